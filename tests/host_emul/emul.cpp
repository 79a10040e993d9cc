// TEST INFRASTRUCTURE.  Compiles the product's device functions
// (smallvcm_amd/csrc/vcm_core.h) for the HOST with g++ and drives them
// serially, so that `-m "not gpu"` tests can compare the product's path logic
// and arithmetic against the oracle without a GPU.  This is not a fallback:
// it is never built into libsmallvcm_amd.so and nothing in the package loads it.
#include <vector>
#include <algorithm>
#include <string.h>
#include "../../smallvcm_amd/csrc/vcm_core.h"
#include "../../smallvcm_amd/csrc/vcm_kat.h"
#include "../../smallvcm_amd/csrc/scene_host.h"

using namespace vcm;

#if defined(VCM_BVH_PROFILE)   /* measurement build (libemul_prof.so, profiles/tools/bvh_sim.py): the traversals' event stream */
static std::vector<char> g_bvhLog;
static std::vector<float> g_bvhRays;   /* org, dir of every logged ray, in log order */
namespace vcm { void vcm_bvh_event(char e) { g_bvhLog.push_back(e); }
                void vcm_bvh_ray(const float *o, const float *d, float t) { for (int k = 0; k < 3; k++) g_bvhRays.push_back(o[k]); for (int k = 0; k < 3; k++) g_bvhRays.push_back(d[k]); g_bvhRays.push_back(t); } }
extern "C" long long emul_bvh_log_size() { return (long long)g_bvhLog.size(); }
extern "C" void emul_bvh_log_get(char *out) { memcpy(out, g_bvhLog.data(), g_bvhLog.size()); }
extern "C" void emul_bvh_log_clear() { g_bvhLog.clear(); g_bvhRays.clear(); }
#define EMUL_PATH_MARK(c) g_bvhLog.push_back(c)
/* A wave of 64 lanes running vcm_core.h's while-while traversal over the logged rays, in lockstep: what one wave-level
 * instruction stream costs against what its lanes needed.  kind: 'B' closest hit, 'b' any hit; `which`: 0 = every ray of
 * that kind in log order, k > 0 = only the k-th closest-hit ray of each path (the wave of bounce k, dead paths replaced by
 * the next path as the refill does).  refill > 0: lanes that finish take the next ray as soon as `refill` lanes are idle
 * (persistent lanes with dynamic fetch), paying cTask wave-instructions per fetch round; 0 = one ray per lane, as built.
 * out: [rays, lane cost, wave cost x 64, mean I, mean L, max I, max L, wave rounds, bound x 64], bound = the wave cost if a wave
 * paid max-over-lanes of each event kind's TOTAL (no alternation between the two halves: what postponing leaves could reach) */
static int g_bvhSortMode = 0;   /* 0: (origin cell, direction octant); 1: (origin cell, end-point cell); 2: (end-point cell, origin cell); 3: origin cell alone */
extern "C" void emul_bvh_sort_mode(int m) { g_bvhSortMode = m; }
static int g_bvhSortCells = 0;   /* > 0: the rays of a replay are first ordered by (origin cell on a grid of that many cells per axis over
                                    [-1.5, 1.5]^3, direction octant) -- what a sort of the shadow-ray tasks before K3b / K3c would do */
extern "C" void emul_bvh_sort(int cells) { g_bvhSortCells = cells; }
extern "C" void emul_bvh_simulate(int kind, int which, int pathKind, int refill, double cI, double cP, double cL, double cTask, double *out)
{
    struct RaySpan { size_t b, e; long long key; };
    std::vector<RaySpan> rays;
    size_t rayNo = 0;
    const size_t n = g_bvhLog.size();
    char curPath = 0; int nthB = 0;
    for (size_t i = 0; i < n; i++) {
        const char c = g_bvhLog[i];
        if (c == 'l' || c == 'c') { curPath = c; nthB = 0; continue; }
        if (c == 'B' || c == 'b') {
            if (c == 'B') nthB++;
            size_t j = i + 1;
            while (j < n && g_bvhLog[j] != 'B' && g_bvhLog[j] != 'b' && g_bvhLog[j] != 'l' && g_bvhLog[j] != 'c') j++;
            const bool take = c == (char)kind && (pathKind == 0 || curPath == (char)pathKind) && (which == 0 || (c == 'B' && nthB == which));
            if (take) {
                long long key = 0;
                if (g_bvhSortCells > 0 && (rayNo + 1) * 7 <= g_bvhRays.size()) {
                    const float *r = &g_bvhRays[rayNo * 7];
                    int c[3];
                    for (int k = 0; k < 3; k++) { int v = (int)((r[k] + 1.5f) / 3.0f * g_bvhSortCells); c[k] = v < 0 ? 0 : (v >= g_bvhSortCells ? g_bvhSortCells - 1 : v); }
                    const int oct = (r[3] < 0.f ? 1 : 0) | (r[4] < 0.f ? 2 : 0) | (r[5] < 0.f ? 4 : 0);
                    key = ((((long long)c[2] * g_bvhSortCells + c[1]) * g_bvhSortCells + c[0]) << 3) | oct;
                    if (g_bvhSortMode > 0) {
                        int e[3];
                        const float t = r[6] > 10.f ? 10.f : r[6];
                        for (int k = 0; k < 3; k++) { int v = (int)((r[k] + r[3 + k] * t + 1.5f) / 3.0f * g_bvhSortCells); e[k] = v < 0 ? 0 : (v >= g_bvhSortCells ? g_bvhSortCells - 1 : v); }
                        const long long ko = ((long long)c[2] * g_bvhSortCells + c[1]) * g_bvhSortCells + c[0], ke = ((long long)e[2] * g_bvhSortCells + e[1]) * g_bvhSortCells + e[0];
                        const long long n3 = (long long)g_bvhSortCells * g_bvhSortCells * g_bvhSortCells;
                        key = g_bvhSortMode == 1 ? ko * n3 + ke : (g_bvhSortMode == 2 ? ke * n3 + ko : ko);
                    }
                }
                rays.push_back(RaySpan{ i + 1, j, key });
            }
            rayNo++;
            i = j - 1;
        }
    }
    if (g_bvhSortCells > 0) std::stable_sort(rays.begin(), rays.end(), [](const RaySpan &a, const RaySpan &b) { return a.key < b.key; });
    double laneCost = 0, waveCost = 0, sumI = 0, sumL = 0, maxI = 0, maxL = 0, rounds = 0;
    for (const RaySpan &r : rays) {
        double nI = 0, nL = 0;
        for (size_t i = r.b; i < r.e; i++) {
            const char c = g_bvhLog[i];
            if (c == 'I') { laneCost += cI; nI++; } else if (c == 'p' || c == 'P') laneCost += cP; else if (c == 'L') { laneCost += cL; nL++; }
        }
        sumI += nI; sumL += nL; maxI = std::max(maxI, nI); maxL = std::max(maxL, nL);
    }
    size_t next = 0;
    while (next < rays.size()) {
        size_t pos[64], end[64]; int live = 0;
        for (int l = 0; l < 64; l++) { if (next < rays.size()) { pos[l] = rays[next].b; end[l] = rays[next].e; next++; live++; } else pos[l] = end[l] = 0; }
        if (refill > 0) { waveCost += cTask; laneCost += cTask * live / 64.0 * 0; }
        auto ev = [&](int l) -> char { return pos[l] < end[l] ? g_bvhLog[pos[l]] : 0; };
        for (;;) {
            bool anyLeft = false;
            for (int l = 0; l < 64; l++) if (pos[l] < end[l]) anyLeft = true;
            if (!anyLeft) break;
            rounds++;
            for (;;) {   /* the inner-node loop */
                bool in[64]; bool any = false;
                for (int l = 0; l < 64; l++) { in[l] = ev(l) == 'I'; if (in[l]) { any = true; pos[l]++; } }
                if (!any) break;
                waveCost += cI;
                for (;;) {
                    bool anyP = false;
                    for (int l = 0; l < 64; l++) if (in[l] && ev(l) == 'p') { anyP = true; pos[l]++; } else in[l] = false;
                    if (!anyP) break;
                    waveCost += cP;
                }
            }
            int mL = 0, mP = 0;
            for (int l = 0; l < 64; l++) { int k = 0; while (ev(l) == 'L') { pos[l]++; k++; } mL = std::max(mL, k); }
            waveCost += mL * cL;
            for (int l = 0; l < 64; l++) { int k = 0; while (ev(l) == 'P') { pos[l]++; k++; } mP = std::max(mP, k); }
            waveCost += mP * cP;
            for (int l = 0; l < 64; l++) if (ev(l) == 'O') pos[l]++;
            if (refill > 0) {   /* dynamic fetch: idle lanes take new rays once enough of them wait */
                int idle = 0;
                for (int l = 0; l < 64; l++) if (pos[l] >= end[l]) idle++;
                if (idle >= refill && next < rays.size()) {
                    for (int l = 0; l < 64; l++) if (pos[l] >= end[l] && next < rays.size()) { pos[l] = rays[next].b; end[l] = rays[next].e; next++; }
                    waveCost += cTask;
                }
            }
        }
    }
    double bound = 0;
    for (size_t b = 0; b < rays.size(); b += 64) {
        double mI = 0, mP = 0, mL = 0;
        for (size_t r = b; r < std::min(rays.size(), b + 64); r++) {
            double nI = 0, nP = 0, nL = 0;
            for (size_t i = rays[r].b; i < rays[r].e; i++) { const char c = g_bvhLog[i]; if (c == 'I') nI++; else if (c == 'p' || c == 'P') nP++; else if (c == 'L') nL++; }
            mI = std::max(mI, nI); mP = std::max(mP, nP); mL = std::max(mL, nL);
        }
        bound += mI * cI + mP * cP + mL * cL;
    }
    out[8] = bound * 64.0;
    out[0] = (double)rays.size(); out[1] = laneCost; out[2] = waveCost * 64.0;
    out[3] = rays.empty() ? 0 : sumI / rays.size(); out[4] = rays.empty() ? 0 : sumL / rays.size(); out[5] = maxI; out[6] = maxL; out[7] = rounds;
}
#else
#define EMUL_PATH_MARK(c) ((void)0)
#endif

/* the ray-casting functions are instantiated per kind of scene (vcm_core.h SceneList / SceneBvh): pick like the
   product's launches do */
template <class F> static void with_scene(const DScene &sc, F &&f)
{
    if (sc.nNodes > 0) f(static_cast<const SceneBvh &>(sc));
    else if (sc.fastOnePlane && sc.nFastRects[0] + sc.nFastRects[1] + sc.nFastRects[2] > 0) f(static_cast<const SceneRects &>(sc));
    else if (sc.fastOnePlane) f(static_cast<const SceneQuads &>(sc));
    else f(static_cast<const SceneList &>(sc));
}

struct Emul {
    SceneHost host;   /* owned scene arrays + packed pairs / BVH */
    DScene sc;        /* what the device functions see: offsets from THIS object into `host` (an Emul never moves) */
    bool useVM, useVC, lightTraceOnly, ppm;
    int renderer;
    float baseRadius, radiusAlpha;
    int seed, iterations;
    int resX, resY, N, p0, nLocal;
    IterParams P;
    std::vector<F4> v0 /* the light store: 4 fields per slot */, g1, g2, camOut;
    std::vector<F2> g3;
    std::vector<float> gx, gy, gz, fb, records;
    std::vector<unsigned char> count, rngL, rngC;
    std::vector<int> cellStart;
    GridHeader hdr;
    LaneStats ls;
};

extern "C" {

static void *emul_finish_create(Emul *e, int algorithm, float radiusFactor, float radiusAlpha, int seed, int rank, int world);
void *emul_create(const vcm_scene_desc *scene, int algorithm, float radiusFactor, float radiusAlpha, int seed,
                  int rank, int world)
{
    Emul *e = new Emul();
    std::string err;
    if (!scene_host_from_desc(*scene, e->host, err)) { delete e; return NULL; }
    return emul_finish_create(e, algorithm, radiusFactor, radiusAlpha, seed, rank, world);
}
void *emul_create2(const vcm_scene_desc2 *scene, int algorithm, float radiusFactor, float radiusAlpha, int seed,
                   int rank, int world)
{
    Emul *e = new Emul();
    std::string err;
    if (!scene_host_from_desc2(*scene, e->host, err)) { delete e; return NULL; }
    return emul_finish_create(e, algorithm, radiusFactor, radiusAlpha, seed, rank, world);
}
static void *emul_finish_create(Emul *e, int algorithm, float radiusFactor, float radiusAlpha, int seed, int rank, int world)
{
    scene_host_build_accel(e->host, scene_host_force_bvh());
    e->host.view(e->sc);
    const SceneHost *scene = &e->host;
    e->useVM = e->useVC = e->lightTraceOnly = e->ppm = false;
    e->renderer = 0;
    switch (algorithm) {
    case VCM_ALGO_LIGHT_TRACE: e->lightTraceOnly = true; break;
    case VCM_ALGO_PPM: e->ppm = true; e->useVM = true; break;
    case VCM_ALGO_BPM: e->useVM = true; break;
    case VCM_ALGO_BPT: e->useVC = true; break;
    case VCM_ALGO_PATH_TRACE: e->renderer = 1; break;
    case VCM_ALGO_EYE_LIGHT: e->renderer = 2; break;
    default: e->useVC = true; e->useVM = true; break;
    }
    if (e->ppm) {
        for (size_t i = 0; i < scene->materials.size(); i++) {
            const vcm_material &m = scene->materials[i];
            if (((vmax3(ld3(m.diffuse)) > 0) || (vmax3(ld3(m.phong)) > 0)) && ((vmax3(ld3(m.mirror)) > 0) || (m.ior > 0))) {
                e->ppm = false; break;
            }
        }
    }
    e->baseRadius = radiusFactor * scene->sceneRadius;
    e->radiusAlpha = radiusAlpha;
    e->seed = seed; e->iterations = 0;
    e->resX = (int)scene->camera.resolution[0]; e->resY = (int)scene->camera.resolution[1];
    e->N = e->resX * e->resY;
    e->p0 = (int)((long long)e->N * rank / world);
    e->nLocal = (int)((long long)e->N * (rank + 1) / world) - e->p0;
    e->fb.assign((size_t)e->N * 3, 0.f);
    return e;
}
void emul_destroy(void *h) { delete (Emul *)h; }

void emul_run_iteration(void *h, int iteration, unsigned minLen, unsigned maxLen)
{
    Emul &e = *(Emul *)h;
    IterParams &P = e.P;
    memset(&P, 0, sizeof(P));
    const int S = (maxLen >= 2) ? (int)maxLen - 1 : 1;
    P.seed = (uint32_t)e.seed; P.localIter = (uint32_t)e.iterations;
    P.minLen = minLen; P.maxLen = maxLen;
    P.resX = e.resX; P.resY = e.resY; P.N = e.N; P.p0 = e.p0; P.nLocal = e.nLocal; P.S = S;
    P.useVM = e.useVM; P.useVC = e.useVC; P.lightTraceOnly = e.lightTraceOnly; P.ppm = e.ppm;
    P.lightSubPathCount = float(e.resX * e.resY);
    float radius = e.baseRadius;
    radius /= dm_powf(float(iteration + 1), 0.5f * (1 - e.radiusAlpha));
    radius = smax(radius, 1e-7f);
    const float radiusSqr = sqr(radius);
    P.radius = radius; P.radiusSqr = radiusSqr;
    P.vmNormalization = 1.f / (radiusSqr * VCM_PI_F * P.lightSubPathCount);
    const float etaVCM = (VCM_PI_F * radiusSqr) * P.lightSubPathCount;
    P.misVmWeightFactor = e.useVM ? mis(etaVCM) : 0.f;
    P.misVcWeightFactor = e.useVC ? mis(1.f / etaVCM) : 0.f;
    P.cellSize = radius * 2.f;
    P.invCellSize = 1.f / P.cellSize;
    P.nCells = e.N;

    P.renderer = e.renderer; P.iteration = iteration;
    P.qblockVertex = P.qblockDI = 512; P.qblockVC = 2048; P.nBuckets = VCM_QSORT_BUCKETS;
    if (e.renderer) {   /* PathTracer / EyeLight: pixel loop, then AddColor in pixel order */
        e.rngL.assign((size_t)e.nLocal, 0); e.rngC.assign((size_t)e.nLocal, 0);
        e.records.clear();
        lane_stats_zero(e.ls);
        e.camOut.assign((size_t)e.nLocal, mk4(0, 0, 0, 0));
        for (int lp = 0; lp < e.nLocal; lp++) {
            if (e.renderer == 1) {
                PtPath path;
                pt_path_begin(e.sc, P, path, lp);
                with_scene(e.sc, [&](const auto &sc) { while (pt_path_step(sc, P, path, e.ls)) {} });
                e.camOut[lp] = mk4(path.color.x, path.color.y, path.color.z, u2f((uint32_t)raster_target(P, path.sx, path.sy)));
                e.rngC[lp] = (unsigned char)path.rng.k;
            } else {
                V3 color = sp3(0.f); float sx, sy; uint32_t drawn;
                bool hit = false;
                with_scene(e.sc, [&](const auto &sc) { hit = eyelight_path(sc, P, lp, color, sx, sy, drawn, e.ls); });
                e.camOut[lp] = mk4(color.x, color.y, color.z, u2f((uint32_t)(hit ? raster_target(P, sx, sy) : -1)));
                e.rngC[lp] = (unsigned char)drawn;
            }
        }
        for (int lp = 0; lp < e.nLocal; lp++) {
            const int t = (int)f2u(e.camOut[lp].w);
            if (t < 0) continue;
            float *px = &e.fb[(size_t)t * 3];
            px[0] = px[0] + e.camOut[lp].x; px[1] = px[1] + e.camOut[lp].y; px[2] = px[2] + e.camOut[lp].z;
        }
        e.iterations++;
        return;
    }
    const size_t slots = (size_t)S * e.nLocal;
    e.v0.assign(slots * VCM_LV_FIELDS, mk4(0, 0, 0, 0));
    e.count.assign((size_t)e.nLocal, 0); e.rngL.assign((size_t)e.nLocal, 0); e.rngC.assign((size_t)e.nLocal, 0);
    lane_stats_zero(e.ls);
    std::vector<uint32_t> lenMask((size_t)e.nLocal, 0u);
    LightStore store; store.v = e.v0.data(); store.count = e.count.data(); store.lenMask = lenMask.data();

    /* K1 */
    for (int lp = 0; lp < e.nLocal; lp++) {
        LightPath path;
        EMUL_PATH_MARK('l');
        light_path_begin(e.sc, P, path, lp);
        LaneBox box; lane_box_init(box);
        with_scene(e.sc, [&](const auto &sc) { while (light_path_step<0>(sc, P, path, store, e.fb.data(), e.ls, box)) {} });
        e.count[lp] = (unsigned char)path.nStored;
        lenMask[lp] = path.lenMask;
        e.rngL[lp] = (unsigned char)path.rng.k;
    }
    /* K1b: records in reference order */
    e.records.clear();
    for (int lp = 0; lp < e.nLocal; lp++)
        for (int j = 0; j < e.count[lp]; j++) {
            const size_t slot = (size_t)j * e.nLocal + lp;
            const F4 a = lv(store, slot, 0), b = lv(store, slot, 1), d = lv(store, slot, 3);
            const F4 w = light_vertex_wdir_contprob(e.sc, a, lv(store, slot, 2), d, false);
            const float r[13] = { a.x, a.y, a.z, w.x, w.y, w.z, b.x, b.y, b.z, b.w, d.w, w.w, u2f(f2u(a.w) & 0xffu) };
            e.records.insert(e.records.end(), r, r + 13);
        }
    const int n = (int)(e.records.size() / 13);
    /* K2: stable counting sort by cell */
    memset(&e.hdr, 0, sizeof(e.hdr));
    e.hdr.nRecords = n;
    for (int c = 0; c < 3; c++) { e.hdr.bboxMin[c] = 1e36f; e.hdr.bboxMax[c] = -1e36f; }
    e.cellStart.assign((size_t)P.nCells + 1, 0);
    e.gx.assign((size_t)n + VCM_MERGE_UNROLL, 0.f); e.gy = e.gx; e.gz = e.gx;
    e.g1.assign((size_t)n + 1, mk4(0, 0, 0, 0)); e.g2 = e.g1; { F2 z; z.x = z.y = 0.f; e.g3.assign((size_t)n + 1, z); }
    if (e.useVM) {
        for (int i = 0; i < n; i++)
            for (int c = 0; c < 3; c++) {
                e.hdr.bboxMax[c] = smax(e.hdr.bboxMax[c], e.records[(size_t)i * 13 + c]);
                e.hdr.bboxMin[c] = smin(e.hdr.bboxMin[c], e.records[(size_t)i * 13 + c]);
            }
        std::vector<int> cell((size_t)n);
        for (int i = 0; i < n; i++) {
            const float *r = &e.records[(size_t)i * 13];
            cell[i] = grid_cell_of_point(mk3(r[0], r[1], r[2]), ld3(e.hdr.bboxMin), P.invCellSize, P.nCells);
            e.cellStart[cell[i] + 1]++;
        }
        for (int c = 0; c < P.nCells; c++) e.cellStart[c + 1] += e.cellStart[c];
        std::vector<int> fill(e.cellStart.begin(), e.cellStart.end() - 1);
        for (int i = 0; i < n; i++) {
            const float *r = &e.records[(size_t)i * 13];
            const int dst = fill[cell[i]]++;
            e.gx[dst] = r[0]; e.gy[dst] = r[1]; e.gz[dst] = r[2];
            e.g1[dst] = mk4(r[3], r[4], r[5], r[11]);
            e.g2[dst] = mk4(r[6], r[7], r[8], r[9]);
            e.g3[dst].x = r[10]; e.g3[dst].y = r[12];
        }
    }
    /* K3 */
    GridStore grid; grid.cellStart = e.cellStart.data(); grid.gx = e.gx.data(); grid.gy = e.gy.data(); grid.gz = e.gz.data(); grid.g1 = e.g1.data();
    grid.g2 = e.g2.data(); grid.g3 = e.g3.data(); grid.hdr = &e.hdr;
    if (!e.lightTraceOnly) {
        e.camOut.assign((size_t)e.nLocal, mk4(0, 0, 0, 0));
        for (int lp = 0; lp < e.nLocal; lp++) {
            CameraPath path;
            uint32_t q[VCM_MERGE_Q + 1];
            MergeScratch ms; ms.q = q; ms.stride = 1; ms.cap = VCM_MERGE_Q;
            EMUL_PATH_MARK('c');
            camera_path_begin(e.sc, P, path, lp);
            VertexStore vs; memset(&vs, 0, sizeof(vs));
            int wqState[6] = {0, 0, 0, 0, 0, 0};
            CameraWaveQueues wqs; wqs.v.p = wqState; wqs.di.p = wqState + 2; wqs.vc.p = wqState + 4; wqs.pendingVertex = -1; wqs.pendingArrival = 0;
            QueryKey qk = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1u, 1u, 0, 0, 0 };
            with_scene(e.sc, [&](const auto &sc) { while (camera_path_step<0>(sc, P, path, store, grid, e.ls, ms, vs, wqs, qk)) {} });
            e.camOut[lp] = mk4(path.color.x, path.color.y, path.color.z, u2f((uint32_t)camera_path_target(P, path)));
            e.rngC[lp] = (unsigned char)path.rng.k;
        }
        /* K5 in path order */
        for (int lp = 0; lp < e.nLocal; lp++) {
            const int t = (int)f2u(e.camOut[lp].w);
            if (t < 0) continue;
            float *px = &e.fb[(size_t)t * 3];
            px[0] = px[0] + e.camOut[lp].x; px[1] = px[1] + e.camOut[lp].y; px[2] = px[2] + e.camOut[lp].z;
        }
    }
    e.iterations++;
}

void emul_get_framebuffer(void *h, float *out) { Emul &e = *(Emul *)h; memcpy(out, e.fb.data(), e.fb.size() * 4); }
void emul_get_counts(void *h, unsigned char *l, unsigned char *c)
{
    Emul &e = *(Emul *)h;
    memcpy(l, e.rngL.data(), e.rngL.size()); memcpy(c, e.rngC.data(), e.rngC.size());
}
long long emul_record_count(void *h) { return (long long)(((Emul *)h)->records.size() / 13); }
void emul_get_records(void *h, float *out) { Emul &e = *(Emul *)h; memcpy(out, e.records.data(), e.records.size() * 4); }
void emul_get_stats(void *h, long long *out9)
{
    const LaneStats &s = ((Emul *)h)->ls;
    const uint32_t v[9] = { s.lightRays, s.cameraRays, s.shadowRays, s.mergeQueries, s.mergeCandidates, s.mergeAccepted,
                            s.connections, s.lightSplats, s.stored };
    for (int i = 0; i < 9; i++) out9[i] = v[i];
}
float emul_sinf(float x) { return dm_sinf(x); }
float emul_cosf(float x) { return dm_cosf(x); }
float emul_powf(float x, float y) { return dm_powf(x, y); }
float emul_path_float(uint32_t seed, uint32_t iter, uint32_t path, uint32_t kind, uint32_t k)
{
    PathRng r; rng_init(r, seed, iter, path, kind);
    float f = 0;
    for (uint32_t i = 0; i <= k; i++) f = rng_float(r);
    return f;
}
/* how often the certified filter handed a ray to the reference loop (per ray here; a GPU wave does it for 64) */
void emul_filter_stats(unsigned long long *out4, int reset)
{
    out4[0] = g_filterStats.isect; out4[1] = g_filterStats.isectExact; out4[2] = g_filterStats.occl; out4[3] = g_filterStats.occlExact;
    if (reset) g_filterStats = FilterStats{ 0, 0, 0, 0 };
}
int emul_scene_cornell(int resX, int resY, unsigned mask, vcm_scene_desc *out);
/* function-level known answers: the product's device functions, one call per record (vcm_kat.h) */
void emul_kat(const vcm_scene_desc *scene, int op, int n, const float *in, float *out)
{
    SceneHost h;
    std::string err;
    if (!scene_host_from_desc(*scene, h, err)) return;
    scene_host_build_accel(h, scene_host_force_bvh());
    DScene view;
    h.view(view);
    with_scene(view, [&](const auto &sc) {
        for (int i = 0; i < n; i++) kat_eval(sc, op, in + (size_t)i * VCM_KAT_FLOATS, out + (size_t)i * VCM_KAT_FLOATS);
    });
}
/* the same over a version-2 scene (BVH for more than VCM_MAX_PRIMS primitives) */
void emul_kat2(const vcm_scene_desc2 *scene, int op, int n, const float *in, float *out)
{
    SceneHost h;
    std::string err;
    if (!scene_host_from_desc2(*scene, h, err)) return;
    scene_host_build_accel(h, scene_host_force_bvh());
    DScene view;
    h.view(view);
    with_scene(view, [&](const auto &sc) {
        for (int i = 0; i < n; i++) kat_eval(sc, op, in + (size_t)i * VCM_KAT_FLOATS, out + (size_t)i * VCM_KAT_FLOATS);
    });
}

} // extern "C"
