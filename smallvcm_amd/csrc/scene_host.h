// scene_host.h -- host side of the scene: copies a C-ABI scene description (vcm_scene_desc: the reference's built-in
// boxes, fixed capacities; vcm_scene_desc2: any counts) into owned arrays and builds what the intersection code walks:
//   * <= VCM_MAX_PRIMS primitives: GeometryList order, consecutive triangles packed in pairs (vcm_core.h TriPair) --
//     the brute-force loop of Scene::Intersect (scene.hxx:53-70), which is what the reference does for every scene;
//   * more (or SMALLVCM_AMD_FORCE_BVH=1): a binary BVH over the primitives' boxes, SAH-binned over the three axes, at most 2 primitives per
//     leaf, nodes in depth-first order with escape indices (stackless traversal, vcm_core.h bvh_intersect).  The
//     reference has no acceleration structure (README:208-209); results are the same because the traversal only decides
//     WHICH primitives are tested, never how (vcm_core.h explains the tie rule).
// Host only (std::vector); vcm_api.hip uploads the arrays and hands the kernels a DScene of device pointers,
// tests/host_emul walks the same structure on the CPU.
#ifndef SMALLVCM_AMD_SCENE_HOST_H
#define SMALLVCM_AMD_SCENE_HOST_H

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vcm_core.h"

namespace vcm {

struct SceneHost {
    std::vector<vcm_prim> prims;
    std::vector<vcm_material> materials;
    std::vector<int> mat2light;
    std::vector<vcm_light> lights;
    int backgroundLight;
    float sceneCenter[3], sceneRadius, invSceneRadiusSqr;
    vcm_camera camera;
    std::vector<PrimOp> ops;
    std::vector<TriPair> pairs;
    std::vector<FastPair> fastPairs;      /* the filter's view of the list (vcm_core.h): triangles two by two, */
    std::vector<FastSphere> fastSpheres;  /* and the spheres */
    std::vector<FastRect> fastRects;      /* the same pairs as axis-aligned rectangles, grouped by normal axis -- if ALL of them are */
    int nFastRects[3]; float fastGmax;
    float fastRw2, fastCenter[3], fastRadius;
    std::vector<BvhNode> nodes;
    std::vector<BvhWide> wide;            /* one per inner node: both children's boxes (the ordered traversals, vcm_core.h) */
    std::vector<int> leafPrims;
    std::vector<LeafPrim> leafData;       /* the leaves' primitives in leaf order (what the traversals read) */

    /* the view the device functions take, filled IN PLACE: the arrays are addressed relative to the DScene object
       itself (vcm_core.h), so `d` must stay where it is while it is in use (host emulation) */
    void view(DScene &d) const
    {
        fill_scalars(d);
        const char *base = reinterpret_cast<const char *>(&d);
        d.offPrims = (const char *)prims.data() - base; d.offMaterials = (const char *)materials.data() - base;
        d.offMat2light = (const char *)mat2light.data() - base; d.offLights = (const char *)lights.data() - base;
        d.offOps = (const char *)ops.data() - base; d.offPairs = (const char *)pairs.data() - base;
        d.offNodes = (const char *)nodes.data() - base; d.offLeafPrims = (const char *)leafPrims.data() - base;
        d.offFastPairs = (const char *)fastPairs.data() - base; d.offFastSpheres = (const char *)fastSpheres.data() - base;
        d.offWide = (const char *)wide.data() - base; d.offLeafData = (const char *)leafData.data() - base;
        d.offFastRects = (const char *)fastRects.data() - base;
    }
    void fill_scalars(DScene &d) const
    {
        d.nPrims = (int)prims.size(); d.nMaterials = (int)materials.size(); d.nLights = (int)lights.size();
        d.backgroundLight = backgroundLight;
        for (int k = 0; k < 3; k++) d.sceneCenter[k] = sceneCenter[k];
        d.sceneRadius = sceneRadius; d.invSceneRadiusSqr = invSceneRadiusSqr;
        d.camera = camera;
        d.nOps = (int)ops.size(); d.nNodes = (int)nodes.size();
        d.fastRw2 = fastRw2; d.fastRadius = fastRadius;
        d.nFastPairs = (int)fastPairs.size(); d.nFastSpheres = (int)fastSpheres.size();
        d.fastOnePlane = 1;
        for (const FastPair &f : fastPairs) if (!(f.flags & 4)) d.fastOnePlane = 0;
        { const char *e = getenv("SMALLVCM_AMD_NO_ONEPLANE"); if (e && e[0] == '1') d.fastOnePlane = 0; }   /* measurement switch */
        for (int k = 0; k < 3; k++) d.fastCenter[k] = fastCenter[k];
        for (int k = 0; k < 3; k++) d.nFastRects[k] = nFastRects[k];
        d.fastGmax = fastGmax;
        { const char *e = getenv("SMALLVCM_AMD_NO_RECTS"); if (e && e[0] == '1') d.nFastRects[0] = d.nFastRects[1] = d.nFastRects[2] = 0; }   /* measurement switch */
    }
};

inline bool scene_host_check(const SceneHost &s, std::string &err)
{
    if (s.lights.empty()) { err = "scene has no light"; return false; }
    if (s.materials.empty()) { err = "scene has no material"; return false; }
    if (s.materials.size() >= (1u << 24)) { err = "more than 2^24 materials"; return false; }
    for (const vcm_prim &p : s.prims)
        if (p.matID < 0 || p.matID >= (int)s.materials.size()) { err = "primitive with a material index out of range"; return false; }
    for (int l : s.mat2light)
        if (l >= (int)s.lights.size()) { err = "mat2light entry out of range"; return false; }
    if (s.backgroundLight >= (int)s.lights.size()) { err = "backgroundLight out of range"; return false; }
    return true;
}

inline bool scene_host_from_desc(const vcm_scene_desc &sc, SceneHost &s, std::string &err)
{
    if (sc.nPrims < 0 || sc.nPrims > VCM_MAX_PRIMS || sc.nMaterials < 0 || sc.nMaterials > VCM_MAX_MATERIALS || sc.nLights < 1 ||
        sc.nLights > VCM_MAX_LIGHTS) { err = "scene exceeds the fixed capacities of vcm_scene_desc (use vcm_scene_desc2)"; return false; }
    s.prims.assign(sc.prims, sc.prims + sc.nPrims);
    s.materials.assign(sc.materials, sc.materials + sc.nMaterials);
    s.mat2light.assign(sc.mat2light, sc.mat2light + sc.nMaterials);
    s.lights.assign(sc.lights, sc.lights + sc.nLights);
    s.backgroundLight = sc.backgroundLight;
    for (int k = 0; k < 3; k++) s.sceneCenter[k] = sc.sceneCenter[k];
    s.sceneRadius = sc.sceneRadius; s.invSceneRadiusSqr = sc.invSceneRadiusSqr;
    s.camera = sc.camera;
    return scene_host_check(s, err);
}

inline bool scene_host_from_desc2(const vcm_scene_desc2 &sc, SceneHost &s, std::string &err)
{
    if (sc.nPrims < 0 || sc.nMaterials < 1 || sc.nLights < 1 || (sc.nPrims > 0 && !sc.prims) || !sc.materials || !sc.mat2light || !sc.lights) {
        err = "vcm_scene_desc2: bad counts or NULL arrays"; return false;
    }
    s.prims.assign(sc.prims, sc.prims + sc.nPrims);
    s.materials.assign(sc.materials, sc.materials + sc.nMaterials);
    s.mat2light.assign(sc.mat2light, sc.mat2light + sc.nMaterials);
    s.lights.assign(sc.lights, sc.lights + sc.nLights);
    s.backgroundLight = sc.backgroundLight;
    for (int k = 0; k < 3; k++) s.sceneCenter[k] = sc.sceneCenter[k];
    s.sceneRadius = sc.sceneRadius; s.invSceneRadiusSqr = sc.invSceneRadiusSqr;
    s.camera = sc.camera;
    return scene_host_check(s, err);
}

/* ---- brute-force list: consecutive triangles in pairs, fields interleaved (vcm_core.h TriPair) ---- */
inline void scene_host_build_pairs(SceneHost &s)
{
    s.ops.clear(); s.pairs.clear();
    const int n = (int)s.prims.size();
    for (int i = 0; i < n; ) {
        PrimOp op;
        if (s.prims[i].type != VCM_PRIM_TRIANGLE) { op.kind = 1; op.index = i; s.ops.push_back(op); i++; continue; }
        op.kind = 0; op.index = (int)s.pairs.size();
        s.ops.push_back(op);
        TriPair tp;
        std::memset(&tp, 0, sizeof(tp));
        const bool two = (i + 1 < n) && s.prims[i + 1].type == VCM_PRIM_TRIANGLE;
        for (int h = 0; h < 2; h++) {
            const vcm_prim &t = s.prims[(h == 1 && two) ? i + 1 : i];
            tp.p0x[h] = t.p0[0]; tp.p0y[h] = t.p0[1]; tp.p0z[h] = t.p0[2];
            tp.p1x[h] = t.p1[0]; tp.p1y[h] = t.p1[1]; tp.p1z[h] = t.p1[2];
            tp.p2x[h] = t.p2[0]; tp.p2y[h] = t.p2[1]; tp.p2z[h] = t.p2[2];
            tp.nx[h] = t.n[0]; tp.ny[h] = t.n[1]; tp.nz[h] = t.n[2];
            tp.matID[h] = t.matID;
            tp.prim[h] = (h == 1 && two) ? i + 1 : i;
        }
        tp.valid1 = two ? 1 : 0;
        s.pairs.push_back(tp);
        i += two ? 2 : 1;
    }
}

/* ---- the pairs as axis-aligned rectangles (vcm_core.h FastRect), if EVERY pair is one: the reference's own boxes ---- */
inline bool scene_host_rect_triangle(const vcm_prim &t, int k, const float *(&diag)[2], float c[2], float g[2])
{   /* the reference's edge functions (geometry.hxx:133-139): V(c, b), V(b, a), V(a, c); one must be the diagonal, one
       run along u = k+1 (constant v), one along v = k+2 (constant u) */
    const int u = (k + 1) % 3, v = (k + 2) % 3;
    const float *e[3][2] = { { t.p2, t.p1 }, { t.p1, t.p0 }, { t.p0, t.p2 } };
    bool haveU = false, haveV = false, haveD = false;
    for (int i = 0; i < 3; i++) {
        const float *P = e[i][0], *Q = e[i][1];
        const bool sameU = P[u] == Q[u], sameV = P[v] == Q[v];
        if (sameV && !sameU) { if (haveU) return false; haveU = true; c[0] = P[v]; g[0] = (float)((double)P[u] - (double)Q[u]); }        /* along u: V = d_k (P_u - Q_u)(c - X_v) */
        else if (sameU && !sameV) { if (haveV) return false; haveV = true; c[1] = P[u]; g[1] = (float)((double)Q[v] - (double)P[v]); }   /* along v: V = d_k (Q_v - P_v)(c - X_u) */
        else if (!sameU && !sameV) { if (haveD) return false; haveD = true; diag[0] = P; diag[1] = Q; }
        else return false;   /* a degenerate edge */
    }
    return haveU && haveV && haveD;
}
inline void scene_host_build_rects(SceneHost &s)
{
    s.fastRects.clear();
    s.nFastRects[0] = s.nFastRects[1] = s.nFastRects[2] = 0; s.fastGmax = 0.f;
    std::vector<FastRect> byAxis[3];
    for (const FastPair &f : s.fastPairs) {
        if ((f.flags & 7) != 7) return;   /* two triangles, shared diagonal, one plane */
        const vcm_prim &t = s.prims[(size_t)f.prim[0]], &w = s.prims[(size_t)f.prim[1]];
        int k = -1;
        for (int a = 0; a < 3; a++) if (std::fabs(t.n[a]) == 1.f && t.n[(a + 1) % 3] == 0.f && t.n[(a + 2) % 3] == 0.f) k = a;
        if (k < 0) return;
        const float *vs[6] = { t.p0, t.p1, t.p2, w.p0, w.p1, w.p2 };
        for (int i = 1; i < 6; i++) if (vs[i][k] != vs[0][k]) return;
        if (w.n[k] != t.n[k]) return;
        FastRect r;
        std::memset(&r, 0, sizeof(r));
        const float *da[2], *db[2];
        if (!scene_host_rect_triangle(t, k, da, r.c, r.g) || !scene_host_rect_triangle(w, k, db, r.c + 2, r.g + 2)) return;
        /* B's diagonal must be A's, reversed: then B's third edge function is exactly minus A's */
        bool rev = true;
        for (int a = 0; a < 3; a++) rev = rev && da[0][a] == db[1][a] && da[1][a] == db[0][a];
        if (!rev) return;
        {   /* V(P, Q) = Dot(dir, P x Q) + Dot(o x dir, Q - P), binary64, rounded once (as scene_host_build_fast) */
            const double p[3] = { da[0][0], da[0][1], da[0][2] }, q[3] = { da[1][0], da[1][1], da[1][2] };
            r.NEd[0] = (float)(p[1] * q[2] - p[2] * q[1]); r.NEd[1] = (float)(p[2] * q[0] - p[0] * q[2]); r.NEd[2] = (float)(p[0] * q[1] - p[1] * q[0]);
            for (int a = 0; a < 3; a++) r.NEd[3 + a] = (float)(q[a] - p[a]);
        }
        r.pk = t.p0[k]; r.nk = t.n[k];
        r.prim[0] = f.prim[0]; r.prim[1] = f.prim[1];
        for (int a = 0; a < 4; a++) s.fastGmax = std::max(s.fastGmax, std::fabs(r.g[a]) * 1.000001f);
        byAxis[k].push_back(r);
    }
    if (s.fastPairs.empty()) return;
    for (int k = 0; k < 3; k++) { s.nFastRects[k] = (int)byAxis[k].size(); s.fastRects.insert(s.fastRects.end(), byAxis[k].begin(), byAxis[k].end()); }
}

/* ---- the filter's view of the list (vcm_core.h "certified filters"): planes and Pluecker edge data, computed in
 *      binary64 and rounded once; the two triangles of a quad share their diagonal, which the second one reuses ---- */
inline void scene_host_build_fast(SceneHost &s)
{
    const int n = (int)s.prims.size();
    s.fastPairs.clear(); s.fastSpheres.clear();
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 }, rw2 = 0.0;
    auto edge = [](const float *P, const float *Q, float *NE) {   /* V(P, Q) = Dot(dir, P x Q) + Dot(o x dir, Q - P) */
        const double p[3] = { P[0], P[1], P[2] }, q[3] = { Q[0], Q[1], Q[2] };
        NE[0] = (float)(p[1] * q[2] - p[2] * q[1]); NE[1] = (float)(p[2] * q[0] - p[0] * q[2]); NE[2] = (float)(p[0] * q[1] - p[1] * q[0]);
        for (int k = 0; k < 3; k++) NE[3 + k] = (float)(q[k] - p[k]);
    };
    auto same = [](const float *a, const float *b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; };
    for (int i = 0; i < n; ) {
        const vcm_prim &t = s.prims[i];
        if (t.type != VCM_PRIM_TRIANGLE) {
            FastSphere f;
            std::memset(&f, 0, sizeof(f));
            for (int k = 0; k < 3; k++) f.c[k] = t.p0[k];
            f.radius = t.p1[0]; f.prim = i;
            s.fastSpheres.push_back(f);
            i++;
            continue;
        }
        const bool two = (i + 1 < n) && s.prims[i + 1].type == VCM_PRIM_TRIANGLE;
        const vcm_prim &u = s.prims[two ? i + 1 : i];
        FastPair f;
        std::memset(&f, 0, sizeof(f));
        for (int k = 0; k < 3; k++) { f.p0[0][k] = t.p0[k]; f.n[0][k] = t.n[k]; f.p0[1][k] = u.p0[k]; f.n[1][k] = u.n[k]; }
        f.prim[0] = i; f.prim[1] = two ? i + 1 : i;
        f.flags = two ? 1 : 0;
        /* the reference's three edge functions (geometry.hxx:133-139): v0 = V(c, b), v1 = V(b, a), v2 = V(a, c);
           the sign test is symmetric in them, so each triangle may list them in any order */
        const float *te[3][2] = { { t.p2, t.p1 }, { t.p1, t.p0 }, { t.p0, t.p2 } };
        const float *ue[3][2] = { { u.p2, u.p1 }, { u.p1, u.p0 }, { u.p0, u.p2 } };
        int ta = 2, ub = 2;
        bool shared = false;
        if (two)
            for (int a = 0; a < 3 && !shared; a++)
                for (int b = 0; b < 3 && !shared; b++)
                    if (same(te[a][0], ue[b][1]) && same(te[a][1], ue[b][0])) { ta = a; ub = b; shared = true; }
        const int to[3] = { (ta + 1) % 3, (ta + 2) % 3, ta }, uo[3] = { (ub + 1) % 3, (ub + 2) % 3, ub };
        for (int k = 0; k < 3; k++) edge(te[to[k]][0], te[to[k]][1], f.NE[k]);
        edge(ue[uo[0]][0], ue[uo[0]][1], f.NE[3]);
        edge(ue[uo[1]][0], ue[uo[1]][1], f.NE[4]);
        edge(ue[uo[2]][0], ue[uo[2]][1], f.NE[5]);
        if (shared) f.flags |= 2;
        if (two) {   /* one plane part for both? (FastPair::flags bit 2) */
            bool same = true;
            for (int k = 0; k < 3; k++) same = same && (t.n[k] == u.n[k]) && (t.n[k] == 0.f || t.p0[k] == u.p0[k]);
            if (same) f.flags |= 4;
        }
        s.fastPairs.push_back(f);
        for (int w = 0; w < (two ? 2 : 1); w++) {
            const vcm_prim &q = s.prims[i + w];
            const float *vs[3] = { q.p0, q.p1, q.p2 };
            for (int v = 0; v < 3; v++) {
                double l2 = 0.0;
                for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], (double)vs[v][k]); hi[k] = std::max(hi[k], (double)vs[v][k]); l2 += (double)vs[v][k] * vs[v][k]; }
                rw2 = std::max(rw2, l2);
            }
        }
        i += two ? 2 : 1;
    }
    double c[3] = { 0, 0, 0 }, r2 = 0.0;
    if (lo[0] <= hi[0]) for (int k = 0; k < 3; k++) c[k] = 0.5 * (lo[k] + hi[k]);
    for (int i = 0; i < n; i++) {
        const vcm_prim &t = s.prims[i];
        if (t.type != VCM_PRIM_TRIANGLE) continue;
        const float *vs[3] = { t.p0, t.p1, t.p2 };
        for (int v = 0; v < 3; v++) {
            double d2 = 0.0;
            for (int k = 0; k < 3; k++) d2 += (vs[v][k] - c[k]) * (vs[v][k] - c[k]);
            r2 = std::max(r2, d2);
        }
    }
    scene_host_build_rects(s);
    /* rounded up: they enter error BOUNDS */
    s.fastRw2 = (float)(rw2 * 1.0001) + 1e-30f;
    s.fastRadius = (float)(std::sqrt(r2) * 1.0001) + 1e-30f;
    for (int k = 0; k < 3; k++) s.fastCenter[k] = (float)c[k];
}

/* ---- BVH ---- */
/* primitives per leaf.  A wave pays for its slowest lane in both halves of the traversal, and with up to four primitives per
   leaf the leaf half was the longer one: replayed as waves of 64 rays over the mesh scene (profiles/tools/bvh_sim.py) a
   closest-hit ray costs 9.0 inner steps + 2.7 triangle tests with two, 7.4 + 7.2 with four -- a fifth fewer
   wave-instructions at the same lane utilisation.  (The descriptor holds up to 15: coincident centroids stay together.) */
#ifndef VCM_BVH_LEAF_MAX
#define VCM_BVH_LEAF_MAX 2
#endif
struct BvhBuildPrim { float lo[3], hi[3], c[3]; int index; };

inline void bvh_prim_box(const vcm_prim &p, float lo[3], float hi[3])
{
    if (p.type == VCM_PRIM_TRIANGLE) {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(p.p0[k], std::min(p.p1[k], p.p2[k]));
            hi[k] = std::max(p.p0[k], std::max(p.p1[k], p.p2[k]));
        }
    } else {   /* sphere: p0 = centre, p1[0] = radius */
        for (int k = 0; k < 3; k++) { lo[k] = p.p0[k] - p.p1[0]; hi[k] = p.p0[k] + p.p1[0]; }
    }
}

inline int bvh_build_node(SceneHost &s, std::vector<BvhBuildPrim> &bp, int first, int count, float pad)
{
    const int me = (int)s.nodes.size();
    s.nodes.push_back(BvhNode());
    float lo[3] = { 1e36f, 1e36f, 1e36f }, hi[3] = { -1e36f, -1e36f, -1e36f }, clo[3] = { 1e36f, 1e36f, 1e36f }, chi[3] = { -1e36f, -1e36f, -1e36f };
    for (int i = first; i < first + count; i++)
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], bp[i].lo[k]); hi[k] = std::max(hi[k], bp[i].hi[k]);
            clo[k] = std::min(clo[k], bp[i].c[k]); chi[k] = std::max(chi[k], bp[i].c[k]);
        }
    /* grown: the traversal's slab test and the primitives' own hit computations round (relative 1e-6); a box that is
       larger by 1e-4 of the scene can only add visits */
    for (int k = 0; k < 3; k++) { s.nodes[me].bmin[k] = lo[k] - pad; s.nodes[me].bmax[k] = hi[k] + pad; }
    int axis = 0;
    for (int k = 1; k < 3; k++) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
    const bool flat = !(chi[axis] - clo[axis] > 0.f);
    if (count <= VCM_BVH_LEAF_MAX || (flat && count <= 15)) {
        s.nodes[me].leaf = ((int)s.leafPrims.size() << 4) | count;
        for (int i = first; i < first + count; i++) s.leafPrims.push_back(bp[i].index);
        s.nodes[me].escape = (int)s.nodes.size();
        return me;
    }
    /* binned surface-area heuristic over all three axes (16 bins each; the axis of the largest centroid extent alone --
       rounds 2-3 -- left the room's large wall triangles in the floor mesh's subtrees); median split as the fallback */
    int mid = first + count / 2;
    if (!flat) {
        const int B = 16;
        auto area = [](const float *l, const float *h) { const float x = h[0] - l[0], y = h[1] - l[1], z = h[2] - l[2]; return x * y + y * z + z * x; };
        float best = 1e36f; int bestSplit = -1, bestAxis = axis;
        for (int ax = 0; ax < 3; ax++) {
            if (!(chi[ax] - clo[ax] > 0.f)) continue;
            float blo[B][3], bhi[B][3]; int bn[B];
            for (int b = 0; b < B; b++) { bn[b] = 0; for (int k = 0; k < 3; k++) { blo[b][k] = 1e36f; bhi[b][k] = -1e36f; } }
            const float scale = (float)B / (chi[ax] - clo[ax]);
            for (int i = first; i < first + count; i++) {
                int b = (int)((bp[i].c[ax] - clo[ax]) * scale); b = b < 0 ? 0 : (b >= B ? B - 1 : b);
                bn[b]++;
                for (int k = 0; k < 3; k++) { blo[b][k] = std::min(blo[b][k], bp[i].lo[k]); bhi[b][k] = std::max(bhi[b][k], bp[i].hi[k]); }
            }
            for (int sp = 1; sp < B; sp++) {
                float l0[3] = { 1e36f, 1e36f, 1e36f }, h0[3] = { -1e36f, -1e36f, -1e36f }, l1[3] = { 1e36f, 1e36f, 1e36f }, h1[3] = { -1e36f, -1e36f, -1e36f };
                int n0 = 0, n1 = 0;
                for (int b = 0; b < B; b++) {
                    if (!bn[b]) continue;
                    float *l = b < sp ? l0 : l1, *h = b < sp ? h0 : h1;
                    (b < sp ? n0 : n1) += bn[b];
                    for (int k = 0; k < 3; k++) { l[k] = std::min(l[k], blo[b][k]); h[k] = std::max(h[k], bhi[b][k]); }
                }
                if (!n0 || !n1) continue;
                const float cost = area(l0, h0) * n0 + area(l1, h1) * n1;
                if (cost < best) { best = cost; bestSplit = sp; bestAxis = ax; }
            }
        }
        if (bestSplit > 0) {
            axis = bestAxis;
            const float scale = (float)B / (chi[axis] - clo[axis]);
            auto bin_of = [&](const BvhBuildPrim &p) { int b = (int)((p.c[axis] - clo[axis]) * scale); return b < 0 ? 0 : (b >= B ? B - 1 : b); };
            auto it = std::stable_partition(bp.begin() + first, bp.begin() + first + count, [&](const BvhBuildPrim &p) { return bin_of(p) < bestSplit; });
            mid = (int)(it - bp.begin());
        }
    }
    if (mid <= first || mid >= first + count) {   /* degenerate: all centroids in one bin */
        std::stable_sort(bp.begin() + first, bp.begin() + first + count, [&](const BvhBuildPrim &a, const BvhBuildPrim &b) { return a.c[axis] < b.c[axis]; });
        mid = first + count / 2;
    }
    s.nodes[me].leaf = -1;
    bvh_build_node(s, bp, first, mid - first, pad);
    bvh_build_node(s, bp, mid, first + count - mid, pad);
    s.nodes[me].escape = (int)s.nodes.size();
    return me;
}

inline void scene_host_build_bvh(SceneHost &s)
{
    s.nodes.clear(); s.leafPrims.clear();
    const int n = (int)s.prims.size();
    if (n == 0) return;
    std::vector<BvhBuildPrim> bp((size_t)n);
    float lo[3] = { 1e36f, 1e36f, 1e36f }, hi[3] = { -1e36f, -1e36f, -1e36f };
    for (int i = 0; i < n; i++) {
        bvh_prim_box(s.prims[i], bp[i].lo, bp[i].hi);
        bp[i].index = i;
        for (int k = 0; k < 3; k++) {
            bp[i].c[k] = 0.5f * (bp[i].lo[k] + bp[i].hi[k]);
            lo[k] = std::min(lo[k], bp[i].lo[k]); hi[k] = std::max(hi[k], bp[i].hi[k]);
        }
    }
    float extent = 0.f;
    for (int k = 0; k < 3; k++) extent = std::max(extent, std::max(hi[k] - lo[k], std::max(std::fabs(lo[k]), std::fabs(hi[k]))));
    const float pad = 1e-4f * extent + 1e-30f;
    s.nodes.reserve((size_t)n);
    bvh_build_node(s, bp, 0, n, pad);
    /* the wide view: inner nodes numbered in depth-first order, each holding the boxes of its two children -- the left
       child is the next node in memory, the right one the left's escape */
    s.wide.clear();
    for (size_t i = 0; i < s.nodes.size(); i++) if (s.nodes[i].leaf < 0) { s.nodes[i].leaf = -1 - (int)s.wide.size(); s.wide.push_back(BvhWide()); }
    for (size_t i = 0; i < s.nodes.size(); i++) {
        if (s.nodes[i].leaf >= 0) continue;
        BvhWide &w = s.wide[(size_t)(-1 - s.nodes[i].leaf)];
        const int l = (int)i + 1, r = s.nodes[(size_t)l].escape;
        const BvhNode &nl = s.nodes[(size_t)l], &nr = s.nodes[(size_t)r];
        for (int k = 0; k < 3; k++) { w.lmin[k] = nl.bmin[k]; w.lmax[k] = nl.bmax[k]; w.rmin[k] = nr.bmin[k]; w.rmax[k] = nr.bmax[k]; }
        w.lnode = l; w.lref = nl.leaf; w.rnode = r; w.rref = nr.leaf;
    }
    s.leafData.resize(s.leafPrims.size());
    for (size_t i = 0; i < s.leafPrims.size(); i++) { s.leafData[i].prim = s.prims[(size_t)s.leafPrims[i]]; s.leafData[i].index = s.leafPrims[i]; s.leafData[i].pad = 0; }
}

/* what the intersection code walks: the packed list for the reference's own scenes, the BVH beyond */
inline void scene_host_build_accel(SceneHost &s, bool forceBvh)
{
    s.ops.clear(); s.pairs.clear(); s.nodes.clear(); s.wide.clear(); s.leafPrims.clear(); s.leafData.clear(); s.fastPairs.clear(); s.fastSpheres.clear(); s.fastRects.clear();
    s.nFastRects[0] = s.nFastRects[1] = s.nFastRects[2] = 0; s.fastGmax = 0.f;
    s.fastRw2 = s.fastRadius = 0.f; s.fastCenter[0] = s.fastCenter[1] = s.fastCenter[2] = 0.f;
    if ((int)s.prims.size() > VCM_MAX_PRIMS || (forceBvh && !s.prims.empty())) scene_host_build_bvh(s);
    else { scene_host_build_pairs(s); scene_host_build_fast(s); }
}
inline bool scene_host_force_bvh()
{
    const char *e = getenv("SMALLVCM_AMD_FORCE_BVH");
    return e && e[0] == '1';
}

} // namespace vcm
#endif
