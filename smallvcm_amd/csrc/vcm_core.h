// vcm_core.h -- device functions of the VCM hot path (host+device so that the
// unit tests can also drive them on the CPU; the shipped library only ever
// runs them inside the HIP kernels of vcm_kernels.h).
//
// One lane owns one sub-path.  Every function states which reference code it
// replaces; expression order is the reference's (see vcm_math.h).
//
//   geometry     src/geometry.hxx:65-237, src/scene.hxx:53-102
//   BSDF         src/bsdf.hxx:95-566
//   lights       src/lights.hxx:112-514
//   samplers     src/utils.hxx:36-259
//   integrator   src/vertexcm.hxx:284-1006; src/pathtracer.hxx:45-215; src/eyelight.hxx:46-77
//   hash grid    src/hashgrid.hxx:110-201 (query side; build is in the kernels)
#ifndef SMALLVCM_AMD_VCM_CORE_H
#define SMALLVCM_AMD_VCM_CORE_H

#include "../../include/smallvcm_amd.h"
#include "vcm_math.h"
#include "detmath.h"
#include "philox.h"

namespace vcm {

/* ------------------------------------------------------------------ */
/* per-iteration constants (vertexcm.hxx:288-308) + launch geometry     */
struct IterParams {
    uint32_t seed, localIter;
    uint32_t minLen, maxLen;
    int   resX, resY, N;
    int   p0, nLocal;          /* local path range [p0, p0+nLocal) */
    int   S;                   /* light-vertex slots per path = max(1, maxLen-1) */
    int   useVM, useVC, lightTraceOnly, ppm;
    float radius, radiusSqr, vmNormalization, misVmWeightFactor, misVcWeightFactor;
    float lightSubPathCount;
    float cellSize, invCellSize;   /* hashgrid.hxx:47-48 */
    int   nCells;                  /* = N (vertexcm.hxx:406) */
    int   wavefront;               /* 1: DI / VC / merge deferred to task kernels (default) */
    int   renderer;                /* 0: VertexCM family; 1: PathTracer (pathtracer.hxx); 2: EyeLight (eyelight.hxx) */
    int   iteration;               /* aIteration as passed to RunIteration (EyeLight reads it, eyelight.hxx:61) */
    int   qblockVertex, qblockDI, qblockVC;   /* slots a wave of K3 reserves per atomic (wave_queue_alloc) */
    int   nBuckets;                /* entries of the query-sort bucket table in use (<= VCM_QSORT_BUCKETS) */
    int   foreignOut;              /* 1: the merge kernels evaluate queries another rank sent (merge sharded by space): a query's term goes to
                                      mergeOut[its position in the query array], not to its path slot */
};

/* device-resident hash-grid header: bbox is reduced on the device */
struct GridHeader {
    uint32_t bboxMinU[3], bboxMaxU[3];   /* order-preserving uint encoding during the reduce */
    float bboxMin[3], bboxMax[3];
    int   nRecords;                      /* vertices in the grid */
    int   nLocalRecords;                 /* vertices stored by this rank */
    int   pad[2];
};

/* Light-vertex store: vertex j of local path lp lives in slot [j * nLocal + lp], a record of FOUR 16-byte fields = 64
 * bytes, 64-byte aligned: whoever gathers a vertex (vertex connection K3c, camera connection K1c, the cell-sorted copy of
 * the grid build) moves ONE 128-byte line for it.  Rounds 1-3 kept a fifth field (WorldDirFix | ContinuationProb, what only
 * the merge records want): an 80-byte record lies in 1.5 lines, and the grid build, which wants fields 0, 1, 3 and 4, read
 * two.  Round 4 rebuilds the fifth where the records are made (light_vertex_wdir_contprob): ~200 operations per vertex in
 * kernels that wait for memory, against 0.14 GB less written by K1 and ~3 GB less gathered per 2048^2 iteration.
 * History: the record was five 16-byte fields (80 B).  K1 writes slot-major, so a wave (64 consecutive paths at the same bounce) fills one
 * contiguous 5 KB region; the consumers GATHER a vertex (vertex connection K3c, the camera connection K1c, the
 * cell-sorted copy of the grid build) and a gather moves whole 128-byte lines (profiles/archive/r05a_fetch_calib.json): an
 * 80-byte record lies in 1.5 of them on average.  Round 3 measured the alternative -- the four fields the connections
 * read as a 64-byte aligned record (one line), the fifth in an array of its own: the connections
 * gain, the grid build (which wants fields 0, 1, 3 AND 4: two lines) loses and slows the camera pass it runs next to:
 * 868 against 875 Mpaths/s (profiles/archive/r05h_ab_summary.txt).  The camera-vertex records, below, ARE split.
 * (Five separate arrays were measured in round 1: five lines per gather, k_cell_rank_gather 1.16 ms instead of 0.6.)
 * Replaces the AoS std::vector<LightVertex> (vertexcm.hxx:79-101, 120 B/vertex). */
#define VCM_LV_FIELDS 4
struct LightStore {
    F4 *v;    /* [slot * 5 + k]:
                 k=0 hitpoint.xyz | pathLength (bits 0-7) , matID (bits 8-15)
                 k=1 throughput.xyz | dVCM
                 k=2 isect.normal.xyz | dVC
                 k=3 localDirFix.xyz | dVM                                    */
    unsigned char *count;   /* stored vertices per local path (mPathEnds, :395) */
    uint32_t *lenMask;      /* per local path: bit L set <=> a vertex with pathLength L is stored (stored vertices have
                               increasing pathLength, so vertex j is the j-th set bit); valid while maxPathLength <= 31 */
};
VCM_HD F4 &lv(const LightStore &s, size_t slot, int k) { return s.v[slot * VCM_LV_FIELDS + (size_t)k]; }

/* Hash grid, vertices sorted by cell (replaces mIndices indirection,
 * hashgrid.hxx:83-88): cell c = [cellStart[c], cellStart[c+1]) */
struct alignas(8) F2 { float x, y; };
struct GridStore {
    const int *cellStart;   /* nCells+1 */
    const float *gx, *gy, *gz;   /* position, one array per axis (padded by VCM_MERGE_UNROLL): the distance
                                    test reads nothing else, and consecutive candidates of a cell arrive as the
                                    two halves of a packed fp32 operand */
    const float *gb;        /* the positions once more, BLOCKED: photons 4b .. 4b+3 as x0..x3 y0..y3 z0..z3 (48 bytes, 16-byte aligned):
                               the three loads of a step of k_merge_pairs fall into ONE line (or two) instead of three arrays'
                               (grid_blocked_index) */
    const F4 *g1;           /* WorldDirFix.xyz | light ContinuationProb */
    const F4 *g2;           /* throughput.xyz | dVCM */
    const F2 *g3;           /* dVM | pathLength bits */
    /* (the three as ONE array of 48-byte records -- one or two cache lines per accepted photon instead of three -- was
       measured in round 4: K4 itself 3.0 -> 2.7 ms, the grid build's strided writes slower by as much: profiles/archive/r06i_ab.txt) */
    const GridHeader *hdr;
};

VCM_HD size_t grid_blocked_index(int i, int axis) { return (size_t)(i >> 2) * 12u + (size_t)(axis * 4 + (i & 3)); }

/* Camera vertices of one iteration (wavefront mode).  Direct illumination,
 * vertex connection and merging only ADD to the pixel colour, they never steer
 * the path (vertexcm.hxx:487-538), so the camera pass appends one 80-byte
 * record per non-delta vertex plus one task per light connection and dense
 * kernels evaluate them afterwards; k_resolve then replays the additions of
 * every path in the reference's order. */
struct alignas(16) I4 { int x, y, z, w; };
struct VertexStore {
    /* the record of camera vertex i: four 16-byte fields, contiguous and 64-byte aligned (q[i * 4 + k]) = what the merge
       (K4, which GATHERS the vertices in cell order: one 128-byte line each) and the connections read, plus a fifth
       in an array of its own (q4[i]) that only K3b / K3c want.  As one 80-byte record (round 2) a vertex
       straddled two lines half of the time: 864 -> 875 Mpaths/s with the split (profiles/archive/r05h_ab_summary.txt); as five
       separate arrays (round 1) a wave's append was five partly written lines per step and every gather touched five
       (K4 3.55 -> 3.39 ms when they were joined, profiles/archive/r02r_ab_summary.txt)
         k=0 hitpoint.xyz | local path index
         k=1 isect.normal.xyz | pathLength (bits 0-7), matID (8-15)
         k=2 localDirFix.xyz | dVCM
         k=3 throughput.xyz | dVM
         k=4 dVC | the position of the 3 random floats of DirectIllumination (:672-673) in the path's stream */
    F4 *q, *q4;
    size_t qcap;     /* records allocated */
    I4 *meta;        /* per PATH SLOT: DI task (-1: none) | first VC task | number of VC tasks | 0 */
    int *count;      /* [0] vertices  [1] DI tasks  [2] VC tasks                  */
    int *diTask;     /* DI task -> vertex                                         */
    int *vcTask;     /* VC task -> (vertex, index j of the light vertex)          */
    /* What k_resolve reads per vertex is indexed by the vertex's PATH SLOT (pathLength-1)*nLocal + lp, not by its
       queue position: lanes of k_resolve hold neighbouring paths, so these reads coalesce (the queue order is the
       order in which waves happened to append).  k_resolve is bound by its fetches (2.3 GB for 1.0 GB of use: the planes
       of the longer path lengths are sparse, and a 16-byte entry read there is a 64-byte fetch from each array); three
       other layouts were measured in round 4 and none kept:  indexed by queue position (dense arrays + a 4-byte
       slot -> vertex plane) k_resolve fetched 20 % MORE -- a wave of K3 appends the vertices of 64 paths at different
       depths, neighbouring paths' vertices of one length are not neighbours in the queue (profiles/archive/r06v_fetch.txt);
       one 32-byte record per slot {di.xyz, mg.xyz, first VC task, count} -35 % fetch and k_resolve 413 -> 320 us, one
       48-byte record {meta, di, mg} -23 % and 353 us -- but K3b's and K4's stores, 16 bytes next to their neighbours' in
       an array of their own, become strided partial lines: K3b 420 -> 525 / 574 us, the iteration 1.5 % / 3 % SLOWER
       (profiles/archive/r06w_ab_slot32.txt, r06x_ab_slot48.txt); {meta, di} as one 32-byte record that K3b writes whole, so that K3 has
       no slot store of its own: K3 -0.23 GB written, k_resolve -39 us, K3b +52 us, the iteration equal (r06y_ab_metadi.txt). */
    F4 *diOut;       /* per path slot: throughput * DirectIllumination()  (:491)   */
    F4 *vcOut;       /* per VC task:   throughput * lvThroughput * ConnectVertices() (:523) */
    F4 *mergeOut;    /* per path slot: throughput * vmNormalization * contrib (:534) */
    /* K4a folded into K3 when the grid exists before the camera pass (single-rank order light -> grid -> camera):
       the vertex takes its bucket key and its place in the bucket the moment it is appended; NULL otherwise */
    const GridHeader *sortHdr;
    int *sortKey, *sortArrival, *bucketCount;
};
struct alignas(8) VcTaskPair { int vertex, j; };   /* one entry of VertexStore::vcTask */
VCM_HD F4 &vq(const VertexStore &vs, int k, size_t i) { return k < 4 ? vs.q[i * 4 + (size_t)k] : vs.q4[i]; }
VCM_HD size_t path_slot(const IterParams &P, uint32_t pathLength, uint32_t lp)
{
    return (size_t)(pathLength - 1u) * (size_t)P.nLocal + (size_t)lp;
}
/* where the merge term of camera vertex `vi` goes */
VCM_HD size_t merge_out_slot(const IterParams &P, uint32_t pathLength, uint32_t lp, int vi)
{
    return P.foreignOut ? (size_t)vi : path_slot(P, pathLength, lp);
}

struct LaneStats {
    uint32_t lightRays, cameraRays, shadowRays, mergeQueries, mergeCandidates, mergeAccepted,
             connections, lightSplats, stored;
};
VCM_HD void lane_stats_zero(LaneStats &s)
{
    s.lightRays = s.cameraRays = s.shadowRays = s.mergeQueries = s.mergeCandidates = 0;
    s.mergeAccepted = s.connections = s.lightSplats = s.stored = 0;
}

/* Measurement build only (make variant NAME=rc EXTRA=-DVCM_REGION_CLOCK, profiles/tools/region_clock.py): where
 * the shader-clock time of a wave goes inside the big kernels.  RC_MARK(id) charges the cycles since the previous
 * mark of this wave to region `id` (first active lane: one atomic per wave and mark).  Compiles to nothing otherwise. */
#if defined(VCM_REGION_CLOCK) && defined(__HIPCC__)
#define VCM_RC_SLOTS 2048   /* one row per (wave mod SLOTS): the marks of a launch would otherwise queue on a dozen words */
#define VCM_RC_IDS 64
__device__ unsigned long long g_regionClock[VCM_RC_SLOTS * 3 * VCM_RC_IDS];   /* row: [id] cycles, [IDS + id] marks, [2 IDS + id] cycles x lanes at the mark */
#endif
#if defined(VCM_REGION_CLOCK) && defined(__HIP_DEVICE_COMPILE__)
/* The clock is the WAVE's (one word of LDS per wave): a mark charges the cycles since the wave's previous mark -- whichever
   lanes executed that one -- so the branches of a divergent region, which the wave runs one after the other, are each
   charged their own time, with the lanes that ran them. */
__device__ __forceinline__ unsigned long long *rc_slot()
{
    __shared__ unsigned long long rcT[16];
    return &rcT[(threadIdx.x >> 6) & 15];
}
__device__ __forceinline__ void rc_reset() { *rc_slot() = clock64(); }
__device__ __forceinline__ void rc_mark(int id)
{
    const unsigned long long n = clock64();
    const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
    unsigned long long *t = rc_slot();
    if ((int)__lane_id() == __ffsll((long long)m) - 1) {
        const unsigned long long d = n - *t;
        unsigned long long *row = g_regionClock + (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (VCM_RC_SLOTS - 1)) * (3 * VCM_RC_IDS);
        atomicAdd(&row[id], d);
        atomicAdd(&row[VCM_RC_IDS + id], 1ull);
        atomicAdd(&row[2 * VCM_RC_IDS + id], d * (unsigned long long)__popcll(m));
        *t = clock64();
    }
}
#define RC_DECL rc_reset()
#define RC_MARK(id) rc_mark(id)
#define RC_RESET rc_reset()
#else
#define RC_DECL
#define RC_MARK(id)
#define RC_RESET
#endif

/* float atomic add to the framebuffer (light splats land on arbitrary
 * pixels: vertexcm.hxx:931).  Order of concurrent adds is not defined, which
 * is the one place results are not bit-reproducible (DESIGN.md "Parity"). */
VCM_HD void fb_atomic_add(float *addr, float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsafeAtomicAdd(addr, v);
#else
    *addr = *addr + v;
#endif
}

/* ------------------------------------------------------------------ */
/* Ray / Isect: ray.hxx:34-65                                           */
struct Ray { V3 org, dir; float tmin; };
struct Isect { float dist; int matID; int lightID; V3 normal; int prim; /* index in vcm_scene_desc::prims of the hit */ };

/* wave-level "any lane" (one lane on the host build) */
VCM_HD bool wave_any(bool x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    /* the ballot of a condition IS its lane mask (s_and with exec, s_cmp); __any() turns the condition into 0 / 1 per
       lane and compares that again (v_cndmask + v_cmp, per call -- three per step of K4's scan) */
    return __builtin_amdgcn_ballot_w64(x) != 0ull;
#else
    return x;
#endif
}
VCM_HD bool wave_all(bool x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(!x) == 0ull;
#else
    return x;
#endif
}

/* ---- two-wide fp32 (packed v_pk_mul_f32 / v_pk_add_f32 on gfx950: IEEE per half,
 *      twice the scalar fp32 rate; never fused, the build has -ffp-contract=off) */
#if defined(__HIP_DEVICE_COMPILE__)
typedef float f2 __attribute__((ext_vector_type(2)));
VCM_HD f2 f2_mk(float a, float b) { f2 r = { a, b }; return r; }
VCM_HD float f2_get(f2 a, int i) { return i ? a.y : a.x; }
#else
struct f2 { float x, y; };
VCM_HD f2 f2_mk(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
VCM_HD float f2_get(f2 a, int i) { return i ? a.y : a.x; }
VCM_HD f2 operator+(f2 a, f2 b) { return f2_mk(a.x + b.x, a.y + b.y); }
VCM_HD f2 operator-(f2 a, f2 b) { return f2_mk(a.x - b.x, a.y - b.y); }
VCM_HD f2 operator*(f2 a, f2 b) { return f2_mk(a.x * b.x, a.y * b.y); }
#endif
VCM_HD f2 f2_ld(const float *p) { return f2_mk(p[0], p[1]); }
VCM_HD f2 f2_sp(float a) { return f2_mk(a, a); }
/* Dot for both halves, same order as vcm::dot (T res(0); res += ...) */
VCM_HD f2 f2_dot(f2 ax, f2 ay, f2 az, f2 bx, f2 by, f2 bz)
{
    f2 r = f2_sp(0.f);
    r = r + ax * bx; r = r + ay * by; r = r + az * bz;
    return r;
}

/* Device-side scene: the C-ABI description plus the primitive list regrouped
 * for the intersection loop -- consecutive triangles of GeometryList::mGeometry
 * (geometry.hxx:104) in PAIRS whose fields are interleaved {tri a, tri b}, so a
 * wave-uniform scalar load of 8 bytes feeds both halves of a packed operation.
 * List order is preserved (closest-hit ties go to the first primitive). */
struct alignas(8) TriPair {
    float p0x[2], p0y[2], p0z[2], p1x[2], p1y[2], p1z[2], p2x[2], p2y[2], p2z[2], nx[2], ny[2], nz[2];
    int matID[2];
    int prim[2];  /* index of each triangle in vcm_scene_desc::prims */
    int valid1;   /* 0: the pair holds only one triangle */
    int pad;
};
struct PrimOp { int kind; int index; };   /* kind 0: TriPair pairs[index]; kind 1: sphere prims[index] */

/* The brute-force list as the CERTIFIED FILTER sees it (scene_intersect / scene_occluded below): the triangles in
 * GeometryList order, two per entry, and the spheres.  Per triangle the vertex and normal the reference's plane part
 * reads, and the three edge functions of Triangle::Intersect (geometry.hxx:133-142) in Pluecker form,
 *     Dot(Cross(P - o, Q - o), dir)  =  Dot(dir, Cross(P, Q)) + Dot(Cross(o, dir), Q - P),
 * so that an edge costs six fused multiply-adds on the ray's direction and moment instead of the reference's two
 * vertex offsets, a cross product and a dot product.  The two triangles of a quad share their diagonal with opposite
 * orientation: the second triangle's third edge function is MINUS the first one's third (exactly: every term changes
 * sign), so a pair takes five edge evaluations.  One entry = one burst of scalar loads (the index is wave-uniform)
 * and one wait: the loop is bound by that latency as much as by its arithmetic. */
struct alignas(16) FastPair {
    float p0[2][3], n[2][3];    /* vertex 0 and Triangle::mNormal of the two triangles */
    float NE[6][6];             /* {N, E} of edges A0 A1 A2 B0 B1 B2:  W = Dot(dir, N) + Dot(Cross(o, dir), E) */
    int   prim[2];              /* indices in prims[] */
    int   flags;                /* bit 0: triangle B present; bit 1: B2 = -A2 (the entry holds B2 all the same: testing
                                   the bit would split the burst of loads in two and cost more than six fma);
                                   bit 2: B's plane operands Dot(n, p0 - o), Dot(n, dir) equal A's for EVERY ray (same
                                   normal up to the sign of zeros, and p0 differs only along axes where the normal is
                                   zero: the two triangles of an axis-aligned quad) -- one plane part serves both */
    int   pad;
};
struct alignas(16) FastSphere { float c[3], radius; int prim; int pad[3]; };
/* An AXIS-ALIGNED RECTANGLE of the list -- two consecutive triangles with a common plane normal to a coordinate axis
 * k, sharing their diagonal, the other four edges parallel to the axes u = k+1, v = k+2 (cyclic): every quad of the
 * reference's Cornell boxes (scene.hxx:215-330).  For an edge P -> Q in the plane the reference's edge function is
 *     V(P, Q) = Dot(dir, Cross(P - o, Q - o)) = dir_k * [ (P_u - X_u)(Q_v - X_v) - (P_v - X_v)(Q_u - X_u) ],
 * X = the point where the ray meets the plane (the offsets to o differ from those to X by multiples of dir, which drop
 * out of the triple product).  For an edge along u (P_v = Q_v = c) that is  dir_k * (P_u - Q_u) * (c - X_v),  for one
 * along v (P_u = Q_u = c)  dir_k * (Q_v - P_v) * (c - X_u):  a coordinate difference of the hit point times two
 * factors known per rectangle and ray -- against six fused multiply-adds for the same edge in Pluecker form.  Only the
 * diagonal keeps its Pluecker evaluation.  64 bytes instead of FastPair's 208: the burst of scalar loads per entry,
 * on which the loop stalls, shrinks with it.  Entries are grouped by the axis k (three loops with compile-time
 * components; the filter's answer does not depend on the order of the entries). */
struct alignas(16) FastRect {
    float pk, nk;        /* the plane: coordinate along k of the vertices; Triangle::mNormal's component along k (+-1) */
    float c[4];          /* the constant coordinate of: A's edge along u (a v value), A's edge along v (a u value), B's two likewise */
    float g[4];          /* their signed extents: V = dir_k * g * (c - X_w) */
    float NEd[6];        /* the diagonal as triangle A's edge function (B's is its negative), Pluecker form */
    int   prim[2];       /* list indices of A and B */
};

/* One node of the bounding-volume hierarchy used for scenes with more primitives than the brute-force loop is
 * meant for (the reference has no acceleration structure: README:208-209, Scene::Intersect scene.hxx:53-70).
 * Nodes are stored in depth-first order and THREADED: a traversal needs no stack -- when the ray meets a node's box
 * it moves on to the next node in memory (the first child, or for a leaf: after its primitives), otherwise it jumps
 * to `escape`, the node after the subtree.  32 bytes, two 16-byte loads. */
struct alignas(16) BvhNode {
    float bmin[3]; int escape;        /* index of the node after this subtree (nNodes at the end) */
    float bmax[3]; int leaf;          /* >= 0: (first << 4) | count into leafPrims, count <= 15; < 0: inner node, -1 - its index
                                         in the array of wide nodes */
};
/* The same hierarchy as the ordered traversals walk it: one 64-byte record per INNER node holding the boxes of BOTH
 * children and what they are, so a level costs one load (four 16-byte words) and two slab tests -- with BvhNode alone it
 * was the node, its left child and, dependent on that, the right child: two dependent round trips per level.  A child
 * is named twice: `node` = its BvhNode (what goes on the closest-hit stack: the box is tested again when it is popped,
 * against the distance held by then), `ref` = what to do with it: >= 0 a leaf (the descriptor of BvhNode::leaf),
 * < 0 an inner node, -1 - its wide index.  (With the two boxes interleaved per bound and the slab arithmetic packed --
 * six v_pk_add_f32 + six v_pk_mul_f32 for twelve + twelve -- the mesh scene ran at 451 against 455 Mpaths/s: the
 * traversal is not bound by those instructions.  Round 4, not kept.) */
/* The primitives of the leaves, copied in LEAF order with their list index: a leaf's (at most 15, usually <= 4)
 * primitives are one contiguous run of 64-byte records -- through leafPrims[] -> prims[] every triangle test began with
 * two dependent gathers. */
struct alignas(16) LeafPrim { vcm_prim prim; int index; int pad; };
struct alignas(16) BvhWide {
    float lmin[3]; int lnode;
    float lmax[3]; int lref;
    float rmin[3]; int rnode;
    float rmax[3]; int rref;
};

/* The scene as the device functions see it.  Scalars are those of vcm_scene_desc (the C-ABI struct the scene arrives
 * in); primitives, materials, lights and the structure the intersection code walks are arrays of any length, so the
 * same code serves the reference's built-in boxes (<= 32 primitives, brute force in list order over packed triangle
 * pairs) and arbitrary scenes (vcm_scene_desc2: any counts, BVH).  Built by scene_host.h, uploaded once per context.
 *
 * The arrays are addressed as BYTE OFFSETS FROM THIS STRUCT, not through stored pointers: a kernel gets the scene as
 * `const DScene *__restrict__`, and only addresses derived from that argument carry its no-alias guarantee.  With
 * pointer members the compiler has to assume that any store of the kernel may overwrite the triangle data, so it
 * re-reads it with per-lane VECTOR loads into VGPRs instead of scalar loads into SGPRs (the list index is
 * wave-uniform): K3 122 -> 133 registers, 4 -> 3 waves per SIMD, 713 -> 576 Mpaths/s on the same box (r02g). */
struct DScene {
    /* (as a type, see the kinds below: may the kernels assume that every Phong exponent is an integer in [1, 65536]?) */
    static constexpr bool kIntPhong = false;
    int nPrims, nMaterials, nLights, backgroundLight;
    float sceneCenter[3], sceneRadius, invSceneRadiusSqr;
    vcm_camera camera;
    /* brute force: GeometryList order, triangles in pairs (nOps > 0 and nNodes == 0); BVH: nNodes > 0 */
    int nOps, nNodes;
    long long offPrims, offMaterials, offMat2light, offLights, offOps, offPairs, offNodes, offLeafPrims, offFastPairs, offFastSpheres, offWide, offLeafData, offFastRects;
    /* scene constants of the filter's error bounds: max |vertex|^2 over the triangles; a sphere around their vertices */
    float fastRw2, fastCenter[3], fastRadius;
    int nFastPairs, nFastSpheres;
    int fastOnePlane;   /* every FastPair has flags bit 2: the loops then contain no per-entry branch (one burst of loads) */
    int nFastRects[3];  /* every FastPair is an axis-aligned rectangle: their FastRect view, grouped by normal axis (else 0, 0, 0) */
    float fastGmax;     /* the longest rectangle edge (enters an error bound) */
    template <class T> VCM_HD const T *at(long long off) const { return reinterpret_cast<const T *>(reinterpret_cast<const char *>(this) + off); }
    VCM_HD const vcm_prim *prims() const { return at<vcm_prim>(offPrims); }
    VCM_HD const vcm_material *materials() const { return at<vcm_material>(offMaterials); }
    VCM_HD const int *mat2light() const { return at<int>(offMat2light); }
    VCM_HD const vcm_light *lights() const { return at<vcm_light>(offLights); }
    VCM_HD const PrimOp *ops() const { return at<PrimOp>(offOps); }
    VCM_HD const TriPair *pairs() const { return at<TriPair>(offPairs); }
    VCM_HD const BvhNode *nodes() const { return at<BvhNode>(offNodes); }
    VCM_HD const int *leafPrims() const { return at<int>(offLeafPrims); }
    VCM_HD const BvhWide *wide() const { return at<BvhWide>(offWide); }
    VCM_HD const LeafPrim *leafData() const { return at<LeafPrim>(offLeafData); }
    VCM_HD const FastRect *fastRects() const { return at<FastRect>(offFastRects); }
    VCM_HD const FastPair *fastPairs() const { return at<FastPair>(offFastPairs); }
    VCM_HD const FastSphere *fastSpheres() const { return at<FastSphere>(offFastSpheres); }
};
/* Which of the two a scene carries, as a TYPE: every kernel that casts rays exists once per kind (the launch picks by
 * nNodes), so the brute-force kernels hold no traversal code and the BVH kernels no list loop.  Compiled together the
 * two paths cost the headline kernels 11-27 VGPRs, i.e. a wave per SIMD (K3 122 -> 133 registers: 713 -> 576
 * Mpaths/s on the same box, profiles/archive/r02g_*).  Functions that cast rays take `const SC &`, the rest `const DScene &`. */
/* kIntPhong: the host found every Phong exponent in use to be an integer in [1, 65536] (the reference's scenes: 90), so
 * pow(x, n) of the lobe is detmath.h's binary exponentiation and nothing else -- the kernels of such a kind hold no call
 * of the general powf where a lobe is only EVALUATED (the call's register constraints cost k_merge_walk its fourth wave
 * per SIMD: 135 registers against 120).  Scenes with other exponents take SceneList / SceneBvhG. */
struct SceneList : DScene { static constexpr bool kBvh = false; static constexpr bool kOnePlane = false; static constexpr bool kRects = false; static constexpr bool kIntPhong = false; };
struct SceneBvh : DScene { static constexpr bool kBvh = true; static constexpr bool kOnePlane = false; static constexpr bool kRects = false; static constexpr bool kIntPhong = true; };
struct SceneBvhG : DScene { static constexpr bool kBvh = true; static constexpr bool kOnePlane = false; static constexpr bool kRects = false; static constexpr bool kIntPhong = false; };
/* a list whose triangle pairs all share their plane part (FastPair::flags bit 2: axis-aligned quads, i.e. the reference's
   Cornell boxes): its kernels carry only that loop */
struct SceneQuads : DScene { static constexpr bool kBvh = false; static constexpr bool kOnePlane = true; static constexpr bool kRects = false; static constexpr bool kIntPhong = true; };
/* a list whose triangle pairs are all axis-aligned rectangles (FastRect): the reference's own boxes */
struct SceneRects : DScene { static constexpr bool kBvh = false; static constexpr bool kOnePlane = true; static constexpr bool kRects = true; static constexpr bool kIntPhong = true; };

/* ---- the scene's small tables in LDS ----
 * A lane's material, the primitive its ray ends on, the light it samples are GATHERS (the index is per lane): as global
 * loads, one dependent trip through the vector memory path in front of the component probabilities, one inside every
 * branch of BSDF::Sample, one per Evaluate / Pdf, one for the winner of Scene::Intersect, one per light sample -- in
 * kernels that wait for latency, not bandwidth.  Every kernel that traces or shades copies the tables (materials,
 * primitives, material -> light, lights: the reference's scenes have at most 10 / 26 / 10 / 2 entries) into LDS when it
 * starts (stage_scene_tables: all threads of the block, one barrier) and scene_material() / scene_prim() /
 * scene_mat2light() / scene_light() read them from there; a table with more entries than its room keeps the global path
 * (a scalar branch on the count).  Materials alone: 887 / 893 -> 924 / 917 Mpaths/s, same box (profiles/archive/r05s_ab.txt).
 * RULE: a kernel that can reach one of the accessors calls stage_scene_tables() first -- nothing else initialises the
 * copy (the merge kernels do not, and pass lds = false where they read a material). */
#define VCM_LDS_MATERIALS 32
#define VCM_LDS_PRIMS 32
#define VCM_LDS_LIGHTS 4
#define VCM_LDS_MAT_WORDS 11
#define VCM_LDS_PRIM_WORDS 14
#define VCM_LDS_LIGHT_WORDS 24
#define VCM_LDS_OFF_PRIMS (VCM_LDS_MATERIALS * VCM_LDS_MAT_WORDS)
#define VCM_LDS_OFF_M2L (VCM_LDS_OFF_PRIMS + VCM_LDS_PRIMS * VCM_LDS_PRIM_WORDS)
#define VCM_LDS_OFF_LIGHTS (VCM_LDS_OFF_M2L + VCM_LDS_MATERIALS)
#define VCM_LDS_SCENE_WORDS (VCM_LDS_OFF_LIGHTS + VCM_LDS_LIGHTS * VCM_LDS_LIGHT_WORDS)
#if defined(__HIP_DEVICE_COMPILE__)
__shared__ uint32_t g_ldsScene[VCM_LDS_SCENE_WORDS];
typedef const __attribute__((address_space(3))) uint32_t *LdsWords;
__device__ __forceinline__ LdsWords lds_scene(int offset) { return (LdsWords)g_ldsScene + offset; }
#endif
VCM_HD void stage_scene_tables(const DScene &sc)
{
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(vcm_material) == VCM_LDS_MAT_WORDS * 4 && sizeof(vcm_prim) == VCM_LDS_PRIM_WORDS * 4 &&
                  sizeof(vcm_light) == VCM_LDS_LIGHT_WORDS * 4, "table entries are whole words");
    const int t = (int)threadIdx.x, nt = (int)blockDim.x;
    if (sc.nMaterials <= VCM_LDS_MATERIALS) {
        const uint32_t *src = (const uint32_t *)sc.materials();
        for (int i = t; i < sc.nMaterials * VCM_LDS_MAT_WORDS; i += nt) g_ldsScene[i] = src[i];
        const uint32_t *m2l = (const uint32_t *)sc.mat2light();
        for (int i = t; i < sc.nMaterials; i += nt) g_ldsScene[VCM_LDS_OFF_M2L + i] = m2l[i];
    }
    if (sc.nPrims <= VCM_LDS_PRIMS) {
        const uint32_t *src = (const uint32_t *)sc.prims();
        for (int i = t; i < sc.nPrims * VCM_LDS_PRIM_WORDS; i += nt) g_ldsScene[VCM_LDS_OFF_PRIMS + i] = src[i];
    }
    if (sc.nLights <= VCM_LDS_LIGHTS) {
        const uint32_t *src = (const uint32_t *)sc.lights();
        for (int i = t; i < sc.nLights * VCM_LDS_LIGHT_WORDS; i += nt) g_ldsScene[VCM_LDS_OFF_LIGHTS + i] = src[i];
    }
    dm_stage_tables();   /* powf's two tables (detmath.h), 512 bytes */
    __syncthreads();
#else
    (void)sc;
#endif
}
/* lds = false: the caller's kernel does not stage the tables (the merge kernels: two material reads per QUERY are
   nothing next to its candidates, and the copy's registers cost k_merge_walk its fourth wave per SIMD) */
VCM_HD vcm_material scene_material(const DScene &sc, int matID, bool lds = true)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (lds && sc.nMaterials <= VCM_LDS_MATERIALS) {   /* wave-uniform */
        LdsWords p = lds_scene(matID * VCM_LDS_MAT_WORDS);
        vcm_material m;
        m.diffuse[0] = u2f(p[0]); m.diffuse[1] = u2f(p[1]); m.diffuse[2] = u2f(p[2]);
        m.phong[0] = u2f(p[3]); m.phong[1] = u2f(p[4]); m.phong[2] = u2f(p[5]);
        m.phongExp = u2f(p[6]);
        m.mirror[0] = u2f(p[7]); m.mirror[1] = u2f(p[8]); m.mirror[2] = u2f(p[9]);
        m.ior = u2f(p[10]);
        return m;
    }
#endif
    return sc.materials()[matID];
}
VCM_HD vcm_prim scene_prim(const DScene &sc, int index)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (sc.nPrims <= VCM_LDS_PRIMS) {
        LdsWords p = lds_scene(VCM_LDS_OFF_PRIMS + index * VCM_LDS_PRIM_WORDS);
        vcm_prim r;
        r.type = (int)p[0]; r.matID = (int)p[1];
        for (int k = 0; k < 3; k++) { r.p0[k] = u2f(p[2 + k]); r.p1[k] = u2f(p[5 + k]); r.p2[k] = u2f(p[8 + k]); r.n[k] = u2f(p[11 + k]); }
        return r;
    }
#endif
    return sc.prims()[index];
}
VCM_HD int scene_mat2light(const DScene &sc, int matID)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (sc.nMaterials <= VCM_LDS_MATERIALS) return (int)lds_scene(VCM_LDS_OFF_M2L)[matID];
#endif
    return sc.mat2light()[matID];
}
VCM_HD vcm_light scene_light(const DScene &sc, int index)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (sc.nLights <= VCM_LDS_LIGHTS) {
        LdsWords p = lds_scene(VCM_LDS_OFF_LIGHTS + index * VCM_LDS_LIGHT_WORDS);
        vcm_light l;
        l.type = (int)p[0];
        for (int k = 0; k < 3; k++) {
            l.p0[k] = u2f(p[1 + k]); l.e1[k] = u2f(p[4 + k]); l.e2[k] = u2f(p[7 + k]);
            l.frameX[k] = u2f(p[10 + k]); l.frameY[k] = u2f(p[13 + k]); l.frameZ[k] = u2f(p[16 + k]); l.intensity[k] = u2f(p[19 + k]);
        }
        l.invArea = u2f(p[22]); l.scale = u2f(p[23]);
        return l;
    }
#endif
    return sc.lights()[index];
}


/* ---- utils.hxx ---------------------------------------------------- */
VCM_HD float luminance(V3 c)
{   /* :36-41 */
    return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z;
}
VCM_HD float fresnel_dielectric(float cosInc, float ior)
{   /* :43-74 */
    if (ior < 0.f) return 1.f;
    float eta;
    if (cosInc < 0.f) { cosInc = -cosInc; eta = ior; }
    else              { eta = 1.f / ior; }
    const float sinTrans2 = sqr(eta) * (1.f - sqr(cosInc));
    const float cosTrans = sqrtf(smax(0.f, 1.f - sinTrans2));
    const float term1 = eta * cosTrans;
    const float rParallel = (cosInc - term1) / (cosInc + term1);
    const float term2 = eta * cosInc;
    const float rPerp = (term2 - cosTrans) / (term2 + cosTrans);
    return 0.5f * (sqr(rParallel) + sqr(rPerp));
}
VCM_HD V3 reflect_local(V3 v) { return mk3(-v.x, -v.y, v.z); }   /* :77-80 */

/* (sin, cos of term1 = 2 pi sx come from the caller: BSDF::Sample's diffuse and Phong branches both start from them,
   same argument, and a wave that holds lanes of both kinds would evaluate dm_sincosf once per branch) */
VCM_HD V3 sample_power_cos_hemisphere(float s, float c, float sy, float power)
{   /* :85-103, oPdfW == NULL at its only call site (bsdf.hxx:296) */
    const float term2 = dm_powf(sy, 1.f / (power + 1.f), true);   /* the kernels that sample stage detmath's tables */
    const float term3 = sqrtf(1.f - term2 * term2);
    return mk3(c * term3, s * term3, term2);
}
VCM_HD float power_cos_hemisphere_pdf(V3 n, V3 d, float power, bool intPhong = false)
{   /* :105-113 */
    const float cosTheta = smax(0.f, dot(n, d));
    return (power + 1.f) * dm_powf_wave(cosTheta, power, true, intPhong) * (VCM_INV_PI_F * 0.5f);
}
VCM_HD void sample_concentric_disc(float sx, float sy, float &ox, float &oy)
{   /* :119-160 */
    float phi, r;
    const float a = 2 * sx - 1;
    const float b = 2 * sy - 1;
    if (a > -b) {
        if (a > b) { r = a;  phi = (VCM_PI_F / 4.f) * (b / a); }
        else       { r = b;  phi = (VCM_PI_F / 4.f) * (2.f - (a / b)); }
    } else {
        if (a < b) { r = -a; phi = (VCM_PI_F / 4.f) * (4.f + (b / a)); }
        else {
            r = -b;
            if (b != 0) phi = (VCM_PI_F / 4.f) * (6.f - (a / b));
            else        phi = 0;
        }
    }
    float s, c;
    dm_sincosf_cold(phi, s, c);
    ox = r * c;
    oy = r * s;
}
VCM_HD float concentric_disc_pdf_a() { return VCM_INV_PI_F; }   /* :162-165 */
VCM_HD V3 sample_cos_hemisphere_sc(float s, float c, float sy, float &pdfW)
{   /* :173-190 with sin, cos of term1 = 2 pi sx given */
    const float term2 = sqrtf(1.f - sy);
    const V3 ret = mk3(c * term2, s * term2, sqrtf(sy));
    pdfW = ret.z * VCM_INV_PI_F;
    return ret;
}
VCM_HD V3 sample_cos_hemisphere(float sx, float sy, float &pdfW)
{   /* :173-190 */
    const float term1 = 2.f * VCM_PI_F * sx;
    float s, c;
    dm_sincosf(term1, s, c);
    return sample_cos_hemisphere_sc(s, c, sy, pdfW);
}
VCM_HD float cos_hemisphere_pdf(V3 n, V3 d) { return smax(0.f, dot(n, d)) * VCM_INV_PI_F; }   /* :192-197 */
VCM_HD void sample_uniform_triangle(float sx, float sy, float &u, float &v)
{   /* :202-207 */
    const float term = sqrtf(sx);
    u = 1.f - term;
    v = sy * term;
}
VCM_HD V3 sample_uniform_sphere(float sx, float sy, float &pdf)
{   /* :212-230 */
    const float term1 = 2.f * VCM_PI_F * sx;
    const float term2 = 2.f * sqrtf(sy - sy * sy);
    float s, c;
    dm_sincosf_cold(term1, s, c);
    const V3 ret = mk3(c * term2, s * term2, 1.f - 2.f * sy);
    pdf = VCM_INV_PI_F * 0.25f;
    return ret;
}
VCM_HD float uniform_sphere_pdf() { return VCM_INV_PI_F * 0.25f; }   /* :232-236 */
VCM_HD float pdf_w_to_a(float pdfW, float dist, float cosThere)
{   /* :245-251 */
    return pdfW * fabsf(cosThere) / sqr(dist);
}

VCM_HD float pdf_a_to_w(float pdfA, float dist, float cosThere)
{   /* :253-259 */
    return pdfA * sqr(dist) / fabsf(cosThere);
}

/* ---- geometry.hxx ------------------------------------------------- */
VCM_HD bool sph_intersect(const vcm_prim &s, int primIndex, const Ray &ray, Isect &res)
{   /* Sphere::Intersect :198-237.  The discriminant is evaluated in float and
       only then widened (:211); sqrt, q, t0, t1 are double (:216-220). */
    const V3 center = ld3(s.p0);
    const float radius = s.p1[0];
    const V3 to = ray.org - center;
    const float A = dot(ray.dir, ray.dir);
    const float B = 2 * dot(ray.dir, to);
    const float C = dot(to, to) - (radius * radius);
    const float discF = B * B - 4 * A * C;
    const double disc = discF;
    if (disc < 0) return false;
    const double discSqrt = sqrt(disc);
    const double q = (B < 0) ? ((-B - discSqrt) / 2.f) : ((-B + discSqrt) / 2.f);
    double t0 = q / A;
    double t1 = C / q;
    if (t0 > t1) { const double t = t0; t0 = t1; t1 = t; }
    float resT;
    if (t0 > ray.tmin && t0 < res.dist)      resT = float(t0);
    else if (t1 > ray.tmin && t1 < res.dist) resT = float(t1);
    else return false;
    res.dist = resT;
    res.matID = s.matID;
    res.prim = primIndex;
    res.normal = normalize(to + sp3(resT) * ray.dir);
    return true;
}
/* Triangle::Intersect (:125-156) for the two triangles of a pair, in two parts: the plane part (ao = p0 - org,
 * num = Dot(normal, ao), den = Dot(normal, dir): the operands of `distance = num / den`, :144-147) and the edge
 * functions (:133-142), both with packed operations -- IEEE per half, the reference's expression trees. */
struct TriPairPlane { f2 ox, oy, oz, dx, dy, dz, aox, aoy, aoz, nx, ny, nz, num, den; };
VCM_HD void tri_pair_plane(const TriPair &t, V3 org, V3 dir, TriPairPlane &p)
{
    p.ox = f2_sp(org.x); p.oy = f2_sp(org.y); p.oz = f2_sp(org.z);
    p.dx = f2_sp(dir.x); p.dy = f2_sp(dir.y); p.dz = f2_sp(dir.z);
    p.aox = f2_ld(t.p0x) - p.ox; p.aoy = f2_ld(t.p0y) - p.oy; p.aoz = f2_ld(t.p0z) - p.oz;
    p.nx = f2_ld(t.nx); p.ny = f2_ld(t.ny); p.nz = f2_ld(t.nz);
    p.num = f2_dot(p.nx, p.ny, p.nz, p.aox, p.aoy, p.aoz);
    p.den = f2_dot(p.nx, p.ny, p.nz, p.dx, p.dy, p.dz);
}
VCM_HD void tri_pair_inside(const TriPair &t, const TriPairPlane &p, bool inside[2])
{
    const f2 box = f2_ld(t.p1x) - p.ox, boy = f2_ld(t.p1y) - p.oy, boz = f2_ld(t.p1z) - p.oz;
    const f2 cox = f2_ld(t.p2x) - p.ox, coy = f2_ld(t.p2y) - p.oy, coz = f2_ld(t.p2z) - p.oz;
    const f2 aox = p.aox, aoy = p.aoy, aoz = p.aoz;
    /* v0 = Cross(co, bo), v1 = Cross(bo, ao), v2 = Cross(ao, co)  (math.hxx:154-162) */
    const f2 v0x = coy * boz - coz * boy, v0y = coz * box - cox * boz, v0z = cox * boy - coy * box;
    const f2 v1x = boy * aoz - boz * aoy, v1y = boz * aox - box * aoz, v1z = box * aoy - boy * aox;
    const f2 v2x = aoy * coz - aoz * coy, v2y = aoz * cox - aox * coz, v2z = aox * coy - aoy * cox;
    const f2 v0d = f2_dot(v0x, v0y, v0z, p.dx, p.dy, p.dz);
    const f2 v1d = f2_dot(v1x, v1y, v1z, p.dx, p.dy, p.dz);
    const f2 v2d = f2_dot(v2x, v2y, v2z, p.dx, p.dy, p.dz);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int h = 0; h < 2; h++) {
        const float a = f2_get(v0d, h), b = f2_get(v1d, h), c = f2_get(v2d, h);
        inside[h] = ((a < 0.f) && (b < 0.f) && (c < 0.f)) || ((a >= 0.f) && (b >= 0.f) && (c >= 0.f));
    }
}
/* the two closest-hit updates are applied in list order (exactly what two consecutive calls of
 * Triangle::Intersect do) */
VCM_HD bool tri_pair_intersect(const TriPair &t, const Ray &ray, Isect &res)
{
    TriPairPlane p;
    tri_pair_plane(t, ray.org, ray.dir, p);
    bool inside[2];
    tri_pair_inside(t, p, inside);
    bool anyHit = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int h = 0; h < 2; h++) {
        const float distance = f2_get(p.num, h) / f2_get(p.den, h);
        if ((h == 0 || t.valid1) && inside[h] && (distance > ray.tmin) && (distance < res.dist)) {
            res.normal = mk3(f2_get(p.nx, h), f2_get(p.ny, h), f2_get(p.nz, h));
            res.matID = t.matID[h];
            res.prim = t.prim[h];
            res.dist = distance;
            anyHit = true;
        }
    }
    return anyHit;
}
/* Any-hit form for Scene::Occluded (GeometryList::IntersectP returns at the first primitive that reports a hit,
 * geometry.hxx:80-91, so every primitive is tested against the SAME interval (0, tmax)).  Before the edge
 * functions (3/4 of the work) the plane part decides, EXACTLY, whether a triangle can report a hit at all:
 * `distance = num / den` (fp32 division, correctly rounded) is positive only if num and den have the same sign,
 * and it is < tmax only if |num| < tmax * |den| -- rounding is monotonic, so |num| >= tmax * |den| in exact
 * arithmetic implies fl(num / den) >= tmax; the test below allows for the rounding of its own two
 * multiplications (1.000001f > (1 - 2^-24)^-2).  The condition is a superset of the hits, never an
 * approximation of them: a triangle that passes it takes the full test.  A shadow segment between two surface
 * points of the Cornell box passes it for none of the walls (they lie behind its start or beyond its end), and
 * the wave skips the edge functions when no lane needs them. */
VCM_HD bool tri_pair_occluded(const TriPair &t, V3 org, V3 dir, float tmax)
{
    TriPairPlane p;
    tri_pair_plane(t, org, dir, p);
    bool can[2];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int h = 0; h < 2; h++) {
        const float a = f2_get(p.num, h), b = f2_get(p.den, h);
        const bool sameSign = ((f2u(a) ^ f2u(b)) & 0x80000000u) == 0u;
        const bool beyond = fabsf(a) >= 1.000001f * (tmax * fabsf(b));   /* false for NaN: the full test decides */
        can[h] = sameSign && !beyond && (h == 0 || t.valid1);
    }
#if !defined(VCM_NO_OCCLUSION_PRETEST)   /* measurement switch: evaluate every triangle in full */
    if (!wave_any(can[0] || can[1])) return false;
#endif
    bool inside[2];
    tri_pair_inside(t, p, inside);
    bool hit = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int h = 0; h < 2; h++) {
        const float distance = f2_get(p.num, h) / f2_get(p.den, h);
        if ((h == 0 || t.valid1) && inside[h] && (distance > 0.f) && (distance < tmax)) hit = true;
    }
    return hit;
}
/* Triangle::Intersect (:125-156) for ONE triangle of the primitive array (BVH leaves; the brute-force loop uses the
 * packed pairs): edge functions, then the plane distance.  `accept` decides what the closest-hit loop does with
 * the distance (see bvh_intersect). */
VCM_HD bool tri_inside(const vcm_prim &t, V3 org, V3 dir, float &distance)
{
    const V3 ao = ld3(t.p0) - org, bo = ld3(t.p1) - org, co = ld3(t.p2) - org;
    const V3 v0 = cross(co, bo), v1 = cross(bo, ao), v2 = cross(ao, co);
    const float v0d = dot(v0, dir), v1d = dot(v1, dir), v2d = dot(v2, dir);
    const bool inside = ((v0d < 0.f) && (v1d < 0.f) && (v2d < 0.f)) || ((v0d >= 0.f) && (v1d >= 0.f) && (v2d >= 0.f));
    distance = 0.f;
    if (inside) {   /* the reference divides only here too (:144-147); a wave skips it when no lane is inside */
        const V3 n = ld3(t.n);
        distance = dot(n, ao) / dot(n, dir);
    }
    return inside;
}

/* Does the ray meet the node's box within [0, tmax]?  The boxes are grown at build time by far more than the
 * rounding of this test and of the primitives' own hit computations (scene_host.h), so the answer errs only towards
 * "yes": the traversal visits a superset of the primitives that can report a hit. */
VCM_HD bool bvh_box_hit(const BvhNode &nd, V3 org, V3 invDir, float tmax)
{
    const float ax = (nd.bmin[0] - org.x) * invDir.x, bx = (nd.bmax[0] - org.x) * invDir.x;
    const float ay = (nd.bmin[1] - org.y) * invDir.y, by = (nd.bmax[1] - org.y) * invDir.y;
    const float az = (nd.bmin[2] - org.z) * invDir.z, bz = (nd.bmax[2] - org.z) * invDir.z;
    /* fminf / fmaxf return the other operand for a NaN (0 * inf on a slab plane): the slab then does not constrain */
    const float tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
    const float tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    return tnear <= tfar * 1.0000004f;
}

/* GeometryList::Intersect (geometry.hxx:65-78) literally: every primitive in list order, each keeping its hit only
 * if strictly closer than what is held.  O(nPrims): the BVH traversal falls back to it for the rare ray whose
 * outcome depends on the list order in a way a hierarchy cannot see (below). */
VCM_HD bool list_intersect(const DScene &sc, const Ray &ray, Isect &res)
{
    bool any = false;
    for (int pi = 0; pi < sc.nPrims; pi++) {
        const vcm_prim &pr = sc.prims()[pi];
        if (pr.type == VCM_PRIM_TRIANGLE) {
            float distance;
            const bool inside = tri_inside(pr, ray.org, ray.dir, distance);
            if (inside && (distance > ray.tmin) && (distance < res.dist)) {
                res.normal = ld3(pr.n); res.matID = pr.matID; res.prim = pi; res.dist = distance; any = true;
            }
        } else if (sph_intersect(pr, pi, ray, res)) any = true;
    }
    if (any) res.lightID = sc.mat2light()[res.matID];
    return any;
}

/* Scene::Intersect over a BVH.  The reference walks GeometryList in order and keeps a hit only if it is strictly
 * closer (geometry.hxx:65-78, :150, :226-234), i.e. it returns the hit with the smallest distance, ties going to the
 * lower list index.  A hierarchy visits the primitives in another order, so the comparison is made explicitly
 * lexicographic on (distance, primitive index): same winner, bit for bit.
 * One case depends on the list order beyond that: Sphere::Intersect compares its binary64 root with the binary32
 * distance held so far and then holds the root rounded to binary32 (:216-234), so when a sphere and another
 * primitive are hit within an ulp of each other the winner depends on which came first.  Such a ray (two surfaces
 * within 1e-7 of each other along it: the contact point of a sphere resting on the floor) is re-done by the
 * list walk. */
VCM_HD bool near_tie(float a, float b) { const int d = (int)f2u(a) - (int)f2u(b); return d >= -2 && d <= 2; }
/* entry distance of the ray into a box (bvh_box_hit's tnear); hit = the box is met within [0, tmax] */
VCM_HD bool bvh_box_near6(const float *bmin, const float *bmax, V3 org, V3 invDir, float tmax, float &tnear)
{
    const float ax = (bmin[0] - org.x) * invDir.x, bx = (bmax[0] - org.x) * invDir.x;
    const float ay = (bmin[1] - org.y) * invDir.y, by = (bmax[1] - org.y) * invDir.y;
    const float az = (bmin[2] - org.z) * invDir.z, bz = (bmax[2] - org.z) * invDir.z;
    tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
    const float tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    return tnear <= tfar * 1.0000004f;
}
VCM_HD bool bvh_box_near(const BvhNode &nd, V3 org, V3 invDir, float tmax, float &tnear)
{
    return bvh_box_near6(nd.bmin, nd.bmax, org, invDir, tmax, tnear);
}
/* host-only measurement hook (profiles/tools/bvh_sim.py): the event stream of a traversal -- B/b begin (closest / any hit),
   I inner step, p pop inside it, L leaf primitive, P pop after a leaf, O end of an outer round.  Nothing on the device. */
#if defined(VCM_BVH_PROFILE) && !defined(__HIP_DEVICE_COMPILE__)
void vcm_bvh_event(char e);
void vcm_bvh_ray(const float *org, const float *dir, float tmax);
#define VCM_BVH_EV(e) vcm_bvh_event(e)
#define VCM_BVH_RAY(r, t) do { const float o_[3] = { (r).org.x, (r).org.y, (r).org.z }, d_[3] = { (r).dir.x, (r).dir.y, (r).dir.z }; vcm_bvh_ray(o_, d_, t); } while (0)
#else
#define VCM_BVH_EV(e) ((void)0)
#define VCM_BVH_RAY(r, t) ((void)0)
#endif
/* the primitives of one leaf (descriptor (first << 4) | count) against the hit held so far: lexicographic on
   (distance, list index), see below */
VCM_HD void bvh_leaf(const DScene &sc, int leaf, const Ray &ray, Isect &res, bool &any, bool &ambiguous, bool &bestIsSphere)
{
    const int first = leaf >> 4, count = leaf & 15;
    for (int k = 0; k < count; k++) {
        VCM_BVH_EV('L');
        const LeafPrim &lp = sc.leafData()[first + k];
        const vcm_prim &pr = lp.prim;
        const int pi = lp.index;
        if (pr.type == VCM_PRIM_TRIANGLE) {
            float distance;
            const bool inside = tri_inside(pr, ray.org, ray.dir, distance);
            if (inside && (distance > ray.tmin)) {
                if (any && bestIsSphere && near_tie(distance, res.dist)) ambiguous = true;
                if (distance < res.dist || (distance == res.dist && any && pi < res.prim)) {
                    res.normal = ld3(pr.n); res.matID = pr.matID; res.prim = pi; res.dist = distance; any = true;
                    bestIsSphere = false;
                }
            }
        } else {
            /* Sphere::Intersect offers ONE distance -- the nearer root beyond tmin, else the farther one --
               whatever res.dist is (if the nearer root fails "< res.dist" so does the farther) */
            Isect s; s.dist = 1e36f; s.matID = 0; s.lightID = -1; s.normal = sp3(0.f); s.prim = -1;
            if (sph_intersect(pr, pi, ray, s)) {
                if (any && near_tie(s.dist, res.dist)) ambiguous = true;
                if (s.dist < res.dist || (s.dist == res.dist && any && pi < res.prim)) {
                    res.normal = s.normal; res.matID = s.matID; res.prim = pi; res.dist = s.dist; any = true;
                    bestIsSphere = true;
                }
            }
        }
    }
}
/* any primitive of the leaf hit within (0, tmax)?  (GeometryList::IntersectP, geometry.hxx:80-91, for a subset) */
VCM_HD bool bvh_leaf_occluded(const DScene &sc, int leaf, const Ray &ray, float tmaxp)
{
    const int first = leaf >> 4, count = leaf & 15;
    bool occluded = false;
    for (int k = 0; k < count; k++) {
        VCM_BVH_EV('L');
        const LeafPrim &lp = sc.leafData()[first + k];
        const vcm_prim &pr = lp.prim;
        const int pi = lp.index;
        if (pr.type == VCM_PRIM_TRIANGLE) {
            float distance;
            const bool inside = tri_inside(pr, ray.org, ray.dir, distance);
            if (inside && (distance > 0.f) && (distance < tmaxp)) occluded = true;
        } else {
            Isect s; s.dist = tmaxp; s.matID = 0; s.lightID = -1; s.normal = sp3(0.f); s.prim = -1;
            if (sph_intersect(pr, pi, ray, s)) occluded = true;
        }
    }
    return occluded;
}

/* The traversal stack: 32 levels per lane, in LDS on the device ([level][thread], no bank conflicts; 32 KB per block of
 * 256 lanes), ONE array for the closest-hit and the any-hit traversal -- a kernel that runs both (strict mode) never
 * has both alive.  A deeper tree finishes the ray with the threaded walk. */
#ifndef VCM_BVH_STACK
#define VCM_BVH_STACK 32
#endif
#define VCM_BVH_NONE 0x7fffffff
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int *bvh_stack(int &stride)
{
    __shared__ int stackNode[VCM_BVH_STACK][256];
    stride = 256;
    return &stackNode[0][threadIdx.x];
}
#endif

/* Closest hit, ORDERED, one wide node per level: both children's boxes are tested, the nearer is descended first and
 * the farther goes on the stack, to be dropped unvisited if a hit closer than its box has been found by the time it is
 * popped (its own BvhNode is loaded then and tested against the distance held NOW) -- the threaded walk (the first
 * version of this function) visits the subtrees in memory order whatever the ray's direction and prunes only by what
 * it happens to have found.  The loop is "while-while": a lane keeps descending until it holds a leaf, and the leaves
 * are intersected when no lane of the wave has an inner node left -- so the slab tests run with the lanes that
 * descend and the triangle tests with the lanes that hold leaves, instead of every step paying for both.
 * The visiting order cannot change the result: the winner is the minimum of (distance, list index), and a
 * primitive within 2 ulp of the winner is never pruned (the boxes are padded by 1e-4 of the scene, scene_host.h), so
 * the near-tie rule sees the same pairs. */
VCM_HD bool bvh_intersect(const DScene &sc, const Ray &ray, Isect &res)
{
    const V3 invDir = mk3(1.f / ray.dir.x, 1.f / ray.dir.y, 1.f / ray.dir.z);
    const Isect start = res;
    bool any = false, ambiguous = false, bestIsSphere = false, overflow = false;
#if defined(__HIP_DEVICE_COMPILE__)
    int stride;
    int *sn = bvh_stack(stride);
#else
    int stackNode[VCM_BVH_STACK];
    int *sn = stackNode;
    const int stride = 1;
#endif
    int sp = 0, ref = VCM_BVH_NONE;
    VCM_BVH_EV('B'); VCM_BVH_RAY(ray, res.dist);
#if defined(VCM_BVH_THREADED)   /* measurement switch: the threaded walk only */
    overflow = true;
#else
    if (sc.nNodes > 0) {
        const BvhNode root = sc.nodes()[0];
        float t;
        if (bvh_box_near(root, ray.org, invDir, res.dist, t)) ref = root.leaf;
    }
#endif
    for (;;) {
        while (ref < 0) {   /* an inner node: one 64-byte record, two slab tests */
            VCM_BVH_EV('I');
            const BvhWide w = sc.wide()[-1 - ref];
            float tl, tr;
            const bool hl = bvh_box_near6(w.lmin, w.lmax, ray.org, invDir, res.dist, tl);
            const bool hr = bvh_box_near6(w.rmin, w.rmax, ray.org, invDir, res.dist, tr);
            if (hl && hr) {
                const bool leftFirst = tl <= tr;
                if (sp < VCM_BVH_STACK) { sn[sp * stride] = leftFirst ? w.rnode : w.lnode; sp++; }
                else overflow = true;
                ref = leftFirst ? w.lref : w.rref;
            } else if (hl) ref = w.lref;
            else if (hr) ref = w.rref;
            else {   /* next pending subtree that can still hold a closer (or equal: the tie rule) hit */
                ref = VCM_BVH_NONE;
                while (ref == VCM_BVH_NONE && sp > 0) {
                    VCM_BVH_EV('p');
                    sp--;
                    const BvhNode nd = sc.nodes()[sn[sp * stride]];
                    float t;
                    if (bvh_box_near(nd, ray.org, invDir, res.dist, t)) ref = nd.leaf;
                }
            }
        }
        if (ref == VCM_BVH_NONE) break;
        bvh_leaf(sc, ref, ray, res, any, ambiguous, bestIsSphere);
        ref = VCM_BVH_NONE;
        while (ref == VCM_BVH_NONE && sp > 0) {
            VCM_BVH_EV('P');
            sp--;
            const BvhNode nd = sc.nodes()[sn[sp * stride]];
            float t;
            if (bvh_box_near(nd, ray.org, invDir, res.dist, t)) ref = nd.leaf;
        }
        VCM_BVH_EV('O');
        if (ref == VCM_BVH_NONE) break;
    }
    if (overflow) {   /* deeper than the stack: the threaded walk over the whole tree (order-free, same result) */
        int nodeT = 0;
        while (nodeT < sc.nNodes) {
            const BvhNode nd = sc.nodes()[nodeT];
            if (!bvh_box_hit(nd, ray.org, invDir, res.dist)) { nodeT = nd.escape; continue; }
            if (nd.leaf >= 0) bvh_leaf(sc, nd.leaf, ray, res, any, ambiguous, bestIsSphere);
            nodeT++;
        }
    }
    if (ambiguous) { res = start; return list_intersect(sc, ray, res); }
    if (any) res.lightID = sc.mat2light()[res.matID];
    return any;
}

/* Scene::Occluded over the BVH: any hit in (0, tmax).  The answer does not depend on the order, and tmax does not
 * shrink: a child whose box is met is simply remembered by its REF and needs no second test when it is popped -- one
 * 64-byte record per inner node is all the traversal loads above the leaves (the threaded walk of rounds 1-2 loaded a
 * 32-byte node per visited node, each load dependent on the one before).  While-while as above; the nearer child
 * first, because an occluder near the origin ends the ray. */
VCM_HD bool bvh_occluded(const DScene &sc, const Ray &ray, float tmaxp)
{
    const V3 invDir = mk3(1.f / ray.dir.x, 1.f / ray.dir.y, 1.f / ray.dir.z);
    bool occluded = false, overflow = false;
#if defined(__HIP_DEVICE_COMPILE__)
    int stride;
    int *sn = bvh_stack(stride);
#else
    int stackNode[VCM_BVH_STACK];
    int *sn = stackNode;
    const int stride = 1;
#endif
    int sp = 0, ref = VCM_BVH_NONE;
    VCM_BVH_EV('b'); VCM_BVH_RAY(ray, tmaxp);
#if defined(VCM_BVH_THREADED)
    overflow = true;
#else
    if (sc.nNodes > 0) {
        const BvhNode root = sc.nodes()[0];
        if (bvh_box_hit(root, ray.org, invDir, tmaxp)) ref = root.leaf;
    }
#endif
    while (!overflow) {
        while (ref < 0) {
            VCM_BVH_EV('I');
            const BvhWide w = sc.wide()[-1 - ref];
            float tl, tr;
            const bool hl = bvh_box_near6(w.lmin, w.lmax, ray.org, invDir, tmaxp, tl);
            const bool hr = bvh_box_near6(w.rmin, w.rmax, ray.org, invDir, tmaxp, tr);
            if (hl && hr) {
                const bool leftFirst = tl <= tr;
                if (sp < VCM_BVH_STACK) { sn[sp * stride] = leftFirst ? w.rref : w.lref; sp++; ref = leftFirst ? w.lref : w.rref; }
                else { overflow = true; ref = VCM_BVH_NONE; }
            } else if (hl) ref = w.lref;
            else if (hr) ref = w.rref;
            else if (sp > 0) { sp--; ref = sn[sp * stride]; }
            else ref = VCM_BVH_NONE;
        }
        if (ref == VCM_BVH_NONE) break;
        if (bvh_leaf_occluded(sc, ref, ray, tmaxp)) { occluded = true; break; }
        VCM_BVH_EV('O');
        if (sp > 0) { sp--; ref = sn[sp * stride]; }
        else break;
    }
    if (overflow && !occluded) {   /* deeper than the stack: the threaded walk over the whole tree */
        int node = 0;
        while (node < sc.nNodes && !occluded) {
            const BvhNode nd = sc.nodes()[node];
            if (!bvh_box_hit(nd, ray.org, invDir, tmaxp)) { node = nd.escape; continue; }
            if (nd.leaf >= 0) occluded = bvh_leaf_occluded(sc, nd.leaf, ray, tmaxp);
            node++;
        }
    }
    return occluded;
}

/* GeometryList::Intersect (geometry.hxx:65-78) over the packed pairs: the reference's operations, every primitive
 * in list order (the op index is wave-uniform, so the primitive data comes in through scalar loads). */
VCM_HD bool pairs_intersect(const DScene &sc, const Ray &ray, Isect &res)
{
    bool any = false;
    for (int i = 0; i < sc.nOps; i++) {
        const PrimOp op = sc.ops()[i];
        const bool hit = (op.kind == 0) ? tri_pair_intersect(sc.pairs()[op.index], ray, res)
                                        : sph_intersect(sc.prims()[op.index], op.index, ray, res);
        if (hit) any = hit;
    }
    if (any) res.lightID = sc.mat2light()[res.matID];
    return any;
}
/* GeometryList::IntersectP (geometry.hxx:80-91) over the packed pairs */
VCM_HD bool pairs_occluded(const DScene &sc, const Ray &ray, float tmaxp)
{
    bool occluded = false;
    for (int i = 0; i < sc.nOps; i++) {
        const PrimOp op = sc.ops()[i];
        if (!occluded) {
            bool hit;
            if (op.kind == 0) hit = tri_pair_occluded(sc.pairs()[op.index], ray.org, ray.dir, tmaxp);
            else {
                Isect isect;
                isect.dist = tmaxp; isect.matID = 0; isect.lightID = -1; isect.normal = sp3(0.f); isect.prim = -1;
                hit = sph_intersect(sc.prims()[op.index], op.index, ray, isect);
            }
            if (hit) occluded = true;
        }
    }
    return occluded;
}

/* ---- certified filters in front of the reference's intersection arithmetic ---------------------------------------
 * What has to come out of Scene::Intersect is WHICH primitive the reference's loop ends up holding and the distance,
 * normal and material it computed for it; of Scene::Occluded only a boolean.  The reference's expression trees
 * (offsets, cross products, dot products per triangle; binary64 roots per sphere) DEFINE those answers, but for all
 * but a vanishing fraction of the rays the answers do not depend on the last bits: a ray either passes an edge at a
 * distance far above every rounding error or it does not.  So every primitive is first put through a cheap
 * approximation with a RIGOROUS error bound (u = 2^-24; Lw = max(|o|, max |vertex|), Lo >= the distance from the
 * ray's origin to every triangle vertex; |dir| = 1 up to rounding, the bounds are scaled by max(1, |dir|^2)):
 *
 *   edge function   W  = fma chain of Dot(dir, N) + Dot(Cross(o, dir), E)                      6 fma
 *                   |W - exact| <= 29 u Lw^2  (N, E rounded once from binary64; the moment; six roundings of the chain)
 *                   |reference's v0d - exact| <= 14 u Lo^2  (three rounded offsets, cross, dot: standard forward bound)
 *                   => |W| > tauW = u (64 Lw^2 + 32 Lo^2)  (twice the sum) : the reference's sign is the sign of W
 *   plane distance  the reference's own two operands num = Dot(n, p0 - o), den = Dot(n, dir) (same trees, geometry.hxx
 *                   :144-147: 13 operations), then t = num * rcp(den) instead of the correctly rounded division:
 *                   |t - reference's distance| <= eps = 8 u |t|.  (A Hessian-form plane, 6 fma, was measured first:
 *                   its ABSOLUTE error of ~1e-6 cannot tell on which side of a surface a shadow ray starts that
 *                   leaves it at a grazing angle -- 0.5 % of the shadow rays, a quarter of the waves, fell back.)
 *   sphere roots    the reference's own binary32 A, B, C and discriminant (same tree, geometry.hxx:205-211; "no real
 *                   root" is therefore exact), then its roots in binary32 instead of binary64 (:216-220) with the
 *                   error of every step carried along: fast_sphere below
 *
 * A primitive is then classified as certainly hit / certainly missed / unknown, and
 *   closest hit:  the winner is certain when the candidate with the smallest lower bound t - eps is a certain hit and
 *                 its upper bound t + eps lies below the lower bound of every other candidate; the reference's own
 *                 arithmetic then runs for THAT primitive only (plane distance of one triangle, or the binary64
 *                 roots of one sphere), so distance and normal are the reference's bits;
 *   any hit:      one certain hit decides "occluded", all primitives certainly missed decides "free".
 * Whenever a lane of the wave is left without a certain answer (a ray within ~1e-5 of an edge, of a tangent, of two
 * surfaces meeting) the WAVE re-does the query with the reference's arithmetic over the whole list -- the loops
 * above, which are also what the filter replaced.  Results are therefore bit-identical to the brute-force loop by
 * construction; the parity tests compare full frames (tests/test_gpu_parity.py).  Measured: DESIGN.md section 5. */
#if defined(__HIP_DEVICE_COMPILE__)
VCM_HD float approx_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     /* 1 ulp */
VCM_HD float approx_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }   /* 1 ulp */
#else
VCM_HD float approx_rcp(float x) { return 1.f / x; }
VCM_HD float approx_sqrt(float x) { return sqrtf(x); }
/* how often the filter hands a ray to the reference loop (host emulation only: tests/test_core_emul.py reports it) */
struct FilterStats { unsigned long long isect, isectExact, occl, occlExact; };
inline FilterStats g_filterStats = { 0, 0, 0, 0 };
#endif
/* v_min_f32 / v_max_f32 as they are (minNum / maxNum: a NaN operand loses, which the filter relies on); fminf / fmaxf
   compile to the same instruction behind a canonicalising v_max_f32 x, x of every operand that might be a signalling NaN */
#if defined(__HIP_DEVICE_COMPILE__)
VCM_HD float filter_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
VCM_HD float filter_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
VCM_HD float filter_min(float a, float b) { return fminf(a, b); }
VCM_HD float filter_max(float a, float b) { return fmaxf(a, b); }
#endif
#define VCM_FILTER_U 5.9604645e-8f          /* 2^-24 */
#define VCM_FILTER_INF 3.0e38f
struct FastRay { V3 o, d, m; float tauW; };
VCM_HD void fast_ray_setup(const DScene &sc, V3 org, V3 dir, FastRay &r)
{
    r.o = org; r.d = dir;
    r.m = mk3(__builtin_fmaf(org.y, dir.z, -(org.z * dir.y)), __builtin_fmaf(org.z, dir.x, -(org.x * dir.z)),
              __builtin_fmaf(org.x, dir.y, -(org.y * dir.x)));
    const float oo = __builtin_fmaf(org.x, org.x, __builtin_fmaf(org.y, org.y, org.z * org.z));
    const float dd = fmaxf(1.f, __builtin_fmaf(dir.x, dir.x, __builtin_fmaf(dir.y, dir.y, dir.z * dir.z)));
    const float lw2 = fmaxf(oo, sc.fastRw2);
    const V3 oc = org - ld3(sc.fastCenter);
    const float lo = approx_sqrt(__builtin_fmaf(oc.x, oc.x, __builtin_fmaf(oc.y, oc.y, oc.z * oc.z))) * 1.00001f + sc.fastRadius;
    r.tauW = (VCM_FILTER_U * dd) * __builtin_fmaf(64.f, lw2, 32.f * (lo * lo));
}
struct FastHit { float L, U; bool certIn, certOut; };
/* plane part: the reference's operands num = Dot(n, p0 - o), den = Dot(n, dir), bit for bit; t = num * rcp(den) */
VCM_HD void fast_tri_plane(const float *p0, const float *nrm, const FastRay &r, FastHit &h, float &num, float &den)
{
    const V3 n = ld3(nrm);
    num = dot(n, ld3(p0) - r.o);
    den = dot(n, r.d);
    const float t = num * approx_rcp(den);
    const float eps = (VCM_FILTER_U * 8.f) * fabsf(t);
    const bool known = fabsf(t) < 1e30f;      /* false for inf / NaN (den = 0 or denormal, overflow) */
    h.L = known ? t - eps : -VCM_FILTER_INF;
    h.U = known ? t + eps : VCM_FILTER_INF;
}
VCM_HD float fast_edge(const float *NE, const FastRay &r)
{
    return __builtin_fmaf(r.d.x, NE[0], __builtin_fmaf(r.d.y, NE[1], __builtin_fmaf(r.d.z, NE[2],
           __builtin_fmaf(r.m.x, NE[3], __builtin_fmaf(r.m.y, NE[4], r.m.z * NE[5])))));
}
VCM_HD void fast_classify(float w0, float w1, float w2, float tau, FastHit &h)
{
    /* geometry.hxx:141-142 with certain signs: all three below -tau or all three above tau = certainly inside; one below
       -tau and one above tau = certainly outside.  Taken on the smallest and the largest of the three (v_min3 / v_max3,
       four comparisons, two mask operations -- six comparisons and ten mask operations when written per edge, and the
       scalar unit issues no faster than a SIMD: profiles/archive/r05o_ab.txt, +1.3 %).  min / max skip a NaN operand, so a
       caller whose operands can be NaN makes tau infinite instead (fast_rect_edges; Pluecker edges are finite). */
    const float lo = fminf(fminf(w0, w1), w2), hi = fmaxf(fmaxf(w0, w1), w2);
    h.certIn = (bool)((int)(hi < -tau) | (int)(lo > tau));
    h.certOut = (bool)((int)(lo < -tau) & (int)(hi > tau));
}
/* the edge functions of both triangles of an entry */
VCM_HD void fast_pair_edges(const FastPair &p, const FastRay &r, FastHit &ha, FastHit &hb)
{
    const float a0 = fast_edge(p.NE[0], r), a1 = fast_edge(p.NE[1], r), a2 = fast_edge(p.NE[2], r);
    const float b0 = fast_edge(p.NE[3], r), b1 = fast_edge(p.NE[4], r), b2 = fast_edge(p.NE[5], r);
    fast_classify(a0, a1, a2, r.tauW, ha);
    fast_classify(b0, b1, b2, r.tauW, hb);
}
/* The reference's binary32 part of Sphere::Intersect (geometry.hxx:205-211: A, B, C and the discriminant, same
 * trees, so "no real root" is exact), then ITS two roots approximately.  They are not simply the roots of the
 * quadratic: the reference takes q = (-B - sqrt(disc)) / 2 for B < 0 and (-B + sqrt(disc)) / 2 otherwise (:217) --
 * the CANCELLING combination -- in binary64, from a discriminant that was rounded to binary32, and t0 = q / A,
 * t1 = C / q.  The far root C / q therefore carries the relative error of B*B - disc against 4AC, up to ~1e-5 for a
 * ray that starts on the sphere (C small): a feature of the reference, reproduced here.  In binary32 the
 * cancellation is avoided algebraically, q = +-(B*B - disc) / (2 (|B| + sqrt(disc))), with B*B - disc evaluated
 * without loss (B*B as an exact two-term product, the subtraction exact by Sterbenz or harmless), and the error of
 * every step is carried along: |root - reference's| <= e = (rel + 16 u) |root|, rel = 2 u (|e1| + |lo|) / |D|.
 * ok = false: roots unknown (q = 0, overflow, NaN).  The factor 4 on e covers the two roots changing places when
 * they nearly coincide (geometry.hxx:222). */
struct FastRoots { float lo, hi, eLo, eHi; bool noRoot, ok; };
VCM_HD void fast_sphere(const FastSphere &p, V3 org, V3 dir, FastRoots &fr)
{
    const V3 to = org - ld3(p.c);
    const float radius = p.radius;
    const float A = dot(dir, dir);
    const float B = 2 * dot(dir, to);
    const float C = dot(to, to) - (radius * radius);
    const float discF = B * B - 4 * A * C;
    fr.noRoot = discF < 0;
    const float s = approx_sqrt(fmaxf(discF, 0.f));
    const float b = fabsf(B);
    const float hi = b * b, lo = __builtin_fmaf(b, b, -hi);   /* b*b = hi + lo exactly */
    const float e1 = hi - discF;
    const float D = e1 + lo;                                  /* B*B - disc */
    const float rel = (VCM_FILTER_U * 2.f) * (fabsf(e1) + fabsf(lo)) * approx_rcp(fabsf(D)) + VCM_FILTER_U * 16.f;
    const float qa = D * approx_rcp(2.f * (b + s));
    const float q = (B < 0) ? qa : -qa;
    const float t0 = q * approx_rcp(A), t1 = C * approx_rcp(q);
    fr.ok = (fabsf(q) > 1e-30f) && (A > 1e-30f) && (fabsf(t0) < 1e30f) && (fabsf(t1) < 1e30f) && (rel < 0.01f);   /* false for NaN */
    fr.lo = fminf(t0, t1); fr.hi = fmaxf(t0, t1);
    fr.eLo = (4.f * rel) * fabsf(fr.lo) + 1e-30f;
    fr.eHi = (4.f * rel) * fabsf(fr.hi) + 1e-30f;
}

/* running choice of the closest-hit filter: the candidate with the smallest lower bound, and the second smallest */
struct FastBest { float minL1, minL2, bestU; int best; bool bestCertain, poison /* some entry's bounds are unknown: nothing is certain */; };
VCM_HD float filter_med3(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(a, b, c);
#else
    return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
#endif
}
VCM_HD void fast_offer(FastBest &fb, bool cand, bool cert, float L, float U, int prim)
{
    const float Lc = cand ? L : VCM_FILTER_INF;
    const bool isBest = Lc < fb.minL1;
    fb.minL2 = filter_med3(fb.minL1, fb.minL2, Lc);   /* minL1 <= minL2: the second smallest of the three */
    fb.minL1 = filter_min(fb.minL1, Lc);
    fb.best = isBest ? prim : fb.best;
    fb.bestU = isBest ? U : fb.bestU;
    fb.bestCertain = (bool)(((int)isBest & (int)cert) | ((int)!isBest & (int)fb.bestCertain));   /* mask logic (scalar unit): no select of 0 / 1, no branch */
}
/* ---- the filter for lists of axis-aligned rectangles (FastRect) ----
 * Per rectangle: the plane distance t from the reference's own operands (for a normal along axis k they are
 * num = n_k (p_k - o_k), den = n_k dir_k exactly, up to the sign of a zero: the other two products of the reference's
 * dot are zeros); the hit point's in-plane coordinates X~_u = fma(t, dir_u, o_u), X~_v likewise; then the four axis
 * edges as  dir_k g (c - X~_w)  and the diagonal in Pluecker form, classified exactly as the general filter does.
 * Error bound (u = 2^-24): t~ carries a relative error <= 8 u against the exact quotient (one rounded subtraction,
 * the reciprocal, one product), so |X~_w - X_w| <= e_p = 16 u (|t| + |o|) (|dir| <= 1 up to rounding; generous);
 * the two products of the proxy add 2 u relative.  The proxy w~ of an axis edge therefore differs from the exact edge
 * function V by at most |dir_k| |g| e_p (1 + 2 u) + 2 u |w~|, and the reference's computed value from V by at most
 * 14 u Lo^2 (general filter, above).  With tau = max(tauW, tauRef + 2.5 |dir_k| gmax e_p), tauRef = 32 u Lo^2 -- twice
 * the reference's bound -- |w~| > tau certifies the sign the reference computes; the diagonal keeps tauW. */
struct FastRayRect { float tauRef, lw1, gmax; };
VCM_HD void fast_ray_setup_rect(const DScene &sc, V3 org, V3 dir, FastRay &r, FastRayRect &rr)
{
    r.o = org; r.d = dir;
    r.m = mk3(__builtin_fmaf(org.y, dir.z, -(org.z * dir.y)), __builtin_fmaf(org.z, dir.x, -(org.x * dir.z)),
              __builtin_fmaf(org.x, dir.y, -(org.y * dir.x)));
    const float oo = __builtin_fmaf(org.x, org.x, __builtin_fmaf(org.y, org.y, org.z * org.z));
    const float dd = fmaxf(1.f, __builtin_fmaf(dir.x, dir.x, __builtin_fmaf(dir.y, dir.y, dir.z * dir.z)));
    const float lw2 = fmaxf(oo, sc.fastRw2);
    const V3 oc = org - ld3(sc.fastCenter);
    const float lo = approx_sqrt(__builtin_fmaf(oc.x, oc.x, __builtin_fmaf(oc.y, oc.y, oc.z * oc.z))) * 1.00001f + sc.fastRadius;
    const float lo2 = lo * lo;
    r.tauW = (VCM_FILTER_U * dd) * __builtin_fmaf(64.f, lw2, 32.f * lo2);
    rr.tauRef = (VCM_FILTER_U * 32.f * dd) * lo2;
    rr.lw1 = approx_sqrt(oo) * 1.00001f;
    rr.gmax = sc.fastGmax * 2.5f * dd;
}
template <int K> VCM_HD float v3c(V3 a) { return K == 0 ? a.x : (K == 1 ? a.y : a.z); }
/* plane part + the five edge functions of one rectangle: L, U = bounds of the reference's distance; a0, a1, a2 the
   edge functions of triangle A, b0, b1 (and -a2) those of B, as proxies with a common certainty threshold tau */
/* (n_k = +-1 exactly -- scene_host_build_rects admits nothing else -- so the reference's operands are
   num = +-(p_k - o_k), den = +-dir_k with the SAME sign, and num * rcp(den) = (p_k - o_k) * rcp(dir_k) bit for bit:
   one reciprocal per axis group instead of one per rectangle; `inv` = rcp(dir_k).) */
template <int K>
VCM_HD void fast_rect_plane(const FastRect &p, const FastRay &r, float inv, float &num, float &t, float &L, float &U)
{
    num = p.pk - v3c<K>(r.o);
    t = num * inv;
    /* an unknown t (inf / NaN: den = 0 or denormal, overflow) leaves L and U inf / NaN: no comparison with them holds,
       fast_rect_edges certifies nothing for it, and the closest-hit side poisons its choice (rect_offer_one) */
    const float eps = (VCM_FILTER_U * 8.f) * fabsf(t);
    L = t - eps;
    U = t + eps;
}
template <int K>
VCM_HD void fast_rect_edges(const FastRect &p, const FastRay &r, const FastRayRect &rr, float t, FastHit &ha, FastHit &hb)
{
    constexpr int KU = (K + 1) % 3, KV = (K + 2) % 3;
    const float dk = v3c<K>(r.d);
    const float xu = __builtin_fmaf(t, v3c<KU>(r.d), v3c<KU>(r.o)), xv = __builtin_fmaf(t, v3c<KV>(r.d), v3c<KV>(r.o));
    const float ep = (VCM_FILTER_U * 16.f) * (fabsf(t) + rr.lw1);
    /* an unknown t (inf / NaN) makes ep and the proxies NaN: nothing may be certain then */
    const float tauKnown = filter_max(r.tauW, __builtin_fmaf(fabsf(dk) * rr.gmax, ep, rr.tauRef));
    const float tau = (fabsf(t) < 1e30f) ? tauKnown : VCM_FILTER_INF;
    const float a0 = (dk * p.g[0]) * (p.c[0] - xv);
    const float a1 = (dk * p.g[1]) * (p.c[1] - xu);
    const float b0 = (dk * p.g[2]) * (p.c[2] - xv);
    const float b1 = (dk * p.g[3]) * (p.c[3] - xu);
    const float a2 = fast_edge(p.NEd, r);
    fast_classify(a0, a1, a2, tau, ha);
    fast_classify(b0, b1, -a2, tau, hb);
}
/* a value every lane holds alike, kept in a scalar register (the compiler otherwise selects between the ADDRESSES of
   prim[0] and prim[1] per lane and fetches the index with a vector load the loop then waits for, once per entry) */
/* nothing is scheduled across this point (the read-ahead below must be ISSUED before the arithmetic it overlaps) */
#if defined(__HIP_DEVICE_COMPILE__)
#define VCM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
/* first use of a scalar that a load in flight delivers: the s_waitcnt lands HERE.  Scalar loads return out of order,
   so the only wait there is is lgkmcnt(0), for all of them: the entry being waited for must be the only one in flight,
   i.e. the next entry's loads are issued after this point, not before */
#define VCM_AWAIT_SCALAR(x) asm volatile("; await" :: "s"(x))
#else
#define VCM_SCHED_FENCE()
#define VCM_AWAIT_SCALAR(x)
#endif
VCM_HD int wave_uniform(int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}
/* The entries are read one AHEAD (scalar loads: the next entry's burst is in flight while this one is evaluated;
   issued and awaited in the same trip, the loads were what the loop waited for, four waves of a SIMD together), in a
   loop unrolled by two with two register sets: with one set the compiler copies the 17 prefetched words into place
   with 34 s_mov per entry, and the scalar unit issues no faster than a SIMD does (one instruction per four cycles
   each): the loop then ran at the pace of its ~60 scalar instructions instead of its 58 vector ones. */
template <int K>
VCM_HD void rect_offer_one(const FastRect &p, const FastRay &r, const FastRayRect &rr, float inv, float tmin, float resDist, FastBest &fb)
{
    FastHit ha, hb;
    float num, t;
    fast_rect_plane<K>(p, r, inv, num, t, ha.L, ha.U);
    fast_rect_edges<K>(p, r, rr, t, ha, hb);
    /* one plane, two triangles: the same distance bounds for both; at most one of them contains the point */
    const bool reach = !(ha.U <= tmin) && !(ha.L >= resDist), sure = (ha.L > tmin) && (ha.U < resDist);
    const bool candA = reach && !ha.certOut, candB = reach && !hb.certOut;
    const bool certA = sure && ha.certIn && hb.certOut, certB = sure && hb.certIn && ha.certOut;
    const int primA = wave_uniform(p.prim[0]), primB = wave_uniform(p.prim[1]);
    fb.poison = fb.poison || !(fabsf(t) < 1e30f);
    fast_offer(fb, candA || candB, certA || certB, ha.L, ha.U, certB ? primB : primA);
}
template <int K>
VCM_HD void rects_offer(const FastRect *rects, int n, const FastRay &r, const FastRayRect &rr, float tmin, float resDist, FastBest &fb)
{
    if (n <= 0) return;
    const float inv = approx_rcp(v3c<K>(r.d));
    FastRect a = rects[0];
    int i = 0;
    for (; i + 1 < n; i += 2) {
        VCM_AWAIT_SCALAR(a.pk);
        VCM_SCHED_FENCE();
        const FastRect b = rects[i + 1];
        VCM_SCHED_FENCE();
        rect_offer_one<K>(a, r, rr, inv, tmin, resDist, fb);
        VCM_AWAIT_SCALAR(b.pk);
        VCM_SCHED_FENCE();
        if (i + 2 < n) a = rects[i + 2];
        VCM_SCHED_FENCE();
        rect_offer_one<K>(b, r, rr, inv, tmin, resDist, fb);
    }
    if (i < n) rect_offer_one<K>(a, r, rr, inv, tmin, resDist, fb);
}
/* Can the plane part report a hit in (0, tmax) at all?  fl(num / den) > 0 needs equal signs, < tmax needs
   |num| < tmax |den| up to the rounding of this test's own products (occluded_pairs below has the argument; num and
   dir_k are the reference's operands up to their common sign n_k).  Both in ONE unsigned comparison of bit patterns:
   with the sign of dir_k folded into num, "sign clear and below the bound" is "pattern below the bound's pattern"
   (patterns of non-negative floats order as the floats do, every pattern with the sign set lies above them; a NaN
   numerator reaches nothing, which is the reference's answer for it: no comparison with a NaN distance holds).  One
   comparison is also what lets the wave's "any lane?" be the comparison's own lane mask: of a combination of
   conditions the compiler first makes a 0 / 1 per lane and compares that again. */
template <int K>
VCM_HD void rect_occluded_one(const FastRect &p, const FastRay &r, const FastRayRect &rr, float inv, uint32_t dkSign, uint32_t reachBits,
                              float tmaxp, bool &occ, bool &unknown)
{
    FastHit ha, hb;
    float num, t;
    fast_rect_plane<K>(p, r, inv, num, t, ha.L, ha.U);
    const bool reach = (f2u(num) ^ dkSign) < reachBits;
    if (!wave_any(reach)) return;   /* no lane can report a hit: the wave skips the edges */
    fast_rect_edges<K>(p, r, rr, t, ha, hb);
    const bool inRange = (ha.L > 0.f) && (ha.U < tmaxp);
    const bool hitA = reach && ha.certIn && inRange, missA = !reach || ha.certOut;
    const bool hitB = reach && hb.certIn && inRange, missB = !reach || hb.certOut;
    occ = occ || hitA || hitB;
    unknown = unknown || !(hitA || missA) || !(hitB || missB);
}
template <int K>
VCM_HD void rects_occluded(const FastRect *rects, int n, const FastRay &r, const FastRayRect &rr, float tmaxp, bool &occ, bool &unknown)
{
    if (n <= 0) return;
    const float dk = v3c<K>(r.d), inv = approx_rcp(dk);
    const float reachDen = 1.000001f * (tmaxp * fabsf(dk));
    const uint32_t dkSign = f2u(dk) & 0x80000000u;
    const uint32_t reachBits = (reachDen > 0.f) ? f2u(reachDen) : ((reachDen <= 0.f) ? 0u : 0x7fc00000u);   /* NaN bound: everything reaches */
    FastRect a = rects[0];
    int i = 0;
    for (; i + 1 < n; i += 2) {
        VCM_AWAIT_SCALAR(a.pk);
        VCM_SCHED_FENCE();
        const FastRect b = rects[i + 1];
        VCM_SCHED_FENCE();
        rect_occluded_one<K>(a, r, rr, inv, dkSign, reachBits, tmaxp, occ, unknown);
        VCM_AWAIT_SCALAR(b.pk);
        VCM_SCHED_FENCE();
        if (i + 2 < n) a = rects[i + 2];
        VCM_SCHED_FENCE();
        rect_occluded_one<K>(b, r, rr, inv, dkSign, reachBits, tmaxp, occ, unknown);
    }
    if (i < n) rect_occluded_one<K>(a, r, rr, inv, dkSign, reachBits, tmaxp, occ, unknown);
}

/* Scene::Intersect over the list with the filter in front.  `certain` = this lane's answer is final. */
template <bool ONE_PLANE, bool RECTS = false>
VCM_HD bool list_intersect_filtered(const DScene &sc, const Ray &ray, Isect &res, bool &certain)
{
    FastRay r;
    FastBest fb;
    fb.minL1 = fb.minL2 = fb.bestU = VCM_FILTER_INF; fb.best = -1; fb.bestCertain = false; fb.poison = false;
    if (RECTS) {
        FastRayRect rr;
        fast_ray_setup_rect(sc, ray.org, ray.dir, r, rr);
        const FastRect *rc = sc.fastRects();
        rects_offer<0>(rc, sc.nFastRects[0], r, rr, ray.tmin, res.dist, fb);
        rects_offer<1>(rc + sc.nFastRects[0], sc.nFastRects[1], r, rr, ray.tmin, res.dist, fb);
        rects_offer<2>(rc + sc.nFastRects[0] + sc.nFastRects[1], sc.nFastRects[2], r, rr, ray.tmin, res.dist, fb);
    } else fast_ray_setup(sc, ray.org, ray.dir, r);
    if (RECTS) {
    } else if (ONE_PLANE) {
        for (int i = 0; i < sc.nFastPairs; i++) {
            const FastPair &p = sc.fastPairs()[i];
            FastHit ha, hb;
            float num, den;
            fast_tri_plane(p.p0[0], p.n[0], r, ha, num, den);
            /* one plane, two triangles: the same distance bounds for both; at most one of them contains the point */
            hb.L = ha.L; hb.U = ha.U;
            fast_pair_edges(p, r, ha, hb);
            const bool reach = !(ha.U <= ray.tmin) && !(ha.L >= res.dist), sure = (ha.L > ray.tmin) && (ha.U < res.dist);
            const bool candA = reach && !ha.certOut, candB = reach && !hb.certOut;
            const bool certA = sure && ha.certIn && hb.certOut, certB = sure && hb.certIn && ha.certOut;
            fast_offer(fb, candA || candB, certA || certB, ha.L, ha.U, certB ? p.prim[1] : p.prim[0]);
        }
    } else {
        for (int i = 0; i < sc.nFastPairs; i++) {
            const FastPair &p = sc.fastPairs()[i];
            FastHit ha, hb;
            float num, den;
            fast_tri_plane(p.p0[0], p.n[0], r, ha, num, den);
            fast_tri_plane(p.p0[1], p.n[1], r, hb, num, den);
            fast_pair_edges(p, r, ha, hb);
            const bool two = (p.flags & 1) != 0;
            fast_offer(fb, !ha.certOut && !(ha.U <= ray.tmin) && !(ha.L >= res.dist),
                       ha.certIn && (ha.L > ray.tmin) && (ha.U < res.dist), ha.L, ha.U, p.prim[0]);
            fast_offer(fb, two && !hb.certOut && !(hb.U <= ray.tmin) && !(hb.L >= res.dist),
                       hb.certIn && (hb.L > ray.tmin) && (hb.U < res.dist), hb.L, hb.U, p.prim[1]);
        }
    }
    FastSphere nextSphere;
    if (sc.nFastSpheres > 0) nextSphere = sc.fastSpheres()[0];
    for (int i = 0; i < sc.nFastSpheres; i++) {
        const FastSphere p = nextSphere;   /* read one ahead, as the rectangles are */
        if (i + 1 < sc.nFastSpheres) nextSphere = sc.fastSpheres()[i + 1];
        FastRoots fr;
        fast_sphere(p, ray.org, ray.dir, fr);
        /* Sphere::Intersect offers the first root beyond tmin (:226-234) */
        const bool loValid = fr.lo - fr.eLo > ray.tmin, loInvalid = fr.lo + fr.eLo <= ray.tmin;
        const float t = loInvalid ? fr.hi : fr.lo, e = loInvalid ? fr.eHi : fr.eLo;
        const float L = fr.ok ? t - e : -VCM_FILTER_INF, U = fr.ok ? t + e : VCM_FILTER_INF;
        const bool cand = !fr.noRoot && !(fr.ok && loInvalid && (fr.hi + fr.eHi <= ray.tmin)) && !(L >= res.dist);
        const bool cert = !fr.noRoot && fr.ok && (loValid || (loInvalid && (fr.hi - fr.eHi > ray.tmin))) && (U < res.dist);
        fast_offer(fb, cand, cert, L, U, wave_uniform(p.prim));
    }
    RC_MARK(38);
    if (fb.poison) { certain = false; return false; }
    if (fb.best < 0) { certain = true; return false; }   /* every primitive certainly missed */
    certain = fb.bestCertain && (fb.minL2 > fb.bestU);
    if (!certain) return false;
    /* the reference's arithmetic for the winner alone (per-lane index: a gather, once per ray -- out of LDS for the
       reference's scenes) */
    const vcm_prim pr = scene_prim(sc, fb.best);
    bool hit;
    if (pr.type == VCM_PRIM_TRIANGLE) {
        const V3 n = ld3(pr.n);
        const V3 ao = ld3(pr.p0) - ray.org;
        const float distance = dot(n, ao) / dot(n, ray.dir);                 /* geometry.hxx:144-147 */
        hit = (distance > ray.tmin) && (distance < res.dist);               /* certified to hold */
        if (hit) { res.normal = n; res.matID = pr.matID; res.prim = fb.best; res.dist = distance; }
        else certain = false;                                               /* cannot happen; the wave would re-do it */
    } else {
        hit = sph_intersect(pr, fb.best, ray, res);
        if (!hit) certain = false;
        RC_MARK(41);
    }
    RC_MARK(39);
    if (hit) res.lightID = scene_mat2light(sc, res.matID);
    return hit;
}
/* the triangle entries of Scene::Occluded; ONE_PLANE: every entry's two triangles share the plane part (no branch
 * inside the loop: the loads of an entry stay one burst) */
template <bool ONE_PLANE>
VCM_HD void occluded_pairs(const DScene &sc, const FastRay &r, float tmaxp, bool &occ, bool &unknown)
{
    for (int i = 0; i < sc.nFastPairs; i++) {
        const FastPair &p = sc.fastPairs()[i];
        FastHit ha, hb;
        float numA, denA, numB, denB;
        fast_tri_plane(p.p0[0], p.n[0], r, ha, numA, denA);
        if (ONE_PLANE) { hb.L = ha.L; hb.U = ha.U; numB = numA; denB = denA; }
        else fast_tri_plane(p.p0[1], p.n[1], r, hb, numB, denB);
        /* can the plane part report a hit in (0, tmax) at all?  Exact (see tri_pair_occluded): fl(num / den) > 0
           needs equal signs, < tmax needs |num| < tmax |den| up to the rounding of this test's own products */
        const bool reachA = (((f2u(numA) ^ f2u(denA)) & 0x80000000u) == 0u) && !(fabsf(numA) >= 1.000001f * (tmaxp * fabsf(denA)));
        const bool reachB = ONE_PLANE ? reachA
                                      : ((p.flags & 1) && (((f2u(numB) ^ f2u(denB)) & 0x80000000u) == 0u) && !(fabsf(numB) >= 1.000001f * (tmaxp * fabsf(denB))));
#if defined(VCM_FILTER_NOSKIP)   /* measurement switch: the edge functions of every entry */
        const bool edges = true;
#else
        const bool edges = wave_any((reachA || reachB) && !occ);   /* no lane can report a hit: the wave skips them */
#endif
        if (edges) {
            fast_pair_edges(p, r, ha, hb);
            const bool hitA = reachA && ha.certIn && (ha.L > 0.f) && (ha.U < tmaxp), missA = !reachA || ha.certOut;
            const bool hitB = reachB && hb.certIn && (hb.L > 0.f) && (hb.U < tmaxp), missB = !reachB || hb.certOut;
            occ = occ || hitA || hitB;
            unknown = unknown || !(hitA || missA) || !(hitB || missB);
        }
    }
}
/* Scene::Occluded over the list with the filter in front */
template <bool ONE_PLANE, bool RECTS = false>
VCM_HD bool list_occluded_filtered(const DScene &sc, const Ray &ray, float tmaxp, bool &certain)
{
    FastRay r;
    bool occ = false, unknown = false;
    if (RECTS) {
        FastRayRect rr;
        fast_ray_setup_rect(sc, ray.org, ray.dir, r, rr);
        const FastRect *rc = sc.fastRects();
        rects_occluded<0>(rc, sc.nFastRects[0], r, rr, tmaxp, occ, unknown);
        rects_occluded<1>(rc + sc.nFastRects[0], sc.nFastRects[1], r, rr, tmaxp, occ, unknown);
        rects_occluded<2>(rc + sc.nFastRects[0] + sc.nFastRects[1], sc.nFastRects[2], r, rr, tmaxp, occ, unknown);
    } else {
        fast_ray_setup(sc, ray.org, ray.dir, r);
        occluded_pairs<ONE_PLANE>(sc, r, tmaxp, occ, unknown);
    }
    FastSphere nextSphere;
    if (sc.nFastSpheres > 0) nextSphere = sc.fastSpheres()[0];
    for (int i = 0; i < sc.nFastSpheres; i++) {
        const FastSphere p = nextSphere;
        if (i + 1 < sc.nFastSpheres) nextSphere = sc.fastSpheres()[i + 1];
        FastRoots fr;
        fast_sphere(p, ray.org, ray.dir, fr);
        /* geometry.hxx:226-234 with res.dist = tmax: a hit iff one of the roots lies in (0, tmax) */
        const bool hitLo = (fr.lo - fr.eLo > 0.f) && (fr.lo + fr.eLo < tmaxp), hitHi = (fr.hi - fr.eHi > 0.f) && (fr.hi + fr.eHi < tmaxp);
        const bool missLo = (fr.lo + fr.eLo <= 0.f) || (fr.lo - fr.eLo >= tmaxp), missHi = (fr.hi + fr.eHi <= 0.f) || (fr.hi - fr.eHi >= tmaxp);
        const bool hitC = !fr.noRoot && fr.ok && (hitLo || hitHi);
        const bool missC = fr.noRoot || (fr.ok && missLo && missHi);
        occ = occ || hitC;
        unknown = unknown || !(hitC || missC);
    }
    certain = occ || !unknown;
    return occ;
}

/* Scene::Intersect scene.hxx:53-70 (+ GeometryList::Intersect geometry.hxx:65-78): brute force in list order for
 * the reference's own scenes (<= 32 primitives), the BVH for larger ones. */
template <class SC>
VCM_HD bool scene_intersect(const SC &sc, const Ray &ray, Isect &res)
{
    if constexpr (SC::kBvh) return bvh_intersect(sc, ray, res);
#if !defined(VCM_NO_FILTER)
    bool certain;
    Isect fast = res;
    const bool hit = list_intersect_filtered<SC::kOnePlane, SC::kRects>(sc, ray, fast, certain);
#if !defined(__HIP_DEVICE_COMPILE__)
    g_filterStats.isect++; if (!certain) g_filterStats.isectExact++;
#endif
#if defined(VCM_FILTER_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    if (certain) {
        Isect ex = res;
        const bool eh = pairs_intersect(sc, ray, ex);
        if (eh != hit || (hit && (ex.dist != fast.dist || ex.prim != fast.prim || ex.matID != fast.matID)))
            printf("FILTER MISMATCH isect: org %.9g %.9g %.9g dir %.9g %.9g %.9g tmin %g | exact hit %d prim %d dist %.9g | fast hit %d prim %d dist %.9g\n",
                   ray.org.x, ray.org.y, ray.org.z, ray.dir.x, ray.dir.y, ray.dir.z, ray.tmin, (int)eh, ex.prim, ex.dist, (int)hit, fast.prim, fast.dist);
    }
#endif
    if (!wave_any(!certain)) { res = fast; return hit; }
#endif
    const bool exactHit = pairs_intersect(sc, ray, res);
    RC_MARK(40);
    return exactHit;
}
/* Scene::Occluded scene.hxx:72-85 (+ GeometryList::IntersectP geometry.hxx:80-91) */
template <class SC>
VCM_HD bool scene_occluded(const SC &sc, V3 point, V3 dir, float tmax)
{
    Ray ray;
    ray.org = point + dir * VCM_EPS_RAY;
    ray.dir = dir;
    ray.tmin = 0;
    const float tmaxp = tmax - 2 * VCM_EPS_RAY;
    if constexpr (SC::kBvh) return bvh_occluded(sc, ray, tmaxp);
#if !defined(VCM_NO_FILTER)
    bool certain;
    const bool occ = list_occluded_filtered<SC::kOnePlane, SC::kRects>(sc, ray, tmaxp, certain);
#if !defined(__HIP_DEVICE_COMPILE__)
    g_filterStats.occl++; if (!certain) g_filterStats.occlExact++;
#endif
#if defined(VCM_FILTER_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    if (certain && pairs_occluded(sc, ray, tmaxp) != occ)
        printf("FILTER MISMATCH occluded: org %.9g %.9g %.9g dir %.9g %.9g %.9g tmax %.9g | fast %d\n",
               ray.org.x, ray.org.y, ray.org.z, ray.dir.x, ray.dir.y, ray.dir.z, tmaxp, (int)occ);
#endif
    if (!wave_any(!certain)) return occ;
#endif
    return pairs_occluded(sc, ray, tmaxp);
}

/* ---- BSDF: bsdf.hxx:61-576 ---------------------------------------- */
enum { kDiffuse = 1, kPhong = 2, kReflect = 4, kRefract = 8, kSpecular = 12 };
struct Bsdf {
    int   matID;          /* < 0: invalid (bsdf.hxx:260) */
    Frame frame;
    V3    localDirFix;
    bool  isDelta;
    float diffProb, phongProb, reflProb, refrProb;
    float contProb;
    float reflectCoeff;
};

VCM_HD void bsdf_component_probabilities(Bsdf &b, const vcm_material &m)
{   /* GetComponentProbabilities :528-566 */
    b.reflectCoeff = fresnel_dielectric(b.localDirFix.z, m.ior);
    const float albedoDiffuse = luminance(ld3(m.diffuse));
    const float albedoPhong   = luminance(ld3(m.phong));
    const float albedoReflect = b.reflectCoeff * luminance(ld3(m.mirror));
    const float albedoRefract = (1.f - b.reflectCoeff) * (m.ior > 0.f ? 1.f : 0.f);
    const float totalAlbedo = albedoDiffuse + albedoPhong + albedoReflect + albedoRefract;
    if (totalAlbedo < 1e-9f) {
        b.diffProb = b.phongProb = b.reflProb = b.refrProb = 0.f;
        b.contProb = 0.f;
    } else {
        b.diffProb  = albedoDiffuse / totalAlbedo;
        b.phongProb = albedoPhong / totalAlbedo;
        b.reflProb  = albedoReflect / totalAlbedo;
        b.refrProb  = albedoRefract / totalAlbedo;
        b.contProb = vmax3(ld3(m.diffuse) + ld3(m.phong) + b.reflectCoeff * ld3(m.mirror)) +
                     (1.f - b.reflectCoeff);
        b.contProb = smin(1.f, smax(0.f, b.contProb));
    }
}
/* Setup :95-117.  rayDir = the incoming ray direction, normal = isect.normal, prim = isect.prim */
VCM_HD void bsdf_setup(Bsdf &b, V3 rayDir, V3 normal, int matID, int prim, const DScene &sc)
{
    (void)prim;
    b.matID = -1;
    frame_from_z(b.frame, normal);
    b.localDirFix = to_local(b.frame, -rayDir);
    if (fabsf(b.localDirFix.z) < VCM_EPS_COSINE) return;
    bsdf_component_probabilities(b, scene_material(sc, matID));
    b.isDelta = (b.diffProb == 0.f) && (b.phongProb == 0.f);
    b.matID = matID;
}
/* What a stored vertex keeps of its surface, in the upper 24 bits of the word that holds its path length: matID */
VCM_HD uint32_t shade_code(int matID, int /*prim*/) { return (uint32_t)matID & 0xffffffu; }
/* Rebuild the BSDF of a STORED vertex from (isect.normal, mLocalDirFix, matID): the same operations Setup ran,
 * hence the same bits. */
VCM_HD void bsdf_restore(Bsdf &b, V3 normal, V3 localDirFix, uint32_t code, const DScene &sc, bool ldsMaterials = true)
{
    const int matID = (int)(code & 0xffffffu);
    frame_from_z(b.frame, normal);
    b.localDirFix = localDirFix;
    bsdf_component_probabilities(b, scene_material(sc, matID, ldsMaterials));
    b.isDelta = false;
    b.matID = matID;
}
VCM_HD V3 bsdf_eval_diffuse(const Bsdf &b, const vcm_material &m, V3 gen, float *dirPdf, float *revPdf)
{   /* EvaluateDiffuse :393-412 */
    if (b.diffProb == 0.f) return sp3(0.f);
    if (b.localDirFix.z < VCM_EPS_COSINE || gen.z < VCM_EPS_COSINE) return sp3(0.f);
    if (dirPdf) *dirPdf += b.diffProb * smax(0.f, gen.z * VCM_INV_PI_F);
    if (revPdf) *revPdf += b.diffProb * smax(0.f, b.localDirFix.z * VCM_INV_PI_F);
    return ld3(m.diffuse) * VCM_INV_PI_F;
}
VCM_HD V3 bsdf_eval_phong(const Bsdf &b, const vcm_material &m, V3 gen, float *dirPdf, float *revPdf, bool intPhong = false)
{   /* EvaluatePhong :414-446 */
    if (b.phongProb == 0.f) return sp3(0.f);
    if (b.localDirFix.z < VCM_EPS_COSINE || gen.z < VCM_EPS_COSINE) return sp3(0.f);
    const V3 refl = reflect_local(b.localDirFix);
    const float dot_R_Wi = dot(refl, gen);
    if (dot_R_Wi <= VCM_EPS_PHONG) return sp3(0.f);
    /* pow(dot_R_Wi, n) is needed by the pdf (PowerCosHemispherePdfW, whose
       cosTheta = max(0, dot) == dot here) and by the value: evaluate once */
    const float pw = dm_powf_wave(dot_R_Wi, m.phongExp, true, intPhong);
    if (dirPdf || revPdf) {
        const float pdfW = b.phongProb * ((m.phongExp + 1.f) * pw * (VCM_INV_PI_F * 0.5f));
        if (dirPdf) *dirPdf += pdfW;
        if (revPdf) *revPdf += pdfW;
    }
    const V3 rho = ld3(m.phong) * (m.phongExp + 2.f) * 0.5f * VCM_INV_PI_F;
    return rho * pw;
}
VCM_HD void bsdf_pdf_diffuse(const Bsdf &b, V3 gen, float *dirPdf, float *revPdf)
{   /* PdfDiffuse :456-472 (no EPS_COSINE guard, unlike EvaluateDiffuse) */
    if (b.diffProb == 0.f) return;
    if (dirPdf) *dirPdf += b.diffProb * smax(0.f, gen.z * VCM_INV_PI_F);
    if (revPdf) *revPdf += b.diffProb * smax(0.f, b.localDirFix.z * VCM_INV_PI_F);
}
VCM_HD void bsdf_pdf_phong(const Bsdf &b, const vcm_material &m, V3 gen, float *dirPdf, float *revPdf, bool intPhong = false)
{   /* PdfPhong :474-503 */
    if (b.phongProb == 0.f) return;
    const V3 refl = reflect_local(b.localDirFix);
    const float dot_R_Wi = dot(refl, gen);
    if (dot_R_Wi <= VCM_EPS_PHONG) return;
    const float pdfW = power_cos_hemisphere_pdf(refl, gen, m.phongExp, intPhong) * b.phongProb;
    if (dirPdf) *dirPdf += pdfW;
    if (revPdf) *revPdf += pdfW;
}
/* (S = the scene as the CALLER holds it -- a kind, or plain DScene on the host: S::kIntPhong picks the pow) */
template <class S>
VCM_HD V3 bsdf_evaluate(const Bsdf &b, const S &sc, V3 worldDirGen, float &cosThetaGen,
                        float *dirPdf, float *revPdf)
{   /* Evaluate :128-153 */
    V3 result = sp3(0.f);
    if (dirPdf) *dirPdf = 0.f;
    if (revPdf) *revPdf = 0.f;
    const V3 gen = to_local(b.frame, worldDirGen);
    if (gen.z * b.localDirFix.z < 0.f) return result;
    cosThetaGen = fabsf(gen.z);
    const vcm_material m = scene_material(sc, b.matID);
    result = result + bsdf_eval_diffuse(b, m, gen, dirPdf, revPdf);
    result = result + bsdf_eval_phong(b, m, gen, dirPdf, revPdf, S::kIntPhong);
    return result;
}
template <class S>
VCM_HD float bsdf_pdf(const Bsdf &b, const S &sc, V3 worldDirGen, bool evalRev)
{   /* Pdf :161-180 */
    const V3 gen = to_local(b.frame, worldDirGen);
    if (gen.z * b.localDirFix.z < 0.f) return 0.f;
    const vcm_material m = scene_material(sc, b.matID);
    float directPdfW = 0.f, reversePdfW = 0.f;
    bsdf_pdf_diffuse(b, gen, &directPdfW, &reversePdfW);
    bsdf_pdf_phong(b, m, gen, &directPdfW, &reversePdfW, S::kIntPhong);
    return evalRev ? reversePdfW : directPdfW;
}
/* Sample :191-257 with SampleDiffuse :274, SamplePhong :290, SampleReflect :320,
 * SampleRefract :335.  fixIsLight is the reference's template argument. */
template <class S>
VCM_HD V3 bsdf_sample(const Bsdf &b, const S &sc, bool fixIsLight, float r0, float r1, float r2,
                      V3 &worldDirGen, float &pdfW, float &cosThetaGen, uint32_t &sampledEvent)
{
    if (r2 < b.diffProb) sampledEvent = kDiffuse;
    else if (r2 < b.diffProb + b.phongProb) sampledEvent = kPhong;
    else if (r2 < b.diffProb + b.phongProb + b.reflProb) sampledEvent = kReflect;
    else sampledEvent = kRefract;

    const vcm_material m = scene_material(sc, b.matID);
    pdfW = 0.f;
    V3 result = sp3(0.f);
    V3 gen = sp3(0.f);
    float sinPhi = 0.f, cosPhi = 0.f;   /* of term1 = 2 pi r0 (utils.hxx:91, :177), for the two branches that sample a lobe */
    if (sampledEvent == kDiffuse || sampledEvent == kPhong) dm_sincosf(2.f * VCM_PI_F * r0, sinPhi, cosPhi);
    RC_MARK(32);

    if (sampledEvent == kDiffuse) {
        if (b.localDirFix.z < VCM_EPS_COSINE) return sp3(0.f);
        float unweightedPdfW;
        gen = sample_cos_hemisphere_sc(sinPhi, cosPhi, r1, unweightedPdfW);
        pdfW += unweightedPdfW * b.diffProb;
        result = result + ld3(m.diffuse) * VCM_INV_PI_F;
        if (iszero(result)) return sp3(0.f);
        result = result + bsdf_eval_phong(b, m, gen, &pdfW, (float *)0, S::kIntPhong);
        RC_MARK(33);
    } else if (sampledEvent == kPhong) {
        gen = sample_power_cos_hemisphere(sinPhi, cosPhi, r1, m.phongExp);
        const V3 refl = reflect_local(b.localDirFix);
        {
            Frame fr;
            frame_from_z(fr, refl);
            gen = to_world(fr, gen);
        }
        const float dot_R_Wi = dot(refl, gen);
        if (dot_R_Wi <= VCM_EPS_PHONG) return sp3(0.f);
        /* PdfPhong(:309) and the value (:317) use the same pow */
        const float pw = dm_powf_wave(dot_R_Wi, m.phongExp, true, S::kIntPhong);
        if (b.phongProb != 0.f)
            pdfW += ((m.phongExp + 1.f) * pw * (VCM_INV_PI_F * 0.5f)) * b.phongProb;
        const V3 rho = ld3(m.phong) * (m.phongExp + 2.f) * 0.5f * VCM_INV_PI_F;
        result = result + rho * pw;
        if (iszero(result)) return sp3(0.f);
        result = result + bsdf_eval_diffuse(b, m, gen, &pdfW, (float *)0);
        RC_MARK(34);
    } else if (sampledEvent == kReflect) {
        gen = reflect_local(b.localDirFix);
        pdfW += b.reflProb;
        result = result + b.reflectCoeff * ld3(m.mirror) / fabsf(gen.z);
        if (iszero(result)) return sp3(0.f);
        RC_MARK(35);
    } else {
        if (m.ior < 0.f) return sp3(0.f);
        float cosI = b.localDirFix.z;
        float cosT, eta;
        if (cosI < 0.f) { eta = m.ior; cosI = -cosI; cosT = 1.f; }
        else            { eta = 1.f / m.ior; cosT = -1.f; }
        const float sinI2 = 1.f - cosI * cosI;
        const float sinT2 = sqr(eta) * sinI2;
        if (sinT2 < 1.f) {
            cosT *= sqrtf(smax(0.f, 1.f - sinT2));
            gen = mk3(-eta * b.localDirFix.x, -eta * b.localDirFix.y, cosT);
            pdfW += b.refrProb;
            const float refractCoeff = 1.f - b.reflectCoeff;
            if (!fixIsLight) result = result + sp3(refractCoeff * sqr(eta) / fabsf(cosT));
            else             result = result + sp3(refractCoeff / fabsf(cosT));
        } else {
            return sp3(0.f);
        }
        if (iszero(result)) return sp3(0.f);
        RC_MARK(36);
    }

    cosThetaGen = fabsf(gen.z);
    if (cosThetaGen < VCM_EPS_COSINE) return sp3(0.f);
    worldDirGen = to_world(b.frame, gen);
    return result;
}

/* ---- lights.hxx ---------------------------------------------------- */
VCM_HD bool light_is_finite(const vcm_light &l) { return l.type == VCM_LIGHT_AREA || l.type == VCM_LIGHT_POINT; }
VCM_HD bool light_is_delta(const vcm_light &l) { return l.type == VCM_LIGHT_DIRECTIONAL || l.type == VCM_LIGHT_POINT; }
VCM_HD vcm_light get_light(const DScene &sc, int idx)
{   /* Scene::GetLightPtr scene.hxx:98-102 */
    idx = (sc.nLights - 1 < idx) ? sc.nLights - 1 : idx;
    return scene_light(sc, idx);
}

VCM_HD V3 light_illuminate(const vcm_light &l, const DScene &sc, V3 recvPos, float rx, float ry,
                           V3 &dirToLight, float &distance, float &directPdfW, float &emissionPdfW,
                           float &cosAtLight)
{
    if (l.type == VCM_LIGHT_AREA) {   /* AreaLight::Illuminate :129-166 */
        float u, v;
        sample_uniform_triangle(rx, ry, u, v);
        const V3 lightPoint = ld3(l.p0) + ld3(l.e1) * u + ld3(l.e2) * v;
        dirToLight = lightPoint - recvPos;
        const float distSqr = lensqr(dirToLight);
        distance = sqrtf(distSqr);
        dirToLight = dirToLight / distance;
        const float cosNormalDir = dot(ld3(l.frameZ), -dirToLight);
        if (cosNormalDir < VCM_EPS_COSINE) return sp3(0.f);
        directPdfW = l.invArea * distSqr / cosNormalDir;
        cosAtLight = cosNormalDir;
        emissionPdfW = l.invArea * cosNormalDir * VCM_INV_PI_F;
        return ld3(l.intensity);
    } else if (l.type == VCM_LIGHT_DIRECTIONAL) {   /* :245-265 */
        dirToLight = -ld3(l.frameZ);
        distance = 1e36f;
        directPdfW = 1.f;
        cosAtLight = 1.f;
        emissionPdfW = concentric_disc_pdf_a() * sc.invSceneRadiusSqr;
        return ld3(l.intensity);
    } else if (l.type == VCM_LIGHT_POINT) {   /* :330-353 */
        dirToLight = ld3(l.p0) - recvPos;
        const float distSqr = lensqr(dirToLight);
        directPdfW = distSqr;
        distance = sqrtf(distSqr);
        dirToLight = dirToLight / distance;
        cosAtLight = 1.f;
        emissionPdfW = uniform_sphere_pdf();
        return ld3(l.intensity);
    } else {   /* BackgroundLight::Illuminate :410-437 */
        dirToLight = sample_uniform_sphere(rx, ry, directPdfW);
        const V3 radiance = ld3(l.intensity) * l.scale;
        distance = 1e36f;
        emissionPdfW = directPdfW * concentric_disc_pdf_a() * sc.invSceneRadiusSqr;
        cosAtLight = 1.f;
        return radiance;
    }
}

VCM_HD V3 light_emit(const vcm_light &l, const DScene &sc, float dx, float dy, float px, float py,
                     V3 &position, V3 &direction, float &emissionPdfW, float &directPdfA, float &cosThetaLight)
{
    if (l.type == VCM_LIGHT_AREA) {   /* AreaLight::Emit :168-198 */
        float u, v;
        sample_uniform_triangle(px, py, u, v);
        position = ld3(l.p0) + ld3(l.e1) * u + ld3(l.e2) * v;
        V3 localDirOut = sample_cos_hemisphere(dx, dy, emissionPdfW);
        emissionPdfW *= l.invArea;
        localDirOut.z = smax(localDirOut.z, VCM_EPS_COSINE);
        Frame f; f.mX = ld3(l.frameX); f.mY = ld3(l.frameY); f.mZ = ld3(l.frameZ);
        direction = to_world(f, localDirOut);
        directPdfA = l.invArea;
        cosThetaLight = localDirOut.z;
        return ld3(l.intensity) * localDirOut.z;
    } else if (l.type == VCM_LIGHT_DIRECTIONAL) {   /* :267-294 */
        float x, y;
        sample_concentric_disc(px, py, x, y);
        position = ld3(sc.sceneCenter) + sc.sceneRadius * (-ld3(l.frameZ) + ld3(l.frameX) * x + ld3(l.frameY) * y);
        direction = ld3(l.frameZ);
        emissionPdfW = concentric_disc_pdf_a() * sc.invSceneRadiusSqr;
        directPdfA = 1.f;
        cosThetaLight = 1.f;
        return ld3(l.intensity);
    } else if (l.type == VCM_LIGHT_POINT) {   /* :355-376 */
        position = ld3(l.p0);
        direction = sample_uniform_sphere(dx, dy, emissionPdfW);
        directPdfA = 1.f;
        cosThetaLight = 1.f;
        return ld3(l.intensity);
    } else {   /* BackgroundLight::Emit :439-481 */
        float directPdf;
        direction = sample_uniform_sphere(dx, dy, directPdf);
        const V3 radiance = ld3(l.intensity) * l.scale;
        float x, y;
        sample_concentric_disc(px, py, x, y);
        Frame frame;
        frame_from_z(frame, direction);
        position = ld3(sc.sceneCenter) + sc.sceneRadius * (-direction + frame.mX * x + frame.mY * y);
        emissionPdfW = directPdf * concentric_disc_pdf_a() * sc.invSceneRadiusSqr;
        directPdfA = directPdf;
        cosThetaLight = 1.f;
        return radiance;
    }
}

VCM_HD V3 light_get_radiance(const vcm_light &l, const DScene &sc, V3 rayDir,
                             float &directPdfA, float &emissionPdfW)
{
    if (l.type == VCM_LIGHT_AREA) {   /* :200-221 */
        const float cosOutL = smax(0.f, dot(ld3(l.frameZ), -rayDir));
        if (cosOutL == 0.f) return sp3(0.f);
        directPdfA = l.invArea;
        emissionPdfW = cos_hemisphere_pdf(ld3(l.frameZ), -rayDir);
        emissionPdfW *= l.invArea;
        return ld3(l.intensity);
    } else if (l.type == VCM_LIGHT_BACKGROUND) {   /* :483-504 */
        const float directPdf = uniform_sphere_pdf();
        const V3 radiance = ld3(l.intensity) * l.scale;
        const float positionPdf = concentric_disc_pdf_a() * sc.invSceneRadiusSqr;
        directPdfA = directPdf;
        emissionPdfW = directPdf * positionPdf;
        return radiance;
    }
    return sp3(0.f);   /* directional :296-304, point :378-386 */
}

/* ------------------------------------------------------------------ */
/* sub-path state: vertexcm.hxx:64-76                                   */
struct SubPathState {
    V3 origin, direction, throughput;
    uint32_t pathLength;
    uint32_t isFiniteLight;
    uint32_t specularPath;
    float dVCM, dVC, dVM;
};
VCM_HD float mis(float pdf) { return pdf; }   /* :553-557 (balance heuristic) */

/* SampleScattering<tLightSample> :938-1006 */
template <class S>
VCM_HD bool sample_scattering(const S &sc, const IterParams &P, bool lightSample, PathRng &rng,
                              const Bsdf &bsdf, V3 hitPoint, SubPathState &st)
{
    /* the 3 floats of BSDF::Sample (:944) and the Russian-roulette float (:964), which only counts as drawn if
       the sample is non-zero: 4 consecutive floats, generated at one site */
    float rnd[4];
    RC_DECL;
    rng_peek(rng, rng.k, rnd, 4);
    RC_MARK(lightSample ? 23 : 27);
    rng.k += 3u;
    const float r0 = rnd[0], r1 = rnd[1], r2 = rnd[2];
    float bsdfDirPdfW, cosThetaOut;
    uint32_t sampledEvent;
    const V3 bsdfFactor = bsdf_sample(bsdf, sc, lightSample, r0, r1, r2, st.direction, bsdfDirPdfW, cosThetaOut,
                                      sampledEvent);
    RC_MARK(lightSample ? 24 : 28);
    if (iszero(bsdfFactor)) return false;
    float bsdfRevPdfW = bsdfDirPdfW;
    if ((sampledEvent & kSpecular) == 0) bsdfRevPdfW = bsdf_pdf(bsdf, sc, st.direction, true);
    RC_MARK(lightSample ? 25 : 29);
    const float contProb = bsdf.contProb;
    rng.k += 1u;
    if (rnd[3] > contProb) return false;
    bsdfDirPdfW *= contProb;
    bsdfRevPdfW *= contProb;
    if (sampledEvent & kSpecular) {
        st.dVCM = 0.f;
        st.dVC *= mis(cosThetaOut);
        st.dVM *= mis(cosThetaOut);
        st.specularPath &= 1;
    } else {
        st.dVC = mis(cosThetaOut / bsdfDirPdfW) * (st.dVC * mis(bsdfRevPdfW) + st.dVCM + P.misVmWeightFactor);
        st.dVM = mis(cosThetaOut / bsdfDirPdfW) * (st.dVM * mis(bsdfRevPdfW) + st.dVCM * P.misVcWeightFactor + 1.f);
        st.dVCM = mis(1.f / bsdfDirPdfW);
        st.specularPath &= 0;
    }
    st.origin = hitPoint;
    st.throughput = st.throughput * (bsdfFactor * (cosThetaOut / bsdfDirPdfW));
    RC_MARK(lightSample ? 26 : 30);
    return true;
}

/* ================= light sub-path (vertexcm.hxx:321-396) ============== */

/* GenerateLightSample :816-858 */
VCM_HD void generate_light_sample(const DScene &sc, const IterParams &P, PathRng &rng, SubPathState &st)
{
    const int lightCount = sc.nLights;
    const float lightPickProb = 1.f / lightCount;
    float rnd[5];   /* :822-824 */
    rng_peek(rng, rng.k, rnd, 5);
    rng.k += 5u;
    const int lightID = int(rnd[0] * lightCount);
    const float dx = rnd[1];
    const float dy = rnd[2];
    const float px = rnd[3];
    const float py = rnd[4];
    const vcm_light &light = get_light(sc, lightID);
    float emissionPdfW, directPdfA, cosLight;
    st.throughput = light_emit(light, sc, dx, dy, px, py, st.origin, st.direction, emissionPdfW, directPdfA, cosLight);
    emissionPdfW *= lightPickProb;
    directPdfA *= lightPickProb;
    st.throughput = st.throughput / emissionPdfW;
    st.pathLength = 1;
    st.isFiniteLight = light_is_finite(light) ? 1u : 0u;
    st.specularPath = 0;
    st.dVCM = mis(directPdfA / emissionPdfW);
    if (!light_is_delta(light)) {
        const float usedCosLight = light_is_finite(light) ? cosLight : 1.f;
        st.dVC = mis(usedCosLight / emissionPdfW);
    } else {
        st.dVC = 0.f;
    }
    st.dVM = st.dVC * P.misVcWeightFactor;
}

/* ConnectToCamera :862-933; the splat is an atomic add (Framebuffer::AddColor
 * framebuffer.hxx:43-57 on an arbitrary pixel) */
/* splatOut == NULL: the splat is an fp32 atomic add on fb (strict mode);
 * otherwise *splatOut receives (contrib.rgb, pixel) -- pixel -1 when nothing is
 * splatted -- and k_splat_apply adds the splats of a pixel in vertex order. */
template <class SC>
VCM_HD void connect_to_camera(const SC &sc, const IterParams &P, const SubPathState &st, V3 hitpoint,
                              const Bsdf &bsdf, float *fb, LaneStats &ls, F4 *splatOut = 0)
{
    if (splatOut) *splatOut = mk4(0.f, 0.f, 0.f, u2f(0xffffffffu));
    RC_DECL;
    const vcm_camera &cam = sc.camera;
    V3 directionToCamera = ld3(cam.position) - hitpoint;
    if (dot(ld3(cam.forward), -directionToCamera) <= 0.f) return;
    const V3 ip = transform_point(cam.worldToRaster, hitpoint);
    if (!(ip.x >= 0 && ip.y >= 0 && ip.x < cam.resolution[0] && ip.y < cam.resolution[1])) return;
    const float distEye2 = lensqr(directionToCamera);
    const float distance = sqrtf(distEye2);
    directionToCamera = directionToCamera / distance;
    float cosToCamera, bsdfDirPdfW, bsdfRevPdfW;
    const V3 bsdfFactor = bsdf_evaluate(bsdf, sc, directionToCamera, cosToCamera, &bsdfDirPdfW, &bsdfRevPdfW);
    if (iszero(bsdfFactor)) return;
    bsdfRevPdfW *= bsdf.contProb;
    const float cosAtCamera = dot(ld3(cam.forward), -directionToCamera);
    const float imagePointToCameraDist = cam.imagePlaneDist / cosAtCamera;
    const float imageToSolidAngleFactor = sqr(imagePointToCameraDist) / cosAtCamera;
    const float imageToSurfaceFactor = imageToSolidAngleFactor * fabsf(cosToCamera) / sqr(distance);
    const float cameraPdfA = imageToSurfaceFactor;
    const float wLight = mis(cameraPdfA / P.lightSubPathCount) *
                         (P.misVmWeightFactor + st.dVCM + st.dVC * mis(bsdfRevPdfW));
    const float misWeight = P.lightTraceOnly ? 1.f : (1.f / (wLight + 1.f));
    const float surfaceToImageFactor = 1.f / imageToSurfaceFactor;
    const V3 contrib = misWeight * st.throughput * bsdfFactor / (P.lightSubPathCount * surfaceToImageFactor);
    RC_MARK(17);
    if (!iszero(contrib)) {
        ls.shadowRays++;
        const bool occluded = scene_occluded(sc, hitpoint, directionToCamera, distance);
        RC_MARK(18);
        if (occluded) return;
        const int x = int(ip.x), y = int(ip.y);
        if (splatOut) {
            *splatOut = mk4(contrib.x, contrib.y, contrib.z, u2f((uint32_t)(x + y * P.resX)));
        } else {
            float *px = fb + (size_t)(x + y * P.resX) * 3;
            fb_atomic_add(px + 0, contrib.x);
            fb_atomic_add(px + 1, contrib.y);
            fb_atomic_add(px + 2, contrib.z);
        }
        ls.lightSplats++;
    }
}

struct LightPath {
    SubPathState st;
    PathRng rng;
    int lp;          /* local path index */
    int nStored;
    uint32_t lenMask;   /* see LightStore::lenMask */
};

/* bounding box of the vertices a lane has stored (HashGrid::Build takes it over all of them, hashgrid.hxx:50-61):
 * K1 keeps it while it writes them, so that the grid build does not read every position again */
struct LaneBox { float mn[3], mx[3]; };
VCM_HD void lane_box_init(LaneBox &b) { for (int k = 0; k < 3; k++) { b.mn[k] = 3.0e38f; b.mx[k] = -3.0e38f; } }

VCM_HD void light_path_begin(const DScene &sc, const IterParams &P, LightPath &lp, int localPath)
{
    lp.lp = localPath;
    lp.nStored = 0;
    lp.lenMask = 0u;
    rng_init(lp.rng, P.seed, P.localIter, (uint32_t)(P.p0 + localPath), 0u);
    generate_light_sample(sc, P, lp.rng, lp.st);
}

/* one iteration of the for(;;) at :328-393; returns false when the path ends.
 * MODE 1 (wavefront): ConnectToCamera (:380-384) is left to k_connect_camera,
 * which runs it for every stored vertex. */
template <int MODE, class SC>
VCM_HD bool light_path_step(const SC &sc, const IterParams &P, LightPath &lp, const LightStore &store,
                            float *fb, LaneStats &ls, LaneBox &box)
{
    SubPathState &st = lp.st;
    Ray ray; ray.org = st.origin + st.direction * VCM_EPS_RAY; ray.dir = st.direction; ray.tmin = 0;
    Isect isect; isect.dist = 1e36f; isect.matID = 0; isect.lightID = -1; isect.normal = sp3(0.f); isect.prim = -1;
    ls.lightRays++;
    RC_DECL;
    const bool hitSomething = scene_intersect(sc, ray, isect);
    RC_MARK(0);
    if (!hitSomething) return false;
    const V3 hitPoint = ray.org + ray.dir * isect.dist;
    isect.dist += VCM_EPS_RAY;
    Bsdf bsdf;
    bsdf_setup(bsdf, ray.dir, isect.normal, isect.matID, isect.prim, sc);
    if (bsdf.matID < 0) return false;
    {   /* :351-360 */
        if (st.pathLength > 1 || st.isFiniteLight == 1) st.dVCM *= mis(sqr(isect.dist));
        st.dVCM /= mis(fabsf(bsdf.localDirFix.z));
        st.dVC  /= mis(fabsf(bsdf.localDirFix.z));
        st.dVM  /= mis(fabsf(bsdf.localDirFix.z));
    }
    RC_MARK(1);
    if (!bsdf.isDelta && (P.useVC || P.useVM || (MODE == 1 && P.lightTraceOnly))) {   /* :364-377 */
        const size_t slot = (size_t)lp.nStored * (size_t)P.nLocal + (size_t)lp.lp;
        lv(store, slot, 0) = mk4(hitPoint.x, hitPoint.y, hitPoint.z, u2f(st.pathLength | (shade_code(bsdf.matID, isect.prim) << 8)));
        lv(store, slot, 1) = mk4(st.throughput.x, st.throughput.y, st.throughput.z, st.dVCM);
        lv(store, slot, 2) = mk4(isect.normal.x, isect.normal.y, isect.normal.z, st.dVC);
        lv(store, slot, 3) = mk4(bsdf.localDirFix.x, bsdf.localDirFix.y, bsdf.localDirFix.z, st.dVM);
        lp.nStored++;
        box.mn[0] = fminf(box.mn[0], hitPoint.x); box.mn[1] = fminf(box.mn[1], hitPoint.y); box.mn[2] = fminf(box.mn[2], hitPoint.z);
        box.mx[0] = fmaxf(box.mx[0], hitPoint.x); box.mx[1] = fmaxf(box.mx[1], hitPoint.y); box.mx[2] = fmaxf(box.mx[2], hitPoint.z);
        lp.lenMask |= (st.pathLength < 32u) ? (1u << st.pathLength) : 0u;
        if (P.useVC || P.useVM) ls.stored++;   /* the reference stores nothing in light-trace mode (:364) */
    }
    if (MODE == 0 && !bsdf.isDelta && (P.useVC || P.lightTraceOnly)) {   /* :380-384 */
        if (st.pathLength + 1 >= P.minLen) connect_to_camera(sc, P, st, hitPoint, bsdf, fb, ls);
    }
    RC_MARK(2);
    if (st.pathLength + 2 > P.maxLen) return false;   /* :387 */
    const bool goesOn = sample_scattering(sc, P, true, lp.rng, bsdf, hitPoint, st);
    RC_MARK(3);
    if (!goesOn) return false;
    ++st.pathLength;
    return true;
}

/* WorldDirFix() | ContinuationProb() of a STORED light vertex (bsdf.hxx:260-264; what RangeQuery::Process reads of the light
 * BSDF, vertexcm.hxx:140, :161), rebuilt from the record: Setup's frame from the stored normal, ToWorld of the stored
 * mLocalDirFix, GetComponentProbabilities of the stored material -- the operations the light pass ran, on the same values,
 * hence the same bits. */
VCM_HD F4 light_vertex_wdir_contprob(const DScene &sc, F4 field0, F4 field2, F4 field3, bool ldsMaterials)
{
    Bsdf b;
    bsdf_restore(b, mk3(field2.x, field2.y, field2.z), mk3(field3.x, field3.y, field3.z), f2u(field0.w) >> 8, sc, ldsMaterials);
    const V3 w = to_world(b.frame, b.localDirFix);
    return mk4(w.x, w.y, w.z, b.contProb);
}

/* ConnectToCamera (:380-384, :862-933) for a STORED light vertex (wavefront mode) */
template <class SC>
VCM_HD void connect_stored_vertex_to_camera(const SC &sc, const IterParams &P, const LightStore &store,
                                            size_t slot, float *fb, LaneStats &ls, F4 *splatOut)
{
    const F4 a = lv(store, slot, 0), b = lv(store, slot, 1), c = lv(store, slot, 2), d = lv(store, slot, 3);
    SubPathState st;
    st.pathLength = f2u(a.w) & 0xffu;
    *splatOut = mk4(0.f, 0.f, 0.f, u2f(0xffffffffu));
    if (!(st.pathLength + 1 >= P.minLen)) return;
    st.throughput = mk3(b.x, b.y, b.z);
    st.dVCM = b.w; st.dVC = c.w; st.dVM = d.w;
    Bsdf bsdf;
    bsdf_restore(bsdf, mk3(c.x, c.y, c.z), mk3(d.x, d.y, d.z), f2u(a.w) >> 8, sc);
    connect_to_camera(sc, P, st, mk3(a.x, a.y, a.z), bsdf, fb, ls, splatOut);
}

/* ================= camera sub-path (vertexcm.hxx:415-545) ============= */

/* GetLightRadiance :617-658 */
VCM_HD V3 get_light_radiance(const DScene &sc, const IterParams &P, const vcm_light &light,
                             const SubPathState &st, V3 rayDir)
{
    const int lightCount = sc.nLights;
    const float lightPickProb = 1.f / lightCount;
    float directPdfA = 0.f, emissionPdfW = 0.f;
    const V3 radiance = light_get_radiance(light, sc, rayDir, directPdfA, emissionPdfW);
    if (iszero(radiance)) return sp3(0.f);
    if (st.pathLength == 1) return radiance;
    if (P.useVM && !P.useVC) return st.specularPath ? radiance : sp3(0.f);
    directPdfA *= lightPickProb;
    emissionPdfW *= lightPickProb;
    const float wCamera = mis(directPdfA) * st.dVCM + mis(emissionPdfW) * st.dVC;
    const float misWeight = 1.f / (1.f + wCamera);
    return misWeight * radiance;
}

/* DirectIllumination :663-738 */
template <class SC>
VCM_HD V3 direct_illumination(const SC &sc, const IterParams &P, float rPick, float rx, float ry,
                              const SubPathState &st, V3 hitpoint, const Bsdf &bsdf, LaneStats &ls)
{   /* rPick, rx, ry: the three floats drawn at :672-673 */
    RC_DECL;
    const int lightCount = sc.nLights;
    const float lightPickProb = 1.f / lightCount;
    const int lightID = int(rPick * lightCount);
    const vcm_light &light = get_light(sc, lightID);
    V3 directionToLight;
    float distance, directPdfW, emissionPdfW, cosAtLight;
    const V3 radiance = light_illuminate(light, sc, hitpoint, rx, ry, directionToLight, distance, directPdfW,
                                         emissionPdfW, cosAtLight);
    if (iszero(radiance)) return sp3(0.f);
    float bsdfDirPdfW, bsdfRevPdfW, cosToLight;
    const V3 bsdfFactor = bsdf_evaluate(bsdf, sc, directionToLight, cosToLight, &bsdfDirPdfW, &bsdfRevPdfW);
    if (iszero(bsdfFactor)) return sp3(0.f);
    const float continuationProbability = bsdf.contProb;
    bsdfDirPdfW *= light_is_delta(light) ? 0.f : continuationProbability;
    bsdfRevPdfW *= continuationProbability;
    const float wLight = mis(bsdfDirPdfW / (lightPickProb * directPdfW));
    const float wCamera = mis(emissionPdfW * cosToLight / (directPdfW * cosAtLight)) *
                          (P.misVmWeightFactor + st.dVCM + st.dVC * mis(bsdfRevPdfW));
    const float misWeight = 1.f / (wLight + 1.f + wCamera);
    const V3 contrib = (misWeight * cosToLight / (lightPickProb * directPdfW)) * (radiance * bsdfFactor);
    RC_MARK(9);
    if (iszero(contrib)) return sp3(0.f);
    ls.shadowRays++;
    const bool occluded = scene_occluded(sc, hitpoint, directionToLight, distance);
    RC_MARK(10);
    if (occluded) return sp3(0.f);
    return contrib;
}

/* ConnectVertices :743-809; the light vertex comes from the LightStore */
template <class SC>
VCM_HD V3 connect_vertices(const SC &sc, const IterParams &P, V3 lvHitpoint, const Bsdf &lvBsdf,
                           float lvdVCM, float lvdVC, const Bsdf &cameraBsdf, V3 cameraHitpoint,
                           const SubPathState &st, LaneStats &ls)
{
    ls.connections++;
    RC_DECL;
    V3 direction = lvHitpoint - cameraHitpoint;
    const float dist2 = lensqr(direction);
    const float distance = sqrtf(dist2);
    direction = direction / distance;
    float cosCamera, cameraBsdfDirPdfW, cameraBsdfRevPdfW;
    const V3 cameraBsdfFactor = bsdf_evaluate(cameraBsdf, sc, direction, cosCamera, &cameraBsdfDirPdfW, &cameraBsdfRevPdfW);
    if (iszero(cameraBsdfFactor)) return sp3(0.f);
    const float cameraCont = cameraBsdf.contProb;
    cameraBsdfDirPdfW *= cameraCont;
    cameraBsdfRevPdfW *= cameraCont;
    float cosLight, lightBsdfDirPdfW, lightBsdfRevPdfW;
    const V3 lightBsdfFactor = bsdf_evaluate(lvBsdf, sc, -direction, cosLight, &lightBsdfDirPdfW, &lightBsdfRevPdfW);
    if (iszero(lightBsdfFactor)) return sp3(0.f);
    const float lightCont = lvBsdf.contProb;
    lightBsdfDirPdfW *= lightCont;
    lightBsdfRevPdfW *= lightCont;
    const float geometryTerm = cosLight * cosCamera / dist2;
    if (geometryTerm < 0.f) return sp3(0.f);
    const float cameraBsdfDirPdfA = pdf_w_to_a(cameraBsdfDirPdfW, distance, cosLight);
    const float lightBsdfDirPdfA = pdf_w_to_a(lightBsdfDirPdfW, distance, cosCamera);
    const float wLight = mis(cameraBsdfDirPdfA) * (P.misVmWeightFactor + lvdVCM + lvdVC * mis(lightBsdfRevPdfW));
    const float wCamera = mis(lightBsdfDirPdfA) * (P.misVmWeightFactor + st.dVCM + st.dVC * mis(cameraBsdfRevPdfW));
    const float misWeight = 1.f / (wLight + 1.f + wCamera);
    const V3 contrib = (misWeight * geometryTerm) * cameraBsdfFactor * lightBsdfFactor;
    RC_MARK(12);
    if (iszero(contrib)) return sp3(0.f);
    ls.shadowRays++;
    const bool occluded = scene_occluded(sc, cameraHitpoint, direction, distance);
    RC_MARK(13);
    if (occluded) return sp3(0.f);
    return contrib;
}

/* HashGrid::GetCellIndex(Vec3i) hashgrid.hxx:179-187 */
VCM_HD int grid_cell_hash(int cx, int cy, int cz, int nCells)
{
    const uint32_t x = (uint32_t)cx, y = (uint32_t)cy, z = (uint32_t)cz;
    return (int)(((x * 73856093u) ^ (y * 19349663u) ^ (z * 83492791u)) % (uint32_t)nCells);
}
/* HashGrid::GetCellIndex(Vec3f) :189-201 */
VCM_HD int grid_cell_of_point(V3 p, V3 bboxMin, float invCellSize, int nCells)
{
    const V3 distMin = p - bboxMin;
    const float fx = floorf(invCellSize * distMin.x);
    const float fy = floorf(invCellSize * distMin.y);
    const float fz = floorf(invCellSize * distMin.z);
    return grid_cell_hash(int(fx), int(fy), int(fz), nCells);
}

/* Per-lane queue of accepted photon indices, in LDS on the device:
 * entry k of this lane is q[k * stride], k = 0..VCM_MERGE_Q (one spare row: the
 * scan writes every candidate at the tail and only advances it on acceptance). */
#ifndef VCM_MERGE_Q
#define VCM_MERGE_Q 32   /* 16 -> 32: the wave drains when ONE lane is nearly full; deeper queues even out the lanes (-2.8 % K4) */
#endif
#define VCM_MERGE_UNROLL 4
struct MergeScratch { uint32_t *q; int stride; int cap; /* entries per lane (+1 spare row) */ };

/* Everything of RangeQuery::Process (vertexcm.hxx:130-169) and of the camera
 * BSDF::Evaluate (bsdf.hxx:128-153, :393-446) that does not depend on the
 * photon, computed once per query.  The per-photon part below performs the
 * remaining operations of the reference in the same order on the same values,
 * so the sum is bit-identical to evaluating bsdf_evaluate() per
 * photon (the material would otherwise be re-fetched with a lane-varying index
 * for every accepted photon). */
struct MergeEval {
    Frame frame;
    V3 refl;              /* ReflectLocal(mLocalDirFix), bsdf.hxx:423 */
    V3 diffuseVal;        /* mDiffuseReflectance * INV_PI_F, :411 */
    V3 rho;               /* mPhongReflectance * (n+2) * 0.5 * INV_PI_F, :442-443 */
    float ldfz;
    float diffProb, phongProb, phongExp;
    float revPdfDiffuse;  /* diffProb * max(0, mLocalDirFix.z * INV_PI_F), :408 */
    float camContProb;
    float camTerm;        /* mCameraState.dVCM * mMisVcWeightFactor, vertexcm.hxx:160 */
    float camdVM;
    uint32_t pathLength;
    bool cosOk;           /* !(mLocalDirFix.z < EPS_COSINE) */
};
VCM_HD void merge_eval_setup(MergeEval &e, const DScene &sc, const IterParams &P, const Bsdf &b,
                             const SubPathState &st, bool ldsMaterials = true)
{
    const vcm_material m = scene_material(sc, b.matID, ldsMaterials);
    e.frame = b.frame;
    e.refl = reflect_local(b.localDirFix);
    e.diffuseVal = ld3(m.diffuse) * VCM_INV_PI_F;
    e.rho = ld3(m.phong) * (m.phongExp + 2.f) * 0.5f * VCM_INV_PI_F;
    e.ldfz = b.localDirFix.z;
    e.diffProb = b.diffProb; e.phongProb = b.phongProb; e.phongExp = m.phongExp;
    e.revPdfDiffuse = b.diffProb * smax(0.f, b.localDirFix.z * VCM_INV_PI_F);
    e.camContProb = b.contProb;
    e.camTerm = st.dVCM * P.misVcWeightFactor;
    e.camdVM = st.dVM;
    e.pathLength = st.pathLength;
    e.cosOk = !(b.localDirFix.z < VCM_EPS_COSINE);
}
template <bool IP /* the scene's Phong exponents are integers (DScene::kIntPhong) */>
VCM_HD void merge_eval_photon(const MergeEval &e, const IterParams &P, uint32_t lvLen, V3 lightDirection,
                              float lvContProb, V3 lvThroughput, float lvdVCM, float lvdVM, V3 &contrib)
{
    /* written with selects instead of early returns: on the GPU every early
       return is an exec-mask save/branch/restore; the values and the order of
       the operations are those of the reference */
    const V3 gen = to_local(e.frame, lightDirection);                                   /* bsdf.hxx:140 */
    const bool valid = !((lvLen + e.pathLength > P.maxLen) || (lvLen + e.pathLength < P.minLen))   /* :133-135 */
                       && !(gen.z * e.ldfz < 0.f);                                      /* bsdf.hxx:142 */
    const bool ok = e.cosOk && !(gen.z < VCM_EPS_COSINE);                               /* :402, :423 */
    const bool dOn = valid && ok && (e.diffProb != 0.f);
    /* EvaluateDiffuse: the pdfs start at 0, "0 + x" with x >= 0 is x */
    float dirPdf = dOn ? e.diffProb * smax(0.f, gen.z * VCM_INV_PI_F) : 0.f;
    float revPdf = dOn ? e.revPdfDiffuse : 0.f;
    V3 result = sp3(0.f) + (dOn ? e.diffuseVal : sp3(0.f));
    /* EvaluatePhong */
    const float dot_R_Wi = dot(e.refl, gen);
    V3 ph = sp3(0.f);
    if (valid && ok && (e.phongProb != 0.f) && !(dot_R_Wi <= VCM_EPS_PHONG)) {
        const float pw = dm_powf_wave(dot_R_Wi, e.phongExp, false, IP);   /* the merge kernels stage no tables */
        const float pdfW = e.phongProb * ((e.phongExp + 1.f) * pw * (VCM_INV_PI_F * 0.5f));
        dirPdf += pdfW;
        revPdf += pdfW;
        ph = e.rho * pw;
    }
    result = result + ph;
    if (valid && !iszero(result)) {                                                     /* :145-146 */
        dirPdf *= e.camContProb;                                                        /* :148 */
        revPdf *= lvContProb;                                                           /* :153 */
        const float wLight = lvdVCM * P.misVcWeightFactor + lvdVM * mis(dirPdf);        /* :156-157 */
        const float wCamera = e.camTerm + e.camdVM * mis(revPdf);                       /* :160-161 */
        const float misWeight = P.ppm ? 1.f : 1.f / (wLight + 1.f + wCamera);           /* :164-166 */
        contrib = contrib + misWeight * result * lvThroughput;                          /* :168 */
    }
}

struct MergePhoton { float lenBits; F4 b, c; float dVM; };
VCM_HD void merge_photon_load(const GridStore &g, const MergeScratch &ms, int k, int qn, MergePhoton &p)
{
    /* lanes without an entry k read photon 0 (always allocated); the value is not used */
    const uint32_t idx = (k < qn) ? ms.q[k * ms.stride] : 0u;
    const F2 t = g.g3[idx];
    p.lenBits = t.y;
    p.b = g.g1[idx];
    p.c = g.g2[idx];
    p.dVM = t.x;
}
template <bool IP>
VCM_HD void merge_drain(const IterParams &P, const GridStore &g, const MergeEval &e, const MergeScratch &ms, int qn,
                        V3 &contrib)
{
    /* software-pipelined: the loads of entry k+1 are in flight while entry k is evaluated.  Unrolled by two with the two
       register sets taking turns (round 4): as a rotating pair the compiler copied the eleven registers of `nxt` into `cur`
       every step, 10 of the step's ~150 instructions (profiles/archive/r06n_ab.txt) */
    if (!wave_any(0 < qn)) return;
    MergePhoton pa, pb;
    merge_photon_load(g, ms, 0, qn, pa);
    for (int k = 0; k < ms.cap; k += 2) {
        const bool more1 = wave_any(k + 1 < qn);
        if (more1) merge_photon_load(g, ms, k + 1, qn, pb);
        if (k < qn)
            merge_eval_photon<IP>(e, P, f2u(pa.lenBits), mk3(pa.b.x, pa.b.y, pa.b.z), pa.b.w,
                                  mk3(pa.c.x, pa.c.y, pa.c.z), pa.c.w, pa.dVM, contrib);
        if (!more1) break;
        const bool more2 = wave_any(k + 2 < qn);
        if (more2) merge_photon_load(g, ms, k + 2, qn, pa);
        if (k + 1 < qn)
            merge_eval_photon<IP>(e, P, f2u(pb.lenBits), mk3(pb.b.x, pb.b.y, pb.b.z), pb.b.w,
                                  mk3(pb.c.x, pb.c.y, pb.c.z), pb.c.w, pb.dVM, contrib);
        if (!more2) break;
    }
}

/* HashGrid::Process hashgrid.hxx:110-169: the 8 hashed cells toward the
 * nearer faces, in the reference's order (duplicates included, :142-155).
 *
 * GPU shape: the cheap part (distance test, :162-165) and the expensive part
 * (RangeQuery::Process, a BSDF evaluation) are separated.  All lanes of the
 * wave walk the 8 cells in lockstep, each testing up to 4 candidates of ITS
 * cell range per step with the 4 loads in flight together (the vertices of a
 * cell are contiguous: cell-sorted SoA); accepted indices go to a per-lane
 * LDS queue.  When some lane's queue is nearly full the whole wave drains:
 * every lane evaluates its queued photons, so the evaluation runs with most
 * lanes active instead of the ~18 % that accept at any one candidate.  Each
 * lane still processes ITS photons in the reference's order, so the sum
 * (:168) is bit-identical. */
template <bool IP>
VCM_HD V3 merge_query(const DScene &sc, const IterParams &P, const GridStore &g, const Bsdf &cameraBsdf,
                      const SubPathState &st, V3 queryPos, LaneStats &ls, const MergeScratch &ms, bool ldsMaterials = true)
{
    V3 contrib = sp3(0.f);
    const V3 bmin = ld3(g.hdr->bboxMin), bmax = ld3(g.hdr->bboxMax);
    const V3 distMin = queryPos - bmin;
    const V3 distMax = bmax - queryPos;
    const bool inside = !(distMin.x < 0.f || distMax.x < 0.f || distMin.y < 0.f || distMax.y < 0.f ||
                          distMin.z < 0.f || distMax.z < 0.f);   /* :116-122 */
    const V3 cellPt = P.invCellSize * distMin;
    const V3 coordF = mk3(floorf(cellPt.x), floorf(cellPt.y), floorf(cellPt.z));
    const int px = int(coordF.x), py = int(coordF.y), pz = int(coordF.z);
    const V3 fractCoord = cellPt - coordF;
    const int pxo = px + (fractCoord.x < 0.5f ? -1 : +1);
    const int pyo = py + (fractCoord.y < 0.5f ? -1 : +1);
    const int pzo = pz + (fractCoord.z < 0.5f ? -1 : +1);
    MergeEval ev;
    merge_eval_setup(ev, sc, P, cameraBsdf, st, ldsMaterials);
    int qn = 0;
    /* the range of cell j+1 is fetched while cell j is scanned: one dependent memory round trip per cell less */
    int nlo = 0, nhi = 0;
    if (inside) {
        const int cell = grid_cell_hash(px, py, pz, P.nCells);
        nlo = g.cellStart[cell];
        nhi = g.cellStart[cell + 1];
    }
    for (int j = 0; j < 8; j++) {
        int lo = nlo, hi = nhi;
        if (inside && j < 7) {
            const int k = j + 1;
            const int cx = (k & 4) ? pxo : px;
            const int cy = (k & 2) ? pyo : py;
            const int cz = (k & 1) ? pzo : pz;
            const int cell = grid_cell_hash(cx, cy, cz, P.nCells);
            nlo = g.cellStart[cell];
            nhi = g.cellStart[cell + 1];
        }
        ls.mergeCandidates += (uint32_t)(hi - lo);   /* one distance test per entry (:162-165) */
        /* entries past hi are read but never used (the arrays are padded by VCM_MERGE_UNROLL elements), so one
           address serves all 4 candidates of a step.  LenSqr of (query - position), hashgrid.hxx:162, math.hxx:107. */
#if defined(__HIP_DEVICE_COMPILE__)
        /* 3 x 16-byte loads per step, software-pipelined: the candidates of step s+1 are in flight while step s
           is tested; two candidates per packed operation (IEEE per half, same operation order) */
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        const f2 qx = f2_sp(queryPos.x), qy = f2_sp(queryPos.y), qz = f2_sp(queryPos.z);
        f4u X = *(const f4u *)(g.gx + lo), Y = *(const f4u *)(g.gy + lo), Z = *(const f4u *)(g.gz + lo);
#endif
        while (wave_any(lo < hi)) {
            const int nextLo = (lo + VCM_MERGE_UNROLL < hi) ? lo + VCM_MERGE_UNROLL : hi;
            float distSqr[VCM_MERGE_UNROLL];
#if defined(__HIP_DEVICE_COMPILE__)
            const f4u Xn = *(const f4u *)(g.gx + nextLo), Yn = *(const f4u *)(g.gy + nextLo), Zn = *(const f4u *)(g.gz + nextLo);
            {
                const f2 dxa = qx - X.xy, dya = qy - Y.xy, dza = qz - Z.xy;
                const f2 dxb = qx - X.zw, dyb = qy - Y.zw, dzb = qz - Z.zw;
                const f2 da = dxa * dxa + dya * dya + dza * dza;
                const f2 db = dxb * dxb + dyb * dyb + dzb * dzb;
                distSqr[0] = da.x; distSqr[1] = da.y; distSqr[2] = db.x; distSqr[3] = db.y;
            }
            X = Xn; Y = Yn; Z = Zn;
#else
            for (int u = 0; u < VCM_MERGE_UNROLL; u++)
                distSqr[u] = lensqr(queryPos - mk3(g.gx[lo + u], g.gy[lo + u], g.gz[lo + u]));
#endif
            /* branch-free: every candidate writes its index at the queue tail, only an
               accepted one advances the tail (qn <= VCM_MERGE_Q, row VCM_MERGE_Q is spare) */
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int u = 0; u < VCM_MERGE_UNROLL; u++) {
                const int idx = lo + u;
                const bool acc = (idx < hi) & (distSqr[u] <= P.radiusSqr);   /* :165 */
                ms.q[qn * ms.stride] = (uint32_t)idx;
                qn += acc ? 1 : 0;
            }
            lo = nextLo;
            if (wave_any(qn > ms.cap - VCM_MERGE_UNROLL)) {
                ls.mergeAccepted += (uint32_t)qn;
                merge_drain<IP>(P, g, ev, ms, qn, contrib);
                qn = 0;
            }
        }
    }
    ls.mergeAccepted += (uint32_t)qn;
    merge_drain<IP>(P, g, ev, ms, qn, contrib);
    return contrib;
}

/* ---- K4a bucket keys (counting sort of the camera vertices by the cell they lie in) ---- */
#ifndef VCM_QSORT_BITS
#define VCM_QSORT_BITS 8
#endif
#define VCM_QSORT_BUCKETS (1 << (3 * VCM_QSORT_BITS))   /* entries of the bucket table: upper bound, IterParams::nBuckets is in use */
/* Bucket index of a cell: row-major over the cells the photon bbox spans, with per-axis coarsening only as far
 * as the bucket table requires.  (The first version used a Morton code with 8 bits per axis: the moment the grid
 * passed 256 cells on one axis -- 2048^2: iteration 9, the radius shrinks every iteration -- every bucket became a
 * 2x2x2 block of cells in arrival order, a wave of K4 then touched ~21 distinct cells instead of ~13, and K4 jumped
 * from 3.9 to 5.4 ms.  Halving ONE axis at a time keeps a bucket at 1, 2, 4 ... cells, and the full table is used
 * before anything is coarsened: 257 x 251 x 257 still fits 2^24.) */
/* (Keyed by the lattice CORNER nearest to the query instead -- floor(cellPt + 0.5): the 2x2x2 block HashGrid::Process probes,
 * hashgrid.hxx:124-141, so equal keys walk the same eight buckets -- K4 was no faster and the iteration 1.5 % slower,
 * three runs of 40 iterations each way: profiles/archive/r06k_ab.txt.  With ~2 queries per cell there is little to share.) */
struct QueryBuckets { uint32_t nx, ny; int sx, sy, sz; };
VCM_HD QueryBuckets query_buckets(const IterParams &P, const GridHeader *hdr)
{   /* wave-uniform */
    const V3 ext = P.invCellSize * (ld3(hdr->bboxMax) - ld3(hdr->bboxMin));
    const uint32_t cx = (uint32_t)fmaxf(ext.x, 0.f) + 1u, cy = (uint32_t)fmaxf(ext.y, 0.f) + 1u, cz = (uint32_t)fmaxf(ext.z, 0.f) + 1u;
    QueryBuckets b; b.sx = b.sy = b.sz = 0;
    for (;;) {
        const unsigned long long nx = ((cx - 1u) >> b.sx) + 1u, ny = ((cy - 1u) >> b.sy) + 1u, nz = ((cz - 1u) >> b.sz) + 1u;
        if (nx * ny * nz <= (unsigned long long)P.nBuckets) { b.nx = (uint32_t)nx; b.ny = (uint32_t)ny; break; }
        if (nx >= ny && nx >= nz) b.sx++; else if (ny >= nz) b.sy++; else b.sz++;
    }
    return b;
}

/* what the key needs of the grid header, read ONCE per kernel into wave-uniform (scalar) registers: the photon bbox is
 * final before the camera pass starts.  Read where the key is formed -- once per appended camera vertex -- the header
 * came through two VECTOR loads (the compiler issues no scalar load for memory other kernels write), and the wait for
 * them was a wait for every record store the vertex had just issued; the bucket geometry was recomputed as well. */
struct QueryKey { float bminx, bminy, bminz, bmaxx, bmaxy, bmaxz; uint32_t nx, ny; int sx, sy, sz; };   /* scalars: an array member would live in scratch */
VCM_HD float wave_uniform_f(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
#else
    return v;
#endif
}
VCM_HD QueryKey query_key_load(const IterParams &P, const GridHeader *hdr)
{
    QueryKey k;
    const QueryBuckets b = query_buckets(P, hdr);
    k.bminx = wave_uniform_f(hdr->bboxMin[0]); k.bminy = wave_uniform_f(hdr->bboxMin[1]); k.bminz = wave_uniform_f(hdr->bboxMin[2]);
    k.bmaxx = wave_uniform_f(hdr->bboxMax[0]); k.bmaxy = wave_uniform_f(hdr->bboxMax[1]); k.bmaxz = wave_uniform_f(hdr->bboxMax[2]);
    k.nx = (uint32_t)wave_uniform((int)b.nx); k.ny = (uint32_t)wave_uniform((int)b.ny);
    k.sx = wave_uniform(b.sx); k.sy = wave_uniform(b.sy); k.sz = wave_uniform(b.sz);
    return k;
}
VCM_HD int query_sort_key(const IterParams &P, const QueryKey &k, V3 queryPos)
{
    const V3 distMin = queryPos - mk3(k.bminx, k.bminy, k.bminz);
    const V3 distMax = mk3(k.bmaxx, k.bmaxy, k.bmaxz) - queryPos;
    if (distMin.x < 0.f || distMax.x < 0.f || distMin.y < 0.f || distMax.y < 0.f || distMin.z < 0.f || distMax.z < 0.f)
        return -1;   /* outside the photon bbox: HashGrid::Process returns at once (:116-122) */
    const V3 cellPt = P.invCellSize * distMin;
    const uint32_t cx = (uint32_t)floorf(cellPt.x) >> k.sx, cy = (uint32_t)floorf(cellPt.y) >> k.sy,
                   cz = (uint32_t)floorf(cellPt.z) >> k.sz;
    return (int)((cz * k.ny + cy) * k.nx + cx);
}
VCM_HD int query_sort_key(const IterParams &P, const GridHeader *hdr, V3 queryPos)
{
    return query_sort_key(P, query_key_load(P, hdr), queryPos);
}


struct CameraPath {
    SubPathState st;
    PathRng rng;
    V3 color;
    int lp;
    float sx, sy;    /* the jittered screen sample (:576) */
    uint32_t queryMask;   /* wavefront mode: bit L set = a vertex record was appended at path length L */
    uint32_t lightLenMask;   /* wavefront mode: LightStore::lenMask of the light path of the same index, fetched when the
                             path starts (it was a dependent gather in front of every vertex's VC tasks) */
};

/* ---- wave-level block allocator for the device queues --------------------
 * A single hot counter word sustains only ~88 returning atomics per
 * microsecond on MI355X, and the camera pass appends ~30 M items per
 * iteration.  So a wave takes BLOCKS of slots from the global counter (one
 * atomic per block) and hands slots to its lanes with ballot + prefix
 * popcount; the unused tail of a block is filled with hole markers that the
 * consumers skip.  On the host build (one lane) it degenerates to a counter. */
#define VCM_QBLOCK_VERTEX 512   /* upper bounds (buffer sizing); the sizes in use are in IterParams */
#define VCM_QBLOCK_DI     512
#define VCM_QBLOCK_VC     2048
/* allocator state of ONE wave: {next free slot, slots left in the block}.  It
 * lives in LDS (one pair per wave and queue): the allocation runs in divergent
 * code, so a register copy would go stale in the lanes that sit a call out. */
/* The two words of a queue live in LDS and are reached through an LDS pointer with relaxed atomic loads and stores:
 * ds_read / ds_write that the compiler neither caches in a register nor hoists out of the loop.  (They were a
 * `volatile int *` first: a GENERIC pointer, so every access was a flat_load / flat_store with system scope followed
 * by s_waitcnt vmcnt(0) -- twelve of them per vertex K3 appended, each also waiting for every record store in flight:
 * 28 % of K3's wave time, profiles/archive/r05z_region_clock.txt.) */
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) int *WaveQueueWords;
#else
typedef int *WaveQueueWords;
#endif
VCM_HD int wq_load(WaveQueueWords p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __atomic_load_n(p, __ATOMIC_RELAXED);
#else
    return *p;
#endif
}
VCM_HD void wq_store(WaveQueueWords p, int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __atomic_store_n(p, v, __ATOMIC_RELAXED);
#else
    *p = v;
#endif
}
struct WaveQueue { WaveQueueWords p; };

VCM_HD uint32_t lanes_below_mask_popc(unsigned long long m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#else
    (void)m; return 0u;
#endif
}
/* n in [0,31] items for this lane (only active lanes call; n <= maxPathLength - 2 <= 29 in wavefront mode,
 * vcm_begin_iteration): returns the lane's first slot; refills the wave's block from *counter when it runs
 * out -- with as many blocks as the wave's request needs (64 lanes x 29 items exceed one block), taken with
 * ONE atomic so that they are contiguous.  holeFill(first, count, rank, nActive) must mark slots
 * [first, first+count) as holes; it is called by the nActive active lanes, rank = 0..nActive-1.
 * Space: a refill abandons fewer slots than the request it could not serve, so holes <= items; the buffers are
 * sized for 2 x the worst-case item count + one block per wave (arena_ensure). */
template <typename HoleFill>
VCM_HD int wave_queue_alloc(const WaveQueue &wq, int *counter, int blockSize, int n, HoleFill holeFill)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    const int rank = (int)lanes_below_mask_popc(act);
    int prefix = 0, total = 0;
#pragma unroll
    for (int bit = 0; bit < 5; bit++) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(((n >> bit) & 1) != 0);
        prefix += (int)lanes_below_mask_popc(m) << bit;
        total += __popcll(m) << bit;
    }
    int base = wq_load(wq.p), left = wq_load(wq.p + 1);   /* same values in every active lane */
    if (total > left) {   /* wave-uniform */
        holeFill(base, left, rank, (int)__popcll(act));
        const int take = ((total + blockSize - 1) / blockSize) * blockSize;
        int nb = 0;
        if (rank == 0) nb = atomicAdd(counter, take);
        base = __shfl(nb, __ffsll((long long)act) - 1, 64);
        left = take;
    }
    if (rank == 0) { wq_store(wq.p, base + total); wq_store(wq.p + 1, left - total); }
    return base + prefix;
#else
    (void)wq; (void)blockSize; (void)holeFill;
    const int r = *counter; *counter += n; return r;
#endif
}
/* The same on a copy of the queue's two words that the caller has read (and writes back): K3 appends to three queues
 * per vertex, and three read - modify - write round trips through LDS, one after the other, were a fifth of its step.
 * BITS = how many bits of n can be set (0: n is 1 in every active lane -- rank and count of the active lanes are the
 * prefix and the total). */
template <int BITS, typename HoleFill>
VCM_HD int wave_queue_take(int &base, int &left, int *counter, int blockSize, int n, unsigned long long act, int rank, HoleFill holeFill)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int prefix = rank, total = (int)__popcll(act);
    if (BITS > 0) {
        prefix = 0; total = 0;
#pragma unroll
        for (int bit = 0; bit < BITS; bit++) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(((n >> bit) & 1) != 0);
            prefix += (int)lanes_below_mask_popc(m) << bit;
            total += __popcll(m) << bit;
        }
    }
    if (total > left) {   /* wave-uniform */
        holeFill(base, left, rank, (int)__popcll(act));
        const int take = ((total + blockSize - 1) / blockSize) * blockSize;
        int nb = 0;
        if (rank == 0) nb = atomicAdd(counter, take);
        base = __shfl(nb, __ffsll((long long)act) - 1, 64);
        left = take;
    }
    const int mine = base + prefix;
    base += total; left -= total;
    return mine;
#else
    (void)base; (void)left; (void)blockSize; (void)holeFill; (void)act; (void)rank;
    const int r = *counter; *counter += n; return r;
#endif
}

/* GenerateCameraSample :564-606 (+ Camera::GenerateRay camera.hxx:108-117) */
VCM_HD void camera_path_begin(const DScene &sc, const IterParams &P, CameraPath &cp, int localPath, const uint32_t *lightLenMask = 0)
{
    const vcm_camera &cam = sc.camera;
    const int pathIdx = P.p0 + localPath;
    cp.lp = localPath;
    cp.color = sp3(0.f);
    cp.queryMask = 0u;
    cp.lightLenMask = (lightLenMask && P.useVC) ? lightLenMask[localPath] : 0u;
    rng_init(cp.rng, P.seed, P.localIter, (uint32_t)pathIdx, 1u);
    const int x = pathIdx % P.resX;
    const int y = pathIdx / P.resX;
    float jit[2];   /* :576, the first two floats of the path */
    rng_peek_block(cp.rng, 0u, jit, 2);
    cp.rng.k = 2u;
    const float jx = jit[0];
    const float jy = jit[1];
    cp.sx = float(x) + jx;
    cp.sy = float(y) + jy;
    const V3 worldRaster = transform_point(cam.rasterToWorld, mk3(cp.sx, cp.sy, 0.f));
    const V3 org = ld3(cam.position);
    const V3 dir = normalize(worldRaster - org);
    const float cosAtCamera = dot(ld3(cam.forward), dir);
    const float imagePointToCameraDist = cam.imagePlaneDist / cosAtCamera;
    const float imageToSolidAngleFactor = sqr(imagePointToCameraDist) / cosAtCamera;
    const float cameraPdfW = imageToSolidAngleFactor;
    SubPathState &st = cp.st;
    st.origin = org;
    st.direction = dir;
    st.throughput = sp3(1.f);
    st.pathLength = 1;
    st.specularPath = 1;
    st.isFiniteLight = 0;
    st.dVCM = mis(P.lightSubPathCount / cameraPdfW);
    st.dVC = 0.f;
    st.dVM = 0.f;
}

/* one iteration of the for(;;) at :423-542; returns false when the path ends */
/* MODE 0 ("strict"): everything inside the path, every addition in the
 *         reference's order as it happens.
 * MODE 1 ("wavefront", default): the path only traces and scatters; per
 *         non-delta vertex it appends a VertexStore record and its DI / VC
 *         tasks.  Returns false when the path ends. */
struct CameraWaveQueues {   /* wave-uniform allocator state of K3 */
    WaveQueue v, di, vc;
    /* per lane: the place a vertex took in its query-sort bucket (a RETURNING atomic, issued when the vertex is
       appended) is written to sortArrival by the caller AFTER the step: stored on the spot, the wave waited for the
       atomic's round trip -- and with it for every store of the record it had just issued -- once per bounce */
    int pendingVertex, pendingArrival;
};

template <int MODE, class SC>
VCM_HD bool camera_path_step(const SC &sc, const IterParams &P, CameraPath &cp, const LightStore &store,
                             const GridStore &grid, LaneStats &ls, const MergeScratch &ms, const VertexStore &vs,
                             CameraWaveQueues &wqs, const QueryKey &qk /* wavefront mode with the query sort: query_key_load, once per kernel */)
{
    SubPathState &st = cp.st;
    Ray ray; ray.org = st.origin + st.direction * VCM_EPS_RAY; ray.dir = st.direction; ray.tmin = 0;
    Isect isect; isect.dist = 1e36f; isect.matID = 0; isect.lightID = -1; isect.normal = sp3(0.f); isect.prim = -1;
    ls.cameraRays++;
    RC_DECL;
    const bool hitSomething = scene_intersect(sc, ray, isect);
    RC_MARK(4);
    if (!hitSomething) {   /* :434-447 */
        if (sc.backgroundLight >= 0) {
            if (st.pathLength >= P.minLen)
                cp.color = cp.color + st.throughput * get_light_radiance(sc, P, sc.lights()[sc.backgroundLight], st, ray.dir);
        }
        return false;
    }
    const V3 hitPoint = ray.org + ray.dir * isect.dist;
    isect.dist += VCM_EPS_RAY;
    Bsdf bsdf;
    bsdf_setup(bsdf, ray.dir, isect.normal, isect.matID, isect.prim, sc);
    if (bsdf.matID < 0) return false;
    {   /* :459-464 */
        st.dVCM *= mis(sqr(isect.dist));
        st.dVCM /= mis(fabsf(bsdf.localDirFix.z));
        st.dVC  /= mis(fabsf(bsdf.localDirFix.z));
        st.dVM  /= mis(fabsf(bsdf.localDirFix.z));
    }
    if (isect.lightID >= 0) {   /* :468-479 */
        const vcm_light &light = get_light(sc, isect.lightID);
        if (st.pathLength >= P.minLen)
            cp.color = cp.color + st.throughput * get_light_radiance(sc, P, light, st, ray.dir);
        return false;
    }
    if (st.pathLength >= P.maxLen) return false;   /* :482 */
    RC_MARK(5);

    if (MODE == 1) {
        if (!bsdf.isDelta && (P.useVC || P.useVM)) {
            /* DirectIllumination draws its 3 floats here, in path order (:672-673): the task records WHERE in
               the path's stream they are and K3b generates them */
            uint32_t diK = 0u;
            int hasDI = 0;
            if (P.useVC && st.pathLength + 1 >= P.minLen) {
                diK = cp.rng.k;
                cp.rng.k += 3u;
                hasDI = 1;
            }
            /* the light vertices this camera vertex connects to (:508-521) */
            /* vertex j of the light path has the j-th smallest stored pathLength; the window of :512-521
               (minLen <= lvLen + 1 + camLen <= maxLen) is a range of lengths, hence a range of j: pure bit
               arithmetic on the path's length mask instead of one dependent 16-byte gather per light vertex */
            uint32_t jmask = 0u;
            if (P.useVC) {
                const uint32_t M = cp.lightLenMask;
                const int loLen = (int)P.minLen - 1 - (int)st.pathLength;    /* lvLen >= loLen */
                const int hiLen = (int)P.maxLen - 1 - (int)st.pathLength;    /* lvLen <= hiLen */
                if (hiLen >= 0) {
                    const uint32_t below = (loLen <= 0) ? 0u : (loLen >= 32 ? 0xffffffffu : ((1u << loLen) - 1u));
                    const uint32_t upto = (hiLen >= 31) ? 0xffffffffu : ((1u << (hiLen + 1)) - 1u);
                    const int jlo = __builtin_popcount(M & below), jhi = __builtin_popcount(M & upto);
                    if (jhi > jlo) jmask = ((jhi >= 32) ? 0xffffffffu : ((1u << jhi) - 1u)) & ~((1u << jlo) - 1u);
                }
            }
            const int nvc = __builtin_popcount(jmask);
#if defined(__HIP_DEVICE_COMPILE__)
            /* the three queues' words: read together, written back together by the first active lane */
            const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
            const int rank = (int)lanes_below_mask_popc(act);
            int vb = wq_load(wqs.v.p), vl = wq_load(wqs.v.p + 1), db = wq_load(wqs.di.p), dl = wq_load(wqs.di.p + 1), cb = wq_load(wqs.vc.p), cl = wq_load(wqs.vc.p + 1);
            const int vi = wave_queue_take<0>(vb, vl, &vs.count[0], P.qblockVertex, 1, act, rank,
                [&](int first, int cnt, int rk, int na) {
                    for (int i = rk; i < cnt; i += na) {
                        vq(vs, 0, first + i) = mk4(0.f, 0.f, 0.f, u2f(0xffffffffu));
                        if (vs.sortKey) vs.sortKey[first + i] = -1;
                    } });
            const int di = wave_queue_take<1>(db, dl, &vs.count[1], P.qblockDI, hasDI, act, rank,
                [&](int first, int cnt, int rk, int na) { for (int i = rk; i < cnt; i += na) vs.diTask[first + i] = -1; });
            const int vc0 = wave_queue_take<5>(cb, cl, &vs.count[2], P.qblockVC, nvc, act, rank,
                [&](int first, int cnt, int rk, int na) { for (int i = rk; i < cnt; i += na) vs.vcTask[2 * (first + i)] = -1; });
            if (rank == 0) { wq_store(wqs.v.p, vb); wq_store(wqs.v.p + 1, vl); wq_store(wqs.di.p, db); wq_store(wqs.di.p + 1, dl); wq_store(wqs.vc.p, cb); wq_store(wqs.vc.p + 1, cl); }
#else
            const int vi = wave_queue_alloc(wqs.v, &vs.count[0], P.qblockVertex, 1, [](int, int, int, int) {});
            const int di = wave_queue_alloc(wqs.di, &vs.count[1], P.qblockDI, hasDI, [](int, int, int, int) {});
            const int vc0 = wave_queue_alloc(wqs.vc, &vs.count[2], P.qblockVC, nvc, [](int, int, int, int) {});
#endif
            vq(vs, 0, vi) = mk4(hitPoint.x, hitPoint.y, hitPoint.z, u2f((uint32_t)cp.lp));
            vq(vs, 1, vi) = mk4(isect.normal.x, isect.normal.y, isect.normal.z, u2f(st.pathLength | (shade_code(bsdf.matID, isect.prim) << 8)));
            vq(vs, 2, vi) = mk4(bsdf.localDirFix.x, bsdf.localDirFix.y, bsdf.localDirFix.z, st.dVCM);
            vq(vs, 3, vi) = mk4(st.throughput.x, st.throughput.y, st.throughput.z, st.dVM);
            vq(vs, 4, vi) = mk4(st.dVC, u2f(diK), 0.f, 0.f);
            I4 m; m.x = hasDI ? di : -1; m.y = vc0; m.z = nvc; m.w = 0;
            vs.meta[path_slot(P, st.pathLength, (uint32_t)cp.lp)] = m;
#if defined(__HIP_DEVICE_COMPILE__)
            if (vs.sortKey && P.useVM) {   /* K4a's histogram pass, here (see VertexStore) */
                const int k = query_sort_key(P, qk, hitPoint);
                vs.sortKey[vi] = k;
#if defined(VCM_NO_DEFER)   /* measurement switch: store on the spot */
                if (k >= 0) vs.sortArrival[vi] = atomicAdd(&vs.bucketCount[k], 1);
#else
                /* the place the atomic hands back goes to memory at the lane's NEXT append (or when the kernel ends): a whole bounce
                   later, so no wave waits for the round trip (stored at the end of the same step -- rounds 3-4 -- the wait was
                   still visible at the end of the step: +0.6 %, four pairs of 40 iterations, profiles/archive/r06zd_defer_ab.txt) */
                if (wqs.pendingVertex >= 0) vs.sortArrival[wqs.pendingVertex] = wqs.pendingArrival;
                wqs.pendingVertex = -1;
                if (k >= 0) { wqs.pendingVertex = vi; wqs.pendingArrival = atomicAdd(&vs.bucketCount[k], 1); }
#endif
                else vs.mergeOut[path_slot(P, st.pathLength, (uint32_t)cp.lp)] = mk4(0.f, 0.f, 0.f, 0.f);   /* empty query */
            }
#endif
            if (hasDI) vs.diTask[di] = vi;
            int t = vc0;
            while (jmask) {
                const int j = __builtin_ctz(jmask);
                jmask &= jmask - 1u;
                VcTaskPair pr; pr.vertex = vi; pr.j = j;
                reinterpret_cast<VcTaskPair *>(vs.vcTask)[t] = pr;   /* one 8-byte store (the array is 256-byte aligned) */
                t++;
            }
            cp.queryMask |= 1u << st.pathLength;
            if (P.useVM) {
                ls.mergeQueries++;
                if (P.ppm) return false;   /* :537 */
            }
        }
    } else {
        if (!bsdf.isDelta && P.useVC) {   /* :487-494 */
            if (st.pathLength + 1 >= P.minLen) {
                float rnd[3];
                rng_peek(cp.rng, cp.rng.k, rnd, 3);
                cp.rng.k += 3u;
                cp.color = cp.color + st.throughput * direct_illumination(sc, P, rnd[0], rnd[1], rnd[2], st, hitPoint, bsdf, ls);
            }
        }
        if (!bsdf.isDelta && P.useVC) {   /* :498-526: the light path of the same index */
            const int n = store.count[cp.lp];
            for (int j = 0; j < n; j++) {
                const size_t slot = (size_t)j * (size_t)P.nLocal + (size_t)cp.lp;
                const F4 a = lv(store, slot, 0);
                const uint32_t lvLen = f2u(a.w) & 0xffu;
                if (lvLen + 1 + st.pathLength < P.minLen) continue;
                if (lvLen + 1 + st.pathLength > P.maxLen) break;
                const F4 b = lv(store, slot, 1);
                const F4 c = lv(store, slot, 2);
                const F4 d = lv(store, slot, 3);
                Bsdf lvBsdf;
                bsdf_restore(lvBsdf, mk3(c.x, c.y, c.z), mk3(d.x, d.y, d.z), f2u(a.w) >> 8, sc);
                cp.color = cp.color + st.throughput * mk3(b.x, b.y, b.z) *
                           connect_vertices(sc, P, mk3(a.x, a.y, a.z), lvBsdf, b.w, c.w, bsdf, hitPoint, st, ls);
            }
        }
        if (!bsdf.isDelta && P.useVM) {   /* :530-538 */
            ls.mergeQueries++;
            const V3 contrib = merge_query<SC::kIntPhong>(sc, P, grid, bsdf, st, hitPoint, ls, ms);
            cp.color = cp.color + st.throughput * P.vmNormalization * contrib;
            if (P.ppm) return false;
        }
    }
    RC_MARK(6);
    const bool goesOn = sample_scattering(sc, P, false, cp.rng, bsdf, hitPoint, st);
    RC_MARK(7);
    if (!goesOn) return false;
    ++st.pathLength;
    return true;
}

/* ---- wavefront tasks: the deferred parts of a camera vertex ------------- */
struct CamVertex {
    V3 hit, throughput;
    Bsdf bsdf;
    SubPathState st;      /* pathLength, dVCM, dVC, dVM are valid */
    uint32_t lp;
    uint32_t diK;         /* position of DirectIllumination's 3 floats in the path's random stream */
};
VCM_HD void load_cam_vertex(const DScene &sc, const VertexStore &vs, int vi, CamVertex &v)
{
    const F4 a = vq(vs, 0, vi), b = vq(vs, 1, vi), c = vq(vs, 2, vi), d = vq(vs, 3, vi), e = vq(vs, 4, vi);
    v.hit = mk3(a.x, a.y, a.z);
    v.lp = f2u(a.w);
    bsdf_restore(v.bsdf, mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), f2u(b.w) >> 8, sc);
    v.throughput = mk3(d.x, d.y, d.z);
    v.st.pathLength = f2u(b.w) & 0xffu;
    v.st.dVCM = c.w; v.st.dVM = d.w; v.st.dVC = e.x;
    v.st.throughput = v.throughput;
    v.diK = f2u(e.y);
}
/* the addend of :491  (color += throughput * DirectIllumination(...)) */
template <class SC>
VCM_HD V3 eval_di_task(const SC &sc, const IterParams &P, const VertexStore &vs, int vi, LaneStats &ls,
                       size_t &pathSlot)
{
    CamVertex v;
    load_cam_vertex(sc, vs, vi, v);
    pathSlot = path_slot(P, v.st.pathLength, v.lp);
    PathRng rng;
    rng_init(rng, P.seed, P.localIter, (uint32_t)(P.p0 + (int)v.lp), 1u);
    float rnd[3];
    rng_peek(rng, v.diK, rnd, 3);
    return v.throughput * direct_illumination(sc, P, rnd[0], rnd[1], rnd[2], v.st, v.hit, v.bsdf, ls);
}
/* the addend of :523  (color += throughput * lightVertex.mThroughput * ConnectVertices(...)) */
template <class SC>
VCM_HD V3 eval_vc_task(const SC &sc, const IterParams &P, const VertexStore &vs, const LightStore &store,
                       int vi, int j, LaneStats &ls)
{
    CamVertex v;
    load_cam_vertex(sc, vs, vi, v);
    const size_t slot = (size_t)j * (size_t)P.nLocal + (size_t)v.lp;
    const F4 a = lv(store, slot, 0), b = lv(store, slot, 1), c = lv(store, slot, 2), d = lv(store, slot, 3);
    Bsdf lvBsdf;
    bsdf_restore(lvBsdf, mk3(c.x, c.y, c.z), mk3(d.x, d.y, d.z), f2u(a.w) >> 8, sc);
    return v.throughput * mk3(b.x, b.y, b.z) *
           connect_vertices(sc, P, mk3(a.x, a.y, a.z), lvBsdf, b.w, c.w, v.bsdf, v.hit, v.st, ls);
}
/* the addend of :534  (color += throughput * mVmNormalization * query.GetContrib()) */
template <bool IP>
VCM_HD V3 eval_merge_task(const DScene &sc, const IterParams &P, const VertexStore &vs, const GridStore &g,
                          int vi, LaneStats &ls, const MergeScratch &ms, size_t &pathSlot, bool ldsMaterials = true)
{
    const F4 a = vq(vs, 0, vi), b = vq(vs, 1, vi), c = vq(vs, 2, vi), d = vq(vs, 3, vi);
    pathSlot = path_slot(P, f2u(b.w) & 0xffu, f2u(a.w));
    Bsdf bsdf;
    bsdf_restore(bsdf, mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), f2u(b.w) >> 8, sc, ldsMaterials);
    SubPathState st;
    st.pathLength = f2u(b.w) & 0xffu; st.dVCM = c.w; st.dVM = d.w;
    const V3 contrib = merge_query<IP>(sc, P, g, bsdf, st, mk3(a.x, a.y, a.z), ls, ms, ldsMaterials);
    return mk3(d.x, d.y, d.z) * P.vmNormalization * contrib;
}
/* Replays vertexcm.hxx:417-544 for one camera path: colour starts at 0, every
 * vertex adds its DI term, its VC terms (increasing j), its merge term; the
 * emission / background term of the last segment comes last (the path ends
 * there).  Same addends, same order => same bits as the serial reference. */
VCM_HD V3 replay_path_color(const IterParams &P, const VertexStore &vs, int lp, uint32_t mask, V3 emission)
{
    /* pure latency: per vertex the three slot-indexed loads go out together, the VC addends in batches of 4 */
    V3 color = sp3(0.f);
    while (mask) {
        const int L = __builtin_ctz(mask);
        mask &= mask - 1u;
        const size_t ps = path_slot(P, (uint32_t)L, (uint32_t)lp);
        const I4 m = vs.meta[ps];
        const F4 di = vs.diOut[ps];
        const F4 mg = P.useVM ? vs.mergeOut[ps] : mk4(0.f, 0.f, 0.f, 0.f);
        if (m.x >= 0) color = color + mk3(di.x, di.y, di.z);
        for (int k0 = 0; k0 < m.z; k0 += 4) {
            F4 t[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int u = 0; u < 4; u++) t[u] = (k0 + u < m.z) ? vs.vcOut[m.y + k0 + u] : mk4(0.f, 0.f, 0.f, 0.f);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int u = 0; u < 4; u++) if (k0 + u < m.z) color = color + mk3(t[u].x, t[u].y, t[u].z);
        }
        if (P.useVM) color = color + mk3(mg.x, mg.y, mg.z);
    }
    return color + emission;
}

/* ================= PathTracer::RunIteration (pathtracer.hxx:45-215) ================= */
/* The reference's second renderer (SURVEY section 8(f) "next" #2): same device functions, one lane per
 * pixel, next-event estimation + BSDF sampling combined with Mis2.  Random floats of a path, in order:
 * jitter x,y (:59); per vertex [light pick, 2 for Illuminate (:148-154)] if the vertex does NEE, the Sample
 * triplet (:185), the Russian-roulette float only if contProb < 1 (:201-203). */
struct PtPath {
    V3 org, dir;          /* the next ray; the FIRST one starts at the camera without the EPS_RAY offset (:61) */
    V3 weight, color;
    uint32_t pathLength, lastSpecular;
    float lastPdfW;
    float sx, sy;
    PathRng rng;
    int lp;
};
VCM_HD float mis2(float samplePdf, float otherPdf) { return mis(samplePdf) / (mis(samplePdf) + mis(otherPdf)); }   /* :226-231 */
VCM_HD void pt_path_begin(const DScene &sc, const IterParams &P, PtPath &pp, int localPath)
{
    const vcm_camera &cam = sc.camera;
    const int pathIdx = P.p0 + localPath;
    pp.lp = localPath;
    rng_init(pp.rng, P.seed, P.localIter, (uint32_t)pathIdx, 1u);
    float jit[2];
    rng_peek_block(pp.rng, 0u, jit, 2);
    pp.rng.k = 2u;
    pp.sx = float(pathIdx % P.resX) + jit[0];   /* :56-59 */
    pp.sy = float(pathIdx / P.resX) + jit[1];
    const V3 worldRaster = transform_point(cam.rasterToWorld, mk3(pp.sx, pp.sy, 0.f));   /* camera.hxx:108-117 */
    pp.org = ld3(cam.position);
    pp.dir = normalize(worldRaster - pp.org);
    pp.weight = sp3(1.f);
    pp.color = sp3(0.f);
    pp.pathLength = 1;
    pp.lastSpecular = 1;
    pp.lastPdfW = 1.f;
}
/* one turn of the for(;; ++pathLength) at :71-213; false when the path ends */
template <class SC>
VCM_HD bool pt_path_step(const SC &sc, const IterParams &P, PtPath &pp, LaneStats &ls)
{
    const int lightCount = sc.nLights;
    const float lightPickProb = 1.f / lightCount;   /* :48-49 */
    Ray ray; ray.org = pp.org; ray.dir = pp.dir; ray.tmin = 0;
    Isect isect; isect.dist = 1e36f; isect.matID = 0; isect.lightID = -1; isect.normal = sp3(0.f); isect.prim = -1;
    ls.cameraRays++;
    if (!scene_intersect(sc, ray, isect)) {   /* :73-97 */
        if (pp.pathLength < P.minLen) return false;
        if (sc.backgroundLight < 0) return false;
        float directPdfW = 0.f, emissionPdfW = 0.f;
        const V3 contrib = light_get_radiance(sc.lights()[sc.backgroundLight], sc, ray.dir, directPdfW, emissionPdfW);
        if (iszero(contrib)) return false;
        float misWeight = 1.f;
        if (pp.pathLength > 1 && !pp.lastSpecular) misWeight = mis2(pp.lastPdfW, directPdfW * lightPickProb);
        pp.color = pp.color + pp.weight * misWeight * contrib;
        return false;
    }
    const V3 hitPoint = ray.org + ray.dir * isect.dist;   /* :99-100 */
    isect.dist += VCM_EPS_RAY;
    Bsdf bsdf;
    bsdf_setup(bsdf, ray.dir, isect.normal, isect.matID, isect.prim, sc);
    if (bsdf.matID < 0) return false;
    if (isect.lightID >= 0) {   /* :107-129 */
        if (pp.pathLength < P.minLen) return false;
        const vcm_light &light = get_light(sc, isect.lightID);
        float directPdfA = 0.f, emissionPdfW = 0.f;
        const V3 contrib = light_get_radiance(light, sc, ray.dir, directPdfA, emissionPdfW);
        if (iszero(contrib)) return false;
        float misWeight = 1.f;
        if (pp.pathLength > 1 && !pp.lastSpecular) {
            const float directPdfW = pdf_a_to_w(directPdfA, isect.dist, bsdf.localDirFix.z);   /* CosThetaFix, bsdf.hxx:263 */
            misWeight = mis2(pp.lastPdfW, directPdfW * lightPickProb);
        }
        pp.color = pp.color + pp.weight * misWeight * contrib;
        return false;
    }
    if (pp.pathLength >= P.maxLen) return false;   /* :131 */
    if (bsdf.contProb == 0.f) return false;        /* :134 */
    if (!bsdf.isDelta && pp.pathLength + 1 >= P.minLen) {   /* next event estimation :138-179 */
        float rnd[3];
        rng_peek(pp.rng, pp.rng.k, rnd, 3);
        pp.rng.k += 3u;
        const int lightID = int(rnd[0] * lightCount);
        const vcm_light &light = get_light(sc, lightID);
        V3 directionToLight;
        float distance, directPdfW, emissionPdfW, cosAtLight;
        const V3 radiance = light_illuminate(light, sc, hitPoint, rnd[1], rnd[2], directionToLight, distance, directPdfW,
                                             emissionPdfW, cosAtLight);
        if (!iszero(radiance)) {
            float bsdfPdfW, cosThetaOut;
            const V3 factor = bsdf_evaluate(bsdf, sc, directionToLight, cosThetaOut, &bsdfPdfW, NULL);
            if (!iszero(factor)) {
                float weight = 1.f;
                if (!light_is_delta(light)) {
                    const float contProb = bsdf.contProb;
                    bsdfPdfW *= contProb;
                    weight = mis2(directPdfW * lightPickProb, bsdfPdfW);
                }
                const V3 contrib = (weight * cosThetaOut / (lightPickProb * directPdfW)) * (radiance * factor);
                ls.shadowRays++;
                if (!scene_occluded(sc, hitPoint, directionToLight, distance)) pp.color = pp.color + pp.weight * contrib;
            }
        }
    }
    {   /* continue the random walk :182-212 */
        float rnd[4];
        rng_peek(pp.rng, pp.rng.k, rnd, 4);
        pp.rng.k += 3u;
        float pdf, cosThetaOut;
        uint32_t sampledEvent;
        V3 newDir;
        const V3 factor = bsdf_sample(bsdf, sc, false, rnd[0], rnd[1], rnd[2], newDir, pdf, cosThetaOut, sampledEvent);
        if (iszero(factor)) return false;
        const float contProb = bsdf.contProb;
        pp.lastSpecular = (sampledEvent & kSpecular) != 0 ? 1u : 0u;
        pp.lastPdfW = pdf * contProb;
        if (contProb < 1.f) {
            pp.rng.k += 1u;
            if (rnd[3] > contProb) return false;
            pdf *= contProb;
        }
        pp.weight = pp.weight * (factor * (cosThetaOut / pdf));
        pp.dir = newDir;
        pp.org = hitPoint + VCM_EPS_RAY * newDir;   /* :208 */
    }
    ++pp.pathLength;
    return true;
}

/* ================= EyeLight::RunIteration (eyelight.hxx:46-77) ================= */
/* returns the colour and the jittered sample; hit = false: nothing is added (:68) */
template <class SC>
VCM_HD bool eyelight_path(const SC &sc, const IterParams &P, int localPath, V3 &color, float &sx, float &sy,
                          uint32_t &floatsDrawn, LaneStats &ls)
{
    const vcm_camera &cam = sc.camera;
    const int pathIdx = P.p0 + localPath;
    float jit[2] = { 0.5f, 0.5f };
    floatsDrawn = 0u;
    if (P.iteration != 1) {   /* :60-61: iteration 1 samples the pixel centres and draws nothing */
        PathRng rng;
        rng_init(rng, P.seed, P.localIter, (uint32_t)pathIdx, 1u);
        rng_peek_block(rng, 0u, jit, 2);
        floatsDrawn = 2u;
    }
    sx = float(pathIdx % P.resX) + jit[0];
    sy = float(pathIdx / P.resX) + jit[1];
    Ray ray;
    const V3 worldRaster = transform_point(cam.rasterToWorld, mk3(sx, sy, 0.f));
    ray.org = ld3(cam.position);
    ray.dir = normalize(worldRaster - ray.org);
    ray.tmin = 0;
    Isect isect; isect.dist = 1e36f; isect.matID = 0; isect.lightID = -1; isect.normal = sp3(0.f); isect.prim = -1;
    ls.cameraRays++;
    if (!scene_intersect(sc, ray, isect)) return false;
    const float dotLN = dot(isect.normal, -ray.dir);   /* :70-75 */
    color = (dotLN > 0) ? sp3(dotLN) : mk3(-dotLN, 0.f, 0.f);
    return true;
}

VCM_HD int raster_target(const IterParams &P, float sx, float sy)
{   /* Framebuffer::AddColor framebuffer.hxx:43-57 */
    const float rx = (float)P.resX, ry = (float)P.resY;
    if (sx < 0 || sx >= rx) return -1;
    if (sy < 0 || sy >= ry) return -1;
    return int(sx) + int(sy) * P.resX;
}

/* Framebuffer::AddColor(screenSample, color) vertexcm.hxx:544, framebuffer.hxx:43-57:
 * the pixel comes from the JITTERED sample; float(x)+jitter can round up to
 * x+1, so the colour may belong to the next pixel or be dropped at the edge. */
VCM_HD int camera_path_target(const IterParams &P, const CameraPath &cp)
{
    const float rx = (float)P.resX, ry = (float)P.resY;
    if (cp.sx < 0 || cp.sx >= rx) return -1;
    if (cp.sy < 0 || cp.sy >= ry) return -1;
    return int(cp.sx) + int(cp.sy) * P.resX;
}

} // namespace vcm
#endif
