// vcm_api.hip -- implementation of the C-ABI (include/smallvcm_amd.h): context
// management, device memory, kernel launches for one VCM iteration.
//
// No host<->device synchronisation happens inside an iteration: vertex counts
// stay on the device (GridHeader) and the grid-build kernels are grid-stride
// over a device-resident count.  The only sync points are the read-back entry
// points (framebuffer, stats, light-record count for the multi-GPU exchange).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <new>
#include <mutex>
#include <vector>
#include <condition_variable>
#include <atomic>

#include "vcm_kernels.h"
#include "vcm_kat.h"
#include "scene_host.h"

using namespace vcm;

static thread_local std::string g_err;
static thread_local bool g_hipFailed;   /* the last failure on this thread came from the HIP runtime */
static int fail(const char *what, const char *detail)
{
    g_err = std::string(what) + ": " + (detail ? detail : "");
    return -1;
}
#define HIPCHK(expr)                                                            \
    do {                                                                        \
        hipError_t e_ = (expr);                                                 \
        if (e_ != hipSuccess) { g_hipFailed = true; return fail(#expr, hipGetErrorString(e_)); } \
    } while (0)

#define VCM_MAX_TRACE_WAVES (256 * 32)   /* upper bound of the persistent waves of K1 / K3 */
#define VCM_STAMP_RING 64   /* iterations of phase timestamps and counters kept on the device */
#define VCM_STAT_SLOTS (STAT_COUNT + 1)   /* + the number of vertices in the grid */
enum { EV_START = 0, EV_LIGHT_K0, EV_LIGHT_K1, EV_LIGHT, EV_GRID_K0, EV_GRID, EV_CAMERA_K0, EV_CAMERA_K1, EV_CONNECT_K1,
       EV_MERGE_K0, EV_SORT_K1, EV_MERGE_K1, EV_CAMERA, EV_COUNT };

/* Iteration scratch: everything that is dead once an iteration has been
 * resolved into the framebuffer.  It lives in ARENAS that all single-rank contexts of a device share: the
 * reference's render() creates one renderer per host core and runs them concurrently (smallvcm.cxx:66, :82-108)
 * -- 256 private copies (1.5 GB each at 512^2, 24 GB at 2048^2) do not fit in HBM.  A context borrows an arena
 * from vcm_begin_iteration to vcm_end_iteration; consecutive borrowers of one arena are ordered on the GPU by an
 * event, the host never waits for the device.  A device keeps a small POOL of arenas (as many as fit a quarter of
 * its memory, at most 8; vcm_set_arena_limit): iterations of different renderers then overlap on the GPU -- one
 * renderer's tails and latency-bound helper kernels run next to another's wide kernels, which is what fills the
 * chip at 512^2, where a single iteration is too small to do it. */
#define VCM_RSORT_MAX_BLOCKS 4096   /* workgroups of K2's radix sort (sort_cells_radix) */
struct Scratch {
    LightStore store;                 /* S*nLocal slots (+ count[nLocal]) */
    int *dPathStart;                  /* nLocal+1 */
    int *dLocalTotal;                 /* 1 */
    int *dTileSums[4];                /* scan scratch: [0] main stream, [1] side stream (grid build), [2] splat stream, [3] sort stream */
    int *dPixCount, *dPixStart;       /* N+2 each: light splats per pixel, and the start of every pixel's list (K1d) */
    int *dSplatArrival;               /* per light vertex: the place of its splat in its pixel's list */
    F4 *dSplatList;                   /* per light vertex: the splats grouped by pixel */
    float *dRecordsLocal;             /* S*nLocal records */
    int *dSlotOfVertex;               /* S*nLocal: dense vertex index -> slot in the light store */
    F4 *dSplat;                       /* S*nLocal: splat of each light vertex (rgb | pixel) */
    float *dRecordsAll;               /* S*N records (multi-rank only) */
    int *dCellCount, *dCellStart;     /* N+2 each */
    int *dCellId;                     /* per record */
    I4 *dUnsorted;                    /* per record: {index, slot, cell} list entries, grouped by cell; the radix sort's two {vertex, slot} lists */
    int *dRadixHist;                  /* 2 x 256 x VCM_RSORT_MAX_BLOCKS: digit histogram per workgroup, raw and scanned */
    float *dGx, *dGy, *dGz, *dGb; F4 *dG1, *dG2; F2 *dG3;
    int *dSortedIndex;                /* parity: grid position -> record index */
    F4 *dCamOut;                      /* nLocal */
    uint32_t *dCamMask;               /* nLocal: path lengths at which a vertex record was appended */
    VertexStore vs;                   /* camera vertices + DI/VC tasks (wavefront mode) */
    int *dQueryKey;                   /* per camera vertex: sort key */
    int *dSortedVertex;               /* camera vertices sorted by key */
    int *dQueryStart, *dQueryCount;   /* VCM_QSORT_BUCKETS+2 each (dQueryStart also >= N+2) */
    int *dQueryArrival;               /* per camera vertex: its place inside its bucket */
};

struct vcm_ctx;
struct ArenaPool;
struct Arena {
    /* *mtx guards the fields below (the pool's mutex, or the arena's own for the private arena of a sharded
       context) and is only ever held INSIDE an API call.  The arena itself is lent with the `busy` token, not by
       holding a mutex from vcm_begin_iteration to vcm_end_iteration: the two calls may come from different host
       threads, and vcm_create / vcm_destroy of other contexts must not block behind an open iteration. */
    std::mutex *mtx;
    std::condition_variable *cv;
    std::mutex ownMtx;
    std::condition_variable ownCv;
    ArenaPool *pool;                  /* NULL: private to one sharded context */
    bool busy;                        /* lent to a context (between begin and end) */
    unsigned long long ownerThread;   /* host thread that borrowed it, 0 = none */
    int device;
    Scratch s;
    size_t capLocal, capN; int capS, capL; bool capSharded;
    size_t bytes;                     /* device memory held */
    bool allocated;
    hipEvent_t lastUse; bool eventReady, lastValid;
    vcm_ctx *lastUser;                /* whose iteration the buffers still hold (NULL: nobody's) */
};
#define VCM_MAX_ARENAS 8
struct ArenaPool {
    std::mutex m;
    std::condition_variable cv;
    int device;
    int users;                        /* live single-rank contexts on this device */
    int limit;                        /* arenas this pool may hold (<= VCM_MAX_ARENAS); 0 = by memory budget */
    int count;
    Arena *arenas[VCM_MAX_ARENAS];
};
static std::atomic<unsigned long long> g_threadCounter{0};
static thread_local unsigned long long g_threadId = 0;
static unsigned long long this_thread_id() { if (!g_threadId) g_threadId = ++g_threadCounter; return g_threadId; }

/* Kernels that cast rays exist once per kind of scene (vcm_core.h SceneList / SceneBvh); the launch picks. */
#define LAUNCH_SC(c, K, ...) do { if (!(c)->scene->nodes.empty()) { if ((c)->intPhong) hipLaunchKernelGGL((K<SceneBvh>), __VA_ARGS__); \
                                                                     else hipLaunchKernelGGL((K<SceneBvhG>), __VA_ARGS__); } \
                                  else if (!(c)->intPhong) hipLaunchKernelGGL((K<SceneList>), __VA_ARGS__); \
                                  else if ((c)->sceneRects) hipLaunchKernelGGL((K<SceneRects>), __VA_ARGS__); \
                                  else if ((c)->sceneQuads) hipLaunchKernelGGL((K<SceneQuads>), __VA_ARGS__); \
                                  else hipLaunchKernelGGL((K<SceneList>), __VA_ARGS__); } while (0)
#define LAUNCH_SC_MODE(c, K, M, ...) do { if (!(c)->scene->nodes.empty()) { if ((c)->intPhong) hipLaunchKernelGGL((K<M, SceneBvh>), __VA_ARGS__); \
                                                                             else hipLaunchKernelGGL((K<M, SceneBvhG>), __VA_ARGS__); } \
                                          else if (!(c)->intPhong) hipLaunchKernelGGL((K<M, SceneList>), __VA_ARGS__); \
                                          else if ((c)->sceneRects) hipLaunchKernelGGL((K<M, SceneRects>), __VA_ARGS__); \
                                          else if ((c)->sceneQuads) hipLaunchKernelGGL((K<M, SceneQuads>), __VA_ARGS__); \
                                          else hipLaunchKernelGGL((K<M, SceneList>), __VA_ARGS__); } while (0)

#ifndef VCM_MERGE_DEFAULT
#define VCM_MERGE_DEFAULT VCM_MERGE_PAIRS
#endif
struct vcm_ctx : Scratch {
    SceneHost *scene;                 /* host copy of the scene + the structure the intersection code walks */
    bool useVM, useVC, lightTraceOnly, ppm;
    int renderer;                     /* 0 VertexCM family, 1 PathTracer, 2 EyeLight */
    float baseRadius, radiusAlpha;
    int seed, device, rank, world;
    int resX, resY, N, p0, nLocal;
    int iterations;

    hipStream_t stream;
    bool ownStream;
    hipStream_t side;                 /* the grid build runs here, next to the camera pass on `stream` */
    hipEvent_t evFork, evBbox, evGrid; /* main -> side, side -> main (bbox known), side -> main (grid complete) */
    bool gridInFlight;
    hipStream_t splat;                /* small frames: K1c / K1d (light splats) run here, next to the camera pass */
    hipEvent_t evSplatFork, evSplatDone;
    hipEvent_t evSortFork, evSorted;  /* the query sort's scan + scatter run on `side`, behind the grid build and next to K3b (round 4) */
    hipEvent_t evZero;                /* the iteration's tables are zero (side stream, next to K1) */
    bool prezeroed, sortInFlight;
    bool splatInFlight;
    bool resolveInFlight;             /* K5 of the LAST iteration runs on the splat stream (beside the next iteration's K1): survives vcm_end_iteration */
    hipEvent_t evMergeDone, evSplatWork, evResolved;
    hipEvent_t evPreMerge;            /* K4 runs on the SIDE stream (round 5), so that the main stream can start the NEXT iteration's K1 beside it */
    bool mergeInFlight;               /* K4 of the last iteration may still run: what it reads (grid header floats, cell ranges, camera vertices) must not be rewritten yet */
    bool deviceReady;
    ArenaPool *pool;                  /* the device's shared arenas (NULL: sharded context with a private one) */
    Arena *arena; bool holdsArena;    /* the arena of the current / last iteration */

    /* per context: survives the iteration */
    DScene *dScene;                   /* the scene header in device memory = the start of dSceneBlob */
    char *dSceneBlob;                 /* header, primitives, materials, lights, pairs / BVH: one allocation */
    float *dFb;                       /* N*3, running sum (mFramebuffer, renderer.hxx:68) */
    unsigned char *dRngLight, *dRngCam;   /* the random-number tape of the last iteration */
    GridHeader *dHdr;
    unsigned long long *dStats;       /* the current iteration's slot of dStatsRing */
    unsigned long long *dStatsRing;   /* VCM_STAMP_RING x VCM_STAT_SLOTS: counters of the last iterations */
    unsigned long long *dStamps;      /* VCM_STAMP_RING x EV_COUNT device wall-clock readings */
    float radiusRing[VCM_STAMP_RING];
    double stampKHz;

    bool importedRecords;
    bool importedSorted;              /* vcm_import_sorted_light_records: the grid build merges the ranks' cell-sorted slabs */
    bool sortedExchange;              /* sharded context that expects the sorted exchange: K1b does not materialise the records */
    SortedSlabs sortedIn;
    /* the merge sharded by space (round 6 prototype, include/smallvcm_amd.h "merge sharded by SPACE") */
    SpaceSlabs space; bool spaceSet, mergeImported;
    float presetMin[3];               /* host copy of the box vcm_set_grid_bbox installed */
    int *dSpaceMatrix, *dSpaceScanned, *dSpaceTotals, *dSpaceHist;
    int *dWhereDest, *dWherePos; size_t whereCap;
    F4 *fQ, *fRes; int *fKey, *fArrival, *fSorted, *fCount; size_t fCap;   /* the queries other ranks sent, their sort, their terms */
    bool gridBuilt, cameraTraced, merged, splatsPending, recordsValid, countedInCamera, scatteredInDI, bboxPreset;
    bool bboxFromLight;               /* K1 of this iteration accumulated the vertices' box into dHdr (single rank) */
    bool bboxFinal;                   /* ... and k_compact_records has turned it into floats already */
    bool strictOrder;
    int mergeKind;                    /* VCM_MERGE_* */
    bool sceneQuads;                  /* every triangle pair of the list shares its plane part: the SceneQuads kernels */
    bool sceneRects;                  /* ... and is an axis-aligned rectangle: the SceneRects kernels */
    bool intPhong;                    /* every Phong exponent in use is an integer in [1, 65536]: the kernels whose pow is the binary
                                         exponentiation alone (detmath.h); otherwise the SceneList / SceneBvhG kernels and the general merge */
    IterParams P;
    bool inIteration;
    hipEvent_t ev[EV_COUNT];
    bool evValid;
    unsigned long long *pend[2][4]; int nPend[2];   /* stamp slots waiting for the next kernel on [0] main, [1] side stream */
    vcm_stats lastStats;
};

static int use_device(vcm_ctx *c) { HIPCHK(hipSetDevice(c->device)); return 0; }

static thread_local size_t g_allocBytes;   /* bytes dalloc handed out on this thread (arena_ensure reads the difference) */
template <typename T> static int dalloc(T **p, size_t n)
{
    void *v = NULL;
    HIPCHK(hipMalloc(&v, (n ? n : 1) * sizeof(T)));
    g_allocBytes += (n ? n : 1) * sizeof(T);
    *p = (T *)v;
    return 0;
}
#define DFREE(p) do { if (p) { (void)hipFree(p); p = NULL; } } while (0)

/* ---- arenas ------------------------------------------------------------- */
static std::mutex g_arenaRegistryMutex;
static ArenaPool *g_pools[64];

static Arena *arena_new(int device, ArenaPool *pool)
{
    Arena *a = new Arena();
    a->device = device;
    memset((void *)&a->s, 0, sizeof(Scratch));
    a->capLocal = a->capN = 0; a->capS = a->capL = 0; a->capSharded = false; a->bytes = 0;
    a->allocated = false; a->eventReady = a->lastValid = false; a->lastUser = NULL;
    a->pool = pool; a->ownerThread = 0; a->busy = false;
    a->mtx = pool ? &pool->m : &a->ownMtx;
    a->cv = pool ? &pool->cv : &a->ownCv;
    return a;
}
static ArenaPool *pool_get(int device)
{
    if (device < 0 || device >= 64) return NULL;
    std::lock_guard<std::mutex> g(g_arenaRegistryMutex);
    if (!g_pools[device]) {
        ArenaPool *p = new ArenaPool();
        p->device = device; p->users = 0; p->count = 0;
        memset(p->arenas, 0, sizeof(p->arenas));
        const char *e = getenv("SMALLVCM_AMD_ARENAS");
        p->limit = (e && atoi(e) > 0) ? (atoi(e) < VCM_MAX_ARENAS ? atoi(e) : VCM_MAX_ARENAS) : 0;
        g_pools[device] = p;
    }
    return g_pools[device];
}

static void arena_free_buffers(Arena *a)
{
    Scratch &s = a->s;
    DFREE(s.store.v); DFREE(s.store.count); DFREE(s.store.lenMask);
    DFREE(s.dPathStart); DFREE(s.dLocalTotal); DFREE(s.dTileSums[0]); DFREE(s.dTileSums[1]); DFREE(s.dTileSums[2]); DFREE(s.dTileSums[3]);
    DFREE(s.dPixCount); DFREE(s.dPixStart); DFREE(s.dSplatArrival); DFREE(s.dSplatList);
    DFREE(s.dRecordsLocal); DFREE(s.dRecordsAll); DFREE(s.dSlotOfVertex); DFREE(s.dSplat);
    DFREE(s.dCellCount); DFREE(s.dCellStart); DFREE(s.dCellId); DFREE(s.dUnsorted); DFREE(s.dRadixHist);
    DFREE(s.dGx); DFREE(s.dGy); DFREE(s.dGz); DFREE(s.dGb); DFREE(s.dG1); DFREE(s.dG2); DFREE(s.dG3); DFREE(s.dSortedIndex);
    DFREE(s.dCamOut); DFREE(s.dCamMask);
    DFREE(s.vs.q); DFREE(s.vs.q4); DFREE(s.vs.meta); DFREE(s.vs.count);
    DFREE(s.vs.diTask); DFREE(s.vs.vcTask); DFREE(s.vs.diOut); DFREE(s.vs.vcOut); DFREE(s.vs.mergeOut);
    DFREE(s.dQueryKey); DFREE(s.dSortedVertex); DFREE(s.dQueryStart); DFREE(s.dQueryCount); DFREE(s.dQueryArrival);
    a->allocated = false;
    a->capLocal = a->capN = 0; a->capS = a->capL = 0; a->capSharded = false;
}

/* called by the context that holds the busy token; grows the arena to what this context needs */
static int arena_ensure(Arena *a, size_t nLocal, size_t N, int S, int L, bool sharded)
{
    if (!a->eventReady) { HIPCHK(hipEventCreateWithFlags(&a->lastUse, hipEventDisableTiming)); a->eventReady = true; }
    if (a->allocated && nLocal <= a->capLocal && N <= a->capN && S <= a->capS && L <= a->capL && (!sharded || a->capSharded))
        return 0;
    HIPCHK(hipDeviceSynchronize());   /* rare: another context's kernels may still use the old buffers */
    const size_t cl = nLocal > a->capLocal ? nLocal : a->capLocal, cn = N > a->capN ? N : a->capN;
    const int cs = S > a->capS ? S : a->capS, cL = L > a->capL ? L : a->capL;
    const bool sh = sharded || a->capSharded;
    arena_free_buffers(a);
    a->lastValid = false;
    a->lastUser = NULL;   /* the previous borrower's Scratch copy points at freed memory now */
    a->bytes = 0;
    const size_t bytesBefore = g_allocBytes;
    Scratch &s = a->s;
    const size_t slots = (size_t)cs * cl;
    const size_t allRecs = (size_t)cs * cn;
    if (dalloc(&s.store.v, slots * VCM_LV_FIELDS) || dalloc(&s.store.count, cl) || dalloc(&s.store.lenMask, cl)) return -1;
    if (dalloc(&s.dPathStart, cl + 1) || dalloc(&s.dLocalTotal, 1)) return -1;
    size_t maxScan = (cn > cl ? cn : cl) + 1;
    if (maxScan < (size_t)VCM_QSORT_BUCKETS + 1) maxScan = (size_t)VCM_QSORT_BUCKETS + 1;
    for (int w = 0; w < 4; w++) if (dalloc(&s.dTileSums[w], maxScan / VCM_SCAN_TILE + 2)) return -1;
    if (dalloc(&s.dRecordsLocal, slots * VCM_MERGE_RECORD_FLOATS)) return -1;
    if (dalloc(&s.dSlotOfVertex, slots) || dalloc(&s.dSplat, slots)) return -1;
    if (dalloc(&s.dPixCount, cn + 2) || dalloc(&s.dPixStart, cn + 2) || dalloc(&s.dSplatArrival, slots) || dalloc(&s.dSplatList, slots)) return -1;
    if (sh && dalloc(&s.dRecordsAll, allRecs * VCM_MERGE_RECORD_FLOATS)) return -1;
    if (dalloc(&s.dCellCount, cn + 2) || dalloc(&s.dCellStart, cn + 2)) return -1;
    if (dalloc(&s.dCellId, allRecs) || dalloc(&s.dUnsorted, allRecs) || dalloc(&s.dRadixHist, (size_t)2 * 256 * VCM_RSORT_MAX_BLOCKS)) return -1;
    if (dalloc(&s.dGx, allRecs + VCM_MERGE_UNROLL) || dalloc(&s.dGy, allRecs + VCM_MERGE_UNROLL) ||
        dalloc(&s.dGz, allRecs + VCM_MERGE_UNROLL) || dalloc(&s.dGb, 3 * (allRecs + 2 * VCM_MERGE_UNROLL)) || dalloc(&s.dG1, allRecs) || dalloc(&s.dG2, allRecs) ||
        dalloc(&s.dG3, allRecs) || dalloc(&s.dSortedIndex, allRecs)) return -1;
    if (dalloc(&s.dCamOut, cl) || dalloc(&s.dCamMask, cl) || dalloc(&s.vs.count, 32)) return -1;
    /* wavefront buffers, worst case: <= L vertices per path; a vertex at path length l connects to
       light vertices of length <= L-1-l, so <= (L-1)(L-2)/2 VC tasks per path; + one partly used
       block per wave (holes) */
    /* Holes: when a wave's block cannot serve a request it abandons what is left of it, which is less than the
       request itself (wave_queue_alloc): <= 64 of every >= 256 slots for the vertex / DI queues (one item per
       lane), up to as many slots as items for the VC queue (<= 29 items per lane). */
    const size_t maxWaves = (size_t)VCM_MAX_TRACE_WAVES;
    const size_t vitems = (size_t)(cL > 0 ? cL : 1) * cl;
    const size_t vslots = vitems + vitems / 3 + maxWaves * VCM_QBLOCK_VERTEX;
    const size_t vcPerPath = (cL >= 3) ? (size_t)(cL - 1) * (size_t)(cL - 2) / 2 : 1;
    const size_t vcslots = 2 * vcPerPath * cl + maxWaves * VCM_QBLOCK_VC;
    s.vs.qcap = vslots;
    if (dalloc(&s.vs.q, vslots * 4) || dalloc(&s.vs.q4, vslots) || dalloc(&s.vs.meta, vslots) ||
        dalloc(&s.vs.diTask, vslots) || dalloc(&s.vs.diOut, vslots) ||
        dalloc(&s.vs.mergeOut, vslots) || dalloc(&s.vs.vcTask, 2 * vcslots) || dalloc(&s.vs.vcOut, vcslots) ||
        dalloc(&s.dQueryKey, vslots) || dalloc(&s.dSortedVertex, vslots)) return -1;
    const size_t qsN = (cn > (size_t)VCM_QSORT_BUCKETS ? cn : (size_t)VCM_QSORT_BUCKETS) + 2;   /* also pixStart of K1d */
    if (dalloc(&s.dQueryStart, qsN) || dalloc(&s.dQueryCount, (size_t)VCM_QSORT_BUCKETS + 2) ||
        dalloc(&s.dQueryArrival, vslots)) return -1;
    a->capLocal = cl; a->capN = cn; a->capS = cs; a->capL = cL; a->capSharded = sh;
    a->allocated = true;
    a->bytes = g_allocBytes - bytesBefore;
    return 0;
}

/* may the pool grow by one more arena?  Budget: a quarter of the device's memory, the existing arenas as the
   estimate of the next one's size (a pool with an explicit limit just counts) */
static bool pool_may_grow(ArenaPool *p)
{
    if (p->count >= VCM_MAX_ARENAS) return false;
    if (p->limit > 0) return p->count < p->limit;
    if (p->count == 0) return true;
    size_t held = 0, largest = 0;
    for (int i = 0; i < p->count; i++) { held += p->arenas[i]->bytes; if (p->arenas[i]->bytes > largest) largest = p->arenas[i]->bytes; }
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return false;
    return held + largest <= totalB / 4 && largest + (totalB / 16) <= freeB;
}

/* borrow an arena for one iteration */
static int arena_acquire(vcm_ctx *c, int S, int L)
{
    Arena *a = NULL;
    bool wait;
    if (c->pool) {
        ArenaPool *p = c->pool;
        std::unique_lock<std::mutex> lk(p->m);
        for (;;) {
            Arena *mine = NULL, *idle = NULL, *lru = NULL;
            bool holdsOne = false;
            for (int i = 0; i < p->count; i++) {
                Arena *x = p->arenas[i];
                if (x->busy) { if (x->ownerThread == this_thread_id()) holdsOne = true; continue; }
                if (x->lastUser == c) mine = x;
                else if (!x->lastValid || hipEventQuery(x->lastUse) == hipSuccess) idle = x;
                else lru = x;
            }
            (void)hipGetLastError();   /* hipEventQuery reports hipErrorNotReady through the sticky error too */
            a = mine ? mine : idle;
            if (!a && pool_may_grow(p)) { a = arena_new(p->device, p); p->arenas[p->count++] = a; }
            if (!a) a = lru;
            if (a) break;
            /* Waiting while holding is the dead-lock: two threads that each hold one arena and each open a second
               iteration would wait for each other forever once the pool cannot grow.  A thread that holds nothing waits. */
            if (holdsOne)
                return fail("vcm_begin_iteration", "this thread is already inside an iteration on this device and no further "
                                                   "scratch arena is free: single-rank contexts share the iteration scratch, "
                                                   "end an iteration first (or raise vcm_set_arena_limit)");
            p->cv.wait(lk);
        }
        a->busy = true;
        a->ownerThread = this_thread_id();
        wait = a->lastValid && a->lastUser != c;
        if (a->lastUser != c) a->lastUser = NULL;   /* from here on the buffers are this context's */
        /* the arena this context used before (if another one) no longer holds anything it may read */
        if (c->arena && c->arena != a && c->arena->lastUser == c) c->arena->lastUser = NULL;
        c->arena = a;
    } else {
        a = c->arena;
        std::unique_lock<std::mutex> lk(*a->mtx);
        if (a->busy && a->ownerThread == this_thread_id())
            return fail("vcm_begin_iteration", "this thread is already inside an iteration of this sharded context");
        a->cv->wait(lk, [a] { return !a->busy; });
        a->busy = true;
        a->ownerThread = this_thread_id();
        wait = a->lastValid && a->lastUser != c;
        if (a->lastUser != c) a->lastUser = NULL;
    }
    c->holdsArena = true;
    if (arena_ensure(a, (size_t)c->nLocal, (size_t)c->N, S, L, c->world > 1)) return -1;
    *static_cast<Scratch *>(c) = a->s;
    if (wait && a->lastValid) HIPCHK(hipStreamWaitEvent(c->stream, a->lastUse, 0));
    return 0;
}
static void arena_release(vcm_ctx *c, bool recordEvent)
{
    if (!c->holdsArena) return;
    Arena *a = c->arena;
    /* (K5 on the splat stream: the iteration's LAST kernel is there, behind everything of the main stream) */
    const bool recorded = recordEvent && a->eventReady && hipEventRecord(a->lastUse, c->resolveInFlight ? c->splat : c->stream) == hipSuccess;
    c->holdsArena = false;
    {
        std::lock_guard<std::mutex> g(*a->mtx);
        if (recorded) { a->lastValid = true; a->lastUser = c; }
        a->busy = false;
        a->ownerThread = 0;
    }
    a->cv->notify_all();
}
/* A failed phase call ENDS the iteration -- a HIP error as well as a call-order error: its work is abandoned,
 * the arena goes back (other contexts must not dead-lock behind it) and the next call has to be
 * vcm_begin_iteration.  What the iteration had already added to the framebuffer stays there. */
static int abort_iteration(vcm_ctx *c, int rc)
{
    if (rc != 0 && c && c->inIteration && c->holdsArena) {
        const std::string keep = g_err;   /* the message of the failure, not of the clean-up */
        (void)hipStreamSynchronize(c->stream);
        if (c->deviceReady) { (void)hipStreamSynchronize(c->side); (void)hipStreamSynchronize(c->splat); }
        c->mergeInFlight = false;
        c->gridInFlight = false;
        c->splatInFlight = false;
        c->resolveInFlight = false;
        c->sortInFlight = false;
        c->inIteration = false;
        arena_release(c, false);
        g_err = keep;
    } else if (rc != 0 && c && c->holdsArena && !c->inIteration) {
        arena_release(c, false);   /* vcm_begin_iteration failed after borrowing */
    }
    return rc;
}

/* the scratch of the last iteration is readable until another context borrows the arena */
static int scratch_readable(vcm_ctx *c, const char *what)
{
    if (!c || !c->deviceReady || (!c->inIteration && c->iterations == 0)) return fail(what, "no iteration has run");
    if (!c->holdsArena) {
        if (!c->arena) return fail(what, "no iteration has run");
        std::lock_guard<std::mutex> g(*c->arena->mtx);
        if (c->arena->lastUser != c)
            return fail(what, "the iteration scratch has since been used by another context of this device");
    }
    return 0;
}

static int ensure_device(vcm_ctx *c)
{
    if (use_device(c)) return -1;
    if (!c->deviceReady) {
        if (c->ownStream) HIPCHK(hipStreamCreate(&c->stream));
        for (int i = 0; i < EV_COUNT; i++) HIPCHK(hipEventCreate(&c->ev[i]));
        /* the helper streams: equal priorities (lowest priority for them bought 0.5-0.9 % at 2048^2 and cost 28 % at 512^2, a CU mask
           for them cost 7-12 %: profiles/archive/r05k_prio.txt, r05r_configs.txt, profiles/archive/r07l_ab_2048_vcm_s1.txt; both switches retired in round 6) */
        HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&c->splat, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->evSortFork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->evSorted, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->evZero, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->evSplatFork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->evSplatDone, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->evMergeDone, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->evSplatWork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->evResolved, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->evPreMerge, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->evBbox, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->evGrid, hipEventDisableTiming));
        {   /* the scene in ONE device allocation: the DScene header, then the arrays it addresses by offset */
            const SceneHost &h = *c->scene;
            struct Part { const void *src; size_t bytes; size_t off; } parts[13] = {
                { h.prims.data(), h.prims.size() * sizeof(vcm_prim), 0 }, { h.materials.data(), h.materials.size() * sizeof(vcm_material), 0 },
                { h.mat2light.data(), h.mat2light.size() * sizeof(int), 0 }, { h.lights.data(), h.lights.size() * sizeof(vcm_light), 0 },
                { h.ops.data(), h.ops.size() * sizeof(PrimOp), 0 }, { h.pairs.data(), h.pairs.size() * sizeof(TriPair), 0 },
                { h.nodes.data(), h.nodes.size() * sizeof(BvhNode), 0 }, { h.leafPrims.data(), h.leafPrims.size() * sizeof(int), 0 },
                { h.fastPairs.data(), h.fastPairs.size() * sizeof(FastPair), 0 }, { h.fastSpheres.data(), h.fastSpheres.size() * sizeof(FastSphere), 0 },
                { h.wide.data(), h.wide.size() * sizeof(BvhWide), 0 }, { h.leafData.data(), h.leafData.size() * sizeof(LeafPrim), 0 },
                { h.fastRects.data(), h.fastRects.size() * sizeof(FastRect), 0 } };
            size_t total = (sizeof(DScene) + 255) & ~(size_t)255;
            for (Part &p : parts) { p.off = total; total += (p.bytes + 255) & ~(size_t)255; }
            if (dalloc(&c->dSceneBlob, total + 256)) return -1;
            for (const Part &p : parts)
                if (p.bytes) HIPCHK(hipMemcpy(c->dSceneBlob + p.off, p.src, p.bytes, hipMemcpyHostToDevice));
            DScene view;
            h.fill_scalars(view);
            c->sceneQuads = h.nodes.empty() && view.fastOnePlane != 0;
            c->sceneRects = c->sceneQuads && (view.nFastRects[0] + view.nFastRects[1] + view.nFastRects[2]) > 0;
            c->intPhong = true;
            for (const vcm_material &m : h.materials)
                if ((m.phong[0] != 0.f || m.phong[1] != 0.f || m.phong[2] != 0.f) && !dm_pow_is_int_case(m.phongExp)) c->intPhong = false;
            { const char *e = getenv("SMALLVCM_AMD_GENERAL_POW"); if (e && e[0] == '1') c->intPhong = false; }   /* tests: the general kernels on the reference's scenes */
            view.offPrims = (long long)parts[0].off; view.offMaterials = (long long)parts[1].off;
            view.offMat2light = (long long)parts[2].off; view.offLights = (long long)parts[3].off;
            view.offOps = (long long)parts[4].off; view.offPairs = (long long)parts[5].off;
            view.offNodes = (long long)parts[6].off; view.offLeafPrims = (long long)parts[7].off;
            view.offFastPairs = (long long)parts[8].off; view.offFastSpheres = (long long)parts[9].off;
            view.offWide = (long long)parts[10].off; view.offLeafData = (long long)parts[11].off; view.offFastRects = (long long)parts[12].off;
            c->dScene = reinterpret_cast<DScene *>(c->dSceneBlob);
            HIPCHK(hipMemcpy(c->dScene, &view, sizeof(DScene), hipMemcpyHostToDevice));
        }
        if (dalloc(&c->dFb, (size_t)c->N * 3)) return -1;
        HIPCHK(hipMemset(c->dFb, 0, (size_t)c->N * 3 * sizeof(float)));
        if (dalloc(&c->dRngLight, (size_t)c->nLocal)) return -1;
        if (dalloc(&c->dRngCam, (size_t)c->nLocal)) return -1;
        HIPCHK(hipMemset(c->dRngLight, 0, (size_t)c->nLocal));
        HIPCHK(hipMemset(c->dRngCam, 0, (size_t)c->nLocal));
        if (dalloc(&c->dHdr, 1)) return -1;
        HIPCHK(hipMemset(c->dHdr, 0, sizeof(GridHeader)));
        if (dalloc(&c->dStatsRing, (size_t)VCM_STAMP_RING * VCM_STAT_SLOTS)) return -1;
        HIPCHK(hipMemset(c->dStatsRing, 0, (size_t)VCM_STAMP_RING * VCM_STAT_SLOTS * sizeof(unsigned long long)));
        c->dStats = c->dStatsRing;
        if (dalloc(&c->dStamps, (size_t)VCM_STAMP_RING * EV_COUNT)) return -1;
        HIPCHK(hipMemset(c->dStamps, 0, (size_t)VCM_STAMP_RING * EV_COUNT * sizeof(unsigned long long)));
        { int khz = 0; HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device)); c->stampKHz = khz > 0 ? (double)khz : 100000.0; }
        c->deviceReady = true;
    }
    return 0;
}

/* Phase / kernel boundaries of every iteration are marked twice on the context's stream: a HIP event, and the
 * device's constant-rate wall clock read by a one-lane kernel.  Both give the same intervals (checked:
 * SMALLVCM_AMD_TIMING=events makes vcm_get_stats report the events); the stamps are kept for the last
 * VCM_STAMP_RING iterations, so a host can run a batch of iterations back to back and ask for per-iteration
 * figures afterwards -- reading back between iterations lets the GPU idle and return at a lower clock, which
 * made kernel times come out 10-20 % longer than in the batch. */
/* Framebuffer::SaveBMP / SaveHDR pixel encodings (framebuffer.hxx:194-214, :229-247), one lane per pixel */
static __global__ void k_encode_image(const float *fb, int resX, int resY, int format, float scale, float invGamma, unsigned char *out)
{
    const int n = resX * resY;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        if (format == VCM_IMAGE_BGR8) {
            const int x = p % resX, y = p / resX;
            const float *c = fb + (size_t)(x + (resY - y - 1) * resX) * 3;   /* bottom-up (:200) */
            const float r = c[0] * scale, g = c[1] * scale, b = c[2] * scale;
            const float v[3] = { dm_powf(b, invGamma) * 255.f, dm_powf(g, invGamma) * 255.f, dm_powf(r, invGamma) * 255.f };
            for (int k = 0; k < 3; k++) out[(size_t)p * 3 + k] = (unsigned char)fminf(255.f, fmaxf(0.f, v[k]));
        } else {
            const float *c = fb + (size_t)p * 3;
            const float r = c[0] * scale, g = c[1] * scale, b = c[2] * scale;
            unsigned char e4[4] = { 0, 0, 0, 0 };
            float v = fmaxf(r, fmaxf(g, b));
            if (v >= 1e-32f) {
                int e;
                v = (float)(frexp((double)v, &e) * 256.f / v);   /* :241, evaluated in double like the reference */
                e4[0] = (unsigned char)(r * v); e4[1] = (unsigned char)(g * v); e4[2] = (unsigned char)(b * v);
                e4[3] = (unsigned char)(e + 128);
            }
            for (int k = 0; k < 4; k++) out[(size_t)p * 4 + k] = e4[k];
        }
    }
}
static __global__ void k_set_bbox(GridHeader *hdr, float x0, float y0, float z0, float x1, float y1, float z1)
{
    hdr->bboxMin[0] = x0; hdr->bboxMin[1] = y0; hdr->bboxMin[2] = z0;
    hdr->bboxMax[0] = x1; hdr->bboxMax[1] = y1; hdr->bboxMax[2] = z1;
}
static __global__ void k_note_grid_vertices(const GridHeader *hdr, unsigned long long *out, StampArgs st) { stamp_entry(st); *out = (unsigned long long)hdr->nRecords; }
static __global__ void k_stamp_many(StampArgs st) { stamp_entry(st); }
/* Zeroing up to four device ranges in ONE launch (16-byte stores plus a tail of 4-byte ones; every range starts
 * 16-byte aligned and is a whole number of 4-byte words, except byte tables, whose size is rounded up to 4: they are
 * allocated in larger units).  hipMemsetAsync reached 380 GB/s on the 16 MB tables and cost a launch per
 * range: nine of them were 0.43 ms of an 11 ms iteration (profiles/archive/r02k_kernel_stats.csv). */
/* vcm_begin_iteration zeroes the first six words of the grid header: the order keys K1 accumulates the box into */
static_assert(offsetof(GridHeader, bboxMinU) == 0 && offsetof(GridHeader, bboxMaxU) == 3 * sizeof(uint32_t), "GridHeader layout");
struct ZeroArgs { void *p[4]; unsigned long long n16[4]; unsigned tail4[4]; };
static __global__ void __launch_bounds__(256) k_zero_ranges(ZeroArgs z)
{
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        uint4 *q = (uint4 *)z.p[r];
        for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < z.n16[r]; i += (unsigned long long)gridDim.x * blockDim.x) q[i] = zero;
        if (blockIdx.x == 0 && threadIdx.x < z.tail4[r]) ((unsigned *)(q + z.n16[r]))[threadIdx.x] = 0u;
    }
}
static int zero_ranges(hipStream_t stream, void *p0, size_t b0, void *p1 = NULL, size_t b1 = 0, void *p2 = NULL, size_t b2 = 0, void *p3 = NULL, size_t b3 = 0)
{
    ZeroArgs z;
    void *p[4] = { p0, p1, p2, p3 };
    const size_t b[4] = { b0, b1, b2, b3 };
    size_t most = 0;
    for (int r = 0; r < 4; r++) {
        z.p[r] = p[r]; z.n16[r] = p[r] ? b[r] / 16 : 0; z.tail4[r] = p[r] ? (unsigned)((b[r] % 16 + 3) / 4) : 0u;
        if (z.n16[r] + 1 > most && p[r] && b[r]) most = z.n16[r] + 1;
    }
    if (!most) return 0;
#if defined(VCM_HIP_MEMSET)   /* measurement switch: one hipMemsetAsync per range, as before */
    for (int r = 0; r < 4; r++) if (p[r] && b[r]) HIPCHK(hipMemsetAsync(p[r], 0, b[r], stream));
    return 0;
#endif
    size_t blocks = (most + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_zero_ranges, dim3((unsigned)blocks), dim3(256), 0, stream, z);
    HIPCHK(hipGetLastError());
    return 0;
}
static StampArgs take_stamps(vcm_ctx *c, hipStream_t stream)
{
    const int w = (stream == c->side) ? 1 : 0;
    StampArgs st;
    for (int k = 0; k < 4; k++) st.p[k] = (k < c->nPend[w]) ? c->pend[w][k] : NULL;
    c->nPend[w] = 0;
    return st;
}
/* stamps still pending when a phase has no further kernel: one tiny launch writes them */
static int flush_stamps(vcm_ctx *c, hipStream_t stream)
{
    const int w = (stream == c->side) ? 1 : 0;
    if (c->nPend[w] == 0) return 0;
    hipLaunchKernelGGL(k_stamp_many, dim3(1), dim3(1), 0, stream, take_stamps(c, stream));
    HIPCHK(hipGetLastError());
    return 0;
}
/* mark a phase boundary on `stream`: the NEXT kernel launched there writes the device clock into the mark's slot when
 * it starts (take_stamps / stamp_entry) -- the boundary costs no launch of its own.  SMALLVCM_AMD_TIMING=events additionally
 * records a HIP event per mark (the two agree; the events only remember the last iteration). */
static int mark_on(vcm_ctx *c, int ev, hipStream_t stream)
{
    static int events = -1;
    if (events < 0) { const char *e = getenv("SMALLVCM_AMD_TIMING"); events = (e && !strcmp(e, "events")) ? 1 : 0; }
    if (events) HIPCHK(hipEventRecord(c->ev[ev], stream));
    const int w = (stream == c->side) ? 1 : 0;
    if (c->nPend[w] == 4 && flush_stamps(c, stream)) return -1;
    c->pend[w][c->nPend[w]++] = c->dStamps + (size_t)(c->iterations % VCM_STAMP_RING) * EV_COUNT + ev;
    return 0;
}
static int mark(vcm_ctx *c, int ev) { return mark_on(c, ev, c->stream); }

/* exclusive scan of n ints/bytes: two launches (vcm_kernels.h); `which` = 0 main stream, 1 side stream, 2 splat stream --
   each has its own tile sums in the arena, because the scans of different streams may run at the same time */
template <typename T>
static int launch_scan_on(vcm_ctx *c, int which, hipStream_t stream, const T *in, int n, int *out, int *totalOut, int writeTotalAtN,
                          StampArgs st)
{
    if (n <= 0) {   /* nothing to scan: the total is zero (not a failure of the iteration) */
        if (totalOut) HIPCHK(hipMemsetAsync(totalOut, 0, sizeof(int), stream));
        if (writeTotalAtN) HIPCHK(hipMemsetAsync(out, 0, sizeof(int), stream));
        return 0;
    }
    const int nTiles = (n + VCM_SCAN_TILE - 1) / VCM_SCAN_TILE;
    /* k_scan_apply: every tile adds up the tile sums before it -- nTiles^2 / 2 reads out of L2, 134 MB at the 8192 tiles
       of the largest table in use (the 16.8 M-entry bucket table of the query sort).  Quadratic: refuse what the
       scratch was not sized for instead of crawling */
    if (nTiles > 16384) return fail("launch_scan", "more than 16384 tiles: use a three-level scan");
    int *tileSums = c->dTileSums[which];
    hipLaunchKernelGGL((k_scan_tile_sums<T>), dim3(nTiles), dim3(VCM_SCAN_BLOCK), 0, stream, in, n, tileSums, st);
    hipLaunchKernelGGL((k_scan_apply<T>), dim3(nTiles), dim3(VCM_SCAN_BLOCK), 0, stream, in, n, (const int *)tileSums, out, totalOut, writeTotalAtN);
    HIPCHK(hipGetLastError());
    return 0;
}
template <typename T>
static int launch_scan(vcm_ctx *c, const T *in, int n, int *out, int *totalOut, int writeTotalAtN)
{
    return launch_scan_on<T>(c, 0, c->stream, in, n, out, totalOut, writeTotalAtN, take_stamps(c, c->stream));
}

/* Launch shapes are chosen per frame size below ("Launch shapes follow the frame"); ONE switch overrides any of them
   for sweeps and tests: SMALLVCM_AMD_SHAPE="key=value,key=value", keys: trace_waves, light_waves, trace_chunk (persistent waves of
   K3 / of K1, paths per chunk), task_blocks (K1c, K3b, K3c), merge_blocks, merge_chunk (K4), aux_blocks (streaming helpers),
   resolve_blocks (K5), grid_sort_blocks (K2's radix sort), buckets_per_path (the query sort).  0 / absent: the default.
   (Rounds 2-5 had one environment variable per knob: ten of them.) */
static int shape_knob(const char *key)
{
    static const char *spec = getenv("SMALLVCM_AMD_SHAPE");
    if (!spec) return 0;
    const size_t n = strlen(key);
    for (const char *p = spec; *p;) {
        if (!strncmp(p, key, n) && p[n] == '=') { const int v = atoi(p + n + 1); return v > 0 ? v : 0; }
        const char *q = strchr(p, ',');
        if (!q) break;
        p = q + 1;
    }
    return 0;
}
static void trace_launch_shape(int nLocal, int *blocks, int *chunk, bool lightPass = false)
{
    /* persistent waves: enough to fill 256 CUs several times over, each wave
       owning a contiguous chunk of paths (>= 64) */
    int waves = (nLocal + VCM_WAVE - 1) / VCM_WAVE;
    const int tw = shape_knob("trace_waves");
    /* 4096 = 16 waves per CU: measured best (fewer, longer-lived waves leave fewer partly used queue blocks) */
    int maxWaves = tw ? tw : 256 * 16;
    /* a 512^2 frame is exactly 4096 waves of one path per lane: every wave then lives as long as its longest path.  With
       3072 waves a third of the lanes take a second path: K3 0.41 -> 0.36 ms (profiles/archive/r05c_ab_summary.txt; at 1024^2
       4096 is best) */
    /* ... and so does 1024^2 (round 5): 3072 waves and chunks of 128 -- a third of the chunks dealt dynamically -- K3 0.83 -> 0.72 ms
       on scene 3, 1.02 -> 0.82 on scene 1 (1068 -> 1116 and 808 -> 896 Mpaths/s, profiles/archive/r07g_ab_1024_*.txt); no effect at 2048^2 */
    if (!tw && nLocal <= (1 << 20)) maxWaves = 256 * 12;
    const int lw = shape_knob("light_waves");   /* K1 needs fewer registers than K3: 5 waves per SIMD fit */
    if (lightPass && lw) maxWaves = lw;
    if (maxWaves > VCM_MAX_TRACE_WAVES) maxWaves = VCM_MAX_TRACE_WAVES;   /* the queue buffers hold one spare block per wave */
    if (waves > maxWaves) waves = maxWaves;
    if (waves < 1) waves = 1;
    const int wavesPerBlock = VCM_TRACE_BLOCK / VCM_WAVE;
    *blocks = (waves + wavesPerBlock - 1) / wavesPerBlock;
    const int totalWaves = *blocks * wavesPerBlock;
    /* indices are dealt out in chunks (vcm_kernels.h wave_work_take): the first chunk of a wave is fixed, the rest
       comes from a counter.  Up to 1024^2 a wave gets everything in its first chunk (64 .. 256 paths: no atomics, as
       before); above, chunks of 256 -- four per wave at 2048^2.  Smaller chunks balance no better and cost a grab
       every step where paths are short (environment light: K1 0.24 -> 0.34 ms at 1024^2 with 64-path chunks, r03f) */
    int ch = nLocal / totalWaves;
    const int ce = shape_knob("trace_chunk");
    if (ce) ch = ce;
    /* up to 1024^2 (round 5): chunks of 128.  At 1024^2 a third of the 8192 chunks is then dealt dynamically; at 512^2 it means
       2048 waves with two paths per lane instead of 3072 with one and a third (584 -> 608 Mpaths/s, profiles/archive/r07h_ab_512_vcm_s1.txt;
       chunks of 64: 566) */
    if (!ce && !tw && nLocal <= (1 << 20)) {
        ch = 128;
        const int need = (nLocal + ch - 1) / ch;   /* waves that get a first chunk at all */
        if (totalWaves > need) *blocks = (need + wavesPerBlock - 1) / wavesPerBlock;
    }
    *chunk = ch < 64 ? 64 : (ch > 256 && !ce ? 256 : ch);
}

extern "C" {

const char *vcm_last_error(void) { return g_err.c_str(); }

/* which build this is: "default", or the name and flags of a measurement variant (csrc/Makefile `variant`) -- so that an
   A/B run can say which library actually served it */
#ifndef VCM_BUILD_TAG
#define VCM_BUILD_TAG "default"
#endif
const char *vcm_build_tag(void) { return VCM_BUILD_TAG; }

int vcm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* the part of vcm_create common to both scene descriptions; takes ownership of `h` */
static vcm_ctx *create_from_host(SceneHost *h, int algorithm, float radiusFactor, float radiusAlpha, int seed, int device,
                                 int rank, int worldSize)
{
    if (worldSize < 1 || rank < 0 || rank >= worldSize) { delete h; fail("vcm_create", "bad rank/worldSize"); return NULL; }
    int ndev = vcm_device_count();
    if (ndev <= 0) { delete h; fail("vcm_create", "no HIP device available (this library has no CPU path)"); return NULL; }
    if (device < 0 || device >= ndev) { delete h; fail("vcm_create", "device index out of range"); return NULL; }
    vcm_ctx *c = new (std::nothrow) vcm_ctx();
    if (!c) { delete h; fail("vcm_create", "out of host memory"); return NULL; }
    memset((void *)c, 0, sizeof(*c));
    c->scene = h;
    scene_host_build_accel(*h, scene_host_force_bvh());
    /* VertexCM::VertexCM vertexcm.hxx:222-244 */
    switch (algorithm) {
    case VCM_ALGO_LIGHT_TRACE: c->lightTraceOnly = true; break;
    case VCM_ALGO_PPM: c->ppm = true; c->useVM = true; break;
    case VCM_ALGO_BPM: c->useVM = true; break;
    case VCM_ALGO_BPT: c->useVC = true; break;
    case VCM_ALGO_VCM: c->useVC = true; c->useVM = true; break;
    case VCM_ALGO_PATH_TRACE: c->renderer = 1; break;   /* PathTracer(scene, seed), config.hxx:120-121 */
    case VCM_ALGO_EYE_LIGHT: c->renderer = 2; break;    /* EyeLight(scene, seed), config.hxx:118-119 */
    default: delete h; delete c; fail("vcm_create", "unknown algorithm"); return NULL;
    }
    if (c->ppm) {   /* PPM -> BPM downgrade :246-278 */
        for (const vcm_material &m : h->materials) {
            const bool hasNonSpecular = (vmax3(ld3(m.diffuse)) > 0) || (vmax3(ld3(m.phong)) > 0);
            const bool hasSpecular = (vmax3(ld3(m.mirror)) > 0) || (m.ior > 0);
            if (hasNonSpecular && hasSpecular) {
                fprintf(stderr, "smallvcm_amd: scene mixes specular and non-specular BSDFs in one material, "
                                "switching from PPM to BPM (vertexcm.hxx:262-275)\n");
                c->ppm = false;
                break;
            }
        }
    }
    c->baseRadius = radiusFactor * h->sceneRadius;   /* :280 */
    c->radiusAlpha = radiusAlpha;
    c->seed = seed;
    c->device = device; c->rank = rank; c->world = worldSize;
    c->resX = (int)h->camera.resolution[0];
    c->resY = (int)h->camera.resolution[1];
    c->N = c->resX * c->resY;
    if (c->N <= 0) { delete h; delete c; fail("vcm_create", "empty resolution"); return NULL; }
    c->p0 = (int)((long long)c->N * rank / worldSize);
    c->nLocal = (int)((long long)c->N * (rank + 1) / worldSize) - c->p0;
    c->ownStream = true;
    if (worldSize == 1) {   /* single-rank contexts share the device's arenas */
        c->pool = pool_get(device);
        if (!c->pool) { delete h; delete c; fail("vcm_create", "device index out of range"); return NULL; }
        std::lock_guard<std::mutex> g(c->pool->m);
        c->pool->users++;
    } else {                /* a sharded context's iteration spans host-side collectives: private arena */
        c->arena = arena_new(device, NULL);
    }
    const char *so = getenv("SMALLVCM_AMD_STRICT_ORDER");
    c->strictOrder = (so && so[0] == '1');
    { const char *e = getenv("SMALLVCM_AMD_SORTED_EXCHANGE");   /* 0: the host will use the unsorted exchange of rounds 1-4 */
      c->sortedExchange = worldSize > 1 && worldSize <= VCM_SORTED_MAX_SHARDS && c->useVM && !(e && e[0] == '0'); }
    { const char *e = getenv("SMALLVCM_AMD_MERGE");
      c->mergeKind = (e && !strcmp(e, "walk")) ? VCM_MERGE_WALK : (e && !strcmp(e, "pairs")) ? VCM_MERGE_PAIRS : VCM_MERGE_DEFAULT; }
    return c;
}

vcm_ctx *vcm_create_sharded(const vcm_scene_desc *scene, int algorithm, float radiusFactor, float radiusAlpha,
                            int seed, int device, int rank, int worldSize)
{
    if (!scene) { fail("vcm_create", "scene is NULL"); return NULL; }
    SceneHost *h = new (std::nothrow) SceneHost();
    std::string err;
    if (!h || !scene_host_from_desc(*scene, *h, err)) { delete h; fail("vcm_create", err.c_str()); return NULL; }
    return create_from_host(h, algorithm, radiusFactor, radiusAlpha, seed, device, rank, worldSize);
}

vcm_ctx *vcm_create_sharded2(const vcm_scene_desc2 *scene, int algorithm, float radiusFactor, float radiusAlpha,
                             int seed, int device, int rank, int worldSize)
{
    if (!scene) { fail("vcm_create2", "scene is NULL"); return NULL; }
    SceneHost *h = new (std::nothrow) SceneHost();
    std::string err;
    if (!h || !scene_host_from_desc2(*scene, *h, err)) { delete h; fail("vcm_create2", err.c_str()); return NULL; }
    return create_from_host(h, algorithm, radiusFactor, radiusAlpha, seed, device, rank, worldSize);
}

/* Which device a renderer-per-host-core host puts its next renderer on (vcm_next_device).  The reference's driver
 * builds one renderer per host core and runs them concurrently (smallvcm.cxx:61-72, :99-108): on a multi-GPU node the
 * drop-in deals them round-robin over the visible devices -- every GPU renders whole iterations of its renderers
 * (replicas, no per-iteration exchange) and the driver's own framebuffer average (smallvcm.cxx:116-142) is the
 * reduce.  SMALLVCM_AMD_DEVICES = "all" (default of vcm_next_device), "current" (the calling thread's hipGetDevice)
 * or a list "0,2,5".  vcm_create itself stays on the caller's CURRENT device unless the variable is set: a host that
 * passes its own stream or device buffers allocated them there. */
static int next_device(bool createDefault)
{
    static std::mutex m;
    static std::vector<int> devs;
    static unsigned long long counter = 0;
    static bool current = false, init = false, envSet = false;
    std::lock_guard<std::mutex> g(m);
    const int n = vcm_device_count();
    if (!init) {
        init = true;
        const char *e = getenv("SMALLVCM_AMD_DEVICES");
        envSet = e && *e;
        if (e && !strcmp(e, "current")) current = true;
        else if (e && strcmp(e, "all") && *e) {
            for (const char *p = e; *p;) {
                char *end = NULL;
                const long d = strtol(p, &end, 10);
                if (end == p) break;
                if (d >= 0 && d < n) devs.push_back((int)d);
                p = (*end == ',') ? end + 1 : end;
            }
        }
        if (!current && devs.empty()) for (int d = 0; d < n; d++) devs.push_back(d);
    }
    if (current || devs.empty() || (createDefault && !envSet)) { int dev = 0; if (n > 0) (void)hipGetDevice(&dev); return dev; }
    return devs[(size_t)(counter++ % devs.size())];
}
int vcm_next_device(void) { return next_device(false); }

vcm_ctx *vcm_create(const vcm_scene_desc *scene, int algorithm, float radiusFactor, float radiusAlpha, int seed)
{
    return vcm_create_sharded(scene, algorithm, radiusFactor, radiusAlpha, seed, next_device(true), 0, 1);
}
vcm_ctx *vcm_create2(const vcm_scene_desc2 *scene, int algorithm, float radiusFactor, float radiusAlpha, int seed)
{
    return vcm_create_sharded2(scene, algorithm, radiusFactor, radiusAlpha, seed, next_device(true), 0, 1);
}

void vcm_destroy(vcm_ctx *c)
{
    if (!c) return;
    if (c->deviceReady) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->splat);   /* K5 of the last iteration */
    }
    arena_release(c, false);
    if (c->pool) {
        ArenaPool *p = c->pool;
        std::unique_lock<std::mutex> lk(p->m);
        for (int i = 0; i < p->count; i++)
            if (p->arenas[i]->lastUser == c) p->arenas[i]->lastUser = NULL;   /* the events stay valid: synchronised above */
        if (--p->users == 0) {   /* no context left, so nobody holds or wants an arena: give the memory back */
            (void)hipSetDevice(p->device);
            (void)hipDeviceSynchronize();
            for (int i = 0; i < p->count; i++) {
                Arena *a = p->arenas[i];
                if (a->allocated) arena_free_buffers(a);
                if (a->eventReady) (void)hipEventDestroy(a->lastUse);
                delete a;
                p->arenas[i] = NULL;
            }
            p->count = 0;
        }
    } else if (c->arena) {
        Arena *a = c->arena;
        if (a->allocated) { (void)hipSetDevice(a->device); (void)hipDeviceSynchronize(); arena_free_buffers(a); }
        if (a->eventReady) (void)hipEventDestroy(a->lastUse);
        delete a;
    }
    c->arena = NULL;
    delete c->scene;
    c->scene = NULL;
    if (c->deviceReady) {
        DFREE(c->dSpaceMatrix); DFREE(c->dSpaceScanned); DFREE(c->dSpaceTotals); DFREE(c->dSpaceHist); DFREE(c->dWhereDest); DFREE(c->dWherePos);
        DFREE(c->fQ); DFREE(c->fRes); DFREE(c->fKey); DFREE(c->fArrival); DFREE(c->fSorted); DFREE(c->fCount);
        c->dScene = NULL; DFREE(c->dSceneBlob); DFREE(c->dFb); DFREE(c->dRngLight); DFREE(c->dRngCam); DFREE(c->dHdr); DFREE(c->dStatsRing); DFREE(c->dStamps);
        for (int i = 0; i < EV_COUNT; i++) (void)hipEventDestroy(c->ev[i]);
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamSynchronize(c->splat);
        (void)hipEventDestroy(c->evSortFork); (void)hipEventDestroy(c->evSorted); (void)hipEventDestroy(c->evZero);
        (void)hipEventDestroy(c->evFork); (void)hipEventDestroy(c->evBbox); (void)hipEventDestroy(c->evGrid);
        (void)hipEventDestroy(c->evSplatFork); (void)hipEventDestroy(c->evSplatDone);
        (void)hipEventDestroy(c->evMergeDone); (void)hipEventDestroy(c->evSplatWork); (void)hipEventDestroy(c->evResolved);
        (void)hipEventDestroy(c->evPreMerge);
        (void)hipStreamDestroy(c->side);
        (void)hipStreamDestroy(c->splat);
        if (c->ownStream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

int vcm_set_strict_order(vcm_ctx *c, int on)
{
    if (!c) return fail("vcm_set_strict_order", "ctx is NULL");
    if (c->inIteration) return fail("vcm_set_strict_order", "iteration in progress");
    c->strictOrder = on != 0;
    return 0;
}

int vcm_set_merge_kernel(vcm_ctx *c, int kind)
{
    if (!c) return fail("vcm_set_merge_kernel", "ctx is NULL");
    if (c->inIteration) return fail("vcm_set_merge_kernel", "iteration in progress");
    if (kind == VCM_MERGE_LANE || kind == VCM_MERGE_STAGED) return fail("vcm_set_merge_kernel", "k_merge_lane / k_merge_staged were retired in round 6: VCM_MERGE_WALK or VCM_MERGE_PAIRS");
    if (kind != VCM_MERGE_WALK && kind != VCM_MERGE_PAIRS) return fail("vcm_set_merge_kernel", "unknown kernel");
    c->mergeKind = kind;
    return 0;
}

int vcm_set_stream(vcm_ctx *c, void *hipStream)
{
    if (!c) return fail("vcm_set_stream", "ctx is NULL");
    if (c->inIteration) return fail("vcm_set_stream", "iteration in progress");
    if (hipStream) {   /* kernels of this context are launched with c->device current: the stream must live there */
        int sdev = -1;
        const hipError_t e = hipStreamGetDevice((hipStream_t)hipStream, &sdev);
        (void)hipGetLastError();
        if (e == hipSuccess && sdev != c->device) return fail("vcm_set_stream", "the stream belongs to another device than the context");
    }
    if (c->deviceReady) {
        if (use_device(c)) return -1;
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->resolveInFlight) { HIPCHK(hipStreamSynchronize(c->splat)); c->resolveInFlight = false; c->splatInFlight = false; }
        if (c->ownStream) { (void)hipStreamDestroy(c->stream); }
    }
    if (hipStream) { c->stream = (hipStream_t)hipStream; c->ownStream = false; }
    else if (c->deviceReady) { HIPCHK(hipStreamCreate(&c->stream)); c->ownStream = true; }
    else { c->stream = NULL; c->ownStream = true; }
    return 0;
}

/* How many iteration-scratch arenas the single-rank contexts of `device` may share (1..8; 0 = as many as fit a
 * quarter of the device memory, the default; environment SMALLVCM_AMD_ARENAS).  1 serialises the iterations of
 * different renderers on the GPU. */
int vcm_set_arena_limit(int device, int maxArenas)
{
    if (maxArenas < 0 || maxArenas > VCM_MAX_ARENAS) return fail("vcm_set_arena_limit", "limit must be 0..8");
    ArenaPool *p = pool_get(device);
    if (!p) return fail("vcm_set_arena_limit", "device index out of range");
    std::lock_guard<std::mutex> g(p->m);
    p->limit = maxArenas;   /* arenas beyond a lowered limit stay until the last context of the device is destroyed */
    return 0;
}

static int vcm_begin_iteration_impl(vcm_ctx *c, int iteration, unsigned minLen, unsigned maxLen)
{   /* vertexcm.hxx:288-316 */
    if (!c) return fail("vcm_begin_iteration", "ctx is NULL");
    if (c->inIteration) return fail("vcm_begin_iteration", "previous iteration not ended");
    if (maxLen > 255) return fail("vcm_begin_iteration", "maxPathLength > 255 unsupported (8-bit vertex counts)");
    /* the two-launch scans (launch_scan_on) take at most 16384 tiles of 2048 entries: say so here, not in the middle of an
       iteration -- 33.5 M paths per rank, a 5792^2 frame */
    if ((long long)c->N / VCM_SCAN_TILE + 2 > 16384 || (long long)c->nLocal / VCM_SCAN_TILE + 2 > 16384)
        return fail("vcm_begin_iteration", "more than 33.5 M paths: the scans of the iteration are two-level (16384 tiles of 2048); shard the frame over more ranks");
    /* light-vertex slots per path / camera vertices per path; PathTracer and EyeLight store neither */
    const int S = c->renderer ? 1 : ((maxLen >= 2) ? (int)maxLen - 1 : 1);
    const int L = c->renderer ? 1 : ((maxLen >= 1) ? (int)maxLen : 1);
    if (ensure_device(c)) return -1;
    if (arena_acquire(c, S, L)) return -1;   /* the wrapper gives the arena back if it was borrowed */

    IterParams &P = c->P;
    memset(&P, 0, sizeof(P));
    P.seed = (uint32_t)c->seed;
    P.localIter = (uint32_t)c->iterations;
    P.minLen = minLen; P.maxLen = maxLen;
    P.resX = c->resX; P.resY = c->resY; P.N = c->N;
    P.p0 = c->p0; P.nLocal = c->nLocal; P.S = S;
    P.useVM = c->useVM; P.useVC = c->useVC; P.lightTraceOnly = c->lightTraceOnly; P.ppm = c->ppm;
    P.lightSubPathCount = float(c->resX * c->resY);                       /* :292 */
    float radius = c->baseRadius;                                         /* :295 */
    radius /= dm_powf(float(iteration + 1), 0.5f * (1 - c->radiusAlpha)); /* :296 */
    radius = smax(radius, 1e-7f);                                         /* :298 */
    const float radiusSqr = sqr(radius);
    P.radius = radius; P.radiusSqr = radiusSqr;
    P.vmNormalization = 1.f / (radiusSqr * VCM_PI_F * P.lightSubPathCount);   /* :303 */
    const float etaVCM = (VCM_PI_F * radiusSqr) * P.lightSubPathCount;        /* :306 */
    P.misVmWeightFactor = c->useVM ? mis(etaVCM) : 0.f;                       /* :307 */
    P.misVcWeightFactor = c->useVC ? mis(1.f / etaVCM) : 0.f;                 /* :308 */
    P.cellSize = radius * 2.f;                                                /* hashgrid.hxx:47 */
    P.invCellSize = 1.f / P.cellSize;                                         /* :48 */
    P.nCells = c->N;                                                          /* vertexcm.hxx:406 */
    P.wavefront = vcm_is_wavefront(c, maxLen);
    P.renderer = c->renderer;
    P.iteration = iteration;
    /* Queue blocks of K3: the unused tail of every wave's last block is holes the task kernels step over (half a
       block per wave and queue on average), while small blocks mean more atomics on one hot counter word.
       Measured (VC block 2048 / 1024 / 512 / 256): 703 / 708 / 689 / 584 Mpaths/s at 2048^2, 252 / 283 / 291 / 281
       at 512^2. */
    const bool big = c->nLocal >= (1 << 20);
    P.qblockVertex = big ? 512 : 256; P.qblockDI = big ? 512 : 256; P.qblockVC = big ? 1024 : 512;
    /* Query-sort bucket table: its scan and memset cost 0.3 ms per iteration at the full 2^24 entries whatever the
       resolution (the cell count follows the radius, not the pixel count), 14 % of a 512^2 iteration; with few
       queries per cell anyway a bucket may as well span a few cells there. */
    {
        const int perPath = shape_knob("buckets_per_path");   /* buckets of the query sort per path */
        /* (round 5: 4 buckets per path up to 1024^2 -- the in-line scan of the table is on the critical path there: 1024^2 scene 3
           1123 -> 1188 Mpaths/s, 512^2 588 -> 605, profiles/archive/r07h_ab_*.txt; 16 from 2^21 paths, where it makes no difference) */
        const int per = perPath ? perPath : (c->nLocal > (1 << 20) ? 16 : 4);
        long long nb = (long long)per * c->nLocal;
        if (nb < (1 << 18)) nb = 1 << 18;
        P.nBuckets = nb < VCM_QSORT_BUCKETS ? (int)nb : VCM_QSORT_BUCKETS;
    }
    P.foreignOut = 0;

    c->nPend[0] = c->nPend[1] = 0;
    if (mark(c, EV_START)) return -1;
    c->dStats = c->dStatsRing + (size_t)(c->iterations % VCM_STAMP_RING) * VCM_STAT_SLOTS;
    c->radiusRing[c->iterations % VCM_STAMP_RING] = radius;
    if (zero_ranges(c->stream, c->dStats, VCM_STAT_SLOTS * sizeof(unsigned long long), c->store.count, (size_t)c->nLocal /* :311-312 */,
                    c->vs.count, 32 * sizeof(int) /* queue counts [0..2]; chunk counter of K3 / k_path_trace [8] and of K1 [16] */,
                    c->dHdr, 6 * sizeof(uint32_t) /* bboxMinU / bboxMaxU: K1 accumulates into them with atomicMax */)) return -1;
    c->importedRecords = false;
    c->importedSorted = false;
    c->gridBuilt = c->cameraTraced = c->merged = c->splatsPending = c->recordsValid = c->countedInCamera = c->scatteredInDI = c->gridInFlight = c->splatInFlight = c->bboxPreset = c->bboxFromLight = c->bboxFinal = c->prezeroed = c->sortInFlight = c->spaceSet = c->mergeImported = false;
    c->splatInFlight = c->resolveInFlight;   /* whoever reads the framebuffer still has the last iteration's K5 to wait for */
    c->inIteration = true;
    c->evValid = false;
    return 0;
}

/* K1c + K1d: connect every light vertex to the camera and add the splats in the reference's
 * order (vertexcm.hxx:373-378 -> :863-934).  A sharded context defers this until the host has
 * started the all-gather of the light records (they are complete after K1b), so that the
 * exchange overlaps it; it must run before the grid build, whose scratch it borrows. */
/* workgroups of the dense task kernels (K1c, K3b, K3c: grid-stride loops over the tasks).  At 70-80 VGPRs 1536 of them
 * are resident (6 waves per SIMD); the 2048 of round 1 meant a second round that occupied a third of the chip.
 * Measured (1024 / 1536 / 1792 / 2048 / 3072 / 4608 / 8192): K3b+c 2.16 / 1.96 / 2.06 / 1.99 / 1.83 / 1.82 / 1.88 ms
 * next to the tail of the grid build (profiles/archive/r03j_ab_summary.txt). */
static int task_blocks(int nLocal)
{
    const int n = shape_knob("task_blocks");
    /* Small frames (round 5, profiles/archive/r07c_ab_*.txt, r07d_ab_*.txt; 400-iteration runs): rounds 1-4 kept 2048 workgroups up to
       1024^2 -- 8192 waves where ~6144 are resident, for one or two tasks per thread, next to K3c / K4 which want wave slots at
       the same time.  512 workgroups at 512^2 (five tasks per thread; 256 / 384 / 512 / 768 / 1024 / 1536 / 2048: 531 / 534 /
       543 / 503 / 497 / 447 / 426 Mpaths/s) and 1024 at 1024^2 (512 / 768 / 1024 / 1536 / 2048 / 3072: 1013 / 1036 / 1070 / 1006 /
       1001 / 957 on scene 3): +19 % and +7 % for the iteration. */
    if (n) return n;
    if (nLocal >= (1 << 21)) return 256 * 12;
    const int b = nLocal / 1024;
    return b < 512 ? 512 : (b > 1024 ? 1024 : b);
}
/* workgroups of k_merge_walk (multiple of 8: they are dealt to the XCDs).  1024 are resident (126 VGPRs, 37 KB of
 * LDS); with 2048 every workgroup walked ~20 batches of 256 queries and the last ones to finish set the kernel's time,
 * with 16384 it is two or three batches each: 3.29 -> 3.15 ms; beyond that the launch itself shows (32768: 3.63 ms;
 * profiles/archive/r03k, r03l). */
static int merge_blocks(int nLocal, int N)
{
    const int n = shape_knob("merge_blocks") >= 8 ? (shape_knob("merge_blocks") & ~7) : 0;
    /* smaller frames have fewer batches than that.  2048 until round 4 (16384 at 1024^2: 0.40 -> 1.08 ms, r03m); round 5
       measured the small end: 512^2 with 512 / 768 / 1024 / 1536 / 2048 / 4096 workgroups: K4 0.18 / 0.19 / 0.21 / 0.24 / 0.27 /
       0.31 ms (571 / 559 / 543 / 529 / 497 / 475 Mpaths/s with 512 task workgroups); 1024^2 scene 3 with 1024 / 2048 / 4096 /
       8192: K4 0.40 / 0.42 / 0.52 / 0.86 ms (profiles/archive/r07d_ab_*.txt) */
    if (n) return n;
    if (nLocal >= (1 << 21)) return 16384;
    int b = (nLocal / 1024) & ~7;
    b = b < 512 ? 512 : (b > 2048 ? 2048 : b);
    /* a SHARD of a large frame: its queries are as heavy as the whole frame's (the photon density follows the frame, 210
       candidates per query at 2048^2 against 29 at 512^2), so it gets the whole frame's workgroups per query */
    if (N >= (1 << 21) && nLocal < N) { const long long d = (16384ll * nLocal / N) & ~7ll; if (d > b) b = (int)d; }
    return b;
}
/* workgroups of the streaming helper kernels (compaction, grid build, splat lists, query scatter, resolve: grid-stride loops over
 * paths, vertices or pixels).  2048 = 524 288 threads; a 512^2 frame has 262 144 paths and ~560 000 vertices: half the threads
 * found nothing to do. */
static int aux_blocks(int nLocal)
{
    const int n = shape_knob("aux_blocks");
    return n ? n : 2048;
}
/* K5; `heavy` = beside the next iteration's K1 with the addends of a VC algorithm to replay: with 2048
   workgroups its 8192 waves took the wave slots K1's persistent waves were about to claim.  At 2048^2 VCM (K5 replays 41 M addends, 2.3 GB) 2048 / 1024 / 768 / 512 / 256 workgroups:
   1014 / 1035 / 1058 / 1062 / 1044 Mpaths/s (5 pairs of 40 iterations, profiles/archive/r11l_ab_summary.txt); BPM at 2048^2, 1024^2 and
   512^2 do not care down to 512 and lose below (r11m). */
static int resolve_blocks(int nLocal, bool heavy) { const int n = shape_knob("resolve_blocks"); return n ? n : ((heavy && nLocal >= (1 << 21)) ? 512 : aux_blocks(nLocal)); }
/* K2's sort (vcm_kernels.h, "K2 as a radix sort"): SMALLVCM_AMD_GRID_SORT=count keeps the reference's counting sort with one
   atomic per vertex (rounds 1-5); the default is the radix sort. */
static bool grid_sort_is_radix(const vcm_ctx *c)
{
    static int mode = -2;
    if (mode == -2) { const char *e = getenv("SMALLVCM_AMD_GRID_SORT"); mode = !e ? -1 : (!strcmp(e, "radix") ? 1 : (!strcmp(e, "count") ? 0 : -1)); }
    return mode != 0;
}
static int radix_sort_blocks(int nLocal)
{
    const int forced = shape_knob("grid_sort_blocks");
    if (forced) return forced < VCM_RSORT_MAX_BLOCKS ? forced : VCM_RSORT_MAX_BLOCKS;
    int v = nLocal / 2048;   /* ~2200 vertices per workgroup at the reference's path lengths (4096 / 2048 / 1024 / 512 workgroups at 2048^2: 1000-1009 / 1006-1016 / 1001-1022 / 1008-1019 Mpaths/s, profiles/archive/r11h) */
    return v < 64 ? 64 : (v > VCM_RSORT_MAX_BLOCKS ? VCM_RSORT_MAX_BLOCKS : v);
}
/* the main stream continues only after the splat stream's K1c / K1d (before anything else touches the framebuffer) */
static int join_splats(vcm_ctx *c)
{
    if (!c->splatInFlight) return 0;
    HIPCHK(hipStreamWaitEvent(c->stream, c->evSplatDone, 0));
    c->splatInFlight = false;
    c->resolveInFlight = false;   /* (evSplatDone is recorded behind K5 as well) */
    return 0;
}
static int flush_light_splats(vcm_ctx *c)
{
    if (!c->splatsPending) return 0;
    c->splatsPending = false;
    {
        /* K1c / K1d only read the light-vertex store and add to the framebuffer; nothing of the camera pass touches the
           framebuffer before K5.  They run on a stream of their own next to the grid build and the camera pass and are
           joined before K5: +4.5 % at 512^2, +7 % at 1024^2, +3 % at 2048^2 (profiles/archive/r05c_ab_summary.txt; in round 1,
           when they still shared their scratch with the grid build, the overlap had bought nothing).  Strict order keeps them in line. */
        /* (round 5: sharded contexts too -- a rank's K1c / K1d then run beside its camera pass while its light vertices
           travel, instead of in line in front of it) */
        const bool overlap = !c->strictOrder;
        hipStream_t q = overlap ? c->splat : c->stream;
        const StampArgs none = { { NULL, NULL, NULL, NULL } };
        if (overlap) {
            HIPCHK(hipEventRecord(c->evSplatFork, c->stream));
            HIPCHK(hipStreamWaitEvent(q, c->evSplatFork, 0));
        } else if (join_splats(c)) return -1;   /* in line: K5 of the LAST iteration may still be adding to dFb on the splat stream */
        int *pixCount = c->dPixCount, *arrival = c->dSplatArrival, *pixStart = c->dPixStart;
        F4 *list = c->dSplatList;
        if (c->prezeroed) HIPCHK(hipStreamWaitEvent(q, c->evZero, 0));
        else if (zero_ranges(q, pixCount, ((size_t)c->N + 1) * sizeof(int))) return -1;
        LAUNCH_SC(c, k_connect_camera, dim3(task_blocks(c->nLocal)), dim3(256), 0, q, c->dScene, c->P, c->store,
                           (const int *)c->dSlotOfVertex, (const int *)c->dLocalTotal, c->dFb, c->dSplat, pixCount,
                           arrival, c->dStats);
        if (launch_scan_on<int>(c, overlap ? 2 : 0, q, pixCount, c->N, pixStart, NULL, 1, overlap ? none : take_stamps(c, q))) return -1;
        hipLaunchKernelGGL(k_splat_scatter, dim3(aux_blocks(c->nLocal)), dim3(256), 0, q, (const F4 *)c->dSplat,
                           (const int *)c->dLocalTotal, (const int *)pixStart, (const int *)arrival, list, pixCount);
        /* pixels with more than VCM_SPLAT_REG splats are queued (pixCount[0] = their number, `arrival` = the queue:
           both dead since the scatter) and handled by one wave each; `sorted` = the vertex-ordered splat array */
        static int splatLong = -1;   /* SMALLVCM_AMD_SPLAT_LONG: tests send short lists down the one-wave-per-pixel path too */
        if (splatLong < 0) { const char *e = getenv("SMALLVCM_AMD_SPLAT_LONG"); splatLong = (e && atoi(e) >= VCM_SPLAT_REG) ? atoi(e) : VCM_SPLAT_LONG; }
        hipLaunchKernelGGL(k_splat_apply, dim3(aux_blocks(c->nLocal)), dim3(256), 0, q, c->N, (const int *)pixStart,
                           (const F4 *)list, c->dFb, arrival, pixCount, splatLong);
        hipLaunchKernelGGL(k_splat_apply_long, dim3(1024), dim3(256), 0, q, (const int *)pixStart, (const F4 *)list,
                           c->dSplat, c->dFb, (const int *)arrival, (const int *)pixCount);
        HIPCHK(hipGetLastError());
        if (overlap) {
            HIPCHK(hipEventRecord(c->evSplatDone, q));
            c->splatInFlight = true;
        }
    }
    return 0;
}

static int vcm_trace_light_impl(vcm_ctx *c)
{   /* vertexcm.hxx:321-396 */
    if (!c || !c->inIteration) return fail("vcm_trace_light", "no iteration in progress");
    if (use_device(c)) return -1;
    if (c->renderer) {   /* PathTracer / EyeLight have no light pass */
        if (mark(c, EV_LIGHT_K0)) return -1;
        if (mark(c, EV_LIGHT_K1)) return -1;
        HIPCHK(hipMemsetAsync(c->dLocalTotal, 0, sizeof(int), c->stream));
        if (mark(c, EV_LIGHT)) return -1;   /* written by the phase's last kernel as it starts */
        hipLaunchKernelGGL(k_set_counts, dim3(1), dim3(1), 0, c->stream, c->dHdr, c->dLocalTotal, 1, 0, take_stamps(c, c->stream));
        HIPCHK(hipGetLastError());
        return 0;
    }
    int blocks, chunk;
    trace_launch_shape(c->nLocal, &blocks, &chunk, true);
    if (c->world == 1 && !c->strictOrder) {
        /* The tables the passes after K1 count into -- cells of the hash grid, pixels of the light splats, buckets of the
           query sort: 100 MB at 2048^2 -- are zeroed NOW, on the side stream next to K1 (VALU-bound, 0.9 ms).  Zeroed where
           they are used, the two memsets sat between K1 and K3 on the critical path and took 0.3 + 0.5 ms there, because
           they shared the memory system with the light splats and the cell count (profiles/archive/r06d_timeline2048.txt: K3
           started 0.9 ms after K1 had ended).  The consumers wait for evZero on their own streams. */
        HIPCHK(hipEventRecord(c->evFork, c->stream));   /* behind the previous users of the arena */
        HIPCHK(hipStreamWaitEvent(c->side, c->evFork, 0));
        if (zero_ranges(c->side, c->useVM ? c->dQueryCount : NULL, c->useVM ? ((size_t)c->P.nBuckets + 1) * sizeof(int) : 0,
                        (c->useVM && !grid_sort_is_radix(c)) ? c->dCellCount : NULL, c->useVM ? ((size_t)c->P.nCells + 1) * sizeof(int) : 0,
                        (c->useVC || c->lightTraceOnly) ? c->dPixCount : NULL, ((size_t)c->N + 1) * sizeof(int))) return -1;
        HIPCHK(hipEventRecord(c->evZero, c->side));
        c->prezeroed = true;
    }
    if (mark(c, EV_LIGHT_K0)) return -1;
    const bool wf = !c->strictOrder;
    if (!wf && join_splats(c)) return -1;   /* the fused K1 splats straight into dFb: behind a K5 the last (aside) iteration left in flight */
    if (wf)
        LAUNCH_SC_MODE(c, k_light_trace, 1, dim3(blocks), dim3(VCM_TRACE_BLOCK), 0, c->stream, c->dScene, c->P, c->store,
                           c->dFb, c->dRngLight, c->dStats, chunk, take_stamps(c, c->stream), c->vs.count + 16, c->dHdr);
    else
        LAUNCH_SC_MODE(c, k_light_trace, 0, dim3(blocks), dim3(VCM_TRACE_BLOCK), 0, c->stream, c->dScene, c->P, c->store,
                           c->dFb, c->dRngLight, c->dStats, chunk, take_stamps(c, c->stream), c->vs.count + 16, c->dHdr);
    HIPCHK(hipGetLastError());
    c->bboxFromLight = c->world == 1;   /* K1 keeps the box of what it stores (k_bbox, a pass of its own over the vertices, serves imported records) */
    if (mark(c, EV_LIGHT_K1)) return -1;
    /* mPathEnds (:395) = scan of the per-path counts, then the contiguous
       record array in the reference's vertex order */
    if (launch_scan<unsigned char>(c, c->store.count, c->nLocal, c->dPathStart, c->dLocalTotal, 0)) return -1;
    /* K4 of the LAST iteration (merge stream) reads the grid header's box and counts, the cell ranges and the camera vertices:
       everything from here on rewrites them (compaction first: the header), so the main stream waits for it now -- K1 above ran
       beside it */
    if (c->mergeInFlight) { HIPCHK(hipStreamWaitEvent(c->stream, c->evMergeDone, 0)); c->mergeInFlight = false; }
    bool countsSet = false;
    if (c->useVM || wf) {
        /* a sharded renderer ships the records to the other ranks; a single-rank one builds its grid straight
           from the store and materialises them only when somebody asks (ensure_records).  On a single rank the kernel
           also publishes the counts and the box K1 kept (k_set_counts / k_bbox_finalize folded in). */
        c->recordsValid = c->useVM && c->world > 1 && !c->sortedExchange;   /* (vcm_export_light_records still materialises them on demand) */
        const bool fold = c->world == 1;
        hipLaunchKernelGGL(k_compact_records, dim3(aux_blocks(c->nLocal)), dim3(256), 0, c->stream, (const DScene *)c->dScene, c->P, c->store, c->dPathStart,
                           c->dRecordsLocal, c->dSlotOfVertex, c->recordsValid ? 1 : 0, fold ? c->dHdr : (GridHeader *)NULL,
                           (const int *)c->dLocalTotal, (fold && c->bboxFromLight) ? 1 : 0);
        HIPCHK(hipGetLastError());
        countsSet = fold;
        if (fold && c->bboxFromLight) c->bboxFinal = true;
    }
    if (wf && (c->useVC || c->lightTraceOnly)) {
        c->splatsPending = true;
        if (c->world == 1 && flush_light_splats(c)) return -1;
    }
    if (mark(c, EV_LIGHT)) return -1;   /* written by the next kernel of the stream as it starts */
    if (!countsSet) {
        hipLaunchKernelGGL(k_set_counts, dim3(1), dim3(1), 0, c->stream, c->dHdr, c->dLocalTotal, 1, 0, take_stamps(c, c->stream));
        HIPCHK(hipGetLastError());
    }
    return 0;
}

/* materialise the 13-float merge records of the local light vertices (reference order) if K1b skipped them */
static int ensure_records(vcm_ctx *c)
{
    if (c->recordsValid || !c->useVM) return 0;
    hipLaunchKernelGGL(k_compact_records, dim3(aux_blocks(c->nLocal)), dim3(256), 0, c->stream, (const DScene *)c->dScene, c->P, c->store, c->dPathStart,
                       c->dRecordsLocal, c->dSlotOfVertex, 1, (GridHeader *)NULL, (const int *)c->dLocalTotal, 0);
    HIPCHK(hipGetLastError());
    c->recordsValid = true;
    return 0;
}

int vcm_light_records(vcm_ctx *c, void **devPtr, long long *count)
{
    if (scratch_readable(c, "vcm_light_records")) return -1;
    if (use_device(c)) return -1;
    int n = 0;
    if (devPtr && ensure_records(c)) return -1;
    HIPCHK(hipMemcpyAsync(&n, c->dLocalTotal, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (devPtr) *devPtr = c->dRecordsLocal;
    if (count) *count = n;
    return 0;
}

/* The bounding box of this rank's light vertices and their number, after vcm_trace_light: what a sharded host
 * exchanges (7 numbers per rank) instead of only the count, so that every rank knows the box of ALL vertices
 * before they have arrived (HashGrid::Build takes it over the whole array, hashgrid.hxx:50-61; min / max over
 * ranks of the per-rank min / max is the same box, bit for bit).  Empty: min = +1e36, max = -1e36 (:47-48). */
static int vcm_local_light_bbox_impl(vcm_ctx *c, float *min3, float *max3, long long *count)
{
    if (!c || !c->inIteration || !min3 || !max3) return fail("vcm_local_light_bbox", "call it between vcm_trace_light and vcm_build_grid");
    if (use_device(c)) return -1;
    if (!c->renderer && !c->bboxFinal) {
        /* K1 kept the box of what it stored in the header's key words (minimum inverted), as on a single rank: one tiny
           launch turns them into floats -- rounds 2-4 gathered every local vertex's position once more for this (k_bbox) */
        hipLaunchKernelGGL(k_bbox_finalize, dim3(1), dim3(64), 0, c->stream, c->dHdr, 1);
    } else if (!c->bboxFinal) {
        VertexSource src; src.records = NULL; src.store = c->store; src.slotOfVertex = c->dSlotOfVertex;
        hipLaunchKernelGGL(k_grid_init, dim3(1), dim3(64), 0, c->stream, c->dHdr, take_stamps(c, c->stream));
        hipLaunchKernelGGL(k_bbox, dim3(512), dim3(256), 0, c->stream, src, c->dHdr);
        hipLaunchKernelGGL(k_bbox_finalize, dim3(1), dim3(64), 0, c->stream, c->dHdr, 0);
    }
    HIPCHK(hipGetLastError());
    GridHeader h;
    HIPCHK(hipMemcpyAsync(&h, c->dHdr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 3; i++) { min3[i] = h.bboxMin[i]; max3[i] = h.bboxMax[i]; }
    if (count) *count = h.nLocalRecords;
    /* the key words K1 left (minimum inverted) have been overwritten by k_grid_init / k_bbox: vcm_build_grid must not
       finalise them a second time.  On a single rank the box just computed IS the box of all vertices
       (hashgrid.hxx:50-61); a sharded host still has to call vcm_set_grid_bbox with the box of all ranks. */
    c->bboxFromLight = false;
    if (c->world == 1) c->bboxPreset = true;
    return 0;
}

/* The box of all ranks' vertices, computed by the host from vcm_local_light_bbox of every rank.  With it the
 * camera pass can take over the query-sort histogram (it needs the box, not the grid) although the grid is built
 * only after the exchange, and vcm_build_grid skips its own reduction. */
static int vcm_set_grid_bbox_impl(vcm_ctx *c, const float *min3, const float *max3)
{
    if (!c || !c->inIteration || !min3 || !max3) return fail("vcm_set_grid_bbox", "no iteration in progress");
    if (c->gridBuilt) return fail("vcm_set_grid_bbox", "the grid is built already");
    if (use_device(c)) return -1;
    hipLaunchKernelGGL(k_set_bbox, dim3(1), dim3(1), 0, c->stream, c->dHdr, min3[0], min3[1], min3[2], max3[0], max3[1], max3[2]);
    HIPCHK(hipGetLastError());
    for (int k = 0; k < 3; k++) c->presetMin[k] = min3[k];
    c->bboxPreset = true;
    return 0;
}

int vcm_export_light_records(vcm_ctx *c, void *dstDev, long long count)
{
    if (!dstDev) return fail("vcm_export_light_records", "bad argument");
    if (scratch_readable(c, "vcm_export_light_records")) return -1;
    if (use_device(c)) return -1;
    if (ensure_records(c)) return -1;
    if (count > 0)
        HIPCHK(hipMemcpyAsync(dstDev, c->dRecordsLocal, (size_t)count * VCM_MERGE_RECORD_FLOATS * sizeof(float),
                              hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

static __global__ void k_scale_copy(const float *src, float *dst, size_t n, float scale)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i] * scale;
}
/* dst = framebuffer * scale (renderer.hxx:53-54 with scale = 1 / mIterations), device to device, asynchronous */
int vcm_export_framebuffer_scaled(vcm_ctx *c, void *dstDev, float scale)
{
    if (!c || !dstDev) return fail("vcm_export_framebuffer_scaled", "bad argument");
    if (ensure_device(c)) return -1;
    if (join_splats(c)) return -1;   /* the light splats of an open iteration run on a stream of their own */
    hipLaunchKernelGGL(k_scale_copy, dim3(1024), dim3(256), 0, c->stream, (const float *)c->dFb, (float *)dstDev, (size_t)c->N * 3, scale);
    HIPCHK(hipGetLastError());
    return 0;
}

int vcm_export_framebuffer(vcm_ctx *c, void *dstDev)
{
    if (!c || !dstDev) return fail("vcm_export_framebuffer", "bad argument");
    if (ensure_device(c)) return -1;
    if (join_splats(c)) return -1;
    HIPCHK(hipMemcpyAsync(dstDev, c->dFb, (size_t)c->N * 3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

static int vcm_import_light_records_impl(vcm_ctx *c, const void *devPtr, const long long *counts, int nSeg, long long strideRecords)
{
    if (!c || !c->inIteration) return fail("vcm_import_light_records", "no iteration in progress");
    if (c->world <= 1 || !c->dRecordsAll) return fail("vcm_import_light_records", "context is not sharded");
    if (use_device(c)) return -1;
    long long total = 0;
    const size_t recBytes = VCM_MERGE_RECORD_FLOATS * sizeof(float);
    for (int s = 0; s < nSeg; s++) {
        if (counts[s] < 0) return fail("vcm_import_light_records", "negative count");
        if (total + counts[s] > (long long)c->arena->capS * (long long)c->arena->capN) return fail("vcm_import_light_records", "too many records");
        if (counts[s] > 0)
            HIPCHK(hipMemcpyAsync((char *)c->dRecordsAll + (size_t)total * recBytes,
                                  (const char *)devPtr + (size_t)s * (size_t)strideRecords * recBytes,
                                  (size_t)counts[s] * recBytes, hipMemcpyDeviceToDevice, c->stream));
        total += counts[s];
    }
    hipLaunchKernelGGL(k_set_counts, dim3(1), dim3(1), 0, c->stream, c->dHdr, c->dLocalTotal, 0, (int)total, take_stamps(c, c->stream));
    HIPCHK(hipGetLastError());
    c->importedRecords = true;
    return 0;
}

/* the vertices of `recs` sorted by cell, stable: cellStart and the list {vertex, slot} in the grid's order (returned) */
static const I2 *sort_cells_radix(vcm_ctx *c, hipStream_t q, int scanSlot, const VertexSource &recs)
{
    const int nCells = c->P.nCells;
    int bits = 1;
    while (bits < 31 && (1ll << bits) < (long long)nCells) bits++;
    const int passes = (bits + 7) / 8;
    const int V = radix_sort_blocks(c->nLocal);
    static_assert(256 * VCM_RSORT_MAX_BLOCKS <= VCM_QSORT_BUCKETS, "the scan scratch (dTileSums) is sized for the bucket table: the digit matrix must not be larger");
    const StampArgs none = { { NULL, NULL, NULL, NULL } };
    uint32_t *key[2] = { (uint32_t *)c->dCellId, (uint32_t *)c->dSortedIndex };   /* (the index is written by the gather, after the keys are dead) */
    I2 *pay[2] = { (I2 *)c->dUnsorted, (I2 *)c->dUnsorted + (size_t)c->arena->capS * c->arena->capN };   /* an I4 per record = two lists of I2 */
    int *hist = c->dRadixHist, *scanned = c->dRadixHist + 256 * VCM_RSORT_MAX_BLOCKS;
    hipLaunchKernelGGL(k_cell_keys, dim3(V), dim3(256), 0, q, c->P, recs, (const GridHeader *)c->dHdr, key[0], pay[0], hist, take_stamps(c, q));
    int cur = 0;
    for (int p = 0; p < passes; p++) {
        if (p > 0) hipLaunchKernelGGL(k_radix_hist, dim3(V), dim3(256), 0, q, (const uint32_t *)key[cur], (const GridHeader *)c->dHdr, 8 * p, hist);
        if (hipGetLastError() != hipSuccess) { fail("vcm_build_grid", "radix histogram launch"); return NULL; }
        if (launch_scan_on<int>(c, scanSlot, q, hist, 256 * V, scanned, NULL, 0, none)) return NULL;
        hipLaunchKernelGGL(k_radix_scatter, dim3(V), dim3(256), 0, q, (const uint32_t *)key[cur], (const I2 *)pay[cur], key[cur ^ 1], pay[cur ^ 1],
                           (const GridHeader *)c->dHdr, 8 * p, (const int *)scanned);
        cur ^= 1;
    }
    hipLaunchKernelGGL(k_cell_starts, dim3(aux_blocks(c->nLocal)), dim3(256), 0, q, (const uint32_t *)key[cur], (const GridHeader *)c->dHdr, nCells, c->dCellStart);
    if (hipGetLastError() != hipSuccess) { fail("vcm_build_grid", "radix sort launch"); return NULL; }
    return pay[cur];
}

/* ---- the sorted exchange of a sharded renderer (include/smallvcm_amd.h; kernels: vcm_kernels.h "K2 of a SHARDED renderer") ---- */
static int sorted_shape(vcm_ctx *c, long long strideRecords, int *K, int *nBlocks, long long *slabWords, const char *who)
{
    if (!c) return fail(who, "ctx is NULL");
    if (c->world <= 1) return fail(who, "the context is not sharded");
    if (!c->useVM) return fail(who, "the algorithm does not merge: nothing to exchange");
    if (c->world > VCM_SORTED_MAX_SHARDS) return fail(who, "more than 64 shards: use the unsorted exchange");
    if (strideRecords < 1 || strideRecords >= (1ll << 24)) return fail(who, "1 <= records per shard < 2^24 (the index shares a word with the path length): use the unsorted exchange");
    *K = sorted_block_cells(c->world);
    *nBlocks = (c->N + *K - 1) / *K;   /* nCells = pathCount (vertexcm.hxx:406) */
    const long long w = strideRecords * VCM_SORTED_WORDS + (long long)*nBlocks + 1;
    *slabWords = (w + 3) & ~3ll;
    return 0;
}
extern "C" long long vcm_sorted_slab_words(vcm_ctx *c, long long strideRecords)
{
    int K, nBlocks; long long words;
    if (sorted_shape(c, strideRecords, &K, &nBlocks, &words, "vcm_sorted_slab_words")) return -1;
    return words;
}
static int vcm_sort_light_records_impl(vcm_ctx *c, void *dstDev, long long strideRecords)
{
    if (!c || !c->inIteration || !dstDev) return fail("vcm_sort_light_records", "call it between vcm_set_grid_bbox and vcm_build_grid");
    if (!c->bboxPreset) return fail("vcm_sort_light_records", "call vcm_set_grid_bbox first: the cell of a vertex depends on the box of all ranks' vertices");
    if (c->gridBuilt) return fail("vcm_sort_light_records", "the grid is built already");
    int K, nBlocks; long long words;
    if (sorted_shape(c, strideRecords, &K, &nBlocks, &words, "vcm_sort_light_records")) return -1;
    if (use_device(c)) return -1;
    /* HashGrid::Build (hashgrid.hxx:41-107) over THIS rank's vertices, with the box of all of them: the kernels of the
       single-rank build, 1 / worldSize of its work; hdr->nRecords still is the local count here (k_set_counts) */
    const int nCells = c->P.nCells;
    const dim3 g(aux_blocks(c->nLocal)), b(256);
    VertexSource recs; recs.records = c->recordsValid ? c->dRecordsLocal : NULL; recs.store = c->store; recs.slotOfVertex = c->dSlotOfVertex;
    const I2 *sorted = NULL;
    if (grid_sort_is_radix(c)) {
        if (!(sorted = sort_cells_radix(c, c->stream, 0, recs))) return -1;
    } else {
        if (c->prezeroed) HIPCHK(hipStreamWaitEvent(c->stream, c->evZero, 0));   /* the side stream's zeroing must not overtake the counts */
        if (zero_ranges(c->stream, c->dCellCount, ((size_t)nCells + 1) * sizeof(int))) return -1;
        hipLaunchKernelGGL(k_cell_count, g, b, 0, c->stream, c->P, recs, (const GridHeader *)c->dHdr, c->dCellId, c->dSortedIndex, c->dCellCount,
                           take_stamps(c, c->stream));
        HIPCHK(hipGetLastError());
        if (launch_scan<int>(c, c->dCellCount, nCells, c->dCellStart, NULL, 1)) return -1;
        hipLaunchKernelGGL(k_cell_scatter, g, b, 0, c->stream, (const GridHeader *)c->dHdr, (const int *)c->dCellId, (const int *)c->dSortedIndex,
                           (const int *)c->dCellStart, recs.records ? (const int *)NULL : (const int *)c->dSlotOfVertex, c->dUnsorted);
    }
    uint32_t *slab = (uint32_t *)dstDev;
    hipLaunchKernelGGL(k_cell_rank_pack, g, b, 0, c->stream, (const DScene *)c->dScene, (const GridHeader *)c->dHdr, recs, (const int *)c->dCellStart,
                       (const I4 *)c->dUnsorted, sorted, slab, (int *)(slab + (size_t)strideRecords * VCM_SORTED_WORDS), nCells, K, nBlocks);
    HIPCHK(hipGetLastError());
    return 0;
}
static int vcm_import_sorted_light_records_impl(vcm_ctx *c, const void *gathered, const long long *counts, int nSeg, long long strideRecords)
{
    if (!c || !c->inIteration || !gathered || !counts) return fail("vcm_import_sorted_light_records", "no iteration in progress");
    if (nSeg != c->world) return fail("vcm_import_sorted_light_records", "one slab per rank");
    if (nSeg > VCM_SORTED_MAX_SHARDS) return fail("vcm_import_sorted_light_records", "more than 64 shards: use the unsorted exchange");
    int K, nBlocks; long long words;
    if (sorted_shape(c, strideRecords, &K, &nBlocks, &words, "vcm_import_sorted_light_records")) return -1;
    if (use_device(c)) return -1;
    SortedSlabs &in = c->sortedIn;
    in.base = (const uint32_t *)gathered; in.slabWords = words; in.strideRecords = strideRecords; in.S = nSeg; in.K = K; in.nBlocks = nBlocks;
    long long total = 0;
    for (int s = 0; s < nSeg; s++) {
        if (counts[s] < 0 || counts[s] > strideRecords) return fail("vcm_import_sorted_light_records", "a count exceeds the slab");
        in.rankBase[s] = (int)total;
        total += counts[s];
    }
    in.rankBase[nSeg] = (int)total;
    if (total > (long long)c->arena->capS * (long long)c->arena->capN) return fail("vcm_import_sorted_light_records", "too many records");
    hipLaunchKernelGGL(k_set_counts, dim3(1), dim3(1), 0, c->stream, c->dHdr, c->dLocalTotal, 0, (int)total, take_stamps(c, c->stream));
    HIPCHK(hipGetLastError());
    c->importedSorted = true;
    c->importedRecords = false;
    return 0;
}
extern "C" int vcm_sort_light_records(vcm_ctx *c, void *dstDev, long long strideRecords)
{
    g_hipFailed = false;
    return abort_iteration(c, vcm_sort_light_records_impl(c, dstDev, strideRecords));
}
extern "C" int vcm_import_sorted_light_records(vcm_ctx *c, const void *gathered, const long long *counts, int nSeg, long long strideRecords)
{
    g_hipFailed = false;
    return abort_iteration(c, vcm_import_sorted_light_records_impl(c, gathered, counts, nSeg, strideRecords));
}

/* the main stream continues only after the side stream's grid build */
static int join_grid(vcm_ctx *c)
{
    if (!c->gridInFlight) return 0;
    HIPCHK(hipStreamWaitEvent(c->stream, c->evGrid, 0));
    c->gridInFlight = false;
    return 0;
}

static int vcm_build_grid_impl(vcm_ctx *c)
{   /* vertexcm.hxx:403-408 -> HashGrid::Build hashgrid.hxx:41-107 */
    if (!c || !c->inIteration) return fail("vcm_build_grid", "no iteration in progress");
    if (use_device(c)) return -1;
    if (flush_light_splats(c)) return -1;
    c->gridBuilt = true;
    if (c->useVM) {
        /* The grid build is atomic- and gather-bound, the camera pass that follows in the single-rank order
           (K3, K3b, K3c) is VALU-bound and does not read the grid: the build runs on a side stream next to it.
           Main waits for the bounding box only (K3 derives the query-sort keys from it); vcm_merge waits for the
           rest.  Strict mode merges inside K3 and waits at once. */
        hipStream_t q = c->side;
        HIPCHK(hipEventRecord(c->evFork, c->stream));
        HIPCHK(hipStreamWaitEvent(q, c->evFork, 0));
        if (mark_on(c, EV_GRID_K0, q)) return -1;
        if (c->importedSorted) {
            /* every rank sorted its own vertices (vcm_sort_light_records); what is left of HashGrid::Build is ONE streaming
               merge of the slabs, cell block by cell block: cellStart, the cell-sorted arrays, the index for the read-out */
            if (!c->bboxPreset) return fail("vcm_build_grid", "sorted records without the box they were sorted with");
            const int blocks = c->sortedIn.nBlocks < 8192 ? c->sortedIn.nBlocks : 8192;
            hipLaunchKernelGGL(k_grid_merge_blocks, dim3(blocks), dim3(256), 0, q, c->P, (const GridHeader *)c->dHdr, c->sortedIn, c->dCellStart,
                               c->dGx, c->dGy, c->dGz, c->dGb, c->dG1, c->dG2, c->dG3, c->dSortedIndex, take_stamps(c, q));
            HIPCHK(hipGetLastError());
            if (mark_on(c, EV_GRID, q)) return -1;
            hipLaunchKernelGGL(k_note_grid_vertices, dim3(1), dim3(1), 0, q, (const GridHeader *)c->dHdr, c->dStats + STAT_COUNT, take_stamps(c, q));
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c->evGrid, q));
            c->gridInFlight = true;
            if (!c->P.wavefront && join_grid(c)) return -1;
            return 0;
        }
        VertexSource recs;
        recs.records = c->importedRecords ? c->dRecordsAll : (c->recordsValid ? c->dRecordsLocal : NULL);
        recs.store = c->store;
        recs.slotOfVertex = c->dSlotOfVertex;
        const int nCells = c->P.nCells;
        const dim3 g(aux_blocks(c->nLocal)), b(256);
        const bool radix = grid_sort_is_radix(c);
        if (c->prezeroed) HIPCHK(hipStreamWaitEvent(q, c->evZero, 0));   /* (q is the side stream itself) */
        else if (!radix && zero_ranges(q, c->dCellCount, ((size_t)nCells + 1) * sizeof(int))) return -1;
        bool boxOnSide = false;
        if (!c->bboxPreset) {   /* a sharded host has exchanged the ranks' boxes already (vcm_set_grid_bbox) */
            if (c->bboxFromLight && !recs.records) {   /* K1 left the box of what it stored in the header's key words */
                if (!c->bboxFinal) { hipLaunchKernelGGL(k_bbox_finalize, dim3(1), dim3(64), 0, q, c->dHdr, 1); boxOnSide = true; }
            } else {
                boxOnSide = true;
                hipLaunchKernelGGL(k_grid_init, dim3(1), dim3(64), 0, q, c->dHdr, take_stamps(c, q));
                hipLaunchKernelGGL(k_bbox, dim3(512), b, 0, q, recs, c->dHdr);
                hipLaunchKernelGGL(k_bbox_finalize, dim3(1), dim3(64), 0, q, c->dHdr, 0);
            }
        }
        HIPCHK(hipEventRecord(c->evBbox, q));
        const I2 *sorted = NULL;
        if (radix) {
            if (!(sorted = sort_cells_radix(c, q, 1, recs))) return -1;
        } else {
            hipLaunchKernelGGL(k_cell_count, g, b, 0, q, c->P, recs, (const GridHeader *)c->dHdr, c->dCellId,
                               c->dSortedIndex /* arrival: dead before k_cell_rank_gather writes the index */, c->dCellCount, take_stamps(c, q));
            HIPCHK(hipGetLastError());
            if (launch_scan_on<int>(c, 1, q, c->dCellCount, nCells, c->dCellStart, NULL, 1, take_stamps(c, q))) return -1;
            hipLaunchKernelGGL(k_cell_scatter, g, b, 0, q, (const GridHeader *)c->dHdr, (const int *)c->dCellId,
                               (const int *)c->dSortedIndex, (const int *)c->dCellStart,
                               recs.records ? (const int *)NULL : (const int *)c->dSlotOfVertex, c->dUnsorted);
        }
        hipLaunchKernelGGL(k_cell_rank_gather, g, b, 0, q, (const DScene *)c->dScene, (const GridHeader *)c->dHdr, recs,
                           (const int *)c->dCellStart, (const I4 *)c->dUnsorted, sorted, c->dGx,
                           c->dGy, c->dGz, c->dGb, c->dG1, c->dG2, c->dG3, c->dSortedIndex);
        if (mark_on(c, EV_GRID, q)) return -1;
        hipLaunchKernelGGL(k_note_grid_vertices, dim3(1), dim3(1), 0, q, (const GridHeader *)c->dHdr, c->dStats + STAT_COUNT, take_stamps(c, q));
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->evGrid, q));
        /* K3 needs the bounding box (query-sort keys), nothing else of the build: when k_compact_records has finalised the
           box K1 kept, or the host has set it, the main stream does not wait for the side stream at all */
        if (boxOnSide) HIPCHK(hipStreamWaitEvent(c->stream, c->evBbox, 0));
        c->gridInFlight = true;
        if (!c->P.wavefront && join_grid(c)) return -1;
    } else {
        if (mark(c, EV_GRID_K0)) return -1;
        if (mark(c, EV_GRID)) return -1;
    }
    return 0;
}

static GridStore grid_of(vcm_ctx *c)
{
    GridStore grid;
    grid.cellStart = c->dCellStart; grid.gx = c->dGx; grid.gy = c->dGy; grid.gz = c->dGz; grid.gb = c->dGb; grid.g1 = c->dG1; grid.g2 = c->dG2; grid.g3 = c->dG3;
    grid.hdr = c->dHdr;
    return grid;
}

static int vcm_trace_camera_impl(vcm_ctx *c)
{   /* vertexcm.hxx:415-545 without the merge (:530-538) in wavefront mode */
    if (!c || !c->inIteration) return fail("vcm_trace_camera", "no iteration in progress");
    if (use_device(c)) return -1;
    if (flush_light_splats(c)) return -1;
    c->cameraTraced = true;
    if (c->lightTraceOnly) return 0;
    int blocks, chunk;
    trace_launch_shape(c->nLocal, &blocks, &chunk);
    /* the camera pass overwrites what K5 of the LAST iteration replays (path colours, masks, per-slot addends): if that K5 runs
       on the splat stream, beside this iteration's K1, it has to be over now */
    if (c->resolveInFlight) HIPCHK(hipStreamWaitEvent(c->stream, c->evResolved, 0));
    if (mark(c, EV_CAMERA_K0)) return -1;
    if (c->renderer) {   /* PathTracer / EyeLight: colour + jittered pixel per path; K5 adds them in path order */
        if (c->renderer == 1)
            LAUNCH_SC(c, k_path_trace, dim3(blocks), dim3(VCM_TRACE_BLOCK), 0, c->stream, c->dScene, c->P, c->dCamOut,
                               c->dRngCam, c->dStats, chunk, take_stamps(c, c->stream), c->vs.count + 8);
        else
            LAUNCH_SC(c, k_eye_light, dim3(2048), dim3(256), 0, c->stream, c->dScene, c->P, c->dCamOut, c->dRngCam,
                               c->dStats, take_stamps(c, c->stream));
        HIPCHK(hipGetLastError());
        if (mark(c, EV_CAMERA_K1)) return -1;
        if (mark(c, EV_CONNECT_K1)) return -1;
        return 0;
    }
    if (c->P.wavefront) {
        /* K3: needs the light-vertex store, NOT the hash grid.  If the grid exists already (single-rank order
           light -> grid -> camera) every camera vertex takes its K4a bucket key and place when it is appended. */
        c->countedInCamera = c->useVM && (c->gridBuilt || c->bboxPreset);
        c->vs.sortHdr = c->countedInCamera ? c->dHdr : NULL;
        c->vs.sortKey = c->countedInCamera ? c->dQueryKey : NULL;
        c->vs.sortArrival = c->countedInCamera ? c->dQueryArrival : NULL;
        c->vs.bucketCount = c->countedInCamera ? c->dQueryCount : NULL;
        if (c->countedInCamera) {
            if (c->prezeroed) HIPCHK(hipStreamWaitEvent(c->stream, c->evZero, 0));
            else if (zero_ranges(c->stream, c->dQueryCount, ((size_t)c->P.nBuckets + 1) * sizeof(int))) return -1;
        }
        LAUNCH_SC_MODE(c, k_camera_trace, 1, dim3(blocks), dim3(VCM_TRACE_BLOCK), 0, c->stream, c->dScene, c->P,
                           c->store, grid_of(c), c->vs, c->dCamOut, c->dCamMask, c->dRngCam, c->dStats, chunk, take_stamps(c, c->stream));
        if (mark(c, EV_CAMERA_K1)) return -1;
        if (c->useVC) {   /* K3b, K3c: dense DI / VC tasks */
            /* K3c reads what K3 appended and the light store, and only K5 reads what it writes: it runs on the splat
               stream, next to K3b and K4, joined before K5: +2 % at 512^2, +6 % at 1024^2 (profiles/archive/r05d_ab_summary.txt);
               at 2048^2, where K4 is issue-bound, it neither gains nor loses with equal stream priorities (r05f) and gained
               1.4 % with the helper streams at low priority (r05l). */
            /* with the histogram done in K3 (countedInCamera) and one DI task per camera vertex (every vertex of a
               VC algorithm has one unless minPathLength cuts it off), K3b also does the scatter of the query sort */
            c->scatteredInDI = c->countedInCamera && c->P.minLen <= 2;
            /* Small frames (round 5): K3b and K3c are grid-stride kernels whose workgroups hold every wave slot of the chip
               until they end, so a scan on another stream gets its workgroups only when K3b's are done and its second launch
               when K3c's are: at 512^2 the two launches took 101 + 169 us (12 us alone), K4 started 200 us after K3b had ended
               and the main stream idled for 16 % of the iteration (profiles/archive/r06t_timeline512.txt).  Up to 1024^2 the scan
               therefore runs IN LINE between K3 and K3b (nothing else is resident at that moment: its 12-25 us are all it
               costs) and K3b scatters the sorted order as it goes.  At 2048^2 the in-line scan shared the memory system with the
               tail of the counting-sort grid build and took 0.9 ms (r06d); beside the radix-sort build it takes 0.1-0.2 ms and
               frees the side stream, whose scan + k_query_scatter (0.42 ms behind the build) were what K4 waited for: in line
               there too, +1.7 % (profiles/archive/r11l_ab_summary.txt); the side-stream form stays for large frames with the counting-sort build
               (SMALLVCM_AMD_GRID_SORT=count). */
            const bool inlineSort = c->scatteredInDI && (c->nLocal <= (1 << 20) || grid_sort_is_radix(c));
            if (c->countedInCamera && c->world == 1 && !inlineSort) {
                /* The scan of the bucket table (16.8 M entries at 2048^2) and the scatter of the sorted order run on the SIDE
                   stream, behind the grid build and next to K3b: in line, between K3 and K3b, the scan took 0.9 ms -- 70 us
                   alone, but it shared the memory system with the tail of the grid build and the light splats while the VALU
                   idled (profiles/archive/r06d_timeline2048.txt).  K4 waits for evSorted.  (Not a stream of its own: HIP maps the
                   streams of a process onto four hardware queues, a fifth stream shared the splat stream's and its work
                   queued behind K3c -- at 512^2 the main stream then idled 0.3 of 1.33 ms, profiles/archive/r06r_timeline512.txt.) */
                HIPCHK(hipEventRecord(c->evSortFork, c->stream));   /* behind K3 */
                HIPCHK(hipStreamWaitEvent(c->side, c->evSortFork, 0));
                const StampArgs none = { { NULL, NULL, NULL, NULL } };
                if (launch_scan_on<int>(c, 3, c->side, c->dQueryCount, c->P.nBuckets, c->dQueryStart, NULL, 1, none)) return -1;
                hipLaunchKernelGGL(k_query_scatter, dim3(aux_blocks(c->nLocal)), dim3(256), 0, c->side, c->vs, (const int *)c->dQueryKey,
                                   (const int *)c->dQueryArrival, (const int *)c->dQueryStart, c->dSortedVertex);
                HIPCHK(hipEventRecord(c->evSorted, c->side));
                c->sortInFlight = true;
                c->scatteredInDI = false;
            }
            if (c->scatteredInDI && launch_scan<int>(c, c->dQueryCount, c->P.nBuckets, c->dQueryStart, NULL, 1)) return -1;
            {
                /* behind K3 -- and behind the in-line scan of the bucket table: launched at the same moment as K3c, the scan's
                   8192 workgroups shared the chip with K3c's resident ones and took 93 + 127 us instead of ~25 at 1024^2, on the
                   critical path (profiles/archive/r07f_timeline1024.txt) */
                HIPCHK(hipEventRecord(c->evSplatFork, c->stream));
                HIPCHK(hipStreamWaitEvent(c->splat, c->evSplatFork, 0));
                LAUNCH_SC(c, k_connect_vc, dim3(task_blocks(c->nLocal)), dim3(VCM_TASK_BLOCK), 0, c->splat, c->dScene, c->P, c->vs,
                                   c->store, c->dStats);
                HIPCHK(hipEventRecord(c->evSplatDone, c->splat));   /* also behind the light splats: same stream */
                c->splatInFlight = true;
            }
            LAUNCH_SC(c, k_connect_di, dim3(task_blocks(c->nLocal)), dim3(VCM_TASK_BLOCK), 0, c->stream, c->dScene, c->P, c->vs,
                               c->dStats, c->scatteredInDI ? (const int *)c->dQueryStart : (const int *)NULL,
                               c->scatteredInDI ? c->dSortedVertex : (int *)NULL, take_stamps(c, c->stream));
        }
        if (mark(c, EV_CONNECT_K1)) return -1;
    } else {
        if (c->useVM && !c->gridBuilt) return fail("vcm_trace_camera", "strict mode merges inside the camera pass: call vcm_build_grid first");
        LAUNCH_SC_MODE(c, k_camera_trace, 0, dim3(blocks), dim3(VCM_TRACE_BLOCK), 0, c->stream, c->dScene, c->P,
                           c->store, grid_of(c), c->vs, c->dCamOut, c->dCamMask, c->dRngCam, c->dStats, chunk, take_stamps(c, c->stream));
        if (mark(c, EV_CAMERA_K1)) return -1;
        if (mark(c, EV_CONNECT_K1)) return -1;
    }
    HIPCHK(hipGetLastError());
    if (c->world > 1 && flush_stamps(c, c->stream)) return -1;   /* the exchange may sit between this phase and the merge */
    return 0;
}

/* ---- the merge sharded by SPACE (round 6 prototype; include/smallvcm_amd.h, kernels: vcm_kernels.h) ---- */
#define VCM_SPACE_BLOCKS 512
static int space_scratch(vcm_ctx *c)
{
    if (c->dSpaceMatrix) return 0;
    HIPCHK(hipMalloc((void **)&c->dSpaceMatrix, sizeof(int) * VCM_SPACE_MAX_SLABS * VCM_SPACE_BLOCKS));
    HIPCHK(hipMalloc((void **)&c->dSpaceScanned, sizeof(int) * (VCM_SPACE_MAX_SLABS * VCM_SPACE_BLOCKS + 1)));
    HIPCHK(hipMalloc((void **)&c->dSpaceTotals, sizeof(int) * VCM_SPACE_MAX_SLABS));
    HIPCHK(hipMalloc((void **)&c->dSpaceHist, sizeof(int) * 256));
    return 0;
}
static int space_ready(vcm_ctx *c, const char *who)
{
    if (!c || !c->inIteration) return fail(who, "no iteration in progress");
    if (c->world <= 1) return fail(who, "the context is not sharded");
    if (!c->useVM || !c->P.wavefront) return fail(who, "needs a merging algorithm in wavefront mode");
    if (use_device(c)) return -1;
    return space_scratch(c);
}
static int vcm_space_histogram_impl(vcm_ctx *c, int axis, float *lo, float *binWidth, int *hist256)
{
    if (space_ready(c, "vcm_space_histogram")) return -1;
    if (axis < 0 || axis > 2 || !lo || !binWidth || !hist256) return fail("vcm_space_histogram", "bad argument");
    if (ensure_records(c)) return -1;
    const float R = c->scene->sceneRadius, L = c->scene->sceneCenter[axis] - R, w = 2.f * R / 256.f;
    HIPCHK(hipMemsetAsync(c->dSpaceHist, 0, sizeof(int) * 256, c->stream));
    hipLaunchKernelGGL(k_space_hist, dim3(VCM_SPACE_BLOCKS), dim3(256), 0, c->stream, (const float *)c->dRecordsLocal, (const int *)c->dLocalTotal, axis, L, 1.f / w, c->dSpaceHist);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(hist256, c->dSpaceHist, sizeof(int) * 256, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *lo = L; *binWidth = w;
    return 0;
}
static int vcm_space_set_slabs_impl(vcm_ctx *c, int axis, const float *splits, int nSlabs)
{
    if (space_ready(c, "vcm_space_set_slabs")) return -1;
    if (!c->bboxPreset) return fail("vcm_space_set_slabs", "call vcm_set_grid_bbox first: slabs are ranges of cells");
    if (axis < 0 || axis > 2 || !splits || nSlabs != c->world || nSlabs > VCM_SPACE_MAX_SLABS) return fail("vcm_space_set_slabs", "one slab per shard, at most 64");
    SpaceSlabs &sl = c->space;
    sl.S = nSlabs; sl.axis = axis;
    sl.X[0] = -(1 << 30); sl.X[nSlabs] = 1 << 30;
    for (int s2 = 1; s2 < nSlabs; s2++) {
        const float f = floorf(c->P.invCellSize * (splits[s2] - c->presetMin[axis]));
        int x = f < -1e9f ? -(1 << 30) : (f > 1e9f ? (1 << 30) : (int)f);
        if (x < sl.X[s2 - 1]) x = sl.X[s2 - 1];
        sl.X[s2] = x;
    }
    c->spaceSet = true;
    return 0;
}
/* the stable partition of `n` elements (device count) by destination slab; the counts per destination come back to the host */
static int space_partition(vcm_ctx *c, int kind, const char *who, const float *src, const int *nPtr, void *dstDev, long long stride, long long *counts,
                           int *whereDest, int *wherePos)
{
    if (!c->spaceSet) return fail(who, "call vcm_space_set_slabs first");
    if (!dstDev || !counts || stride < 1) return fail(who, "bad argument");
    const int S = c->space.S, V = VCM_SPACE_BLOCKS;
    HIPCHK(hipMemsetAsync(c->dSpaceTotals, 0, sizeof(int) * VCM_SPACE_MAX_SLABS, c->stream));
    if (kind == 0) hipLaunchKernelGGL(k_space_count<0>, dim3(V), dim3(256), 0, c->stream, c->P, (const GridHeader *)c->dHdr, c->space, src, nPtr, c->dSpaceMatrix, c->dSpaceTotals);
    else hipLaunchKernelGGL(k_space_count<1>, dim3(V), dim3(256), 0, c->stream, c->P, (const GridHeader *)c->dHdr, c->space, src, nPtr, c->dSpaceMatrix, c->dSpaceTotals);
    HIPCHK(hipGetLastError());
    if (launch_scan<int>(c, c->dSpaceMatrix, S * V, c->dSpaceScanned, NULL, 0)) return -1;
    if (kind == 0) hipLaunchKernelGGL(k_space_scatter<0>, dim3(V), dim3(256), 0, c->stream, c->P, (const GridHeader *)c->dHdr, c->space, src, nPtr, (const int *)c->dSpaceScanned,
                                      (float *)dstDev, stride, whereDest, wherePos);
    else hipLaunchKernelGGL(k_space_scatter<1>, dim3(V), dim3(256), 0, c->stream, c->P, (const GridHeader *)c->dHdr, c->space, src, nPtr, (const int *)c->dSpaceScanned,
                            (float *)dstDev, stride, whereDest, wherePos);
    HIPCHK(hipGetLastError());
    int h[VCM_SPACE_MAX_SLABS];
    HIPCHK(hipMemcpyAsync(h, c->dSpaceTotals, sizeof(int) * (size_t)S, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int d = 0; d < S; d++) {
        counts[d] = h[d];
        if (h[d] > stride) return fail(who, "a destination's share exceeds the stride of the buffer");
    }
    return 0;
}
static int vcm_space_partition_light_impl(vcm_ctx *c, void *dstDev, long long strideRecords, long long *counts)
{
    if (space_ready(c, "vcm_space_partition_light")) return -1;
    if (ensure_records(c)) return -1;
    return space_partition(c, 0, "vcm_space_partition_light", (const float *)c->dRecordsLocal, (const int *)c->dLocalTotal, dstDev, strideRecords, counts, NULL, NULL);
}
static int vcm_space_partition_queries_impl(vcm_ctx *c, void *dstDev, long long strideQueries, long long *counts)
{
    if (space_ready(c, "vcm_space_partition_queries")) return -1;
    if (!c->cameraTraced) return fail("vcm_space_partition_queries", "call vcm_trace_camera first");
    if (c->whereCap < c->vs.qcap) {
        if (c->dWhereDest) { (void)hipFree(c->dWhereDest); (void)hipFree(c->dWherePos); c->dWhereDest = c->dWherePos = NULL; }
        HIPCHK(hipMalloc((void **)&c->dWhereDest, sizeof(int) * c->vs.qcap));
        HIPCHK(hipMalloc((void **)&c->dWherePos, sizeof(int) * c->vs.qcap));
        c->whereCap = c->vs.qcap;
    }
    return space_partition(c, 1, "vcm_space_partition_queries", (const float *)c->vs.q, (const int *)c->vs.count, dstDev, strideQueries, counts, c->dWhereDest, c->dWherePos);
}
/* K4a + K4 over the queries the other ranks sent, against this rank's grid; their terms in the queries' order */
static int vcm_space_merge_impl(vcm_ctx *c, const void *queriesDev, const long long *counts, int nSeg, long long stride, void *resultsDev)
{
    if (space_ready(c, "vcm_space_merge")) return -1;
    if (!c->gridBuilt) return fail("vcm_space_merge", "call vcm_build_grid first");
    if (!queriesDev || !counts || !resultsDev || nSeg != c->world || stride < 1) return fail("vcm_space_merge", "bad argument");
    long long M = 0;
    for (int s2 = 0; s2 < nSeg; s2++) { if (counts[s2] < 0 || counts[s2] > stride) return fail("vcm_space_merge", "a count exceeds the stride"); M += counts[s2]; }
    const size_t need = (size_t)(M > 0 ? M : 1);
    if (c->fCap < need) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->fQ) { (void)hipFree(c->fQ); (void)hipFree(c->fRes); (void)hipFree(c->fKey); (void)hipFree(c->fArrival); (void)hipFree(c->fSorted); }
        c->fCap = need + need / 8 + 1024;
        HIPCHK(hipMalloc((void **)&c->fQ, sizeof(F4) * 4 * c->fCap));
        HIPCHK(hipMalloc((void **)&c->fRes, sizeof(F4) * c->fCap));
        HIPCHK(hipMalloc((void **)&c->fKey, sizeof(int) * c->fCap));
        HIPCHK(hipMalloc((void **)&c->fArrival, sizeof(int) * c->fCap));
        HIPCHK(hipMalloc((void **)&c->fSorted, sizeof(int) * c->fCap));
    }
    if (!c->fCount) { HIPCHK(hipMalloc((void **)&c->fCount, sizeof(int) * 32)); HIPCHK(hipMemsetAsync(c->fCount, 0, sizeof(int) * 32, c->stream)); }
    long long at = 0;
    for (int s2 = 0; s2 < nSeg; s2++) {   /* the segments back to back: positions = the order the terms go back in */
        if (counts[s2] > 0)
            HIPCHK(hipMemcpyAsync((char *)c->fQ + (size_t)at * 64, (const char *)queriesDev + (size_t)s2 * (size_t)stride * 64, (size_t)counts[s2] * 64, hipMemcpyDeviceToDevice, c->stream));
        at += counts[s2];
    }
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, c->stream, c->fCount, (int)M);
    if (join_grid(c)) return -1;
    if (M > 0) {
        IterParams P2 = c->P;
        P2.foreignOut = 1;
        VertexStore fvs = c->vs;
        fvs.q = c->fQ; fvs.q4 = NULL; fvs.qcap = c->fCap; fvs.count = c->fCount; fvs.mergeOut = c->fRes;
        fvs.sortHdr = NULL; fvs.sortKey = NULL; fvs.sortArrival = NULL; fvs.bucketCount = NULL;
        const int nb = c->P.nBuckets;
        if (c->prezeroed) HIPCHK(hipStreamWaitEvent(c->stream, c->evZero, 0));
        if (zero_ranges(c->stream, c->dQueryCount, ((size_t)nb + 1) * sizeof(int))) return -1;
        const StampArgs none = { { NULL, NULL, NULL, NULL } };
        const int nLoc = (int)(M < (1ll << 30) ? M : (1ll << 30));
        hipLaunchKernelGGL(k_query_count, dim3(aux_blocks(nLoc)), dim3(256), 0, c->stream, P2, fvs, (const GridHeader *)c->dHdr, c->fKey, c->fArrival, c->dQueryCount, none);
        HIPCHK(hipGetLastError());
        if (launch_scan<int>(c, c->dQueryCount, nb, c->dQueryStart, NULL, 1)) return -1;
        hipLaunchKernelGGL(k_query_scatter, dim3(aux_blocks(nLoc)), dim3(256), 0, c->stream, fvs, (const int *)c->fKey, (const int *)c->fArrival, (const int *)c->dQueryStart, c->fSorted);
        const int blocks = merge_blocks(nLoc, c->N);
        int kind = c->mergeKind;
        if (kind == VCM_MERGE_PAIRS && (int)c->scene->materials.size() > VCM_PAIR_MATERIALS) kind = VCM_MERGE_WALK;
        const int chunk = shape_knob("merge_chunk") ? shape_knob("merge_chunk") : 16;
        if (kind == VCM_MERGE_PAIRS) {
            if (c->intPhong) hipLaunchKernelGGL(k_merge_pairs<true>, dim3(blocks), dim3(VCM_MERGE_BLOCK), 0, c->stream, c->dScene, P2, grid_of(c), fvs, (const int *)c->fSorted, (const int *)(c->dQueryStart + nb), c->dStats, chunk, none);
            else hipLaunchKernelGGL(k_merge_pairs<false>, dim3(blocks), dim3(VCM_MERGE_BLOCK), 0, c->stream, c->dScene, P2, grid_of(c), fvs, (const int *)c->fSorted, (const int *)(c->dQueryStart + nb), c->dStats, chunk, none);
        } else {
            if (c->intPhong) hipLaunchKernelGGL(k_merge_walk<true>, dim3(blocks), dim3(VCM_MERGE_BLOCK), 0, c->stream, c->dScene, P2, grid_of(c), fvs, (const int *)c->fSorted, (const int *)(c->dQueryStart + nb), c->dStats, chunk, none);
            else hipLaunchKernelGGL(k_merge_walk<false>, dim3(blocks), dim3(VCM_MERGE_BLOCK), 0, c->stream, c->dScene, P2, grid_of(c), fvs, (const int *)c->fSorted, (const int *)(c->dQueryStart + nb), c->dStats, chunk, none);
        }
        HIPCHK(hipGetLastError());
    }
    at = 0;
    for (int s2 = 0; s2 < nSeg; s2++) {
        if (counts[s2] > 0)
            HIPCHK(hipMemcpyAsync((char *)resultsDev + (size_t)s2 * (size_t)stride * 16, (const char *)c->fRes + (size_t)at * 16, (size_t)counts[s2] * 16, hipMemcpyDeviceToDevice, c->stream));
        at += counts[s2];
    }
    return 0;
}
static int vcm_space_import_results_impl(vcm_ctx *c, const void *resultsDev, long long stride)
{
    if (space_ready(c, "vcm_space_import_results")) return -1;
    if (!c->dWhereDest || !resultsDev) return fail("vcm_space_import_results", "call vcm_space_partition_queries first");
    hipLaunchKernelGGL(k_space_results, dim3(aux_blocks(c->nLocal)), dim3(256), 0, c->stream, c->P, c->vs, (const int *)c->dWhereDest, (const int *)c->dWherePos, (const F4 *)resultsDev, stride);
    HIPCHK(hipGetLastError());
    c->mergeImported = true;
    return 0;
}
extern "C" int vcm_space_histogram(vcm_ctx *c, int axis, float *lo, float *binWidth, int *hist256) { g_hipFailed = false; return abort_iteration(c, vcm_space_histogram_impl(c, axis, lo, binWidth, hist256)); }
extern "C" int vcm_space_set_slabs(vcm_ctx *c, int axis, const float *splits, int nSlabs) { g_hipFailed = false; return abort_iteration(c, vcm_space_set_slabs_impl(c, axis, splits, nSlabs)); }
extern "C" int vcm_space_partition_light(vcm_ctx *c, void *dstDev, long long strideRecords, long long *counts) { g_hipFailed = false; return abort_iteration(c, vcm_space_partition_light_impl(c, dstDev, strideRecords, counts)); }
extern "C" int vcm_space_partition_queries(vcm_ctx *c, void *dstDev, long long strideQueries, long long *counts) { g_hipFailed = false; return abort_iteration(c, vcm_space_partition_queries_impl(c, dstDev, strideQueries, counts)); }
extern "C" int vcm_space_merge(vcm_ctx *c, const void *queriesDev, const long long *counts, int nSeg, long long stride, void *resultsDev) { g_hipFailed = false; return abort_iteration(c, vcm_space_merge_impl(c, queriesDev, counts, nSeg, stride, resultsDev)); }
extern "C" int vcm_space_import_results(vcm_ctx *c, const void *resultsDev, long long stride) { g_hipFailed = false; return abort_iteration(c, vcm_space_import_results_impl(c, resultsDev, stride)); }

static int vcm_merge_impl(vcm_ctx *c)
{   /* vertexcm.hxx:530-538 for all camera vertices, then :544 */
    if (!c || !c->inIteration) return fail("vcm_merge", "no iteration in progress");
    if (!c->cameraTraced) return fail("vcm_merge", "call vcm_trace_camera first");
    if (use_device(c)) return -1;
    if (!c->lightTraceOnly) {
        bool mergeAside = false;
        if (mark(c, EV_MERGE_K0)) return -1;
        if (c->P.wavefront && c->useVM && c->mergeImported) {
            /* merge sharded by space: the slabs' owners evaluated this rank's queries, vcm_space_import_results wrote their terms */
            if (mark(c, EV_SORT_K1)) return -1;
        } else if (c->P.wavefront && c->useVM) {
            if (!c->gridBuilt) return fail("vcm_merge", "call vcm_build_grid first");
            /* K4a: counting sort of the camera vertices by the Morton code of their base cell */
            const int nb = c->P.nBuckets;
            if (!c->countedInCamera) {
                /* (camera pass before the grid build on a single rank: the side stream's zeroing of this very table, started
                   next to K1, must be over before it is zeroed and counted into again here -- ADVICE r4) */
                if (c->prezeroed) HIPCHK(hipStreamWaitEvent(c->stream, c->evZero, 0));
                if (zero_ranges(c->stream, c->dQueryCount, ((size_t)nb + 1) * sizeof(int))) return -1;
                hipLaunchKernelGGL(k_query_count, dim3(aux_blocks(c->nLocal)), dim3(256), 0, c->stream, c->P, c->vs,
                                   (const GridHeader *)c->dHdr, c->dQueryKey, c->dQueryArrival, c->dQueryCount, take_stamps(c, c->stream));
            }
            if (c->sortInFlight) {   /* scan + scatter ran next to K3b */
                HIPCHK(hipStreamWaitEvent(c->stream, c->evSorted, 0));
                c->sortInFlight = false;
            } else if (!c->scatteredInDI) {
                if (launch_scan<int>(c, c->dQueryCount, nb, c->dQueryStart, NULL, 1)) return -1;
                hipLaunchKernelGGL(k_query_scatter, dim3(aux_blocks(c->nLocal)), dim3(256), 0, c->stream, c->vs, (const int *)c->dQueryKey,
                                   (const int *)c->dQueryArrival, (const int *)c->dQueryStart, c->dSortedVertex);
            }
            if (mark(c, EV_SORT_K1)) return -1;
            /* K4: needs the grid; the query sort above only needed its bounding box, so in the sharded order
               (camera pass before the grid build) it ran next to the build.  If the stream has to wait for the side
               stream, the pending marks are written first (they must not absorb the wait). */
            if (c->gridInFlight && hipEventQuery(c->evGrid) != hipSuccess) { (void)hipGetLastError(); if (flush_stamps(c, c->stream)) return -1; }
            if (join_grid(c)) return -1;
            /* Round 5: K4 on the SIDE stream, behind the grid build and the query sort it needs anyway.  It is the iteration's
               longest kernel, bound by its cache-line requests (VALU 0.42); the main stream is free when K3b has ended, so the
               NEXT iteration's K1 (VALU-bound) runs beside it and waits for it only when compaction is about to rewrite the grid
               header (vcm_trace_light); the next grid build and table zeroing queue behind it on the side stream, which they have
               to.  K5 follows K4 through an event (it runs on the splat stream, see below).  Same conditions as for K5.  (A stream of
               its own was the first attempt: HIP gave it the splat stream's hardware queue and K4 waited for K3c,
               profiles/archive/r09b_timeline2048.txt.) */
            /* Measured (profiles/archive/r09c_ab_*.txt, pairs of 40-400 iteration runs, bit-exact): 1024^2 scene 3 1206 -> 1344 Mpaths/s
               (+11 %), 1024^2 scene 1 962 -> 1049 (+9 %), 2048^2 BPM 1240 -> 1264 (+2 %), 512^2 equal -- and 2048^2 VCM 1012 -> 989
               (-2 %): there K4 already shares the chip with K3c, and K1 beside both costs more than it hides.  So: by default up to
               1024^2, and at any size for the algorithms without vertex connection (re-measured with k_merge_pairs in round 6: forced at
               2048^2 VCM 1129 -> 1057, profiles/r13l). */
            mergeAside = (c->nLocal <= (1 << 20) || !c->useVC) && c->world == 1 && !c->strictOrder;   /* (the conditions of `aside` below: K5 follows K4 through evMergeDone only there) */
            hipStream_t ks = c->stream;
            if (mergeAside) {
                if (flush_stamps(c, c->stream)) return -1;   /* the marks so far belong to what ran on the main stream */
                HIPCHK(hipEventRecord(c->evPreMerge, c->stream));
                HIPCHK(hipStreamWaitEvent(c->side, c->evPreMerge, 0));
                ks = c->side;
            }
            static int mergeChunk = 0;
            if (!mergeChunk) mergeChunk = shape_knob("merge_chunk") ? shape_knob("merge_chunk") : 16;
            /* Two kernels, same bits (vcm_set_merge_kernel).  k_merge_pairs (default since round 6): every lane walks its own non-empty
               runs back to back, the accepted (query, photon) pairs of a wave are evaluated 64 at a time -- 2.73 ms against 3.23 for
               k_merge_walk, whose lanes evaluate their own queues at 38 % occupancy (profiles/r13m_kstats_2048_vcm.txt). */
            int mergeStaged = c->mergeKind;
            /* k_merge_pairs keeps a material table of VCM_PAIR_MATERIALS rows in LDS */
            if (mergeStaged == VCM_MERGE_PAIRS && (int)c->scene->materials.size() > VCM_PAIR_MATERIALS) mergeStaged = VCM_MERGE_WALK;
            if (mergeStaged == VCM_MERGE_PAIRS) {
                if (c->intPhong)
                    hipLaunchKernelGGL(k_merge_pairs<true>, dim3(merge_blocks(c->nLocal, c->N)), dim3(VCM_MERGE_BLOCK), 0, ks, c->dScene, c->P, grid_of(c),
                                       c->vs, (const int *)c->dSortedVertex, (const int *)(c->dQueryStart + nb), c->dStats, mergeChunk, take_stamps(c, c->stream));
                else
                    hipLaunchKernelGGL(k_merge_pairs<false>, dim3(merge_blocks(c->nLocal, c->N)), dim3(VCM_MERGE_BLOCK), 0, ks, c->dScene, c->P, grid_of(c),
                                       c->vs, (const int *)c->dSortedVertex, (const int *)(c->dQueryStart + nb), c->dStats, mergeChunk, take_stamps(c, c->stream));
            }
            else {
                if (c->intPhong)
                    hipLaunchKernelGGL(k_merge_walk<true>, dim3(merge_blocks(c->nLocal, c->N)), dim3(VCM_MERGE_BLOCK), 0, ks, c->dScene, c->P, grid_of(c),
                                       c->vs, (const int *)c->dSortedVertex, (const int *)(c->dQueryStart + nb), c->dStats, mergeChunk, take_stamps(c, c->stream));
                else
                    hipLaunchKernelGGL(k_merge_walk<false>, dim3(merge_blocks(c->nLocal, c->N)), dim3(VCM_MERGE_BLOCK), 0, ks, c->dScene, c->P, grid_of(c),
                                       c->vs, (const int *)c->dSortedVertex, (const int *)(c->dQueryStart + nb), c->dStats, mergeChunk, take_stamps(c, c->stream));
            }
        } else {
            if (mark(c, EV_SORT_K1)) return -1;
        }
        if (mergeAside) {
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c->evMergeDone, c->side));
            c->mergeInFlight = true;
        }
        if (mark(c, EV_MERGE_K1)) return -1;
        /* K5: the first kernel since the light splats that touches the framebuffer.
           Round 5: on the SPLAT stream, behind K1c / K1d / K3c (what join_splats enforced) and behind K4 (an event) -- the main
           stream is free the moment K4 ends, so the NEXT iteration's K1 (VALU-bound, touches neither the framebuffer nor
           anything K5 reads) runs beside K5 (bound by its fetches) instead of behind it.  The next camera pass waits for
           evResolved, the next light splats follow K5 on its stream, every reader of the framebuffer joins the splat stream as
           before (sharded and strict-order contexts keep it in line). */
        const bool aside = c->world == 1 && !c->strictOrder;
        if (aside) {
            if (!mergeAside) HIPCHK(hipEventRecord(c->evMergeDone, c->stream));   /* (else: recorded behind K4 on its own stream) */
            HIPCHK(hipEventRecord(c->evSplatWork, c->splat));    /* K1c, K1d, K3c of this iteration: the light store is dead behind it */
            HIPCHK(hipStreamWaitEvent(c->splat, c->evMergeDone, 0));
            hipLaunchKernelGGL(k_resolve, dim3(resolve_blocks(c->nLocal, c->useVC)), dim3(256), 0, c->splat, c->P, (const F4 *)c->dCamOut,
                               (const uint32_t *)c->dCamMask, c->vs, c->dFb, take_stamps(c, c->stream));
            HIPCHK(hipGetLastError());
            if (mark(c, EV_CAMERA)) return -1;
            if (c->nPend[0] > 0) {   /* the end mark: a one-lane launch behind K5, on its stream */
                hipLaunchKernelGGL(k_stamp_many, dim3(1), dim3(1), 0, c->splat, take_stamps(c, c->stream));
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipEventRecord(c->evResolved, c->splat));
            HIPCHK(hipEventRecord(c->evSplatDone, c->splat));
            /* the next iteration's K1 overwrites the light store: K1c / K3c of THIS iteration (splat stream) must be over --
               they are, long before K4 ends, but the main stream has to know */
            HIPCHK(hipStreamWaitEvent(c->stream, c->evSplatWork, 0));
            c->splatInFlight = true;
            c->resolveInFlight = true;
            c->merged = true;
            return 0;
        }
        if (join_splats(c)) return -1;
        hipLaunchKernelGGL(k_resolve, dim3(resolve_blocks(c->nLocal, false)), dim3(256), 0, c->stream, c->P, (const F4 *)c->dCamOut,
                           (const uint32_t *)c->dCamMask, c->vs, c->dFb, take_stamps(c, c->stream));
        HIPCHK(hipGetLastError());
    }
    if (join_splats(c)) return -1;               /* light tracing alone: nothing else waited for them */
    if (mark(c, EV_CAMERA)) return -1;
    if (flush_stamps(c, c->stream)) return -1;   /* nothing of this iteration follows: the end mark gets its own (one-lane) launch */
    c->merged = true;
    return 0;
}

/* a HIP failure inside a phase ends the iteration and returns the arena */
int vcm_begin_iteration(vcm_ctx *c, int iteration, unsigned minLen, unsigned maxLen)
{
    g_hipFailed = false;
    if (c && c->inIteration) return fail("vcm_begin_iteration", "previous iteration not ended");   /* it stays open */
    return abort_iteration(c, vcm_begin_iteration_impl(c, iteration, minLen, maxLen));
}
/* 1 if an iteration with this maxPathLength runs in wavefront mode (camera pass independent of the grid),
 * 0 if everything is evaluated inside the paths (strict order requested, or maxPathLength > 31) */
int vcm_is_wavefront(vcm_ctx *c, unsigned maxLen)
{
    if (!c) return 0;
    return (!c->strictOrder && !c->lightTraceOnly && !c->renderer && maxLen <= 31) ? 1 : 0;
}
int vcm_import_light_records(vcm_ctx *c, const void *devPtr, const long long *counts, int nSeg, long long strideRecords)
{
    g_hipFailed = false;
    return abort_iteration(c, vcm_import_light_records_impl(c, devPtr, counts, nSeg, strideRecords));
}
int vcm_local_light_bbox(vcm_ctx *c, float *min3, float *max3, long long *count)
{
    g_hipFailed = false;
    return abort_iteration(c, vcm_local_light_bbox_impl(c, min3, max3, count));
}
int vcm_set_grid_bbox(vcm_ctx *c, const float *min3, const float *max3)
{
    g_hipFailed = false;
    return abort_iteration(c, vcm_set_grid_bbox_impl(c, min3, max3));
}
/* allocate now what the first iteration would allocate (context buffers, the device's scratch arena) */
int vcm_reserve(vcm_ctx *c, unsigned maxLen)
{
    if (!c) return fail("vcm_reserve", "ctx is NULL");
    if (c->inIteration) return fail("vcm_reserve", "iteration in progress");
    if (maxLen > 255) return fail("vcm_reserve", "maxPathLength > 255 unsupported");
    g_hipFailed = false;
    if (ensure_device(c)) return -1;
    const int S = c->renderer ? 1 : ((maxLen >= 2) ? (int)maxLen - 1 : 1);
    const int L = c->renderer ? 1 : ((maxLen >= 1) ? (int)maxLen : 1);
    const int rc = arena_acquire(c, S, L);
    if (c->holdsArena) arena_release(c, false);
    return rc;
}
int vcm_trace_light(vcm_ctx *c) { g_hipFailed = false; return abort_iteration(c, vcm_trace_light_impl(c)); }
int vcm_build_grid(vcm_ctx *c) { g_hipFailed = false; return abort_iteration(c, vcm_build_grid_impl(c)); }
int vcm_trace_camera(vcm_ctx *c) { g_hipFailed = false; return abort_iteration(c, vcm_trace_camera_impl(c)); }
int vcm_merge(vcm_ctx *c) { g_hipFailed = false; return abort_iteration(c, vcm_merge_impl(c)); }

static int vcm_end_iteration_impl(vcm_ctx *c)
{
    if (use_device(c)) return -1;
    if (join_grid(c)) return -1;
    if (!c->resolveInFlight && join_splats(c)) return -1;   /* (a merge-free algorithm never waited; K5 aside: the next camera pass waits) */
    if (flush_stamps(c, c->stream) || (c->deviceReady && flush_stamps(c, c->side))) return -1;
    c->iterations++;   /* :547 */
    c->inIteration = false;
    c->evValid = true;
    arena_release(c, true);
    return 0;
}
int vcm_end_iteration(vcm_ctx *c)
{
    if (!c || !c->inIteration) return fail("vcm_end_iteration", "no iteration in progress");
    /* a mis-ordered call is refused and the iteration stays open (the host can still call vcm_merge): tearing it down
       here would leave the light splats in the framebuffer without counting the iteration */
    if (!c->merged) return fail("vcm_end_iteration", "vcm_merge has not run");
    g_hipFailed = false;
    return abort_iteration(c, vcm_end_iteration_impl(c));   /* a HIP failure ends the iteration like every other phase call */
}

int vcm_run_iteration(vcm_ctx *c, int iteration, unsigned minLen, unsigned maxLen)
{
    if (c && c->world > 1)
        return fail("vcm_run_iteration", "sharded context: the host must exchange light records between "
                                         "vcm_trace_light and vcm_build_grid (see smallvcm_amd/renderer.py)");
    if (vcm_begin_iteration(c, iteration, minLen, maxLen)) return -1;
    if (vcm_trace_light(c)) return -1;
    if (vcm_build_grid(c)) return -1;
    if (vcm_trace_camera(c)) return -1;
    if (vcm_merge(c)) return -1;
    return vcm_end_iteration(c);
}

int vcm_synchronize(vcm_ctx *c)
{
    if (!c) return fail("vcm_synchronize", "ctx is NULL");
    if (!c->deviceReady) return 0;
    if (use_device(c)) return -1;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->resolveInFlight) HIPCHK(hipStreamSynchronize(c->splat));   /* the last iteration's K5 */
    return 0;
}

int vcm_read_framebuffer(vcm_ctx *c, float *rgbHost)
{
    if (!c || !rgbHost) return fail("vcm_read_framebuffer", "NULL argument");
    if (!c->deviceReady) { memset(rgbHost, 0, (size_t)c->N * 3 * sizeof(float)); return 0; }
    if (use_device(c)) return -1;
    if (join_splats(c)) return -1;   /* K1c / K1d of an open iteration add to dFb on the splat stream */
    HIPCHK(hipMemcpyAsync(rgbHost, c->dFb, (size_t)c->N * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int vcm_pin_host_memory(void *hostPtr, unsigned long long bytes)
{
    if (!hostPtr || !bytes) return fail("vcm_pin_host_memory", "NULL argument");
    if (vcm_device_count() <= 0) return fail("vcm_pin_host_memory", "no HIP device available");
    HIPCHK(hipHostRegister(hostPtr, (size_t)bytes, hipHostRegisterPortable));
    return 0;
}
int vcm_unpin_host_memory(void *hostPtr)
{
    if (!hostPtr) return fail("vcm_unpin_host_memory", "NULL argument");
    HIPCHK(hipHostUnregister(hostPtr));
    return 0;
}

int vcm_read_image(vcm_ctx *c, int format, float scale, float gamma, unsigned char *outHost)
{
    if (!c || !outHost) return fail("vcm_read_image", "NULL argument");
    if (format != VCM_IMAGE_BGR8 && format != VCM_IMAGE_RGBE) return fail("vcm_read_image", "unknown format");
    if (!(gamma > 0.f)) return fail("vcm_read_image", "gamma must be positive");
    if (ensure_device(c)) return -1;
    const size_t bytes = (size_t)c->N * (format == VCM_IMAGE_BGR8 ? 3 : 4);
    unsigned char *d = NULL;
    if (dalloc(&d, bytes)) return -1;
    if (join_splats(c)) { (void)hipFree(d); return -1; }
    hipLaunchKernelGGL(k_encode_image, dim3(1024), dim3(256), 0, c->stream, (const float *)c->dFb, c->resX, c->resY, format,
                       scale, 1.f / gamma, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(outHost, d, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) { g_hipFailed = true; return fail("vcm_read_image", hipGetErrorString(e)); }
    return 0;
}

int vcm_framebuffer_device(vcm_ctx *c, void **devPtr)
{
    if (!c || !devPtr) return fail("vcm_framebuffer_device", "NULL argument");
    if (ensure_device(c)) return -1;
    if (join_splats(c)) return -1;   /* whoever uses the pointer orders against the context's stream */
    *devPtr = c->dFb;
    return 0;
}

int vcm_clear_framebuffer(vcm_ctx *c)
{
    if (!c) return fail("vcm_clear_framebuffer", "ctx is NULL");
    if (!c->deviceReady) return 0;
    if (use_device(c)) return -1;
    if (join_splats(c)) return -1;
    HIPCHK(hipMemsetAsync(c->dFb, 0, (size_t)c->N * 3 * sizeof(float), c->stream));
    return 0;
}

int vcm_iterations(vcm_ctx *c) { return c ? c->iterations : 0; }

/* counters and phase times of the iteration that ended `ago` iterations before the last one (0 = the last) */
int vcm_get_stats_at(vcm_ctx *c, int ago, vcm_stats *out)
{
    if (!c || !out) return fail("vcm_get_stats", "NULL argument");
    memset(out, 0, sizeof(*out));
    if (!c->deviceReady) return 0;
    if (ago < 0 || ago >= VCM_STAMP_RING) return fail("vcm_get_stats_at", "only the last 64 iterations are kept");
    if (use_device(c)) return -1;
    /* inside an iteration the current slot is the newest; otherwise the last completed one */
    const int newest = c->inIteration ? c->iterations : c->iterations - 1;
    if (newest < 0) return 0;
    if (newest - ago < 0) return fail("vcm_get_stats_at", "no such iteration");
    const int slot = (newest - ago) % VCM_STAMP_RING;
    if (c->resolveInFlight) HIPCHK(hipStreamSynchronize(c->splat));   /* the end mark of the last iteration is written behind its K5 */
    unsigned long long h[VCM_STAT_SLOTS];
    HIPCHK(hipMemcpyAsync(h, c->dStatsRing + (size_t)slot * VCM_STAT_SLOTS, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    out->lightVertices = (long long)h[STAT_STORED];
    out->gridVertices = (long long)h[STAT_COUNT];
    out->lightRays = (long long)h[STAT_LIGHT_RAYS];
    out->cameraRays = (long long)h[STAT_CAMERA_RAYS];
    out->shadowRays = (long long)h[STAT_SHADOW_RAYS];
    out->mergeQueries = (long long)h[STAT_MERGE_QUERIES];
    out->mergeCandidates = (long long)h[STAT_MERGE_CANDIDATES];
    out->mergeAccepted = (long long)h[STAT_MERGE_ACCEPTED];
    out->connections = (long long)h[STAT_CONNECTIONS];
    out->lightSplats = (long long)h[STAT_LIGHT_SPLATS];
    out->radius = c->radiusRing[slot];
    if (!c->inIteration) {
        static int useEvents = -1;
        if (useEvents < 0) { const char *e = getenv("SMALLVCM_AMD_TIMING"); useEvents = (e && !strcmp(e, "events")) ? 1 : 0; }
        unsigned long long t[EV_COUNT];
        HIPCHK(hipMemcpy(t, c->dStamps + (size_t)slot * EV_COUNT, sizeof(t), hipMemcpyDeviceToHost));
        const bool ev = useEvents && ago == 0 && c->evValid;   /* the events only remember the last iteration */
        auto span = [&](int a, int b) -> float {
            float ms = 0;
            if (ev) { if (hipEventElapsedTime(&ms, c->ev[a], c->ev[b]) != hipSuccess) ms = 0; }
            else ms = (float)((double)(t[b] - t[a]) / c->stampKHz);
            return ms;
        };
        out->msLight = span(EV_START, EV_LIGHT);
        out->msGrid = span(EV_GRID_K0, EV_GRID);
        out->msTotal = span(EV_START, EV_CAMERA);
        out->msLightKernel = span(EV_LIGHT_K0, EV_LIGHT_K1);
        if (!c->lightTraceOnly) {
            out->msCamera = span(EV_CAMERA_K0, EV_CONNECT_K1) + span(EV_MERGE_K0, EV_CAMERA);
            out->msCameraKernel = span(EV_CAMERA_K0, EV_CAMERA_K1);
            out->msConnectKernels = span(EV_CAMERA_K1, EV_CONNECT_K1);
            out->msQuerySort = span(EV_MERGE_K0, EV_SORT_K1);
            out->msMergeKernel = span(EV_SORT_K1, EV_MERGE_K1);
        }
    }
    if (ago == 0) c->lastStats = *out;
    return 0;
}
int vcm_get_stats(vcm_ctx *c, vcm_stats *out) { return vcm_get_stats_at(c, 0, out); }

int vcm_get_rng_counts(vcm_ctx *c, unsigned char *lightCounts, unsigned char *cameraCounts)
{
    if (!c) return fail("vcm_get_rng_counts", "ctx is NULL");
    if (!c->deviceReady) return fail("vcm_get_rng_counts", "no iteration has run");
    if (use_device(c)) return -1;
    if (lightCounts) HIPCHK(hipMemcpyAsync(lightCounts, c->dRngLight, (size_t)c->nLocal, hipMemcpyDeviceToHost, c->stream));
    if (cameraCounts) HIPCHK(hipMemcpyAsync(cameraCounts, c->dRngCam, (size_t)c->nLocal, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int vcm_local_path_range(vcm_ctx *c, int *first, int *count)
{
    if (!c) return fail("vcm_local_path_range", "ctx is NULL");
    if (first) *first = c->p0;
    if (count) *count = c->nLocal;
    return 0;
}

/* ---- parity/debug read-backs (used by tests/, not by the render path) ---- */

/* hash grid of the last iteration: cellStart (nCells+1 ints), sortedIndex
 * (grid position -> record index, nRecords ints), bbox (6 floats) */
int vcm_debug_read_grid(vcm_ctx *c, int *cellStart, int *sortedIndex, float *bbox6, long long *nRecords)
{
    if (scratch_readable(c, "vcm_debug_read_grid")) return -1;
    if (use_device(c)) return -1;
    /* the context's stream may be a caller's non-blocking stream (torch): a plain hipMemcpy would not wait for it */
    HIPCHK(hipStreamSynchronize(c->stream));
    GridHeader hdr;
    HIPCHK(hipMemcpy(&hdr, c->dHdr, sizeof(hdr), hipMemcpyDeviceToHost));
    if (nRecords) *nRecords = hdr.nRecords;
    if (cellStart) HIPCHK(hipMemcpy(cellStart, c->dCellStart, ((size_t)c->P.nCells + 1) * sizeof(int), hipMemcpyDeviceToHost));
    if (sortedIndex && hdr.nRecords > 0)
        HIPCHK(hipMemcpy(sortedIndex, c->dSortedIndex, (size_t)hdr.nRecords * sizeof(int), hipMemcpyDeviceToHost));
    if (bbox6) { memcpy(bbox6, hdr.bboxMin, 12); memcpy(bbox6 + 3, hdr.bboxMax, 12); }
    return 0;
}

/* local merge records of the last iteration, host copy (count from vcm_light_records) */
int vcm_debug_read_records(vcm_ctx *c, float *out, long long count)
{
    if (scratch_readable(c, "vcm_debug_read_records")) return -1;
    if (use_device(c)) return -1;
    if (ensure_records(c)) return -1;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (count > 0)
        HIPCHK(hipMemcpy(out, c->dRecordsLocal, (size_t)count * VCM_MERGE_RECORD_FLOATS * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

/* element-wise device evaluation of the numeric spec (detmath / philox):
 * op 0 sinf(a), 1 cosf(a), 2 powf(a,b), 3 a/b, 4 sqrtf(a), 5 dot-style a*b+c via mul,add,
 * 6 powf(a,b) with its tables read from LDS (the path of the kernels that sample Phong lobes), 7 the Phong lobe's pow
 * (dm_powf_wave) as the integer-exponent kernels evaluate it, 8 sinf(a) through the call of the cold sites */
} // extern "C"

namespace vcm {
__global__ void k_numeric_spec(int op, int n, const float *a, const float *b, float *out)
{
    if (op == 6) { dm_stage_tables(); __syncthreads(); }   /* before any thread leaves */
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (op) {
    case 6: r = dm_powf(a[i], b[i], true); break;
    case 7: r = dm_powf_wave(a[i], b[i], false, true); break;
    case 8: { float cc; dm_sincosf_cold(a[i], r, cc); (void)cc; } break;
    case 0: r = dm_sinf(a[i]); break;
    case 1: r = dm_cosf(a[i]); break;
    case 2: r = dm_powf(a[i], b[i]); break;
    case 3: r = a[i] / b[i]; break;
    case 4: r = sqrtf(a[i]); break;
    default: { float t = a[i] * b[i]; t += a[i]; r = t; } break;
    }
    out[i] = r;
}
template <class SC>
__global__ void k_kat(const DScene *__restrict__ scp, int op, int n, const float *in, float *out)
{
    stage_scene_tables(*scp);   /* before any thread leaves: it holds a barrier */
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a[VCM_KAT_FLOATS], r[VCM_KAT_FLOATS];
    for (int k = 0; k < VCM_KAT_FLOATS; k++) a[k] = in[(size_t)i * VCM_KAT_FLOATS + k];
    kat_eval(*static_cast<const SC *>(scp), op, a, r);
    for (int k = 0; k < VCM_KAT_FLOATS; k++) out[(size_t)i * VCM_KAT_FLOATS + k] = r[k];
}
__global__ void k_philox_spec(uint32_t seed, uint32_t iter, uint32_t kind, int nPaths, int nFloats, float *out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nPaths) return;
    PathRng r;
    rng_init(r, seed, iter, (uint32_t)p, kind);
    for (int k = 0; k < nFloats; k++) out[(size_t)p * nFloats + k] = rng_float(r);
}
} // namespace vcm

extern "C" {

int vcm_debug_numeric_spec(int op, int n, const float *a, const float *b, float *out)
{
    float *da = NULL, *db = NULL, *dout = NULL;
    HIPCHK(hipMalloc((void **)&da, (size_t)n * 4)); HIPCHK(hipMalloc((void **)&db, (size_t)n * 4));
    HIPCHK(hipMalloc((void **)&dout, (size_t)n * 4));
    HIPCHK(hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_numeric_spec, dim3((n + 255) / 256), dim3(256), 0, 0, op, n, da, db, dout);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return 0;
}

int vcm_debug_kat(vcm_ctx *c, int op, int n, const float *in, float *out)
{
    if (!c || !in || !out || n < 0 || op < 0 || op >= VCM_KAT_OPS) return fail("vcm_debug_kat", "bad argument");
    if (ensure_device(c)) return -1;
    if (n == 0) return 0;
    float *din = NULL, *dout = NULL;
    const size_t bytes = (size_t)n * VCM_KAT_FLOATS * sizeof(float);
    HIPCHK(hipMalloc((void **)&din, bytes));
    HIPCHK(hipMalloc((void **)&dout, bytes));
    HIPCHK(hipMemcpy(din, in, bytes, hipMemcpyHostToDevice));
    LAUNCH_SC(c, k_kat, dim3((n + 63) / 64), dim3(64), 0, 0, (const DScene *)c->dScene, op, n, (const float *)din, dout);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(din); (void)hipFree(dout);
    if (e != hipSuccess) return fail("vcm_debug_kat", hipGetErrorString(e));
    return 0;
}

int vcm_debug_philox_spec(unsigned seed, unsigned iter, unsigned kind, int nPaths, int nFloats, float *out)
{
    float *dout = NULL;
    const size_t n = (size_t)nPaths * nFloats;
    HIPCHK(hipMalloc((void **)&dout, n * 4));
    hipLaunchKernelGGL(k_philox_spec, dim3((nPaths + 255) / 256), dim3(256), 0, 0, seed, iter, kind, nPaths, nFloats, dout);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dout);
    return 0;
}

/* host-side evaluation of the same spec functions (the radius schedule uses
 * dm_powf on the host: vcm_begin_iteration) */
float vcm_host_sinf(float x) { return dm_sinf(x); }
float vcm_host_cosf(float x) { return dm_cosf(x); }
float vcm_host_powf(float x, float y) { return dm_powf(x, y); }
float vcm_host_path_float(unsigned seed, unsigned iter, unsigned path, unsigned kind, unsigned k)
{
    PathRng r;
    rng_init(r, seed, iter, path, kind);
    float f = 0;
    for (unsigned i = 0; i <= k; i++) f = rng_float(r);
    return f;
}
unsigned vcm_sizeof_scene_desc(void) { return (unsigned)sizeof(vcm_scene_desc); }
unsigned vcm_sizeof_stats(void) { return (unsigned)sizeof(vcm_stats); }

} // extern "C"

#if defined(VCM_K4_TIMES)   /* measurement variant only: profiles/tools/k4_tail.py */
extern "C" int k4_times_read(unsigned long long *out /* 2 x 32768 */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(vcm::g_k4Times), sizeof(unsigned long long) * 2 * 32768) == hipSuccess ? 0 : -1;
}
#endif
#if defined(VCM_K4_REGIONS)   /* measurement variant only: profiles/tools/k4_regions.py */
extern "C" int k4_regions_read(unsigned long long *out /* 16 */, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(vcm::g_k4Regions), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(vcm::g_k4Regions), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
#if defined(VCM_K4_STEPS)   /* measurement variant only: profiles/tools/k4_lanes.py */
extern "C" int k4_steps_read(unsigned short *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(vcm::g_k4Steps), sizeof(unsigned short) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif
#if defined(VCM_REGION_CLOCK)   /* measurement variant only: profiles/tools/region_clock.py */
extern "C" int region_clock_read(unsigned long long *out, int reset)   /* out: 3 x VCM_RC_IDS words */
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    const int W = 3 * VCM_RC_IDS;
    std::vector<unsigned long long> all((size_t)VCM_RC_SLOTS * W);
    if (out) {
        if (hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(vcm::g_regionClock), all.size() * sizeof(unsigned long long)) != hipSuccess) return -1;
        for (int k = 0; k < W; k++) { out[k] = 0; for (int s = 0; s < VCM_RC_SLOTS; s++) out[k] += all[(size_t)s * W + k]; }
    }
    if (reset) {
        std::fill(all.begin(), all.end(), 0ull);
        if (hipMemcpyToSymbol(HIP_SYMBOL(vcm::g_regionClock), all.data(), all.size() * sizeof(unsigned long long)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
