/*
 * smallvcm_amd.h -- C-ABI of the MI355X-native VCM integrator.
 *
 * This is the drop-in boundary for SmallVCM's `VertexCM::RunIteration`
 * (reference: src/vertexcm.hxx:284-548) behind `AbstractRenderer`
 * (src/renderer.hxx:33-70).  Everything crossing it is plain C: POD structs,
 * pointers and sizes.  The C++ shim `smallvcm_amd/dropin/vertexcm.hxx` binds
 * these entry points under the reference's own class name/ctor so that
 * `smallvcm.cxx` and `config.hxx` compile unchanged.
 *
 * Conventions
 *   - all entry points returning int return 0 on success, non-zero on error;
 *     `vcm_last_error()` then holds a message (thread-local).
 *   - "paths" are indexed 0..N-1, N = resX*resY, light path p and camera
 *     path (pixel) p share the index (vertexcm.hxx:321, :415, :504-506).
 *   - random numbers: counter-based Philox4x32-10 keyed by
 *     (seed, renderer-local iteration count), counter (path, kind, block);
 *     replaces src/rng.hxx (see DESIGN.md "RNG").
 */
#ifndef SMALLVCM_AMD_H
#define SMALLVCM_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define VCM_MAX_PRIMS      32
#define VCM_MAX_MATERIALS  16
#define VCM_MAX_LIGHTS      8

/* ---- scene description: POD flattening of the public members of `Scene`
 *      (src/scene.hxx:476-485) ------------------------------------------- */

enum { VCM_PRIM_TRIANGLE = 0, VCM_PRIM_SPHERE = 1 };

/* Triangle {p[3], matID, mNormal} (src/geometry.hxx:174-176) or
 * Sphere {center, radius, matID} (src/geometry.hxx:263-265: p0 = center,
 * p1[0] = radius).  Primitives keep the order of GeometryList::mGeometry
 * (src/geometry.hxx:104); closest-hit ties resolve to the first one. */
typedef struct vcm_prim {
    int   type;
    int   matID;
    float p0[3];
    float p1[3];
    float p2[3];
    float n[3];
} vcm_prim;

/* Material (src/materials.hxx:54-65) */
typedef struct vcm_material {
    float diffuse[3];
    float phong[3];
    float phongExp;
    float mirror[3];
    float ior;
} vcm_material;

enum {
    VCM_LIGHT_AREA = 0,        /* src/lights.hxx:112 */
    VCM_LIGHT_DIRECTIONAL = 1, /* src/lights.hxx:236 */
    VCM_LIGHT_POINT = 2,       /* src/lights.hxx:320 */
    VCM_LIGHT_BACKGROUND = 3   /* src/lights.hxx:401 */
};

/* Tagged union of the public light fields (src/lights.hxx:229-232, 314-315,
 * 395-396, 512-513).
 *   area:        p0,e1,e2, frame (mX,mY,mZ), intensity, invArea
 *   directional: frame, intensity
 *   point:       p0 = mPosition, intensity
 *   background:  intensity = mBackgroundColor, scale = mScale            */
typedef struct vcm_light {
    int   type;
    float p0[3];
    float e1[3];
    float e2[3];
    float frameX[3];
    float frameY[3];
    float frameZ[3];
    float intensity[3];
    float invArea;
    float scale;
} vcm_light;

/* Camera (src/camera.hxx:121-126).  Matrices are the 16 floats of Mat4f in
 * its memory order (column-major, src/math.hxx:255-259). */
typedef struct vcm_camera {
    float position[3];
    float forward[3];
    float resolution[2];
    float rasterToWorld[16];
    float worldToRaster[16];
    float imagePlaneDist;
} vcm_camera;

typedef struct vcm_scene_desc {
    int          nPrims;
    vcm_prim     prims[VCM_MAX_PRIMS];
    int          nMaterials;
    vcm_material materials[VCM_MAX_MATERIALS];
    int          mat2light[VCM_MAX_MATERIALS]; /* Scene::mMaterial2Light, -1 = none */
    int          nLights;
    vcm_light    lights[VCM_MAX_LIGHTS];
    int          backgroundLight;              /* index into lights or -1 (Scene::mBackground) */
    float        sceneCenter[3];               /* SceneSphere (src/lights.hxx:32-40) */
    float        sceneRadius;
    float        invSceneRadiusSqr;
    vcm_camera   camera;
} vcm_scene_desc;

/* Scene description, version 2: the same members with POINTER + count for primitives, materials and lights, for
 * scenes beyond the reference's built-in Cornell boxes (which is all its CLI can load: config.hxx:146-151; "no
 * acceleration structure", README:208-209; Scene::Intersect is a loop over every primitive, scene.hxx:53-70).  A
 * scene with more than VCM_MAX_PRIMS primitives is traced through a BVH built at vcm_create2; the results are those of
 * the reference's list walk (closest hit, ties to the lower list index: geometry.hxx:65-78).  The arrays are copied:
 * they may be freed after the call.  mat2light has nMaterials entries (Scene::mMaterial2Light: an emissive triangle
 * needs a material of its own, as in scene.hxx:333-361); at most 2^24 materials. */
typedef struct vcm_scene_desc2 {
    int                 nPrims;
    const vcm_prim     *prims;
    int                 nMaterials;
    const vcm_material *materials;
    const int          *mat2light;
    int                 nLights;
    const vcm_light    *lights;
    int                 backgroundLight;
    float               sceneCenter[3];
    float               sceneRadius;
    float               invSceneRadiusSqr;
    vcm_camera          camera;
} vcm_scene_desc2;

/* VertexCM::AlgorithmType (src/vertexcm.hxx:182-204) -- same values */
enum {
    VCM_ALGO_LIGHT_TRACE = 0,
    VCM_ALGO_PPM = 1,
    VCM_ALGO_BPM = 2,
    VCM_ALGO_BPT = 3,
    VCM_ALGO_VCM = 4,
    /* the reference's two other renderers behind the same interface (AbstractRenderer, renderer.hxx:33-70;
       created at config.hxx:118-121).  radiusFactor / radiusAlpha are ignored. */
    VCM_ALGO_PATH_TRACE = 5,                   /* PathTracer  src/pathtracer.hxx:45-215 */
    VCM_ALGO_EYE_LIGHT = 6                     /* EyeLight    src/eyelight.hxx:46-77 */
};

/* Per-iteration workload counters (what SURVEY.md section 8(d) calls N_LV, C, A, K, S)
 * of the most recent iteration on this rank, plus kernel times in ms measured
 * with HIP events on the context's stream. */
typedef struct vcm_stats {
    long long lightVertices;   /* N_LV stored on this rank                    */
    long long gridVertices;    /* vertices in the hash grid (all ranks)       */
    long long lightRays;       /* Scene::Intersect calls, light pass          */
    long long cameraRays;      /* Scene::Intersect calls, camera pass         */
    long long shadowRays;      /* Scene::Occluded calls                       */
    long long mergeQueries;    /* HashGrid::Process calls                     */
    long long mergeCandidates; /* C: distance tests (hashgrid.hxx:162-165)    */
    long long mergeAccepted;   /* A: RangeQuery::Process calls                */
    long long connections;     /* K: ConnectVertices calls                    */
    long long lightSplats;     /* S: Framebuffer::AddColor from light paths   */
    float msLight, msGrid, msCamera, msTotal; /* phases: light(+compaction; the light splats run on a stream of their own
                                                 and end inside msCamera), grid build, camera(+resolve) */
    float msLightKernel, msCameraKernel;      /* k_light_trace / k_camera_trace alone        */
    float msMergeKernel;                      /* k_merge_lane (0 in strict-order mode)       */
    float msQuerySort;                        /* camera-vertex counting sort (0 in strict mode) */
    float msConnectKernels;                   /* k_connect_di (0 in strict mode); k_connect_vc runs on another stream and is
                                                 not in this span */
    float radius;              /* merge radius of the iteration               */
} vcm_stats;

typedef struct vcm_ctx vcm_ctx;

/* Record layout used to exchange light vertices between ranks (13 floats =
 * 52 bytes: the part of LightVertex that RangeQuery::Process reads,
 * src/vertexcm.hxx:130-169). */
#define VCM_MERGE_RECORD_FLOATS 13

int         vcm_device_count(void);
const char *vcm_last_error(void);
/* "default", or name and flags of a measurement build of the library (profiles/quick_ab.sh prints it per run) */
const char *vcm_build_tag(void);

/* Replaces `new VertexCM(scene, algo, radiusFactor, radiusAlpha, seed)`
 * (src/vertexcm.hxx:208-282, called from src/config.hxx:124-138).
 * Includes the PPM->BPM downgrade of :246-278.  Device resources are
 * allocated lazily at the first iteration (unused renderers exist:
 * src/renderer.hxx:58, src/smallvcm.cxx:66). */
vcm_ctx *vcm_create(const vcm_scene_desc *scene, int algorithm,
                    float radiusFactor, float radiusAlpha, int seed);
/* vcm_create puts the renderer on the calling thread's CURRENT HIP device (a host that hands over its own stream or
 * device buffers allocated them there).  A renderer-per-host-core host on a multi-GPU node -- the reference's render()
 * builds one renderer per host core, smallvcm.cxx:61-72 -- asks vcm_next_device() for the device of its next renderer
 * and passes it to vcm_create_sharded(..., device, 0, 1): round-robin over the visible devices, so that every GPU
 * renders whole iterations of its share of the renderers and the driver's own framebuffer average is the only reduce
 * (this is what the drop-in does, smallvcm_amd/dropin/gpu_renderer.hxx).  Environment SMALLVCM_AMD_DEVICES: "all"
 * (what vcm_next_device deals over by default), "current" or a list such as "0,2,5"; when it is set, vcm_create
 * follows it too. */
int vcm_next_device(void);

/* Same, for one rank of a sharded renderer: rank r of worldSize traces light
 * paths and pixels [r*N/W, (r+1)*N/W) on HIP device `device`. */
vcm_ctx *vcm_create_sharded(const vcm_scene_desc *scene, int algorithm,
                            float radiusFactor, float radiusAlpha, int seed,
                            int device, int rank, int worldSize);

/* The same for a version-2 scene description. */
vcm_ctx *vcm_create2(const vcm_scene_desc2 *scene, int algorithm,
                     float radiusFactor, float radiusAlpha, int seed);
vcm_ctx *vcm_create_sharded2(const vcm_scene_desc2 *scene, int algorithm,
                             float radiusFactor, float radiusAlpha, int seed,
                             int device, int rank, int worldSize);

void vcm_destroy(vcm_ctx *ctx);

/* Execution mode.  0 (default, "wavefront"): the camera pass only traces and
 * scatters; direct illumination, vertex connections and merges are evaluated
 * by dense task kernels and every path's additions are replayed in the
 * reference's order, light splats included: bit-identical to the reference for
 * every algorithm.  1 ("strict"): everything is evaluated inside the paths as
 * the reference does (about 2x slower; light splats are fp32 atomics there).
 * The environment variable SMALLVCM_AMD_STRICT_ORDER=1 sets the default. */
int vcm_set_strict_order(vcm_ctx *ctx, int on);
/* 1 if an iteration with this maxPathLength will run in wavefront mode, 0 if it falls back to / was set to the
 * strict order (vcm_set_strict_order, or maxPathLength > 31: the per-path vertex masks are 32 bits).  A sharded
 * host asks this to know whether vcm_trace_camera may run before vcm_build_grid. */
int vcm_is_wavefront(vcm_ctx *ctx, unsigned maxPathLength);

/* The merge sharded by SPACE (round 6 prototype).  With the calls above every rank merges ITS pixels' camera
 * vertices against ALL ranks' light vertices (vertexcm.hxx:532-533): the grid and the merge do not shrink with the shard.  Here rank s
 * owns a SLAB of un-hashed grid cells along one axis -- cells [X[s], X[s+1]) -- and one cell of halo: it receives the light vertices of
 * cells [X[s] - 1, X[s+1]] only, builds HashGrid::Build's grid over those (vcm_import_light_records + vcm_build_grid: records arrive grouped by
 * source rank = in global index order, so the in-cell order is the reference's), receives the camera vertices whose base cell lies in its slab
 * from every rank, evaluates RangeQuery::Process for them and sends the 16-byte terms back to the pixels' owner.  A query's accepted photons
 * all lie in its own 2 x 2 x 2 block of cells (hashgrid.hxx:124-155), so its accepted sequence -- and the frame -- are the unsharded
 * renderer's; the candidate COUNT is not (a hash bucket no longer holds the far-away cells that collide into it).
 *   vcm_space_histogram          after vcm_trace_light: 256 bins of the local light vertices' coordinate along `axis` over the scene's
 *                                bounding sphere ([*lo, *lo + 256 * *binWidth)); the ranks sum them and pick S - 1 split coordinates
 *   vcm_space_set_slabs          after vcm_set_grid_bbox: splits[1 .. S-1] = the world coordinates where slab s begins (every rank
 *                                passes the same values)
 *   vcm_space_partition_light    the local light records (13 floats each) grouped by destination slab, index order inside a destination,
 *                                halo vertices once per destination: destination d at dstDev + d * strideRecords records; counts[d] to
 *                                the host (one stream synchronisation).  The host exchanges the groups all-to-all and hands what it received
 *                                to vcm_import_light_records (segment s = what rank s sent), then vcm_build_grid
 *   vcm_space_partition_queries  after vcm_trace_camera: the camera vertices (64-byte records) grouped by the slab of their base cell
 *   vcm_space_merge              after vcm_build_grid: queriesDev = nSeg segments (rank order) of counts[s] queries at a stride of
 *                                `stride` queries; sorts and merges them against this rank's grid; resultsDev gets the terms (16 bytes
 *                                each) in the same layout
 *   vcm_space_import_results     resultsDev = for every destination d the terms of the queries this rank sent to d (the layout
 *                                vcm_space_partition_queries wrote); vcm_merge then skips its own merge and resolves */
int vcm_space_histogram(vcm_ctx *ctx, int axis, float *lo, float *binWidth, int *hist256);
int vcm_space_set_slabs(vcm_ctx *ctx, int axis, const float *splits, int nSlabs);
int vcm_space_partition_light(vcm_ctx *ctx, void *dstDev, long long strideRecords, long long *counts);
int vcm_space_partition_queries(vcm_ctx *ctx, void *dstDev, long long strideQueries, long long *counts);
int vcm_space_merge(vcm_ctx *ctx, const void *queriesDev, const long long *counts, int nSeg, long long stride, void *resultsDev);
int vcm_space_import_results(vcm_ctx *ctx, const void *resultsDev, long long stride);

/* Which kernel evaluates the range merges (HashGrid::Process, hashgrid.hxx:110-169) in wavefront mode: both produce the
 * same bits, they differ in who evaluates an accepted photon (vcm_kernels.h; measured in DESIGN.md 5).
 * The environment variable SMALLVCM_AMD_MERGE=walk|pairs sets the default (pairs). */
#define VCM_MERGE_LANE   0   /* retired in round 6 (the 8 cells in lockstep): vcm_set_merge_kernel refuses it */
#define VCM_MERGE_STAGED 1   /* retired in round 6 (cell lists staged through LDS by a workgroup): refused */
#define VCM_MERGE_WALK   2   /* every lane walks its own non-empty runs back to back and evaluates its own accepted photons */
#define VCM_MERGE_PAIRS  3   /* the scan per lane as in WALK; the accepted (query, photon) pairs of a wave evaluated 64 at a time
                                by whichever lane gets them (scenes with more than 32 materials: WALK) */
int vcm_set_merge_kernel(vcm_ctx *ctx, int kind);

/* Use an externally owned HIP stream (e.g. torch's current stream) for all
 * work of this context; NULL = the context's own stream. */
int vcm_set_stream(vcm_ctx *ctx, void *hipStream);

/* Iteration scratch (light-vertex store, hash grid, camera-vertex queues: 1.5 GB at 512^2, 24 GB at 2048^2) is not
 * owned by a renderer: single-rank renderers of a device borrow it per iteration from a pool of arenas, so the
 * reference's one-renderer-per-host-core driver (smallvcm.cxx:66) fits in HBM and iterations of different renderers
 * overlap on the GPU.  maxArenas 1..8, 0 = as many as fit a quarter of the device memory (default; environment
 * SMALLVCM_AMD_ARENAS). */
int vcm_set_arena_limit(int device, int maxArenas);

/* Optional: allocate up front everything the first iteration with this maxPathLength would allocate lazily
 * (the context's buffers and the device's scratch arena), e.g. before a timed region. */
int vcm_reserve(vcm_ctx *ctx, unsigned maxPathLength);
/* Replaces VertexCM::RunIteration(aIteration) (src/vertexcm.hxx:284-548) for a
 * single-GPU renderer.  minLen/maxLen are AbstractRenderer::mMinPathLength /
 * mMaxPathLength, which the driver assigns after construction
 * (src/smallvcm.cxx:70-71).  Asynchronous on the context's stream. */
int vcm_run_iteration(vcm_ctx *ctx, int iteration, unsigned minLen, unsigned maxLen);

/* Phase-level entry points: vcm_run_iteration == begin, trace_light,
 * build_grid, trace_camera, merge, end.  A multi-GPU host puts the all-gather
 * of the light-vertex records between trace_light and build_grid; in the
 * default (wavefront) mode vcm_trace_camera does not need the grid, so it may
 * run BEFORE vcm_build_grid, overlapping the all-gather. */
/* A phase call that fails ENDS the iteration (HIP error or wrong call order alike): the iteration scratch goes
 * back to the device's arena and the next call must be vcm_begin_iteration. */
int vcm_begin_iteration(vcm_ctx *ctx, int iteration, unsigned minLen, unsigned maxLen); /* :288-316 */
int vcm_trace_light(vcm_ctx *ctx);   /* :321-396, then compaction into merge records */
int vcm_build_grid(vcm_ctx *ctx);    /* :403-408 -> hashgrid.hxx:41-107 */
int vcm_trace_camera(vcm_ctx *ctx);  /* :415-545 (wavefront mode: all but the merge :530-538) */
int vcm_merge(vcm_ctx *ctx);         /* :530-538 for every camera vertex, then AddColor :544 */
int vcm_end_iteration(vcm_ctx *ctx); /* :547 */

/* Local merge records of this rank after vcm_trace_light: device pointer to
 * count x VCM_MERGE_RECORD_FLOATS floats, in the reference's vertex order.
 * `count` is read back from the device (synchronises the stream). */
int vcm_light_records(vcm_ctx *ctx, void **devPtr, long long *count);

/* Sharded hosts: the bounding box (and number) of this rank's light vertices after vcm_trace_light -- empty:
 * min = +1e36, max = -1e36 (hashgrid.hxx:47-48) -- and, the other way, the box of ALL ranks' vertices (min / max of
 * the per-rank boxes), to be set before vcm_trace_camera / vcm_build_grid.  Exchanging 7 numbers per rank instead
 * of the count alone lets the camera pass prepare the query sort while the vertices are still in flight. */
int vcm_local_light_bbox(vcm_ctx *ctx, float *min3, float *max3, long long *count);
int vcm_set_grid_bbox(vcm_ctx *ctx, const float *min3, const float *max3);
/* Copy the local merge records (count from vcm_light_records) / the
 * framebuffer into caller-owned DEVICE memory (e.g. a torch tensor that is
 * then handed to an RCCL collective).  Asynchronous on the context's stream. */
int vcm_export_light_records(vcm_ctx *ctx, void *dstDev, long long count);
int vcm_export_framebuffer(vcm_ctx *ctx, void *dstDev);
/* the same multiplied by `scale` (1 / iterations gives GetFramebuffer, renderer.hxx:49-55): what a multi-GPU host
 * hands to the framebuffer all-reduce */
int vcm_export_framebuffer_scaled(vcm_ctx *ctx, void *dstDev, float scale);

/* Install the all-gathered records: nSeg segments (one per rank, rank order),
 * segment s holds counts[s] records starting at devPtr + s*strideRecords
 * records.  Repacks them contiguously; build_grid then uses them. */
int vcm_import_light_records(vcm_ctx *ctx, const void *devPtr,
                             const long long *counts, int nSeg,
                             long long strideRecords);

/* The SORTED exchange of a sharded renderer (round 5; replaces the three calls above on the hot path -- they stay for hosts
 * that want the records in the reference's order).  Path-index blocks are contiguous per rank, so a cell's vertices in
 * HashGrid::Build's stable order (hashgrid.hxx:83-88) are rank 0's in their local order, then rank 1's, ...: every rank
 * counting-sorts ONLY ITS OWN vertices by hash cell (1 / worldSize of the grid build), the ranks all-gather the sorted
 * slabs, and vcm_build_grid merges them cell block by cell block in one streaming pass -- no rank ever counts, scans or
 * ranks the other ranks' vertices (rounds 2-4: the whole build replicated on every rank).
 *   vcm_sorted_slab_words    4-byte words one rank's slab takes for `strideRecords` records: the records (13 words each,
 *                            cell order) + one block-start word per block of cells (+ padding to 16 bytes); -1 with
 *                            vcm_last_error() when the context cannot use the sorted exchange (not sharded, no merging
 *                            algorithm, more than 64 shards, 2^24 or more records in a shard): use the calls above.
 *   vcm_sort_light_records   after vcm_set_grid_bbox (the cell of a vertex depends on the box of ALL vertices): sorts the
 *                            local vertices and writes the slab to dstDev (device memory of the caller, e.g. the send
 *                            buffer of ncclAllGather).  Asynchronous on the context's stream.
 *   vcm_import_sorted_light_records   `gathered` = worldSize slabs of vcm_sorted_slab_words(strideRecords) words, rank
 *                            order, counts[s] = records of rank s; vcm_build_grid then merges them.  The memory must stay
 *                            valid until vcm_merge has been enqueued. */
long long vcm_sorted_slab_words(vcm_ctx *ctx, long long strideRecords);
int vcm_sort_light_records(vcm_ctx *ctx, void *dstDev, long long strideRecords);
int vcm_import_sorted_light_records(vcm_ctx *ctx, const void *gathered, const long long *counts, int nSeg,
                                    long long strideRecords);

/* The image in the reference's two 8-bit output encodings, converted on the device (a quarter / a third of
 * the bytes of the fp32 framebuffer cross PCIe).  The framebuffer (running sum) is scaled by `scale` first --
 * 1 / iterations gives what GetFramebuffer returns (renderer.hxx:49-55).
 *   VCM_IMAGE_BGR8  Framebuffer::SaveBMP's pixel data (framebuffer.hxx:194-214): rows bottom-up, B,G,R bytes,
 *                   byte(min(255, max(0, pow(c, 1/gamma) * 255))); resX*resY*3 bytes.  pow is the deterministic
 *                   powf of this library, so a byte can differ by one level from a glibc build of the reference.
 *   VCM_IMAGE_RGBE  Framebuffer::SaveHDR's pixel data (:229-247): rows top-down, R,G,B,E bytes; resX*resY*4
 *                   bytes, bit-identical to the reference's (frexp and IEEE double arithmetic only). */
enum { VCM_IMAGE_BGR8 = 0, VCM_IMAGE_RGBE = 1 };
int vcm_read_image(vcm_ctx *ctx, int format, float scale, float gamma, unsigned char *outHost);
/* Framebuffer = running SUM over iterations of this context (the reference's
 * mFramebuffer, src/renderer.hxx:68); W*H*3 floats, row-major, RGB. */
int vcm_read_framebuffer(vcm_ctx *ctx, float *rgbHost);
/* Page-lock / release caller-owned HOST memory (hipHostRegister / hipHostUnregister), so that vcm_read_framebuffer into it is
 * one DMA at link speed instead of a staged copy through the runtime's bounce buffers.  The drop-in registers the reference
 * Framebuffer's own pixel storage once (dropin/gpu_renderer.hxx) and refreshes it after every RunIteration
 * (src/renderer.hxx:49-55).  Both return 0 or -1 with vcm_last_error(); a host that cannot pin simply reads slower. */
int vcm_pin_host_memory(void *hostPtr, unsigned long long bytes);
int vcm_unpin_host_memory(void *hostPtr);
int vcm_framebuffer_device(vcm_ctx *ctx, void **devPtr);
int vcm_clear_framebuffer(vcm_ctx *ctx);

int vcm_iterations(vcm_ctx *ctx); /* AbstractRenderer::mIterations */
int vcm_synchronize(vcm_ctx *ctx);
int vcm_get_stats(vcm_ctx *ctx, vcm_stats *out);
/* The same for an earlier iteration: `ago` = 0 is the last completed one; the counters and the phase times
 * (stamped on the context's stream by every iteration) of the last 64 iterations are kept on the device, so a
 * host can run a batch of iterations without reading anything back in between and still get per-iteration
 * figures afterwards. */
int vcm_get_stats_at(vcm_ctx *ctx, int ago, vcm_stats *out);

/* Number of random floats each path consumed in the last iteration (local
 * paths of this rank): the "tape" that lets the unmodified reference replay
 * the same random numbers (oracle/ref_driver.cpp).  Host buffers of
 * localPathCount bytes each. */
int vcm_get_rng_counts(vcm_ctx *ctx, unsigned char *lightCounts, unsigned char *cameraCounts);
int vcm_local_path_range(vcm_ctx *ctx, int *first, int *count);

/* Host-side restatement of Scene::LoadCornellBox + BuildSceneSphere
 * (src/scene.hxx:132-398) and Camera::Setup (src/camera.hxx:37-76) for hosts
 * that do not link the reference's scene code (Python, bench). boxMask uses
 * Scene::BoxMask bits (src/scene.hxx:112-126). */
int vcm_scene_cornell(int resX, int resY, unsigned boxMask, vcm_scene_desc *out);
/* g_SceneConfigs[sceneID] (src/config.hxx:146-151) */
unsigned vcm_scene_config_mask(int sceneID);

/* Building a version-2 scene: one call per object, each computing the derived members the way the reference's
 * constructor does (same operations, same bits), so that a scene assembled here equals what the reference would hold
 * for the same input:
 *   vcm_make_triangle          Triangle::Triangle        geometry.hxx:111-123  (mNormal)
 *   vcm_make_sphere            Sphere::Sphere            geometry.hxx:184-192
 *   vcm_make_area_light        AreaLight::AreaLight      lights.hxx:116-127    (edges, frame, 1 / area)
 *   vcm_make_directional_light DirectionalLight          lights.hxx:239-243    (frame)
 *   vcm_make_point_light       PointLight                lights.hxx:324-328
 *   vcm_make_background_light  BackgroundLight           lights.hxx:404-408    (the reference's sky colour)
 *   vcm_make_material          Material::Reset           materials.hxx:44-51   (black, exponent 1, no refraction)
 *   vcm_make_camera            Camera::Setup             camera.hxx:37-76
 *   vcm_make_scene_sphere      Scene::BuildSceneSphere   scene.hxx:387-398
 * An emissive triangle is a primitive with a material of its own whose mat2light entry names the area light over the
 * same three points (scene.hxx:333-361). */
void vcm_make_triangle(const float *p0, const float *p1, const float *p2, int matID, vcm_prim *out);
void vcm_make_sphere(const float *center, float radius, int matID, vcm_prim *out);
void vcm_make_area_light(const float *p0, const float *p1, const float *p2, const float *intensity, vcm_light *out);
void vcm_make_directional_light(const float *direction, const float *intensity, vcm_light *out);
void vcm_make_point_light(const float *position, const float *intensity, vcm_light *out);
void vcm_make_background_light(float scale, vcm_light *out);
void vcm_make_material(vcm_material *out);
int  vcm_make_camera(const float *position, const float *forward, const float *up, float horizontalFovDeg, int resX, int resY,
                     vcm_camera *out);
void vcm_make_scene_sphere(const vcm_prim *prims, int nPrims, float *center3, float *radius, float *invRadiusSqr);

/* Scene files.  The reference has no loader ("Scenes are hard-coded", README:210; Scene::LoadCornellBox,
 * scene.hxx:132-398, is the only way a scene comes into being).  vcm_scene_load reads a `.vcmscene` text file --
 * directives processed in order: `obj <file>` (Wavefront OBJ triangles with their MTL library: Kd / Ks+Ns / illum /
 * Ni -> Material, Ke -> one AreaLight per triangle as in scene.hxx:333-361), `sphere`, `camera`, `light
 * point|directional|background`, `mtllib` -- or a bare `.obj` (default camera), and builds the version-2 description
 * with the vcm_make_* constructors above; smallvcm_amd/csrc/scene_file.cpp documents the format.  The description
 * points into the handle: keep it until the renderers are created (vcm_create2 copies).  NULL on failure, with the
 * reason in vcm_scene_load_error(). */
typedef struct vcm_scene_file vcm_scene_file;
vcm_scene_file *vcm_scene_load(const char *path, int resX, int resY);
const vcm_scene_desc2 *vcm_scene_file_desc(const vcm_scene_file *scene);
void vcm_scene_file_free(vcm_scene_file *scene);
const char *vcm_scene_load_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SMALLVCM_AMD_H */
