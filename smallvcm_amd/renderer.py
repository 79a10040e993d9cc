"""Python host of the MI355X VCM integrator.

Mirrors the reference's renderer plug-in surface (src/renderer.hxx:33-70,
src/vertexcm.hxx:61, :182-214) on top of the C-ABI (include/smallvcm_amd.h):

    r = VertexCM(scene, VertexCM.kVcm, radius_factor, radius_alpha, seed)
    r.mMaxPathLength = 10; r.mMinPathLength = 0      # smallvcm.cxx:70-71
    r.RunIteration(i)                                 # vertexcm.hxx:284
    fb = r.GetFramebuffer()                           # renderer.hxx:49-55

All compute happens in libsmallvcm_amd.so (HIP, gfx950).  There is no CPU
fallback: a missing library or GPU raises.

`ShardedVertexCM` is the one-process-per-GPU host: light sub-paths and pixels
are sharded by path index; between the light pass and the grid build every
rank all-gathers the light-vertex merge records (RCCL via torch.distributed),
and the framebuffers are summed once at read-out.
"""
import ctypes as C
import os

import numpy as np

from ._abi import (SceneDesc, SceneDesc2, Stats, SCENE_CONFIGS, VCM_MERGE_RECORD_FLOATS, ALGO_LIGHT_TRACE, ALGO_PPM, ALGO_BPM,
                   ALGO_BPT, ALGO_VCM)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMALLVCM_AMD_LIB") or os.path.join(_HERE, "csrc", "libsmallvcm_amd.so")

_lib = None


def load_library(require_gpu=True):
    """Load the HIP library and declare the C-ABI.  Raises if it is missing
    (build it with `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("smallvcm_amd: %s not built; the HIP library is the only compute path" % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  Two HIP runtimes
        # in one process do not share the GPU ("No HIP GPUs are available" from
        # whichever comes second), so let torch's copy load first; our library
        # then binds to the already-loaded libamdhip64 (same SONAME).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, ip, llp, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_float)
        L.vcm_last_error.restype = C.c_char_p
        L.vcm_device_count.restype = C.c_int
        L.vcm_create.restype = vp
        L.vcm_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_float, C.c_float, C.c_int]
        L.vcm_create_sharded.restype = vp
        L.vcm_create_sharded.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                         C.c_int, C.c_int]
        L.vcm_create_sharded2.restype = vp
        L.vcm_create_sharded2.argtypes = [C.POINTER(SceneDesc2), C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                          C.c_int, C.c_int]
        L.vcm_destroy.argtypes = [vp]
        L.vcm_destroy.restype = None
        L.vcm_set_stream.argtypes = [vp, vp]
        L.vcm_set_strict_order.argtypes = [vp, C.c_int]
        L.vcm_pin_host_memory.argtypes = [vp, C.c_ulonglong]
        L.vcm_unpin_host_memory.argtypes = [vp]
        L.vcm_set_merge_kernel.argtypes = [vp, C.c_int]
        L.vcm_is_wavefront.argtypes = [vp, C.c_uint]
        L.vcm_set_arena_limit.argtypes = [C.c_int, C.c_int]
        L.vcm_run_iteration.argtypes = [vp, C.c_int, C.c_uint, C.c_uint]
        L.vcm_begin_iteration.argtypes = [vp, C.c_int, C.c_uint, C.c_uint]
        for n in ("vcm_trace_light", "vcm_build_grid", "vcm_trace_camera", "vcm_merge", "vcm_end_iteration", "vcm_synchronize",
                  "vcm_clear_framebuffer", "vcm_iterations"):
            getattr(L, n).argtypes = [vp]
        L.vcm_light_records.argtypes = [vp, C.POINTER(vp), llp]
        L.vcm_export_light_records.argtypes = [vp, vp, C.c_longlong]
        L.vcm_export_framebuffer.argtypes = [vp, vp]
        L.vcm_import_light_records.argtypes = [vp, vp, llp, C.c_int, C.c_longlong]
        L.vcm_sorted_slab_words.argtypes = [vp, C.c_longlong]
        L.vcm_sorted_slab_words.restype = C.c_longlong
        L.vcm_sort_light_records.argtypes = [vp, vp, C.c_longlong]
        L.vcm_import_sorted_light_records.argtypes = [vp, vp, llp, C.c_int, C.c_longlong]
        L.vcm_read_framebuffer.argtypes = [vp, fp]
        L.vcm_framebuffer_device.argtypes = [vp, C.POINTER(vp)]
        L.vcm_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.vcm_get_stats_at.argtypes = [vp, C.c_int, C.POINTER(Stats)]
        L.vcm_reserve.argtypes = [vp, C.c_uint]
        L.vcm_read_image.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_ubyte)]
        L.vcm_local_light_bbox.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), llp]
        L.vcm_set_grid_bbox.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.vcm_get_rng_counts.argtypes = [vp, C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte)]
        L.vcm_local_path_range.argtypes = [vp, ip, ip]
        L.vcm_scene_cornell.argtypes = [C.c_int, C.c_int, C.c_uint, C.POINTER(SceneDesc)]
        L.vcm_scene_config_mask.argtypes = [C.c_int]
        L.vcm_scene_config_mask.restype = C.c_uint
        L.vcm_debug_read_grid.argtypes = [vp, ip, ip, fp, llp]
        L.vcm_debug_read_records.argtypes = [vp, fp, C.c_longlong]
        L.vcm_debug_numeric_spec.argtypes = [C.c_int, C.c_int, fp, fp, fp]
        L.vcm_debug_philox_spec.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_int, fp]
        for n in ("vcm_host_sinf", "vcm_host_cosf"):
            getattr(L, n).argtypes = [C.c_float]
            getattr(L, n).restype = C.c_float
        L.vcm_host_powf.argtypes = [C.c_float, C.c_float]
        L.vcm_host_powf.restype = C.c_float
        L.vcm_host_path_float.argtypes = [C.c_uint] * 5
        L.vcm_host_path_float.restype = C.c_float
        L.vcm_sizeof_scene_desc.restype = C.c_uint
        L.vcm_sizeof_stats.restype = C.c_uint
        _lib = L
    if require_gpu and _lib.vcm_device_count() <= 0:
        raise RuntimeError("smallvcm_amd: no HIP device visible; this package has no CPU compute path")
    return _lib


def _check(L, rc, what):
    if rc != 0:
        raise RuntimeError("smallvcm_amd: %s failed: %s" % (what, (L.vcm_last_error() or b"").decode()))


def cornell_scene(scene_id_or_mask, resx=512, resy=512, is_mask=False):
    """Built-in Cornell box (reference src/scene.hxx:132, scene ids of
    src/config.hxx:146-151) as a SceneDesc.  Host-only, needs no GPU."""
    L = load_library(require_gpu=False)
    mask = scene_id_or_mask if is_mask else SCENE_CONFIGS[scene_id_or_mask]
    d = SceneDesc()
    _check(L, L.vcm_scene_cornell(resx, resy, mask, C.byref(d)), "vcm_scene_cornell")
    return d


class HipBackend:
    """One vcm_ctx: the phases of an iteration on one GPU (include/smallvcm_amd.h)."""

    def __init__(self, scene, algorithm, radius_factor, radius_alpha, seed, device=0, rank=0, world=1):
        self.L = load_library()
        self.scene = scene
        self.resx = int(scene.camera.resolution[0])
        self.resy = int(scene.camera.resolution[1])
        self.N = self.resx * self.resy
        create = self.L.vcm_create_sharded2 if isinstance(scene, SceneDesc2) else self.L.vcm_create_sharded
        self.ctx = create(C.byref(scene), algorithm, radius_factor, radius_alpha, seed, device, rank, world)
        if not self.ctx:
            raise RuntimeError("smallvcm_amd: vcm_create failed: %s" % self.L.vcm_last_error().decode())
        first, count = C.c_int(), C.c_int()
        self.L.vcm_local_path_range(self.ctx, C.byref(first), C.byref(count))
        self.first, self.count = first.value, count.value
        self.world = world
        self.device = device
        self._tstream = None
        self._max_len = 2

    def stream_context(self):
        """Run this context's kernels and the caller's torch ops (collectives,
        copies) on ONE non-default torch stream, so they are ordered without
        host synchronisation."""
        import torch
        if self._tstream is None:
            self._tstream = torch.cuda.Stream(device=self.device)
            self.set_stream(self._tstream.cuda_stream)
        return torch.cuda.stream(self._tstream)

    def close(self):
        if getattr(self, "ctx", None):
            self.L.vcm_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_strict_order(self, on):
        """True: DI / VC / merge inside the camera path as the reference does (slower, same bits)."""
        _check(self.L, self.L.vcm_set_strict_order(self.ctx, 1 if on else 0), "vcm_set_strict_order")

    MERGE_KERNELS = {"walk": 2, "pairs": 3}

    def set_merge_kernel(self, kind):
        """which kernel evaluates the range merges: 'walk' or 'pairs' (same bits; include/smallvcm_amd.h)"""
        _check(self.L, self.L.vcm_set_merge_kernel(self.ctx, self.MERGE_KERNELS[kind]), "vcm_set_merge_kernel")

    def set_stream(self, stream_handle):
        _check(self.L, self.L.vcm_set_stream(self.ctx, stream_handle), "vcm_set_stream")

    def reserve(self, max_len):
        """allocate now what the first iteration would allocate"""
        _check(self.L, self.L.vcm_reserve(self.ctx, max_len), "vcm_reserve")

    def begin(self, it, min_len, max_len):
        self._max_len = int(max_len)
        _check(self.L, self.L.vcm_begin_iteration(self.ctx, it, min_len, max_len), "vcm_begin_iteration")

    def trace_light(self):
        _check(self.L, self.L.vcm_trace_light(self.ctx), "vcm_trace_light")

    def build_grid(self):
        _check(self.L, self.L.vcm_build_grid(self.ctx), "vcm_build_grid")

    def trace_camera(self):
        _check(self.L, self.L.vcm_trace_camera(self.ctx), "vcm_trace_camera")

    def merge(self):
        _check(self.L, self.L.vcm_merge(self.ctx), "vcm_merge")

    @property
    def camera_before_grid(self):
        """The camera trace needs only the local light vertices, not the grid -- in wavefront mode, which the LIBRARY
        decides per iteration (strict order requested, or maxPathLength > 31: vcm_is_wavefront): a sharded renderer
        then runs it while the other ranks' vertices are in flight.  A single-rank renderer builds the grid first:
        the camera pass then also takes over the histogram of the query sort."""
        return self.world > 1 and bool(self.L.vcm_is_wavefront(self.ctx, self._max_len))

    def end(self):
        _check(self.L, self.L.vcm_end_iteration(self.ctx), "vcm_end_iteration")

    def run_iteration(self, it, min_len, max_len):
        _check(self.L, self.L.vcm_run_iteration(self.ctx, it, min_len, max_len), "vcm_run_iteration")

    def synchronize(self):
        _check(self.L, self.L.vcm_synchronize(self.ctx), "vcm_synchronize")

    # ---- exchange (multi-GPU) -------------------------------------------
    def local_record_count(self):
        ptr, n = C.c_void_p(), C.c_longlong()
        _check(self.L, self.L.vcm_light_records(self.ctx, C.byref(ptr), C.byref(n)), "vcm_light_records")
        return n.value

    def local_bbox(self):
        """(min3, max3, count) of this rank's light vertices (after trace_light); one stream synchronisation"""
        mn, mx, n = (C.c_float * 3)(), (C.c_float * 3)(), C.c_longlong()
        _check(self.L, self.L.vcm_local_light_bbox(self.ctx, mn, mx, C.byref(n)), "vcm_local_light_bbox")
        return [float(x) for x in mn], [float(x) for x in mx], n.value

    def set_grid_bbox(self, mn, mx):
        """the box of all ranks' light vertices, before trace_camera / build_grid"""
        _check(self.L, self.L.vcm_set_grid_bbox(self.ctx, (C.c_float * 3)(*mn), (C.c_float * 3)(*mx)), "vcm_set_grid_bbox")

    def export_records(self, dst_tensor, count):
        """copy `count` local records into a torch device tensor"""
        _check(self.L, self.L.vcm_export_light_records(self.ctx, dst_tensor.data_ptr(), count),
               "vcm_export_light_records")

    def import_records(self, gathered_tensor, counts, stride_records):
        arr = (C.c_longlong * len(counts))(*[int(c) for c in counts])
        _check(self.L, self.L.vcm_import_light_records(self.ctx, gathered_tensor.data_ptr(), arr, len(counts),
                                                       stride_records), "vcm_import_light_records")

    # the sorted exchange (include/smallvcm_amd.h): every rank sorts its own vertices by hash cell, the slabs travel,
    # build_grid merges them -- SMALLVCM_AMD_SORTED_EXCHANGE=0 keeps the unsorted exchange above
    def sorted_slab_words(self, stride_records):
        """4-byte words of one rank's slab, or -1 when this context has to use the unsorted exchange"""
        if os.environ.get("SMALLVCM_AMD_SORTED_EXCHANGE", "1") == "0":
            return -1
        return int(self.L.vcm_sorted_slab_words(self.ctx, int(stride_records)))

    def sort_records(self, dst_tensor, stride_records):
        _check(self.L, self.L.vcm_sort_light_records(self.ctx, dst_tensor.data_ptr(), int(stride_records)), "vcm_sort_light_records")

    def import_sorted_records(self, gathered_tensor, counts, stride_records):
        arr = (C.c_longlong * len(counts))(*[int(c) for c in counts])
        _check(self.L, self.L.vcm_import_sorted_light_records(self.ctx, gathered_tensor.data_ptr(), arr, len(counts),
                                                              int(stride_records)), "vcm_import_sorted_light_records")

    def export_framebuffer(self, dst_tensor):
        _check(self.L, self.L.vcm_export_framebuffer(self.ctx, dst_tensor.data_ptr()), "vcm_export_framebuffer")

    def new_tensor(self, n_floats):
        import torch
        return torch.empty(int(n_floats), dtype=torch.float32, device="cuda")

    # ---- read-back --------------------------------------------------------
    def framebuffer_sum(self):
        out = np.zeros((self.resy, self.resx, 3), np.float32)
        _check(self.L, self.L.vcm_read_framebuffer(self.ctx, out.ctypes.data_as(C.POINTER(C.c_float))),
               "vcm_read_framebuffer")
        return out

    def clear_framebuffer(self):
        _check(self.L, self.L.vcm_clear_framebuffer(self.ctx), "vcm_clear_framebuffer")

    def stats(self):
        s = Stats()
        _check(self.L, self.L.vcm_get_stats(self.ctx, C.byref(s)), "vcm_get_stats")
        return s.asdict()

    def read_image(self, fmt, scale, gamma=2.2):
        """the framebuffer * scale in the reference's BMP (fmt 0: BGR8, rows bottom-up) or HDR (fmt 1: RGBE) pixel
        encoding, converted on the device (Framebuffer::SaveBMP / SaveHDR, framebuffer.hxx:170-251)"""
        out = np.zeros((self.resy, self.resx, 3 if fmt == 0 else 4), np.uint8)
        _check(self.L, self.L.vcm_read_image(self.ctx, fmt, scale, gamma, out.ctypes.data_as(C.POINTER(C.c_ubyte))),
               "vcm_read_image")
        return out

    def stats_at(self, ago):
        """counters and phase times of the iteration `ago` iterations before the last completed one (<= 63)"""
        st = Stats()
        _check(self.L, self.L.vcm_get_stats_at(self.ctx, int(ago), C.byref(st)), "vcm_get_stats_at")
        return st.asdict()

    def rng_counts(self):
        a = np.zeros(self.count, np.uint8)
        b = np.zeros(self.count, np.uint8)
        _check(self.L, self.L.vcm_get_rng_counts(self.ctx, a.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                                 b.ctypes.data_as(C.POINTER(C.c_ubyte))), "vcm_get_rng_counts")
        return a, b

    def records(self):
        n = self.local_record_count()
        out = np.zeros((n, VCM_MERGE_RECORD_FLOATS), np.float32)
        _check(self.L, self.L.vcm_debug_read_records(self.ctx, out.ctypes.data_as(C.POINTER(C.c_float)), n),
               "vcm_debug_read_records")
        return out

    def grid(self):
        cs = np.zeros(self.N + 1, np.int32)
        nrec = C.c_longlong()
        bbox = np.zeros(6, np.float32)
        _check(self.L, self.L.vcm_debug_read_grid(self.ctx, None, None, None, C.byref(nrec)), "vcm_debug_read_grid")
        idx = np.zeros(max(nrec.value, 1), np.int32)
        _check(self.L, self.L.vcm_debug_read_grid(self.ctx, cs.ctypes.data_as(C.POINTER(C.c_int)),
                                                  idx.ctypes.data_as(C.POINTER(C.c_int)),
                                                  bbox.ctypes.data_as(C.POINTER(C.c_float)), C.byref(nrec)),
               "vcm_debug_read_grid")
        return cs, idx[:nrec.value], bbox


class VertexCM:
    """Single-GPU renderer with the reference's interface (src/vertexcm.hxx:61,
    src/renderer.hxx:33-70)."""

    # AlgorithmType, src/vertexcm.hxx:182-204
    kLightTrace, kPpm, kBpm, kBpt, kVcm = ALGO_LIGHT_TRACE, ALGO_PPM, ALGO_BPM, ALGO_BPT, ALGO_VCM

    def __init__(self, aScene, aAlgorithm, aRadiusFactor, aRadiusAlpha, aSeed=1234, device=0, strict_order=None):
        self.mScene = aScene
        self.mMaxPathLength = 2      # renderer.hxx:40
        self.mMinPathLength = 0      # renderer.hxx:39
        self.backend = HipBackend(aScene, aAlgorithm, aRadiusFactor, aRadiusAlpha, aSeed, device=device)
        if strict_order is not None:
            self.backend.set_strict_order(strict_order)
        self.mIterations = 0

    def RunIteration(self, aIteration):
        self.backend.run_iteration(aIteration, self.mMinPathLength, self.mMaxPathLength)
        self.mIterations += 1

    def WasUsed(self):
        return self.mIterations > 0

    def GetFramebuffer(self):
        """renderer.hxx:49-55: the running sum scaled by 1/mIterations."""
        fb = self.backend.framebuffer_sum()
        if self.mIterations > 0:
            fb = fb * np.float32(1.0 / self.mIterations)
        return fb

    def framebuffer_sum(self):
        return self.backend.framebuffer_sum()

    def stats(self):
        return self.backend.stats()

    def close(self):
        self.backend.close()


class ShardedVertexCM:
    """One rank of a renderer sharded over `world` processes (one GPU each).

    Rank r owns light paths and pixels [r*N/W, (r+1)*N/W).  Per iteration:
      trace_light (local) -> all-gather of the 52-byte merge records (ragged:
      counts first, then one all_gather_into_tensor on max-padded slabs) ->
      identical hash grid on every rank -> trace_camera (local pixels).
    Light splats land on arbitrary pixels, so every rank keeps a full
    framebuffer; they are summed once at read-out (all_reduce).

    `backend` is anything with HipBackend's phase interface -- the CPU tests
    drive this same class with an oracle-backed backend over gloo.
    """

    def __init__(self, backend, rank, world, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.backend = backend
        self.rank, self.world, self.group = rank, world, group
        self.mMaxPathLength = 2
        self.mMinPathLength = 0
        self.mIterations = 0
        self._gather = None
        self._local = None
        self._pending = None

    def RunIteration(self, aIteration):
        self.start_iteration(aIteration)
        self.finish_iteration()

    # An iteration in two host-side halves, so that a scheduler (RenderFarm) can keep TWO renderers in flight on
    # the same GPUs: everything here only enqueues work (kernels on the context's stream, the all-gather on RCCL's),
    # so while one renderer's records cross xGMI the GPU runs the other renderer's kernels.
    def start_iteration(self, aIteration):
        """light pass, start of the exchange, and the part of the camera pass that does not need the grid"""
        with self.backend.stream_context():
            self._start(aIteration)

    def finish_iteration(self):
        """wait for the exchange, grid build, merge, resolve"""
        with self.backend.stream_context():
            self._finish()
        self.mIterations += 1

    def _start(self, aIteration):
        import torch
        b, dist = self.backend, self.dist
        b.begin(aIteration, self.mMinPathLength, self.mMaxPathLength)
        b.trace_light()
        work = counts = stride = gathered = None
        if self.world > 1:
            # 1) counts -- and, if the backend can use it, the bounding box of the local vertices: 7 numbers per rank
            #    in one tiny all-gather.  min / max over the ranks' boxes is the box HashGrid::Build takes over the
            #    whole array (hashgrid.hxx:50-61); knowing it now lets the camera pass below prepare the query sort.
            dev = b.new_tensor(1).device
            boxed = hasattr(b, "local_bbox")
            if boxed:
                mn, mx, n_local = b.local_bbox()
            else:
                mn, mx, n_local = [0.0] * 3, [0.0] * 3, b.local_record_count()
            mine = torch.tensor([float(n_local)] + mn + mx, dtype=torch.float64, device=dev)   # fp32 values, exact
            every = torch.empty(7 * self.world, dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(every, mine, group=self.group)
            rows = every.cpu().reshape(self.world, 7).numpy()
            counts = [int(x) for x in rows[:, 0]]
            stride = max(max(counts), 1)
            if boxed:
                full = rows[rows[:, 0] > 0]
                if len(full):
                    b.set_grid_bbox([float(x) for x in full[:, 1:4].min(axis=0)], [float(x) for x in full[:, 4:7].max(axis=0)])
                else:
                    b.set_grid_bbox([1e36] * 3, [-1e36] * 3)
            # 2) records, padded to the largest shard; asynchronous: the camera
            #    trace below does not need the other ranks' vertices
            #    (sorted exchange, when the backend offers it: the rank's records in CELL order + block starts, one slab)
            words = b.sorted_slab_words(stride) if (boxed and hasattr(b, "sorted_slab_words")) else -1
            sorted_x = words > 0
            need = words if sorted_x else stride * VCM_MERGE_RECORD_FLOATS
            if self._local is None or self._local.numel() < need:
                self._local = b.new_tensor(need)
                self._gather = b.new_tensor(need * self.world)
            local = self._local[:need]
            gathered = self._gather[:need * self.world]
            if sorted_x:
                b.sort_records(local, stride)
            else:
                b.export_records(local, n_local)
            overlap = getattr(b, "camera_before_grid", False)
            work = dist.all_gather_into_tensor(gathered, local, group=self.group, async_op=overlap)
        early = getattr(b, "camera_before_grid", False)
        if early:
            b.trace_camera()            # overlaps the all-gather
        self._pending = (work, counts, stride, gathered, early, self.world > 1 and sorted_x)

    def _finish(self):
        b = self.backend
        work, counts, stride, gathered, early, sorted_x = self._pending
        self._pending = None
        if work is not None and early:
            work.wait()
        if self.world > 1:
            if sorted_x:
                b.import_sorted_records(gathered, counts, stride)
            else:
                b.import_records(gathered, counts, stride)
        b.build_grid()
        if not early:
            b.trace_camera()
        b.merge()
        b.end()

    def WasUsed(self):
        return self.mIterations > 0

    def framebuffer_sum(self):
        """Sum over ranks of the running-sum framebuffers (valid on every rank)."""
        b, dist = self.backend, self.dist
        with b.stream_context():
            t = b.new_tensor(b.N * 3)
            b.export_framebuffer(t)
            if self.world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            out = t.cpu()
        return out.numpy().reshape(b.resy, b.resx, 3)

    def GetFramebuffer(self):
        fb = self.framebuffer_sum()
        if self.mIterations > 0:
            fb = fb * np.float32(1.0 / self.mIterations)
        return fb


def static_schedule(n_iterations, n_threads, tid):
    """The iterations OpenMP's default `schedule(static)` gives thread `tid` of the reference's
    `#pragma omp parallel for` (src/smallvcm.cxx:98-108): contiguous blocks, the first
    n_iterations % n_threads threads get one more."""
    q, r = divmod(int(n_iterations), int(n_threads))
    lo = tid * q + min(tid, r)
    return range(lo, lo + q + (1 if tid < r else 0))


class RenderFarm:
    """The reference's render() (src/smallvcm.cxx:52-151) over the GPUs of one node.

    The reference creates one renderer per host thread, seed mBaseSeed + i (:66-72), hands each a
    block of iterations (:98-108), and averages the used renderers' framebuffers (:116-142).  Here a
    "thread" is a renderer that lives on a GROUP of `shards` ranks: the world of W ranks (one per GPU) is
    cut into W / shards groups; a renderer's ranks split the paths of every iteration (RCCL all-gather of
    the light vertices inside the group).  shards = W is one renderer across all GPUs (minimum latency per
    iteration), shards = 1 is one renderer per GPU (no per-iteration communication); the image is the same
    estimator either way.  The only world-wide collective is the framebuffer reduce at read-out.

    `inflight` renderers share each group and take turns (default 2 when shards > 1): every call of an
    iteration half only ENQUEUES work, so while one renderer's 464 MB exchange crosses xGMI the GPUs run the
    other renderer's kernels -- the exchange is hidden without any second buffer set inside a renderer.
    Each in-flight renderer has its own process group (its own RCCL communicator): collectives of one
    communicator execute in order, and the small count exchange of one renderer must not queue behind the
    large all-gather of the other.  Renderer index = group * inflight + slot; R = (W / shards) * inflight.

    backend_factory(seed, shard_rank, shard_world) -> backend with HipBackend's phase interface.
    """

    def __init__(self, backend_factory, base_seed, rank, world, shards=None, dist=None, inflight=None):
        if dist is None:
            import torch.distributed as dist
        shards = world if shards is None else int(shards)
        if shards < 1 or world % shards:
            raise ValueError("world size %d is not a multiple of %d shards" % (world, shards))
        inflight = (2 if shards > 1 else 1) if inflight is None else int(inflight)
        if inflight < 1:
            raise ValueError("inflight must be >= 1")
        self.dist, self.rank, self.world, self.shards, self.inflight = dist, rank, world, shards, inflight
        self.groups = world // shards
        self.replicas = self.groups * inflight          # renderers = the reference's "threads"
        self.group_index, self.shard = rank // shards, rank % shards
        self.renderers, self.renderer_ids = [], []
        for g in range(self.groups):
            for k in range(inflight):
                grp = None
                if world > 1 and shards > 1 and (shards < world or inflight > 1):
                    # every rank creates every group, in the same order (torch.distributed contract)
                    grp = dist.new_group(ranks=list(range(g * shards, (g + 1) * shards)))
                if g != self.group_index:
                    continue
                rid = g * inflight + k
                backend = backend_factory(base_seed + rid, self.shard, shards)   # smallvcm.cxx:68
                r = ShardedVertexCM(backend, self.shard, shards, group=grp)
                r.dist = dist
                self.renderers.append(r)
                self.renderer_ids.append(rid)
        # first local renderer: what single-renderer callers (bench.py's per-kernel statistics) look at
        self.renderer, self.backend = self.renderers[0], self.renderers[0].backend
        self.replica = self.renderer_ids[0]

    def set_path_lengths(self, min_len, max_len):
        for r in self.renderers:
            r.mMinPathLength, r.mMaxPathLength = min_len, max_len   # smallvcm.cxx:70-71

    def run_iterations(self, per_renderer):
        """per_renderer[k] = the iteration indices of this rank's k-th renderer, in order.  Round-robin over the
        in-flight renderers, one iteration half at a time (start all, then finish all)."""
        for t in range(max(len(x) for x in per_renderer)):
            live = [(r, its[t]) for r, its in zip(self.renderers, per_renderer) if t < len(its)]
            for r, it in live:
                r.start_iteration(it)
            for r, _ in live:
                r.finish_iteration()

    def render(self, n_iterations):
        """smallvcm.cxx:98-108 (iterations based loop)"""
        self.run_iterations([list(static_schedule(n_iterations, self.replicas, rid)) for rid in self.renderer_ids])

    def framebuffer(self):
        """smallvcm.cxx:116-142: mean over the used renderers of (running sum / own iterations).
        One all_reduce over the whole world: the shards of a renderer hold partial sums of it."""
        b0, dist = self.backend, self.dist
        with b0.stream_context():
            t = b0.new_tensor(b0.N * 3)
            t.zero_()
            n_used_local = 0.0
            for r in self.renderers:
                if not r.WasUsed():
                    continue
                with r.backend.stream_context():
                    part = r.backend.new_tensor(b0.N * 3)
                    r.backend.export_framebuffer(part)
                    part *= 1.0 / r.mIterations                                         # renderer.hxx:53-54
                    r.backend.synchronize()
                t += part
                if self.shard == 0:
                    n_used_local += 1.0
            used = t.new_tensor([n_used_local])
            if self.world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                dist.all_reduce(used, op=dist.ReduceOp.SUM)
            n_used = max(float(used.cpu()[0]), 1.0)
            t *= 1.0 / n_used                                                           # smallvcm.cxx:142
            out = t.cpu()
        return out.numpy().reshape(b0.resy, b0.resx, 3)

    def close(self):
        for r in self.renderers:
            r.backend.close()
