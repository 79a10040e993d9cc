"""ctypes bindings of the TEST oracle (oracle/liboracle.so) and of the
unmodified reference harness (oracle/_ref/*.so).  Test infrastructure only."""
import ctypes as C
import os
import numpy as np

from smallvcm_amd._abi import SceneDesc, SceneDesc2, Stats, VCM_MERGE_RECORD_FLOATS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_TAPE_SO = os.path.join(ROOT, "oracle", "_ref", "libsmallvcm_ref_tape.so")
REF_STOCK_SO = os.path.join(ROOT, "oracle", "_ref", "libsmallvcm_ref_stock.so")
REF_TAPE_LIBM_SO = os.path.join(ROOT, "oracle", "_ref", "libsmallvcm_ref_tape_libm.so")   # tape replay, the image's own libm

_u8p = C.POINTER(C.c_ubyte)
_fp = C.POINTER(C.c_float)


def _fptr(a):
    return a.ctypes.data_as(_fp)


def _bptr(a):
    return a.ctypes.data_as(_u8p)


_oracle = None


def load_oracle(path):
    """a build of oracle/vcm_oracle.cpp with its entry points declared (liboracle.so = the checker; liboracle_glibc.so /
    liboracle_cr.so = the same restatement over another sin / cos / pow, for oracle/libm_tolerance.py only)"""
    L = C.CDLL(path)
    _declare(L)
    return L


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = load_oracle(ORACLE_SO)
    return _oracle


def _declare(L):
    if True:
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
        L.oracle_create2.restype = C.c_void_p
        L.oracle_create2.argtypes = [C.POINTER(SceneDesc2), C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
        for n in ("oracle_destroy", "oracle_trace_light", "oracle_build_grid", "oracle_trace_camera",
                  "oracle_end_iteration"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = None
        L.oracle_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.oracle_begin_iteration.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_uint]
        L.oracle_run_iteration.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_uint]
        L.oracle_trace_camera_rows.argtypes = [C.c_void_p, C.c_int]
        L.oracle_trace_camera_window.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_light_record_count.argtypes = [C.c_void_p]
        L.oracle_light_record_count.restype = C.c_longlong
        L.oracle_export_light_records.argtypes = [C.c_void_p, _fp]
        L.oracle_import_light_records.argtypes = [C.c_void_p, _fp, C.c_longlong]
        L.oracle_get_framebuffer.argtypes = [C.c_void_p, _fp]
        L.oracle_add_framebuffer.argtypes = [C.c_void_p, _fp]
        L.oracle_get_counts.argtypes = [C.c_void_p, _u8p, _u8p]
        L.oracle_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.oracle_iterations.argtypes = [C.c_void_p]
        L.oracle_local_range.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_grid_cells.argtypes = [C.c_void_p]
        L.oracle_grid_cells.restype = C.c_longlong
        L.oracle_get_grid.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _fp]
        for n in ("oracle_sinf", "oracle_cosf"):
            getattr(L, n).argtypes = [C.c_float]
            getattr(L, n).restype = C.c_float
        L.oracle_powf.argtypes = [C.c_float, C.c_float]
        L.oracle_powf.restype = C.c_float
        L.oracle_philox.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.oracle_path_float.argtypes = [C.c_uint32] * 5
        L.oracle_path_float.restype = C.c_float
        L.oracle_max_threads.restype = C.c_int


class Oracle:
    """CPU restatement of VertexCM (oracle/vcm_oracle.cpp)."""

    def __init__(self, scene, algo, radius_factor=0.003, radius_alpha=0.75, seed=1234, rank=0, world=1, threads=1, lib=None):
        self.L = lib if lib is not None else oracle()
        self.scene = scene
        create = self.L.oracle_create2 if isinstance(scene, SceneDesc2) else self.L.oracle_create
        self.h = create(C.byref(scene), algo, radius_factor, radius_alpha, seed, rank, world)
        self.resx = int(scene.camera.resolution[0])
        self.resy = int(scene.camera.resolution[1])
        self.N = self.resx * self.resy
        first, count = C.c_int(), C.c_int()
        self.L.oracle_local_range(self.h, C.byref(first), C.byref(count))
        self.first, self.count = first.value, count.value
        self.L.oracle_set_threads(self.h, threads)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_destroy(self.h)
            self.h = None

    def run_iteration(self, it, min_len=0, max_len=10):
        self.L.oracle_run_iteration(self.h, it, min_len, max_len)

    def begin(self, it, min_len=0, max_len=10):
        self.L.oracle_begin_iteration(self.h, it, min_len, max_len)

    def trace_light(self):
        self.L.oracle_trace_light(self.h)

    def build_grid(self):
        self.L.oracle_build_grid(self.h)

    def trace_camera(self, row_stride=1, row_width=1):
        """camera paths of the pixel rows y with y % row_stride < row_width"""
        self.L.oracle_trace_camera_window(self.h, row_stride, row_width)

    def end(self):
        self.L.oracle_end_iteration(self.h)

    def records(self):
        n = self.L.oracle_light_record_count(self.h)
        out = np.zeros((n, VCM_MERGE_RECORD_FLOATS), np.float32)
        if n:
            self.L.oracle_export_light_records(self.h, _fptr(out))
        return out

    def import_records(self, recs):
        recs = np.ascontiguousarray(recs, np.float32)
        self.L.oracle_import_light_records(self.h, _fptr(recs), recs.shape[0])

    def framebuffer(self):
        out = np.zeros((self.resy, self.resx, 3), np.float32)
        self.L.oracle_get_framebuffer(self.h, _fptr(out))
        return out

    def add_framebuffer(self, fb):
        fb = np.ascontiguousarray(fb, np.float32)
        self.L.oracle_add_framebuffer(self.h, _fptr(fb))

    def counts(self):
        a = np.zeros(self.count, np.uint8)
        b = np.zeros(self.count, np.uint8)
        self.L.oracle_get_counts(self.h, _bptr(a), _bptr(b))
        return a, b

    def stats(self):
        s = Stats()
        self.L.oracle_get_stats(self.h, C.byref(s))
        return s.asdict()

    def grid(self):
        n = self.L.oracle_grid_cells(self.h)
        nrec = self.L.oracle_light_record_count(self.h)
        ce = np.zeros(n, np.int32)
        idx = np.zeros(nrec, np.int32)
        bbox = np.zeros(6, np.float32)
        self.L.oracle_get_grid(self.h, ce.ctypes.data_as(C.POINTER(C.c_int)),
                               idx.ctypes.data_as(C.POINTER(C.c_int)), _fptr(bbox))
        return ce, idx, bbox


_ref_tape = None
_ref_tape_libm = None
_ref_stock = None


def have_ref():
    return os.path.exists(REF_TAPE_SO) and os.path.exists(REF_STOCK_SO)


def _declare_tape(L):
    L.ref_flatten_scene.argtypes = [C.c_uint, C.c_int, C.c_int, C.POINTER(SceneDesc)]
    L.ref_detmath_calls.restype = C.c_longlong
    L.ref_run_tape.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                               C.c_int, C.c_int, C.c_uint, C.c_uint, _u8p, _u8p, _fp,
                               C.POINTER(C.c_longlong)]
    return L


def ref_tape_libm():
    """the tape-replay reference linked against the image's OWN libm (no interposition): what is left between the product's
    definition of sinf / cosf / powf and the reference as shipped"""
    global _ref_tape_libm
    if _ref_tape_libm is None:
        _ref_tape_libm = _declare_tape(C.CDLL(REF_TAPE_LIBM_SO))
    return _ref_tape_libm


def ref_tape():
    global _ref_tape
    if _ref_tape is None:
        L = C.CDLL(REF_TAPE_SO)
        L.ref_flatten_scene.argtypes = [C.c_uint, C.c_int, C.c_int, C.POINTER(SceneDesc)]
        L.ref_scene_config_mask.argtypes = [C.c_int]
        L.ref_scene_config_mask.restype = C.c_uint
        L.ref_detmath_calls.restype = C.c_longlong
        L.ref_run_tape.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                   C.c_int, C.c_int, C.c_uint, C.c_uint, _u8p, _u8p, _fp,
                                   C.POINTER(C.c_longlong)]
        _ref_tape = L
    return _ref_tape


def ref_stock():
    global _ref_stock
    if _ref_stock is None:
        L = C.CDLL(REF_STOCK_SO)
        L.ref_flatten_scene.argtypes = [C.c_uint, C.c_int, C.c_int, C.POINTER(SceneDesc)]
        L.ref_render_stock.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_uint, C.c_uint, C.c_float, C.c_float, _fp, C.POINTER(C.c_double)]
        L.ref_render_stock_iters.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int,
                                             C.c_int, C.c_uint, C.c_uint, C.c_float, C.c_float, _fp, C.POINTER(C.c_double)]
        _ref_stock = L
    return _ref_stock


def ref_scene(mask, resx, resy):
    """Scene desc flattened from the REFERENCE's own Scene (needs oracle/_ref)."""
    d = SceneDesc()
    rc = ref_tape().ref_flatten_scene(mask, resx, resy, C.byref(d))
    assert rc == 0, rc
    return d


def ref_run_tape(mask, resx, resy, algo, light_counts, cam_counts, radius_factor=0.003, radius_alpha=0.75,
                 seed=1234, first_iteration=0, n_iter=1, min_len=0, max_len=10, own_libm=False):
    """Unmodified reference VertexCM, replaying the taped random numbers.
    Returns (framebuffer sum, floats consumed, desync flag).  own_libm: the build whose sinf / cosf / powf are the
    image's libm instead of detmath_ref.h."""
    lc = np.ascontiguousarray(light_counts, np.uint8)
    cc = np.ascontiguousarray(cam_counts, np.uint8)
    assert lc.size == n_iter * resx * resy and cc.size == lc.size
    fb = np.zeros((resy, resx, 3), np.float32)
    consumed = C.c_longlong()
    L = ref_tape_libm() if own_libm else ref_tape()
    bad = L.ref_run_tape(mask, resx, resy, algo, radius_factor, radius_alpha, seed, first_iteration, n_iter,
                         min_len, max_len, _bptr(lc), _bptr(cc), _fptr(fb), C.byref(consumed))
    return fb, consumed.value, bad


def ref_run_tape2(scene2, algo, light_counts, cam_counts, radius_factor=0.003, radius_alpha=0.75, seed=1234,
                  first_iteration=0, n_iter=1, min_len=0, max_len=10):
    """ref_run_tape for a version-2 scene: the harness builds a reference Scene from the arrays with the reference's own
    constructors and refuses (rc <= -100) if a derived member of the description is not what they produce."""
    L = ref_tape()
    L.ref_run_tape2.argtypes = [C.POINTER(SceneDesc2), C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint,
                                _u8p, _u8p, _fp, C.POINTER(C.c_longlong)]
    resx, resy = int(scene2.camera.resolution[0]), int(scene2.camera.resolution[1])
    lc = np.ascontiguousarray(light_counts, np.uint8)
    cc = np.ascontiguousarray(cam_counts, np.uint8)
    assert lc.size == n_iter * resx * resy and cc.size == lc.size
    fb = np.zeros((resy, resx, 3), np.float32)
    consumed = C.c_longlong()
    bad = L.ref_run_tape2(C.byref(scene2), algo, radius_factor, radius_alpha, seed, first_iteration, n_iter, min_len, max_len,
                          _bptr(lc), _bptr(cc), _fptr(fb), C.byref(consumed))
    return fb, consumed.value, bad


def ref_check_scene2(scene2):
    L = ref_tape()
    L.ref_check_scene2.argtypes = [C.POINTER(SceneDesc2)]
    return L.ref_check_scene2(C.byref(scene2))


def ref_render_stock(mask, resx, resy, algo, iterations=1, threads=1, seed=1234, min_len=0, max_len=10,
                     radius_factor=0.003, radius_alpha=0.75, config_algo=-1, iteration_index=None):
    """Unmodified reference (mt19937 + glibc), render() semantics. Returns (fb averaged, wall seconds).
    iteration_index: optional list, the aIteration passed for loop index i (sets the radius only)."""
    fb = np.zeros((resy, resx, 3), np.float32)
    wall = C.c_double()
    idx = None
    if iteration_index is not None:
        assert len(iteration_index) == iterations
        idx = (C.c_int * iterations)(*[int(i) for i in iteration_index])
    ref_stock().ref_render_stock_iters(mask, resx, resy, config_algo, algo, iterations, idx, threads, seed, min_len, max_len,
                                       radius_factor, radius_alpha, _fptr(fb), C.byref(wall))
    return fb, wall.value


def host_libm_is_the_restated_one():
    """detmath restates ONE libm bit for bit: glibc 2.35's sinf / cosf / powf in their FMA multiarch variants (x86-64 with
    FMA3: what the reference's image runs on).  Comparisons against the AMBIENT libm.so.6 are only meaningful there; on
    another glibc, or a CPU without FMA (glibc's ifunc then picks the non-fused variant), they would fail with no product bug
     -- the tests that call libm.so.6 directly ask here first and skip that part otherwise."""
    try:
        libc = C.CDLL("libc.so.6")
        libc.gnu_get_libc_version.restype = C.c_char_p
        if libc.gnu_get_libc_version().decode() != "2.35":
            return False
        flags = ""
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = line
                break
        return " fma " in (flags + " ") or flags.rstrip().endswith(" fma")
    except Exception:
        return False
