"""ctypes binding of the C++ multi-GPU host (include/smallvcm_amd_farm.h, smallvcm_amd/host/libsmallvcm_amd_farm.so):
the reference's render() (src/smallvcm.cxx:52-151) over the GPUs of one node -- one host thread per GPU, RCCL between
them.  Python only fills the configuration and reads the result; nothing of an iteration runs here.

  single process, N GPUs    farm_render(scene, ..., ranks=N)                               (`python bench.py --gpus N`)
  one process per GPU       rank 0: ids = unique_ids(1 + N // shards); ship them (any channel: bench.py uses a gloo
                            broadcast); every rank: farm_render(..., ranks=N, first_rank=r, devices=[local_rank],
                            unique_ids=ids)                                                (torch.distributed.run)
"""
import ctypes as C
import os

import numpy as np

from ._abi import SceneDesc, Stats

_HERE = os.path.dirname(os.path.abspath(__file__))
FARM_LIB_PATH = os.environ.get("SMALLVCM_AMD_FARM_LIB") or os.path.join(_HERE, "host", "libsmallvcm_amd_farm.so")
MAX_RANKS = 64


class FarmConfig(C.Structure):
    _fields_ = [("scene", SceneDesc), ("algorithm", C.c_int), ("radiusFactor", C.c_float), ("radiusAlpha", C.c_float),
                ("baseSeed", C.c_int), ("minLen", C.c_uint), ("maxLen", C.c_uint), ("iterations", C.c_int), ("warmup", C.c_int),
                ("sameWindow", C.c_int), ("ranks", C.c_int), ("firstRank", C.c_int), ("localRanks", C.c_int),
                ("devices", C.c_int * MAX_RANKS), ("shards", C.c_int), ("inflight", C.c_int), ("collectives", C.c_int),
                ("uniqueIds", C.c_void_p), ("nUniqueIds", C.c_int)]


class FarmResult(C.Structure):
    _fields_ = [("wallSeconds", C.c_double), ("renderers", C.c_int), ("rcclRanks", C.c_int),
                ("rankIterationMs", C.c_float * MAX_RANKS), ("meanStats", Stats)]


_lib = None


def load_farm_library():
    """Raises if the library is missing: there is no other multi-GPU compute path."""
    global _lib
    if _lib is None:
        if not os.path.exists(FARM_LIB_PATH):
            raise RuntimeError("smallvcm_amd: %s not built (python -c 'import __graft_entry__ as g; g.build()')" % FARM_LIB_PATH)
        from .renderer import load_library
        load_library(require_gpu=False)   # libsmallvcm_amd.so (and, before it, torch's HIP runtime if torch is installed)
        L = C.CDLL(FARM_LIB_PATH)
        L.vcm_farm_render.argtypes = [C.POINTER(FarmConfig), C.POINTER(FarmResult), C.POINTER(C.c_float)]
        L.vcm_farm_last_error.restype = C.c_char_p
        L.vcm_farm_unique_ids.argtypes = [C.c_void_p, C.c_int]
        _lib = L
    return _lib


def unique_ids(n):
    """n RCCL unique ids (bytes), made by the process of world rank 0 and shipped to every process by the launcher"""
    L = load_farm_library()
    buf = C.create_string_buffer(n * L.vcm_farm_unique_id_bytes())
    if L.vcm_farm_unique_ids(buf, n) != 0:
        raise RuntimeError("smallvcm_amd: %s" % L.vcm_farm_last_error().decode())
    return buf.raw


def farm_render(scene, algorithm, iterations, ranks, shards, inflight, devices=None, first_rank=0, warmup=0, same_window=False,
                collectives="rccl", ids=None, radius_factor=0.003, radius_alpha=0.75, seed=1234, min_len=0, max_len=10,
                want_image=True):
    """-> dict(wall_s, renderers, rccl_ranks, rank_iteration_ms, stats, image).  `devices`: the HIP device of each rank
    hosted by this process (default: all `ranks` ranks, device = rank)."""
    L = load_farm_library()
    devices = list(range(ranks)) if devices is None else list(devices)
    cfg = FarmConfig()
    cfg.scene = scene
    cfg.algorithm, cfg.radiusFactor, cfg.radiusAlpha, cfg.baseSeed = algorithm, radius_factor, radius_alpha, seed
    cfg.minLen, cfg.maxLen, cfg.iterations, cfg.warmup, cfg.sameWindow = min_len, max_len, iterations, warmup, 1 if same_window else 0
    cfg.ranks, cfg.firstRank, cfg.localRanks = ranks, first_rank, len(devices)
    for i, d in enumerate(devices):
        cfg.devices[i] = d
    cfg.shards, cfg.inflight, cfg.collectives = shards, inflight, 0 if collectives == "rccl" else 1
    keep = None
    if ids is not None:
        keep = C.create_string_buffer(ids, len(ids))
        cfg.uniqueIds = C.cast(keep, C.c_void_p)
        cfg.nUniqueIds = len(ids) // L.vcm_farm_unique_id_bytes()
    res = FarmResult()
    resx, resy = int(scene.camera.resolution[0]), int(scene.camera.resolution[1])
    img = np.zeros((resy, resx, 3), np.float32) if (want_image and first_rank == 0) else None
    rc = L.vcm_farm_render(C.byref(cfg), C.byref(res), img.ctypes.data_as(C.POINTER(C.c_float)) if img is not None else None)
    if rc != 0:
        raise RuntimeError("smallvcm_amd: vcm_farm_render failed: %s" % L.vcm_farm_last_error().decode())
    return {"wall_s": res.wallSeconds, "renderers": res.renderers, "rccl_ranks": res.rcclRanks,
            "rank_iteration_ms": [float(res.rankIterationMs[r]) for r in range(ranks)], "stats": res.meanStats.asdict(), "image": img}
