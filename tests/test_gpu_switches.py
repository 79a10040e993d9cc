"""Every environment switch of the library that selects another launch shape, stream layout, kernel or filter is a
shipped code path in front of RangeQuery::Process / Scene::Intersect (src/vertexcm.hxx:130-169, src/hashgrid.hxx:110-169,
src/scene.hxx:53-85): each must render the frames of the default build, bit for bit, and count the same work.  The
switches are read once per process, so every case is a process of its own -- the C++ host (no Python, no torch).  GPU only."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle_lib import Oracle
from smallvcm_amd.renderer import cornell_scene

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "smallvcm_amd", "host", "vcm_render")

# (scene, algorithm name, oracle algorithm id, resolution, iterations)
CONFIGS = [(1, "vcm", 4, 256, 2), (3, "bpm", 2, 192, 2)]
SWITCHES = [
    "SMALLVCM_AMD_MERGE_DEAL=slab",            # k_merge_walk: one slab of the sorted queries per XCD, with stealing
    "SMALLVCM_AMD_MERGE_DEAL=slab SMALLVCM_AMD_MERGE_SLAB_BLOCKS=64",
    "SMALLVCM_AMD_MERGE=walk",                 # k_merge_walk (the default is k_merge_pairs)
    "SMALLVCM_AMD_MERGE_CHUNK=3",
    "SMALLVCM_AMD_MERGE_BLOCKS=64",
    "SMALLVCM_AMD_NO_RECTS=1",                 # the Pluecker filter (SceneQuads kernels) instead of the rectangles
    "SMALLVCM_AMD_NO_RECTS=1 SMALLVCM_AMD_NO_ONEPLANE=1",   # ... and the general list (SceneList kernels)
    "SMALLVCM_AMD_FORCE_BVH=1",                # the boxes through the BVH kernels
    "SMALLVCM_AMD_GENERAL_POW=1",              # the kernels that keep the general powf (non-integer Phong exponents)
    "SMALLVCM_AMD_GENERAL_POW=1 SMALLVCM_AMD_FORCE_BVH=1",   # SceneBvhG
    "SMALLVCM_AMD_NO_SIDE=1",                  # grid build in line
    "SMALLVCM_AMD_STREAM_PRIO=1",
    "SMALLVCM_AMD_SPLAT_STREAM=0",             # light splats in line
    "SMALLVCM_AMD_VC_STREAM=1",
    "SMALLVCM_AMD_VC_STREAM=0",
    "SMALLVCM_AMD_NO_K1_BBOX=1",               # k_bbox instead of the box K1 keeps
    "SMALLVCM_AMD_TRACE_WAVES=512 SMALLVCM_AMD_TRACE_CHUNK=64",
    "SMALLVCM_AMD_LIGHT_WAVES=1024",
    "SMALLVCM_AMD_TASK_BLOCKS=96",
    "SMALLVCM_AMD_SPLAT_LONG=8",
    "SMALLVCM_AMD_NO_STAMPS=1",
    "SMALLVCM_AMD_TIMING=events",
    "SMALLVCM_AMD_STRICT_ORDER=1 SMALLVCM_AMD_ARENAS=1",
    # round 5
    "SMALLVCM_AMD_SORT_INLINE=0",              # the query sort's scan + scatter on the side stream whatever the frame size
    "SMALLVCM_AMD_SORT_INLINE=1",
    "SMALLVCM_AMD_BUCKETS_PER_PATH=16",
    "SMALLVCM_AMD_BUCKETS_PER_PATH=1",
    "SMALLVCM_AMD_AUX_BLOCKS=64",
    "SMALLVCM_AMD_MERGE_BLOCKS=16384 SMALLVCM_AMD_TASK_BLOCKS=3072 SMALLVCM_AMD_TRACE_WAVES=4096",   # the launch shapes of a 2048^2 frame
    "SMALLVCM_AMD_MERGE_ASIDE=0",              # K4 on the main stream (rounds 1-4) instead of on the side stream beside the next K1
    "SMALLVCM_AMD_MERGE_ASIDE=1",              # ... forced (the default decides by frame size and algorithm)
    "SMALLVCM_AMD_RESOLVE_ASIDE=0",            # K5 in line on the main stream (rounds 1-4) instead of on the splat stream beside the next K1
    "SMALLVCM_AMD_SPLATS_AFTER_K3=1",          # measurement: K1c / K1d start when K3 has ended
    "SMALLVCM_AMD_HELPER_CUS=4",               # measurement: the helper streams on half of the CUs (hipExtStreamCreateWithCUMask)
    "SMALLVCM_AMD_GRID_SORT=count",            # HashGrid::Build as rounds 1-5 did it: a counter per cell, one atomic per vertex, a ranking pass
    "SMALLVCM_AMD_GRID_SORT_BLOCKS=1",         # the radix sort with ONE workgroup: ~70 000 vertices = 35 tiles in a row
    "SMALLVCM_AMD_GRID_SORT_BLOCKS=7",         # chunks that are not a multiple of the tile
    "SMALLVCM_AMD_GRID_SORT_BLOCKS=4096",      # more workgroups than 256-vertex chunks: most of them have nothing to do
    "SMALLVCM_AMD_GRID_SORT=count SMALLVCM_AMD_SORT_INLINE=0 SMALLVCM_AMD_RESOLVE_BLOCKS=2048",   # the launch plan of the round-4 build
    "SMALLVCM_AMD_RESOLVE_BLOCKS=3",
]

_oracle_cache = {}


def _oracle_frame(cfg):
    if cfg not in _oracle_cache:
        sid, _, algo, res, iters = cfg
        o = Oracle(cornell_scene(sid, res, res), algo, threads=8)
        for it in range(iters):
            o.run_iteration(it, 0, 10)
        _oracle_cache[cfg] = (o.framebuffer() * np.float32(1.0 / iters), o.stats())
    return _oracle_cache[cfg]


def _read_pfm(path):
    raw = open(path, "rb").read()
    a, b, c, data = raw.split(b"\n", 3)
    w, h = (int(x) for x in b.split())
    assert a == b"PF" and c == b"-1"
    return np.frombuffer(data, np.float32).reshape(h, w, 3)


@pytest.mark.skipif(not os.path.exists(HOST), reason="vcm_render not built")
@pytest.mark.parametrize("switch", ["(default)"] + SWITCHES)
def test_switch_renders_the_default_frames(tmp_path, switch):
    env = dict(os.environ)
    strict = False
    if switch != "(default)":
        for kv in switch.split():
            k, v = kv.split("=")
            env[k] = v
            strict = strict or k == "SMALLVCM_AMD_STRICT_ORDER"
    for cfg in CONFIGS:
        sid, name, algo, res, iters = cfg
        out = str(tmp_path / ("s%d.pfm" % sid))
        r = subprocess.run([HOST, "-s", str(sid), "-a", name, "-i", str(iters), "--res", str(res), str(res), "-o", out, "--json"],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, (switch, r.stdout[-1000:] + r.stderr[-2000:])
        info = json.loads(r.stdout.strip().splitlines()[-1])
        ref, stats = _oracle_frame(cfg)
        img = _read_pfm(out)
        if strict and algo == 4:   # strict mode splats with fp32 atomics (DESIGN "Two modes")
            assert np.allclose(img, ref, rtol=2e-5, atol=2e-7), switch
        else:
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (switch, cfg)
        for k in ("lightVertices", "lightRays", "cameraRays", "shadowRays", "mergeQueries", "mergeCandidates", "mergeAccepted",
                  "connections", "lightSplats"):
            assert info["last_iteration_counters"][k] == stats[k], (switch, cfg, k)
