"""Multi-rank host logic (smallvcm_amd.renderer.ShardedVertexCM) on CPU:
world_size 2 and 3 over gloo, the oracle as compute backend.  Sharding by path
index + all-gather of the merge records + framebuffer sum must reproduce the
unsharded result: bit-exact wherever a pixel is written by one rank only."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle_lib import Oracle
from smallvcm_amd.renderer import cornell_scene

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,sid,algo,res,iters,sorted_exchange", [
    (2, 1, 4, 40, 2, 0), (2, 1, 2, 40, 2, 0), (3, 3, 4, 33, 1, 0), (2, 0, 3, 32, 1, 0),
    (8, 1, 4, 40, 1, 0),    # the node's size: eight shards of one renderer
    # the sorted exchange of round 5 (every rank sorts its own vertices by cell, slabs travel, placement by cell):
    # ShardedVertexCM's path for it and the slab layout, restated in numpy by the worker's backend
    (2, 1, 4, 40, 2, 1), (3, 3, 4, 33, 1, 1), (2, 1, 2, 40, 2, 1), (8, 1, 4, 40, 1, 1)])
def test_sharded_equals_unsharded(tmp_path, world, sid, algo, res, iters, sorted_exchange):
    port = _free_port()
    out = str(tmp_path / "fb.npy")
    env = dict(os.environ, SHARDED_WORKER_SORTED=str(sorted_exchange))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), str(r), str(world), str(port),
                               str(sid), str(algo), str(res), str(iters), out], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    fb = np.load(out)
    o = Oracle(cornell_scene(sid, res, res), algo)
    for it in range(iters):
        o.run_iteration(it, 0, 10)
    ref = o.framebuffer()
    # per-rank partial sums are added in a different order than the serial loop
    assert np.allclose(fb, ref, rtol=2e-6, atol=1e-7)
    if algo == 2:   # no light splats: one writer per pixel (up to jitter-shifted colours)
        assert (fb == ref).mean() > 0.999


@pytest.mark.parametrize("world,shards,inflight,sid,algo,res,iters", [
    (4, 2, 1, 1, 4, 32, 5), (2, 1, 1, 1, 4, 32, 3), (3, 1, 1, 3, 2, 24, 2), (2, 2, 1, 1, 4, 32, 2),
    (2, 2, 2, 1, 4, 32, 5),     # one pair, two renderers in flight on it
    (4, 2, 2, 1, 2, 24, 7),     # two pairs x two in flight = 4 renderers, uneven iteration blocks
    (2, 2, 3, 3, 4, 24, 2),     # more renderers than iterations: the unused one must not count
    (8, 2, 2, 1, 4, 24, 9)])    # eight ranks as bench.py --gpus 8 cuts them: four pairs x two in flight = 8 renderers
def test_render_farm_equals_the_reference_render_loop(tmp_path, world, shards, inflight, sid, algo, res, iters):
    """RenderFarm = render() of src/smallvcm.cxx:52-151 with one "thread" per renderer: renderer i has seed
    base + i, runs the static-schedule block of iterations, the used renderers' means are averaged.  Renderers are
    spread over groups of `shards` ranks, `inflight` of them taking turns on each group."""
    from smallvcm_amd.renderer import static_schedule
    port = _free_port()
    out = str(tmp_path / "fb.npy")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), str(r), str(world), str(port),
                               str(sid), str(algo), str(res), str(iters), out, str(shards), str(inflight)])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    fb = np.load(out)
    replicas = (world // shards) * inflight
    total, used = np.zeros((res, res, 3), np.float64), 0
    for g in range(replicas):
        its = static_schedule(iters, replicas, g)
        if len(its) == 0:
            continue
        o = Oracle(cornell_scene(sid, res, res), algo, seed=1234 + g)   # smallvcm.cxx:68
        for it in its:
            o.run_iteration(it, 0, 10)
        total += o.framebuffer().astype(np.float64) / len(its)          # renderer.hxx:53-54
        used += 1
    ref = (total / used).astype(np.float32)                             # smallvcm.cxx:142
    assert used == min(replicas, iters)
    assert np.allclose(fb, ref, rtol=3e-6, atol=1e-7)


def test_static_schedule_is_openmp_static():
    from smallvcm_amd.renderer import static_schedule
    assert [list(static_schedule(10, 4, t)) for t in range(4)] == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]
    assert [len(static_schedule(2, 4, t)) for t in range(4)] == [1, 1, 0, 0]
    assert sorted(i for t in range(7) for i in static_schedule(23, 7, t)) == list(range(23))


@pytest.mark.parametrize("world,nP,nQ", [(2, 3000, 600), (8, 4000, 800), (3, 2500, 500)])
def test_space_sharded_merge_protocol(tmp_path, world, nP, nQ):
    """The merge sharded by SPACE (round 6 prototype: include/smallvcm_amd.h, vcm_farm.cpp step_finish_space) restated in numpy over gloo:
    slabs of cells from the summed histogram, light vertices to the owners of their cell + one cell of halo, queries to the owner of
    their base cell, HashGrid::Process against the owner's vertices only -- every query's ACCEPTED SEQUENCE (order included: it is the
    order of the sum, vertexcm.hxx:168) equals the walk over all vertices, at 2, 3 and 8 ranks; fewer than 2 x the vertices travel."""
    import json
    port = _free_port()
    out = str(tmp_path / "space.json")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "space_worker.py"), str(r), str(world), str(port), "77", str(nP), str(nQ), out])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    d = json.load(open(out))
    assert d["mismatches"] == 0 and d["accepted"] > nQ, d
    assert nP <= d["sent_photons"] < 2 * nP, d   # the halo: some vertices go to two owners, none to all
