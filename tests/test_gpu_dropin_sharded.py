"""GPU tests of the two host bindings of the C-ABI:
  * the C++ drop-in: the reference's UNCHANGED smallvcm.cxx + config.hxx built
    over smallvcm_amd/dropin/vertexcm.hxx (prebuilt: dropin/smallvcm);
  * the sharded host (ShardedVertexCM + vcm_create_sharded / export / import),
    here with 2-4 ranks as threads on ONE GPU and an in-process stand-in for
    the collectives (RCCL itself needs one GPU per rank)."""
import os
import subprocess
import threading

import numpy as np
import pytest

from smallvcm_amd.renderer import HipBackend, RenderFarm, ShardedVertexCM, VertexCM, cornell_scene, static_schedule

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "smallvcm_amd", "dropin", "smallvcm")


def _rgbe(fb):
    """Framebuffer::SaveHDR (reference src/framebuffer.hxx:219-251)"""
    v = fb.max(axis=2)
    out = np.zeros(fb.shape[:2] + (4,), np.uint8)
    ok = v >= 1e-32
    m, e = np.frexp(v.astype(np.float32))
    scale = np.where(ok, (m.astype(np.float32) * np.float32(256.0) / np.where(ok, v, 1)).astype(np.float32), 0).astype(np.float32)
    out[..., :3] = (fb * scale[..., None]).astype(np.uint8)
    out[..., 3] = np.where(ok, e + 128, 0).astype(np.uint8)
    out[~ok] = 0
    return out


def _read_hdr(path):
    raw = open(path, "rb").read()
    head, _, body = raw.partition(b"\n\n")
    dims, _, pix = body.partition(b"\n")
    t = dims.split()
    h, w = int(t[1]), int(t[3])
    return np.frombuffer(pix, np.uint8).reshape(h, w, 4)


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="drop-in binary not built (needs a SmallVCM checkout)")
@pytest.mark.parametrize("algo,name", [(2, "bpm"), (4, "vcm")])
def test_reference_driver_over_dropin(tmp_path, algo, name):
    out = str(tmp_path / ("img_%s.hdr" % name))
    r = subprocess.run([DROPIN, "-s", "1", "-a", name, "-i", "1", "-o", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "done in" in r.stdout
    img = _read_hdr(out)
    # the same render through the Python host: 512x512 (config.hxx:237), seed 1234 (:234), maxPathLength 10 (:235)
    v = VertexCM(cornell_scene(1, 512, 512), algo, 0.003, 0.75, 1234)
    v.mMaxPathLength, v.mMinPathLength = 10, 0
    v.RunIteration(0)
    mine = _rgbe(v.GetFramebuffer())
    v.close()
    d = np.abs(img.astype(np.int32) - mine.astype(np.int32))
    assert d.max() == 0   # deterministic, also with light splats


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="drop-in binary not built (needs a SmallVCM checkout)")
@pytest.mark.parametrize("algo,name", [(5, "pt"), (6, "el")])
def test_reference_driver_runs_path_tracer_and_eye_light_on_the_gpu(tmp_path, algo, name):
    """`-a pt` / `-a el` of the unchanged driver over dropin/pathtracer.hxx / eyelight.hxx"""
    out = str(tmp_path / ("img_%s.hdr" % name))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(["taskset", "-c", "0", DROPIN, "-s", "0", "-a", name, "-i", "2", "-o", out], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    img = _read_hdr(out)
    # one host core => render() creates one renderer (smallvcm.cxx:66, :280), seed 1234, iterations 0 and 1
    v = VertexCM(cornell_scene(0, 512, 512), algo, 0.003, 0.75, 1234)
    v.mMaxPathLength, v.mMinPathLength = 10, 0
    v.RunIteration(0)
    v.RunIteration(1)
    mine = _rgbe(v.GetFramebuffer())
    v.close()
    assert np.array_equal(img, mine)


def test_no_gpu_error_path_is_loud(tmp_path):
    if not os.path.exists(DROPIN):
        pytest.skip("drop-in binary not built")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    r = subprocess.run([DROPIN, "-s", "1", "-a", "vcm", "-i", "1", "-o", str(tmp_path / "x.hdr")], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode == 2 and "no HIP device" in (r.stdout + r.stderr)


class _ThreadCollectives:
    """all_gather_into_tensor / all_reduce / new_group for ranks that are threads of one process."""

    class ReduceOp:
        SUM = "sum"

    class _Group:
        def __init__(self, ranks):
            self.ranks = list(ranks)
            self.bar = threading.Barrier(len(self.ranks))
            self.slots = [None] * len(self.ranks)

    def __init__(self, world):
        import torch
        self.torch = torch
        self.world = world
        self.all = self._Group(range(world))
        self.bar = self.all.bar
        self.groups, self.lock = {}, threading.Lock()
        self.tls = threading.local()

    def bind(self, rank):
        self.tls.rank = rank

    def new_group(self, ranks):
        """every rank (thread) creates every group in the same order: the n-th call with these ranks is the same
        communicator on all of them"""
        seen = self.tls.__dict__.setdefault("created", {})
        n = seen[tuple(ranks)] = seen.get(tuple(ranks), 0) + 1
        with self.lock:
            return self.groups.setdefault((tuple(ranks), n), self._Group(ranks))

    def _exchange(self, t, group):
        g = group or self.all
        self.torch.cuda.current_stream().synchronize()
        g.slots[g.ranks.index(self.tls.rank)] = t
        g.bar.wait()
        return g, list(g.slots)

    class _Done:
        def wait(self):
            return True

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        g, parts = self._exchange(inp, group)
        n = inp.numel()
        for r, p in enumerate(parts):
            out[r * n:(r + 1) * n].copy_(p)
        self.torch.cuda.current_stream().synchronize()
        g.bar.wait()
        return self._Done()

    def all_reduce(self, t, op=None, group=None):
        g, parts = self._exchange(t.clone(), group)
        acc = parts[0].clone()
        for p in parts[1:]:
            acc += p
        t.copy_(acc)
        self.torch.cuda.current_stream().synchronize()
        g.bar.wait()


@pytest.mark.parametrize("world,sid,algo,res,iters", [(2, 1, 4, 128, 2), (4, 1, 2, 96, 2), (3, 3, 4, 100, 1),
                                                      (8, 1, 4, 64, 1), (5, 2, 2, 50, 2)])   # eight shards; a point light, five shards
def test_sharded_contexts_equal_single_context(world, sid, algo, res, iters):
    sc = cornell_scene(sid, res, res)
    coll = _ThreadCollectives(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            coll.bind(rank)
            b = HipBackend(sc, algo, 0.003, 0.75, 1234, device=0, rank=rank, world=world)
            r = ShardedVertexCM(b, rank, world)
            r.dist = coll
            r.mMaxPathLength, r.mMinPathLength = 10, 0
            for it in range(iters):
                r.RunIteration(it)
            grid = b.grid() if algo in (1, 2, 4) else None
            results[rank] = (r.framebuffer_sum(), b.stats(), grid)
            b.close()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))
            try:
                coll.bar.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not errors, errors
    one = VertexCM(sc, algo, 0.003, 0.75, 1234)
    one.mMaxPathLength, one.mMinPathLength = 10, 0
    for it in range(iters):
        one.RunIteration(it)
    ref, st1 = one.framebuffer_sum(), one.stats()
    grid1 = one.backend.grid() if algo in (1, 2, 4) else None
    one.close()
    for _, _, grid in results:
        if grid is not None:   # the box exchanged as 7 numbers per rank and the grid built over the gathered records
            assert np.array_equal(grid[2], grid1[2]), "bounding box"
            assert np.array_equal(grid[0], grid1[0]) and np.array_equal(grid[1], grid1[1]), "cell ranges / in-cell order"
    for fb, st, _ in results:
        assert np.allclose(fb, ref, rtol=2e-6, atol=2e-7)
        assert st["gridVertices"] == st1["gridVertices"]          # every rank built the full grid
    assert sum(st["lightVertices"] for _, st, _ in results) == st1["lightVertices"]
    assert sum(st["mergeAccepted"] for _, st, _ in results) == st1["mergeAccepted"]
    assert sum(st["mergeCandidates"] for _, st, _ in results) == st1["mergeCandidates"]
    assert sum(st["connections"] for _, st, _ in results) == st1["connections"]


# ---- many renderers on one device: the shared iteration-scratch arena -------------------------------------------
@pytest.mark.skipif(not os.path.exists(DROPIN), reason="drop-in binary not built (needs a SmallVCM checkout)")
def test_reference_driver_time_mode_uses_every_host_core(tmp_path):
    """`-t` makes render() run one renderer per host core concurrently (smallvcm.cxx:66, :82-108): on the GPU
    box that is 256 contexts on one device, which only fit because the iteration scratch is shared."""
    out = str(tmp_path / "img_t.hdr")
    # (-t counts the CPU time of the process, smallvcm.cxx:74-83: 8 "seconds" on 256 threads are 31 ms of wall time -- enough for
    # the OpenMP team to wake up before the limit has passed; with 2 the margin was 8 ms)
    r = subprocess.run([DROPIN, "-s", "1", "-a", "vcm", "-t", "8", "-o", out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "done in" in r.stdout
    img = _read_hdr(out).astype(np.int32)
    assert img.shape[:2] == (512, 512) and img[..., 3].max() > 0
    # more iterations of the same estimator: close to a 1-iteration render, pixel for pixel on average
    v = VertexCM(cornell_scene(1, 512, 512), 4, 0.003, 0.75, 1234)
    v.mMaxPathLength, v.mMinPathLength = 10, 0
    for i in range(4):
        v.RunIteration(i)
    mine = v.GetFramebuffer()
    v.close()
    rgb = img[..., :3] * np.exp2(img[..., 3:4] - 136.0) * (img[..., 3:4] > 0)
    assert abs(rgb.mean() - mine.mean()) < 0.05 * mine.mean()


def test_concurrent_contexts_share_scratch_and_stay_exact():
    """8 host threads, each with its own context on device 0, iterate concurrently; every image must equal the
    image of the same renderer run alone (the arena hand-over is ordered on the GPU by an event)."""
    sc = cornell_scene(1, 96, 80)
    seeds = [1234 + 7 * i for i in range(8)]

    def render(seed, iters=3):
        v = VertexCM(sc, 4, 0.003, 0.75, seed)
        v.mMaxPathLength, v.mMinPathLength = 10, 0
        for it in range(iters):
            v.RunIteration(it)
        fb = v.framebuffer_sum()
        v.close()
        return fb

    alone = [render(s) for s in seeds]
    together, errors = [None] * len(seeds), []

    def run(i):
        try:
            together[i] = render(seeds[i])
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=run, args=(i,)) for i in range(len(seeds))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not errors, errors
    for a, b in zip(alone, together):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_interleaving_two_iterations_on_one_thread():
    """The iteration scratch is borrowed from a per-device pool of arenas.  With ONE arena a thread that opens a second
    iteration before ending the first is refused (it would wait for itself); with two, both iterations are open at
    once, overlap on the GPU, and each renderer's scratch stays readable afterwards."""
    from smallvcm_amd.renderer import load_library
    L = load_library()
    sc = cornell_scene(0, 32, 32)
    alone = []
    for seed in (1, 2):
        x = HipBackend(sc, 4, 0.003, 0.75, seed)
        x.run_iteration(0, 0, 10)
        alone.append((x.framebuffer_sum(), x.records()))
        x.close()
    assert L.vcm_set_arena_limit(0, 1) == 0
    a = HipBackend(sc, 4, 0.003, 0.75, 1)
    b = HipBackend(sc, 4, 0.003, 0.75, 2)
    a.begin(0, 0, 10)
    with pytest.raises(RuntimeError, match="already inside an iteration"):
        b.begin(0, 0, 10)
    a.trace_light(); a.build_grid(); a.trace_camera(); a.merge(); a.end()
    b.begin(0, 0, 10)
    b.trace_light(); b.build_grid(); b.trace_camera(); b.merge(); b.end()
    # the scratch now belongs to b's last iteration: a's records are gone, and saying so beats returning b's
    with pytest.raises(RuntimeError, match="used by another context"):
        a.records()
    assert b.records().shape[0] > 0
    a.close(); b.close()
    assert L.vcm_set_arena_limit(0, 2) == 0
    a = HipBackend(sc, 4, 0.003, 0.75, 1)
    b = HipBackend(sc, 4, 0.003, 0.75, 2)
    a.begin(0, 0, 10); b.begin(0, 0, 10)
    a.trace_light(); b.trace_light(); a.build_grid(); b.build_grid(); b.trace_camera(); a.trace_camera()
    a.merge(); b.merge(); b.end(); a.end()
    for x, (fb, recs) in zip((a, b), alone):
        assert np.array_equal(x.framebuffer_sum().view(np.uint32), fb.view(np.uint32))
        assert np.array_equal(x.records().view(np.uint32), recs.view(np.uint32))
    a.close(); b.close()
    assert L.vcm_set_arena_limit(0, 0) == 0


@pytest.mark.parametrize("world,shards,inflight,algo,res,iters", [(4, 2, 1, 4, 96, 5), (3, 1, 1, 4, 64, 4), (2, 2, 2, 4, 96, 5),
                                                                  (4, 2, 2, 2, 64, 9)])
def test_render_farm_on_one_gpu(world, shards, inflight, algo, res, iters):
    """RenderFarm (replica groups x path shards) with threads as ranks: equals the mean of the group renderers run
    alone with seeds base + g over their static-schedule iteration blocks (smallvcm.cxx:61-142)."""
    sc = cornell_scene(1, res, res)
    coll = _ThreadCollectives(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            coll.bind(rank)
            farm = RenderFarm(lambda seed, s, S: HipBackend(sc, algo, 0.003, 0.75, seed, device=0, rank=s, world=S),
                              1234, rank, world, shards=shards, dist=coll, inflight=inflight)
            farm.set_path_lengths(0, 10)
            farm.render(iters)
            results[rank] = farm.framebuffer()
            farm.close()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))
            for g in [coll.all] + list(coll.groups.values()):
                g.bar.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not errors, errors
    replicas = (world // shards) * inflight
    total = np.zeros((res, res, 3), np.float64)
    for g in range(replicas):
        its = static_schedule(iters, replicas, g)
        assert len(its) > 0
        v = VertexCM(sc, algo, 0.003, 0.75, 1234 + g)
        v.mMaxPathLength, v.mMinPathLength = 10, 0
        for it in its:
            v.RunIteration(it)
        total += v.GetFramebuffer().astype(np.float64)
        v.close()
    ref = (total / replicas).astype(np.float32)
    for fb in results:
        assert np.allclose(fb, ref, rtol=3e-6, atol=2e-7)


def test_rccl_plumbing_with_one_rank():
    """see tests/nccl_single_rank.py"""
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_single_rank.py"), str(port)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---- the C++ host over the C-ABI alone (smallvcm_amd/host/vcm_render.cpp) ------------------------------------------
HOST = os.path.join(ROOT, "smallvcm_amd", "host", "vcm_render")


def _read_pfm(path):
    raw = open(path, "rb").read()
    a, b, c, data = raw.split(b"\n", 3)
    w, h = (int(x) for x in b.split())
    assert a == b"PF" and c == b"-1"   # Framebuffer::SavePFM, framebuffer.hxx:137-146
    return np.frombuffer(data, np.float32).reshape(h, w, 3)


@pytest.mark.skipif(not os.path.exists(HOST), reason="vcm_render not built")
@pytest.mark.parametrize("name,algo,renderers,iters", [("vcm", 4, 2, 5), ("bpm", 2, 1, 2), ("pt", 5, 3, 4)])
def test_cpp_host_equals_python_host(tmp_path, name, algo, renderers, iters):
    """non-square resolution, seed, path-length window and renderer count from the command line; raw fp32 PFM"""
    import json
    out = str(tmp_path / "img.pfm")
    r = subprocess.run([HOST, "-s", "1", "-a", name, "-i", str(iters), "--res", "96", "72", "--seed", "77", "--minlen", "1",
                        "--maxlen", "6", "--renderers", str(renderers), "-o", out, "--json"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["res"] == [96, 72] and info["Mpaths_s"] > 0
    img = _read_pfm(out)
    sc = cornell_scene(1, 96, 72)
    acc, used = None, 0
    for g in range(renderers):
        its = static_schedule(iters, renderers, g)
        if len(its) == 0:
            continue
        v = VertexCM(sc, algo, 0.003, 0.75, 77 + g)
        v.mMinPathLength, v.mMaxPathLength = 1, 6
        for it in its:
            v.RunIteration(it)
        f = v.framebuffer_sum() * np.float32(1.0 / len(its))
        acc = f if acc is None else acc + f
        used += 1
        v.close()
    ref = acc * np.float32(1.0 / used)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert abs(info["image_mean"][0] - float(ref[..., 0].mean())) < 1e-4


@pytest.mark.skipif(not os.path.exists(HOST), reason="vcm_render not built")
def test_cpp_host_rejects_bad_arguments():
    r = subprocess.run([HOST, "-a", "nope"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2


@pytest.mark.skipif(not (os.path.exists(HOST) and os.path.exists(DROPIN)), reason="needs vcm_render and the drop-in binary")
def test_cpp_host_writes_the_reference_file_formats(tmp_path):
    """vcm_render -o x.hdr / x.bmp (pixels encoded on the device, vcm_read_image) against the files the reference's
    own writers produce from the same render (unchanged smallvcm.cxx over the drop-in)."""
    ref_hdr, ref_bmp = str(tmp_path / "ref.hdr"), str(tmp_path / "ref.bmp")
    mine_hdr, mine_bmp = str(tmp_path / "mine.hdr"), str(tmp_path / "mine.bmp")
    for out in (ref_hdr, ref_bmp):
        r = subprocess.run([DROPIN, "-s", "1", "-a", "vcm", "-i", "1", "-o", out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
    for out in (mine_hdr, mine_bmp):
        r = subprocess.run([HOST, "-s", "1", "-a", "vcm", "-i", "1", "-o", out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
    assert open(ref_hdr, "rb").read() == open(mine_hdr, "rb").read()          # header and RGBE pixels, byte for byte
    a, b = open(ref_bmp, "rb").read(), open(mine_bmp, "rb").read()
    assert len(a) == len(b) == 54 + 512 * 512 * 3
    assert a == b   # the gamma curve is pow(x, 1 / 2.2): the library's pow IS glibc's since round 4, so every byte agrees


def test_read_image_encodings():
    sc = cornell_scene(1, 96, 64)
    v = VertexCM(sc, 4, 0.003, 0.75, 1234)
    v.mMaxPathLength = 10
    for it in range(3):
        v.RunIteration(it)
    fb = v.GetFramebuffer()
    rgbe = v.backend.read_image(1, 1.0 / 3.0)
    assert np.array_equal(rgbe, _rgbe(v.framebuffer_sum() * np.float32(1.0 / 3.0)))
    bgr = v.backend.read_image(0, 1.0 / 3.0, 2.2)
    ref = np.clip(np.power((v.framebuffer_sum() * np.float32(1.0 / 3.0)).astype(np.float64), 1 / 2.2) * 255.0, 0, 255)
    ref = ref[::-1, :, ::-1]                      # bottom-up, BGR
    assert np.abs(bgr.astype(np.float64) - np.floor(ref)).max() <= 1
    assert np.allclose(fb, v.framebuffer_sum() / 3)
    v.close()


def _read_bmp_mean(path):
    """mean of a 24-bit BMP as Framebuffer::SaveBMP writes it (framebuffer.hxx:170-216): 54-byte header, BGR rows"""
    raw = open(path, "rb").read()
    assert raw[:2] == b"BM"
    w = int.from_bytes(raw[18:22], "little", signed=True)
    h = int.from_bytes(raw[22:26], "little", signed=True)
    off = int.from_bytes(raw[10:14], "little")
    px = np.frombuffer(raw, np.uint8, count=w * abs(h) * 3, offset=off).reshape(abs(h), w, 3)
    return w, abs(h), px.astype(np.float64).mean(axis=(0, 1))[::-1]   # -> RGB


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="drop-in binary not built (needs a SmallVCM checkout)")
def test_reference_driver_full_report_over_dropin(tmp_path):
    """`smallvcm --report -t 2` (FullReport, smallvcm.cxx:156-263; html_writer.hxx) of the UNCHANGED driver over the
    drop-in: every scene x algorithm combination renders on the GPU (7 algorithms x 4 scenes, one renderer per core the
    process may use each), index.html and the 28 images are written, and the images of one scene agree: the unbiased / consistent
    estimators (pt, bpt, bpm, vcm) have the same mean up to Monte-Carlo error and 8-bit gamma quantisation."""
    # The driver's time limit is clock() -- the CPU time of the PROCESS (smallvcm.cxx:74-83) -- so with one renderer per host core
    # "1 second" on a 256-thread host is 4 ms of wall time from `startT = clock()` to every thread's first look at the clock; when
    # waking the OpenMP team takes longer than that, no renderer runs an iteration and the reference writes a 0 x 0 image (seen
    # once in round 5).  32 cores and 2 seconds: 60 ms, and 32 instead of 256 renderers to create per combination.
    cores = sorted(os.sched_getaffinity(0))[:32]
    r = subprocess.run(["taskset", "-c", ",".join(str(c) for c in cores), DROPIN, "--report", "-t", "2"], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Whole run took" in r.stdout
    html = open(os.path.join(str(tmp_path), "index.html")).read()
    bmps = sorted(f for f in os.listdir(str(tmp_path)) if f.endswith(".bmp"))
    assert len(bmps) == 28, bmps
    for f in bmps:
        assert f in html
    by_scene = {}
    for f in bmps:
        w, h, mean = _read_bmp_mean(os.path.join(str(tmp_path), f))
        assert (w, h) == (512, 512) and mean.max() > 1.0, f
        # DefaultFilename (config.hxx:153-218): <scene acronym>_<algorithm acronym>.bmp
        scene, algo = f[:-4].rsplit("_", 1)
        by_scene.setdefault(scene, {})[algo] = mean
    assert len(by_scene) == 4 and all(len(v) == 7 for v in by_scene.values()), {k: sorted(v) for k, v in by_scene.items()}
    for scene, imgs in by_scene.items():
        ref = imgs["vcm"]
        # bpm and vcm sample every light path of these scenes (the caustics through the glass sphere included): same mean
        assert np.all(np.abs(imgs["bpm"] - ref) < 0.08 * ref + 2.0), (scene, imgs["bpm"], ref)
        # the others miss or under-sample parts of the transport by construction (pt / bpt: caustics of the delta lights,
        # lt: everything seen through a specular surface, ppm: biased, el: a shading preview) -- same picture, darker
        for algo in ("pt", "bpt", "lt", "ppm"):
            assert np.all(imgs[algo] > 0.35 * ref - 2.0) and np.all(imgs[algo] < 1.3 * ref + 2.0), (scene, algo, imgs[algo], ref)


# ---- the C++ multi-GPU host (smallvcm_amd/host/vcm_farm.cpp): ranks = host threads, RCCL between them -----------------
def _farm(tmp_path, tag, *extra, iters=5, res=(96, 80), algo="vcm", scene=1, env=None):
    import json
    out = str(tmp_path / ("farm_%s.pfm" % tag))
    r = subprocess.run([HOST, "-s", str(scene), "-a", algo, "-i", str(iters), "--res", str(res[0]), str(res[1]), "-o", out, "--json",
                        *extra], capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return _read_pfm(out), json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.skipif(not os.path.exists(HOST), reason="vcm_render not built")
def test_cpp_farm_with_rccl_on_one_rank_is_bit_exact(tmp_path):
    """--gpus 1: the real RCCL calls (ncclCommInitAll, the framebuffer path) with one rank = one renderer; its image is
    the single renderer's, bit for bit."""
    ref, _ = _farm(tmp_path, "ref", "--renderers", "1")
    one, info = _farm(tmp_path, "rccl1", "--gpus", "1", "--shards", "1")
    assert info["gpus"] == 1 and info["renderers"] == 1
    assert np.array_equal(ref.view(np.uint32), one.view(np.uint32))
    # the cross-process form of the small collectives (a second communicator from ncclCommSplit, a stream of its own: so
    # that a 32-byte exchange never waits behind another renderer's all-gather) as far as one rank can take it
    two, info = _farm(tmp_path, "rccl1x", "--gpus", "1", "--shards", "1", env={"SMALLVCM_AMD_FARM_RCCL_EXCHANGE": "1"})
    assert info["gpus"] == 1 and np.array_equal(ref.view(np.uint32), two.view(np.uint32))


@pytest.mark.skipif(not os.path.exists(HOST), reason="vcm_render not built")
@pytest.mark.parametrize("ranks,shards,inflight,algo,iters", [(2, 2, 1, "vcm", 4), (4, 2, 2, "vcm", 9), (3, 3, 1, "bpm", 3), (4, 1, 1, "vcm", 6),
                                                              (2, 2, 2, "bpt", 5),
                                                              # eight ranks, the shapes `bench.py --gpus 8` launches: pairs x 2 in
                                                              # flight (default), one renderer on eight shards, eight replicas
                                                              (8, 2, 2, "vcm", 10), (8, 8, 1, "vcm", 3), (8, 1, 1, "vcm", 9)])
def test_cpp_farm_thread_ranks_equal_single_gpu_renderers(tmp_path, ranks, shards, inflight, algo, iters):
    """Several ranks on the one GPU (in-process stand-in for RCCL, same rank logic: 7-number exchange, all-gather of the
    merge records on the second stream, import, grid, merge, framebuffer reduce): the image equals the mean of the
    renderers run alone (seeds base + g, static-schedule iteration blocks) up to the order of the final sum."""
    img, info = _farm(tmp_path, "t", "--gpus", str(ranks), "--shards", str(shards), "--inflight", str(inflight), "--collectives",
                      "threads", iters=iters, algo=algo)
    renderers = (ranks // shards) * inflight
    assert info["renderers"] == renderers
    ref, _ = _farm(tmp_path, "r", "--renderers", str(renderers), iters=iters, algo=algo)
    assert np.allclose(img, ref, rtol=3e-6, atol=2e-7)
    assert img.max() > 0


@pytest.mark.skipif(not os.path.exists(HOST), reason="vcm_render not built")
@pytest.mark.parametrize("ranks,algo,scene,res", [(8, "vcm", 1, (64, 48)), (5, "bpm", 3, (50, 50)), (2, "vcm", 2, (96, 80)), (3, "ppm", 0, (40, 56))])
def test_sorted_and_unsorted_exchange_render_the_same_bits(tmp_path, ranks, algo, scene, res):
    """ONE renderer on `ranks` shards, twice: with the sorted exchange of round 5 (every rank sorts its own light vertices by
    hash cell, k_grid_merge_blocks places the slabs: the default) and with the unsorted exchange of rounds 1-4
    (SMALLVCM_AMD_SORTED_EXCHANGE=0: records in the reference's order, the whole grid build on every rank).  Both build
    HashGrid::Build's grid (hashgrid.hxx:41-107) -- same cells, same in-cell order -- so the frames are the same bits, and
    serialising the ranks (the scaling model's measurement mode) changes nothing either.  Shard counts that do not divide
    the frame, more shards than cell blocks, a point light whose photons crowd a few cells, an environment light."""
    args = ("--gpus", str(ranks), "--shards", str(ranks), "--inflight", "1", "--collectives", "threads")
    a, info = _farm(tmp_path, "sorted", *args, iters=3, res=res, algo=algo, scene=scene)
    b, _ = _farm(tmp_path, "unsorted", *args, iters=3, res=res, algo=algo, scene=scene, env={"SMALLVCM_AMD_SORTED_EXCHANGE": "0"})
    c, _ = _farm(tmp_path, "turns", *args, iters=3, res=res, algo=algo, scene=scene, env={"SMALLVCM_AMD_FARM_SERIALIZE": "1"})
    assert info["renderers"] == 1 and a.max() > 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32))
    one, _ = _farm(tmp_path, "one", "--renderers", "1", iters=3, res=res, algo=algo, scene=scene)
    assert np.allclose(a, one, rtol=3e-6, atol=2e-7)   # the shards' partial sums meet in the all-reduce: another order of the final sum


@pytest.mark.skipif(not os.path.exists(HOST), reason="vcm_render not built")
@pytest.mark.parametrize("ranks,algo,res", [(4, "vcm", (96, 80)), (8, "bpm", (64, 48)), (3, "vcm", (50, 50))])
def test_direct_exchange_equals_the_all_gather(tmp_path, ranks, algo, res):
    """SMALLVCM_AMD_FARM_EXCHANGE=direct (every rank hands its slab of light vertices to each peer, all
    transfers at once, instead of one ring-class all-gather): the same bytes in the same places, so the same bits -- here with
    the in-process stand-in, whose direct form has every rank WRITE into its peers' receive buffers; the RCCL form
    (grouped ncclSend / ncclRecv) runs wherever two GPUs exist: bench.py --gpus N --selftest."""
    args = ("--gpus", str(ranks), "--shards", str(ranks), "--inflight", "1", "--collectives", "threads")
    a, _ = _farm(tmp_path, "ag", *args, iters=3, res=res, algo=algo)
    b, _ = _farm(tmp_path, "direct", *args, iters=3, res=res, algo=algo, env={"SMALLVCM_AMD_FARM_EXCHANGE": "direct"})
    c, _ = _farm(tmp_path, "direct_unsorted", *args, iters=3, res=res, algo=algo, env={"SMALLVCM_AMD_FARM_EXCHANGE": "direct", "SMALLVCM_AMD_SORTED_EXCHANGE": "0"})
    assert a.max() > 0 and np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(a.view(np.uint32), c.view(np.uint32))


@pytest.mark.skipif(not os.path.exists(HOST), reason="vcm_render not built")
def test_eight_shards_at_full_size_equal_one_renderer(tmp_path):
    """BASELINE.json's fifth configuration -- scene 1, VCM, 2048 x 2048 on 8 shards (vertexcm.hxx:504-506: vertex connection
    stays local; :532-533: the merge sees every rank's light vertices) -- at FULL size, not the <= 128^2 of the tests above
    (VERDICT r5): eight rank threads on this one GPU (the in-process stand-in for RCCL, ranks taking turns), one
    iteration (the first: the largest radius, the most merge candidates).  The work counters, summed over the shards, are the single renderer's exactly -- every path traced once, every
    light vertex in every rank's grid, every camera vertex merged once --, and the frame is the single renderer's up to the
    order of ONE sum: eight partial framebuffers of light splats meet in the all-reduce."""
    args = ("--gpus", "8", "--shards", "8", "--inflight", "1", "--collectives", "threads")
    one, i1 = _farm(tmp_path, "one", "--renderers", "1", iters=1, res=(2048, 2048))
    eight, i8 = _farm(tmp_path, "eight", *args, iters=1, res=(2048, 2048), env={"SMALLVCM_AMD_FARM_SERIALIZE": "1"})
    assert i8["renderers"] == 1 and i8["gpus"] == 8
    for k, v in i1["last_iteration_counters"].items():
        assert i8["last_iteration_counters"][k] == v, (k, v, i8["last_iteration_counters"][k])
    assert i1["last_iteration_counters"]["mergeAccepted"] > 100_000_000
    assert np.allclose(eight, one, rtol=3e-6, atol=2e-7)
    d = eight.astype(np.float64) - one.astype(np.float64)
    assert float(np.sqrt(np.mean(d * d))) < 1e-7 * float(one.mean())


@pytest.mark.skipif(not os.path.exists(HOST), reason="vcm_render not built")
@pytest.mark.parametrize("ranks,algo,scene,res", [(2, "vcm", 1, (96, 80)), (4, "vcm", 1, (96, 80)), (8, "vcm", 1, (64, 48)), (5, "bpm", 3, (50, 50)),
                                                  (3, "ppm", 0, (40, 56)), (4, "bpm", 2, (72, 72))])
def test_merge_sharded_by_space_renders_the_same_bits(tmp_path, ranks, algo, scene, res):
    """SMALLVCM_AMD_FARM_MERGE=space (round 6 prototype): the ranks own slabs of grid cells; light vertices travel to the owners
    of their cell (+ one cell of halo), camera vertices to the owner of their base cell, the merge terms back -- instead of every rank
    merging against ALL light vertices (vertexcm.hxx:532-533).  A query's accepted photons all lie in its own 2 x 2 x 2 block of cells
    (hashgrid.hxx:124-155), each rank's grid keeps HashGrid::Build's in-cell order on the photons it holds, so the frame is the index-sharded
    renderer's bit for bit and every work counter but the candidate count is equal (a hash bucket no longer holds the far-away cells that
    collide into it: fewer distance tests, the same acceptances)."""
    args = ("--gpus", str(ranks), "--shards", str(ranks), "--inflight", "1", "--collectives", "threads")
    a, ia = _farm(tmp_path, "index", *args, iters=1, res=res, algo=algo, scene=scene)
    b, ib = _farm(tmp_path, "space", *args, iters=1, res=res, algo=algo, scene=scene, env={"SMALLVCM_AMD_FARM_MERGE": "space"})
    c, _ = _farm(tmp_path, "turns", *args, iters=3, res=res, algo=algo, scene=scene, env={"SMALLVCM_AMD_FARM_MERGE": "space", "SMALLVCM_AMD_FARM_SERIALIZE": "1"})
    d, _ = _farm(tmp_path, "index3", *args, iters=3, res=res, algo=algo, scene=scene)
    assert a.max() > 0 and np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(c.view(np.uint32), d.view(np.uint32))
    for k, v in ia["last_iteration_counters"].items():
        if k == "mergeCandidates":
            assert 0 < ib["last_iteration_counters"][k] <= v
        else:
            assert ib["last_iteration_counters"][k] == v, (k, v, ib["last_iteration_counters"][k])


@pytest.mark.skipif(not os.path.exists(HOST), reason="vcm_render not built")
def test_merge_sharded_by_space_at_full_size(tmp_path):
    """... and at BASELINE's fifth configuration: scene 1 VCM 2048 x 2048 on 8 shards, one iteration, ranks taking turns."""
    args = ("--gpus", "8", "--shards", "8", "--inflight", "1", "--collectives", "threads")
    a, ia = _farm(tmp_path, "index", *args, iters=1, res=(2048, 2048), env={"SMALLVCM_AMD_FARM_SERIALIZE": "1"})
    b, ib = _farm(tmp_path, "space", *args, iters=1, res=(2048, 2048), env={"SMALLVCM_AMD_FARM_SERIALIZE": "1", "SMALLVCM_AMD_FARM_MERGE": "space"})
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert ia["last_iteration_counters"]["mergeAccepted"] == ib["last_iteration_counters"]["mergeAccepted"] > 100_000_000
    assert ib["last_iteration_counters"]["mergeCandidates"] < ia["last_iteration_counters"]["mergeCandidates"]


def test_sorted_exchange_and_pinning_entry_points_say_what_they_refuse():
    """vcm_sorted_slab_words / vcm_sort_light_records / vcm_import_sorted_light_records (include/smallvcm_amd.h): the slab size
    is records x 13 words + one block-start word per block of cells (+ padding to 16 bytes); contexts that cannot use the
    sorted exchange get -1 and a reason; sorting before the box of ALL ranks is known is refused (the cell of a vertex depends
    on it, hashgrid.hxx:47-61, :189-201) and ends the iteration like every failed phase call.  vcm_pin_host_memory: a
    page-locked destination reads the same framebuffer."""
    import ctypes as C
    from smallvcm_amd.renderer import load_library
    L = load_library()
    sc = cornell_scene(1, 64, 64)

    def err():
        return (L.vcm_last_error() or b"").decode()
    one = HipBackend(sc, 4, 0.003, 0.75, 1234, device=0, rank=0, world=1)
    assert L.vcm_sorted_slab_words(one.ctx, 100) == -1 and "not sharded" in err()
    two = HipBackend(sc, 4, 0.003, 0.75, 1234, device=0, rank=0, world=2)
    n_blocks = (64 * 64 + 1023) // 1024                      # K = 1024 cells per block for 2 shards, nCells = pathCount (vertexcm.hxx:406)
    assert L.vcm_sorted_slab_words(two.ctx, 100) == (100 * 13 + n_blocks + 1 + 3) // 4 * 4
    assert L.vcm_sorted_slab_words(two.ctx, 1 << 24) == -1 and "2^24" in err()
    assert L.vcm_sorted_slab_words(two.ctx, 0) == -1
    bpt = HipBackend(sc, 3, 0.003, 0.75, 1234, device=0, rank=0, world=2)
    assert L.vcm_sorted_slab_words(bpt.ctx, 100) == -1 and "does not merge" in err()
    bpt.close()
    two.begin(0, 0, 10)
    two.trace_light()
    buf = two.new_tensor(L.vcm_sorted_slab_words(two.ctx, 20000))
    assert L.vcm_sort_light_records(two.ctx, C.c_void_p(buf.data_ptr()), 20000) == -1 and "vcm_set_grid_bbox" in err()
    two.begin(1, 0, 10)                                       # the failed call ended the iteration: a new one opens
    two.trace_light()
    mn, mx, n = two.local_bbox()
    two.set_grid_bbox(mn, mx)
    assert n > 0 and L.vcm_sort_light_records(two.ctx, C.c_void_p(buf.data_ptr()), 20000) == 0
    arr = (C.c_longlong * 1)(n)
    assert L.vcm_import_sorted_light_records(two.ctx, C.c_void_p(buf.data_ptr()), arr, 1, 20000) == -1 and "one slab per rank" in err()
    two.close()
    # page-locked read-out
    assert L.vcm_pin_host_memory(None, 0) == -1
    one.run_iteration(0, 0, 10)
    a = one.framebuffer_sum()
    b = np.zeros_like(a)
    assert L.vcm_pin_host_memory(C.c_void_p(b.ctypes.data), b.nbytes) == 0
    assert L.vcm_read_framebuffer(one.ctx, b.ctypes.data_as(C.POINTER(C.c_float))) == 0
    assert L.vcm_unpin_host_memory(C.c_void_p(b.ctypes.data)) == 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a.max() > 0
    one.close()


# ---- the same host behind its C-ABI (include/smallvcm_amd_farm.h), as bench.py --gpus N drives it ---------------------
def test_farm_binding_equals_the_command_line_host(tmp_path):
    """smallvcm_amd.farm (ctypes onto libsmallvcm_amd_farm.so) and vcm_render --gpus run the same C++ farm: same image,
    bit for bit; per-rank iteration times and rank 0's counters come back"""
    from smallvcm_amd import farm
    sc = cornell_scene(1, 96, 80)
    r = farm.farm_render(sc, 4, iterations=6, ranks=4, shards=2, inflight=2, devices=[0, 0, 0, 0], collectives="threads")
    img, info = _farm(tmp_path, "cli", "--gpus", "4", "--shards", "2", "--inflight", "2", "--collectives", "threads", iters=6)
    assert r["renderers"] == info["renderers"] == 4 and r["rccl_ranks"] == 0
    assert np.array_equal(r["image"].view(np.uint32), img.view(np.uint32))
    assert len(r["rank_iteration_ms"]) == 4 and min(r["rank_iteration_ms"]) > 0
    assert r["stats"]["lightVertices"] > 0 and r["stats"]["msTotal"] > 0 and r["wall_s"] > 0


def test_farm_over_rccl_with_shipped_ids():
    """the one-process-per-GPU path as far as one GPU can take it: the RCCL communicators come from ids made by
    vcm_farm_unique_ids and passed in as bytes (what bench.py broadcasts under torch.distributed.run); one rank, image =
    the single renderer's, bit for bit; the benchmark schedule gives every renderer the same iteration window"""
    from smallvcm_amd import farm
    sc = cornell_scene(1, 64, 64)
    ids = farm.unique_ids(2)
    assert len(ids) == 256 and ids[:128] != ids[128:]
    r = farm.farm_render(sc, 4, iterations=3, ranks=1, shards=1, inflight=1, devices=[0], collectives="rccl", ids=ids, warmup=2,
                         same_window=True)
    assert r["rccl_ranks"] == 1 and r["renderers"] == 1
    v = VertexCM(sc, 4, 0.003, 0.75, 1234)
    v.mMaxPathLength = 10
    for it in range(2):          # the farm's warm-up: iterations 0, 1, framebuffer cleared afterwards
        v.RunIteration(it)
    v.backend.clear_framebuffer()
    v.mIterations = 0
    for it in (2, 3, 4):         # same window: warmup .. warmup + n - 1
        v.RunIteration(it)
    assert np.array_equal(r["image"].view(np.uint32), v.GetFramebuffer().view(np.uint32))
    v.close()


def test_bench_line_for_several_gpus_runs_as_typed():
    """`python bench.py --gpus N` (no torch.distributed.run) drives the C++ farm in-process and prints ONE short line: the
    default is north_star's decomposition (one renderer sharded over all GPUs: "strong"), the replica hybrid beside it;
    everything else is in bench_detail.json.  --collectives threads lets the two ranks share this box's one GPU"""
    import json
    import sys
    detail = os.path.join(ROOT, "gpurun_out", "test_bench_detail.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) \
        else os.path.join("/tmp", "test_bench_detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--collectives", "threads", "--res", "256",
                        "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, SMALLVCM_AMD_BENCH_DETAIL=detail))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096          # the driver parses the LAST line of a tail of stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == 3 and d["scaling"] == "strong"
    assert d["config"]["paths_per_step"] == 2 * 256 * 256          # ONE renderer across both ranks
    assert d["hybrid_decomposition"]["scaling"] == "weak" and d["hybrid_decomposition"]["value"] > 0
    assert d["hybrid_decomposition"]["paths_per_step"] == 2 * 256 * 256 * 2    # two renderers on the pair, two in flight
    assert d["roofline"]["frac"] is None or 0 < d["roofline"]["frac"] <= 1
    full = json.load(open(detail))
    assert "C++" in full["config"]["host"] and full["config"]["rccl_ranks"] == 0
    assert len(full["rank_iteration_ms"]) == 2 and min(full["rank_iteration_ms"]) > 0
    assert full["roofline"]["per_kernel"]["k_merge"]["ms"] > 0
