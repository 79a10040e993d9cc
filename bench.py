#!/usr/bin/env python
"""bench.py -- Mpaths/s of the MI355X VCM integrator (reference: VertexCM::RunIteration,
src/vertexcm.hxx:284-548) on BASELINE.json's workloads.

  python bench.py [--gpus N] [--steps K] [--warmup W]            headline: scene 1, VCM, 2048x2048, maxPathLength 10
  python bench.py --scene 3 --algo vcm --res 1024                any other built-in configuration (named in the line)

A "step" is one RunIteration = N light sub-paths + N camera sub-paths, N = res*res.
Mpaths/s = 2*N*steps / seconds / 1e6.  One JSON line on stdout:

  value         whole-job Mpaths/s of the timed region (K steps after W warm-up steps, barrier + synchronize
                on both sides, max over ranks); `metric` and `config.workload` name what was actually run;
  roofline      for the kernel with the largest mean time in the timed region: its share of the algorithmic
                bytes (split per kernel in DESIGN.md section 5, evaluated with the run's
                own counters) / its HIP-event time / 8 TB/s; `traffic` = HBM bytes per launch of that kernel
                measured by TWO CHILD RUNS of this file under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
                (counter passes on their own, FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM), i.e. of
                this binary on this box; if the profiler is not available the figure of profiles/*_traffic.json is
                used when it was collected on the same kernel sources (hash compared), else null;
  configs       N = 1, headline run only: BASELINE.json's other single-GPU configurations (C1 scene 1 vcm 512^2,
                C2 scene 3 vcm 1024^2, C3 scene 1 bpm 2048^2), each timed the same way with its own roofline block;
  cpu_baseline  rank 0, N = 1: the UNMODIFIED reference (oracle/_ref/libsmallvcm_ref_stock.so = /root/reference/src
                built with the flags of its Makefile:5, its own render() loop: one renderer per thread, iterations
                dealt out, smallvcm.cxx:52-151) timed on this box's host cores in this run, wall clock, on the same
                resolution and the same iteration (= radius) window as the GPU; `port` = the oracle restatement,
                path-parallel over all cores, on a bounded sample.  Baseline only.

N > 1 is launched by torch.distributed.run (one rank per GPU) and runs the reference's render() decomposition with a
GROUP of --shards GPUs as one "thread" (smallvcm_amd.renderer.RenderFarm): inside a group the paths of an
iteration are sharded by index and the light-vertex merge records are all-gathered (RCCL) every iteration; two
renderers take turns on every group (--inflight); one framebuffer all-reduce at read-out.  --shards N = one
renderer across all GPUs ("strong"); the default is pairs ("weak").  DESIGN.md section 6 has the reasoning.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
# kernel-name PREFIXES as rocprofv3 prints them (the ray-casting kernels are templates over the kind of scene:
# "vcm::k_camera_trace<1, vcm::SceneList>")
KERNEL_KEYS = {"k_light_trace": ["vcm::k_light_trace<1"], "k_camera_trace": ["vcm::k_camera_trace<1"],
               "k_connect_di+vc": ["vcm::k_connect_di", "vcm::k_connect_vc"],
               "k_merge": ["vcm::k_merge_walk", "vcm::k_merge_staged", "vcm::k_merge_lane"]}


def _is_kernel(name, prefixes):
    return any(name.startswith(p) for p in prefixes)
KERNEL_SOURCES = ["vcm_api.hip", "vcm_kernels.h", "vcm_core.h", "vcm_math.h", "detmath.h", "philox.h", "Makefile"]
# BASELINE.json configs that fit one GPU, besides the headline (C4 at one GPU)
# (name, scene, algorithm, resolution, renderers in flight): "x4" = four renderers (seeds 1234..1237, the reference's
# iteration-parallel threads, smallvcm.cxx:61-108) taking turns on the one GPU, their iterations overlapping
OTHER_CONFIGS = [("C1", 1, "vcm", 512, 1), ("C1x4", 1, "vcm", 512, 4), ("C2", 3, "vcm", 1024, 1), ("C2x2", 3, "vcm", 1024, 2),
                 ("C3", 1, "bpm", 2048, 1), ("C4x2", 1, "vcm", 2048, 2),
                 # not a BASELINE config: SURVEY 8(f) #3, a scene beyond the Cornell boxes through the BVH
                 # (tests/mesh_scenes.py bumpy_room: a tessellated height-field floor, 10 380 primitives)
                 ("M1", "mesh:72", "vcm", 1024, 1)]


def kernel_source_hash():
    """identifies the kernels a traffic figure belongs to: sha256 over the HIP sources and build flags"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "smallvcm_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def recorded_traffic(kernel):
    """profiles/*_traffic.json collected on THESE kernel sources, newest first; (bytes, file) or (None, None)"""
    want = kernel_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("kernel_src_sha16") != want:
                continue
            t = d["kernels"]
            tot = sum(v["fetch_bytes_x2"] + v["write_bytes"] for k, v in t.items() if _is_kernel(k, KERNEL_KEYS[kernel]))
            if tot > 0:
                return int(tot), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def _pmc_mean(csv_path, kernel_names, counter, skip):
    """mean counter value per dispatch of the named kernels, first `skip` dispatches of each dropped"""
    seen, vals = {}, []
    for r in csv.DictReader(open(csv_path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not _is_kernel(k, kernel_names) or r["Counter_Name"] != counter:
            continue
        seen[k] = seen.get(k, 0) + 1
        if seen[k] > skip:
            vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals)) if vals else None


def live_traffic(kernel, args):
    """HBM bytes per launch of `kernel`: two child runs of this workload under rocprofv3, one counter each
    (never combined with trace domains other than --kernel-trace).  None if the profiler is unavailable."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    out = {}
    warm, steps = 2, 6
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="vcm_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--child", "--res", str(args.res), "--scene", str(args.scene), "--algo", args.algo,
               "--steps", str(steps), "--warmup", str(warm)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, timeout=300)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            v = _pmc_mean(files[0], KERNEL_KEYS[kernel], counter, warm)
            if v is None:
                return None, "no %s rows for %s" % (counter, kernel)
            out[counter] = v
        except Exception as e:   # a profiler problem must never cost the benchmark line
            return None, "rocprofv3 --pmc %s: %r" % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # both counters are in KB; FETCH_SIZE reports half the bytes of wide reads on gfx950 (MI355X_MICROARCH.md, HBM)
    return int(2 * 1024 * out["FETCH_SIZE"] + 1024 * out["WRITE_SIZE"]), \
        "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child runs of this workload (%d launches each after %d warm-up), " \
        "2 x FETCH_SIZE + WRITE_SIZE" % (steps, warm)


def algorithmic_bytes(st, n_paths, n_cells):
    """SURVEY.md section 8(d), split per kernel"""
    nlv, A, Cc, K, S = st["lightVertices"], st["mergeAccepted"], st["mergeCandidates"], st["connections"], st["lightSplats"]
    gridv = st["gridVertices"]
    parts = {"light": 68 * nlv + 24 * S,                 # vertex store + splat RMW
             "grid": 20 * gridv + 8 * n_cells,           # position read, index write, histogram + scan
             "camera": 68 * K + 24 * n_paths,            # one LightVertex per connection + framebuffer RMW
             "merge": 52 * A + 16 * (Cc - A)}            # accepted photons 52 B, rejected candidates 16 B
    return sum(parts.values()), parts


def roofline_block(st, n_local, n_paths):
    """per-kernel times (HIP events / device clock stamps of the timed launches) against the algorithmic bytes"""
    b_iter, b_parts = algorithmic_bytes(st, n_local, n_paths)
    kernels = {"k_light_trace": (st["msLightKernel"], b_parts["light"]),
               "k_camera_trace": (st["msCameraKernel"], 24 * n_local),
               "k_connect_di+vc": (st["msConnectKernels"], b_parts["camera"] - 24 * n_local),
               "k_merge": (st["msMergeKernel"], b_parts["merge"])}
    dom = max(kernels, key=lambda k: kernels[k][0])
    dom_s, dom_b = kernels[dom][0] / 1e3, kernels[dom][1]
    achieved = dom_b / dom_s / 1e9 if dom_s > 0 else 0.0
    iter_s = st["msTotal"] / 1e3
    return dom, {
        "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
        "algorithmic_bytes_per_launch": int(dom_b), "kernel_ms": round(dom_s * 1e3, 3),
        "iteration_algorithmic_bytes": int(b_iter), "iteration_ms": round(iter_s * 1e3, 3),
        "iteration_achieved_GBs": round(b_iter / iter_s / 1e9, 2) if iter_s > 0 else 0.0,
        "iteration_frac": round(b_iter / iter_s / 1e9 / HBM_PEAK_GBS, 5) if iter_s > 0 else 0.0,
        "per_kernel": {k: {"ms": round(v[0], 3), "algorithmic_bytes": int(v[1]),
                           "GBs": round(v[1] / (v[0] / 1e3) / 1e9, 2) if v[0] > 0 else 0.0} for k, v in kernels.items()},
        "query_sort_ms": round(st["msQuerySort"], 3), "grid_build_ms_side_stream": round(st["msGrid"], 3),
        "scope": "rank 0 shard, mean over the timed iterations"}


def workload_name(scene, algo, res, replicas, first, last):
    if isinstance(scene, str):
        scene_txt = "tests/mesh_scenes.py bumpy_room(grid=%s), BVH," % scene.split(":")[1]
        return ("%s -a %s %dx%d maxPathLength 10 minPathLength 0 radiusFactor 0.003 radiusAlpha 0.75 seed 1234, iterations %d..%d timed"
                % (scene_txt, algo, res, res, first, last))
    return ("scene %d -a %s %dx%d maxPathLength 10 minPathLength 0 radiusFactor 0.003 radiusAlpha 0.75 seed 1234%s, "
            "iterations %d..%d timed" % (scene, algo, res, res, "..%d" % (1234 + replicas - 1) if replicas > 1 else "", first, last))


def cpu_reference(scene, algo_name, res, first_iter, n_iter, threads):
    """The unmodified reference on the host cores: `threads` renderers (seeds 1234 + i), one iteration each, radius
    indices cycling through the GPU's timed window; wall clock of the parallel loop (smallvcm.cxx:99-108)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from smallvcm_amd._abi import ALGO_BY_NAME, SCENE_CONFIGS
    if not os.path.exists(oracle_lib.REF_STOCK_SO):
        raise RuntimeError("oracle/_ref/libsmallvcm_ref_stock.so not built")
    idx = [first_iter + (i % n_iter) for i in range(threads)]
    _, wall = oracle_lib.ref_render_stock(SCENE_CONFIGS[scene], res, res, ALGO_BY_NAME[algo_name], iterations=threads,
                                          threads=threads, iteration_index=idx)
    return {"value": round(2.0 * res * res * threads / wall / 1e6, 4), "unit": "Mpaths/s", "cores": threads, "kind": "reference",
            "wall_s": round(wall, 2),
            "sample": "unmodified reference (oracle/_ref stock build: mt19937, glibc, its own render() loop), scene %d -a %s "
                      "%dx%d maxPathLength 10: %d renderers on %d threads, one RunIteration each, radius indices %d..%d (the "
                      "GPU's timed window), wall clock of the parallel loop: %.1f s" % (scene, algo_name, res, res, threads,
                                                                                     threads, first_iter, first_iter + n_iter - 1, wall)}


def cpu_port(scene, algo, res, iteration, budget_rows=8):
    """Oracle restatement on all host cores: full light pass + grid build, camera pass on every `budget_rows`-th pixel
    row, extrapolated to the full frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle, oracle
    from smallvcm_amd.renderer import cornell_scene
    cores = oracle().oracle_max_threads()
    o = Oracle(cornell_scene(scene, res, res), algo, threads=cores)
    n = res * res
    t0 = time.perf_counter()
    o.begin(iteration, 0, 10)
    o.trace_light()
    t1 = time.perf_counter()
    o.build_grid()
    t2 = time.perf_counter()
    o.trace_camera(row_stride=budget_rows)
    t3 = time.perf_counter()
    o.end()
    rows = len(range(0, res, budget_rows))
    t_full = (t1 - t0) + (t2 - t1) + (t3 - t2) * (res / rows)
    return {"value": round(2 * n / t_full / 1e6, 4), "unit": "Mpaths/s", "cores": cores, "kind": "port",
            "sample": "oracle/vcm_oracle.cpp, path-parallel, radius index %d: all %d light paths + grid build, camera paths of "
                      "every %dth pixel row, camera time scaled by %.2f; %.1f s light, %.1f s grid, %.1f s camera sample"
                      % (iteration, n, budget_rows, res / rows, t1 - t0, t2 - t1, t3 - t2)}


def timed_run(farm, steps, warmup, sync):
    """W warm-up steps, then exactly K steps bracketed by sync(); returns (seconds, mean statistics of the K steps)"""
    replicas = farm.replicas
    it = 0
    for _ in range(warmup):
        farm.run_iterations([[it * replicas + rid] for rid in farm.renderer_ids])
        it += 1
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        farm.run_iterations([[it * replicas + rid] for rid in farm.renderer_ids])
        it += 1
    sync()
    elapsed = time.perf_counter() - t0
    # Per-kernel times and counters of the K timed iterations, read AFTER the timed region: every iteration stamps
    # its kernel boundaries on the context's stream (HIP events + the device wall clock, vcm_api.hip: mark()) and the
    # library keeps the last 64 iterations' stamps and counters on the device (vcm_get_stats_at).
    hist = [farm.backend.stats_at(ago) for ago in range(min(steps, 64))]
    st = {k: sum(h[k] for h in hist) / len(hist) for k in hist[0]}
    return elapsed, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", type=int, default=2048)
    ap.add_argument("--scene", type=int, default=1)
    ap.add_argument("--algo", default="vcm")
    ap.add_argument("--shards", type=int, default=0,
                    help="GPUs that share one iteration (default: 2 when --gpus is even, else all)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="renderers taking turns on each group of --shards GPUs (default: 2 when shards > 1, else 1)")
    ap.add_argument("--cpu-baseline", default="reference", choices=["reference", "port", "none"],
                    help="reference: the unmodified reference in this run (minutes of host CPU at 2048^2) + the port; "
                         "port: only the oracle restatement (seconds); none")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="same as --cpu-baseline none")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the reference leg (default min(32, cores): "
                                                               "it saturates the host's memory system there)")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE's other single-GPU configs")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 child runs that measure HBM traffic")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)   # inner run under rocprofv3: GPU loop only
    args = ap.parse_args()
    if args.no_cpu_baseline:
        args.cpu_baseline = "none"

    import torch
    from smallvcm_amd._abi import ALGO_BY_NAME
    from smallvcm_amd.renderer import HipBackend, RenderFarm, cornell_scene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def make_farm(scene, algo_name, res, shards, inflight):
        if isinstance(scene, str):   # "mesh:<grid>": a version-2 scene (any number of primitives, BVH)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from mesh_scenes import bumpy_room
            sc = bumpy_room(grid=int(scene.split(":")[1]), resx=res, resy=res)
        else:
            sc = cornell_scene(scene, res, res)
        algo = ALGO_BY_NAME[algo_name]
        farm = RenderFarm(lambda seed, s, S: HipBackend(sc, algo, 0.003, 0.75, seed, device=local_rank, rank=s, world=S),
                          1234, rank, world, shards=shards, dist=dist, inflight=inflight)
        farm.set_path_lengths(0, 10)
        for rr in farm.renderers:   # setup, not a step: device buffers exist before the first (possibly timed) iteration
            rr.backend.reserve(10)
        return farm

    res, n_paths = args.res, args.res * args.res
    shards = args.shards if args.shards > 0 else (2 if world % 2 == 0 else world)
    farm = make_farm(args.scene, args.algo, res, shards, args.inflight if args.inflight > 0 else None)
    replicas = farm.replicas
    elapsed, st = timed_run(farm, args.steps, args.warmup, sync)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if args.child:
        farm.close()
        return
    fb = farm.framebuffer()   # includes the framebuffer reduce over ranks (mean image, smallvcm.cxx:116-142)
    n_local = farm.backend.count
    inflight_used = farm.inflight
    farm.close()
    # N > 1: the pure north_star decomposition next to the default one, same steps: ONE renderer whose paths are sharded
    # over all N GPUs (all-gather of the light vertices over the whole node every iteration, nothing to hide it behind)
    strong = None
    if world > 1 and not args.child and (shards != world or inflight_used != 1):
        f3 = make_farm(args.scene, args.algo, res, world, 1)
        e3, s3 = timed_run(f3, args.steps, args.warmup, sync)
        t = torch.tensor([e3], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e3 = float(t.item())
        f3.close()
        strong = {"value": round(2.0 * n_paths * args.steps / e3 / 1e6, 3), "unit": "Mpaths/s", "scaling": "strong",
                  "ms_per_step": round(e3 / args.steps * 1e3, 3), "paths_per_step": 2 * n_paths,
                  "parallelism": "1 renderer on %d path-index shards (RCCL all-gather of the light vertices every iteration, "
                                 "framebuffer all-reduce at read-out), 1 renderer in flight" % world,
                  "iteration_ms_rank0": round(s3["msTotal"], 3)}

    if rank == 0:
        value = 2.0 * n_paths * args.steps * replicas / elapsed / 1e6
        dom, roof = roofline_block(st, n_local, n_paths)
        headline = (args.scene, args.algo, res) == (1, "vcm", 2048)
        out = {
            "metric": "Mpaths/sec (light+camera), %s scene %d at %d^2" % (args.algo.upper(), args.scene, res),
            "value": round(value, 3), "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            # the default decomposition keeps the work per GPU fixed as N grows (one renderer-iteration per GPU and step:
            # 1 renderer at N = 1, N renderers on N / 2 pairs at N > 1); --shards N at N > 1 is the strong one
            "scaling": "strong" if (replicas == 1 and world > 1) else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (reference's built-in Cornell box scene %d)" % args.scene,
            "config": {"workload": workload_name(args.scene, args.algo, res, replicas, args.warmup * replicas,
                                                 (args.warmup + args.steps) * replicas - 1),
                       "baseline_config": "C4 at 1 GPU (BASELINE.json metric)" if headline and world == 1 else
                                          ("C4" if headline else "other"),
                       "paths_per_step": 2 * n_paths * replicas,
                       "parallelism": "%d renderer(s) (iteration-parallel, smallvcm.cxx:61-108), each on %d path-index shard(s) "
                                      "(RCCL all-gather of light vertices), %d renderer(s) in flight per GPU group"
                                      % (replicas, shards, inflight_used),
                       "merge_kernel": os.environ.get("SMALLVCM_AMD_MERGE", "walk")},
            "roofline": roof,
            "counters": {k: int(st[k]) for k in ("lightVertices", "gridVertices", "mergeQueries", "mergeCandidates",
                                                 "mergeAccepted", "connections", "lightSplats", "lightRays", "cameraRays",
                                                 "shadowRays")},
            "image_mean": [round(float(x), 5) for x in fb.mean(axis=(0, 1))],
        }
        if strong is not None:
            out["strong_decomposition"] = strong
        if world == 1:
            traffic, src = (None, "disabled (--no-traffic)") if args.no_traffic else live_traffic(dom, args)
            if traffic is None:
                rec, path = recorded_traffic(dom)
                if rec is not None:
                    traffic, src = rec, "recorded: %s (same kernel sources, hash %s); live measurement unavailable: %s" % (
                        path, kernel_source_hash(), src)
                else:
                    src = "null: %s; no profiles/*_traffic.json for kernel sources %s" % (src, kernel_source_hash())
            roof["traffic"] = traffic
            roof["traffic_source"] = src
            if traffic:
                roof["achieved_traffic_GBs"] = round(traffic / (roof["kernel_ms"] / 1e3) / 1e9, 2)
                roof["frac_traffic"] = round(traffic / (roof["kernel_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
                roof["traffic_over_algorithmic"] = round(traffic / max(roof["algorithmic_bytes_per_launch"], 1), 4)
        if world == 1 and headline and not args.no_configs:
            cfgs = []
            for name, scene, algo_name, r, nfl in OTHER_CONFIGS:
                try:
                    f2 = make_farm(scene, algo_name, r, 1, nfl)
                    e2, s2 = timed_run(f2, args.steps, args.warmup, sync)
                    nl = f2.backend.count
                    f2.close()
                    _, roof2 = roofline_block(s2, nl, r * r)
                    if nfl > 1:
                        roof2["scope"] = "first of the %d renderers; its kernels share the GPU with the others', so per-kernel " \
                                         "times are longer than alone" % nfl
                    cfgs.append({"name": name, "workload": workload_name(scene, algo_name, r, nfl, args.warmup * nfl,
                                                                         (args.warmup + args.steps) * nfl - 1),
                                 "renderers_in_flight": nfl,
                                 "value": round(2.0 * r * r * args.steps * nfl / e2 / 1e6, 3), "unit": "Mpaths/s",
                                 "ms_per_step": round(e2 / args.steps * 1e3, 3), "steps": args.steps, "warmup": args.warmup,
                                 "paths_per_step": 2 * r * r * nfl,
                                 "roofline": roof2,
                                 "counters": {k: int(s2[k]) for k in ("lightVertices", "mergeQueries", "mergeCandidates",
                                                                      "mergeAccepted", "connections", "lightSplats")}})
                except Exception as e:
                    cfgs.append({"name": name, "error": repr(e)})
            out["configs"] = cfgs
        if world == 1 and args.cpu_baseline != "none":
            base = None
            try:
                port = cpu_port(args.scene, ALGO_BY_NAME[args.algo], res, args.warmup)
            except Exception as e:   # the baseline is a report, never a reason to lose the GPU number
                port = {"error": repr(e)}
            if args.cpu_baseline == "reference":
                try:
                    cores = os.cpu_count() or 1
                    threads = args.cpu_threads if args.cpu_threads > 0 else min(32, cores)
                    base = cpu_reference(args.scene, args.algo, res, args.warmup, args.steps, threads)
                    base["host_cores"] = cores
                    base["port"] = port
                except Exception as e:
                    port["reference_error"] = repr(e)
            out["cpu_baseline"] = base if base is not None else port
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
