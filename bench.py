#!/usr/bin/env python
"""bench.py -- Mpaths/s of the MI355X VCM integrator on BASELINE.json's
headline workload: scene 1 (glossy floor + large mirror sphere + ceiling light
box), algorithm VCM, 2048x2048, maxPathLength 10 (reference: VertexCM::RunIteration,
src/vertexcm.hxx:284-548).

A "step" is one RunIteration = N light sub-paths + N camera sub-paths,
N = 2048*2048.  Mpaths/s = 2*N*steps / seconds / 1e6.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by torch.distributed.run (one rank per GPU): paths are
sharded by index, the light-vertex merge records are all-gathered (RCCL) each
iteration and the framebuffers are summed once at the end.  Total work is
fixed as N grows ("strong" scaling).

The JSON line also carries
  roofline      for the kernel with the largest mean HIP-event time: its share of
                the algorithmic bytes (SURVEY.md section 8(d) formula, split per
                kernel in DESIGN.md, evaluated with the run's own counters)
                / its time / 8 TB/s; per-kernel and whole-iteration figures too;
  cpu_baseline  the oracle port (oracle/vcm_oracle.cpp, OpenMP over paths, all
                host cores) on a bounded sample of the same workload; rank 0,
                N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes(st, n_paths, n_cells):
    """SURVEY.md section 8(d).  Returns (whole iteration, camera kernel share)."""
    nlv, A, Cc, K, S = st["lightVertices"], st["mergeAccepted"], st["mergeCandidates"], st["connections"], st["lightSplats"]
    gridv = st["gridVertices"]
    parts = {"light": 68 * nlv + 24 * S,                 # vertex store + splat RMW
             "grid": 20 * gridv + 8 * n_cells,           # position read, index write, histogram + scan
             "camera": 68 * K + 24 * n_paths,            # one LightVertex per connection + framebuffer RMW
             "merge": 52 * A + 16 * (Cc - A)}            # accepted photons 52 B, rejected candidates 16 B
    return sum(parts.values()), parts


def cpu_baseline(res, budget_rows=8):
    """Oracle port on the host cores: full light pass + grid build, camera pass on
    every `budget_rows`-th pixel row, extrapolated to the full frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle, oracle
    from smallvcm_amd.renderer import cornell_scene
    cores = oracle().oracle_max_threads()
    sc = cornell_scene(1, res, res)
    o = Oracle(sc, 4, threads=cores)
    n = res * res
    t0 = time.perf_counter()
    o.begin(0, 0, 10)
    o.trace_light()
    t1 = time.perf_counter()
    o.build_grid()
    t2 = time.perf_counter()
    o.trace_camera(row_stride=budget_rows)
    t3 = time.perf_counter()
    o.end()
    rows = len(range(0, res, budget_rows))
    t_full = (t1 - t0) + (t2 - t1) + (t3 - t2) * (res / rows)
    return {
        "value": round(2 * n / t_full / 1e6, 4), "unit": "Mpaths/s", "cores": cores, "kind": "port",
        "sample": "scene 1 vcm %dx%d iteration 0: all %d light paths + grid build, camera paths of every %dth pixel "
                  "row (%d of %d rows), camera time scaled by %.2f; measured %.1f s light, %.1f s grid, %.1f s camera sample"
                  % (res, res, n, budget_rows, rows, res, res / rows, t1 - t0, t2 - t1, t3 - t2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", type=int, default=2048)
    ap.add_argument("--scene", type=int, default=1)
    ap.add_argument("--algo", default="vcm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from smallvcm_amd._abi import ALGO_BY_NAME
    from smallvcm_amd.renderer import HipBackend, ShardedVertexCM, cornell_scene

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

    res = args.res
    n_paths = res * res
    algo = ALGO_BY_NAME[args.algo]
    sc = cornell_scene(args.scene, res, res)
    backend = HipBackend(sc, algo, 0.003, 0.75, 1234, device=local_rank, rank=rank, world=world)
    r = ShardedVertexCM(backend, rank, world)
    r.mMaxPathLength, r.mMinPathLength = 10, 0

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        r.RunIteration(it)
        it += 1
    sync()
    t0 = time.perf_counter()
    cam_ms, light_ms, total_ms, merge_ms, sort_ms, conn_ms = [], [], [], [], [], []
    st = None
    for _ in range(args.steps):
        r.RunIteration(it)
        it += 1
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel times / counters of a few extra (untimed) iterations, HIP events on the ctx stream
    for _ in range(min(3, max(1, args.steps))):
        r.RunIteration(it)
        it += 1
        st = backend.stats()
        cam_ms.append(st["msCameraKernel"])
        merge_ms.append(st["msMergeKernel"])
        sort_ms.append(st["msQuerySort"])
        conn_ms.append(st["msConnectKernels"])
        light_ms.append(st["msLightKernel"])
        total_ms.append(st["msTotal"])

    fb = r.framebuffer_sum()   # includes the framebuffer reduce over ranks

    if rank == 0:
        value = 2.0 * n_paths * args.steps / elapsed / 1e6
        b_iter, b_parts = algorithmic_bytes(st, backend.count, n_paths)
        mean = lambda v: sum(v) / len(v)
        kernels = {"k_light_trace": (mean(light_ms), b_parts["light"]),
                   "k_camera_trace": (mean(cam_ms), 24 * backend.count),
                   "k_connect_di+vc": (mean(conn_ms), b_parts["camera"] - 24 * backend.count),
                   "k_merge_lane": (mean(merge_ms), b_parts["merge"])}
        dom = max(kernels, key=lambda k: kernels[k][0])
        dom_s, dom_b = kernels[dom][0] / 1e3, kernels[dom][1]
        achieved = dom_b / dom_s / 1e9 if dom_s > 0 else 0.0
        iter_s = mean(total_ms) / 1e3
        out = {
            "metric": "Mpaths/sec (light+camera), VCM scene 1 at 2048^2",
            "value": round(value, 3), "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (reference's built-in Cornell box scene %d)" % args.scene,
            "config": {"workload": "scene %d -a %s %dx%d maxPathLength 10 minPathLength 0 radiusFactor 0.003 radiusAlpha 0.75 "
                                   "seed 1234, iterations %d..%d timed" % (args.scene, args.algo, res, res, args.warmup,
                                                                           args.warmup + args.steps - 1),
                       "paths_per_step": 2 * n_paths, "parallelism": "path-index shards x%d" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "algorithmic_bytes_per_launch": int(dom_b), "kernel_ms": round(dom_s * 1e3, 3),
                         "iteration_algorithmic_bytes": int(b_iter),
                         "iteration_ms": round(iter_s * 1e3, 3),
                         "iteration_achieved_GBs": round(b_iter / iter_s / 1e9, 2),
                         "iteration_frac": round(b_iter / iter_s / 1e9 / HBM_PEAK_GBS, 5),
                         "per_kernel": {k: {"ms": round(v[0], 3), "algorithmic_bytes": int(v[1]),
                                            "GBs": round(v[1] / (v[0] / 1e3) / 1e9, 2) if v[0] > 0 else 0.0}
                                        for k, v in kernels.items()},
                         "query_sort_ms": round(mean(sort_ms), 3),
                         "connect_kernels_ms": round(mean(conn_ms), 3),
                         "light_kernel_ms": round(sum(light_ms) / len(light_ms), 3),
                         "merge_kernel_ms": round(sum(merge_ms) / len(merge_ms), 3),
                         "scope": "rank 0 shard"},
            "counters": {k: int(st[k]) for k in ("lightVertices", "gridVertices", "mergeQueries", "mergeCandidates",
                                                 "mergeAccepted", "connections", "lightSplats", "lightRays", "cameraRays",
                                                 "shadowRays")},
            "image_mean": [round(float(x), 5) for x in (fb / it).mean(axis=(0, 1))],
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(res)
            except Exception as e:   # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
