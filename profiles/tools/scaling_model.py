#!/usr/bin/env python
"""What ONE rank of north_star's decomposition costs per iteration at S = 1, 2, 4, 8 GPUs -- measured kernel by kernel on
one MI355X -- and what that projects to with an xGMI model for the exchange (run on the GPU box, from the repo root):

    python profiles/tools/scaling_model.py <tag> [--res 2048] [--iters 12] [--warmup 3] [--shards 1,2,4,8]

For every S the C++ farm runs ONE renderer on S path-index shards with the in-process stand-in for RCCL (S rank threads
on this one GPU) in its measurement mode SMALLVCM_AMD_FARM_SERIALIZE=1: the ranks take turns on the device, so a kernel
trace (rocprofv3 --kernel-trace) holds the time every kernel takes for a rank's shard with the chip to itself -- what a
rank of an S-GPU node sees.  Per kernel: mean duration x launches per rank and iteration (warm-up dispatches dropped).

The model (stated, not measured -- the pool has no multi-GPU box):
  busy(S)     = device-busy time per rank and iteration: the union of its kernels' intervals in the trace (its own streams
                overlap inside it as they would on a GPU of its own; S = 1 is the plain iteration)
  exchange    = (S - 1) x slab bytes received per rank / bandwidth; it overlaps the camera pass (K3 + K3b), the rest is exposed
  T(S)        = busy + max(0, exchange - overlap window) + host_sync (0.08 ms: the 32-byte read-back, the 7-number exchange)
with two bandwidths: `rccl` = 300 GB/s bus bandwidth (ring-class all-gather: per-link bound, 2 of 7 links busy) and `direct`
= 7 links x 153 GB/s x 0.75 = 800 GB/s (every rank writing its slab to its 7 peers at once); "free exchange" = busy + sync.
(main / beside / gridmrg = sums of kernel durations by stream, for orientation: they overlap, busy is what counts.)
-> gpurun_out/<tag>_scaling.json/.txt"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MAIN = ["k_light_trace", "k_scan_tile_sums", "k_scan_apply", "k_compact_records", "k_bbox_finalize", "k_bbox", "k_grid_init", "k_cell_count",
        "k_cell_scatter", "k_cell_rank_pack", "k_cell_rank_gather", "k_camera_trace", "k_connect_di", "k_query_scatter", "k_query_count",
        "k_merge_walk", "k_merge_pairs", "k_resolve", "k_zero_ranges", "k_set_counts", "k_stamp_many", "k_set_bbox"]
BESIDE = ["k_connect_camera", "k_splat_scatter", "k_splat_apply", "k_splat_apply_long", "k_connect_vc", "k_note_grid_vertices"]
GRID_MERGE = ["k_grid_merge_blocks"]
OVERLAP_WINDOW = ["k_camera_trace", "k_connect_di"]   # what the exchange hides behind


def base(name):
    n = name.split("(")[0].replace("void ", "").split("<")[0]
    return n.replace("vcm::", "")


def run(tag, S, res, iters, warm):
    d = os.path.join(ROOT, "gpurun_out", "%s_sm%d" % (tag, S))
    exe = os.path.join(ROOT, "smallvcm_amd", "host", "vcm_render")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", exe, "-s", "1", "-a", "vcm", "-i", str(iters), "--warmup", str(warm),
           "--res", str(res), str(res), "--gpus", str(S), "--shards", str(S), "--inflight", "1", "--collectives", "threads", "--same-window", "--json"]
    env = dict(os.environ, TMPDIR="/tmp", SMALLVCM_AMD_FARM_SERIALIZE="1")
    r = subprocess.run(cmd, cwd=os.path.dirname(exe), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    line = [l for l in r.stdout.decode("utf-8", "replace").splitlines() if l.startswith("{")]
    host = json.loads(line[-1]) if line else {}
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if r.returncode != 0 or not f:
        raise RuntimeError("S=%d: rc %d %s" % (S, r.returncode, r.stderr.decode("utf-8", "replace")[-400:]))
    rows = sorted(csv.DictReader(open(f[0])), key=lambda x: int(x["Start_Timestamp"]))
    per = {}
    for x in rows:
        per.setdefault(base(x["Kernel_Name"]), []).append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6)
    # device-busy time per rank and iteration: the union of the kernels' intervals (the stand-in's copies excluded), timed
    # iterations only.  The ranks take turns, so the union never holds two ranks' kernels at once; what overlaps inside it is a
    # rank's OWN streams (K3 beside its splats and its cell sort), as on a GPU of its own.
    ks = [(int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in rows if not base(x["Kernel_Name"]).startswith("__amd")]
    resolve_ends = [int(x["End_Timestamp"]) for x in rows if base(x["Kernel_Name"]) == "k_resolve"]
    t_first = resolve_ends[S * warm - 1] if warm > 0 and len(resolve_ends) >= S * warm else 0   # after the last warm-up iteration of the last rank
    busy, cur_s, cur_e = 0, None, None
    for a, b in ks:
        if b <= t_first:
            continue
        a = max(a, t_first)
        if cur_e is None or a > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    if cur_e is not None:
        busy += cur_e - cur_s
    busy_ms = busy / 1e6 / float(S * iters)
    out = {"__busy__": {"ms": busy_ms, "launches_per_rank_iteration": 0}}
    for k, v in per.items():
        n_iter = S * (iters + warm)
        launches = len(v) / float(n_iter)
        timed = v[int(len(v) * warm / float(iters + warm)):]
        out[k] = {"ms": sum(timed) / len(timed) * launches, "launches_per_rank_iteration": round(launches, 2)}
    return out, host


def main():
    tag = sys.argv[1]
    a = sys.argv[2:]
    opt = lambda k, d: a[a.index(k) + 1] if k in a else d   # noqa: E731
    res, iters, warm = int(opt("--res", "2048")), int(opt("--iters", "12")), int(opt("--warmup", "3"))
    shards = [int(x) for x in opt("--shards", "1,2,4,8").split(",")]
    table, hosts = {}, {}
    for S in shards:
        table[S], hosts[S] = run(tag, S, res, iters, warm)
    kernels = sorted({k for S in shards for k in table[S] if k != "__busy__"}, key=lambda k: -table[shards[0]].get(k, {"ms": 0})["ms"])
    n_cells = res * res
    stats1 = hosts[shards[0]].get("stats", {})
    n_lv = float(stats1.get("lightVertices", 2.126 * res * res))   # the run's own count where the host reports it
    lines = ["scene 1 VCM %dx%d, one MI355X, S rank threads taking turns (SMALLVCM_AMD_FARM_SERIALIZE=1): ms per rank and iteration" % (res, res),
             "%-28s" % "kernel" + "".join("%10s" % ("S=%d" % S) for S in shards)]
    for k in kernels:
        lines.append("%-28s" % k + "".join("%10.3f" % table[S].get(k, {"ms": 0.0})["ms"] for S in shards))
    model = {}
    for S in shards:
        t = table[S]
        g = lambda names: sum(t[k]["ms"] for k in names if k in t)   # noqa: E731
        other = sum(v["ms"] for k, v in t.items() if k not in MAIN + BESIDE + GRID_MERGE and not k.startswith("__"))
        busy_ms = t["__busy__"]["ms"]
        main_ms, beside_ms, merge_ms, window = g(MAIN) + other, g(BESIDE), g(GRID_MERGE), g(OVERLAP_WINDOW)
        K = 16
        while K * 2 <= 4096 // max(S, 1) and K < 1024:
            K *= 2
        slab = (n_lv / S) * 52.0 + 4.0 * (n_cells / K + 1)
        recv = (S - 1) * slab
        m = {"busy_ms": round(busy_ms, 3), "main_ms": round(main_ms, 3), "beside_ms": round(beside_ms, 3), "grid_merge_ms": round(merge_ms, 3), "overlap_window_ms": round(window, 3),
             "serial_sum_ms": round(main_ms + beside_ms + merge_ms, 3), "slab_MB": round(slab / 1e6, 1), "received_MB": round(recv / 1e6, 1)}
        for name, bw in (("rccl_300GBs", 300e9), ("direct_800GBs", 800e9)):
            x = recv / bw * 1e3
            sync = 0.08 if S > 1 else 0.0   # one device -> host read of 32 bytes + the 7-number exchange + relaunch latency
            # busy = everything the rank's GPU does for the iteration, its own overlaps included (the grid merge is in it);
            # what the exchange adds is the part of it the camera pass does not cover, and the host round trip
            T = busy_ms + max(0.0, x - window) + sync
            m[name] = {"exchange_ms": round(x, 3), "exposed_ms": round(max(0.0, x - window), 3), "host_sync_ms": sync, "iteration_ms": round(T, 3)}
        model[S] = m
    base_T = model[shards[0]]["rccl_300GBs"]["iteration_ms"]
    lines.append("")
    lines.append("busy = device-busy ms per rank and iteration (union of its kernels; S = 1: the plain iteration); T = busy + exposed exchange + host sync")
    lines.append("%-4s %8s %8s %8s %8s %8s | %30s | %30s | %s" % ("S", "busy", "main", "beside", "gridmrg", "recv MB", "rccl 300 GB/s: xchg exposed T  x", "direct 800 GB/s: xchg exposed T  x", "free exchange: x"))
    for S in shards:
        m = model[S]
        r1, r2 = m["rccl_300GBs"], m["direct_800GBs"]
        m["rccl_300GBs"]["speedup"] = round(base_T / r1["iteration_ms"], 2)
        m["direct_800GBs"]["speedup"] = round(base_T / r2["iteration_ms"], 2)
        m["free_exchange_speedup"] = round(base_T / (m["busy_ms"] + (0.08 if S > 1 else 0.0)), 2)
        lines.append("%-4d %8.3f %8.3f %8.3f %8.3f %8.1f | %8.3f %7.3f %7.3f %5.2f | %8.3f %7.3f %7.3f %5.2f | %5.2f" % (
            S, m["busy_ms"], m["main_ms"], m["beside_ms"], m["grid_merge_ms"], m["received_MB"], r1["exchange_ms"], r1["exposed_ms"], r1["iteration_ms"],
            m["rccl_300GBs"]["speedup"], r2["exchange_ms"], r2["exposed_ms"], r2["iteration_ms"], m["direct_800GBs"]["speedup"], m["free_exchange_speedup"]))
    txt = "\n".join(lines)
    print(txt)
    out = os.path.join(ROOT, "gpurun_out", "%s_scaling" % tag)
    open(out + ".txt", "w").write(txt + "\n")
    json.dump({"res": res, "iters": iters, "warmup": warm, "kernels_ms": {str(S): table[S] for S in shards}, "model": {str(S): model[S] for S in shards},
               "host_lines": {str(S): hosts[S] for S in shards}}, open(out + ".json", "w"), indent=1)


if __name__ == "__main__":
    main()
