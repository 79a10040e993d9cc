#!/usr/bin/env python
"""bench.py -- Mpaths/s of the MI355X VCM integrator (reference: VertexCM::RunIteration,
src/vertexcm.hxx:284-548) on BASELINE.json's workloads.

  python bench.py [--gpus N] [--steps K] [--warmup W]            headline: scene 1, VCM, 2048x2048, maxPathLength 10
  python bench.py --scene 3 --algo vcm --res 1024                any other built-in configuration (named in the line)

A "step" is one RunIteration = N light sub-paths + N camera sub-paths, N = res*res.
Mpaths/s = 2*N*steps / seconds / 1e6.

STDOUT IS ONE SHORT JSON LINE (< 4 KB: the driver keeps a tail of stdout and parses its last line; round 4's 20 KB line did
not survive that).  Everything else the run measured -- per-kernel blocks, counters, the config blocks, CPU samples --
goes to bench_detail.json (repository root, copied to gpurun_out/ when that exists); the line names the file.

  value         whole-job Mpaths/s of the timed region (K steps after W warm-up steps, barrier + synchronize on both sides,
                max over ranks); `metric` and `config.workload` name what was actually run;
  roofline      for the kernel with the largest mean time in the timed region (device-clock stamps / HIP events on the
                stream the kernel runs on, vcm_api.hip mark()).  `frac` is a fraction (<= 1) or null: with counters it is
                max(measured HBM traffic / kernel time / 8 TB/s, class-weighted VALU issue time / kernel time) and `bound`
                says which; `traffic` = HBM bytes per launch from TWO CHILD RUNS of this file under `rocprofv3 --pmc
                FETCH_SIZE` / `--pmc WRITE_SIZE` (counter passes on their own; FETCH_SIZE doubled per MI355X_MICROARCH.md
                section HBM), a third pass collects the SQ VALU group; without a profiler the recording
                profiles/*_counters_<config>.json is used if it was made on the same kernel sources (hash compared), else
                frac is null.  SURVEY 8(d)'s gather model stays beside it as frac_algorithmic (it exceeds 1 for the
                cell-sorted merge and bounds nothing); iteration_frac = the whole iteration's algorithmic bytes / time / peak;
  configs       N = 1, headline run only: BASELINE.json's other single-GPU configurations (C1 scene 1 vcm 512^2,
                C2 scene 3 vcm 1024^2, C3 scene 1 bpm 2048^2) timed the same way, C1-C3 with their own counter passes;
                one summary each in the line, the blocks in the detail file;
  cpu_baseline  rank 0, N = 1: the UNMODIFIED reference (oracle/_ref/libsmallvcm_ref_stock.so = /root/reference/src
                built with the flags of its Makefile:5, its own render() loop: one renderer per thread, iterations
                dealt out, smallvcm.cxx:52-151) timed on this box's host cores in this run, wall clock, on the same
                resolution and the same iteration (= radius) window as the GPU; `port` = the oracle restatement,
                path-parallel over all cores, on a bounded sample.  Baseline only.

N = 1 and N > 1 are timed by the SAME host: the C++ farm (one rank thread at N = 1; `host_cross_check` repeats the headline
through the Python / ctypes loop).  N > 1 DEFAULT = north_star's decomposition: ONE renderer whose light paths and pixels
are sharded by index over all N GPUs; every iteration each rank sorts its own light vertices by hash cell, the ranks
all-gather them (RCCL), place them into the identical grid, merge their own pixels; one framebuffer all-reduce at
read-out ("scaling": "strong").  The replica hybrid of round 4 (pairs of GPUs x 2 renderers in flight = mostly the
reference's iteration parallelism, smallvcm.cxx:61-72; "weak") is timed beside it as hybrid_decomposition
(--shards / --inflight select any other shape).  Both ways of launching work:
  python bench.py --gpus N                                      one process, N rank threads (ncclCommInitRank x N in a group)
  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N     one process per GPU; rank 0 makes the RCCL ids,
                                                                a gloo broadcast ships them, the farm does the rest
--collectives threads replaces RCCL by an in-process stand-in so that several ranks can share ONE GPU (tests).
DESIGN.md section 6 has the reasoning.  Python here only parses arguments and prints: every iteration, barrier and
collective of the timed region is issued by the C++ host.
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
# VALU issue roofline.  MI355X_MICROARCH.md "Wave scheduling": 4 SIMD-32 per CU, a wave64 instruction issues over 2 cycles
# (157.3 TFLOP/s fp32 = 1024 SIMDs x 32 lanes x 2 x 2.4 GHz).  What a SIMD actually sustains was measured on this chip by
# profiles/tools/valu_bench.hip (profiles/archive/r02h_valu_bench.txt, >= 2 waves per SIMD, in cycles of the nominal clock): plain
# fp32 / int32 2.7, binary64 add / mul / fma 5.2, transcendentals (rcp, sqrt, ...) 8.4.  The roof weighs the run's
# instruction mix with THOSE costs (rounds 2-4 used a flat 4 cycles, which is no peak: kernels printed 1.04-1.12 of it):
#   issue_ms = (plain x 2.7 + f64 x 5.2 + trans x 8.4) / (1024 SIMDs x 2.4 GHz);  frac = issue_ms / kernel_ms  (<= 1)
# binary64 fma has no counter in the group collected here and is priced as plain: issue_ms is a lower bound, frac errs low.
VALU_SIMDS, VALU_CLK_HZ = 1024, 2.4e9
VALU_CYCLES = {"plain": 2.7, "f64": 5.2, "trans": 8.4}
# The same mix at the costs MI355X_MICROARCH.md's SIMD-32 model gives (a wave64 fp32 / int32 instruction over 2 cycles, "Per-instruction
# cycle constants": v_fma_f32 2 cyc; binary64 at half that rate, transcendentals at a quarter): `valu_frac_guide` beside `valu_frac`
# (the measured 2.7 is this builder's own number, the guide's 2 is the hardware's)
VALU_CYCLES_GUIDE = {"plain": 2.0, "f64": 4.0, "trans": 8.0}
LINE_LIMIT = 4096        # the driver keeps a tail of stdout: the JSON line must fit it with room to spare
# kernel-name PREFIXES as rocprofv3 prints them (the ray-casting kernels are templates over the kind of scene:
# "vcm::k_camera_trace<1, vcm::SceneList>")
KERNEL_KEYS = {"k_light_trace": ["vcm::k_light_trace<1"], "k_camera_trace": ["vcm::k_camera_trace<1"],
               "k_connect_di+vc": ["vcm::k_connect_di", "vcm::k_connect_vc"],
               "k_merge": ["vcm::k_merge_pairs", "vcm::k_merge_walk"]}
# How a kernel reads memory decides what FETCH_SIZE means for it (profiles/tools/fetch_calib.hip measures the factor
# per pattern on the GPU box; profiles/fetch_calib.json holds the result).  Until a class is calibrated the guide's
# factor for wide streaming reads (2) is used and the figure is marked "uncalibrated".
KERNEL_FETCH_CLASS = {"k_light_trace": "stream", "k_camera_trace": "stream", "k_connect_di+vc": "gather80", "k_merge": "runs",
                      "k_resolve": "gather16", "k_connect_camera": "gather80", "grid_build": "gather16", "splat": "gather16"}


def _is_kernel(name, prefixes):
    return any(name.startswith(p) for p in prefixes)


def _base(name):
    """'void vcm::k_camera_trace<1, vcm::SceneList>(...)' -> 'vcm::k_camera_trace'"""
    n = name.split("(")[0].replace("void ", "")
    return n.split("<")[0]


KERNEL_SOURCES = ["vcm_api.hip", "vcm_kernels.h", "vcm_core.h", "vcm_math.h", "detmath.h", "philox.h", "Makefile"]
# BASELINE.json configs that fit one GPU, besides the headline (C4 at one GPU)
# (name, scene, algorithm, resolution, renderers in flight): "x4" = four renderers (seeds 1234..1237, the reference's
# iteration-parallel threads, smallvcm.cxx:61-108) taking turns on the one GPU, their iterations overlapping
OTHER_CONFIGS = [("C1", 1, "vcm", 512, 1), ("C1x4", 1, "vcm", 512, 4), ("C2", 3, "vcm", 1024, 1), ("C2x2", 3, "vcm", 1024, 2),
                 ("C3", 1, "bpm", 2048, 1), ("C4x2", 1, "vcm", 2048, 2),
                 # not a BASELINE config: SURVEY 8(f) #3, a scene beyond the Cornell boxes: a scene FILE (OBJ + MTL +
                 # .vcmscene, vcm_scene_load) of 10 380 primitives -- a tessellated height-field floor -- through the BVH
                 ("M1", "file:tests/scenes/bumpy_room.vcmscene", "vcm", 1024, 1)]


def kernel_source_hash():
    """identifies the kernels a traffic figure belongs to: sha256 over the HIP sources and build flags"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "smallvcm_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def fetch_factor(kernel):
    """(factor, source): what FETCH_SIZE has to be multiplied with for this kernel's read pattern"""
    cls = KERNEL_FETCH_CLASS.get(kernel, "stream")
    try:
        cal = json.load(open(os.path.join(ROOT, "profiles", "fetch_calib.json")))
        f = cal["classes"][cls]
        return float(f["factor"]), "%s: x%.2f (profiles/fetch_calib.json, pattern %s)" % (cls, f["factor"], f["pattern"])
    except Exception:
        return 2.0, "%s: x2 (MI355X_MICROARCH.md section HBM, wide streaming reads; this pattern uncalibrated)" % cls


def recorded_traffic(kernel):
    """profiles/*_traffic.json collected on THESE kernel sources, newest first; (bytes, file) or (None, None)"""
    want = kernel_source_hash()
    fac, _ = fetch_factor(kernel)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("kernel_src_sha16") != want:
                continue
            t = d["kernels"]
            tot = sum(v["fetch_bytes_x2"] / 2.0 * fac + v["write_bytes"] for k, v in t.items() if _is_kernel(k, KERNEL_KEYS[kernel]))
            if tot > 0:
                return int(tot), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def counters_file(tag, config_name):
    return os.path.join(ROOT, "profiles", "%s_counters_%s.json" % (tag, config_name))


def save_counters(tag, config_name, workload, counters, note):
    """a live counter set (the three --pmc child runs) as a recording for runs without a profiler: keyed by configuration
    and by the hash of the kernel sources it was collected on"""
    rec = {"config": config_name, "workload": workload, "kernel_src_sha16": kernel_source_hash(), "note": note,
           "units": "FETCH_SIZE / WRITE_SIZE in KB per dispatch (FETCH_SIZE before its calibration factor), SQ_* as counted, _us_* = "
                    "mean kernel duration under the profiler (serialised dispatches), _n_* = dispatches",
           "kernels": counters}
    with open(counters_file(tag, config_name), "w") as f:
        json.dump(_clean(rec), f, indent=1)
        f.write("\n")


def recorded_counters(config_name):
    """profiles/*_counters_<config>.json collected on THESE kernel sources, newest first -> (counters, file) or (None, None)"""
    want = kernel_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters_%s.json" % config_name)), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("kernel_src_sha16") == want and d.get("config") == config_name and d.get("kernels"):
                return d["kernels"], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def _pmc_table(csv_path, skip):
    """{kernel base name: {counter: mean per dispatch, "_us": mean duration, "_n": dispatches per iteration}} with the first
    `skip` dispatches ... of each kernel dropped (a kernel launched k times per iteration: the first skip * k)"""
    rows = list(csv.DictReader(open(csv_path)))
    per = {}
    for r in rows:
        k = _base(r["Kernel_Name"])
        if not k.startswith("vcm::"):
            continue
        per.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(
            (float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    out = {}
    for k, d in per.items():
        o = {}
        for c, v in d.items():
            n = len(v)
            v = v[int(n * skip):] if n * skip >= 1 else v   # skip = fraction of the dispatches that are warm-up
            o[c] = sum(x[0] for x in v) / len(v)
            o["_us"] = sum(x[1] for x in v) / len(v)
            o["_n"] = n
        out[k] = o
    return out


def live_counters(args):
    """Three child runs of this workload under rocprofv3, one counter group each (never combined with trace domains
    other than --kernel-trace): FETCH_SIZE, WRITE_SIZE, and the VALU group.  -> ({kernel base name: {...}}, note) or
    (None, reason).  A profiler problem must never cost the benchmark line."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    warm, steps = args.warmup, args.steps   # the SAME iteration window as the timed run: traffic, VALU counters and kernel_ms agree
    # per group: the counter lists to try in turn (the second VALU list is rounds 2-4's, without the instruction classes)
    groups = {"fetch": [["FETCH_SIZE"]], "write": [["WRITE_SIZE"]],
              "valu": [["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                        "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F32"],
                       ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]]}
    merged = {}
    for gname, alternatives in groups.items():
        err = None
        for counters in alternatives:
            d = tempfile.mkdtemp(prefix="vcm_pmc_", dir="/tmp")
            cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--child", "--res", str(args.res), "--scene", str(args.scene), "--algo", args.algo,
                   "--steps", str(steps), "--warmup", str(warm)]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                   stderr=subprocess.PIPE, timeout=600)
                files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
                if r.returncode != 0 or not files:
                    err = "rocprofv3 --pmc %s failed (rc %d): %s" % (" ".join(counters), r.returncode,
                                                                     r.stderr.decode("utf-8", "replace").strip()[-160:])
                    continue
                for k, v in _pmc_table(files[0], warm / float(warm + steps)).items():
                    m = merged.setdefault(k, {})
                    for c, x in v.items():
                        m[c if not c.startswith("_") else c + "_" + gname] = x
                err = None
                break
            except Exception as e:
                err = "rocprofv3 --pmc %s: %r" % (counters[0], e)
            finally:
                shutil.rmtree(d, ignore_errors=True)
        if err is not None:
            return None, err
    return merged, ("live: rocprofv3 --pmc child runs of this workload (FETCH_SIZE | WRITE_SIZE | SQ VALU group), iterations %d..%d "
                    "= the timed window" % (warm, warm + steps - 1))


def _launches_per_iteration(counters, v, name):
    """a counter value is the mean per DISPATCH; a kernel launched k times per iteration (the radix sort's passes) counts k times.
    k = its dispatches / the dispatches of a kernel that runs once per iteration (K5, else the rarest kernel of the run)"""
    grp = "_n_fetch" if name == "FETCH_SIZE" else "_n_write" if name == "WRITE_SIZE" else "_n_valu"
    n = v.get(grp)
    ref = counters.get("vcm::k_resolve", {}).get(grp) or min((x[grp] for x in counters.values() if x.get(grp)), default=None)
    return max(1, int(round(n / float(ref)))) if n and ref else 1


def _sum_over(counters, prefixes, name):
    vals = [v[name] * _launches_per_iteration(counters, v, name) for k, v in counters.items()
            if _is_kernel(k, [p.split("<")[0] for p in prefixes]) and name in v]
    return sum(vals) if vals else None


def valu_block(counters, prefixes, kernel_ms):
    """VALU issue roofline of one kernel (group) from the SQ counters of the child run"""
    insts = _sum_over(counters, prefixes, "SQ_INSTS_VALU")
    active = _sum_over(counters, prefixes, "SQ_ACTIVE_INST_VALU")
    thr = _sum_over(counters, prefixes, "SQ_THREAD_CYCLES_VALU")
    if not insts or not active or not thr or kernel_ms <= 0:
        return None
    f64 = (_sum_over(counters, prefixes, "SQ_INSTS_VALU_ADD_F64") or 0) + (_sum_over(counters, prefixes, "SQ_INSTS_VALU_MUL_F64") or 0)
    trans = _sum_over(counters, prefixes, "SQ_INSTS_VALU_TRANS_F32") or 0
    classes = _sum_over(counters, prefixes, "SQ_INSTS_VALU_TRANS_F32") is not None
    plain = max(insts - f64 - trans, 0)
    cycles = plain * VALU_CYCLES["plain"] + f64 * VALU_CYCLES["f64"] + trans * VALU_CYCLES["trans"]
    t_min = cycles / (VALU_SIMDS * VALU_CLK_HZ)
    t_guide = (plain * VALU_CYCLES_GUIDE["plain"] + f64 * VALU_CYCLES_GUIDE["f64"] + trans * VALU_CYCLES_GUIDE["trans"]) / (VALU_SIMDS * VALU_CLK_HZ)
    lane = thr / (active * 64.0)
    frac = t_min / (kernel_ms / 1e3)
    wc, wa = _sum_over(counters, prefixes, "SQ_WAVE_CYCLES"), _sum_over(counters, prefixes, "SQ_WAIT_ANY")
    return {"insts": int(insts), "insts_f64": int(f64), "insts_trans": int(trans), "cycles_per_inst": round(cycles / insts, 3),
            "class_costs": VALU_CYCLES if classes else "classes not collected: every instruction at %.1f cycles" % VALU_CYCLES["plain"],
            "simds": VALU_SIMDS, "clk_GHz": VALU_CLK_HZ / 1e9,
            "issue_ms": round(t_min * 1e3, 3), "lane_util": round(lane, 3), "frac": round(frac, 4),
            "frac_guide": round(t_guide / (kernel_ms / 1e3), 4), "guide_costs": VALU_CYCLES_GUIDE,
            "frac_useful_lanes": round(frac * lane, 4), "wait_share_of_wave_cycles": round(wa / wc, 3) if wc and wa else None}


def algorithmic_bytes(st, n_paths, n_cells):
    """SURVEY.md section 8(d), split per kernel"""
    nlv, A, Cc, K, S = st["lightVertices"], st["mergeAccepted"], st["mergeCandidates"], st["connections"], st["lightSplats"]
    gridv = st["gridVertices"]
    parts = {"light": 68 * nlv + 24 * S,                 # vertex store + splat RMW
             "grid": 20 * gridv + 8 * n_cells,           # position read, index write, histogram + scan
             "camera": 68 * K + 24 * n_paths,            # one LightVertex per connection + framebuffer RMW
             "merge": 52 * A + 16 * (Cc - A)}            # accepted photons 52 B, rejected candidates 16 B
    return sum(parts.values()), parts


def design_bytes(st, n_paths):
    """What THIS design has to move per kernel if every distinct datum crosses HBM once ("byte model
    of the design"): N paths, V light vertices, Q camera vertices, K vertex connections, S light splats, G grid vertices."""
    N, V, K, S, G = n_paths, st["lightVertices"], st["connections"], st["lightSplats"], st["gridVertices"]
    Q = st["mergeQueries"] if st["mergeQueries"] > 0 else st["cameraRays"]
    return {"k_light_trace": 80 * V + 6 * N,                      # 80-byte store records; count, length mask, tape per path
            "k_camera_trace": 92 * Q + 8 * K + 29 * N,            # 80-byte vertex record + DI task + sort key / place; VC tasks; per path 29
            "k_connect_di+vc": 116 * Q + 24 * K + 80 * Q + 80 * V,  # DI: task, record, addend, sort scatter; VC: task, addend, each record / light vertex once
            "k_merge": 100 * Q + 52 * G,                          # sorted index, record, addend per query; the cell-sorted photon set once
            "k_resolve": 44 * N + 32 * Q + 16 * K,                # camOut, mask, fb RMW per pixel; DI + merge addends per vertex; VC addends
            "k_connect_camera": 100 * V + 12 * S,                 # slot, store record, splat; place + histogram per splat
            "grid_build": 16 * G + 52 * G + 8 * N,                # position+slot side array in, 52-byte cell-sorted copy out, cell table
            "splat": 12 * N + 16 * V + 64 * S}


def roofline_block(st, n_local, n_paths):
    """per-kernel times (HIP events / device clock stamps of the timed launches) against the algorithmic bytes"""
    b_iter, b_parts = algorithmic_bytes(st, n_local, n_paths)
    dsg = design_bytes(st, n_local)
    kernels = {"k_light_trace": (st["msLightKernel"], b_parts["light"]),
               "k_camera_trace": (st["msCameraKernel"], 24 * n_local),
               "k_connect_di+vc": (st["msConnectKernels"], b_parts["camera"] - 24 * n_local),
               "k_merge": (st["msMergeKernel"], b_parts["merge"])}
    dom = max(kernels, key=lambda k: kernels[k][0])
    dom_s, dom_b = kernels[dom][0] / 1e3, kernels[dom][1]
    achieved = dom_b / dom_s / 1e9 if dom_s > 0 else 0.0
    iter_s = st["msTotal"] / 1e3
    frac = achieved / HBM_PEAK_GBS
    return dom, {
        "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(frac, 5),
        # SURVEY 8(d) prices every candidate / accepted photon as an HBM read; a cell-sorted, query-sorted merge serves most
        # of them from cache, so the model can exceed the peak: it then bounds nothing -- read frac_traffic and valu instead
        "frac_model_invalid": bool(frac > 1.0),
        "traffic": None,
        "algorithmic_bytes_per_launch": int(dom_b), "design_bytes_per_launch": int(dsg[dom]), "kernel_ms": round(dom_s * 1e3, 3),
        "frac_design": round(dsg[dom] / dom_s / 1e9 / HBM_PEAK_GBS, 5) if dom_s > 0 else 0.0,
        "iteration_algorithmic_bytes": int(b_iter), "iteration_ms": round(iter_s * 1e3, 3),
        "iteration_achieved_GBs": round(b_iter / iter_s / 1e9, 2) if iter_s > 0 else 0.0,
        "iteration_frac": round(b_iter / iter_s / 1e9 / HBM_PEAK_GBS, 5) if iter_s > 0 else 0.0,
        "iteration_design_bytes": int(sum(dsg.values())),
        "per_kernel": {k: {"ms": round(v[0], 3), "algorithmic_bytes": int(v[1]), "design_bytes": int(dsg[k]),
                           "GBs": round(v[1] / (v[0] / 1e3) / 1e9, 2) if v[0] > 0 else 0.0,
                           "frac_design": round(dsg[k] / (v[0] / 1e3) / 1e9 / HBM_PEAK_GBS, 5) if v[0] > 0 else 0.0}
                       for k, v in kernels.items()},
        "query_sort_ms": round(st["msQuerySort"], 3), "grid_build_ms_side_stream": round(st["msGrid"], 3),
        "scope": "rank 0 shard, mean over the timed iterations"}


def add_counters(roof, dom, counters, note, st, n_local):
    """fold the child runs' counters into the roofline block: HBM traffic and the VALU issue roofline, per kernel"""
    dsg = design_bytes(st, n_local)
    groups = dict(KERNEL_KEYS)
    groups.update({"k_resolve": ["vcm::k_resolve"], "k_connect_camera": ["vcm::k_connect_camera"],
                   "grid_build": ["vcm::k_cell_", "vcm::k_radix_", "vcm::k_grid_", "vcm::k_bbox"], "splat": ["vcm::k_splat_"]})
    for name, prefixes in groups.items():
        pk = roof["per_kernel"].setdefault(name, {"design_bytes": int(dsg.get(name, 0))})
        fetch = _sum_over(counters, prefixes, "FETCH_SIZE")
        write = _sum_over(counters, prefixes, "WRITE_SIZE")
        us_alone = _sum_over(counters, prefixes, "_us_valu")
        if "ms" not in pk and us_alone:
            pk["ms_serialised_under_profiler"] = round(us_alone / 1e3, 3)
        ms = pk.get("ms") or pk.get("ms_serialised_under_profiler") or 0.0
        if fetch is not None and write is not None:
            fac, fsrc = fetch_factor(name)
            t = int(fac * 1024 * fetch + 1024 * write)
            pk["traffic"] = t
            pk["traffic_read"] = int(fac * 1024 * fetch)
            pk["traffic_written"] = int(1024 * write)
            pk["fetch_factor"] = fsrc
            if ms > 0:
                pk["frac_traffic"] = round(t / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
            if pk.get("design_bytes"):
                pk["traffic_over_design"] = round(t / float(pk["design_bytes"]), 3)
        v = valu_block(counters, prefixes, ms)
        if v:
            pk["valu"] = v
    d = roof["per_kernel"][dom]
    if "traffic" in d:
        roof["traffic"] = d["traffic"]
        roof["traffic_source"] = note + "; " + d["fetch_factor"]
        roof["achieved_traffic_GBs"] = round(d["traffic"] / (roof["kernel_ms"] / 1e3) / 1e9, 2)
        roof["frac_traffic"] = d.get("frac_traffic")
        roof["traffic_over_algorithmic"] = round(d["traffic"] / max(roof["algorithmic_bytes_per_launch"], 1), 4)
        roof["traffic_over_design"] = d.get("traffic_over_design")
    if "valu" in d:
        roof["valu"] = d["valu"]
        roof["limiter"] = "valu-issue + gather latency" if d["valu"]["frac"] > (roof.get("frac_traffic") or 0) else "hbm"
    tot = sum(pk.get("traffic", 0) for pk in roof["per_kernel"].values())
    if tot:
        roof["iteration_traffic"] = int(tot)
        roof["iteration_traffic_over_design"] = round(tot / float(max(roof["iteration_design_bytes"], 1)), 3)


def finalize_roofline(roof):
    """The line's `frac` is the fraction of the roof that BINDS the dominant kernel, and it is a fraction: <= 1, or null.

    SURVEY 8(d)'s gather model prices every merge candidate as an HBM read; the cell-sorted, query-sorted merge serves most
    of them from cache, so for K4 the model reads more than the 8 TB/s peak and bounds nothing; for K1 / K3 it counts 24
    bytes per pixel and says nothing either.  So, with counters (live child runs, or a recording made on these kernel
    sources): frac = max(measured HBM traffic / time / 8 TB/s, class-weighted VALU issue time / time), `bound` says
    which, achieved / peak / unit belong to it.  Without counters: the model's figure if it is <= 1, else null -- never a
    number above 1.  The model's own figure always stays beside it as frac_algorithmic.  Returns the block with the
    figures a reader needs first."""
    alg = {"frac_algorithmic": roof["frac"], "achieved_algorithmic_GBs": roof["achieved"]}
    ft = roof.get("frac_traffic")
    vf = (roof.get("valu") or {}).get("frac")
    head = {"bound": "hbm", "kernel": roof["kernel"], "achieved": roof["achieved"], "peak": roof["peak"], "unit": roof["unit"],
            "frac": roof["frac"]}
    if ft is not None or vf is not None:
        if vf is not None and vf >= (ft or 0.0):
            v = roof["valu"]
            peak = VALU_SIMDS * VALU_CLK_HZ / v["cycles_per_inst"] / 1e9
            head.update({"bound": "valu", "achieved": round(v["insts"] / (roof["kernel_ms"] / 1e3) / 1e9, 2), "peak": round(peak, 1),
                         "unit": "G wave-instr/s (1024 SIMDs x 2.4 GHz / %.2f cyc; the cost per instruction is MEASURED on this chip, "
                                 "profiles/tools/valu_bench.hip -- at the guide's 2 / 4 / 8 cycles: valu_frac_guide)" % v["cycles_per_inst"], "frac": vf})
        else:
            head.update({"bound": "hbm", "achieved": roof["achieved_traffic_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s (measured HBM traffic)",
                         "frac": ft})
    elif roof["kernel"] in ("k_light_trace", "k_camera_trace"):
        # SURVEY 8(d) gives the trace kernels 68 B per stored vertex / 24 B per pixel: they are bound by the latency of their
        # dependent chain and by VALU issue, and a byte model says nothing about them -- no counters, no fraction
        head["frac_note"] = "no counters: SURVEY 8(d)'s bytes do not bound a trace kernel; see frac_algorithmic"
        head["frac"] = None
    if head["frac"] is not None and not (0.0 <= head["frac"] <= 1.0):
        head["frac_note"] = ("%.3f of the %s roof is not a fraction (%s): reported as null" %
                             (head["frac"], head["bound"], "SURVEY 8(d)'s gather model, no counters" if ft is None and vf is None
                              else "the roof's peak is not one for this kernel"))
        head["frac"] = None
    head["limiter"] = roof.get("limiter")
    if roof.get("valu"):
        head["valu_frac"] = roof["valu"]["frac"]
        head["valu_frac_guide"] = roof["valu"].get("frac_guide")
        head["valu_lane_util"] = roof["valu"]["lane_util"]
    head["traffic"] = roof.get("traffic")
    head["frac_traffic"] = ft
    head.update(alg)
    head["frac_model_invalid"] = roof.get("frac_model_invalid")
    for k, v in roof.items():
        if k not in head and k not in ("frac", "achieved", "peak", "unit", "bound"):
            head[k] = v
    return head


def _clean(o):
    """NaN / inf never reach the line (json.dumps(allow_nan=False) would refuse them): they become null"""
    if isinstance(o, float):
        return o if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _clean(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v) for v in o]
    return o


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def write_detail(full):
    """everything the run measured (per-kernel blocks, counters, config blocks, CPU samples) -> bench_detail.json; the
    stdout line only names the file.  -> path relative to the repository (or absolute, if the tree is read-only)"""
    text = json.dumps(_clean(full), indent=1, allow_nan=False)
    want = os.environ.get("SMALLVCM_AMD_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
    for path in (want, os.path.join(tempfile.gettempdir(), "smallvcm_amd_bench_detail.json")):
        try:
            with open(path, "w") as f:
                f.write(text + "\n")
            out_dir = os.path.join(ROOT, "gpurun_out")   # on a gpurun box this directory travels back
            if os.path.isdir(out_dir) and os.path.dirname(os.path.abspath(path)) != out_dir:
                shutil.copyfile(path, os.path.join(out_dir, "bench_detail.json"))
            return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
        except OSError:
            continue
    return None


ROOF_LINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "frac_algorithmic",
                  "valu_frac", "valu_frac_guide", "valu_lane_util", "kernel_ms", "iteration_frac", "iteration_ms", "traffic_source", "frac_note")


def short_line(full, detail_path):
    """The ONE stdout line: what the driver's contract names, the roofline and CPU baseline in figures, one summary per
    configuration, and where the rest is.  Always below LINE_LIMIT bytes: optional parts go first if it ever is not."""
    roof = full.get("roofline") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype")}
    line["data"] = _short(full.get("data", "synthetic"), 80)
    cfg = full.get("config") or {}
    line["config"] = {"workload": _short(cfg.get("workload", ""), 200), "baseline_config": _short(cfg.get("baseline_config", ""), 48),
                      "paths_per_step": cfg.get("paths_per_step"), "parallelism": _short(cfg.get("parallelism", ""), 200)}
    r = {k: roof.get(k) for k in ROOF_LINE_KEYS if k in roof or k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    if "unit" in r:
        r["unit"] = _short(r["unit"], 60)
    if r.get("traffic_source"):
        r["traffic_source"] = _short(r["traffic_source"], 140)
    if r.get("frac_note"):
        r["frac_note"] = _short(r["frac_note"], 120)
    line["roofline"] = r
    cb = full.get("cpu_baseline")
    if cb:
        c = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "wall_s", "host_cores") if k in cb}
        if cb.get("sample"):
            c["sample"] = _short(cb["sample"], 160)
        if cb.get("host_ram_GB") is not None:
            c["host_ram_GB"] = cb["host_ram_GB"]
        if isinstance(cb.get("all_cores_at_this_res"), dict):
            c["all_cores_at_this_res"] = {k: cb["all_cores_at_this_res"].get(k) for k in ("needed_GB", "fits", "value", "cores", "wall_s") if cb["all_cores_at_this_res"].get(k) is not None}
        if isinstance(cb.get("all_cores"), dict) and "value" in cb["all_cores"]:
            c["all_cores"] = {k: cb["all_cores"].get(k) for k in ("value", "cores", "wall_s", "res")}
        port = cb.get("port") if cb.get("kind") == "reference" else None
        if isinstance(port, dict) and "value" in port:
            c["port"] = {"value": port["value"], "cores": port.get("cores")}
        if "error" in cb:
            c["error"] = _short(cb["error"], 120)
        line["cpu_baseline"] = c
    if full.get("configs"):
        line["configs"] = []
        for c in full["configs"]:
            if "error" in c:
                line["configs"].append({"name": c.get("name"), "error": _short(c["error"], 60)})
                continue
            cr = c.get("roofline") or {}
            line["configs"].append({"name": c["name"], "value": c["value"], "ms_per_step": c["ms_per_step"],
                                    "iteration_frac": cr.get("iteration_frac"), "frac": cr.get("frac"), "bound": cr.get("bound"),
                                    "kernel": cr.get("kernel")})
    if full.get("dropin"):
        line["dropin"] = {k: ({kk: v.get(kk) for kk in ("value", "ratio", "error") if v.get(kk) is not None} if isinstance(v, dict) else v)
                          for k, v in full["dropin"].items() if k.startswith("renderers_") or k == "error"}
    if full.get("host_cross_check"):
        line["host_cross_check"] = {"value": full["host_cross_check"]["value"], "host": "python/ctypes"}
    if full.get("strong_decomposition"):
        sd = full["strong_decomposition"]
        line["strong_decomposition"] = {k: sd.get(k) for k in ("value", "ms_per_step", "scaling", "paths_per_step")}
    if full.get("selftest"):
        line["selftest"] = {k: full["selftest"].get(k) for k in ("ran", "ok", "why", "ranks", "direct_equals_allgather_bitwise", "error") if full["selftest"].get(k) is not None}
    if full.get("hybrid_decomposition"):
        sd = full["hybrid_decomposition"]
        line["hybrid_decomposition"] = {k: sd.get(k) for k in ("value", "ms_per_step", "scaling", "paths_per_step", "shards", "inflight")}
    if full.get("image_mean"):
        line["image_mean"] = full["image_mean"]
    line["detail"] = detail_path
    line = _clean(line)
    for drop in (None, "image_mean", "host_cross_check", "cpu_baseline.sample", "configs", "dropin", "config.parallelism"):
        if drop:
            a, _, b = drop.partition(".")
            if b:
                (line.get(a) or {}).pop(b, None)
            else:
                line.pop(a, None)
        if len(json.dumps(line, allow_nan=False)) < LINE_LIMIT:
            break
    return line


def emit(full):
    """detail file + the one short line on stdout"""
    line = short_line(full, write_detail(full))
    print(json.dumps(line, allow_nan=False), flush=True)


def workload_name(scene, algo, res, replicas, first, last):
    if isinstance(scene, str):
        scene_txt = ("scene file %s (vcm_scene_load), BVH," % scene.split(":", 1)[1] if scene.startswith("file:")
                     else "tests/mesh_scenes.py bumpy_room(grid=%s), BVH," % scene.split(":")[1])
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


def host_memory_GB():
    """(total, available) in GB as THIS PROCESS may use it: /proc/meminfo capped by the container's cgroup limit (a pod on a 3 TB
    host may own a fraction of it -- round 6 lost a box to an all-cores leg that trusted /proc/meminfo alone); (None, None) if unreadable"""
    try:
        kv = {}
        for ln in open("/proc/meminfo"):
            k, _, v = ln.partition(":")
            kv[k.strip()] = float(v.split()[0]) / 1048576.0
        total, avail = kv["MemTotal"], kv.get("MemAvailable", kv["MemFree"])
    except Exception:
        return None, None
    for lim, use in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            t = open(lim).read().strip()
            if t and t != "max":
                cap = float(t) / 2 ** 30
                used = float(open(use).read().strip()) / 2 ** 30
                if cap < total:
                    total, avail = cap, min(avail, max(cap - used, 0.0))
            break
        except Exception:
            continue
    return round(total, 1), round(avail, 1)


REF_GB_PER_RENDERER_2048 = 1.3   # one VertexCM of the reference at 2048^2: mLightVertices 8.9 M x 120 B + the hash grid + the framebuffer


def dropin_leg(res, steps, warmup):
    """The north_star surface -- AbstractRenderer::RunIteration + GetFramebuffer (renderer.hxx:49-55) over the drop-in, the host
    Framebuffer refreshed after EVERY iteration -- timed by dropin/dropin_rate with one renderer and with two on two host
    threads (the unchanged render() creates one per core, smallvcm.cxx:61-72), next to the C-ABI alone in the same process."""
    exe = os.path.join(ROOT, "smallvcm_amd", "dropin", "dropin_rate")
    if not os.path.exists(exe):
        return {"error": "smallvcm_amd/dropin/dropin_rate not built (needs the reference checkout at build time)"}
    out = {}
    for n in (1, 2):
        cmd = [exe, str(res), str(steps), str(warmup)] + (["1", "vcm", str(n)] if n > 1 else [])
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            out["renderers_%d" % n] = {"value": d["dropin_Mpaths_s"], "cabi": d["cabi_Mpaths_s"], "ratio": d["dropin_over_cabi"],
                                       "pixels_that_differ": d.get("pixels_that_differ")}
        except Exception as e:
            out["renderers_%d" % n] = {"error": repr(e)[:200]}
    out["unit"] = "Mpaths/s"
    out["what"] = ("reference's Scene / VertexCM interface over dropin/vertexcm.hxx at %d^2, host Framebuffer refreshed after every "
                   "RunIteration; cabi = vcm_run_iteration alone in the same process" % res)
    return out


def rccl_selftest(n_ranks=2, res=256, iters=3):
    """The FIRST execution of the RCCL path with more than one rank should not be the timed run: wherever two GPUs
    are visible, render scene 1 VCM at 256^2 for three iterations with ONE renderer on `n_ranks` shards over real RCCL -- once with
    ncclAllGather, once with the direct exchange (grouped ncclSend / ncclRecv) -- and compare both frames with one renderer on
    one GPU (equal up to the order of the final framebuffer sum; the two exchanges bit for bit).  A few seconds."""
    import numpy as np
    exe = os.path.join(ROOT, "smallvcm_amd", "host", "vcm_render")
    if not os.path.exists(exe):
        return {"ran": False, "why": "smallvcm_amd/host/vcm_render not built"}
    sys.path.insert(0, ROOT)
    from smallvcm_amd.renderer import load_library
    visible = load_library(require_gpu=False).vcm_device_count()
    if visible < n_ranks:
        return {"ran": False, "why": "%d GPU(s) visible, %d needed" % (visible, n_ranks), "gpus_visible": visible}

    def pfm(path):
        with open(path, "rb") as f:
            assert f.readline().strip() == b"PF"
            w, h = (int(x) for x in f.readline().split())
            f.readline()
            return np.frombuffer(f.read(w * h * 12), dtype="<f4").copy()

    tmp = tempfile.mkdtemp(prefix="vcm_selftest_")
    frames, info = {}, {"ran": True, "ranks": n_ranks, "res": res, "iterations": iters, "gpus_visible": visible}
    try:
        runs = (("one_gpu", ["--renderers", "1"], {}),
                ("allgather", ["--gpus", str(n_ranks), "--shards", str(n_ranks), "--inflight", "1"], {"SMALLVCM_AMD_FARM_EXCHANGE": "allgather"}),
                ("direct", ["--gpus", str(n_ranks), "--shards", str(n_ranks), "--inflight", "1"], {"SMALLVCM_AMD_FARM_EXCHANGE": "direct"}))
        for name, extra, env in runs:
            out = os.path.join(tmp, name + ".pfm")
            e = dict(os.environ, **env)
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):   # a plain one-process run, whatever launched us
                e.pop(k, None)
            r = subprocess.run([exe, "-s", "1", "-a", "vcm", "-i", str(iters), "--res", str(res), str(res), "-o", out, "--json"] + extra,
                               capture_output=True, text=True, timeout=300, env=e)
            if r.returncode != 0:
                info["ok"] = False
                info["error"] = "%s: rc %d %s" % (name, r.returncode, (r.stderr or r.stdout)[-300:])
                return info
            frames[name] = pfm(out)
        a, g, d = frames["one_gpu"], frames["allgather"], frames["direct"]
        info["allgather_vs_one_gpu_max_rel"] = float(np.max(np.abs(g - a) / np.maximum(np.abs(a), 1e-6)))
        info["direct_equals_allgather_bitwise"] = bool(np.array_equal(g.view(np.uint32), d.view(np.uint32)))
        info["ok"] = bool(np.allclose(g, a, rtol=3e-6, atol=2e-7) and info["direct_equals_allgather_bitwise"] and a.max() > 0)
    except Exception as ex:
        info["ok"] = False
        info["error"] = repr(ex)[:300]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return info


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


# ---------------------------------------------------------------------------------------------------------------------
# N > 1: the C++ host (smallvcm_amd/host/vcm_farm.cpp).  One line, default and strong decomposition.
def multi_gpu(args):
    from smallvcm_amd._abi import ALGO_BY_NAME
    from smallvcm_amd.renderer import cornell_scene, load_library
    from smallvcm_amd import farm as F
    N = args.gpus
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    multi_process = env_world > 1
    if multi_process and env_world != N:
        raise SystemExit("bench.py --gpus %d launched with WORLD_SIZE %d" % (N, env_world))
    L = load_library(require_gpu=False)
    visible = L.vcm_device_count()
    if visible <= 0:
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    threads_standin = args.collectives == "threads"
    if multi_process and threads_standin:
        raise SystemExit("--collectives threads is the one-process stand-in; it cannot be combined with torch.distributed.run")
    dist = None
    if multi_process:   # plumbing only: ships the RCCL ids (gloo, CPU); the farm owns every GPU-side collective
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=env_world)
        devices, first = [local_rank], rank
    else:
        devices = [(k % visible) if threads_standin else k for k in range(N)]
        first = 0
        if not threads_standin and N > visible:
            raise SystemExit("bench.py --gpus %d: only %d device(s) visible (use --collectives threads to share one)" % (N, visible))
    sc = cornell_scene(args.scene, args.res, args.res)
    algo = ALGO_BY_NAME[args.algo]
    n_paths = args.res * args.res

    def run(shards, inflight):
        groups = N // shards
        R = groups * inflight
        ids = None
        if multi_process:
            import torch
            nb = (1 + groups) * F.load_farm_library().vcm_farm_unique_id_bytes()
            t = torch.zeros(nb, dtype=torch.uint8)
            if rank == 0:
                t = torch.tensor(list(F.unique_ids(1 + groups)), dtype=torch.uint8)
            dist.broadcast(t, src=0)
            ids = bytes(t.tolist())
        r = F.farm_render(sc, algo, iterations=args.steps * R, ranks=N, shards=shards, inflight=inflight, devices=devices,
                          first_rank=first, warmup=args.warmup, same_window=True, collectives=args.collectives, ids=ids)
        r["R"], r["shards"], r["inflight"] = R, shards, inflight
        r["value"] = 2.0 * n_paths * args.steps * R / r["wall_s"] / 1e6
        return r

    # Default = north_star's decomposition: ONE renderer, its light paths and pixels sharded over all N GPUs, the light
    # vertices all-gathered every iteration, one framebuffer reduce at read-out ("strong": the work of a step is fixed).
    # The replica hybrid (pairs of GPUs x 2 renderers in flight: mostly the reference's own iteration parallelism,
    # smallvcm.cxx:61-72, "weak") is timed beside it and reported as hybrid_decomposition.
    shards = args.shards if args.shards > 0 else N
    inflight = args.inflight if args.inflight > 0 else 1
    # two real RCCL ranks against one GPU, both exchanges, before anything is timed (a few seconds; rank 0's subprocesses)
    selftest = rccl_selftest() if (rank == 0 and args.collectives == "rccl") else None
    main_run = run(shards, inflight)
    strong = None
    hybrid = run(2, 2) if (args.shards <= 0 and args.inflight <= 0 and N % 2 == 0 and not args.no_hybrid) else None
    if rank == 0:
        st = main_run["stats"]
        n_local = n_paths // shards
        _, roof = roofline_block(st, n_local, n_paths)
        R = main_run["R"]
        host = ("C++ (smallvcm_amd/host/vcm_farm.cpp via include/smallvcm_amd_farm.h): one host thread per GPU; %s"
                % ("%d processes (torch.distributed.run), ncclCommInitRank from ids shipped over gloo" % N if multi_process
                   else "one process, %d rank threads" % N))
        out = {
            "metric": "Mpaths/sec (light+camera), %s scene %d at %d^2" % (args.algo.upper(), args.scene, args.res),
            "value": round(main_run["value"], 3), "unit": "Mpaths/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(main_run["wall_s"] / args.steps * 1e3, 3), "higher_is_better": True,
            # per-GPU work fixed as N grows: every step renders R = N / shards * inflight iterations, one per GPU
            "scaling": "strong" if R == 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (reference's built-in Cornell box scene %d)" % args.scene,
            "config": {"workload": workload_name(args.scene, args.algo, args.res, R, args.warmup, args.warmup + args.steps - 1)
                                   + " by every renderer",
                       "baseline_config": "C4" if (args.scene, args.algo, args.res) == (1, "vcm", 2048) else "other",
                       "paths_per_step": 2 * n_paths * R,
                       "parallelism": "%d renderer(s) (iteration-parallel, smallvcm.cxx:61-108), each on %d path-index shard(s) "
                                      "(RCCL all-gather of the light vertices every iteration), %d renderer(s) in flight per GPU "
                                      "group, framebuffer all-reduce at read-out" % (R, shards, inflight),
                       "host": host, "collectives": "RCCL" if not threads_standin else "in-process stand-in (one GPU shared by the ranks)",
                       "rccl_ranks": main_run["rccl_ranks"]},
            "rank_iteration_ms": [round(x, 3) for x in main_run["rank_iteration_ms"]],
            "roofline": roof,
            "counters": {k: int(st[k]) for k in ("lightVertices", "gridVertices", "mergeQueries", "mergeCandidates",
                                                 "mergeAccepted", "connections", "lightSplats", "lightRays", "cameraRays", "shadowRays")},
            "image_mean": [round(float(x), 5) for x in main_run["image"].mean(axis=(0, 1))],
        }
        roof["scope"] = "world rank 0 (1 of %d shards of renderer 0), mean over its timed iterations" % shards
        out["roofline"] = finalize_roofline(roof)
        if selftest is not None:
            out["selftest"] = selftest
        if hybrid is not None:
            out["hybrid_decomposition"] = {
                "value": round(hybrid["value"], 3), "unit": "Mpaths/s", "scaling": "weak", "shards": 2, "inflight": 2,
                "ms_per_step": round(hybrid["wall_s"] / args.steps * 1e3, 3), "paths_per_step": 2 * n_paths * hybrid["R"],
                "parallelism": "%d renderers (iteration-parallel, smallvcm.cxx:61-108), each on 2 path-index shards, 2 in flight per "
                               "GPU pair: every step renders %d iterations" % (hybrid["R"], hybrid["R"]),
                "rccl_ranks": hybrid["rccl_ranks"], "rank_iteration_ms": [round(x, 3) for x in hybrid["rank_iteration_ms"]]}
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", type=int, default=2048)
    ap.add_argument("--scene", type=int, default=1)
    ap.add_argument("--algo", default="vcm")
    ap.add_argument("--scene-file", default="", help="a .vcmscene / .obj scene file (relative to the repository) instead of --scene")
    ap.add_argument("--shards", type=int, default=0,
                    help="GPUs that share one iteration (default: all of them = north_star's decomposition)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="renderers taking turns on each group of --shards GPUs (default 1)")
    ap.add_argument("--no-hybrid", action="store_true", help="N > 1: skip the second timing (pairs of GPUs x 2 renderers in flight)")
    ap.add_argument("--collectives", default="rccl", choices=["rccl", "threads"],
                    help="threads: in-process stand-in for RCCL so that --gpus N ranks can share one GPU (tests)")
    ap.add_argument("--cpu-baseline", default="reference", choices=["reference", "port", "none"],
                    help="reference: the unmodified reference in this run (minutes of host CPU at 2048^2) + the port; "
                         "port: only the oracle restatement (seconds); none")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="same as --cpu-baseline none")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the reference leg (default min(32, cores): "
                                                               "it saturates the host's memory system there)")
    ap.add_argument("--cpu-all-cores-2048", action="store_true",
                    help="also time the reference on EVERY host thread at the headline resolution (one 2048^2 iteration per thread: "
                         "~1.3 GB each and about ten minutes; refused when the host's memory does not hold it)")
    ap.add_argument("--selftest", action="store_true",
                    help="only run the multi-rank RCCL self-check (2 real ranks, 256^2, 3 iterations, both exchanges against one GPU) and print its result")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in leg (dropin/dropin_rate: the reference's renderer interface)")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE's other single-GPU configs")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 child runs (HBM traffic, VALU counters)")
    ap.add_argument("--no-cross-check", action="store_true", help="skip the second timing of the headline through the Python host")
    ap.add_argument("--record-counters", default="", metavar="TAG",
                    help="keep the live counter sets as profiles/TAG_counters_<config>.json (the fallback of runs without a profiler)")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)   # inner run under rocprofv3: GPU loop only
    args = ap.parse_args()
    if args.no_cpu_baseline:
        args.cpu_baseline = "none"
    if args.scene_file:   # a scene the reference cannot load: no CPU leg, no per-kernel child runs by scene id
        args.scene = "file:" + args.scene_file
        args.cpu_baseline, args.no_traffic = "none", True
    if args.selftest:
        print(json.dumps({"selftest": rccl_selftest(max(2, min(args.gpus, 8)) if args.gpus > 1 else 2)}, allow_nan=False), flush=True)
        return
    if args.gpus > 1:
        return multi_gpu(args)

    from smallvcm_amd._abi import ALGO_BY_NAME
    from smallvcm_amd.renderer import HipBackend, RenderFarm, cornell_scene, load_library
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if args.child and not isinstance(args.scene, str):
        # inner run under rocprofv3: the C++ farm's loop over the C-ABI alone -- no torch import (1-2 s per counter pass)
        from smallvcm_amd import farm as F
        F.farm_render(cornell_scene(args.scene, args.res, args.res), ALGO_BY_NAME[args.algo], iterations=args.steps * max(args.inflight, 1),
                      ranks=1, shards=1, inflight=max(args.inflight, 1), devices=[local_rank], warmup=args.warmup, same_window=True,
                      collectives="threads")
        return

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)

    def sync():
        torch.cuda.synchronize()

    def make_farm(scene, algo_name, res, shards, inflight):
        if isinstance(scene, str) and scene.startswith("file:"):   # a scene file: version-2 description, BVH
            from smallvcm_amd.scene_file import load_scene
            sc = load_scene(os.path.join(ROOT, scene.split(":", 1)[1]), res, res)
        elif isinstance(scene, str):   # "mesh:<grid>": the same room built procedurally
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from mesh_scenes import bumpy_room
            sc = bumpy_room(grid=int(scene.split(":")[1]), resx=res, resy=res)
        else:
            sc = cornell_scene(scene, res, res)
        algo = ALGO_BY_NAME[algo_name]
        farm = RenderFarm(lambda seed, s, S: HipBackend(sc, algo, 0.003, 0.75, seed, device=local_rank, rank=s, world=S),
                          1234, 0, 1, shards=shards, dist=None, inflight=inflight)
        farm.set_path_lengths(0, 10)
        for rr in farm.renderers:   # setup, not a step: device buffers exist before the first (possibly timed) iteration
            rr.backend.reserve(10)
        return farm

    res, n_paths = args.res, args.res * args.res
    from smallvcm_amd import farm as F

    def farm_run(scene_id, algo_name, r, nfl, steps, warmup):
        """one GPU, the C++ host (vcm_farm.cpp, the same loop `--gpus N` times): `nfl` renderers (seeds 1234 ..) taking
        turns on the GPU, every one the iteration window warmup .. warmup + steps - 1.  -> (seconds, mean statistics of
        renderer 0's timed iterations, image, per-rank device ms of an iteration)"""
        rr = F.farm_render(cornell_scene(scene_id, r, r), ALGO_BY_NAME[algo_name], iterations=steps * nfl, ranks=1, shards=1,
                           inflight=nfl, devices=[local_rank], warmup=warmup, same_window=True, collectives="threads")
        return rr["wall_s"], rr["stats"], rr["image"], rr["rank_iteration_ms"]

    builtin = not isinstance(args.scene, str)
    cross = None
    if builtin:
        nfl0 = args.inflight if args.inflight > 0 else 1
        elapsed, st, fb, _ = farm_run(args.scene, args.algo, res, nfl0, args.steps, args.warmup)
        if args.child:
            return
        replicas, inflight_used, n_local = nfl0, nfl0, n_paths
        host = ("C++ (smallvcm_amd/host/vcm_farm.cpp via include/smallvcm_amd_farm.h): one rank thread, the loop `--gpus N` runs; "
                "no collective exists at one rank")
        if not args.no_cross_check:   # the Python / ctypes loop over the same C-ABI, as a cross-check of the host
            pf = make_farm(args.scene, args.algo, res, 1, args.inflight if args.inflight > 0 else None)
            e2, _ = timed_run(pf, args.steps, args.warmup, sync)
            cross = {"host": "Python (smallvcm_amd.renderer.RenderFarm over ctypes, torch.cuda.synchronize around the timed region)",
                     "value": round(2.0 * n_paths * args.steps * pf.replicas / e2 / 1e6, 3), "unit": "Mpaths/s",
                     "ms_per_step": round(e2 / args.steps * 1e3, 3)}
            pf.close()
    else:   # a scene file: the farm takes the built-in scenes only
        farm = make_farm(args.scene, args.algo, res, 1, args.inflight if args.inflight > 0 else None)
        replicas = farm.replicas
        elapsed, st = timed_run(farm, args.steps, args.warmup, sync)
        if args.child:
            farm.close()
            return
        fb = farm.framebuffer()
        n_local = farm.backend.count
        inflight_used = farm.inflight
        farm.close()
        host = "Python (smallvcm_amd.renderer.RenderFarm over ctypes): scene files are not wired into the C++ farm"

    value = 2.0 * n_paths * args.steps * replicas / elapsed / 1e6
    dom, roof = roofline_block(st, n_local, n_paths)
    headline = (args.scene, args.algo, res) == (1, "vcm", 2048)
    out = {
        "metric": "Mpaths/sec (light+camera), %s scene %s at %d^2" % (args.algo.upper(), args.scene, res),
        "value": round(value, 3), "unit": "Mpaths/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        # the N = 1 point of the curve --gpus N continues: ONE renderer, a step = one iteration of the whole frame whatever N
        # (the N > 1 default shards that iteration: "strong"); several renderers in flight (--inflight) are replicas: "weak"
        "scaling": "strong" if replicas == 1 else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic (reference's built-in Cornell box scene %s)" % args.scene,
        "config": {"workload": workload_name(args.scene, args.algo, res, replicas, args.warmup, args.warmup + args.steps - 1)
                               + (" by every renderer" if replicas > 1 else ""),
                   "baseline_config": "C4 at 1 GPU (BASELINE.json metric)" if headline else "other",
                   "paths_per_step": 2 * n_paths * replicas,
                   "parallelism": "%d renderer(s) (iteration-parallel, smallvcm.cxx:61-108) on one GPU, %d in flight"
                                  % (replicas, inflight_used),
                   "host": host,
                   "merge_kernel": os.environ.get("SMALLVCM_AMD_MERGE", "pairs")},
        "roofline": roof,
        "counters": {k: int(st[k]) for k in ("lightVertices", "gridVertices", "mergeQueries", "mergeCandidates",
                                             "mergeAccepted", "connections", "lightSplats", "lightRays", "cameraRays",
                                             "shadowRays")},
        "image_mean": [round(float(x), 5) for x in fb.mean(axis=(0, 1))],
    }
    if cross is not None:
        out["host_cross_check"] = cross
    def counters_for(cfg_name, scene, algo_name, r, workload):
        """live child runs under rocprofv3; without a profiler the recording made on these kernel sources, if there is one"""
        if args.no_traffic:
            return None, "disabled (--no-traffic)"
        if isinstance(scene, str):
            return None, "scene files have no child runs"
        ns = argparse.Namespace(res=r, scene=scene, algo=algo_name, steps=args.steps, warmup=args.warmup)
        counters, src = live_counters(ns)
        if counters is not None:
            if args.record_counters:
                try:
                    save_counters(args.record_counters, cfg_name, workload, counters, src)
                except OSError as e:
                    src += "; not recorded: %r" % (e,)
            return counters, src
        rec, path = recorded_counters(cfg_name)
        if rec is not None:
            return rec, "recorded: %s (same kernel sources, hash %s); live measurement unavailable: %s" % (path, kernel_source_hash(), src)
        return None, src

    cfg_name = {(1, "vcm", 2048): "C4", (1, "vcm", 512): "C1", (3, "vcm", 1024): "C2", (1, "bpm", 2048): "C3"}.get(
        (args.scene, args.algo, res), "other") if replicas == 1 else "other"
    counters, src = counters_for(cfg_name, args.scene, args.algo, res, out["config"]["workload"])
    if counters is not None:
        add_counters(roof, dom, counters, src, st, n_local)
    else:
        rec, path = recorded_traffic(dom) if headline else (None, None)
        if rec is not None:
            roof["traffic"] = rec
            roof["traffic_source"] = "recorded: %s (same kernel sources, hash %s); live measurement unavailable: %s" % (
                path, kernel_source_hash(), src)
            roof["achieved_traffic_GBs"] = round(rec / (roof["kernel_ms"] / 1e3) / 1e9, 2)
            roof["frac_traffic"] = round(rec / (roof["kernel_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
            roof["traffic_over_algorithmic"] = round(rec / max(roof["algorithmic_bytes_per_launch"], 1), 4)
        else:
            roof["traffic_source"] = "null: %s; no recording in profiles/ for kernel sources %s" % (src, kernel_source_hash())
    out["roofline"] = finalize_roofline(roof)
    if headline and not args.no_configs:
        cfgs = []
        for name, scene, algo_name, r, nfl in OTHER_CONFIGS:
            try:
                if isinstance(scene, str):   # a scene file: Python host (the farm takes the built-in scenes)
                    f2 = make_farm(scene, algo_name, r, 1, nfl)
                    e2, s2 = timed_run(f2, args.steps, args.warmup, sync)
                    nl = f2.backend.count
                    f2.close()
                else:
                    e2, s2, _, _ = farm_run(scene, algo_name, r, nfl, args.steps, args.warmup)
                    nl = r * r
                dom2, roof2 = roofline_block(s2, nl, r * r)
                if nfl == 1 and name in ("C1", "C2", "C3"):   # BASELINE configs: their own counter passes
                    c2, src2 = counters_for(name, scene, algo_name, r, workload_name(scene, algo_name, r, nfl, args.warmup, args.warmup + args.steps - 1))
                    if c2 is not None:
                        add_counters(roof2, dom2, c2, src2, s2, nl)
                    else:
                        roof2["traffic_source"] = "null: %s" % src2
                if nfl > 1:
                    roof2["scope"] = "first of the %d renderers; its kernels share the GPU with the others', so per-kernel " \
                                     "times are longer than alone" % nfl
                cfgs.append({"name": name, "workload": workload_name(scene, algo_name, r, nfl, args.warmup, args.warmup + args.steps - 1)
                                                 + (" by every renderer" if nfl > 1 else ""),
                             "host": "Python" if isinstance(scene, str) else "C++ farm",
                             "renderers_in_flight": nfl,
                             "value": round(2.0 * r * r * args.steps * nfl / e2 / 1e6, 3), "unit": "Mpaths/s",
                             "ms_per_step": round(e2 / args.steps * 1e3, 3), "steps": args.steps, "warmup": args.warmup,
                             "paths_per_step": 2 * r * r * nfl,
                             "roofline": finalize_roofline(roof2),
                             "counters": {k: int(s2[k]) for k in ("lightVertices", "mergeQueries", "mergeCandidates",
                                                                  "mergeAccepted", "connections", "lightSplats")}})
            except Exception as e:
                cfgs.append({"name": name, "error": repr(e)})
        out["configs"] = cfgs
    if headline and not args.no_dropin and not args.child:
        try:
            out["dropin"] = dropin_leg(res, args.steps, args.warmup)
        except Exception as e:
            out["dropin"] = {"error": repr(e)}
    if args.cpu_baseline != "none":
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
                ram_total, ram_avail = host_memory_GB()
                need = round(REF_GB_PER_RENDERER_2048 * cores * (res / 2048.0) ** 2, 1)
                base["host_ram_GB"] = ram_total
                base["all_cores_at_this_res"] = {"needed_GB": need, "present_GB": ram_total, "available_GB": ram_avail,
                                                 "fits": bool(ram_avail is not None and need < 0.5 * ram_avail)}   # (half: the reference's allocations peak above their steady state)
                if threads < cores and args.cpu_all_cores_2048:
                    if base["all_cores_at_this_res"]["fits"]:
                        try:
                            full = cpu_reference(args.scene, args.algo, res, args.warmup, args.steps, cores)
                            base["all_cores_at_this_res"].update({"value": full["value"], "cores": cores, "wall_s": full["wall_s"], "sample": full["sample"]})
                        except Exception as e:
                            base["all_cores_at_this_res"]["error"] = repr(e)
                    else:
                        base["all_cores_at_this_res"]["refused"] = "%s GB needed, %s GB available" % (need, ram_avail)
                # BASELINE.md section 2 asks for all host cores.  At 2048^2 that is one 80-second iteration (1.2 GB of
                # light vertices + grid) per core: recorded once (profiles/archive/r01_cpu_reference_timing*.json: 1.89 Mpaths/s on
                # 32 threads, 1.79 on 128 -- the host's memory system saturates), not repeated in every run.  What IS
                # measured here on every core is the resolution the reference's own CLI renders (config.hxx:237).
                if threads < cores and headline:
                    try:
                        allc = cpu_reference(args.scene, args.algo, 512, args.warmup, args.steps, cores)
                        base["all_cores"] = {"value": allc["value"], "unit": "Mpaths/s", "cores": cores, "wall_s": allc["wall_s"],
                                             "res": 512, "sample": allc["sample"],
                                             "recorded_2048": "profiles/archive/r01_cpu_reference_timing.json, r01_cpu_reference_timing_128.json: "
                                                              "1.89 Mpaths/s on 32 threads, 1.79 on 128 threads at 2048^2 on this host type"}
                    except Exception as e:
                        base["all_cores"] = {"error": repr(e)}
            except Exception as e:
                port["reference_error"] = repr(e)
        out["cpu_baseline"] = base if base is not None else port
    emit(out)


if __name__ == "__main__":
    main()
