"""bench.py's bookkeeping (no GPU): the kernel names rocprofv3 prints are matched by prefix (the ray-casting kernels
are templates over the kind of scene), and a recorded traffic figure is only used for the kernel sources it was
collected on."""
import json
import os

import bench


def test_kernel_names_are_matched_by_prefix():
    names = {"k_light_trace": "vcm::k_light_trace<1, vcm::SceneQuads>", "k_camera_trace": "vcm::k_camera_trace<1, vcm::SceneBvh>",
             "k_connect_di+vc": "vcm::k_connect_vc<vcm::SceneList>", "k_merge": "vcm::k_merge_walk"}
    for key, name in names.items():
        assert bench._is_kernel(name, bench.KERNEL_KEYS[key]), (key, name)
    assert not bench._is_kernel("vcm::k_light_trace<0, vcm::SceneQuads>", bench.KERNEL_KEYS["k_light_trace"])   # strict mode
    assert not bench._is_kernel("vcm::k_resolve", sum(bench.KERNEL_KEYS.values(), []))


def test_recorded_traffic_needs_the_same_kernel_sources(tmp_path, monkeypatch):
    root = tmp_path
    (root / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(root))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "aaaa")
    rec = {"kernel_src_sha16": "bbbb", "kernels": {"vcm::k_merge_walk": {"fetch_bytes_x2": 10, "write_bytes": 1}}}
    (root / "profiles" / "r99z_traffic.json").write_text(json.dumps(rec))
    assert bench.recorded_traffic("k_merge") == (None, None)          # other sources: not quoted
    rec["kernel_src_sha16"] = "aaaa"
    (root / "profiles" / "r99z_traffic.json").write_text(json.dumps(rec))
    got, path = bench.recorded_traffic("k_merge")
    assert got == 11 and path.endswith(os.path.join("profiles", "r99z_traffic.json"))


def test_workload_names():
    assert "scene 1 -a vcm 2048x2048" in bench.workload_name(1, "vcm", 2048, 1, 2, 9)
    assert "bumpy_room(grid=72)" in bench.workload_name("mesh:72", "vcm", 1024, 1, 2, 9)
    assert "scene file tests/scenes/bumpy_room.vcmscene" in bench.workload_name("file:tests/scenes/bumpy_room.vcmscene", "vcm", 1024, 1, 2, 9)
    assert os.path.exists(os.path.join(bench.ROOT, [c[1] for c in bench.OTHER_CONFIGS if c[0] == "M1"][0].split(":", 1)[1]))


def _stats(**kw):
    st = {k: 0 for k in ("lightVertices", "gridVertices", "mergeQueries", "mergeCandidates", "mergeAccepted", "connections",
                         "lightSplats", "lightRays", "cameraRays", "shadowRays")}
    st.update({k: 0.0 for k in ("msLight", "msGrid", "msCamera", "msTotal", "msLightKernel", "msCameraKernel", "msMergeKernel",
                                "msQuerySort", "msConnectKernels", "radius")})
    st.update(kw)
    return st


def test_roofline_flags_a_model_that_exceeds_the_peak():
    """SURVEY 8(d) prices every merge candidate as an HBM read; when that exceeds 8 TB/s the line says so"""
    st = _stats(lightVertices=9_000_000, gridVertices=9_000_000, mergeQueries=10_000_000, mergeCandidates=1_250_000_000,
                mergeAccepted=210_000_000, connections=19_000_000, lightSplats=7_000_000, msLightKernel=1.0, msCameraKernel=2.0,
                msConnectKernels=1.8, msMergeKernel=3.0, msTotal=9.8)
    dom, roof = bench.roofline_block(st, 2048 * 2048, 2048 * 2048)
    assert dom == "k_merge" and roof["frac"] > 1.0 and roof["frac_model_invalid"] is True
    assert 0 < roof["frac_design"] < 1.0 and roof["design_bytes_per_launch"] == 100 * 10_000_000 + 52 * 9_000_000
    # the camera kernel's byte model is what the design writes (records, task queues), not 24 bytes per pixel
    assert roof["per_kernel"]["k_camera_trace"]["design_bytes"] > 1_000_000_000
    st["msMergeKernel"] = 30.0
    assert bench.roofline_block(st, 2048 * 2048, 2048 * 2048)[1]["frac_model_invalid"] is False


def test_valu_roofline_and_traffic_from_profiler_rows(tmp_path):
    """the --pmc child runs' CSV rows -> per-kernel counters -> VALU issue roofline and calibrated traffic"""
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"]
    d = 0
    for it in range(8):   # 2 warm-up + 6 timed iterations
        for name, insts in (("void vcm::k_merge_walk(vcm::DScene const*)", 1.5e9), ("void vcm::k_camera_trace<1, vcm::SceneQuads>(int)", 0.8e9)):
            for c, v in (("SQ_INSTS_VALU", insts), ("SQ_ACTIVE_INST_VALU", insts), ("SQ_THREAD_CYCLES_VALU", insts * 64 * 0.5)):
                rows.append("%d,\"%s\",%s,%f,%d,%d" % (d, name, c, v if it >= 2 else 0, 1000 * d, 1000 * d + 3000000))
            d += 1
    f = tmp_path / "x_counter_collection.csv"
    f.write_text("\n".join(rows))
    t = bench._pmc_table(str(f), 2 / 8.0)
    assert t["vcm::k_merge_walk"]["SQ_INSTS_VALU"] == 1.5e9 and t["vcm::k_camera_trace"]["_n"] == 8
    v = bench.valu_block(t, bench.KERNEL_KEYS["k_merge"], 3.0)
    # no instruction classes in these rows: every instruction at the measured cost of plain fp32 (2.7 cycles, r02h)
    assert v["lane_util"] == 0.5 and abs(v["frac"] - (1.5e9 * 2.7 / (1024 * 2.4e9)) / 3e-3) < 1e-3
    assert v["cycles_per_inst"] == 2.7 and isinstance(v["class_costs"], str)
    assert abs(v["frac_useful_lanes"] - v["frac"] * 0.5) < 1e-3
    assert bench.valu_block(t, ["vcm::k_resolve"], 1.0) is None
    fac, src = bench.fetch_factor("k_merge")
    assert fac > 0 and "runs" in src


def test_the_lines_frac_is_the_binding_figure():
    """when SURVEY's gather model exceeds the peak, `frac` is max(measured HBM fraction, VALU issue fraction) <= 1, the
    model's figure stays as frac_algorithmic, and the figures a reader needs come first"""
    st = _stats(lightVertices=9_000_000, gridVertices=9_000_000, mergeQueries=10_000_000, mergeCandidates=1_250_000_000,
                mergeAccepted=210_000_000, connections=19_000_000, lightSplats=7_000_000, msLightKernel=1.0, msCameraKernel=2.0,
                msConnectKernels=1.8, msMergeKernel=3.0, msTotal=9.8)
    dom, roof = bench.roofline_block(st, 2048 * 2048, 2048 * 2048)
    plain = bench.finalize_roofline(dict(roof))
    # no counters: a model figure above the peak is not a fraction -- the line says null and keeps the model's figure aside
    assert plain["bound"] == "hbm" and plain["frac"] is None and plain["frac_algorithmic"] > 1 and "frac_note" in plain
    counters = {"vcm::k_merge_walk": {"FETCH_SIZE": 2.5e6, "WRITE_SIZE": 1.0e5, "SQ_INSTS_VALU": 1.3e9, "SQ_ACTIVE_INST_VALU": 4.0e9,
                                      "SQ_THREAD_CYCLES_VALU": 4.0e9 * 64 * 0.7, "SQ_WAVE_CYCLES": 1e10, "SQ_WAIT_ANY": 5e9, "_us_valu": 3000.0}}
    bench.add_counters(roof, dom, counters, "test", st, 2048 * 2048)
    fin = bench.finalize_roofline(roof)
    assert list(fin)[:6] == ["bound", "kernel", "achieved", "peak", "unit", "frac"]
    assert fin["bound"] in ("valu", "hbm") and fin["frac"] == max(fin["valu_frac"], fin["frac_traffic"]) and 0 < fin["frac"] <= 1.0
    assert abs(fin["achieved"] / fin["peak"] - fin["frac"]) < 2e-3
    assert fin["frac_algorithmic"] > 1 and fin["frac_model_invalid"] is True and 0 < fin["frac_traffic"] <= fin["frac"]
    assert abs(fin["valu_lane_util"] - 0.7) < 1e-3 and fin["traffic"] == int(2 * 1024 * 2.5e6 + 1024 * 1.0e5)


# ---- the stdout line (round 4's verdict: a 20 KB line the driver could not parse) ------------------------------------
def _fixture():
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bench_fixture.json")))


def _full_from_fixture(with_counters=True):
    """what bench.py's main() assembles, from the recorded statistics and counters of a real run"""
    fx = _fixture()
    c4 = fx["C4"]
    n = c4["res"] ** 2
    dom, roof = bench.roofline_block(c4["stats"], n, n)
    if with_counters:
        bench.add_counters(roof, dom, c4["counters"], "live: rocprofv3 --pmc child runs of this workload", c4["stats"], n)
    full = {"metric": "Mpaths/sec (light+camera), VCM scene 1 at 2048^2", "value": round(2.0 * n * c4["steps"] / c4["elapsed_s"] / 1e6, 3),
            "unit": "Mpaths/s", "n_gpus": 1, "steps": c4["steps"], "warmup": c4["warmup"],
            "ms_per_step": round(c4["elapsed_s"] / c4["steps"] * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (reference's built-in Cornell box scene 1)", "config": fx["config"],
            "roofline": bench.finalize_roofline(roof), "counters": {k: v for k, v in c4["stats"].items() if not k.startswith("ms")},
            "image_mean": fx["image_mean"], "host_cross_check": fx["host_cross_check"], "cpu_baseline": fx["cpu_baseline"], "configs": []}
    for name, c in fx["configs"].items():
        nn = c["res"] ** 2
        _, r2 = bench.roofline_block(c["stats"], nn, nn)
        full["configs"].append({"name": name, "value": c["value"], "ms_per_step": c["ms_per_step"], "roofline": bench.finalize_roofline(r2)})
    full["configs"].append({"name": "broken", "error": "RuntimeError('x' * 500)" + "x" * 500})
    return full


def test_the_stdout_line_is_short_parseable_and_its_frac_is_a_fraction(tmp_path, monkeypatch):
    monkeypatch.setenv("SMALLVCM_AMD_BENCH_DETAIL", str(tmp_path / "bench_detail.json"))
    full = _full_from_fixture()
    path = bench.write_detail(full)
    line = bench.short_line(full, path)
    text = json.dumps(line, allow_nan=False)
    assert len(text) < bench.LINE_LIMIT and "\n" not in text
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "configs", "detail"):
        assert k in back, k
    assert back["config"]["workload"].startswith("scene 1 -a vcm 2048x2048") and back["config"]["paths_per_step"] == 2 * 2048 * 2048
    r = back["roofline"]
    assert r["kernel"] == "k_merge" and 0 < r["frac"] <= 1 and r["bound"] in ("hbm", "valu") and r["traffic"] > 0
    assert abs(r["frac"] - max(r["frac_traffic"], r["valu_frac"])) < 1e-9 and r["frac_algorithmic"] > 1
    assert 0 < r["iteration_frac"] <= 1 and r["kernel_ms"] > 0
    # every kernel's VALU figure is a fraction of its roof now (a flat 4 cycles had K1 and K3b above 1)
    detail = json.load(open(tmp_path / "bench_detail.json"))
    for k, pk in detail["roofline"]["per_kernel"].items():
        if "valu" in pk:
            assert 0 < pk["valu"]["frac"] <= 1.0, (k, pk["valu"])
    cb = back["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["cores"] == 32 and cb["host_cores"] == 256 and cb["value"] > 0 and cb["all_cores"]["cores"] == 256
    names = [c["name"] for c in back["configs"]]
    assert names[:3] == ["C1", "C1x4", "C2"] and "C3" in names
    for c in back["configs"]:
        if "error" in c:
            assert len(c["error"]) <= 60
            continue
        assert c["value"] > 0 and c["ms_per_step"] > 0 and 0 < c["iteration_frac"] <= 1
        assert c["frac"] is None or 0 < c["frac"] <= 1, c     # C3 / C4x2 without counters: null, never 1.09
    assert [c for c in back["configs"] if c["name"] == "C3"][0]["frac"] is None


def test_the_line_without_counters_says_null_not_more_than_one():
    full = _full_from_fixture(with_counters=False)
    line = bench.short_line(full, None)
    assert line["roofline"]["frac"] is None and line["roofline"]["frac_algorithmic"] > 1 and "frac_note" in line["roofline"]
    assert len(json.dumps(line, allow_nan=False)) < bench.LINE_LIMIT


def test_the_line_survives_nan_and_oversized_fields():
    full = _full_from_fixture()
    full["roofline"]["iteration_frac"] = float("nan")
    full["config"]["workload"] = "w" * 5000
    full["config"]["parallelism"] = "p" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["configs"] = full["configs"] * 6
    text = json.dumps(bench.short_line(full, "bench_detail.json"), allow_nan=False)
    assert len(text) < bench.LINE_LIMIT and json.loads(text)["roofline"]["iteration_frac"] is None


def test_recorded_counters_need_the_same_kernel_sources_and_configuration(tmp_path, monkeypatch):
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "aaaa")
    bench.save_counters("r98", "C1", "scene 1 -a vcm 512x512", {"vcm::k_merge_walk": {"FETCH_SIZE": 1.0}}, "test")
    got, path = bench.recorded_counters("C1")
    assert got == {"vcm::k_merge_walk": {"FETCH_SIZE": 1.0}} and path.endswith("r98_counters_C1.json")
    assert bench.recorded_counters("C3") == (None, None)
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "bbbb")
    assert bench.recorded_counters("C1") == (None, None)


def test_the_line_of_a_multi_gpu_run_carries_both_decompositions():
    """N > 1: the default is north_star's decomposition ("strong"), the replica hybrid beside it; both fit the line"""
    full = _full_from_fixture()
    full.update({"n_gpus": 8, "scaling": "strong", "rank_iteration_ms": [1.1] * 8,
                 "hybrid_decomposition": {"value": 7000.0, "unit": "Mpaths/s", "scaling": "weak", "shards": 2, "inflight": 2, "ms_per_step": 9.5,
                                          "paths_per_step": 8 * 8388608, "parallelism": "x" * 400, "rccl_ranks": 8, "rank_iteration_ms": [9.0] * 8}})
    full.pop("cpu_baseline")
    full.pop("configs")
    line = bench.short_line(full, "bench_detail.json")
    text = json.dumps(line, allow_nan=False)
    assert len(text) < bench.LINE_LIMIT and line["scaling"] == "strong" and line["n_gpus"] == 8
    assert line["hybrid_decomposition"] == {"value": 7000.0, "ms_per_step": 9.5, "scaling": "weak", "paths_per_step": 8 * 8388608, "shards": 2, "inflight": 2}


def test_a_kernel_launched_three_times_per_iteration_counts_three_times():
    """the counter files hold means per DISPATCH; the radix sort's scatter runs once per digit (bench._sum_over)"""
    c = {"vcm::k_resolve": {"FETCH_SIZE": 10.0, "_n_fetch": 25, "SQ_INSTS_VALU": 5.0, "_us_valu": 1.0, "_n_valu": 25},
         "vcm::k_radix_scatter": {"FETCH_SIZE": 2.0, "_n_fetch": 75, "SQ_INSTS_VALU": 7.0, "_us_valu": 3.0, "_n_valu": 75},
         "vcm::k_radix_hist": {"FETCH_SIZE": 1.0, "_n_fetch": 50, "SQ_INSTS_VALU": 1.0, "_us_valu": 0.5, "_n_valu": 50},
         "vcm::k_cell_keys": {"FETCH_SIZE": 4.0, "_n_fetch": 25, "SQ_INSTS_VALU": 2.0, "_us_valu": 2.0, "_n_valu": 25}}
    pre = ["vcm::k_cell_", "vcm::k_radix_"]
    assert bench._sum_over(c, pre, "FETCH_SIZE") == 4.0 + 3 * 2.0 + 2 * 1.0
    assert bench._sum_over(c, pre, "SQ_INSTS_VALU") == 2.0 + 3 * 7.0 + 2 * 1.0
    assert bench._sum_over(c, pre, "_us_valu") == 2.0 + 3 * 3.0 + 2 * 0.5
    assert bench._sum_over(c, ["vcm::k_resolve"], "FETCH_SIZE") == 10.0
    assert bench._sum_over(c, ["vcm::k_absent"], "FETCH_SIZE") is None
    # the recorded set of the final sources: K2 = keys + 3 scatter passes + 2 histograms + starts + gather
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r30_counters_C4.json")
    if os.path.exists(path):
        k = json.load(open(path))["kernels"]
        grid = ["vcm::k_cell_", "vcm::k_radix_", "vcm::k_grid_", "vcm::k_bbox"]
        gb = (2 * bench._sum_over(k, grid, "FETCH_SIZE") + bench._sum_over(k, grid, "WRITE_SIZE")) * 1024 / 1e9
        assert 3.0 < gb < 4.6, gb
