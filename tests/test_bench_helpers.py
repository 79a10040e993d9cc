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
