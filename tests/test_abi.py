"""The C-ABI library loads (without a GPU) and exports every symbol that
include/smallvcm_amd.h declares; PODs have the sizes the ctypes mirror assumes."""
import ctypes as C
import os
import re
import subprocess

import pytest

from smallvcm_amd import _abi
from smallvcm_amd.renderer import LIB_PATH, load_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vcm_[a-z_0-9]+)\s*\(", src)))


def _exported_symbols():
    out = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], capture_output=True, text=True, check=True).stdout
    return sorted(l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T" and l.split()[2].startswith("vcm_"))


def test_library_is_built():
    assert os.path.exists(LIB_PATH), "run __graft_entry__.build() first"


def test_exports_every_declared_symbol():
    """include/smallvcm_amd.h is the drop-in boundary, include/smallvcm_amd_debug.h the test entry points"""
    L = load_library(require_gpu=False)
    names = _declared_symbols("smallvcm_amd.h")
    assert len(names) >= 25
    missing = [n for n in names + _declared_symbols("smallvcm_amd_debug.h") if not hasattr(L, n)]
    assert not missing, missing


def test_every_exported_symbol_is_declared():
    """no entry point outside the two headers: what a host can call is what the headers document"""
    declared = set(_declared_symbols("smallvcm_amd.h")) | set(_declared_symbols("smallvcm_amd_debug.h"))
    undeclared = [n for n in _exported_symbols() if n not in declared]
    assert not undeclared, undeclared
    # the boundary header itself carries no debug / test entry points
    assert not [n for n in _declared_symbols("smallvcm_amd.h") if n.startswith(("vcm_debug_", "vcm_host_", "vcm_sizeof_"))]


def test_farm_library_exports_what_its_header_declares():
    """include/smallvcm_amd_farm.h = the multi-GPU host's boundary (smallvcm_amd/host/libsmallvcm_amd_farm.so); loads
    without a GPU, exports exactly the declared entry points, PODs as the ctypes mirror assumes"""
    from smallvcm_amd import farm
    L = farm.load_farm_library()
    declared = [n for n in _declared_symbols("smallvcm_amd_farm.h") if n.startswith("vcm_farm_")]
    assert len(declared) >= 6
    out = subprocess.run(["nm", "-D", "--defined-only", farm.FARM_LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T" and l.split()[2].startswith("vcm_"))
    assert exported == sorted(declared), (exported, declared)
    assert L.vcm_farm_sizeof_config() == C.sizeof(farm.FarmConfig)
    assert L.vcm_farm_sizeof_result() == C.sizeof(farm.FarmResult)
    assert L.vcm_farm_unique_id_bytes() == 128


def test_farm_rejects_bad_configurations_without_a_gpu():
    """argument errors are reported before anything touches a device"""
    from smallvcm_amd import farm
    from smallvcm_amd.renderer import cornell_scene
    sc = cornell_scene(1, 16, 16)
    with pytest.raises(RuntimeError, match="multiple of shards"):
        farm.farm_render(sc, 4, iterations=4, ranks=3, shards=2, inflight=1, devices=[0, 0, 0], collectives="threads")
    with pytest.raises(RuntimeError, match="sameWindow"):
        farm.farm_render(sc, 4, iterations=3, ranks=2, shards=1, inflight=1, devices=[0, 0], collectives="threads", same_window=True)
    with pytest.raises(RuntimeError, match="stand-in"):
        farm.farm_render(sc, 4, iterations=2, ranks=2, shards=2, inflight=1, devices=[0], first_rank=0, collectives="threads")
    with pytest.raises(RuntimeError, match="ids"):
        farm.farm_render(sc, 4, iterations=2, ranks=2, shards=2, inflight=1, devices=[0], first_rank=1, collectives="rccl")


def test_pod_sizes_match_ctypes_mirror():
    L = load_library(require_gpu=False)
    assert L.vcm_sizeof_scene_desc() == C.sizeof(_abi.SceneDesc)
    assert L.vcm_sizeof_stats() == C.sizeof(_abi.Stats)


def test_no_cpu_fallback():
    """Without a GPU the product must refuse to create a renderer."""
    L = load_library(require_gpu=False)
    if L.vcm_device_count() > 0:
        pytest.skip("GPU present")
    from smallvcm_amd.renderer import VertexCM, cornell_scene
    sc = cornell_scene(1, 16, 16)
    with pytest.raises(RuntimeError):
        VertexCM(sc, VertexCM.kVcm, 0.003, 0.75)


def test_product_does_not_reference_oracle():
    """Nothing under smallvcm_amd/ (the shipped path) may include, import or link oracle/."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "smallvcm_amd")):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hxx", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                for line in txt.splitlines():
                    if re.search(r"#\s*include.*oracle|import\s+oracle|from\s+oracle|liboracle|oracle_lib", line):
                        bad.append((f, line.strip()))
    assert not bad, bad


@pytest.mark.parametrize("value", ["", "all", "current", "0", "0,0,0", "0,99,0", "7,8", "nonsense", "0,,"])
def test_the_device_list_of_vcm_next_device_is_parsed_safely(value):
    """SMALLVCM_AMD_DEVICES (which devices a renderer-per-host-core host is dealt, smallvcm.cxx:61-72) is read once
    per process: a process per value.  Whatever the variable holds, the answer is a device that exists -- device 0 on a box with
    one GPU or none -- three times in a row (the list is dealt round-robin)."""
    import sys
    code = ("import ctypes as C\n"
            "L = C.CDLL(%r)\n"
            "L.vcm_next_device.restype = C.c_int\n"
            "L.vcm_device_count.restype = C.c_int\n"
            "n = L.vcm_device_count()\n"
            "d = [L.vcm_next_device() for _ in range(3)]\n"
            "assert all(0 <= x < max(n, 1) for x in d), (n, d)\n"
            "print(n, d)\n" % LIB_PATH)
    env = dict(os.environ)
    env["SMALLVCM_AMD_DEVICES"] = value
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-400:]


def test_every_entry_point_refuses_a_null_context():
    """Every function of include/smallvcm_amd.h that takes a vcm_ctx* returns -1 (vcm_destroy: does nothing) for NULL: a host
    whose vcm_create failed and went unchecked gets errors, not a crash.  The prototypes are read from the header; in a
    process of its own."""
    import sys
    code = r'''
import ctypes as C, re
L = C.CDLL(%r)
hdr = re.sub(r'/\*.*?\*/', '', open(%r).read(), flags=re.S)
seen = 0
for m in re.finditer(r'\b(int|void)\s+(vcm_\w+)\s*\(\s*vcm_ctx\s*\*\s*\w+\s*([^)]*)\)\s*;', hdr):
    ret, name, rest = m.group(1), m.group(2), m.group(3)
    args = [C.c_void_p(None)]
    for p in [p.strip() for p in rest.split(',') if p.strip()]:
        if 'float' in p and '*' not in p: args.append(C.c_float(0.0))
        elif '*' in p: args.append(C.c_void_p(None))
        elif 'long long' in p: args.append(C.c_longlong(0))
        else: args.append(C.c_int(0))
    f = getattr(L, name)
    f.restype = C.c_int if ret == 'int' else None
    r = f(*args)
    seen += 1
    if ret == 'int' and name not in ('vcm_is_wavefront', 'vcm_iterations'): assert r == -1, (name, r)
assert seen >= 28, seen
''' % (LIB_PATH, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "smallvcm_amd.h"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-600:])
