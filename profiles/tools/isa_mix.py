#!/usr/bin/env python
"""Static instruction mix of the gfx950 kernels by SOURCE FUNCTION (no GPU needed).

    hipcc <Makefile FLAGS> -gline-tables-only -S --cuda-device-only -o /tmp/vcm_api_g.s smallvcm_amd/csrc/vcm_api.hip
    python profiles/tools/isa_mix.py /tmp/vcm_api_g.s [kernel-substring ...]

Every instruction of a kernel is attributed to the source function whose line range contains its innermost .loc
(file, line); per function: instructions, and a cost in issue cycles per wave (fp32 VALU 2, packed/fp64 and
transcendental 4 or more: MI355X_MICROARCH.md "Per-instruction cycle constants").  Static counts, not executed ones:
they show where the instruction budget of a kernel sits (what DESIGN.md section 5 calls the VALU diet)."""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2:]

# function line ranges of our sources
def func_ranges(fn):
    out = []
    src = open(fn).read().split("\n")
    name, start, depth, base = None, 0, 0, 0
    for i, l in enumerate(src, 1):
        code = re.sub(r"//.*|/\*.*?\*/", "", l)
        if name is None:
            m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:VCM_HD|__device__ __forceinline__|__global__|inline|static)[^;{]*?\b([A-Za-z_0-9]+)\s*\(", code)
            if m and "(" in code:
                name, start, base, opened = m.group(1), i, depth, False
        depth += code.count("{") - code.count("}")
        if name is not None:
            if depth > base or "{" in code:
                opened = True
            if opened and depth == base:
                out.append((start, i, name))
                name = None
            elif not opened and code.rstrip().endswith(";"):
                name = None
    return out

ROOT = "/root/repo/smallvcm_amd/csrc/"
files = {}
ranges = {}
cost_of = lambda op: (8 if op.startswith(("v_div_", "v_rcp_f64", "v_sqrt_f64", "v_rsq_f64")) and "f64" in op else
                      4 if ("f64" in op or op.startswith(("v_pk_", "v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos", "v_mul_hi", "v_mul_lo", "v_mad_u64", "v_mad_i64"))) else
                      2 if op.startswith("v_") else 1)

kernel = None
cur = ("?", 0)
stats = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
totals = collections.defaultdict(lambda: [0, 0, 0])
for line in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', line)
    if m:
        files[int(m.group(1))] = m.group(3)
        continue
    m = re.match(r"^(_Z[A-Za-z0-9_]+):", line)
    if m:
        kernel = m.group(1)
        continue
    if line.startswith(".Lfunc_end"):
        kernel = None
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    if kernel is None:
        continue
    m = re.match(r"^\t([a-z][a-z0-9_]+)", line)
    if not m:
        continue
    op = m.group(1)
    f, ln = cur
    base = f.split("/")[-1]
    if base not in ranges:
        try:
            ranges[base] = func_ranges(ROOT + base)
        except OSError:
            ranges[base] = []
    fn = base
    for a, b, n in ranges[base]:
        if a <= ln <= b:
            fn = base + ":" + n
            break
    s = stats[kernel][fn]
    s[0] += 1
    s[1] += cost_of(op)
    t = totals[kernel]
    t[0] += 1
    t[1] += cost_of(op)
    if op.startswith("v_"):
        t[2] += 1

for k in stats:
    if want and not any(w in k for w in want):
        continue
    t = totals[k]
    print("== %s: %d instructions (%d VALU), %d issue cycles if each ran once" % (k, t[0], t[2], t[1]))
    for fn, (n, c) in sorted(stats[k].items(), key=lambda x: -x[1][1])[:28]:
        print("   %-52s %6d instr %7d cyc  %5.1f %%" % (fn, n, c, 100.0 * c / t[1]))
