"""The plan of K2's radix sort (smallvcm_amd/csrc/vcm_kernels.h, "K2 as a radix sort"; HashGrid::Build, src/hashgrid.hxx:41-107)
restated in numpy, step for step as the kernels do it -- chunks per workgroup, the digit histogram matrix [digit][workgroup] and
its ONE linear scan, tiles of 2048 entries cut into four contiguous wave quarters, rank = running count of the wave + lanes before
me with my digit, entries staged by digit, written out in runs, cellStart from the sorted keys -- and checked against what the
reference's counting sort produces: a STABLE sort of the vertices by cell and the exclusive scan of the cell counts.

This is a model of the launch arithmetic (what a workgroup owns, where its entries go), not the product: the kernels themselves
are compared with the oracle on the GPU (tests/test_gpu_parity.py, tests/test_gpu_switches.py: SMALLVCM_AMD_GRID_SORT*)."""
import numpy as np
import pytest

TILE = 2048   # VCM_RSORT_TILE


def radix_chunk(n, V):
    return (((n + V - 1) // V) + 255) & ~255


def chunk_of(b, n, V):
    chunk = radix_chunk(n, V)
    lo = min(n, b * chunk)
    hi = n if n - lo < chunk else lo + chunk
    return lo, hi


def hist_matrix(key, shift, V):
    """k_cell_keys / k_radix_hist: hist[digit * V + workgroup]"""
    n = len(key)
    h = np.zeros(256 * V, np.int64)
    for b in range(V):
        lo, hi = chunk_of(b, n, V)
        d = (key[lo:hi] >> shift) & 255
        h[np.arange(256) * V + b] = np.bincount(d, minlength=256)
    return h


def scatter(key, pay, shift, V, scanned):
    """k_radix_scatter"""
    n = len(key)
    out_k, out_p = np.full(n, -1, np.int64), np.full(n, -1, np.int64)
    for b in range(V):
        lo, hi = chunk_of(b, n, V)
        g = scanned[np.arange(256) * V + b].copy()          # sGlobal
        for t0 in range(lo, hi, TILE):
            m = min(TILE, hi - t0)
            q = ((m + 255) >> 8) << 6                          # entries per wave: contiguous quarters
            run = np.zeros((4, 256), np.int64)                 # sRun
            rank = np.zeros(m, np.int64)
            wave_of = np.zeros(m, np.int64)
            for w in range(4):
                wlo, whi = t0 + w * q, min(t0 + m, t0 + w * q + q)
                for r0 in range(wlo, whi, 64):                 # one round of the wave
                    idx = np.arange(r0, min(r0 + 64, whi))
                    d = (key[idx] >> shift) & 255
                    for lane, (i, dd) in enumerate(zip(idx, d)):
                        before = int(np.sum(d[:lane] == dd))   # popc(peers & lanes below)
                        rank[i - t0] = run[w, dd] + before
                        wave_of[i - t0] = w
                    for dd in np.unique(d):                    # the first lane of each group adds the group
                        run[w, dd] += int(np.sum(d == dd))
            total = run.sum(axis=0)
            start = np.concatenate(([0], np.cumsum(total)[:-1]))          # sBinStart
            wave_base = start[None, :] + np.concatenate((np.zeros((1, 256), np.int64), np.cumsum(run, axis=0)[:-1]))
            stage_k, stage_p = np.full(m, -1, np.int64), np.full(m, -1, np.int64)
            for j in range(m):
                dd = (key[t0 + j] >> shift) & 255
                pos = wave_base[wave_of[j], dd] + rank[j]
                assert stage_k[pos] == -1
                stage_k[pos], stage_p[pos] = key[t0 + j], pay[t0 + j]
            for j in range(m):
                dd = (stage_k[j] >> shift) & 255
                dst = g[dd] + (j - start[dd])
                assert out_k[dst] == -1
                out_k[dst], out_p[dst] = stage_k[j], stage_p[j]
            g += total
    return out_k, out_p


def cell_starts(key, n_cells):
    """k_cell_starts: position pos starts every cell in (key[pos-1], key[pos]]; position n stands for key = nCells"""
    n = len(key)
    cs = np.full(n_cells + 1, -1, np.int64)
    for pos in range(n + 1):
        prev = int(key[pos - 1]) if pos > 0 else -1
        cur = int(key[pos]) if pos < n else n_cells
        cs[prev + 1:cur + 1] = pos
    return cs


def sort_cells(cells, n_cells, V):
    key, pay = cells.astype(np.int64), np.arange(len(cells), dtype=np.int64)
    bits = 1
    while bits < 31 and (1 << bits) < n_cells:
        bits += 1
    for p in range((bits + 7) // 8):
        h = hist_matrix(key, 8 * p, V)
        scanned = np.concatenate(([0], np.cumsum(h)[:-1]))
        key, pay = scatter(key, pay, 8 * p, V, scanned)
    return key, pay, cell_starts(key, n_cells)


@pytest.mark.parametrize("n,n_cells,V", [
    (0, 64, 1), (1, 1, 1), (5, 3, 7), (300, 256, 1), (2049, 257, 1),      # empty, one cell, fewer vertices than workgroups, one pass, a tile + 1
    (5000, 70000, 3), (6000, 4096, 64), (9000, 1 << 17, 5),                # three passes; more workgroups than chunks; chunks that are no tile multiple
])
def test_radix_plan_is_the_references_stable_counting_sort(n, n_cells, V):
    rng = np.random.default_rng(1234 + n)
    # a crowded grid: most vertices in few cells (what a point light or a caustic does), the rest anywhere
    hot = rng.integers(0, n_cells, max(1, n_cells // 50))
    cells = np.where(rng.random(n) < 0.7, hot[rng.integers(0, len(hot), n)], rng.integers(0, n_cells, n)) if n else np.zeros(0, np.int64)
    key, pay, cs = sort_cells(cells, n_cells, V)
    order = np.argsort(cells, kind="stable")                  # hashgrid.hxx:83-88: in a cell, vertices keep their index order
    assert np.array_equal(pay, order)
    assert np.array_equal(key, cells[order])
    counts = np.bincount(cells, minlength=n_cells) if n else np.zeros(n_cells, np.int64)
    assert np.array_equal(cs, np.concatenate(([0], np.cumsum(counts))))   # :75-81, and cellStart[nCells] = n


def test_chunks_cover_the_vertices_once_and_in_order():
    for n in (0, 1, 255, 256, 257, 70001, 4500000):
        for V in (1, 7, 64, 2048, 4096):
            spans = [chunk_of(b, n, V) for b in range(V)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all((hi - lo) % 256 == 0 or hi == n for lo, hi in spans)
