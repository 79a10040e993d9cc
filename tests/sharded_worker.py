"""Worker for tests/test_sharded_cpu.py: one rank of ShardedVertexCM over gloo,
computing with the ORACLE as backend (test infrastructure; the product backend
is HipBackend).  Usage: sharded_worker.py rank world port scene algo res iters out.npy [shards [inflight]]
With `shards` the rank is one member of a RenderFarm (world/shards replica groups)."""
import contextlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle_lib import Oracle  # noqa: E402
from smallvcm_amd._abi import VCM_MERGE_RECORD_FLOATS  # noqa: E402
from smallvcm_amd.renderer import RenderFarm, ShardedVertexCM, cornell_scene  # noqa: E402


class OracleBackend:
    """HipBackend's phase interface on top of oracle/liboracle.so (CPU tensors)."""

    def __init__(self, scene, algo, rank, world, seed=1234):
        self.o = Oracle(scene, algo, seed=seed, rank=rank, world=world)
        self.resx, self.resy, self.N = self.o.resx, self.o.resy, self.o.N
        self.first, self.count = self.o.first, self.o.count

    def stream_context(self):
        return contextlib.nullcontext()

    def begin(self, it, mn, mx):
        self.o.begin(it, mn, mx)

    def trace_light(self):
        self.o.trace_light()

    def build_grid(self):
        self.o.build_grid()

    camera_before_grid = False   # the oracle's camera pass merges inline

    def trace_camera(self):
        self.o.trace_camera()

    def merge(self):
        pass

    def end(self):
        self.o.end()

    def new_tensor(self, n):
        return torch.zeros(int(n), dtype=torch.float32)

    def local_record_count(self):
        return self.o.records().shape[0]

    def export_records(self, dst, count):
        if count:
            dst[:count * VCM_MERGE_RECORD_FLOATS] = torch.from_numpy(self.o.records().ravel())

    def import_records(self, gathered, counts, stride):
        g = gathered.numpy().reshape(len(counts), stride, VCM_MERGE_RECORD_FLOATS)
        self.o.import_records(np.concatenate([g[s, :c] for s, c in enumerate(counts)], axis=0))

    def synchronize(self):
        pass

    def close(self):
        pass

    def export_framebuffer(self, dst):
        dst.copy_(torch.from_numpy(self.o.framebuffer().ravel()))


def sorted_block_cells(S):
    """cells per block of the sorted exchange (smallvcm_amd/csrc/vcm_kernels.h sorted_block_cells)"""
    k, p = 4096 // max(S, 1), 16
    while p * 2 <= k and p < 1024:
        p *= 2
    return p


class SortedOracleBackend(OracleBackend):
    """+ the SORTED exchange of round 5 (include/smallvcm_amd.h vcm_sort_light_records / vcm_import_sorted_light_records),
    restated in numpy so that ShardedVertexCM's sorted path and the slab layout run over gloo on the CPU: a rank's records
    in cell order (stable: local vertex order inside a cell), word 12 = path length | local index << 8, behind them the start
    of every block of K cells; the receiver places every record at cellStart[c] + (records of lower ranks in c) + (its
    place in its rank's run) -- which must be the stable sort by cell of ALL records in the reference's order
    (HashGrid::Build, hashgrid.hxx:83-88) -- and hands them to the oracle in that reference order.  The cell size is a
    fixed stand-in (the oracle does not expose the iteration's radius): the layout and the placement are what is tested here,
    the device's cells are tested on the GPU (tests/test_gpu_dropin_sharded.py)."""
    INV_CELL = np.float32(1.0) / np.float32(2.0 * 0.0066512)

    def __init__(self, scene, algo, rank, world, seed=1234):
        super().__init__(scene, algo, rank, world, seed=seed)
        self.world = world
        self.K = sorted_block_cells(world)
        self.n_cells = self.N                                   # vertexcm.hxx:406
        self.n_blocks = (self.n_cells + self.K - 1) // self.K

    def local_bbox(self):
        recs = self.o.records()
        if len(recs) == 0:
            return [1e36] * 3, [-1e36] * 3, 0                   # hashgrid.hxx:47-48
        return [float(x) for x in recs[:, :3].min(axis=0)], [float(x) for x in recs[:, :3].max(axis=0)], len(recs)

    def set_grid_bbox(self, mn, mx):
        self.bmin = np.array(mn, np.float32)

    def _cells(self, pos):                                      # hashgrid.hxx:179-201
        f = np.floor(self.INV_CELL * (pos.astype(np.float32) - self.bmin)).astype(np.int64) & 0xffffffff
        h = ((f[:, 0] * 73856093) & 0xffffffff) ^ ((f[:, 1] * 19349663) & 0xffffffff) ^ ((f[:, 2] * 83492791) & 0xffffffff)
        return (h % self.n_cells).astype(np.int64)

    def sorted_slab_words(self, stride):
        if os.environ.get("SMALLVCM_AMD_SORTED_EXCHANGE", "1") == "0" or not (1 <= stride < (1 << 24)):
            return -1
        return (stride * 13 + self.n_blocks + 1 + 3) & ~3

    def sort_records(self, dst, stride):
        recs = self.o.records()
        n = len(recs)
        slab = np.zeros(self.sorted_slab_words(stride), np.uint32)
        cells = self._cells(recs[:, :3]) if n else np.zeros(0, np.int64)
        order = np.argsort(cells, kind="stable")
        w = recs[order].view(np.uint32).copy()
        if n:
            w[:, 12] = (w[:, 12] & 0xff) | (order.astype(np.uint32) << 8)
        slab[:n * 13] = w.ravel()
        edges = np.minimum(np.arange(self.n_blocks + 1, dtype=np.int64) * self.K, self.n_cells)
        slab[stride * 13:stride * 13 + self.n_blocks + 1] = np.searchsorted(cells[order], edges, side="left").astype(np.uint32)
        dst[:len(slab)] = torch.from_numpy(slab.view(np.float32))

    def import_sorted_records(self, gathered, counts, stride):
        S, words = len(counts), self.sorted_slab_words(stride)
        slabs = gathered.numpy().view(np.uint32).reshape(S, words)
        base = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        total = int(base[-1])
        cnt = np.zeros((S, self.n_cells), np.int64)
        recs, cells = [], []
        for r in range(S):
            w = slabs[r, :counts[r] * 13].reshape(counts[r], 13)
            c = self._cells(w[:, :3].view(np.float32)) if counts[r] else np.zeros(0, np.int64)
            assert np.all(np.diff(c) >= 0), "a slab is in cell order"
            edges = np.minimum(np.arange(self.n_blocks + 1, dtype=np.int64) * self.K, self.n_cells)
            assert np.array_equal(slabs[r, stride * 13:stride * 13 + self.n_blocks + 1].astype(np.int64), np.searchsorted(c, edges, side="left")), "block starts"
            cnt[r] = np.bincount(c, minlength=self.n_cells)
            recs.append(w)
            cells.append(c)
        cell_start = np.concatenate([[0], np.cumsum(cnt.sum(axis=0))])          # hashgrid.hxx:75-81
        before = np.cumsum(cnt, axis=0) - cnt                                       # records of lower ranks in the cell
        placed = np.zeros((total, 13), np.uint32)
        index = np.zeros(total, np.int64)
        for r in range(S):
            local_start = np.concatenate([[0], np.cumsum(cnt[r])])
            i = np.arange(counts[r], dtype=np.int64)
            dst = cell_start[cells[r]] + before[r][cells[r]] + (i - local_start[cells[r]])
            placed[dst] = recs[r]
            index[dst] = base[r] + (recs[r][:, 12] >> 8)
        assert len(np.unique(index)) == total
        # = HashGrid::Build's stable counting sort over ALL records in the reference's order
        ref = np.zeros((total, 13), np.uint32)
        ref[index] = placed
        ref[:, 12] &= 0xff
        all_cells = self._cells(ref[:, :3].view(np.float32)) if total else np.zeros(0, np.int64)
        assert np.array_equal(index, np.argsort(all_cells, kind="stable")), "in-cell order = vertex order, rank-major"
        self.o.import_records(ref.view(np.float32))


def main():
    rank, world, port, sid, algo, res, iters = (int(x) for x in sys.argv[1:8])
    out = sys.argv[8]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = cornell_scene(sid, res, res)
    if len(sys.argv) > 9:
        farm = RenderFarm(lambda seed, s, S: OracleBackend(sc, algo, s, S, seed=seed), 1234, rank, world,
                          shards=int(sys.argv[9]), dist=dist, inflight=int(sys.argv[10]) if len(sys.argv) > 10 else None)
        farm.set_path_lengths(0, 10)
        farm.render(iters)
        fb = farm.framebuffer()
    else:
        backend = SortedOracleBackend if os.environ.get("SHARDED_WORKER_SORTED") == "1" else OracleBackend
        r = ShardedVertexCM(backend(sc, algo, rank, world), rank, world)
        r.mMaxPathLength, r.mMinPathLength = 10, 0
        for it in range(iters):
            r.RunIteration(it)
        fb = r.framebuffer_sum()
    if rank == 0:
        np.save(out, fb)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
