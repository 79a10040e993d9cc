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
        r = ShardedVertexCM(OracleBackend(sc, algo, rank, world), rank, world)
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
