"""Run by tests/test_gpu_dropin_sharded.py on the GPU box: the RCCL plumbing of the multi-GPU host with ONE rank
(RCCL refuses two ranks on one GPU, and the box has one): process-group init as bench.py does it, collectives
enqueued on the context's own stream (stream_context), async all-gather overlapped with the camera trace,
sub-groups, the framebuffer all-reduce.  Prints OK."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smallvcm_amd._abi import VCM_MERGE_RECORD_FLOATS  # noqa: E402
from smallvcm_amd.renderer import HipBackend, RenderFarm, cornell_scene  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sc = cornell_scene(1, 96, 96)
b = HipBackend(sc, 4, 0.003, 0.75, 1234)
with b.stream_context():
    b.begin(0, 0, 10)
    b.trace_light()
    n = b.local_record_count()
    cnt = torch.tensor([n], dtype=torch.int64, device="cuda")
    cnts = torch.empty(1, dtype=torch.int64, device="cuda")
    dist.all_gather_into_tensor(cnts, cnt)
    assert int(cnts[0]) == n and n > 0
    local = b.new_tensor(n * VCM_MERGE_RECORD_FLOATS)
    gathered = b.new_tensor(n * VCM_MERGE_RECORD_FLOATS)
    b.export_records(local, n)
    work = dist.all_gather_into_tensor(gathered, local, async_op=True)
    b.trace_camera()
    work.wait()
    b.build_grid()
    b.merge()
    b.end()
    assert torch.equal(gathered, local)
    assert np.array_equal(gathered.cpu().numpy().reshape(n, -1).view(np.uint32), b.records().view(np.uint32))
b.close()

grp = dist.new_group(ranks=[0])
farm = RenderFarm(lambda seed, s, S: HipBackend(sc, 4, 0.003, 0.75, seed, device=0, rank=s, world=S), 1234, 0, 1, shards=1,
                  dist=dist)
farm.set_path_lengths(0, 10)
farm.render(2)
fb = farm.framebuffer()
t = torch.from_numpy(fb.copy()).cuda()
dist.all_reduce(t, group=grp)
assert np.array_equal(t.cpu().numpy(), fb) and fb.mean() > 0
farm.backend.close()
dist.barrier()
dist.destroy_process_group()
print("OK")
