"""Worker of tests/test_sharded_cpu.py::test_space_sharded_merge_protocol: ONE rank of the merge sharded by SPACE
(include/smallvcm_amd.h "merge sharded by SPACE", smallvcm_amd/host/vcm_farm.cpp step_finish_space), restated in numpy over gloo --
the slabs of un-hashed cells cut from the summed histogram, light vertices to the owners of their cell + one cell of halo in index
order, queries to the owner of their base cell, HashGrid::Process (hashgrid.hxx:110-169: 8 hashed probes toward the nearer faces,
duplicates included, vertices of a cell in index order) against the OWNER's vertices only, the accepted sequences back to the
pixels' owner.  Rank 0 compares every query's accepted sequence with the same walk over ALL vertices.
Usage: space_worker.py rank world port seed nPhotons nQueries out.json"""
import json
import sys

import numpy as np
import torch
import torch.distributed as dist

rank, world, port, seed, nP, nQ = (int(x) for x in sys.argv[1:7])
out = sys.argv[7]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)

rng = np.random.default_rng(seed)
# photons and queries on the walls of a unit box (surfaces: whole cell LAYERS are full, as in the Cornell scenes), same on every rank
def on_walls(n):
    p = rng.random((n, 3)).astype(np.float32)
    side = rng.integers(0, 6, n)
    for k in range(3):
        p[side == 2 * k, k] = np.float32(0.001)
        p[side == 2 * k + 1, k] = np.float32(0.999)
    return p
P, Q = on_walls(nP), on_walls(nQ)
radius = np.float32(0.04)
cell, inv = np.float32(2) * radius, np.float32(1) / (np.float32(2) * radius)
nCells = nQ                      # vertexcm.hxx:406: as many buckets as paths
bmin, bmax = P.min(0), P.max(0)  # the box of ALL vertices (the 7-number exchange of the real protocol)

def cell_of(x):                  # hashgrid.hxx:189-193 per axis
    return np.floor(inv * (x - bmin)).astype(np.int64)
def hash_cell(c):                # hashgrid.hxx:179-187
    x, y, z = (c[..., k].astype(np.uint64) & np.uint64(0xffffffff) for k in range(3))
    h = ((x * np.uint64(73856093)) ^ (y * np.uint64(19349663)) ^ (z * np.uint64(83492791))) & np.uint64(0xffffffff)
    return (h % np.uint64(nCells)).astype(np.int64)

def process(queries, photons, photon_ids):
    """accepted photon ids per query, in HashGrid::Process's order, over the given photons (ids ascending = index order)"""
    hp = hash_cell(cell_of(photons))
    order = np.argsort(hp, kind="stable")
    hs = hp[order]
    res = []
    for q in queries:
        dmin, dmax = q - bmin, bmax - q
        if (dmin < 0).any() or (dmax < 0).any():
            res.append([]); continue
        cp = inv * dmin
        cf = np.floor(cp)
        base = cf.astype(np.int64)
        other = base + np.where(cp - cf < np.float32(0.5), -1, 1)
        acc = []
        for j in range(8):
            c = np.array([other[0] if j & 4 else base[0], other[1] if j & 2 else base[1], other[2] if j & 1 else base[2]])
            h = hash_cell(c)
            lo, hi = np.searchsorted(hs, h, "left"), np.searchsorted(hs, h, "right")
            for i in order[lo:hi]:
                d = q - photons[i]
                if np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) <= radius * radius:
                    acc.append(int(photon_ids[i]))
        res.append(acc)
    return res

# this rank's index blocks
def block(n):
    lo = rank * n // world
    return lo, (rank + 1) * n // world
p0, p1 = block(nP)
q0, q1 = block(nQ)
axis = int(np.argmax(bmax - bmin))
# slabs: equal photon counts from the SUM of the ranks' 256-bin histograms (vcm_space_histogram + the second exchange)
lo_w, bw = np.float32(-0.5), np.float32(2.0 / 256)
hist = torch.from_numpy(np.bincount(np.clip(((P[p0:p1, axis] - lo_w) / bw).astype(np.int64), 0, 255), minlength=256).astype(np.int64))
dist.all_reduce(hist)
hist = hist.numpy()
X = [-(1 << 30)]
cum, b = 0, 0
for s in range(1, world):
    want = hist.sum() * s // world
    while b < 256 and cum + hist[b] <= want:
        cum += hist[b]; b += 1
    X.append(max(X[-1], int(np.floor(inv * (np.float32(lo_w + bw * b) - bmin[axis])))))
X.append(1 << 30)
def owner(c):   # slab s = cells [X[s], X[s+1])
    return int(np.searchsorted(np.array(X[1:-1]), c, "right"))

# 1. light vertices to the owners of cells c-1 .. c+1, index order inside a destination
cx = cell_of(P[p0:p1])[:, axis]
send = [[] for _ in range(world)]
for i, c in enumerate(cx):
    for d in range(owner(c - 1), owner(c + 1) + 1):
        send[d].append(p0 + i)
gathered = [None] * world
dist.all_gather_object(gathered, send)
mine_ids = np.array([i for r in range(world) for i in gathered[r][rank]], dtype=np.int64)   # source-rank order = global index order
assert (np.diff(mine_ids) > 0).all()
# 2. queries to the owner of their base cell (those outside the box have no merge at all)
qsend, qwhere = [[] for _ in range(world)], {}
for i in range(q0, q1):
    dmin, dmax = Q[i] - bmin, bmax - Q[i]
    if (dmin < 0).any() or (dmax < 0).any():
        continue
    d = owner(int(np.floor(inv * dmin[axis])))
    qwhere[i] = (d, len(qsend[d]))
    qsend[d].append(i)
qg = [None] * world
dist.all_gather_object(qg, qsend)
# 3. the owner evaluates what it was sent, against ITS vertices
answers = [process(Q[np.array(qg[r][rank], dtype=np.int64)] if qg[r][rank] else [], P[mine_ids], mine_ids) for r in range(world)]
ag = [None] * world
dist.all_gather_object(ag, answers)
# 4. back at the pixels' owner
result = {i: ag[d][rank][pos] for i, (d, pos) in qwhere.items()}
allres = [None] * world
dist.all_gather_object(allres, result)
if rank == 0:
    got = {}
    for r in allres:
        got.update(r)
    ref = process(Q, P, np.arange(nP))
    bad = sum(1 for i in range(nQ) if got.get(i, []) != ref[i])
    json.dump({"queries": nQ, "mismatches": bad, "accepted": sum(len(x) for x in ref), "sent_photons": int(sum(len(x) for r in gathered for x in r)),
               "fullest_slab": int(max(len(np.array([i for r in range(world) for i in gathered[r][d]])) for d in range(world))), "slabs": X[1:-1]}, open(out, "w"))
dist.barrier()
dist.destroy_process_group()
