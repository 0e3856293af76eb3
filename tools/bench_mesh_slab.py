#!/usr/bin/env python3
"""Developer tool: the slab-decomposed mesh CV (mtd_mesh_slab_*) between the ranks of a torch.distributed launch.
On a one-GPU box run it as a REHEARSAL (every rank on cuda:0, control plane gloo): the numbers then show the software
path (barriers, pulls through IPC mappings of the same HBM), not xGMI.

    MTD_BENCH_REHEARSAL=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/bench_mesh_slab.py [n_mesh] [steps]
"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import torch.distributed as dist
import util
from metadynamics import sharded, xgmi

n_mesh = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
rehearsal = os.environ.get("MTD_BENCH_REHEARSAL") == "1"
torch.cuda.set_device(0 if rehearsal else int(os.environ.get("LOCAL_RANK", "0")))
dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
N, L = 1_000_000, 100.0
pos, types = util.snapshot_random(N, L, seed=12345, modulated=True, dtype=np.float32)
pos = (np.mod(pos.astype(np.float64) + L / 2, L) - L / 2).astype(np.float32)
pos[pos >= L / 2] = -L / 2
sl = slice(rank * N // world, (rank + 1) * N // world)
box = xgmi.connect(dist, max_doubles=8)
assert box is not None, "no mailbox on this node"
dpos = torch.from_numpy(util.pack_postype(pos[sl].copy(), types[sl].copy(), np.float32)).cuda()
part = sharded.MeshSlabPart(n_mesh, n_mesh, n_mesh, [1.0, -1.0], dpos, N, L, box, dist)
grid = dict(sigma=[1e-5], cv_min=[0.0], cv_max=[1.0], num_points=[256])
be = sharded.HipCvSetBackend([part], grid, W=1.0, T_shift=7.0, T=1.0, stride=1, mode="well_tempered")
step = sharded.ShardedBiasStep(be, dist, mailbox=box)
for t in range(10):
    step.step(t)
torch.cuda.synchronize(); dist.barrier()
t0 = time.perf_counter()
for t in range(10, 10 + steps):
    step.step(t)
torch.cuda.synchronize(); dist.barrier()
dt = time.perf_counter() - t0
if rank == 0:
    print("mesh %d^3 in %d slabs, %d particles: %.1f us/step, cv = %.15g, mailbox timeouts %d" % (n_mesh, world, N, 1e6 * dt / steps, be.state()["cv"][0], box.timeouts()))
dist.barrier()
be.close(); box.close()
dist.destroy_process_group()
