"""Worker of tests/test_gpu_comm.py::test_mailbox_timeout_is_fatal: two processes on cuda:0, control plane gloo.  Both connect
the xGMI mailbox (self test included); then rank 1 stops sending.  Rank 0 runs one fused step with a short wait bound and
reports what the step left behind (JSON line).  The reference aborts on a failed MPI_Allreduce
(LamellarOrderParameterGPU.cc:69-77); here the step is poisoned and the communicator is dead."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "metadynamics-plugin_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

import util
from metadynamics import _abi, xgmi
from metadynamics.sharded import HipLamellarBackend

GRID = dict(sigma=[0.02, 0.02], cv_min=[-1.0, -1.0], cv_max=[1.0, 1.0], num_points=[64, 48])
KW = dict(W=1.0, T_shift=7.0, T=1.0, stride=1, mode="well_tempered")
CVS = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _abi.load()
    box = xgmi.connect(dist, max_doubles=8)
    out = {"connected": box is not None}
    if box is None:
        if rank == 0:
            print(json.dumps(out), flush=True)
        dist.destroy_process_group()
        return
    N, L = 20000, 30.0
    pos, types = util.snapshot_random(2 * N, L, seed=5, modulated=True, dtype=np.float32)
    sl = slice(rank * N, (rank + 1) * N)
    d_pos = torch.from_numpy(util.pack_postype(pos[sl].copy(), types[sl].copy(), np.float32)).cuda()
    be = HipLamellarBackend(CVS, d_pos, 2 * N, L, GRID, fast_trig=False, fused=True, **KW)
    be.attach_mailbox(box)
    # one good step on both ranks (a hill is deposited), then rank 1 goes silent
    be.step_single(0)
    torch.cuda.synchronize()
    good = be.state()
    dist.barrier()
    if rank == 0:
        n_part = C.c_uint()
        rc_a = lib.mtd_fused_cv_pass(be.h, C.byref(be.lset), be.N, d_pos.data_ptr(), be.dt, C.byref(be.box), be.scratch.data_ptr(), C.byref(n_part), None)
        rc_b = lib.mtd_fused_force_pass(be.h, C.byref(be.lset), be.N, d_pos.data_ptr(), be.fptr, be.dt, be.N_global, C.byref(be.box), 1, None)
        torch.cuda.synchronize()                       # the wait expires (MTD_COMM_TIMEOUT_MS) — it never hangs
        cv, bias = (C.c_double * 2)(), (C.c_double * 2)()
        V, w, ng = C.c_double(), C.c_double(), C.c_uint()
        rc_state = lib.mtd_metad_get_state(be.h, cv, bias, C.byref(V), C.byref(w), C.byref(ng), None, None)
        f = be.forces[0].cpu().numpy()
        grid = np.zeros(64 * 48)
        rc_grid = lib.mtd_metad_get_array(be.h, 0, grid.ctypes.data, None)
        # the communicator is dead: every call that would use it says so
        rc_next = lib.mtd_fused_cv_pass(be.h, C.byref(be.lset), be.N, d_pos.data_ptr(), be.dt, C.byref(be.box), be.scratch.data_ptr(), C.byref(n_part), None)
        v = torch.ones(1, dtype=torch.float64, device="cuda")
        rc_ar = lib.mtd_comm_allreduce_small(box.handle, v.data_ptr(), 1, None)
        out.update(rc_launch=[rc_a, rc_b], rc_state=rc_state, rc_grid=rc_grid, rc_next=rc_next, rc_allreduce=rc_ar,
                   timeouts=box.timeouts(), cv_nan=bool(np.isnan(cv[0]) and np.isnan(cv[1])),
                   bias_nan=bool(np.isnan(bias[0]) and np.isnan(bias[1])), V_nan=bool(np.isnan(V.value)),
                   forces_all_nan=bool(np.isnan(f[:, :3]).all()), num_gaussians=[good["num_gaussians"], ng.value],
                   grid_finite=bool(np.isfinite(grid).all()), grid_max=float(np.abs(grid).max()),
                   status_string=lib.mtd_status_string(-4).decode())
        print(json.dumps(out), flush=True)
    dist.barrier()
    be.attach_mailbox(None)
    be.close()
    box.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
