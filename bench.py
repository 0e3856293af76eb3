#!/usr/bin/env python3
"""bench.py — headline benchmark of the metadynamics bias step on MI355X.

One "step" = one pass of the hot path over one synthetic particle snapshot (BASELINE.json configs[1]):
    lamellar CV values of both CVs (one fused pass over the positions, incl. reduction)
    -> bias-grid update on the device (histogram, well-tempered Gaussian deposit, reweighted
       estimator, accumulate, V(s), w(s), dV/ds_c)           [stride 1: every step deposits]
    -> bias forces of both CVs written for every particle (one fused pass).
Inputs are resident in HBM before the timed region; nothing synchronises with the host inside it.

    python bench.py --gpus N --steps K --warmup W
N > 1: launched by torch.distributed.run, one rank per GPU; particles are sharded (10^6 per rank,
weak scaling), the per-CV partial sums are all-reduced over RCCL each step, the bias grid is
replicated (every rank deposits the same hill, no grid collective).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "metadynamics-plugin_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

import util
from metadynamics import _abi

N_PER_GPU = 1_000_000
BOX_L = 100.0
GRID = dict(sigma=[1e-3, 1e-3], cv_min=[-0.02, -0.02], cv_max=[0.02, 0.02], num_points=[256, 256])
W, DELTA_T, T = 1.0, 7.0, 1.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--stride", type=int, default=1, help="hill deposition stride (headline: 1)")
    ap.add_argument("--particles", type=int, default=N_PER_GPU, help="particles per GPU")
    ap.add_argument("--fast-trig", type=int, default=int(os.environ.get("MTD_FAST_TRIG", "1")))
    ap.add_argument("--path", choices=["fused", "generic"], default="fused",
                    help="fused: two launches per step (headline); generic: separate C-ABI calls per stage")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=4)
    return ap.parse_args()


class Engine:
    """The hot path through the C-ABI, device resident."""

    def __init__(self, n_local, n_global, rank, seed, stride, fast_trig, dist=None, path="fused"):
        self.lib = lib = _abi.load()
        self.dist = dist
        self.fused = path == "fused"
        self.N, self.N_global = n_local, n_global
        L = BOX_L * (n_global / N_PER_GPU) ** (1.0 / 3.0)  # constant density (config 4: L = 200 at 8e6)
        pos, types = util.snapshot_random(n_global, L, seed=seed, dtype=np.float32) if dist is None else \
            self._shard(n_global, L, seed, rank, n_local)
        self.pos_np, self.types_np, self.L = pos, types, L
        self.box = _abi.Box.make(L)
        self.cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]
        self.lset = _abi.LamellarSet.make(self.cvs)
        self.d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
        self.scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(n_local), dtype=torch.float64, device="cuda")
        self.cv_sum = torch.zeros(2, dtype=torch.float64, device="cuda")
        self.forces = [torch.zeros((n_local, 4), dtype=torch.float32, device="cuda") for _ in self.cvs]
        self.fptr = (C.c_void_p * 2)(*[f.data_ptr() for f in self.forces])
        self.h = C.c_void_p()
        _abi.check(lib.mtd_metad_create(C.byref(self.h), 2, util.dbl_array(GRID["sigma"]), util.dbl_array(GRID["cv_min"]),
                                        util.dbl_array(GRID["cv_max"]), util.uint_array(GRID["num_points"]), W, DELTA_T,
                                        T, stride, _abi.MODE_WELL_TEMPERED, 1))
        self.d_bias = lib.mtd_metad_bias_device(self.h)
        self.n_part = C.c_uint()
        _abi.check(lib.mtd_lamellar_set_fast_trig(int(fast_trig)))
        self.t = 0
        self.ev = None

    @staticmethod
    def _shard(n_global, L, seed, rank, n_local):
        # every rank draws the same global snapshot and keeps its contiguous slice (lamellar CVs need no locality)
        pos, types = util.snapshot_random(n_global, L, seed=seed, dtype=np.float32)
        sl = slice(rank * n_local, (rank + 1) * n_local)
        return pos[sl].copy(), types[sl].copy()

    def _register_sources(self):
        lib = self.lib
        for c in range(2):
            if self.dist is None:
                _abi.check(lib.mtd_metad_set_cv_source(self.h, c, self.scratch.data_ptr(), self.n_part.value, 2, c,
                                                       1.0 / self.N_global, 0.0))
            else:
                _abi.check(lib.mtd_metad_set_cv_source(self.h, c, self.cv_sum.data_ptr(), 1, 2, c,
                                                       1.0 / self.N_global, 0.0))

    def step(self):
        lib, lset, box = self.lib, self.lset, self.box
        # launch A: per-CV partial sums over the particles (+ the deferred grid pass of the previous deposit)
        if self.fused:
            _abi.check(lib.mtd_fused_cv_pass(self.h, C.byref(lset), self.N, self.d_pos.data_ptr(), _abi.MTD_F32,
                                             C.byref(box), self.scratch.data_ptr(), C.byref(self.n_part), None))
        else:
            _abi.check(lib.mtd_lamellar_cv_partials(C.byref(lset), self.N, self.d_pos.data_ptr(), _abi.MTD_F32,
                                                    C.byref(box), self.scratch.data_ptr(), C.byref(self.n_part), None))
        if self.dist is not None:
            # local partial sums -> 2 doubles -> RCCL all-reduce -> the grid engine reads the reduced sums
            _abi.check(lib.mtd_reduce_partials(self.scratch.data_ptr(), self.n_part.value, 2, 2, 1.0, 0.0,
                                               self.cv_sum.data_ptr(), None))
            self.dist.all_reduce(self.cv_sum)
        if self.t == 0:
            self._register_sources()
        if self.ev is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # launch B: bias-grid update + bias forces
        if self.fused:
            if self.ev is not None:
                a.record()
            _abi.check(lib.mtd_fused_force_pass(self.h, C.byref(lset), self.N, self.d_pos.data_ptr(), self.fptr,
                                                _abi.MTD_F32, self.N_global, C.byref(box), self.t, None))
        else:
            _abi.check(lib.mtd_metad_update_bias(self.h, self.t, None))
            if self.ev is not None:
                a.record()
            _abi.check(lib.mtd_lamellar_forces(C.byref(lset), self.N, self.d_pos.data_ptr(), self.fptr, _abi.MTD_F32,
                                               self.N_global, self.d_bias, C.byref(box), None))
        if self.ev is not None:
            b.record()
            self.ev.append((a, b))
        self.t += 1

    def state(self):
        cv = (C.c_double * 2)()
        bias = (C.c_double * 2)()
        V, w = C.c_double(), C.c_double()
        ng = C.c_uint()
        _abi.check(self.lib.mtd_metad_get_state(self.h, cv, bias, C.byref(V), C.byref(w), C.byref(ng), None, None))
        return dict(cv=list(cv), bias=list(bias), V=V.value, w=w.value, num_gaussians=ng.value)


def cpu_baseline(pos, types, L, steps):
    """The CPU restatement (oracle, kind "port") of the same step on ONE host core: the reference CPU
    path is serial (MPI ranks only) and cannot be built here (needs HOOMD)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mtd_ref
    rbox = mtd_ref.Box.make(L)
    opt = util.oracle_postype(pos, types)
    r = mtd_ref.Metad(W=W, T_shift=DELTA_T, T=T, stride=1, mode="well_tempered", **GRID)
    cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]
    t0 = time.perf_counter()
    for t in range(steps):
        s = [mtd_ref.lamellar_cv(v, opt, m, rbox) for v, m in cvs]
        b = r.update_bias(t, s)
        for c, (v, m) in enumerate(cvs):
            mtd_ref.lamellar_forces(v, opt, m, rbox, b[c])
    dt = time.perf_counter() - t0
    return pos.shape[0] * len(cvs) * steps / dt, dt


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n_local = args.particles
    n_global = n_local * world
    eng = Engine(n_local, n_global, rank, seed=12345 if world == 1 else 12346, stride=args.stride,
                 fast_trig=args.fast_trig, dist=dist, path=args.path)

    for _ in range(args.warmup):
        eng.step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # dominant kernel (force pass): per-launch durations from events on the launch stream, same loop
    eng.ev = []
    n_ev = min(args.steps, 500)
    for _ in range(n_ev):
        eng.step()
    torch.cuda.synchronize()
    force_us = float(np.median([a.elapsed_time(b) for a, b in eng.ev]) * 1e3)
    force_us_mean = float(np.mean([a.elapsed_time(b) for a, b in eng.ev]) * 1e3)
    eng.ev = None
    st = eng.state()

    if rank == 0:
        steps_per_s = args.steps / elapsed
        value = steps_per_s * n_global * 2
        # dominant kernel's algorithmic bytes per launch: read Scalar4 positions + write one Scalar4
        # force per CV (48 B/particle); the fused kernel also carries the first grid pass of the deposit
        # (per cell: 8 B dV written, 8+8 B reweighted r/w, 4 B hist_delta read)
        force_bytes = n_local * (16 + 2 * 16)
        if args.path == "fused" and args.stride == 1:
            force_bytes += 256 * 256 * 28
        achieved = force_bytes / (force_us_mean * 1e-6) / 1e9
        out = {
            "metric": "particle_cv_evals_per_s",
            "value": value,
            "unit": "particle-CV-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "md_bias_steps_per_s": steps_per_s,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "1xMI355X: 10^6 particles, 2 lamellar CVs (8 Fourier modes each), 256^2 bias grid, well-tempered"
                       if world == 1 else "%dxMI355X: %d particles sharded, 2 lamellar CVs, RCCL all-reduce of CV sums, replicated 256^2 grid" % (world, n_global),
                       "particles_per_gpu": n_local, "n_cv": 2, "modes_per_cv": 8, "grid": "256x256",
                       "stride": args.stride, "fast_trig": int(args.fast_trig), "path": args.path},
            "roofline": {"bound": "hbm", "kernel": "k_fused_force" if args.path == "fused" else "k_lamellar_forces", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": force_bytes, "avg_launch_us": force_us_mean,
                         "median_launch_us": force_us},
            "state": st,
        }
        if not args.no_cpu_baseline:
            pos, types = eng.pos_np, eng.types_np
            v, dt = cpu_baseline(pos, types, eng.L, args.cpu_steps)
            out["cpu_baseline"] = {"value": v, "unit": "particle-CV-evals/s", "cores": 1, "kind": "port",
                                   "sample": "%d full steps of the same 10^6-particle workload (%.1f s) with the oracle's C restatement, gcc -O2, double" % (args.cpu_steps, dt)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
