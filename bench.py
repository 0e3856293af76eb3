#!/usr/bin/env python3
"""bench.py — headline benchmark of the metadynamics bias step on MI355X.

One "step" = one pass of the hot path over one synthetic particle snapshot (BASELINE.json configs[1]):
    lamellar CV values of both CVs (one fused pass over the positions, incl. reduction)
    -> bias-grid update on the device (histogram, well-tempered Gaussian deposit, reweighted
       estimator, accumulate, V(s), w(s), dV/ds_c)           [stride 1: every step deposits]
    -> bias forces of both CVs written for every particle (one fused pass).
Inputs are resident in HBM before the timed region; nothing synchronises with the host inside it.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--walkers]

N > 1: one rank per GPU.  Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...`: WORLD_SIZE is set) or this script does: called plainly with --gpus N it starts that command ITSELF as a child
process — before it has imported torch or touched the GPU in any way — and relays rank 0's single JSON line and the exit code.
  weak (default)  10^6 particles per GPU (config 4 of BASELINE.json at N = 8: 8 x 10^6, L = 200), particles sharded, the
                  per-CV sums of a step through the xGMI mailbox (mtd_comm_*: direct stores between the GPUs, still two
                  launches per step; RCCL all-reduce if the mailbox cannot be set up — config.exchange says which one ran),
                  bias grid replicated (every rank deposits the same hill, no grid collective)
  strong          8 x 10^6 particles in all (L = 200), N_global / N per GPU, same exchange
  --walkers       one independent 10^6-particle box per GPU sharing ONE bias grid: the packed increments
                  {grid_delta, sigma_grid_delta | hist_delta, hist_gauss_delta} are summed over the walkers on every
                  deposit (IntegratorMetaDynamics.cc:393-409) with RCCL (mtd_metad_update_bias_walkers)
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

# The xGMI mailbox shares device buffers between the ranks of a node through HIP IPC handles.  On hosts whose driver only
# supports dmabuf IPC the runtime has to be told so BEFORE it initialises (it reads the variable once); without it
# hipIpcGetMemHandle fails with "invalid argument", the mailbox cannot map its peers and every rank takes the RCCL all-reduce
# (~11 us per step more).  Set here, before anything loads HIP, unless the caller chose a value.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
N_PER_GPU = 1_000_000
N_STRONG = 8_000_000      # 8 x 10^6 particles, L = 200
BOX_L = 100.0
GRID = dict(sigma=[1e-3, 1e-3], cv_min=[-0.02, -0.02], cv_max=[0.02, 0.02], num_points=[256, 256])
W, DELTA_T, T = 1.0, 7.0, 1.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: fp64 vector (non-matrix) peak
PROFILE_ROUND = "r4"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: 10^6 particles per GPU; strong: 8 x 10^6 particles in all (SURVEY.md 8d config 4)")
    ap.add_argument("--walkers", action="store_true", help="one independent 10^6-particle box per GPU sharing one bias grid")
    ap.add_argument("--stride", type=int, default=1, help="hill deposition stride (headline: 1)")
    ap.add_argument("--particles", type=int, default=None, help="particles per GPU (weak / walkers) or in all (strong)")
    ap.add_argument("--fast-trig", type=int, default=int(os.environ.get("MTD_FAST_TRIG", "1")),
                    help="1 (library default): hardware sine / cosine on the phase in turns; 0: ocml sinpi / cospi")
    ap.add_argument("--path", choices=["fused", "generic"], default="fused",
                    help="fused: two launches per step (headline); generic: separate C-ABI calls per stage")
    ap.add_argument("--driver", choices=["host", "abi"], default=None,
                    help="host: metadynamics.integrate API, C++ run loop (default; at N>1 with the xGMI mailbox as its "
                         "communicator); abi: C-ABI calls from Python (N>1 fallback when the all-reduce has to go through RCCL)")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32",
                    help="Scalar of the particle arrays (HOOMD single / double precision build); headline: f32 as in BASELINE.json")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--no-sub-records", action="store_true", help="skip the config 3 / config 5 sub-records (N = 1)")
    ap.add_argument("--no-variants", action="store_true", help="skip extra.steady_state / stride100 / f64 / accurate_trig (N = 1)")
    ap.add_argument("--sub-steps", type=int, default=200, help="steps timed for each sub-record")
    ap.add_argument("--config", type=int, choices=[2, 3, 5], default=2,
                    help="2 (default, the headline): two lamellar CVs; 3: cv.mesh on 128^3 + one lamellar CV; 5: cv.steinhardt on a noisy "
                         "fcc crystal (BASELINE.json configs[2] / configs[4]) — any N through the C++ host classes, System::run")
    ap.add_argument("--mesh", choices=["replicated", "slab"], default="replicated",
                    help="--config 3 at N > 1: every rank keeps the whole mesh and the ranks sum their assignments (default), or the mesh "
                         "itself is decomposed into slabs over the ranks")
    return ap.parse_args(argv)


def visible_gpus():
    """GPUs this process could use, WITHOUT touching HIP: the KFD topology (nodes with SIMDs), capped by a visibility list"""
    n = None
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for node in os.listdir(base):
            props = dict(l.split() for l in open(os.path.join(base, node, "properties")) if len(l.split()) == 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except Exception:
        n = None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            k = len([x for x in v.split(",") if x.strip() != ""])
            n = k if n is None else min(n, k)
    return n


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU with torch.distributed.run as a CHILD process
    (this process never imports torch and never touches the GPU), relay rank 0's JSON line and the exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    n_vis = visible_gpus()
    if n_vis is not None and n_vis < args.gpus and env.get("MTD_BENCH_REHEARSAL") is None:
        # fewer GPUs than ranks (a one-GPU box): every rank on cuda:0, control plane gloo — a rehearsal of the code path whose
        # numbers mean nothing; the line says so (config.rehearsal).  A GPU box admits few processes on its card at once.
        if args.gpus > 4:
            sys.stderr.write("bench.py: %d ranks asked for, %d GPU(s) visible: a rehearsal on one GPU is limited to 4 ranks\n" % (args.gpus, n_vis))
            return 2
        env["MTD_BENCH_REHEARSAL"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    if os.environ.get("MTD_BENCH_DRY_LAUNCH") == "1":         # test hook (tests/test_bench_launcher.py): show the launch, start nothing
        print(json.dumps({"cmd": cmd, "rehearsal": env.get("MTD_BENCH_REHEARSAL"), "ipc": env.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                          "torch_imported": "torch" in sys.modules, "visible_gpus": n_vis}))
        return 0
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    line = None
    for l in p.stdout:
        if l.startswith('{"metric"'):
            line = l.strip()
        else:
            sys.stderr.write(l)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 else (0 if line is not None else 1)


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _args = parse()
    if _args.gpus > 1:
        sys.exit(self_launch(_args, sys.argv[1:]))

for p in (os.path.join(ROOT, "metadynamics-plugin_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

import util
from metadynamics import _abi


CVS = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]


class Engine:
    """The hot path through the C-ABI, device resident (metadynamics.sharded.HipLamellarBackend)."""

    def __init__(self, n_local, n_global, rank, seed, stride, fast_trig, dist=None, path="fused", dtype=np.float32, shard=True,
                 box_particles=None):
        from metadynamics.sharded import HipLamellarBackend, ShardedBiasStep
        self.dist = dist
        n_box = n_global if box_particles is None else box_particles
        L = BOX_L * (n_box / N_PER_GPU) ** (1.0 / 3.0)  # constant density (config 4: L = 200 at 8e6)
        pos, types = util.snapshot_random(n_box, L, seed=seed, dtype=np.float32)
        self.full = None
        if dist is not None and shard:
            # every rank draws the same global snapshot and keeps its contiguous slice (lamellar CVs need no locality)
            if rank == 0:
                self.full = (pos, types)           # rank 0 checks the global CV values against the oracle afterwards
            sl = slice(rank * n_local, (rank + 1) * n_local)
            pos, types = pos[sl].copy(), types[sl].copy()
        self.pos_np, self.types_np, self.L = pos, types, L
        d_pos = torch.from_numpy(util.pack_postype(pos.astype(dtype), types, dtype)).cuda()
        self.be = HipLamellarBackend(CVS, d_pos, n_global, L, GRID, W, DELTA_T, T, stride, "well_tempered",
                                     fast_trig=bool(fast_trig), fused=(path == "fused"),
                                     exchange="partials")   # equal shards: the block partial sums travel (no reduce launch)
        self.sharded = ShardedBiasStep(self.be, dist) if (dist is not None and shard) else None
        self.exchange = None
        self.exchange_note = None
        if dist is not None and shard:
            # the n_cv sums of a step travel through the xGMI mailbox (direct stores between the GPUs, no collective call);
            # RCCL all-reduce of the block partial sums if the mailbox cannot be set up on this node
            from metadynamics import xgmi
            box = xgmi.connect(dist, max_doubles=8)
            if box is not None:
                self.be.attach_mailbox(box)
            fallback = "rccl" if dist.get_backend() == "nccl" else dist.get_backend()
            self.exchange = "xgmi-mailbox" if box is not None else fallback
            if box is None:
                self.exchange_note = ("FALLBACK: the xGMI mailbox could not be set up (%s); the per-step sums go through the %s "
                                      "all-reduce (~11 us per step more)" % (xgmi.last_failure() or "no reason recorded", fallback))
        self.t = 0
        self.ev = None

    def step(self):
        be = self.be
        if self.ev is None:
            if self.sharded is not None:
                self.sharded.step(self.t)      # launch A, (mailbox | reduce + all-reduce of n_cv doubles), launch B
            else:
                be.step_single(self.t)         # launch A, launch B
        else:
            # same step with the dominant kernel bracketed by events on the launch stream
            if self.sharded is not None and be.mailbox is None:
                sums = be.cv_pass()
                self.dist.all_reduce(sums)
            else:
                be.cv_partials()
                sums = None
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if be.fused:
                a.record()
                be.force_pass(sums, self.t)
            else:
                # generic path: the grid kernels run first, the events bracket the force kernel only
                be._set_sources(be.scratch.data_ptr(), be.n_part.value) if sums is None else be._set_sources(sums.data_ptr(), 1)
                _abi.check(be.lib.mtd_metad_update_bias(be.h, self.t, None))
                a.record()
                _abi.check(be.lib.mtd_lamellar_forces(C.byref(be.lset), be.N, be.d_pos.data_ptr(), be.fptr, be.dt,
                                                      be.N_global, be.d_bias, C.byref(be.box), None))
            b.record()
            self.ev.append((a, b))
        self.t += 1

    def state(self):
        return self.be.state()

    def close(self):
        self.be.close()


class HostEngine:
    """The same workload through the reference-shaped API: metadynamics.cv / metadynamics.integrate over the C++
    host classes; the step loop is System::run in C++ (what HOOMD's run loop does)."""

    def __init__(self, pos, types, L, n_global, stride, fast_trig, path, dtype=np.float32, mailbox=None, walkers=None):
        """pos / types: this rank's particles (the whole snapshot at N = 1); mailbox: metadynamics.xgmi.Mailbox of a
        particle-sharded run — it plays the role of HOOMD's MPI communicator in the execution configuration; walkers: an
        mtd_rccl handle (the reference's partition communicator) of a multiple-walker run"""
        from metadynamics import context, cv, integrate
        self.pos_np, self.types_np, self.L = pos, types, L
        self.ctx = context.initialize(pos, types, ["A", "B"], L, dtype=dtype, n_global=n_global)
        if mailbox is not None:
            context.exec_conf.setMailbox(mailbox.handle.value)
        if walkers is not None:
            context.exec_conf.setWalkerCommunicator(walkers)
        self.meta = integrate.mode_metadynamics(dt=0.005, stride=stride, mode="well_tempered", W=W, deltaT=DELTA_T, T=T)
        if walkers is not None:
            self.meta.set_params(multiple_walkers=True)
        self.cvs = []
        for i, vecs in enumerate((util.CV1_VECTORS, util.CV2_VECTORS)):
            c = cv.lamellar(sigma=GRID["sigma"][i], mode=dict(A=1.0, B=-1.0), lattice_vectors=vecs, name="cv%d" % i)
            c.set_grid(GRID["cv_min"][i], GRID["cv_max"][i], GRID["num_points"][i])
            self.cvs.append(c)
        self.meta.cpp_integrator.setFusedPath(path == "fused")
        _abi.check(_abi.load().mtd_lamellar_set_fast_trig(int(fast_trig)))
        self.context = context
        self.meta.update_forces()
        self.ctx.system.run(0)          # prepRun: allocate the grid, first deposit (Q17)

    def run(self, k):
        self.ctx.system.run(k)          # note: every run() starts with prepRun (one extra bias update, as in HOOMD)

    def state(self):
        t = self.ctx.system.getCurrentTimeStep()
        integ = self.meta.cpp_integrator
        # the values the bias was evaluated at (global sums of a sharded run), as the grid engine holds them
        return dict(cv=list(integ.getCurrentValues()), V=integ.getLogValue("bias", t),
                    w=integ.getLogValue("weight", t), num_gaussians=integ.getNumGaussians(), fused=integ.usedFusedPath())

    def grid_checksum(self):
        """sum and sum of squares of the bias grid (walkers: identical on every rank after each exchange)"""
        lib = _abi.load()
        h = C.c_void_p(self.meta.cpp_integrator.getEngineHandle())
        G = lib.mtd_metad_num_elements(h)
        out = np.zeros(G, dtype=np.float64)
        _abi.check(lib.mtd_metad_get_array(h, 0, out.ctypes.data, None))
        return float(out.sum()), float((out * out).sum())


class WalkerEngine:
    """Multiple walkers through the C ABI from Python (rehearsals whose control plane is gloo — RCCL refuses two ranks on one
    device; a real run takes the C++ host classes with an RCCL communicator): metadynamics.sharded.WalkerBiasStep over a
    generic CV set, the packed increments summed by the process group."""

    def __init__(self, pos, types, L, stride, fast_trig, dist, dtype=np.float32):
        from metadynamics.sharded import HipCvSetBackend, LamellarPart, WalkerBiasStep
        N = pos.shape[0]
        self.d_pos = torch.from_numpy(util.pack_postype(pos.astype(dtype), types, dtype)).cuda()
        _abi.check(_abi.load().mtd_lamellar_set_fast_trig(int(fast_trig)))
        parts = [LamellarPart(v, m, self.d_pos, N, L) for v, m in CVS]
        self.be = HipCvSetBackend(parts, GRID, W, DELTA_T, T, stride, "well_tempered")
        self.walk = WalkerBiasStep(self.be, dist)
        self.t = 0

    def run(self, k):
        for _ in range(k):
            self.walk.step(self.t)
            self.t += 1

    def state(self):
        return self.be.state()

    def grid_checksum(self):
        g = self.be.grid_array(0)
        return float(g.sum()), float((g * g).sum())


def cpu_baseline(pos, types, L, steps):
    """The CPU restatement (oracle, kind "port") of the same step on ONE host core: the reference CPU
    path is serial (MPI ranks only) and cannot be built here (needs HOOMD)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mtd_ref
    rbox = mtd_ref.Box.make(L)
    opt = util.oracle_postype(pos, types)
    r = mtd_ref.Metad(W=W, T_shift=DELTA_T, T=T, stride=1, mode="well_tempered", **GRID)
    t0 = time.perf_counter()
    for t in range(steps):
        s = [mtd_ref.lamellar_cv(v, opt, m, rbox) for v, m in CVS]
        b = r.update_bias(t, s)
        for c, (v, m) in enumerate(CVS):
            mtd_ref.lamellar_forces(v, opt, m, rbox, b[c])
    dt = time.perf_counter() - t0
    return pos.shape[0] * len(CVS) * steps / dt, dt


def _timed_host_steps(context, steps, repeats=3):
    """seconds per step: the median of `repeats` runs of `steps` steps each (a shared box shows single runs 1.6 x off)"""
    out = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # run(k) = prepRun + k updates.  prepRun re-evaluates at the timestep of the last update, where cv.mesh and cv.steinhardt are
        # cached (OrderParameterMesh.cc:927-928, SteinhardtQl.cc:64-65): one small launch, not a step — k steps are counted, no more
        context.current.system.run(steps)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps)
    _timed_host_steps.last = out
    return float(np.median(out))


def _traffic_record(kernels, family):
    """(per-kernel counter records of the PMC passes committed under profiles/, stale) — the records only while the kernels' sources
    (`family`: "mesh" / "ql") are the ones the counters were collected on: otherwise (None, True); (None, None) when no file exists"""
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc_mesh_ql_summary.json")
    if not os.path.exists(path):
        return None, None
    rec = json.load(open(path))
    if (rec.get("kernel_source_sha256") or {}).get(family) != kernel_source_sha(family):
        return None, True
    out = {k: rec[k] for k in kernels if k in rec}
    out["_source"] = "profiles/%s/pmc_mesh_ql_summary.json" % PROFILE_ROUND
    return out, False


def sub_record_config3(steps, fast_trig):
    """BASELINE.json configs[2] (SURVEY config 3): 10^6 particles, cv.mesh on 128^3 (bug-compatible) + one lamellar CV, 256^2 bias
    grid, well-tempered, stride 1 — through the reference-shaped API (C++ host classes).  The mesh CV's grid spans value x [0, 2]
    (one untimed evaluation supplies the value), sigma = 1 % of the range.  Bound: HBM.  Algorithmic bytes:
    SURVEY 8d's definition for the reference's fp32 / C2C layout (191 MB) and this build's own count (fp64 meshes, half
    spectrum: DESIGN.md 4.4).  The mesh kernels are also timed on their own through the C ABI (events on the launch stream)."""
    from metadynamics import context, cv, integrate
    N, L = N_PER_GPU, BOX_L
    pos, types = util.snapshot_random(N, L, seed=12345, modulated=True, dtype=np.float32)
    pos = (np.mod(pos.astype(np.float64) + L / 2, L) - L / 2).astype(np.float32)       # an MD engine keeps its particles in the box
    pos[pos >= L / 2] = -L / 2
    _abi.check(_abi.load().mtd_lamellar_set_fast_trig(int(fast_trig)))

    def build(lo, hi, sigma):
        context.initialize(pos, types, ["A", "B"], L, dtype=np.float32)
        meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=W, deltaT=DELTA_T, T=T)
        lam = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=util.CV1_VECTORS)
        lam.set_grid(-1.0, 1.0, 256)
        mesh = cv.mesh(nx=128, mode={"A": 1.0, "B": -1.0}, sigma=sigma)
        mesh.set_grid(lo, hi, 256)
        return meta, mesh

    meta, mesh = build(0.0, 1.0, 1.0)
    context.run(1)                              # one untimed evaluation: the value the grid is centred on
    s0 = mesh.cpp_force.getCurrentValue(context.current.system.getCurrentTimeStep())
    context.current = None
    lo, hi = (0.0, 2.0 * s0) if s0 > 0 else (2.0 * s0, 0.0)
    meta, mesh = build(lo, hi, 0.01 * (hi - lo))
    context.run(150)                            # (the GPU idled during the CPU baseline: tens of milliseconds of work bring the clocks back)
    per_step = _timed_host_steps(context, steps)
    t_now = context.current.system.getCurrentTimeStep()
    s_mesh = mesh.cpp_force.getCurrentValue(t_now)
    integ = meta.cpp_integrator
    hills, bias_f, V_now = integ.getNumGaussians(), list(integ.getBiasFactors()), integ.getLogValue("bias", t_now)
    context.current = None
    # the mesh CV's own launches through the C ABI
    lib = _abi.load()
    h = C.c_void_p()
    mode = (C.c_double * 2)(1.0, -1.0)
    _abi.check(lib.mtd_mesh_create(C.byref(h), 128, 128, 128, mode, 2, N))
    d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
    force = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
    box = _abi.Box.make(L)
    parts, n_parts = C.c_void_p(), C.c_uint()
    bias = torch.ones(1, dtype=torch.float64, device="cuda")

    def cv_call():
        _abi.check(lib.mtd_mesh_compute_cv(h, N, d_pos.data_ptr(), _abi.MTD_F32, C.byref(box), N, C.byref(parts), C.byref(n_parts), None))

    def force_call():
        _abi.check(lib.mtd_mesh_forces(h, N, d_pos.data_ptr(), force.data_ptr(), _abi.MTD_F32, C.byref(box), N, bias.data_ptr(), 0.0, None))

    def timed(fn, n=60):
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / n

    cv_us, f_us = timed(cv_call), timed(force_call)

    def assign_info():
        pl, n_ovf = C.c_int(-1), C.c_uint(0)
        _abi.check(lib.mtd_mesh_assign_info(h, C.byref(pl), C.byref(n_ovf), None))
        return {0: "cells", 1: "counting", 2: "bin"}.get(pl.value, str(pl.value)), int(n_ovf.value)

    pipeline_static, ovf_static = assign_info()
    # the same call on a snapshot that MOVES: two snapshots a random displacement apart (rms 0.1 mesh cells per coordinate, ~16 thermal
    # MD steps at dt = 0.005) alternate, so every step bins on segments planned from other positions (the bin pipeline)
    rng = np.random.default_rng(99)
    pos2 = pos.astype(np.float64) + rng.normal(0.0, 0.1 * L / 128, size=pos.shape)
    pos2 = (np.mod(pos2 + L / 2, L) - L / 2).astype(np.float32)
    pos2[pos2 >= L / 2] = -L / 2
    d_pos2 = torch.from_numpy(util.pack_postype(pos2, types, np.float32)).cuda()
    flip = [0]

    def cv_call_moving():
        flip[0] ^= 1
        p = d_pos2 if flip[0] else d_pos
        _abi.check(lib.mtd_mesh_compute_cv(h, N, p.data_ptr(), _abi.MTD_F32, C.byref(box), N, C.byref(parts), C.byref(n_parts), None))

    cv_moving_us = timed(cv_call_moving)
    pipeline_moving, ovf_moving = assign_info()
    _abi.check(lib.mtd_mesh_destroy(h))
    mesh_traffic, mesh_stale = _traffic_record(["k_tile_bin", "k_tile_scatter", "k_tile_combine_rows", "k_fft_xy_forward", "k_fft_z_spectral",
                                                "k_fft_xy_inverse", "k_tile_forces"], "mesh")
    M, hfrac = 128 ** 3, 72.0 / 128.0
    bytes_survey = 48 * N + 68 * M + (64 - 16) * 1_000_000          # 48 B N + 68 B M, + the lamellar CV sharing the position reads
    bytes_build = 48 * N + 8 * M + (8 + 16 * hfrac) * M + 32 * hfrac * M + 48 * hfrac * M + 32 * hfrac * M + (16 * hfrac + 8) * M + 48 * N
    return {"workload": "1xMI355X: 10^6 particles, OrderParameterMesh CV on 128^3 mesh (bug-compatible) + 1 lamellar CV, 256^2 bias grid, well-tempered",
            "ms_per_step": 1e3 * per_step, "ms_per_step_runs": [1e3 * x for x in _timed_host_steps.last], "value": 2 * N / per_step,
            "unit": "particle-CV-evals/s", "steps": steps, "dtype": "f64 meshes, f32 particles",
            "fast_trig": int(fast_trig), "mesh_cv": s_mesh, "mesh_grid": [lo, hi, 256], "mesh_sigma": 0.01 * (hi - lo),
            "on_grid": bool(lo <= s_mesh < hi), "hills": hills, "bias_factors": bias_f, "V": V_now,
            "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "step_algorithmic_bytes_survey_fp32_c2c": bytes_survey, "step_frac_survey": bytes_survey / per_step / 1e9 / HBM_PEAK_GBS,
                         "step_algorithmic_bytes_this_build_fp64_r2c": int(bytes_build), "step_frac": bytes_build / per_step / 1e9 / HBM_PEAK_GBS,
                         "mesh_compute_cv_us": cv_us, "mesh_forces_us": f_us,
                         "assign": {"pipeline": pipeline_static, "overflow_last_step": ovf_static,
                                    "moving_snapshots": {"mesh_compute_cv_us": cv_moving_us, "pipeline": pipeline_moving, "overflow_last_step": ovf_moving,
                                                         "displacement_rms_cells": 0.1,
                                                         "note": "two snapshots a random displacement apart alternate: every step bins on tile segments planned from the other snapshot"}},
                         "traffic": mesh_traffic, "traffic_stale": mesh_stale,
                         "dominant_kernels": "k_tile_scatter, k_tile_forces, k_fft_z_spectral (per-kernel table: profiles/%s/config3_mesh_kernel_stats.csv)" % PROFILE_ROUND,
                         "timing": "host API: wall clock around System::run, synchronised on both sides; mesh calls: HIP events on the launch stream"}}


def sub_record_config5(steps):
    """BASELINE.json configs[4] (SURVEY config 5): 256 000-particle noisy fcc crystal, cv.steinhardt lmax 6, full neighbour list
    r_cut 1.4, 512-point grid over [0, 2 s] with sigma = 1 % of the range (one untimed evaluation supplies s) —
    through the reference-shaped API.  Bound: fp64 vector ALU (no HBM roofline: ~40 B against ~0.9 kflop per pair); flops per
    pair from DESIGN.md 4.5 (CV pass ~0.5 kflop, contracted force pass ~0.4 kflop); the counter-based VALU utilisation of the
    two kernels is under profiles/."""
    from metadynamics import context, cv, integrate
    pos, L = util.fcc_lattice(40)
    pos = pos + np.random.default_rng(777).normal(0, 0.05, pos.shape)
    N = len(pos)

    def build(hi, sigma):
        context.initialize(pos, np.zeros(N, dtype=np.int32), ["A"], L, dtype=np.float64)
        meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=W, deltaT=DELTA_T, T=T)
        nl = cv.nlist_cell(r_cut=1.4)
        lists = nl.update()
        st = cv.steinhardt(r_cut=1.4, r_on=1.2, lmax=6, Ql_ref=[0, 0, 0, 0, 1, 0, 1], nlist=nl, type="A", sigma=sigma)
        st.set_grid(0.0, hi, 512)
        return meta, st, lists

    meta, st, lists = build(1.0, 1.0)
    context.run(1)
    s0 = st.cpp_force.getCurrentValue(context.current.system.getCurrentTimeStep())
    context.current = None
    hi = 2.0 * s0
    meta, st, lists = build(hi, 0.01 * hi)
    context.run(150)
    per_step = _timed_host_steps(context, steps)
    t_now = context.current.system.getCurrentTimeStep()
    s = st.cpp_force.getCurrentValue(t_now)
    integ = meta.cpp_integrator
    hills, bias_f, V_now = integ.getNumGaussians(), list(integ.getBiasFactors()), integ.getLogValue("bias", t_now)
    context.current = None
    pairs = len(lists[2])
    # flops per step as ISSUED: (2 FMA_F64 + MUL_F64 + ADD_F64) wave instructions x 64 lanes, summed over the launches of a step, from
    # the instruction counters committed under profiles/ (an upper bound: every lane counted as active) — only while the sources are
    # the ones the counters were collected on; without them the line carries round 3's per-entry model and says so
    ql_counters, ql_stale = _traffic_record(["k_ql_accumulate", "k_ql_forces", "k_ql_finalize_chain", "k_ql_forces_half"], "ql")
    flops_c = sum(v.get("fp64_flops_per_launch", 0.0) for k, v in (ql_counters or {}).items() if isinstance(v, dict))
    if flops_c > 0:
        flops, flops_source = flops_c, "counters"
    else:
        flops, flops_source = pairs * (500.0 / 2 + 400.0), "model (no counters for these kernel sources; reads ~1.37 x high against round 3's counters)"
    return {"workload": "1xMI355X: 2.56x10^5 particles (noisy fcc), SteinhardtQl l<=6 CV with full neighbour list (%.1f neighbours), 1D 512-bin bias grid" % (pairs / N),
            "ms_per_step": 1e3 * per_step, "ms_per_step_runs": [1e3 * x for x in _timed_host_steps.last], "value": N / per_step,
            "unit": "particle-CV-evals/s", "steps": steps, "dtype": "f64",
            "pair_entries": pairs, "pair_visits_per_s": 1.5 * pairs / per_step, "steinhardt_cv": s,
            "grid": [0.0, hi, 512], "sigma": 0.01 * hi, "on_grid": bool(0.0 <= s < hi), "hills": hills, "bias_factors": bias_f, "V": V_now,
            "roofline": {"bound": "valu_fp64", "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "achieved": flops / per_step / 1e12,
                         "frac": flops / per_step / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "flops_per_step": flops, "flops_source": flops_source,
                         "counters": ql_counters, "counters_stale": ql_stale,
                         "dominant_kernels": "k_ql_forces, k_ql_accumulate (per-kernel table: profiles/%s/config5_steinhardt_kernel_stats.csv)" % PROFILE_ROUND,
                         "timing": "host API: wall clock around System::run, synchronised on both sides"}}


KERNEL_SOURCES = {"fused": ("fused.hip", "lamellar_device.hpp", "metad_device.hpp", "comm_device.hpp", "mtd_device.hpp"),
                  "mesh": ("mesh.hip", "lamellar_device.hpp", "metad_device.hpp", "comm_device.hpp", "mtd_device.hpp", "exact_div.hpp"),
                  "ql": ("steinhardt.hip", "metad_device.hpp", "mtd_device.hpp")}


def kernel_source_sha(family="fused"):
    """fingerprint of the sources a family of kernels is built from (the PMC figures under profiles/ carry the same)"""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[family]:
        h.update(open(os.path.join(ROOT, "metadynamics-plugin_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def host_cores():
    """CPU cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a 256-thread host can
    hand a container 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def self_check(st, stride):
    """The grid engine of the oracle, driven with the CV values the device reports (the particles do not move, so they are
    the same every step), deposits the same number of hills; V(s) and the reweighting factor w(s) after thousands of
    deposits must agree with the device's.  Checker only: nothing here is timed."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mtd_ref
    r = mtd_ref.Metad(W=W, T_shift=DELTA_T, T=T, stride=1, mode="well_tempered", **GRID)
    n = int(st["num_gaussians"])
    for t in range(n):
        r.update_bias(t, st["cv"])
    v_ref, w_ref = r.curr_bias, r.curr_weight
    return {"hills": n, "V_oracle": v_ref, "w_oracle": w_ref,
            "V_rel_err": abs(st["V"] - v_ref) / max(abs(v_ref), 1e-300), "w_rel_err": abs(st["w"] - w_ref) / max(abs(w_ref), 1e-300)}


def measured_copy_bandwidth():
    """a measured ceiling beside the 8 TB/s of the data sheet: device-to-device copy of 512 MB (read + write counted) and a
    read-only pass (sum) over the same buffer — HIP events, best of several"""
    n = 128 * 1024 * 1024
    src = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)

    def best(fn, nbytes, reps=8):
        out = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            out.append(nbytes / (a.elapsed_time(b) * 1e-3) / 1e9)
        return float(max(out))

    dst.copy_(src)
    copy = best(lambda: dst.copy_(src), 2 * 4 * n)
    read = best(lambda: src.sum(), 4 * n)
    fill = best(lambda: dst.fill_(1.0), 4 * n)
    del src, dst
    return {"copy_512MB_GBs": copy, "read_512MB_GBs": read, "write_512MB_GBs": fill,
            "note": "torch copy_ / sum / fill_ of a 512 MB buffer, best of 8 (HIP events); copy counts bytes read + written"}


def timed_variant(stride, dtype, fast_trig, steps, warmup=300, path="fused"):
    """us per step of the headline workload with one thing changed (same kernels through the C ABI, Python loop: equal to the
    C++ loop at these step counts, profiles/r2/timed_region_fixed_cost.log), synchronised on both sides"""
    np_dtype = np.float32 if dtype == "f32" else np.float64
    eng = Engine(N_PER_GPU, N_PER_GPU, 0, seed=12345, stride=stride, fast_trig=fast_trig, path=path, dtype=np_dtype)
    for _ in range(warmup):
        eng.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.state()
    eng.close()
    scalar4 = 16 if dtype == "f32" else 32
    return {"ms_per_step": 1e3 * dt / steps, "steps": steps, "value": steps / dt * N_PER_GPU * 2, "hills": st["num_gaussians"],
            "step_frac": N_PER_GPU * 4 * scalar4 / (dt / steps) / 1e9 / HBM_PEAK_GBS}


def init_ranks(args):
    """(dist or None, rank, world, rehearsal) — one process per GPU as the driver launches them; a rehearsal puts every rank on cuda:0"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d inside a launch of %d rank(s): call it plainly (it starts its ranks itself) or with "
                         "torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    rehearsal = os.environ.get("MTD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    return dist, rank, world, rehearsal


def main_config(args):
    """--config 3 | 5 at any N: BASELINE.json's mesh and Steinhardt configurations DOMAIN-DECOMPOSED through the reference-shaped
    API — the same metadynamics.cv / integrate script on every rank, System::run in C++, the xGMI mailbox in the role of HOOMD's MPI
    communicator (every CV reduces its own sums inside the host classes: SURVEY.md 8e).
      config 3  particles sharded by index; cv.mesh: replicated mesh, M + 1 doubles summed over the ranks per step (RCCL on real
                GPUs, remote loads through the mailbox's exported buffers in a rehearsal) or --mesh slab; cv.lamellar: one double
      config 5  z slabs of the crystal with ghost layers (full neighbour lists); the (lmax+1)(lmax+2) Q'_lm sums per step
    weak scaling: 10^6 (2.56 x 10^5) particles per GPU, constant density; --scaling strong: the named configuration split."""
    from metadynamics import context, cv, integrate, xgmi
    dist, rank, world, rehearsal = init_ranks(args)
    ctl = "cpu" if (dist is None or dist.get_backend() != "nccl") else "cuda"

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    box = rccl = None
    if dist is not None:
        n_small = 64 if args.config == 3 else 256
        box = xgmi.connect(dist, max_doubles=n_small)
        if box is None:
            raise SystemExit("bench.py --config %d --gpus %d: the xGMI mailbox could not be set up (%s) — the host classes' domain "
                             "decomposition has no other communicator" % (args.config, world, xgmi.last_failure()))
        if dist.get_backend() == "nccl" and args.config == 3 and args.mesh == "replicated":
            from metadynamics.sharded import RcclAllReduce
            rccl = RcclAllReduce(dist)             # the library's own RCCL binding for the M + 1 doubles of the replicated mesh
    _abi.check(_abi.load().mtd_lamellar_set_fast_trig(int(args.fast_trig)))
    strong = args.scaling == "strong"
    info = {}
    if args.config == 3:
        n_base = args.particles if args.particles is not None else N_PER_GPU
        n_global = n_base if strong else n_base * world
        if n_global % world:
            raise SystemExit("--config 3: %d particles do not divide over %d ranks" % (n_global, world))
        n_local = n_global // world
        L = BOX_L * (n_global / N_PER_GPU) ** (1.0 / 3.0)
        nx = 128
        pos, types = util.snapshot_random(n_global, L, seed=12345, modulated=True, dtype=np.float32)
        pos = (np.mod(pos.astype(np.float64) + L / 2, L) - L / 2).astype(np.float32)
        pos[pos >= L / 2] = -L / 2
        sl = slice(rank * n_local, (rank + 1) * n_local)
        n_cv = 2

        def build(lo, hi, sigma):
            context.initialize(pos[sl].copy(), types[sl].copy(), ["A", "B"], L, dtype=np.float32, n_global=n_global)
            if box is not None:
                xgmi.attach(dist, context.exec_conf, box, communicator=rccl.handle if rccl is not None else None)
            meta = integrate.mode_metadynamics(dt=0.005, stride=args.stride, mode="well_tempered", W=W, deltaT=DELTA_T, T=T)
            lam = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=util.CV1_VECTORS)
            lam.set_grid(-1.0, 1.0, 256)
            main_cv = cv.mesh(nx=nx, mode={"A": 1.0, "B": -1.0}, sigma=sigma)
            main_cv.set_grid(lo, hi, 256)
            if dist is not None:
                main_cv.set_decomposition(args.mesh)
            return meta, main_cv

        workload = ("%dxMI355X: %d particles%s, OrderParameterMesh CV on 128^3 mesh (bug-compatible) + 1 lamellar CV, 256^2 bias grid, well-tempered"
                    % (world, n_global, "" if world == 1 else " sharded by index (%s scaling)" % args.scaling))
        M, hfrac = nx ** 3, 72.0 / 128.0
        step_bytes = 48 * n_local + 68 * M + (64 - 16) * n_local                 # SURVEY 8d per GPU: every rank transforms the whole mesh
        info["step_algorithmic_bytes_definition"] = "SURVEY 8d, per GPU: 48 B N_local + 68 B M (fp32 / C2C layout of the reference) + 48 B N_local for the lamellar CV"
    else:
        cells = args.particles if args.particles is not None else 40      # (--particles: fcc cells per box edge here)
        nz_cells = cells if strong else cells * world
        a = np.sqrt(2.0)
        basis = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
        grid = np.stack(np.meshgrid(np.arange(cells), np.arange(cells), np.arange(nz_cells), indexing="ij"), -1).reshape(-1, 3)
        Lbox = np.array([cells * a, cells * a, nz_cells * a])
        pos = (grid[:, None, :] + basis[None, :, :]).reshape(-1, 3) * a - Lbox / 2 + 0.25 * a
        pos = pos + np.random.default_rng(777).normal(0, 0.05, pos.shape)
        pos = np.mod(pos + Lbox / 2, Lbox) - Lbox / 2
        n_global = len(pos)
        r_list = 1.4
        z = pos[:, 2]
        owner = np.minimum((np.mod(z + Lbox[2] / 2, Lbox[2]) / Lbox[2] * world).astype(int), world - 1)
        mine = np.where(owner == rank)[0]
        lo_z, hi_z = -Lbox[2] / 2 + rank * Lbox[2] / world, -Lbox[2] / 2 + (rank + 1) * Lbox[2] / world

        def zdist(u, v):
            d = np.abs(u - v)
            return np.minimum(d, Lbox[2] - d)

        ghosts = np.where((owner != rank) & ((zdist(z, lo_z) <= r_list) | (zdist(z, hi_z) <= r_list)))[0] if world > 1 else np.zeros(0, dtype=int)
        n_local = len(mine)
        L = tuple(float(x) for x in Lbox)
        n_cv = 1
        pair_entries = [0]

        def build(lo, hi, sigma):
            context.initialize(pos[mine], np.zeros(n_local, dtype=np.int32), ["A"], L, dtype=np.float64, n_global=n_global,
                               ghost_positions=pos[ghosts] if len(ghosts) else None, ghost_types=np.zeros(len(ghosts), dtype=np.int32))
            if box is not None:
                xgmi.attach(dist, context.exec_conf, box)
            meta = integrate.mode_metadynamics(dt=0.005, stride=args.stride, mode="well_tempered", W=W, deltaT=DELTA_T, T=T)
            nl = cv.nlist_cell(r_cut=r_list)
            pair_entries[0] = len(nl.update()[2])
            main_cv = cv.steinhardt(r_cut=1.4, r_on=1.2, lmax=6, Ql_ref=[0, 0, 0, 0, 1, 0, 1], nlist=nl, type="A", sigma=sigma)
            main_cv.set_grid(lo, hi, 512)
            return meta, main_cv

        workload = ("%dxMI355X: %d particles (noisy fcc, %s), SteinhardtQl l<=6 CV with full neighbour list, 1D 512-bin bias grid"
                    % (world, n_global, "one box" if world == 1 else "z slabs with ghost layers, %s scaling" % args.scaling))
        step_bytes = None
    # the grid of SURVEY.md 8d: value x [0, 2], sigma 1 % of the range — one untimed evaluation supplies the value (collective)
    barrier()
    meta, main_cv = build(0.0, 1.0, 1.0)
    context.run(1)
    s0 = main_cv.cpp_force.getCurrentValue(context.current.system.getCurrentTimeStep())
    context.current = None
    lo, hi = (0.0, 2.0 * s0) if s0 > 0 else (2.0 * s0, 0.0)
    barrier()
    meta, main_cv = build(lo, hi, 0.01 * (hi - lo))
    exchange_large = context.exec_conf.largeExchangeName()
    context.run(100)
    barrier()
    if args.warmup > 0:
        context.current.system.run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    # run(k) = prepRun + k updates; the main CV of these configurations is cached for the timestep prepRun re-evaluates (one small
    # launch, not a step): exactly `steps` steps are counted
    context.current.system.run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    steady = None
    if not args.no_variants:
        k_steady = max(500, args.steps)
        barrier()
        t1 = time.perf_counter()
        context.current.system.run(k_steady)
        barrier()
        steady = (time.perf_counter() - t1, k_steady)
    t_now = context.current.system.getCurrentTimeStep()
    s_now = main_cv.cpp_force.getCurrentValue(t_now)
    integ = meta.cpp_integrator
    st = dict(cv=list(integ.getCurrentValues()), V=integ.getLogValue("bias", t_now), w=integ.getLogValue("weight", t_now),
              num_gaussians=integ.getNumGaussians(), bias_factors=list(integ.getBiasFactors()), fused=bool(integ.usedFusedPath()))
    timeouts = None
    if dist is not None:
        tt = torch.tensor([elapsed, steady[0] if steady else 0.0], dtype=torch.float64, device=ctl)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0].item())
        if steady:
            steady = (float(tt[1].item()), steady[1])
        # the replicated state must be the same bits on every rank
        mine_bits = torch.from_numpy(np.array(st["cv"] + st["bias_factors"] + [st["V"], st["w"]]).view(np.int64).copy()).to(ctl)
        lo_b, hi_b = mine_bits.clone(), mine_bits.clone()
        dist.all_reduce(lo_b, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_b, op=dist.ReduceOp.MAX)
        st["identical_on_all_ranks"] = bool((lo_b == hi_b).all().item())
        tq = torch.tensor([box.timeouts()], dtype=torch.int64, device=ctl)
        dist.all_reduce(tq, op=dist.ReduceOp.MAX)
        timeouts = int(tq.item())
    if rank == 0:
        per_step = elapsed / args.steps
        out = {"metric": "particle_cv_evals_per_s", "value": n_global * n_cv / per_step, "unit": "particle-CV-evals/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * per_step, "md_bias_steps_per_s": 1.0 / per_step,
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
               "dtype": "f64 meshes, f32 particles" if args.config == 3 else "f64", "data": "synthetic",
               "config": {"workload": workload, "bench_config": args.config, "particles_per_gpu": n_local, "particles_global": n_global,
                          "n_cv": n_cv, "stride": args.stride, "driver": "host", "fast_trig": int(args.fast_trig),
                          "mode": "single" if world == 1 else "sharded", "grid": [lo, hi], "on_grid": bool(lo <= s_now < hi)},
               "state": st}
        if world > 1:
            out["config"]["exchange"] = ("xgmi-mailbox (per-CV sums, mtd_comm_allreduce_small inside the host classes)" +
                                         ("; mesh: %s" % ("slab decomposition over the mailbox's exported buffers" if args.mesh == "slab" else
                                                          "replicated, M + 1 doubles per step through %s" % exchange_large) if args.config == 3 else
                                          "; Q'_lm sums: 56 doubles per step"))
            out["config"]["mailbox_timeouts"] = timeouts
            out["config"]["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
            if args.config == 5:
                out["config"]["ghosts_rank0"] = int(len(ghosts))
        if rehearsal:
            out["config"]["rehearsal"] = "all %d ranks share cuda:0, control plane gloo: the code path is real, the numbers mean nothing" % world
        if args.config == 3:
            out["roofline"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": step_bytes / per_step / 1e9,
                               "frac": step_bytes / per_step / 1e9 / HBM_PEAK_GBS, "traffic": None, "level": "step, per GPU",
                               "step_algorithmic_bytes": step_bytes, "definition": info["step_algorithmic_bytes_definition"]}
        else:
            out["roofline"] = {"bound": "valu_fp64", "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": None, "level": "step, per GPU",
                               "pair_entries_rank0": pair_entries[0], "achieved": None, "frac": None,
                               "note": "flops from the instruction counters: profiles/%s (single-GPU sub-record extra.config5 of the default run)" % PROFILE_ROUND}
        if steady is not None:
            out["extra"] = {"steady_state": {"ms_per_step": 1e3 * steady[0] / steady[1], "steps": steady[1], "value": n_global * n_cv / (steady[0] / steady[1])}}
        if not args.no_cpu_baseline and n_global <= 2_000_000:
            # checker (untimed): the CV the ranks agreed on against the oracle on the WHOLE snapshot
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import mtd_ref
            if args.config == 3:
                rbox, opt = mtd_ref.Box.make(L), util.oracle_postype(pos, types)
                s_ref = mtd_ref.Mesh(nx, nx, nx, [1.0, -1.0]).cv(opt, rbox)
                l_ref = mtd_ref.lamellar_cv(util.CV1_VECTORS, opt, util.MODE_AB, rbox)
                out["cv_check"] = {"cv_oracle": [l_ref, s_ref], "rel_err": [abs(st["cv"][0] - l_ref) / abs(l_ref), abs(st["cv"][1] - s_ref) / abs(s_ref)],
                                   "tolerance": 1e-6}
            elif L[0] == L[2]:
                rbox, pt = mtd_ref.Box.make(L[0]), util.oracle_postype(pos, np.zeros(n_global, dtype=np.int32))
                lists = util.build_nlist(pos, L[0], r_list)
                val = mtd_ref.ql_compute_cv(pt, rbox, *lists, 1.4, 1.2, 6, 0, [0, 0, 0, 0, 1, 0, 1])[0]
                out["cv_check"] = {"cv_oracle": [val], "rel_err": [abs(st["cv"][0] - val) / abs(val)], "tolerance": 1e-6}
        print(json.dumps(out), flush=True)
    context.current = None
    if dist is not None:
        dist.barrier()
        if rccl is not None:
            rccl.close()
        box.close()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.config != 2:
        return main_config(args)
    lib_default_trig = int(_abi.load().mtd_lamellar_get_fast_trig())     # before anything sets the process-wide switch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d inside a launch of %d rank(s): call it plainly (it starts its ranks itself) or with "
                         "torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    dist = None
    # rehearsal of the N>1 code path on a one-GPU box: MTD_BENCH_REHEARSAL=1 puts every rank on cuda:0 and carries the
    # control plane with gloo (RCCL refuses two ranks on one device); numbers from such a run mean nothing
    rehearsal = os.environ.get("MTD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # MTD_BENCH_FORCE_DIST=1: take the N>1 code path (RCCL process group, all-reduce per step) with a single rank — the
    # host-side cost of that path can then be measured on a one-GPU box
    if world > 1 or os.environ.get("MTD_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    walkers = bool(args.walkers)
    if args.scaling == "strong" and walkers:
        raise SystemExit("--walkers is a weak-scaling mode (one box per GPU)")
    if args.scaling == "strong":
        n_global = args.particles if args.particles is not None else N_STRONG
        if n_global % world:
            raise SystemExit("--scaling strong: %d particles do not divide over %d ranks" % (n_global, world))
        n_local = n_global // world
    else:
        n_local = args.particles if args.particles is not None else N_PER_GPU
        n_global = n_local if walkers else n_local * world
    driver = args.driver or "host"
    path = "generic" if walkers else args.path          # the fused step is off in walker mode (the increments are summed between the grid passes)

    def barrier():
        # (plain synchronise: recording an event and polling it first measured ~1 us per step SLOWER at K = 20 —
        # tools/diag/k20_barrier_probe.py, 22.6 against 23.5 us — and the slower state outlasts the event traffic by thousands of steps)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def ctl_device():
        return "cuda" if dist.get_backend() == "nccl" else "cpu"

    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    # the event-bracketed pass over the dominant kernel always goes through the C-ABI backend (same kernels)
    if walkers:
        seed = 12345 + rank                    # every walker its own box
    elif world == 1 and args.scaling == "weak":
        seed = 12345                           # config 2 of SURVEY.md 8d
    else:
        seed = 12346                           # config 4
    eng = Engine(n_local, n_global, rank, seed=seed, stride=args.stride, fast_trig=args.fast_trig, dist=dist, path=path, dtype=np_dtype,
                 shard=not walkers)
    if dist is not None and eng.be.mailbox is not None:
        # the mailbox passed its self test; a short rehearsal of the real step decides whether it carries the timed run: any
        # bounded wait that expired on any rank (a link that does not deliver) sends every rank back to the collective
        # (checked after 1, 5 and 50 steps: a dead link costs one step's bounded waits, not fifty)
        for chunk in (1, 4, 45):
            barrier()
            for _ in range(chunk):
                eng.step()
            barrier()
            tt = torch.tensor([eng.be.mailbox.timeouts()], dtype=torch.int64, device=ctl_device())
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            if int(tt.item()) > 0:
                break
        if int(tt.item()) > 0 or os.environ.get("MTD_BENCH_TEST_FALLBACK") == "1":      # (env: exercises this branch in a rehearsal)
            eng.be.attach_mailbox(None)
            fb = "rccl" if dist.get_backend() == "nccl" else dist.get_backend()
            eng.exchange = fb
            eng.exchange_note = "FALLBACK: the xGMI mailbox gave %d expired waits in the rehearsal steps; the per-step sums go through the %s all-reduce" % (int(tt.item()), fb)
    if driver == "host" and dist is not None and not walkers and eng.be.mailbox is None:
        driver = "abi"                     # no mailbox on this node: the C-ABI backend with the RCCL all-reduce
    if walkers and dist is not None and dist.get_backend() != "nccl":
        driver = "abi"                     # rehearsal: the packed increments through the process group (gloo)
    host = None
    walker_comm = None
    if walkers and dist is not None and driver == "host":
        from metadynamics.sharded import RcclAllReduce
        walker_comm = RcclAllReduce(dist)   # the library's own RCCL binding; the host classes call mtd_metad_update_bias_walkers
        eng.exchange = "rccl (mtd_metad_update_bias_walkers: packed grid increments, 4 G elements per deposit)"
    elif walkers and dist is not None:
        eng.exchange = "%s process group (packed grid increments, 4 G elements per deposit)" % dist.get_backend()
    if driver == "host":
        # every piece of set-up comes first — building the host-API system copies the snapshot again and leaves the GPU idle for
        # ~0.1 s, after which the first milliseconds of work run on lowered clocks (tools/diag/ramp_probe.py: +6-8 % per step after
        # 20-200 ms of idling, and W = 5 warm-up steps are 0.1 ms of work)
        barrier()                          # ranks enter the first exchange (prepRun's deposit) together
        host = HostEngine(eng.pos_np, eng.types_np, eng.L, n_global, args.stride, args.fast_trig, path, dtype=np_dtype,
                          mailbox=eng.be.mailbox, walkers=walker_comm.handle.value if walker_comm is not None else None)
    elif walkers:
        barrier()
        host = WalkerEngine(eng.pos_np, eng.types_np, eng.L, args.stride, args.fast_trig, dist, dtype=np_dtype)
    def run_steps(k):
        if host is not None:
            if isinstance(host, HostEngine):
                host.run(k - 1)             # run(k) = prepRun (one bias update) + k updates: exactly k bias steps
            else:
                host.run(k)
        else:
            for _ in range(k):
                eng.step()

    # A stretch of plain, untimed steps in front of the warm-up: building the host-API system left the GPU idle for ~0.1 s, and after
    # 50 ms of idling a region of 20 steps runs 1.5 us per step slower, the next one still 1 us (tools/diag/k20_shape_probe.py).
    # The W warm-up steps come FIRST and the long untimed stretch directly in front of the timed region (everything before the timed
    # region is untimed warm-up either way): a short run(W) between the stretch and the region left the region of K = 20 steps at
    # 19.5-24.6 us per step (median 20.0), the stretch directly in front of it at 19.4-20.3 (median 19.5) — sixteen runs on one
    # box, profiles/r4/k20_env_probe.log, tools/diag/k20_order_probe.sh.  MTD_BENCH_WARM_ORDER=0: the old order (diagnostic).
    warm_first = os.environ.get("MTD_BENCH_WARM_ORDER", "1") == "1"
    if warm_first and args.warmup > 0:
        run_steps(args.warmup)
        barrier()
    for _ in range(8):
        run_steps(250)
        barrier()
    barrier()                              # ranks enter the first exchange together (the mailbox waits are bounded)
    if args.warmup > 0 and not warm_first:
        run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    st = host.state() if host is not None else eng.state()
    # steady state: the same loop over >= 2000 steps in the same process (a timed region has a fixed cost of ~40 us —
    # doorbell on an idle queue, completion signal — which is 2 us per step at K = 20 and nothing at K = 2000)
    steady = None
    if not args.no_variants:
        k_steady = max(2000, args.steps)
        barrier()
        t1 = time.perf_counter()
        run_steps(k_steady)
        barrier()
        steady = (time.perf_counter() - t1, k_steady)
        st = host.state() if host is not None else eng.state()
    # the driver's call times ONE region of K steps; the same bracket ten more times shows where that one sample sits
    more_regions = None
    if not args.no_variants and dist is None:
        more_regions = []
        for _ in range(10):
            barrier()
            t2 = time.perf_counter()
            run_steps(args.steps)
            barrier()
            more_regions.append(1e3 * (time.perf_counter() - t2) / args.steps)
        st = host.state() if host is not None else eng.state()
    # (measured AFTER the timed regions since round 4: two events around every launch leave the queue in a state in which the
    # following plain steps run ~2.5 us slower for thousands of steps — tools/diag/k20_barrier_probe.py, tools/diag/k20_shape_probe.py: regions
    # of 20 steps take 19.4-19.7 us per step whatever ran before them, 21-22 after event traffic or after 50 ms of idling)
    # dominant kernel (launch B, the force pass): per-launch durations over the same loop from HIP events on the launch stream.
    # Fused path: every launch carries its own start / stop events (hipExtLaunchKernelGGL, armed by mtd_profile_force_begin):
    # the begin and end of that dispatch and nothing else — no subtraction.  Two cross-checks are reported beside it: the
    # differential (n whole steps) - (n launches of launch A alone), and a plain event pair around B minus the cost of an
    # empty pair (which subtracts one packet too many: low).  The rocprofv3 kernel trace of the same command is under
    # profiles/; its average is reported beside the in-process figure (roofline.rocprof_avg_launch_us) — the roofline is priced
    # with the in-process one, the larger of the two.
    for _ in range(20):
        eng.step()
    n_ev = 300          # launches measured one by one (their own loops, outside the timed region: events per launch perturb the step)
    lib = _abi.load()

    def event_pair():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    alt = {}
    if path == "fused":
        timing = "start/stop events of each launch (hipExtLaunchKernelGGL) on the launch stream"
        _abi.check(lib.mtd_profile_force_begin(n_ev))
        for _ in range(n_ev):
            eng.step()
        buf = (C.c_double * n_ev)()
        n_got = C.c_uint()
        _abi.check(lib.mtd_profile_force_end(buf, n_ev, C.byref(n_got)))
        direct = np.array(buf[:n_got.value])
        e0, e1 = event_pair()
        e0.record()
        for _ in range(n_ev):
            eng.step()
        e1.record()
        # launch A alone sends without anybody receiving: no rank may start that while a peer still polls for the last real
        # exchange (the slots alternate between two buffers: a third send would overwrite what the peer is waiting for)
        barrier()
        f0, f1 = event_pair()
        f0.record()
        for _ in range(n_ev):
            eng.be.cv_partials()
        f1.record()
        barrier()
        step_us, a_us = e0.elapsed_time(e1) * 1e3 / n_ev, f0.elapsed_time(f1) * 1e3 / n_ev
        alt["steps_minus_launch_A_alone_us"] = step_us - a_us
        alt["step_us"], alt["launch_A_alone_us"] = step_us, a_us
    else:
        timing = "HIP events around the kernel minus the cost of an empty event pair"
    # an event pair around the kernel, minus an empty pair (the measurement itself on the generic path)
    eng.ev = []
    for _ in range(n_ev):
        eng.step()
    empty = []
    for _ in range(200):
        a, b = event_pair()
        a.record()
        b.record()
        empty.append((a, b))
    torch.cuda.synchronize()
    ev_overhead_us = float(np.median([a.elapsed_time(b) for a, b in empty]) * 1e3)
    pair = np.array([a.elapsed_time(b) for a, b in eng.ev]) * 1e3 - ev_overhead_us
    eng.ev = None
    raw = direct if path == "fused" else pair
    # a sample more than 3x the median is a stall of the queue (clock ramp, host jitter), not a kernel duration: dropped, counted
    keep = raw <= 3.0 * np.median(raw)
    n_stalls = int((~keep).sum())
    force_us_mean, force_us = float(np.mean(raw[keep])), float(np.median(raw))
    if path == "fused":
        alt["event_pair_minus_empty_pair_us"] = float(np.mean(pair[pair <= 3.0 * np.median(pair)]))
    alt["empty_event_pair_us"] = ev_overhead_us

    if dist is not None:
        vals = [elapsed, steady[0] if steady else 0.0]
        tt = torch.tensor(vals, dtype=torch.float64, device=ctl_device())
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0].item())
        if steady:
            steady = (float(tt[1].item()), steady[1])
    mailbox_timeouts = None
    if eng.be.mailbox is not None:
        tt = torch.tensor([eng.be.mailbox.timeouts()], dtype=torch.int64, device=ctl_device())
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        mailbox_timeouts = int(tt.item())
    walker_check = None
    if walkers and host is not None:
        s1, s2 = host.grid_checksum()
        if dist is not None:
            lo = torch.tensor([s1, s2], dtype=torch.float64, device=ctl_device())
            hi = lo.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            same = bool((lo == hi).all().item())
        else:
            same = True
        walker_check = {"grid_sum": s1, "grid_sum_sq": s2, "grid_identical_on_all_ranks": same, "hills_rank0": int(st["num_gaussians"])}

    if rank == 0:
        n_boxes = world if walkers else 1
        steps_per_s = args.steps / elapsed
        value = steps_per_s * n_global * n_boxes * 2
        # dominant kernel's algorithmic bytes per launch: read Scalar4 positions + write one Scalar4
        # force per CV (48 B/particle); the fused kernel also carries the first grid pass of the deposit
        # (per cell: 8 B dV written, 8+8 B reweighted r/w, 4 B hist_delta read)
        scalar4 = 16 if args.dtype == "f32" else 32
        force_bytes = n_local * (scalar4 + 2 * scalar4)
        if path == "fused" and args.stride == 1:
            force_bytes += 256 * 256 * 28
        achieved = force_bytes / (force_us_mean * 1e-6) / 1e9
        # HBM traffic of the dominant kernel from the PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE
        # in separate rocprofv3 runs, gfx950 x2 correction on FETCH_SIZE); only valid for the default workload
        # (a profiler cannot be attached from inside the run; the figure is only reported while the kernel's sources are the
        # ones the counters were collected on — otherwise traffic is null and traffic_stale says so)
        traffic, traffic_src, traffic_stale, rocprof_us = None, None, None, None
        pmc = os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc_summary.json")
        if os.path.exists(pmc) and path == "fused" and args.stride == 1 and n_local == N_PER_GPU and args.dtype == "f32" and world == 1:
            rec = json.load(open(pmc))
            traffic_stale = rec.get("kernel_source_sha256") != kernel_source_sha()
            if not traffic_stale:
                traffic = rec.get("k_fused_force", {}).get("hbm_bytes_per_launch")
                rocprof_us = rec.get("k_fused_force", {}).get("rocprof_avg_launch_us")
                traffic_src = "profiles/%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on these kernel sources)" % PROFILE_ROUND
        # step level: N (P [CV pass read] + P [force pass read] + n_cv P [force writes]) over the measured step
        step_bytes = n_local * 4 * scalar4
        step_frac = step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS
        if world == 1 and args.scaling == "weak":
            workload = "1xMI355X: 10^6 particles, 2 lamellar CVs (8 Fourier modes each), 256^2 bias grid, well-tempered"
        elif walkers:
            workload = "%dxMI355X: %d walkers of %d particles each sharing one 256^2 bias grid, 2 lamellar CVs, packed increments summed over the walkers (%s)" % (world, world, n_local, eng.exchange)
        else:
            workload = "%dxMI355X: %d particles sharded (%s scaling), 2 lamellar CVs, all-reduce of the CV sums (%s), replicated 256^2 grid" % (world, n_global, args.scaling, eng.exchange)
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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": workload,
                       "particles_per_gpu": n_local, "particles_global": n_global * n_boxes, "n_cv": 2, "modes_per_cv": 8, "grid": "256x256",
                       "stride": args.stride, "fast_trig": int(args.fast_trig), "fast_trig_library_default": lib_default_trig,
                       "path": path, "driver": driver, "mode": "walkers" if walkers else "sharded" if world > 1 else "single"},
            "roofline": {"bound": "hbm", "kernel": "k_fused_force" if path == "fused" else "k_lamellar_forces", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "frac_definition": "algorithmic bytes per launch / avg_launch_us (in-process start/stop events of every launch) / 8 TB/s; "
                                            "rocprof_avg_launch_us is the kernel-trace average of the same command under profiles/ (smaller: "
                                            "the trace excludes the dispatch's own start-up), frac_rocprof prices the roofline with it",
                         "rocprof_avg_launch_us": rocprof_us,
                         "frac_rocprof": (force_bytes / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if rocprof_us else None,
                         "step_frac": step_frac, "step_algorithmic_bytes": step_bytes,
                         "step_frac_definition": "N*(P + P + n_cv*P) bytes per step (SURVEY.md 8d: 64 B/particle in f32) / ms_per_step / 8 TB/s, per GPU",
                         "algorithmic_bytes_per_launch": force_bytes, "avg_launch_us": force_us_mean, "median_launch_us": force_us,
                         "timing": timing, "launches_timed": int(keep.sum()), "cross_checks": alt,
                         "stalled_samples_dropped": n_stalls},
            "state": st,
        }
        if rehearsal:
            out["config"]["rehearsal"] = "all %d ranks share cuda:0, control plane gloo: the code path is real, the numbers mean nothing" % world
        if eng.exchange is not None:
            out["config"]["exchange"] = eng.exchange
            if eng.exchange_note:
                out["config"]["exchange_note"] = eng.exchange_note
            out["config"]["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
            if mailbox_timeouts is not None:
                out["config"]["mailbox_timeouts"] = mailbox_timeouts
        if walker_check is not None:
            out["walker_check"] = walker_check
        extra = {}
        if steady is not None:
            sdt, sk = steady
            extra["steady_state"] = {"ms_per_step": 1e3 * sdt / sk, "steps": sk, "value": sk / sdt * n_global * n_boxes * 2,
                                     "step_frac": step_bytes / (sdt / sk) / 1e9 / HBM_PEAK_GBS,
                                     "note": "the same loop in the same process right after the timed region, synchronised on both sides: the "
                                             "fixed cost of a timed region (~40 us: doorbell on an idle queue, completion signal) is 2 us per "
                                             "step at K = 20 and < 0.03 us here"}
            out["roofline"]["step_frac_steady_state"] = extra["steady_state"]["step_frac"]
        if more_regions:
            extra["timed_region_repeats"] = {"ms_per_step": more_regions, "steps": args.steps,
                                             "note": "ten more regions of the same K steps, same barrier + synchronise bracket, right after the steady-state run"}
        if not args.no_cpu_baseline and st.get("num_gaussians") and args.stride == 1 and not walkers:      # any N: the grid is replicated
            out["self_check"] = self_check(st, args.stride)
        if not args.no_cpu_baseline and eng.full is not None:
            # sharded run: the CV values the ranks agreed on against the oracle on the whole snapshot (checker, untimed).
            # On a random snapshot the sums cancel to O(sqrt(N)): the tolerance of the parity tests is 1e-6 n_wave / sqrt(N)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import mtd_ref
            fpos, ftypes = eng.full
            opt = util.oracle_postype(fpos, ftypes)
            rbox = mtd_ref.Box.make(eng.L)
            ref_cv = [mtd_ref.lamellar_cv(v, opt, util.MODE_AB, rbox) for v in (util.CV1_VECTORS, util.CV2_VECTORS)]
            out["cv_check"] = {"cv_oracle": ref_cv, "abs_err": [abs(a - b) for a, b in zip(st["cv"], ref_cv)],
                               "tolerance": 1e-6 * 8 / np.sqrt(n_global)}
        single = world == 1 and args.scaling == "weak" and not walkers and n_local == N_PER_GPU
        if single and not args.no_variants:
            # the same workload with one thing changed each (SURVEY.md 8d "Metric": stride 100, the fp64 build, and the trigonometry
            # mode that is NOT the one timed above), and a measured bandwidth ceiling beside the data-sheet peak
            try:
                if host is not None and isinstance(host, HostEngine):
                    host.context.current = None
                extra["stride100"] = timed_variant(100, "f32", args.fast_trig, 2000)
                extra["f64"] = timed_variant(1, "f64", args.fast_trig, 1000)
                other = 0 if args.fast_trig else 1
                extra["accurate_trig" if other == 0 else "fast_trig"] = timed_variant(1, "f32", other, 2000)
                _abi.check(lib.mtd_lamellar_set_fast_trig(int(args.fast_trig)))
                mp = measured_copy_bandwidth()
                out["roofline"]["measured_peak"] = mp
                out["roofline"]["frac_of_measured_copy"] = achieved / mp["copy_512MB_GBs"]
                ss = extra.get("steady_state", {}).get("ms_per_step", 1e3 * elapsed / args.steps)
                out["roofline"]["step_frac_of_measured_copy"] = step_bytes / (ss * 1e-3) / 1e9 / mp["copy_512MB_GBs"]
            except Exception as e:                                      # never let a variant break the headline line
                extra["variants_error"] = repr(e)
        if not args.no_cpu_baseline and world == 1 and not walkers:
            pos, types = eng.pos_np, eng.types_np
            n_cpu = min(pos.shape[0], N_PER_GPU)                        # bounded sample (strong scaling at N = 1 holds 8 x 10^6)
            v, dt = cpu_baseline(pos[:n_cpu], types[:n_cpu], eng.L, args.cpu_steps)
            out["cpu_baseline"] = {"value": v, "unit": "particle-CV-evals/s", "cores": 1, "kind": "port",
                                   "sample": "%d full steps of the same workload on %d particles (%.1f s) with the oracle's C restatement, gcc -O2, double" % (args.cpu_steps, n_cpu, dt)}
            # informational: the same loops with OpenMP over the particles on every host core ("idealised multi-rank";
            # the reference itself has no threading, one MPI rank per core is its only parallelism)
            try:
                import mtd_ref
                n_thr = host_cores()
                os.environ["OMP_NUM_THREADS"] = str(n_thr)              # read by libgomp when the OpenMP build is loaded
                mtd_ref.use_openmp(True)
                cpu_baseline(pos[:n_cpu], types[:n_cpu], eng.L, 1)      # thread pool start-up
                v2, dt2 = cpu_baseline(pos[:n_cpu], types[:n_cpu], eng.L, 4 * args.cpu_steps)
                out["cpu_baseline_all_cores"] = {"value": v2, "unit": "particle-CV-evals/s", "cores": n_thr, "kind": "port",
                                                 "sample": "%d steps (%.1f s), OpenMP over particles" % (4 * args.cpu_steps, dt2)}
            except Exception as e:                                      # never let the informational leg break the bench line
                out["cpu_baseline_all_cores"] = {"error": str(e)}
            finally:
                mtd_ref.use_openmp(False)
        if single and not args.no_sub_records:
            # the other single-GPU configurations of BASELINE.json, bounded runs (not the headline: reported beside it)
            try:
                if host is not None and isinstance(host, HostEngine):
                    host.context.current = None
                extra["config3"] = sub_record_config3(args.sub_steps, args.fast_trig)
                extra["config5"] = sub_record_config5(args.sub_steps)
            except Exception as e:                                      # never let a sub-record break the headline line
                extra["error"] = repr(e)
        if extra:
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
