#!/usr/bin/env python3
"""Developer tool (GPU box): the bias step as a HIP GRAPH against plain stream launches.

Every configuration runs through the reference-shaped API (metadynamics.cv / integrate, C++ run loop System::run): mode "graph" switches
on System::run's own graph replay (whole periods captured after a settling period, ~24 steps per graph), "stream" runs plain launches
on a stream of its own, "null" on the null stream.  Timed regions are synchronised on both sides (wall clock), as the driver's are.
Fresh process per (config, mode), rounds interleaved.

usage: exp_graph.py [rounds] [config ...]      config in {2, 3, 5}; worker: exp_graph.py --worker <config> <mode>
"""
import ctypes as C
import os
import subprocess
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(config):
    sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
    import numpy as np
    import util
    from metadynamics import context, cv, integrate
    if config == 2:
        N, L = 1_000_000, 100.0
        pos, types = util.snapshot_random(N, L, seed=12345, dtype=np.float32)
        context.initialize(pos, types, ["A", "B"], L, dtype=np.float32)
        meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
        for i, vecs in enumerate((util.CV1_VECTORS, util.CV2_VECTORS)):
            c = cv.lamellar(sigma=1e-3, mode=dict(A=1.0, B=-1.0), lattice_vectors=vecs, name="cv%d" % i)
            c.set_grid(-0.02, 0.02, 256)
    elif config == 3:
        N, L = 1_000_000, 100.0
        pos, types = util.snapshot_random(N, L, seed=12345, modulated=True, dtype=np.float32)
        pos = (np.mod(pos.astype(np.float64) + L / 2, L) - L / 2).astype(np.float32)
        pos[pos >= L / 2] = -L / 2
        context.initialize(pos, types, ["A", "B"], L, dtype=np.float32)
        meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
        lam = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=util.CV1_VECTORS)
        lam.set_grid(-1.0, 1.0, 256)
        mesh = cv.mesh(nx=128, mode={"A": 1.0, "B": -1.0}, sigma=7.6e-6)
        mesh.set_grid(0.0, 7.647e-4, 256)
    else:
        pos, L = util.fcc_lattice(40)
        pos = pos + np.random.default_rng(777).normal(0, 0.05, pos.shape)
        N = len(pos)
        context.initialize(pos, np.zeros(N, dtype=np.int32), ["A"], L, dtype=np.float64)
        meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
        nl = cv.nlist_cell(r_cut=1.4)
        nl.update()
        st = cv.steinhardt(r_cut=1.4, r_on=1.2, lmax=6, Ql_ref=[0, 0, 0, 0, 1, 0, 1], nlist=nl, type="A", sigma=0.8)
        st.set_grid(0.0, 79.8, 512)
    return context, meta


def worker(config, mode):
    import numpy as np
    import torch
    hip = C.CDLL("libamdhip64.so")
    context, meta = build(config)

    def chk(rc, what):
        if rc != 0:
            raise RuntimeError("%s -> hipError %d" % (what, rc))

    stream = C.c_void_p()
    if mode != "null":                  # "null": everything on the NULL stream, device-wide synchronisation (what bench.py did up to round 3)
        chk(hip.hipStreamCreateWithFlags(C.byref(stream), 1), "hipStreamCreateWithFlags")      # non-blocking
        torch.cuda.synchronize()
        context.exec_conf.setStream(stream.value)
    sysm = context.current.system

    def sync():
        if mode == "null":
            torch.cuda.synchronize()
        else:
            chk(hip.hipStreamSynchronize(stream), "hipStreamSynchronize")

    context.run(50)                     # registers the CVs, allocates, first deposits, mesh: the bin pipeline's plan
    sync()
    out = {}
    # System::run(K) = prepRun + K updates.  prepRun re-evaluates at the timestep of the last update: a FULL step for cv.lamellar (it
    # always recomputes, LamellarOrderParameter.h:75-79), one small launch for cv.mesh / cv.steinhardt (cached per timestep) — counted
    # as K + 1 and K steps.  (The first version of this tool captured run(19) and counted 20 steps: for configs 3 and 5 that graph
    # held 19 real steps, which read as a 5 % gain of the graph — profiles/r4/graph_ab.log.)
    sysm.setGraphMode(1 if mode == "graph" else 0)
    for K, reps in ((24, 40), (2000, 5)):
        def go():
            sysm.run(K)
        steps = K + (1 if config == 2 else 0)
        for _ in range(3):
            go()
        sync()
        ts = []
        for _ in range(reps):
            sync()
            t0 = time.perf_counter()
            go()
            sync()
            ts.append(1e6 * (time.perf_counter() - t0) / steps)
        out["K%d_us_per_step_median" % K] = float(np.median(ts))
        out["K%d_us_per_step_min" % K] = float(np.min(ts))
        out["K%d_graph_steps" % K] = sysm.lastRunGraphSteps()
    integ = meta.cpp_integrator
    out["hills"] = integ.getNumGaussians()
    out["cv"] = list(integ.getCurrentValues())
    print("RESULT " + " ".join("%s=%s" % (k, ("%.2f" % v) if isinstance(v, float) else v) for k, v in out.items()), flush=True)


if __name__ == "__main__":
    if "--worker" in sys.argv:
        i = sys.argv.index("--worker")
        worker(int(sys.argv[i + 1]), sys.argv[i + 2])
    else:
        rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        configs = [int(x) for x in sys.argv[2:]] or [2, 5, 3]
        for r in range(rounds):
            for cfg in configs:
                for mode in os.environ.get("EXP_MODES", "stream,graph").split(","):
                    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(cfg), mode], capture_output=True, text=True)
                    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
                    print("config=%d mode=%-6s %s" % (cfg, mode, line[0][7:] if line else "FAILED " + (p.stderr.strip().splitlines() or ["?"])[-1][-300:]), flush=True)
