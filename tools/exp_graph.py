#!/usr/bin/env python3
"""Developer tool (GPU box): the bias step as a HIP GRAPH against plain stream launches.

Every configuration runs through the reference-shaped API (metadynamics.cv / integrate, C++ run loop System::run) on a
stream of its own (ExecutionConfiguration::setStream: the null stream cannot be captured).  K consecutive steps are
captured once (hipStreamBeginCapture around System::run(K - 1) = prepRun + K - 1 updates, i.e. K bias steps) and replayed;
the same K steps as plain launches are timed beside it.  Timed regions are synchronised on both sides (wall clock), as the
driver's are.  Fresh process per (config, mode), rounds interleaved.

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
    for K, reps in ((20, 60), (2000, 5)):
        if mode in ("stream", "null"):
            def go():
                sysm.run(K - 1)
        else:
            k_graph = 20                # a 20-step graph, replayed K / 20 times
            graph, gexec = C.c_void_p(), C.c_void_p()
            chk(hip.hipStreamBeginCapture(stream, 2), "hipStreamBeginCapture")                 # relaxed
            try:
                sysm.run(k_graph - 1)
            finally:
                rc = hip.hipStreamEndCapture(stream, C.byref(graph))
            chk(rc, "hipStreamEndCapture")
            n_nodes = C.c_size_t()
            chk(hip.hipGraphGetNodes(graph, None, C.byref(n_nodes)), "hipGraphGetNodes")
            out["nodes_per_%d_steps" % k_graph] = n_nodes.value
            t0 = time.perf_counter()
            chk(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")
            out["instantiate_us"] = 1e6 * (time.perf_counter() - t0)

            def go():
                for _ in range(K // k_graph):
                    chk(hip.hipGraphLaunch(gexec, stream), "hipGraphLaunch")
        for _ in range(3):
            go()
        sync()
        ts = []
        for _ in range(reps):
            sync()
            t0 = time.perf_counter()
            go()
            sync()
            ts.append(1e6 * (time.perf_counter() - t0) / K)
        out["K%d_us_per_step_median" % K] = float(np.median(ts))
        out["K%d_us_per_step_min" % K] = float(np.min(ts))
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
