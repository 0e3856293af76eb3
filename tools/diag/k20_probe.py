#!/usr/bin/env python3
"""Developer tool (GPU box): fixed cost of bench.py's timed region — wall time of K steps against K for the host-API driver
(System::run in C++) and the C-ABI driver (a Python loop of two launches per step), bracketed exactly as bench.py does.
usage: k20_probe.py"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import bench

def barrier():
    ev = torch.cuda.Event()
    ev.record()
    while not ev.query():
        pass
    torch.cuda.synchronize()

eng = bench.Engine(1_000_000, 1_000_000, 0, seed=12345, stride=1, fast_trig=1)
host = bench.HostEngine(eng.pos_np, eng.types_np, eng.L, 1_000_000, 1, 1, "fused")
for _ in range(300):
    eng.step()
host.run(300)
barrier()
for name, run in (("host", lambda k: host.run(k - 1)), ("abi", lambda k: [eng.step() for _ in range(k)])):
    rows = []
    for K in (1, 2, 5, 10, 20, 40, 100, 400):
        ts = []
        for rep in range(7):
            barrier()
            run(5)
            barrier()
            t0 = time.perf_counter()
            run(K)
            barrier()
            ts.append((time.perf_counter() - t0) * 1e6)
        rows.append((K, float(np.median(ts)), float(np.min(ts))))
    (k1, t1, _), (k2, t2, _) = rows[-2], rows[-1]
    slope = (t2 - t1) / (k2 - k1)
    print("%s driver: us per step from K=100..400: %.2f" % (name, slope))
    for K, med, mn in rows:
        print("   K=%4d: median %8.1f us (min %8.1f)  = %.2f us/step, fixed part %.1f us" % (K, med, mn, med / K, med - slope * K))
# where the fixed part sits: time from the call to the first kernel's start and from the last kernel's end to the return
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(20):
        eng.step()
    ev1.record()
    t_issue = time.perf_counter()
    barrier()
    t1 = time.perf_counter()
    print("abi, 20 steps: wall %.1f us, issue loop %.1f us, events span %.1f us" % ((t1 - t0) * 1e6, (t_issue - t0) * 1e6, ev0.elapsed_time(ev1) * 1e3))
