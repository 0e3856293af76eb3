import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import bench
def b_poll():
    ev = torch.cuda.Event(); ev.record()
    while not ev.query(): pass
    torch.cuda.synchronize()
def b_sync():
    torch.cuda.synchronize()
eng = bench.Engine(1_000_000, 1_000_000, 0, seed=12345, stride=1, fast_trig=1)
host = bench.HostEngine(eng.pos_np, eng.types_np, eng.L, 1_000_000, 1, 1, "fused")
host.run(300); b_sync()
for name, bar in (("poll+sync", b_poll), ("sync only", b_sync), ("poll+sync", b_poll), ("sync only", b_sync)):
    ts = []
    for rep in range(15):
        bar(); host.run(4); bar()
        t0 = time.perf_counter(); host.run(19); bar()
        ts.append((time.perf_counter() - t0) * 1e6 / 20)
    print("%s: K=20 median %.2f us/step, min %.2f, max %.2f" % (name, np.median(ts), min(ts), max(ts)))
