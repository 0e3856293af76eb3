#!/usr/bin/env python3
"""Developer tool (GPU box): which timed regions of the headline step are slow?  A sequence of System::run regions of different
lengths, each bracketed by synchronise like bench.py's, printed as us per step: is it the FIRST region after a region of another
length (or after a pause) that pays, whatever its own length?"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch
import util, bench
pos, types = util.snapshot_random(1_000_000, 100.0, seed=12345, dtype=np.float32)
h = bench.HostEngine(pos, types, 100.0, 1_000_000, 1, 1, "fused")
for _ in range(8):
    h.run(99); torch.cuda.synchronize()
pattern = [20, 20, 20, 5, 20, 20, 2000, 20, 20, 20, 100, 20, 20, 5, 5, 20, "sleep", 20, 20]
out = []
for k in pattern:
    if k == "sleep":
        time.sleep(0.05); out.append("sleep50ms"); continue
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.run(k - 1)
    torch.cuda.synchronize()
    out.append("%d:%.2f" % (k, 1e6 * (time.perf_counter() - t0) / k))
print(" ".join(out))
