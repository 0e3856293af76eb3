#!/usr/bin/env python3
"""Developer tool (GPU box): per-step duration of the headline step for the first steps after an idle gap of the queue
(events around every step).  usage: ramp_probe.py"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import bench
eng = bench.Engine(1_000_000, 1_000_000, 0, seed=12345, stride=1, fast_trig=1)
for _ in range(300):
    eng.step()
torch.cuda.synchronize()
for gap_ms in (0.0, 0.05, 1.0, 20.0, 200.0):
    res = []
    for rep in range(5):
        for _ in range(200):
            eng.step()
        if gap_ms > 0:
            torch.cuda.synchronize()
            time.sleep(gap_ms * 1e-3)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
        ev[0].record()
        for i in range(40):
            eng.step()
            ev[i + 1].record()
        torch.cuda.synchronize()
        res.append([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(40)])
    m = np.median(np.array(res), axis=0)
    print("idle gap %6.2f ms: steps 1-5 %s | 6-10 mean %.1f | 11-20 mean %.1f | 21-40 mean %.1f | first 20 mean %.2f us" %
          (gap_ms, " ".join("%.1f" % x for x in m[:5]), m[5:10].mean(), m[10:20].mean(), m[20:].mean(), m[:20].mean()), flush=True)
