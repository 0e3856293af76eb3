#!/usr/bin/env python3
"""Developer tool (GPU box): would launch B of the headline step gain from finding the positions in the L2s?  Launch B is issued
twice per step (same time step: the second only deposits another hill) — the second one reads positions that a launch with the
same block -> particle mapping read 13 us earlier, i.e. on the same XCDs.  Per-launch durations from the launch's own events.
usage: l2_reuse_probe.py"""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import bench
from metadynamics import _abi
lib = _abi.load()
eng = bench.Engine(1_000_000, 1_000_000, 0, seed=12345, stride=1, fast_trig=1)
be = eng.be
for _ in range(300):
    eng.step()
torch.cuda.synchronize()
n = 200
_abi.check(lib.mtd_profile_force_begin(2 * n))
t = 1000
for i in range(n):
    be.cv_partials()
    be.force_pass(None, t)
    be.force_pass(None, t)
    t += 1
buf = (C.c_double * (2 * n))()
got = C.c_uint()
_abi.check(lib.mtd_profile_force_end(buf, 2 * n, C.byref(got)))
d = np.array(buf[:got.value])
print("launch B after launch A      : median %.2f us" % np.median(d[0::2]))
print("launch B after another launch B: median %.2f us  (positions read 13 us earlier by the same blocks)" % np.median(d[1::2]))
