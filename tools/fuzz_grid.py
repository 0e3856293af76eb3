#!/usr/bin/env python3
"""Developer tool (GPU box): randomised check of the bias-grid engine (mtd_metad_update_bias: the fused grid step for <= 3
variables, the four-launch sequence above that) against the oracle: 1-5 variables, random grids / widths / temperatures,
stride, standard and well-tempered, trajectories that wander on and off the grid and sit exactly on nodes and edges,
reset_histogram, add_hills toggles.  usage: fuzz_grid.py [seconds] [seed]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np
import mtd_ref
from metadynamics import _abi
from test_gpu_metad import GpuMetad, compare

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, t_print, it, steps_total = time.time(), time.time(), 0, 0
tail_cases = 0
while time.time() - t0 < budget:
    it += 1
    if time.time() - t_print > 30.0:
        t_print = time.time()
        print("fuzz_grid: %d cases, %d steps so far" % (it, steps_total), flush=True)
    n_cv = int(rng.choice([1, 1, 2, 2, 3, 4, 5]))
    pts = [int(x) for x in rng.integers(2, 14 if n_cv > 2 else 60, n_cv)]
    lo = [float(x) for x in rng.uniform(-3.0, 1.0, n_cv)]
    hi = [l + float(x) for l, x in zip(lo, rng.uniform(0.5, 4.0, n_cv))]
    kw = dict(sigma=[float(x) for x in rng.uniform(0.03, 0.8, n_cv)], cv_min=lo, cv_max=hi, num_points=pts,
              W=float(rng.uniform(0.1, 3.0)), T_shift=float(rng.uniform(0.5, 10.0)), T=float(rng.uniform(0.3, 3.0)),
              stride=int(rng.integers(1, 4)), mode="well_tempered" if rng.random() < 0.6 else "standard", add_bias=bool(rng.random() < 0.9))
    g, r = GpuMetad(_abi, **kw), mtd_ref.Metad(**kw)
    try:
        s = np.array([rng.uniform(l, h) for l, h in zip(lo, hi)])
        hist_s = []
        for t in range(int(rng.integers(1, 12))):
            steps_total += 1
            s = s + rng.normal(0, 0.15, n_cv) * (np.array(hi) - np.array(lo))
            u = rng.random()
            if u < 0.15:                                                   # exactly on a node / on an edge of the grid
                i = int(rng.integers(0, n_cv))
                delta = (hi[i] - lo[i]) / (pts[i] - 1)
                s[i] = lo[i] + delta * int(rng.integers(0, pts[i])) if rng.random() < 0.7 else (lo[i] if rng.random() < 0.5 else hi[i])
            elif u < 0.25:
                s = np.clip(s, np.array(lo) - 0.2, np.array(hi) + 0.2)
            hist_s.append(s.copy())
            g.step(t, list(s))
            b = r.update_bias(t, list(s))
            try:
                try:
                    compare(g, r, b, label="fuzz_grid case %d step %d %s" % (it, t, kw))
                except AssertionError:
                    # V, w, dV/ds relative to themselves is too strict far out in a Gaussian tail: with the CV exactly on a node,
                    # (s - min) / delta = k + 3e-16 picks node k + 1 with weight 3e-16, and that node can be 10^11 times larger.
                    # There the scalars are checked against the scale of the grid they interpolate (the arrays themselves already
                    # passed inside compare()).
                    st = g.state()
                    scale_v = max(np.abs(r.array("grid")).max(), 1e-300)
                    dmin = min((h - l) / (p - 1) for l, h, p in zip(lo, hi, pts))
                    ok = abs(st["V"] - r.curr_bias) <= 1e-11 * scale_v or (np.isnan(st["V"]) and np.isnan(r.curr_bias))
                    ok = ok and (np.allclose(st["bias"], b, rtol=1e-9, atol=1e-9 * scale_v / dmin) or np.isnan(b).any())
                    ws = max(np.nanmax(np.abs(r.array("weight"))), 1e-300)
                    ok = ok and (abs(st["w"] - r.curr_weight) <= 1e-11 * ws or (np.isnan(st["w"]) and np.isnan(r.curr_weight)))
                    ok = ok and st["num_gaussians"] == r.num_gaussians
                    for name in ("hist", "hist_delta", "hist_gauss", "hist_gauss_delta"):
                        ok = ok and np.array_equal(g.array(name), r.array(name))
                    if not ok: raise
                    tail_cases += 1
            except AssertionError:
                st = g.state()
                np.savez(os.path.join(root, "gpurun_out", "fuzz_grid_fail.npz"), hist=np.array(hist_s), grid_gpu=g.array("grid"), grid_ref=r.array("grid"), kw=repr(kw))
                print("DEBUG s", repr(s), "gpu V w bias", st["V"], st["w"], st["bias"], "ref V w bias", r.curr_bias, r.curr_weight, b, "oob", st["oob"], flush=True)
                raise
    finally:
        g.close()
print("fuzz_grid: %d random cases, %d steps in %.0f s: every grid array, V, w, dV/ds, counters as the oracle's (%d steps judged on the grid's scale: Gaussian tails)" % (it, steps_total, time.time() - t0, tail_cases))
