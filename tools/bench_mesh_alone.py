#!/usr/bin/env python3
"""Developer tool: cv.mesh alone on a 1D grid (10^6 particles, 128^3) through the reference-shaped API; prints us/step."""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch, util
from metadynamics import context, cv, integrate
N, L = 1_000_000, 100.0
pos, types = util.snapshot_random(N, L, seed=12345, modulated=True, dtype=np.float32)
pos = (np.mod(pos.astype(np.float64) + L / 2, L) - L / 2).astype(np.float32); pos[pos >= L / 2] = -L / 2
context.initialize(pos, types, ["A", "B"], L, dtype=np.float32)
meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
mesh = cv.mesh(nx=128, mode={"A": 1.0, "B": -1.0}, sigma=7.6e-6)
mesh.set_grid(0.0, 7.647e-4, 512)
context.run(20); torch.cuda.synchronize()
t0 = time.perf_counter(); context.current.system.run(1000); torch.cuda.synchronize()
print("mesh alone: %.1f us/step, hills %d" % (1e6 * (time.perf_counter() - t0) / 1000, meta.cpp_integrator.getNumGaussians()))
