#!/usr/bin/env python3
"""Developer tool: BASELINE.json configs[2] (10^6 particles, cv.mesh on 128^3 + 1 lamellar CV, 256^2 grid) through the
reference-shaped API; prints us/step.  Run under rocprofv3 --kernel-trace --stats for the per-kernel table.
usage: bench_mesh.py [steps] [sfc|-] [nx]   sfc: particle ids follow a Morton curve over 64^3 cells, as after HOOMD's SFCPack sorter
(the default, ids uncorrelated with positions, is the worst case for the gathers and scattered stores by id)"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import util
from metadynamics import context, cv, integrate
N, L = 1_000_000, 100.0
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pos, types = util.snapshot_random(N, L, seed=12345, modulated=True, dtype=np.float32)
pos = (np.mod(pos.astype(np.float64) + L / 2, L) - L / 2).astype(np.float32)   # an MD engine keeps its particles wrapped into the box
pos[pos >= L / 2] = -L / 2
if len(sys.argv) > 2 and sys.argv[2] == "sfc":
    c = np.minimum(((pos.astype(np.float64) + L / 2) / L * 64).astype(np.int64), 63)
    key = np.zeros(N, dtype=np.int64)
    for b in range(6):
        for d in range(3):
            key |= ((c[:, d] >> b) & 1) << (3 * b + d)
    order = np.argsort(key, kind="stable")
    pos, types = np.ascontiguousarray(pos[order]), np.ascontiguousarray(types[order])
NX = int(sys.argv[3]) if len(sys.argv) > 3 else 128


def build(lo, hi, sigma):
    context.initialize(pos, types, ["A", "B"], L, dtype=np.float32)
    meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
    lam = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=util.CV1_VECTORS)
    lam.set_grid(-1.0, 1.0, 256)
    mesh = cv.mesh(nx=NX, mode={"A": 1.0, "B": -1.0}, sigma=sigma)
    mesh.set_grid(lo, hi, 256)
    return meta, lam, mesh


# the mesh CV's grid spans value x [0, 2], sigma 1 % of the range — one untimed evaluation supplies the value
meta, lam, mesh = build(0.0, 1.0, 1.0)
context.run(1)
s0 = mesh.cpp_force.getCurrentValue(context.current.system.getCurrentTimeStep())
context.current = None
lo, hi = (0.0, 2.0 * s0) if s0 > 0 else (2.0 * s0, 0.0)
meta, lam, mesh = build(lo, hi, 0.01 * (hi - lo))
context.run(1)
print("mesh cv =", s0, "grid", (lo, hi), "lamellar cv =", lam.cpp_force.getCurrentValue(1))
context.run(10)
torch.cuda.synchronize()
t0 = time.perf_counter()
context.current.system.run(steps - 1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
t_now = context.current.system.getCurrentTimeStep()
print("on grid: %s, hills %d, bias factors %s, V = %g" % (lo <= mesh.cpp_force.getCurrentValue(t_now) < hi, meta.cpp_integrator.getNumGaussians(),
                                                         list(meta.cpp_integrator.getBiasFactors()), meta.cpp_integrator.getLogValue("bias", t_now)))
print("config 3%s%s: %.1f us/step  (%.3e particle-CV-evals/s, 2 CVs)" % (" (ids along a space-filling curve)" if len(sys.argv) > 2 and sys.argv[2] == "sfc" else "", "" if NX == 128 else " with a %d^3 mesh" % NX, 1e6 * dt / steps, 2 * N * steps / dt))
