#!/usr/bin/env python3
"""Developer tool: BASELINE.json configs[4] / SURVEY config 5 (256 000-particle noisy fcc crystal, cv.steinhardt lmax 6,
full neighbour list r_cut 1.4, 512-point grid) through the reference-shaped API; prints us/step.
Run under rocprofv3 --kernel-trace --stats for the per-kernel table."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import util
from metadynamics import context, cv, integrate
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dtype = np.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else np.float64
pos, L = util.fcc_lattice(40)
pos = pos + np.random.default_rng(777).normal(0, 0.05, pos.shape)
N = len(pos)


def build(hi, sigma):
    context.initialize(pos, np.zeros(N, dtype=np.int32), ["A"], L, dtype=dtype)
    meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
    nl = cv.nlist_cell(r_cut=1.4)
    lists = nl.update()
    st = cv.steinhardt(r_cut=1.4, r_on=1.2, lmax=6, Ql_ref=[0, 0, 0, 0, 1, 0, 1], nlist=nl, type="A", sigma=sigma)
    st.set_grid(0.0, hi, 512)
    return meta, st, lists


# grid [0, 2 s] x 512, sigma 1 % of the range — one untimed evaluation supplies s
t0 = time.perf_counter()
meta, st, lists = build(1.0, 1.0)
print("N = %d, neighbour list built on the host in %.1f s, %.1f neighbours/particle" % (N, time.perf_counter() - t0, len(lists[2]) / N))
context.run(1)
s0 = st.cpp_force.getCurrentValue(1)
context.current = None
meta, st, lists = build(2.0 * s0, 0.02 * s0)
context.run(1)
print("steinhardt cv =", s0, "grid", (0.0, 2.0 * s0))
context.run(5)
torch.cuda.synchronize()
t0 = time.perf_counter()
context.current.system.run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pairs = len(lists[2])
t_now = context.current.system.getCurrentTimeStep()
print("on grid: %s, hills %d, bias factors %s, V = %g" % (0.0 <= st.cpp_force.getCurrentValue(t_now) < 2.0 * s0, meta.cpp_integrator.getNumGaussians(),
                                                         list(meta.cpp_integrator.getBiasFactors()), meta.cpp_integrator.getLogValue("bias", t_now)))
print("config 5 (%s): %.1f us/step  (%.3e particle-CV-evals/s, %.3e pair visits/s incl. CV + force pass)"
      % (np.dtype(dtype).name, 1e6 * dt / steps, N * steps / dt, 2 * pairs * steps / dt))
