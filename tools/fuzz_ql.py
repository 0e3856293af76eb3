#!/usr/bin/env python3
"""Developer tool (GPU box): randomised cross-check of the Steinhardt kernels — full lists visited twice (half_nlist 0), the
symmetric once-per-pair CV pass (2) and half lists (1) — against the oracle, over random densities (crystal / gas), cut-offs,
lmax, Ql_ref patterns with zeros, one or two types, fp32 / fp64 particles.  usage: fuzz_ql.py [seconds] [seed]"""
import ctypes as C, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np, torch
import util, mtd_ref
from metadynamics import _abi
from test_gpu_steinhardt import run_gpu, run_ref

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, t_print, it = time.time(), time.time(), 0
worst = dict(cv=0.0, qlm=0.0, f=0.0, cv_sym=0.0, f_half=0.0)
while time.time() - t0 < budget:
    it += 1
    if time.time() - t_print > 30.0:
        t_print = time.time()
        print("fuzz_ql: %d cases so far" % it, flush=True)
    if rng.random() < 0.6:
        pos, L = util.fcc_lattice(int(rng.integers(3, 6)))
        pos = pos + rng.normal(0, float(rng.uniform(0.0, 0.12)), pos.shape)
    else:
        L = float(rng.uniform(4.0, 8.0))
        pos = rng.random((int(rng.integers(1, 600)), 3)) * L - L / 2
    N = len(pos)
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    pos = pos.astype(dtype)
    types = (rng.random(N) < 0.25).astype(np.int32) if rng.random() < 0.4 else np.zeros(N, dtype=np.int32)
    lmax = int(rng.choice([2, 4, 5, 6, 8]))
    Ql_ref = [float(x) if rng.random() < 0.6 else 0.0 for x in rng.uniform(0.1, 1.0, lmax + 1)]
    if not any(Ql_ref): Ql_ref[-1] = 1.0
    rcut = float(rng.uniform(1.1, min(1.8, 0.45 * L)))
    ron = float(rng.uniform(0.5, 0.98)) * rcut
    nl_full = util.build_nlist(pos.astype(np.float64), L, rcut + float(rng.uniform(0.0, 0.2)))
    nl_half = util.build_nlist(pos.astype(np.float64), L, rcut + 0.1, half=True)
    n_global = N * int(rng.integers(1, 3))
    r = run_ref(mtd_ref, pos.astype(np.float64), types, L, nl_full, rcut, ron, lmax, 0, Ql_ref, n_global=n_global)
    qs, fs = max(np.abs(r[2]).max(), 1e-300), max(np.abs(r[3][:, :3]).max(), 1e-300)
    ftol = 1e-9 if dtype == np.float64 else 2e-7
    for mode in (0, 2):
        g = run_gpu(_abi, pos, types, L, nl_full, rcut, ron, lmax, 0, Ql_ref, dtype, half=mode, n_global=n_global)
        key = "cv" if mode == 0 else "cv_sym"
        if abs(r[0]) > 1e-6 * np.abs(r[1]).max(): worst[key] = max(worst[key], abs(g[0] - r[0]) / abs(r[0]))
        worst["qlm"] = max(worst["qlm"], np.abs(g[2] - r[2]).max() / qs)
        if fs > 1e-3 * max(Ql_ref) * qs * qs / float(n_global) ** 2: worst["f"] = max(worst["f"], np.abs(g[3][:, :3] - r[3][:, :3]).max() / fs)
        assert np.abs(g[2] - r[2]).max() <= 1e-11 * qs, ("Qlm", mode, N, lmax, dtype)
        # (a Ql_ref pattern with odd degrees only leaves rounding noise of the even-degree scale as the CV of a full list)
        assert abs(g[0] - r[0]) <= 1e-10 * abs(r[0]) + 1e-13 * np.abs(r[1]).max(), ("cv", mode, N, lmax, dtype, g[0], r[0])
        # natural scale of a pair's force: w_l |Q_lm|^2 / N^2 — a pattern of odd degrees only leaves forces of rounding-noise size
        f_floor = 1e-9 * max(Ql_ref) * qs * qs / float(n_global) ** 2
        assert np.abs(g[3][:, :3] - r[3][:, :3]).max() <= ftol * fs + f_floor, ("force", mode, N, lmax, dtype, fs, f_floor)
    rh = run_ref(mtd_ref, pos.astype(np.float64), types, L, nl_half, rcut, ron, lmax, 0, Ql_ref, half=True, n_global=n_global)
    gh = run_gpu(_abi, pos, types, L, nl_half, rcut, ron, lmax, 0, Ql_ref, dtype, half=True, n_global=n_global)
    fsh = max(np.abs(rh[3][:, :3]).max(), 1e-300)
    worst["f_half"] = max(worst["f_half"], np.abs(gh[3][:, :3] - rh[3][:, :3]).max() / fsh)
    assert abs(gh[0] - rh[0]) <= 1e-10 * abs(rh[0]) + 1e-13 * np.abs(rh[1]).max(), ("cv half", N, lmax, dtype)
    # fp32 force array + atomic adds of the reaction forces (like the reference's third-law branch in a single-precision build):
    # every pair term is rounded to float before it is added, so the error is relative to the PAIR terms, not to the (partly
    # cancelling) net force; tolerance stated in BASELINE.json: 1e-5
    devh = np.abs(gh[3][:, :3] - rh[3][:, :3]).max()
    S = max(Ql_ref) * qs * qs / float(n_global) ** 2               # scale of one pair term (the net force of a near-perfect crystal is far below it)
    if dtype == np.float32: worst["f_half_over_pair_scale"] = max(worst.get("f_half_over_pair_scale", 0.0), devh / max(S, 1e-300))
    assert devh <= (1e-9 * fsh + f_floor if dtype == np.float64 else 1e-5 * fsh + 2e-5 * S), ("force half", N, lmax, dtype, devh / fsh, devh / max(S, 1e-300))
print("fuzz_ql: %d random cases in %.0f s, worst relative deviations %s" % (it, time.time() - t0, {k: float("%.2e" % v) for k, v in worst.items()}))
