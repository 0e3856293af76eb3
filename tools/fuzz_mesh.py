#!/usr/bin/env python3
"""Developer tool (GPU box): randomised cross-check of the two assignment / force pipelines of mesh.hip (tiles against
cells) and of both against the oracle, over random mesh sizes (powers of two and not, edge tiles), particle counts,
triclinic boxes, type mixes and particles sitting exactly on the box boundary.  usage: fuzz_mesh.py [seconds] [seed]"""
import ctypes as C, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np, torch
import util, mtd_ref
from metadynamics import _abi
from test_gpu_mesh import GpuMesh

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DIMS = [4, 5, 6, 8, 9, 12, 16, 17, 20, 24, 32, 36, 48, 64]
q9_events = 0
t0, it, worst = time.time(), 0, dict(rho=0.0, cv=0.0, f=0.0, rho_ref=0.0, cv_ref=0.0, f_ref=0.0)
t_print = t0
while time.time() - t0 < budget:
    it += 1
    if time.time() - t_print > 30.0:
        t_print = time.time()
        print("fuzz_mesh: %d cases so far, Q9 events %d" % (it, q9_events), flush=True)
    dims = tuple(int(rng.choice(DIMS)) for _ in range(3))
    if rng.random() < 0.15:
        # meshes 128 cells wide with at least 64 rows: the forward transform reads the tile images itself (k_fft_xy_forward<true>)
        # where the tiles divide the axes, and falls back to the combine launch where they do not (nz = 4, 20)
        dims = (128, int(rng.choice([64, 128])), int(rng.choice([4, 8, 16, 20, 24])))
    N = int(rng.choice([0, 1, 2, 37, 500, 4000, 20000]))
    Ls = tuple(float(x) for x in rng.uniform(3.0, 15.0, 3))
    tilt = dict(xy=float(rng.uniform(-0.3, 0.3)), xz=float(rng.uniform(-0.3, 0.3)), yz=float(rng.uniform(-0.3, 0.3))) if rng.random() < 0.5 else {}
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    n_types = int(rng.integers(1, 4))
    mode = [float(x) for x in rng.uniform(-1.5, 1.5, n_types)]
    f = rng.random((N, 3))
    if N > 3:
        f[0] = [0.0, 0.0, 0.0]; f[1] = [0.999999999, 0.5, 0.0]; f[2] = [0.5, 0.0, 0.999999999]      # on / next to the boundary
    a1 = np.array([Ls[0], 0, 0]); a2 = np.array([tilt.get("xy", 0) * Ls[1], Ls[1], 0])
    a3 = np.array([tilt.get("xz", 0) * Ls[2], tilt.get("yz", 0) * Ls[2], Ls[2]])
    pos = (-0.5 * np.array(Ls) + f[:, :1] * a1 + f[:, 1:2] * a2 + f[:, 2:3] * a3).astype(dtype)
    types = rng.integers(0, n_types, N).astype(np.int32)
    box, rbox = _abi.Box.make(Ls, **tilt), mtd_ref.Box.make(Ls, **tilt)
    dt = _abi.MTD_F32 if dtype == np.float32 else _abi.MTD_F64
    d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda()
    n_global = max(N, 1) * int(rng.integers(1, 3))
    res = {}
    for path in ("tiles", "cells"):
        os.environ["MTD_MESH_ASSIGN"] = path
        g = GpuMesh(_abi, dims, mode, max(N, 1))
        if path == "tiles" and N and it % 2 == 0:
            # the bin pipeline with a plan made for ANOTHER snapshot (overflow lists), then re-planned: the first assignment of a
            # mesh counts and plans, the second bins on that plan, the third on the second's
            other = (-0.5 * np.array(Ls) + (rng.random((N, 1)) ** 3) * a1 + rng.random((N, 1)) * a2 + rng.random((N, 1)) * a3).astype(dtype)
            d_other = torch.from_numpy(util.pack_postype(other, types, dtype)).cuda()
            g.cv(d_other, dt, box, n_global)
            s_stale = g.cv(d_pos, dt, box, n_global)
            rho_stale = g.array(0).copy()
            s = g.cv(d_pos, dt, box, n_global)
            assert s == s_stale and np.array_equal(rho_stale, g.array(0)), ("bin pipeline: stale plan against fresh plan", dims, N)
        else:
            s = g.cv(d_pos, dt, box, n_global)
        rho = g.array(0).copy()
        F = g.forces(d_pos, dt, box, n_global, 0.8) if N else np.zeros((0, 4))
        res[path] = (s, rho, F)
        g.close()
    (s1, r1, F1), (s2, r2, F2) = res["tiles"], res["cells"]
    scale = max(np.abs(r2).max(), 1e-300)
    # the tile path sums in fixed point with one deposit < 2^51: its resolution is 2^-50 of max|mode| over ALL types, which
    # shows when the few particles present all carry a much smaller mode than an absent type
    fx = 2.0 ** -47 * max(abs(a) for a in mode)
    worst["rho"] = max(worst["rho"], np.abs(r1 - r2).max() / scale)
    if s2 != 0.0: worst["cv"] = max(worst["cv"], abs(s1 - s2) / abs(s2))
    if N and np.abs(F2).max() > 0: worst["f"] = max(worst["f"], np.abs(F1 - F2).max() / np.abs(F2).max())
    if np.abs(r1 - r2).max() > 1e-11 * scale + fx:
        r = mtd_ref.Mesh(dims[0], dims[1], dims[2], mode)
        r.cv(util.oracle_postype(pos, types), rbox, n_global=n_global)
        rho_ref = r.array("mesh").real
        print("DEBUG rho: case", it, "tiles-vs-oracle", np.abs(r1 - rho_ref).max(), "cells-vs-oracle", np.abs(r2 - rho_ref).max(), "scale", scale,
              "sum tiles", r1.sum(), "sum cells", r2.sum(), "sum oracle", rho_ref.sum(), "frac", f[:4].tolist(), "dtype", dtype)
    assert np.abs(r1 - r2).max() <= 1e-11 * scale + fx, ("rho", dims, N, tilt)
    assert abs(s1 - s2) <= 1e-9 * max(abs(s2), 1e-300), ("cv", dims, N, tilt, s1, s2)
    ftol = 1e-8 if dtype == np.float64 else 5e-7                      # fp32 force arrays: one rounding
    if N and np.abs(F1 - F2).max() > ftol * max(np.abs(F2).max(), 1e-300):
        k = int(np.argmax(np.abs(F1 - F2).max(axis=1)))
        r = mtd_ref.Mesh(dims[0], dims[1], dims[2], mode)
        opt = util.oracle_postype(pos, types)
        r.cv(opt, rbox, n_global=n_global)
        F_ref = r.forces(opt, rbox, 0.8, n_global=n_global)
        print("DEBUG particle", k, "frac", f[k], "pos", pos[k], "type", types[k], "\n tiles", F1[k], "\n cells", F2[k], "\n oracle", F_ref[k],
              "\n max|F|", np.abs(F2).max(), "n bad", int((np.abs(F1 - F2).max(axis=1) > ftol * np.abs(F2).max()).sum()),
              "tiles-vs-oracle", np.abs(F1[:, :3] - F_ref[:, :3]).max(), "cells-vs-oracle", np.abs(F2[:, :3] - F_ref[:, :3]).max())
    if N: assert np.abs(F1 - F2).max() <= ftol * max(np.abs(F2).max(), 1e-300), ("force", dims, N, tilt, dtype)
    if dims[0] * dims[1] * dims[2] <= 20000 and N <= 4000:
        r = mtd_ref.Mesh(dims[0], dims[1], dims[2], mode)
        opt = util.oracle_postype(pos, types)
        s_ref = r.cv(opt, rbox, n_global=n_global)
        rho_ref = r.array("mesh").real
        sc = max(np.abs(rho_ref).max(), 1e-300)
        worst["rho_ref"] = max(worst["rho_ref"], np.abs(r1 - rho_ref).max() / sc)
        assert np.abs(r1 - rho_ref).max() <= 1e-11 * sc + fx, ("rho vs oracle", dims, N, tilt)
        if s_ref != 0.0:
            worst["cv_ref"] = max(worst["cv_ref"], abs(s1 - s_ref) / abs(s_ref))
            assert abs(s1 - s_ref) <= 1e-8 * abs(s_ref), ("cv vs oracle", dims, N, tilt, s1, s_ref)
        if N:
            F_ref = r.forces(opt, rbox, 0.8, n_global=n_global)
            fm = np.abs(F_ref).max()
            if fm > 0:
                per = np.abs(F1[:, :3] - F_ref[:, :3]).max(axis=1) / fm
                tol = 1e-8 if dtype == np.float64 else 5e-7
                n_bad = int((per > tol).sum())
                # The reference rounds |x| to float inside the TSC derivative (Q9).  The in-cell shift is formed with the
                # reference's operations in the reference's order (mesh.hip: locate), so it is the same double and the rounding
                # falls the same way: EVERY particle within the tolerance, no counted exemption (round 1 allowed one per case).
                if n_bad:
                    q9_events += 1
                assert n_bad == 0, ("force vs oracle", dims, N, tilt, dtype, n_bad, per.max())
                worst["f_ref"] = max(worst["f_ref"], float(per.max()))
print("fuzz_mesh: %d random cases in %.0f s, worst relative deviations %s, Q9 rounding events %d" % (it, time.time() - t0, {k: float("%.2e" % v) for k, v in worst.items()}, q9_events))
