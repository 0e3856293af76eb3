#!/usr/bin/env python3
"""Developer tool (GPU box): randomised check of the fused two-launch lamellar bias step (the headline path) against the
oracle: 1-3 CVs, random Miller indices (incl. second harmonics, which the CV pass folds into their fundamentals, negative and
zero components; |index| <= 6 like the configs of BASELINE.json: the fp32 phase carries its rounding |h|+|k|+|l| times), 1-4 particle types, orthorhombic / triclinic boxes, 1 ... 40 000 particles, stride, standard / well-tempered,
fp32 / fp64 particles, fast and accurate trigonometry, values on and off the grid.  usage: fuzz_fused.py [seconds] [seed]
       fuzz_fused.py --replay <case.json>   one case from a recorded generator state (tests/golden/fuzz_*.json: cases a long campaign
       stopped at; FUZZ_DUMP=<file> makes a failing campaign record the state, FUZZ_OLD_TOL=1 applies round 3's first tolerance formula)"""
import ctypes as C, json, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np, torch
import util, mtd_ref
from metadynamics import _abi
from test_gpu_metad import GpuMetad, compare

lib = _abi.load()
replay = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "--replay" else None
budget = float(sys.argv[1]) if len(sys.argv) > 1 and not replay else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 and not replay else 1)
OLD_TOL = os.environ.get("FUZZ_OLD_TOL") == "1"
t0, t_print, it = time.time(), time.time(), 0
worst = dict(cv=0.0, force=0.0, force_fast=0.0)


def one_case(rng):
    n_cv = int(rng.integers(1, 4))
    n_types = int(rng.integers(1, 5))
    N = int(rng.choice([1, 2, 63, 700, 5000, 40000]))
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    fast = bool(rng.random() < 0.3)
    Ls = tuple(float(x) for x in rng.uniform(5.0, 40.0, 3))
    tilt = dict(xy=float(rng.uniform(-0.3, 0.3)), xz=float(rng.uniform(-0.3, 0.3)), yz=float(rng.uniform(-0.3, 0.3))) if rng.random() < 0.4 else {}
    cvs = []
    for c in range(n_cv):
        vecs = []
        for _ in range(int(rng.integers(1, 9))):
            v = tuple(int(x) for x in rng.integers(-3, 4, 3))          # second harmonics reach 6, the largest index of BASELINE.json's configs
            vecs.append(v)
            # second harmonic — while |h| + |k| + |l| stays <= 9 (BASELINE.json's modes: <= 6): the fp32 phase carries its rounding
            # that many times (2e-7 rad each), and with sums of 18 the 1e-5 force tolerance is reached at its edge
            if rng.random() < 0.4 and 2 * sum(abs(x) for x in v) <= 9: vecs.append(tuple(2 * x for x in v))
        cvs.append((vecs, [float(x) for x in rng.uniform(-1.5, 1.5, n_types)]))
    f = rng.random((N, 3))
    a1 = np.array([Ls[0], 0, 0]); a2 = np.array([tilt.get("xy", 0) * Ls[1], Ls[1], 0])
    a3 = np.array([tilt.get("xz", 0) * Ls[2], tilt.get("yz", 0) * Ls[2], Ls[2]])
    types = rng.integers(0, n_types, N).astype(np.int32)
    amp = float(rng.uniform(0.0, 1.0))
    box, rbox = _abi.Box.make(Ls, **tilt), mtd_ref.Box.make(Ls, **tilt)
    steps = int(rng.integers(1, 5))
    stride = int(rng.integers(1, 3))
    wide = rng.random() < 0.8
    kw = dict(sigma=[float(x) for x in rng.uniform(0.02, 0.3, n_cv)], cv_min=[-2.5 if wide else -0.05] * n_cv, cv_max=[2.5 if wide else 0.05] * n_cv,
              num_points=[int(x) for x in rng.integers(4, 40, n_cv)], W=float(rng.uniform(0.2, 2.0)), T_shift=float(rng.uniform(1.0, 9.0)),
              T=float(rng.uniform(0.5, 2.0)), stride=stride, mode="well_tempered" if rng.random() < 0.7 else "standard")
    _abi.check(lib.mtd_lamellar_set_fast_trig(int(fast)))
    g, r = GpuMetad(_abi, **kw), mtd_ref.Metad(**kw)
    lset = _abi.LamellarSet.make(cvs)
    dt = _abi.MTD_F32 if dtype == np.float32 else _abi.MTD_F64
    scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
    forces = [torch.zeros((N, 4), dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda") for _ in cvs]
    fptr = (C.c_void_p * n_cv)(*[x.data_ptr() for x in forces])
    try:
        for t in range(steps):
            ft = f.copy()
            ft[:, 2] += amp * 0.05 * (t + 1) * np.where(types == 0, 1.0, -1.0) * np.sin(2 * np.pi * 2 * ft[:, 2])      # a drifting density wave
            pos = (-0.5 * np.array(Ls) + ft[:, :1] * a1 + ft[:, 1:2] * a2 + ft[:, 2:3] * a3).astype(dtype)
            d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda()
            n_part = C.c_uint()
            _abi.check(lib.mtd_fused_cv_pass(g.h, C.byref(lset), N, d_pos.data_ptr(), dt, C.byref(box), scratch.data_ptr(), C.byref(n_part), None))
            for c in range(n_cv):
                _abi.check(lib.mtd_metad_set_cv_source(g.h, c, scratch.data_ptr(), n_part.value, n_cv, c, 1.0 / N, 0.0))
            _abi.check(lib.mtd_fused_force_pass(g.h, C.byref(lset), N, d_pos.data_ptr(), fptr, dt, N, C.byref(box), t, None))
            torch.cuda.synchronize()
            F = [x.cpu().numpy().astype(np.float64) for x in forces]
            st = g.state()
            opt = util.oracle_postype(pos, types)
            for c, (v, m) in enumerate(cvs):
                s_ref = mtd_ref.lamellar_cv(v, opt, m, rbox)
                amax = max(abs(x) for x in m)
                # cancellation floor of the parity tests (1e-6 n_wave / sqrt(N)), times the largest Miller index: the phases are
                # rounded to fp32 turns once, and h g_1 + k g_2 + l g_3 carries that rounding |h| + |k| + |l| times
                # (the index SUM, as the line above says: until the end of round 3 this used the largest single index, which is the same
                # for axis-aligned modes and three times too strict for (-3, 3, -3) — one case in 2.5e4 at 1.1 of that tolerance,
                # bit for bit the same value from the library of the round's first half)
                index = max(max(abs(x) for x in hkl) for hkl in v) if OLD_TOL else max(sum(abs(x) for x in hkl) for hkl in v)
                floor = len(v) * amax * max(1, index) / np.sqrt(N)
                tol = (1e-6 if not fast else 3e-6) * max(abs(s_ref), floor)
                worst["cv"] = max(worst["cv"], abs(st["cv"][c] - s_ref) / max(abs(s_ref), floor, 1e-300))
                assert abs(st["cv"][c] - s_ref) <= tol, ("cv", c, st["cv"][c], s_ref, N, n_cv, dtype, fast, cvs[c])
            b = r.update_bias(t, st["cv"])
            compare(g, r, b, label="fuzz step %d" % t)
            for c, (v, m) in enumerate(cvs):
                # reference forces with the DEVICE's bias factor: the factor itself is checked by compare() above (near zero it is a
                # difference of nearly equal V values: its relative error is then the forces' too, in any implementation)
                F_ref = mtd_ref.lamellar_forces(v, opt, m, rbox, float(st["bias"][c]))
                scale = np.abs(F_ref[:, :3]).max()
                # (forces below the fp32 range flush to zero in an fp32 force array; with one or two particles max|F| can itself be
                #  a cancellation between modes, and the fp32 phases — rounded once, carried |h|+|k|+|l| times: <= 4e-6 rad at
                #  index 6 — are then measured against that remainder: the force check needs a population)
                # ... and the bias factor has to be a number, not the rounding remainder of a difference of equal V values (a
                # factor of 1e-9 differs by 1e-5 between the closed form launch B uses and the interpolation of the flushed grid
                # that get_state reports: 1e-14 absolute either way)
                if scale > 1e-30 and np.isfinite(b[c]) and N >= 63 and abs(b[c]) > 1e-6 * kw["W"] / kw["sigma"][c]:
                    dev = np.abs(F[c][:, :3] - F_ref[:, :3]).max() / scale
                    worst["force_fast" if fast else "force"] = max(worst["force_fast" if fast else "force"], dev)
                    if dev > 1e-5:
                        k = int(np.argmax(np.abs(F[c][:, :3] - F_ref[:, :3]).max(axis=1)))
                        print("DEBUG force: cv", cvs[c], "Ls", Ls, "tilt", tilt, "particle", k, "frac", ft[k], "type", types[k], "\n gpu", F[c][k], "\n ref", F_ref[k], "scale", scale)
                    assert dev <= 1e-5, ("force", c, dev, N, n_cv, dtype, fast, b[c])
    finally:
        g.close()


if replay:
    rec = json.load(open(replay))
    st = rec["state"]
    st["state"] = {k: int(v) for k, v in st["state"].items()}
    rng.bit_generator.state = st
    one_case(rng)
    print("fuzz_fused: replayed %s within the tolerances, worst relative deviations %s" % (os.path.basename(replay), {k: float("%.2e" % v) for k, v in worst.items()}))
    sys.exit(0)
while time.time() - t0 < budget:
    it += 1
    if time.time() - t_print > 30.0:
        t_print = time.time()
        print("fuzz_fused: %d cases so far" % it, flush=True)
    state = rng.bit_generator.state
    try:
        one_case(rng)
    except AssertionError as e:
        if os.environ.get("FUZZ_DUMP"):
            json.dump({"tool": "fuzz_fused.py", "case": it, "seed_args": sys.argv[1:], "error": repr(e)[:600], "state": state}, open(os.environ["FUZZ_DUMP"], "w"), default=str)
        raise
_abi.check(lib.mtd_lamellar_set_fast_trig(0))
print("fuzz_fused: %d random cases in %.0f s, worst relative deviations %s" % (it, time.time() - t0, {k: float("%.2e" % v) for k, v in worst.items()}))
