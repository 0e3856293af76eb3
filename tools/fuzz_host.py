#!/usr/bin/env python3
"""Developer tool (GPU box): the reference-shaped API (metadynamics.cv / integrate over the C++ host classes) with the fused
paths on (pure lamellar sets: two launches; mixed sets: lamellar CVs through launch A + the grid-engine launch) against the
same run with setFusedPath(False) (every CV its own kernels, mtd_metad_update_bias): random CV sets of 1-3 variables out of
lamellar / mesh / lamellar-with-umbrella, random grids, strides, modes.  usage: fuzz_host.py [seconds] [seed]
       fuzz_host.py --replay <case.json>   one case from a recorded generator state (FUZZ_DUMP / FUZZ_OLD_TOL as in fuzz_fused.py)"""
import json, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import util
from metadynamics import context, cv, integrate

replay = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "--replay" else None
budget = float(sys.argv[1]) if len(sys.argv) > 1 and not replay else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 and not replay else 1)
OLD_TOL = os.environ.get("FUZZ_OLD_TOL") == "1"
t0, t_print, it, worst = time.time(), time.time(), 0, dict(cv=0.0, V=0.0, force=0.0)
counts = dict(fused=0, mixed=0)


def one_case(rng):
    N = int(rng.choice([200, 2000, 12000]))
    L = float(rng.uniform(8.0, 20.0))
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    pos, types = util.snapshot_random(N, L, seed=int(rng.integers(1, 10**6)), modulated=True, dtype=np.float64)
    pos = (np.mod(pos + L / 2, L) - L / 2).astype(dtype)
    kinds = [str(k) for k in rng.choice(["lam", "lam", "mesh", "lam_umbrella"], size=int(rng.integers(1, 4)))]
    if kinds.count("mesh") > 1: kinds = ["lam" if (k == "mesh" and i > kinds.index("mesh")) else k for i, k in enumerate(kinds)]
    spec = []
    for k in kinds:
        if k == "mesh":
            spec.append(("mesh", int(rng.choice([8, 12, 16]))))
        else:
            vecs = [tuple(int(x) for x in rng.integers(-3, 4, 3)) for _ in range(int(rng.integers(1, 5)))]
            if all(v == (0, 0, 0) for v in vecs): vecs[0] = (0, 0, 2)
            spec.append((k, vecs))
    stride, mode = int(rng.integers(1, 3)), ("well_tempered" if rng.random() < 0.7 else "standard")
    steps = int(rng.integers(2, 6))
    out = {}
    for fused in (True, False):
        context.initialize(pos, types, ["A", "B"], L, dtype=dtype)
        meta = integrate.mode_metadynamics(dt=0.005, stride=stride, mode=mode, W=1.0, deltaT=5.0, T=1.0)
        cvs = []
        for i, (k, par) in enumerate(spec):
            if k == "mesh":
                c = cv.mesh(nx=par, mode={"A": 1.0, "B": -0.8}, sigma=0.05, name="m%d" % i)
                c.set_grid(0.0, 3.0, 24)
            else:
                c = cv.lamellar(sigma=0.05, mode=dict(A=1.0, B=-1.0), lattice_vectors=par, name="l%d" % i)
                c.set_grid(-1.5, 1.5, 24)
                if k == "lam_umbrella":
                    c.set_params(umbrella="harmonic", kappa=0.7, cv0=0.1)
            cvs.append(c)
        meta.cpp_integrator.setFusedPath(fused)
        context.run(steps)
        t = context.current.system.getCurrentTimeStep()
        integ = meta.cpp_integrator
        out[fused] = dict(cv=list(integ.getCurrentValues()), V=integ.getLogValue("bias", t), n=integ.getNumGaussians(), bias=list(integ.getBiasFactors()),
                          F=[c.cpp_force.getForces().astype(np.float64) for c in cvs], used=integ.usedFusedPath())
        context.current = None
    a, b = out[True], out[False]
    if os.environ.get("FUZZ_HOST_VERBOSE"):
        print("case", it, spec, "N", N, "steps", steps, "\n fused", a["cv"], a["V"], a["n"], a["used"], float(np.abs(a["F"][0]).max()),
              "\n plain", b["cv"], b["V"], b["n"], b["used"], float(np.abs(b["F"][0]).max()), flush=True)
    if a["used"]: counts["fused"] += 1
    elif any(k == "lam" for k, _ in spec) and len(spec) <= 3: counts["mixed"] += 1
    assert a["n"] == b["n"]
    for (k, par), x, y in zip(spec, a["cv"], b["cv"]):
        worst["cv"] = max(worst["cv"], abs(x - y) / max(abs(y), 1e-3))
        # lamellar sums that cancel: the parity tests' floor 1e-6 n_wave / sqrt(N) (the one-launch step, MTD_FUSED_STEP=1, groups
        # the fp32 per-thread sums differently from the generic kernels)
        floor = 1e-6 * len(par) / np.sqrt(N) if k.startswith("lam") else 0.0
        assert abs(x - y) <= max(2e-6 * max(abs(y), 1e-3), floor), ("cv", spec, a["cv"], b["cv"])
    if abs(b["V"]) > 1e-12:
        worst["V"] = max(worst["V"], abs(a["V"] - b["V"]) / abs(b["V"]))
        assert abs(a["V"] - b["V"]) <= 1e-4 * abs(b["V"]), ("V", spec, a["V"], b["V"])
    one_launch = os.environ.get("MTD_FUSED_STEP") == "1"
    for c, (fa, fb) in enumerate(zip(a["F"], b["F"])):
        if spec[c][0].startswith("lam") and abs(a["bias"][c]) > 1e-300 and abs(b["bias"][c]) > 1e-300:
            # The particles do not move, so every hill lands on the same point and dV/ds there is a cancelling difference: the
            # one-launch step — and, since round 3, a mixed set whose lamellar sums ride in the mesh's binning kernel — groups its
            # fp32 sums differently, its CV value differs by ~1e-9 and the bias FACTOR by up to 1e-3 of
            # itself.  What the force kernels contribute is the force per unit bias factor: compared on that.
            # (an umbrella adds its own derivative to the factor the force kernel multiplies with — CollectiveVariable.cc:22-66,
            # harmonic: bias + kappa (s - cv0) — so THAT sum is the unit; dividing by the grid's bias factor alone left
            # kappa (s - cv0) (1 / bias_a - 1 / bias_b) in the comparison: 2.5e-4 in one of 1.7e5 sets once the riders were the default)
            ua = 0.7 * (a["cv"][c] - 0.1) if spec[c][0] == "lam_umbrella" and not OLD_TOL else 0.0
            ub = 0.7 * (b["cv"][c] - 0.1) if spec[c][0] == "lam_umbrella" and not OLD_TOL else 0.0
            if abs(a["bias"][c] + ua) > 1e-300 and abs(b["bias"][c] + ub) > 1e-300:
                fa, fb = fa / (a["bias"][c] + ua), fb / (b["bias"][c] + ub)
        sc = np.abs(fb).max()
        if sc > 1e-20:
            worst["force"] = max(worst["force"], np.abs(fa - fb).max() / sc)
            assert np.abs(fa - fb).max() <= 2e-4 * sc, ("force", c, spec, np.abs(fa - fb).max() / sc)


if replay:
    rec = json.load(open(replay))
    st = rec["state"]
    st["state"] = {k: int(v) for k, v in st["state"].items()}
    rng.bit_generator.state = st
    one_case(rng)
    print("fuzz_host: replayed %s within the tolerances, worst relative deviations %s" % (os.path.basename(replay), {k: float("%.2e" % v) for k, v in worst.items()}))
    sys.exit(0)
while time.time() - t0 < budget:
    it += 1
    if time.time() - t_print > 30.0:
        t_print = time.time()
        print("fuzz_host: %d cases so far" % it, flush=True)
    state = rng.bit_generator.state
    try:
        one_case(rng)
    except AssertionError as e:
        if os.environ.get("FUZZ_DUMP"):
            json.dump({"tool": "fuzz_host.py", "case": it, "seed_args": sys.argv[1:], "error": repr(e)[:600], "state": state}, open(os.environ["FUZZ_DUMP"], "w"), default=str)
        raise
print("fuzz_host: %d random CV sets in %.0f s (%d took the two-launch step, %d the mixed-set launches), worst relative deviations fused vs generic %s"
      % (it, time.time() - t0, counts["fused"], counts["mixed"], {k: float("%.2e" % v) for k, v in worst.items()}))
