#!/usr/bin/env python3
"""Developer tool (GPU box): mtd_fused_force_pass_slots (the lamellar CVs of a mixed set served by the grid-engine launch, any
slot map) against mtd_metad_update_bias + mtd_lamellar_forces on an identical engine: random grids of 1-3 variables, a random
subset of them lamellar in random order, the others host scalars, random strides / modes / particle counts / dtypes.
usage: fuzz_slots.py [seconds] [seed]"""
import ctypes as C, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import util
from metadynamics import _abi

lib = _abi.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dbl = lambda v: (C.c_double * len(v))(*[float(x) for x in v])
t0, t_print, it, worst = time.time(), time.time(), 0, dict(force=0.0, bias=0.0)
while time.time() - t0 < budget:
    it += 1
    if time.time() - t_print > 30.0:
        t_print = time.time()
        print("fuzz_slots: %d cases so far" % it, flush=True)
    n_grid = int(rng.integers(1, 4))
    n_lam = int(rng.integers(1, n_grid + 1))
    slots = [int(x) for x in rng.permutation(n_grid)[:n_lam]]
    n_types = int(rng.integers(1, 4))
    N = int(rng.choice([1, 63, 700, 5000, 30000]))
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    L = float(rng.uniform(6.0, 30.0))
    cvs = [([tuple(int(x) for x in rng.integers(-3, 4, 3)) for _ in range(int(rng.integers(1, 6)))], [float(x) for x in rng.uniform(-1.5, 1.5, n_types)])
           for _ in range(n_lam)]
    for v, _ in cvs:
        if all(h == (0, 0, 0) for h in v): v[0] = (1, 0, 0)
    lset = _abi.LamellarSet.make(cvs)
    box = _abi.Box.make(L)
    dt = _abi.MTD_F32 if dtype == np.float32 else _abi.MTD_F64
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    pts = [int(x) for x in rng.integers(4, 24, n_grid)]
    args = (dbl(rng.uniform(0.05, 0.4, n_grid)), dbl([-2.0] * n_grid), dbl([2.0] * n_grid), (C.c_uint * n_grid)(*pts),
            float(rng.uniform(0.3, 2.0)), float(rng.uniform(1.0, 8.0)), float(rng.uniform(0.5, 2.0)), int(rng.integers(1, 3)),
            int(rng.random() < 0.7), 1)
    ha, hb = C.c_void_p(), C.c_void_p()
    _abi.check(lib.mtd_metad_create(C.byref(ha), n_grid, *args))
    _abi.check(lib.mtd_metad_create(C.byref(hb), n_grid, *args))
    scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
    fa = [torch.zeros((N, 4), dtype=tdt, device="cuda") for _ in range(n_lam)]
    fb = [torch.zeros((N, 4), dtype=tdt, device="cuda") for _ in range(n_lam)]
    pa = (C.c_void_p * n_lam)(*[x.data_ptr() for x in fa])
    cslots = (C.c_uint * n_lam)(*slots)
    types = rng.integers(0, n_types, N).astype(np.int32)
    base = rng.random((N, 3))
    for t in range(int(rng.integers(1, 5))):
        f = base.copy()
        f[:, 2] += 0.08 * (t + 1) * np.where(types == 0, 1.0, -1.0) * np.sin(2 * np.pi * 2 * f[:, 2])
        pos = ((f - 0.5) * L).astype(dtype)
        d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda()
        n_part = C.c_uint()
        _abi.check(lib.mtd_lamellar_cv_partials(C.byref(lset), N, d_pos.data_ptr(), dt, C.byref(box), scratch.data_ptr(), C.byref(n_part), None))
        for h in (ha, hb):
            for g in range(n_grid):
                if g in slots:
                    _abi.check(lib.mtd_metad_set_cv_source(h, g, scratch.data_ptr(), n_part.value, n_lam, slots.index(g), 1.0 / N, 0.0))
                else:
                    _abi.check(lib.mtd_metad_set_cv_value(h, g, 0.3 * np.sin(0.7 * t + g)))
        _abi.check(lib.mtd_fused_force_pass_slots(ha, C.byref(lset), cslots, N, d_pos.data_ptr(), pa, dt, N, C.byref(box), t, None))
        _abi.check(lib.mtd_metad_update_bias(hb, t, None))
        d_bias = lib.mtd_metad_bias_device(hb)
        for c in range(n_lam):
            one = _abi.LamellarSet.make([cvs[c]])
            _abi.check(lib.mtd_lamellar_forces(C.byref(one), N, d_pos.data_ptr(), (C.c_void_p * 1)(fb[c].data_ptr()), dt, N,
                                               d_bias + 8 * slots[c], C.byref(box), None))
        torch.cuda.synchronize()
        ba, bb = (C.c_double * n_grid)(), (C.c_double * n_grid)()
        ca, cb = (C.c_double * n_grid)(), (C.c_double * n_grid)()
        _abi.check(lib.mtd_metad_get_state(ha, ca, ba, None, None, None, None, None))
        _abi.check(lib.mtd_metad_get_state(hb, cb, bb, None, None, None, None, None))
        assert list(ca) == list(cb), ("cv", list(ca), list(cb))
        bs = max(max(abs(x) for x in bb), 1e-300)
        worst["bias"] = max(worst["bias"], max(abs(x - y) for x, y in zip(ba, bb)) / bs)
        assert max(abs(x - y) for x, y in zip(ba, bb)) <= 1e-9 * bs, ("bias", list(ba), list(bb), slots)
        for c in range(n_lam):
            A, B = fa[c].cpu().numpy().astype(np.float64), fb[c].cpu().numpy().astype(np.float64)
            sc = np.abs(B).max()
            if sc > 1e-30 and abs(bb[slots[c]]) > 1e-9 * bs and N >= 63:     # (with one particle max|F| is itself a cancellation between modes)
                worst["force"] = max(worst["force"], np.abs(A - B).max() / sc)
                assert np.abs(A - B).max() <= 5e-6 * sc, ("force", c, slots, N, dtype, np.abs(A - B).max() / sc)
    for h in (ha, hb):
        _abi.check(lib.mtd_metad_destroy(h))
print("fuzz_slots: %d random cases in %.0f s, worst relative deviations %s" % (it, time.time() - t0, {k: float("%.2e" % v) for k, v in worst.items()}))
