#!/usr/bin/env python3
"""Developer tool (GPU box): the headline step (10^6 particles, 2 lamellar CVs x 8 modes, 256^2 grid, stride 1) through
mtd_fused_step (one persistent launch) and through the two-launch form, back to back, HIP events around n steps.
usage: bench_step.py [steps] [particles] [dtype f32|f64]"""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import util
from metadynamics import _abi

lib = _abi.load()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dtype = np.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else np.float32
L = 100.0
pos, types = util.snapshot_random(N, L, seed=12345, dtype=np.float32)
cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]
lset = _abi.LamellarSet.make(cvs)
box = _abi.Box.make(L)
dt = _abi.MTD_F32 if dtype == np.float32 else _abi.MTD_F64
d_pos = torch.from_numpy(util.pack_postype(pos.astype(dtype), types, dtype)).cuda()
scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
forces = [torch.zeros((N, 4), dtype=d_pos.dtype, device="cuda") for _ in cvs]
fptr = (C.c_void_p * 2)(*[f.data_ptr() for f in forces])
lib.mtd_lamellar_set_fast_trig(1)
dbl = lambda v: (C.c_double * len(v))(*v)


def engine():
    h = C.c_void_p()
    _abi.check(lib.mtd_metad_create(C.byref(h), 2, dbl([1e-3, 1e-3]), dbl([-0.02, -0.02]), dbl([0.02, 0.02]), (C.c_uint * 2)(256, 256),
                                    1.0, 7.0, 1.0, 1, 1, 1))
    return h


def run(one, n):
    h = engine()
    _abi.check(lib.mtd_fused_step_set_mode(h, 1 if one else 0))
    t = [0]
    n_part = C.c_uint()

    def step():
        if one:
            _abi.check(lib.mtd_fused_step(h, C.byref(lset), N, d_pos.data_ptr(), fptr, dt, N, C.byref(box), scratch.data_ptr(), t[0], None))
        else:
            _abi.check(lib.mtd_fused_cv_pass(h, C.byref(lset), N, d_pos.data_ptr(), dt, C.byref(box), scratch.data_ptr(), C.byref(n_part), None))
            if t[0] == 0:
                for c in range(2):
                    _abi.check(lib.mtd_metad_set_cv_source(h, c, scratch.data_ptr(), n_part.value, 2, c, 1.0 / N, 0.0))
            _abi.check(lib.mtd_fused_force_pass(h, C.byref(lset), N, d_pos.data_ptr(), fptr, dt, N, C.byref(box), t[0], None))
        t[0] += 1

    for _ in range(200):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        step()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / n
    cv, bias = (C.c_double * 2)(), (C.c_double * 2)()
    V, w, ng = C.c_double(), C.c_double(), C.c_uint()
    _abi.check(lib.mtd_metad_get_state(h, cv, bias, C.byref(V), C.byref(w), C.byref(ng), None, None))
    launches = lib.mtd_fused_step_launches(h) if one else 2
    _abi.check(lib.mtd_metad_destroy(h))
    return us, dict(cv=list(cv), V=V.value, w=w.value, hills=ng.value, launches=launches)


for rep in range(3):
    for one in (True, False):
        us, st = run(one, steps)
        print("%s: %.2f us/step  (%.1f%% of 8 TB/s on 64 B/particle)  %s" % ("one launch " if one else "two launches", us, 100 * 64.0 * N / (us * 1e-6) / 8e12, st), flush=True)
