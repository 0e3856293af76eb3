#!/usr/bin/env python3
"""Developer tool (GPU box): the headline step (10^6 particles, 2 lamellar CVs x 8 modes, 256^2 grid, stride 1, two launches)
under each value of MTD_EXP given on the command line, every value in a fresh process, several rounds, interleaved (box-to-box
and minute-to-minute variation exceeds most kernel tweaks).  Prints us/step, launch A alone, the difference, and the state
(CV values, V, w, hills) so that variants can be compared for equal results.
usage: exp_headline.py [rounds] exp0 exp1 ...      (worker: exp_headline.py --worker)"""
import ctypes as C, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
    import numpy as np, torch
    import util
    from metadynamics import _abi
    lib = _abi.load()
    N, L, steps = int(os.environ.get("EXP_N", "1000000")), 100.0, int(os.environ.get("EXP_STEPS", "3000"))
    pos, types = util.snapshot_random(N, L, seed=12345, dtype=np.float32)
    cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]
    lset, box = _abi.LamellarSet.make(cvs), _abi.Box.make(L)
    d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
    scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
    forces = [torch.zeros((N, 4), dtype=torch.float32, device="cuda") for _ in cvs]
    fptr = (C.c_void_p * 2)(*[f.data_ptr() for f in forces])
    lib.mtd_lamellar_set_fast_trig(int(os.environ.get("EXP_TRIG", "1")))
    dbl = lambda v: (C.c_double * len(v))(*v)
    h = C.c_void_p()
    _abi.check(lib.mtd_metad_create(C.byref(h), 2, dbl([1e-3, 1e-3]), dbl([-0.02, -0.02]), dbl([0.02, 0.02]), (C.c_uint * 2)(256, 256),
                                    1.0, 7.0, 1.0, 1, 1, 1))
    n_part, t = C.c_uint(), [0]

    def cv_pass():
        _abi.check(lib.mtd_fused_cv_pass(h, C.byref(lset), N, d_pos.data_ptr(), 0, C.byref(box), scratch.data_ptr(), C.byref(n_part), None))

    def step():
        cv_pass()
        if t[0] == 0:
            for c in range(2):
                _abi.check(lib.mtd_metad_set_cv_source(h, c, scratch.data_ptr(), n_part.value, 2, c, 1.0 / N, 0.0))
        _abi.check(lib.mtd_fused_force_pass(h, C.byref(lset), N, d_pos.data_ptr(), fptr, 0, N, C.byref(box), t[0], None))
        t[0] += 1

    def timed(fn, n):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / n

    for _ in range(500):
        step()
    us = min(timed(step, steps) for _ in range(3))
    a_us = min(timed(cv_pass, steps) for _ in range(2))
    for _ in range(3):
        step()
    cv, bias = (C.c_double * 2)(), (C.c_double * 2)()
    V, w, ng = C.c_double(), C.c_double(), C.c_uint()
    _abi.check(lib.mtd_metad_get_state(h, cv, bias, C.byref(V), C.byref(w), C.byref(ng), None, None))
    f0 = forces[0][:4, :3].cpu().numpy().ravel().tolist()
    print("RESULT step=%.2f A=%.2f B=%.2f cv=%r bias=%r V=%.15g w=%.15g hills=%d f0=%s" % (us, a_us, us - a_us, list(cv), list(bias), V.value, w.value, ng.value,
                                                                                 ["%.6e" % x for x in f0[:3]]), flush=True)


if __name__ == "__main__":
    if "--worker" in sys.argv:
        worker()
    else:
        rounds = int(sys.argv[1])
        for r in range(rounds):
            for e in sys.argv[2:]:
                env = dict(os.environ)
                for kv in e.split(","):          # "3" or "3,MTD_LAM_CV_BLOCKS=512"
                    if "=" in kv:
                        k, v = kv.split("=")
                        env[k] = v
                    else:
                        env["MTD_EXP"] = kv
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True)
                line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
                print("exp=%-28s %s" % (e, line[0][7:] if line else "FAILED " + out.stderr[-400:]), flush=True)
