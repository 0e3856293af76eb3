"""GPU parity: the fused two-launch bias step (mtd_fused_cv_pass / mtd_fused_force_pass) vs the oracle.

The fused path forms dV/ds_c from closed-form post-deposit node values and defers the second
reweighting pass into the next launch; after a flush (get_state / get_array) every grid array,
V(s), w(s), the bias factors and the per-particle forces must match the oracle's
updateBiasPotential + computeBiasForces sequence.
"""
import ctypes as C

import numpy as np
import pytest

import util
from test_gpu_metad import GpuMetad, compare

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

CVS = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]


def make_traj(N, L, steps, dtype=np.float32):
    """a modulated snapshot whose modulation amplitude drifts, so both CVs move from step to step"""
    rng = np.random.default_rng(42)
    base = rng.random((N, 3)) * L - L / 2
    types = (np.arange(N) % 2).astype(np.int32)
    a = np.where(types == 0, 1.0, -1.0)
    out = []
    for t in range(steps):
        pos = base.copy()
        pos[:, 2] += (1.0 + 0.15 * t) * a * np.sin(2 * np.pi * 3 * pos[:, 2] / L)
        pos[:, 0] += (0.4 + 0.1 * t) * a * np.sin(2 * np.pi * (pos[:, 0] + pos[:, 1] + pos[:, 2]) / L)
        out.append(pos.astype(dtype))
    return out, types


class Fused:
    """one bias step through the C ABI: the two-launch form (mtd_fused_cv_pass + mtd_fused_force_pass) or the one-launch
    persistent kernel (mtd_fused_step)"""

    def __init__(self, abi, g, N, dtype, one_launch=False, cvs=None):
        self.abi, self.g, self.N = abi, g, N
        self.lib = abi.load()
        self.cvs = cvs or CVS
        self.n_cv = len(self.cvs)
        self.lset = abi.LamellarSet.make(self.cvs)
        self.dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
        self.tdt = torch.float32 if dtype == np.float32 else torch.float64
        self.scratch = torch.zeros(self.lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
        self.forces = [torch.zeros((N, 4), dtype=self.tdt, device="cuda") for _ in self.cvs]
        self.fptr = (C.c_void_p * self.n_cv)(*[f.data_ptr() for f in self.forces])
        self.registered = False
        self.one_launch = one_launch

    def step(self, t, d_pos, box, n_global=None):
        lib, abi = self.lib, self.abi
        if self.one_launch:
            abi.check(lib.mtd_fused_step_set_mode(self.g.h, 1))
            abi.check(lib.mtd_fused_step(self.g.h, C.byref(self.lset), self.N, abi.ptr(d_pos) if self.N else None, self.fptr, self.dt,
                                         n_global or self.N, C.byref(box), abi.ptr(self.scratch), t, None))
            assert lib.mtd_fused_step_launches(self.g.h) == 1
            return
        n_part = C.c_uint()
        abi.check(lib.mtd_fused_cv_pass(self.g.h, C.byref(self.lset), self.N, abi.ptr(d_pos), self.dt, C.byref(box),
                                        abi.ptr(self.scratch), C.byref(n_part), None))
        if not self.registered:
            for c in range(self.n_cv):
                abi.check(lib.mtd_metad_set_cv_source(self.g.h, c, abi.ptr(self.scratch), n_part.value, self.n_cv, c,
                                                      1.0 / (n_global or self.N), 0.0))
            self.registered = True
        abi.check(lib.mtd_fused_force_pass(self.g.h, C.byref(self.lset), self.N, abi.ptr(d_pos), self.fptr, self.dt,
                                           n_global or self.N, C.byref(box), t, None))


@pytest.mark.parametrize("one_launch", [False, True], ids=["two_launches", "one_launch"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("stride,add_bias", [(1, True), (3, True), (1, False)])
def test_fused_sequence(abi, ref, dtype, stride, add_bias, one_launch):
    N, L, steps = 30011, 50.0, 7
    traj, types = make_traj(N, L, steps, dtype)
    box, rbox = abi.Box.make(L), ref.Box.make(L)
    kw = dict(sigma=[0.02, 0.01], cv_min=[-0.6, -0.3], cv_max=[0.4, 0.3], num_points=[64, 48], W=1.0, T_shift=7.0,
              T=1.0, stride=stride, mode="well_tempered", add_bias=add_bias)

    # run 1: flush (get_state) after every step and compare everything step by step
    g = GpuMetad(abi, **kw)
    r = ref.Metad(**kw)
    cv_log = []
    try:
        f = Fused(abi, g, N, dtype, one_launch)
        for t in range(steps):
            d_pos = torch.from_numpy(util.pack_postype(traj[t], types, dtype)).cuda()
            f.step(t, d_pos, box)
            torch.cuda.synchronize()
            # the forces were computed from the closed-form bias BEFORE any flush: grab them first
            F = [x.cpu().numpy().astype(np.float64) for x in f.forces]
            st = g.state()
            opt = util.oracle_postype(traj[t], types)
            s_ref = [ref.lamellar_cv(v, opt, m, rbox) for v, m in CVS]
            assert abs(st["cv"][0] - s_ref[0]) <= 1e-6 * abs(s_ref[0])
            assert abs(st["cv"][1] - s_ref[1]) <= max(1e-6 * abs(s_ref[1]), 1e-6 * 8 / np.sqrt(N))
            cv_log.append(st["cv"].copy())
            b = r.update_bias(t, st["cv"])          # oracle grid driven with the device's CV values
            compare(g, r, b, label="fused step %d" % t)
            for c, (v, m) in enumerate(CVS):
                F_ref = ref.lamellar_forces(v, opt, m, rbox, b[c])
                scale = np.abs(F_ref[:, :3]).max()
                if scale > 0:
                    assert np.abs(F[c][:, :3] - F_ref[:, :3]).max() <= 1e-5 * scale, (t, c)
                else:
                    assert np.all(F[c][:, :3] == 0.0)
                assert np.all(F[c][:, 3] == 0.0)
    finally:
        g.close()

    # run 2: no host read-back between steps — the deferred apply rides in the next CV launch
    g = GpuMetad(abi, **kw)
    r = ref.Metad(**kw)
    try:
        f = Fused(abi, g, N, dtype, one_launch)
        d_traj = [torch.from_numpy(util.pack_postype(p, types, dtype)).cuda() for p in traj]
        for t in range(steps):
            f.step(t, d_traj[t], box)
        for t in range(steps):
            b = r.update_bias(t, cv_log[t])
        compare(g, r, b, label="fused deferred")
        assert np.array_equal(g.state()["cv"], cv_log[-1])   # deterministic: same bits as run 1
    finally:
        g.close()


@pytest.mark.parametrize("one_launch", [False, True], ids=["two_launches", "one_launch"])
def test_fused_sharded_n_global(abi, ref, one_launch):
    """a shard of a larger system: N_global != N scales CV and forces (what each rank does multi-GPU)"""
    N, L = 10007, 30.0
    traj, types = make_traj(N, L, 1)
    box, rbox = abi.Box.make(L), ref.Box.make(L)
    kw = dict(sigma=[0.02, 0.01], cv_min=[-0.6, -0.3], cv_max=[0.4, 0.3], num_points=[32, 32], W=1.0, T_shift=7.0,
              T=1.0, stride=1, mode="well_tempered")
    g = GpuMetad(abi, **kw)
    r = ref.Metad(**kw)
    try:
        f = Fused(abi, g, N, np.float32, one_launch)
        d_pos = torch.from_numpy(util.pack_postype(traj[0], types, np.float32)).cuda()
        f.step(0, d_pos, box, n_global=4 * N)
        torch.cuda.synchronize()
        F = f.forces[0].cpu().numpy().astype(np.float64)
        st = g.state()
        opt = util.oracle_postype(traj[0], types)
        s_ref = ref.lamellar_cv(util.CV1_VECTORS, opt, util.MODE_AB, rbox, n_global=4 * N)
        assert abs(st["cv"][0] - s_ref) <= 1e-6 * abs(s_ref)
        b = r.update_bias(0, st["cv"])
        compare(g, r, b)
        F_ref = ref.lamellar_forces(util.CV1_VECTORS, opt, util.MODE_AB, rbox, b[0], n_global=4 * N)
        assert np.abs(F[:, :3] - F_ref[:, :3]).max() <= 1e-5 * np.abs(F_ref[:, :3]).max()
    finally:
        g.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fused_force_pass_with_slot_map(abi, dtype):
    """mtd_fused_force_pass_slots: a 3-variable grid whose variable 0 is a host-provided scalar and whose variables 2 and 1
    (in that order) are the two lamellar CVs of the set — against mtd_metad_update_bias + mtd_lamellar_forces on an
    identical engine"""
    import ctypes as C
    lib = abi.load()
    N, L = 30011, 24.0
    traj, types = make_traj(N, L, 4, dtype)
    box = abi.Box.make(L)
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    lset = abi.LamellarSet.make(CVS)
    slots = (C.c_uint * 2)(2, 1)
    dbl = lambda v: (C.c_double * len(v))(*[float(x) for x in v])

    def engine():
        h = C.c_void_p()
        abi.check(lib.mtd_metad_create(C.byref(h), 3, dbl([0.3, 0.03, 0.04]), dbl([-2.0, -1.0, -1.0]), dbl([2.0, 1.0, 1.0]),
                                       (C.c_uint * 3)(12, 20, 16), 1.0, 7.0, 1.0, 1, 1, 1))
        return h

    ha, hb = engine(), engine()
    scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
    fa = [torch.zeros((N, 4), dtype=tdt, device="cuda") for _ in range(2)]
    fb = [torch.zeros((N, 4), dtype=tdt, device="cuda") for _ in range(2)]
    pa, pb = (C.c_void_p * 2)(*[f.data_ptr() for f in fa]), (C.c_void_p * 2)(*[f.data_ptr() for f in fb])
    for t, pos in enumerate(traj):
        d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda()
        n_part = C.c_uint()
        abi.check(lib.mtd_lamellar_cv_partials(C.byref(lset), N, d_pos.data_ptr(), dt, C.byref(box), scratch.data_ptr(), C.byref(n_part), None))
        for h in (ha, hb):
            abi.check(lib.mtd_metad_set_cv_value(h, 0, 0.1 * t - 0.15))
            abi.check(lib.mtd_metad_set_cv_source(h, 2, scratch.data_ptr(), n_part.value, 2, 0, 1.0 / N, 0.0))   # set CV 0 -> variable 2
            abi.check(lib.mtd_metad_set_cv_source(h, 1, scratch.data_ptr(), n_part.value, 2, 1, 1.0 / N, 0.0))   # set CV 1 -> variable 1
        abi.check(lib.mtd_fused_force_pass_slots(ha, C.byref(lset), slots, N, d_pos.data_ptr(), pa, dt, N, C.byref(box), t, None))
        abi.check(lib.mtd_metad_update_bias(hb, t, None))
        d_bias = lib.mtd_metad_bias_device(hb)
        for c, slot in enumerate((2, 1)):
            one = abi.LamellarSet.make([CVS[c]])
            abi.check(lib.mtd_lamellar_forces(C.byref(one), N, d_pos.data_ptr(), (C.c_void_p * 1)(fb[c].data_ptr()), dt, N,
                                              d_bias + 8 * slot, C.byref(box), None))
        torch.cuda.synchronize()
        for c in range(2):
            A, B = fa[c].cpu().numpy().astype(np.float64), fb[c].cpu().numpy().astype(np.float64)
            assert np.abs(B).max() > 0
            assert np.abs(A - B).max() <= 2e-6 * np.abs(B).max()
    cv_a, cv_b = (C.c_double * 3)(), (C.c_double * 3)()
    ba, bb = (C.c_double * 3)(), (C.c_double * 3)()
    Va, Vb = C.c_double(), C.c_double()
    abi.check(lib.mtd_metad_get_state(ha, cv_a, ba, C.byref(Va), None, None, None, None))
    abi.check(lib.mtd_metad_get_state(hb, cv_b, bb, C.byref(Vb), None, None, None, None))
    assert list(cv_a) == list(cv_b)
    assert np.allclose(list(ba), list(bb), rtol=1e-9, atol=1e-12) and Va.value == pytest.approx(Vb.value, rel=1e-12)
    assert lib.mtd_fused_force_pass_slots(ha, C.byref(lset), (C.c_uint * 2)(3, 1), N, d_pos.data_ptr(), pa, dt, N, C.byref(box), 9, None) == -1
    for h in (ha, hb):
        abi.check(lib.mtd_metad_destroy(h))


@pytest.mark.parametrize("one_launch", [True, False], ids=["one_launch", "two_launches"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("fast", [1, 0])
def test_fused_full_size_config2(abi, ref, dtype, fast, one_launch):
    """The instantiation bench.py times (fused two-launch step; fast_trig = 1 is the bench default) at BASELINE.json
    configs[1] size — 10^6 particles, 2 lamellar CVs x 8 modes, well-tempered, modulated parity snapshot: CV values to 1e-6
    (LamellarOrderParameter.cc:42-74), every grid array / V / w after three deposits against the oracle's
    updateBiasPotential (IntegratorMetaDynamics.cc:314-588), per-particle forces against the oracle's computeBiasForces
    (LamellarOrderParameter.cc:77-140) on a 100 000-particle slice to 1e-5 of max|F|, and linearity in the bias factor over
    all 10^6 particles (size-independent property)."""
    lib = abi.load()
    N, L = 1_000_000, 100.0
    pos, types = util.snapshot_random(N, L, seed=12345, modulated=True, dtype=dtype)
    box, rbox = abi.Box.make(L), ref.Box.make(L)
    opt = util.oracle_postype(pos, types)
    kw = dict(sigma=[0.05, 0.01], cv_min=[-1.0, -1.0], cv_max=[1.0, 1.0], num_points=[256, 256], W=1.0, T_shift=7.0,
              T=1.0, stride=1, mode="well_tempered")
    s_ref = [ref.lamellar_cv(v, opt, m, rbox) for v, m in CVS]
    assert abs(s_ref[0]) > 0.05
    lib.mtd_lamellar_set_fast_trig(fast)
    g = GpuMetad(abi, **kw)
    r = ref.Metad(**kw)
    try:
        f = Fused(abi, g, N, dtype, one_launch)
        d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda()
        F_step = []
        for t in range(3):
            f.step(t, d_pos, box)
            torch.cuda.synchronize()
            F_step.append([x.cpu().numpy().astype(np.float64) for x in f.forces])
        st = g.state()
        assert abs(st["cv"][0] - s_ref[0]) <= 1e-6 * abs(s_ref[0]), (st["cv"][0], s_ref[0])
        assert abs(st["cv"][1] - s_ref[1]) <= 1e-6 * 8 / np.sqrt(N), (st["cv"][1], s_ref[1])
        for t in range(3):
            b = r.update_bias(t, st["cv"])           # static snapshot: the same CV values every step
        compare(g, r, b, label="fused 10^6 fast=%d" % fast)
        assert abs(b[0]) > 0 and abs(b[1]) > 0
        sl = slice(0, 100_000)
        for c, (v, m) in enumerate(CVS):
            F_ref = ref.lamellar_forces(v, opt[sl], m, rbox, b[c], n_global=N)
            scale = np.abs(F_ref[:, :3]).max()
            F = F_step[2][c]
            assert np.abs(F[sl, :3] - F_ref[:, :3]).max() <= 1e-5 * scale, (c, fast)
            assert np.all(F[:, 3] == 0.0)
        # all 10^6 particles: the force of step 1 is the force of step 2 times the ratio of the bias factors (same positions;
        # the kernel multiplies the unscaled force by the bias factor at store time)
        r2 = ref.Metad(**kw)
        b_hist = [r2.update_bias(t, st["cv"]) for t in range(3)]
        for c in range(2):
            ratio = b_hist[1][c] / b_hist[2][c]
            scale = np.abs(F_step[2][c][:, :3]).max()
            assert np.abs(F_step[1][c][:, :3] - ratio * F_step[2][c][:, :3]).max() <= 4e-6 * scale * max(1.0, abs(ratio))
    finally:
        lib.mtd_lamellar_set_fast_trig(0)
        g.close()


@pytest.mark.parametrize("n_cv", [1, 2, 3])
@pytest.mark.parametrize("N", [0, 1, 777, 300_000])
def test_one_launch_step_shapes(abi, ref, n_cv, N):
    """mtd_fused_step (persistent kernel) across its envelope: 1-3 collective variables (1-, 2-, 3-dimensional grids of odd
    sizes so that the blocks' grid slices are ragged), empty / single-particle / few-block / many-block launches, standard and
    well-tempered mode, a stride, CV values that leave the grid (the reference's warning path: V = 0, :677-683) — every grid
    array, V, w, the bias factors and the forces against the oracle after every step"""
    lib = abi.load()
    L = 20.0
    cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB), ([(0, 0, 3), (1, 2, 0), (2, 0, -1)], [0.5, -1.5])][:n_cv]
    steps = 5
    traj, types = make_traj(max(N, 1), L, steps, np.float32)
    traj = [p[:N] for p in traj]
    types = types[:N]
    box, rbox = abi.Box.make(L), ref.Box.make(L)
    for mode, stride, cv_max in (("well_tempered", 1, 0.4), ("standard", 2, 0.4), ("well_tempered", 1, 0.05)):
        kw = dict(sigma=[0.02, 0.01, 0.03][:n_cv], cv_min=[-0.6, -0.3, -0.5][:n_cv], cv_max=[cv_max, 0.3, 0.5][:n_cv],
                  num_points=[37, 21, 9][:n_cv], W=1.0, T_shift=7.0, T=1.0, stride=stride, mode=mode)
        g = GpuMetad(abi, **kw)
        r = ref.Metad(**kw)
        try:
            f = Fused(abi, g, N, np.float32, True, cvs)
            for t in range(steps):
                d_pos = torch.from_numpy(util.pack_postype(traj[t], types, np.float32)).cuda() if N else torch.zeros((1, 4), device="cuda")
                f.step(t, d_pos, box, n_global=max(N, 1))
                torch.cuda.synchronize()
                F = [x.cpu().numpy().astype(np.float64) for x in f.forces]
                st = g.state()
                opt = util.oracle_postype(traj[t], types)
                s_ref = [ref.lamellar_cv(v, opt, m, rbox, n_global=max(N, 1)) for v, m in cvs]
                for c in range(n_cv):
                    assert abs(st["cv"][c] - s_ref[c]) <= max(1e-6 * abs(s_ref[c]), 1e-6 * 8 / np.sqrt(max(N, 1))), (mode, t, c)
                b = r.update_bias(t, st["cv"])
                compare(g, r, b, label="one launch %s N=%d n_cv=%d step %d" % (mode, N, n_cv, t))
                # (the engine counts steps whose CV point is off the grid, the oracle every interpolateGrid call that warned)
                if st["oob"] > 0:
                    assert r.num_oob_warnings > 0, (mode, t)
                for c, (v, m) in enumerate(cvs):
                    if N == 0:
                        continue
                    F_ref = ref.lamellar_forces(v, opt, m, rbox, b[c], n_global=max(N, 1))
                    scale = np.abs(F_ref[:, :3]).max()
                    if scale > 0:
                        assert np.abs(F[c][:, :3] - F_ref[:, :3]).max() <= 1e-5 * scale, (mode, t, c)
                    else:
                        assert np.all(F[c][:, :3] == 0.0)
        finally:
            g.close()


def test_one_launch_step_falls_back_outside_its_envelope(abi, ref):
    """more particles than the persistent kernel holds in registers (4096 per compute unit): mtd_fused_step runs the
    two-launch form with the caller's scratch buffer — same results"""
    lib = abi.load()
    N, L = 1_100_000, 100.0
    pos, types = util.snapshot_random(N, L, seed=4, modulated=True, dtype=np.float32)
    box, rbox = abi.Box.make(L), ref.Box.make(L)
    kw = dict(sigma=[0.05, 0.01], cv_min=[-1.0, -1.0], cv_max=[1.0, 1.0], num_points=[64, 64], W=1.0, T_shift=7.0, T=1.0, stride=1,
              mode="well_tempered")
    g = GpuMetad(abi, **kw)
    r = ref.Metad(**kw)
    try:
        f = Fused(abi, g, N, np.float32, False)
        d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
        abi.check(lib.mtd_fused_step_set_mode(g.h, 1))
        for t in range(2):
            abi.check(lib.mtd_fused_step(g.h, C.byref(f.lset), N, abi.ptr(d_pos), f.fptr, f.dt, N, C.byref(box), abi.ptr(f.scratch), t, None))
            assert lib.mtd_fused_step_launches(g.h) == 2
        st = g.state()
        for t in range(2):
            b = r.update_bias(t, st["cv"])
        compare(g, r, b, label="fallback")
        opt = util.oracle_postype(pos[:50000], types[:50000])
        F_ref = ref.lamellar_forces(util.CV1_VECTORS, opt, util.MODE_AB, rbox, b[0], n_global=N)
        F = f.forces[0][:50000].cpu().numpy().astype(np.float64)
        assert np.abs(F[:, :3] - F_ref[:, :3]).max() <= 1e-5 * np.abs(F_ref[:, :3]).max()
    finally:
        g.close()
