"""GPU parity: lamellar CV / force kernels (through the C-ABI) vs the CPU oracle.

Tolerances are BASELINE.json's: CV value 1e-6 relative (absolute floor 1e-6*n_wave/sqrt(N) on
un-modulated random snapshots, where s ~ 1e-3 is pure cancellation), per-particle force 1e-5
relative to max|F|.
"""
import ctypes as C

import numpy as np
import pytest

import util

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


def _box(abi, ref, L, **tilt):
    return abi.Box.make(L, **tilt), ref.Box.make(L, **tilt)


def gpu_cv(abi, cvs, postype_np, box, n_global=None, fast=False):
    lib = abi.load()
    abi.check(lib.mtd_lamellar_set_fast_trig(int(fast)))
    N = postype_np.shape[0]
    n_global = N if n_global is None else n_global
    dt = abi.MTD_F32 if postype_np.dtype == np.float32 else abi.MTD_F64
    d_pos = torch.from_numpy(postype_np).cuda()
    lset = abi.LamellarSet.make(cvs)
    scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
    n_part = C.c_uint(0)
    abi.check(lib.mtd_lamellar_cv_partials(C.byref(lset), N, abi.ptr(d_pos), dt, C.byref(box), abi.ptr(scratch),
                                           C.byref(n_part), None))
    out = torch.zeros(len(cvs), dtype=torch.float64, device="cuda")
    abi.check(lib.mtd_reduce_partials(abi.ptr(scratch), n_part.value, len(cvs), len(cvs), 1.0 / n_global, 0.0,
                                      abi.ptr(out), None))
    torch.cuda.synchronize()
    abi.check(lib.mtd_lamellar_set_fast_trig(0))
    return out.cpu().numpy()


def gpu_forces(abi, cvs, postype_np, box, bias, n_global=None, fast=False):
    lib = abi.load()
    abi.check(lib.mtd_lamellar_set_fast_trig(int(fast)))
    N = postype_np.shape[0]
    n_global = N if n_global is None else n_global
    dt = abi.MTD_F32 if postype_np.dtype == np.float32 else abi.MTD_F64
    tdt = torch.float32 if dt == abi.MTD_F32 else torch.float64
    d_pos = torch.from_numpy(postype_np).cuda()
    lset = abi.LamellarSet.make(cvs)
    forces = [torch.full((N, 4), 7.0, dtype=tdt, device="cuda") for _ in cvs]
    fptr = (C.c_void_p * len(cvs))(*[f.data_ptr() for f in forces])
    d_bias = torch.tensor(bias, dtype=torch.float64, device="cuda")
    abi.check(lib.mtd_lamellar_forces(C.byref(lset), N, abi.ptr(d_pos), fptr, dt, n_global, abi.ptr(d_bias),
                                      C.byref(box), None))
    torch.cuda.synchronize()
    abi.check(lib.mtd_lamellar_set_fast_trig(0))
    return [f.cpu().numpy().astype(np.float64) for f in forces]


def check_forces(F_gpu, F_ref, tol=1e-5):
    scale = np.abs(F_ref[:, :3]).max()
    assert scale > 0
    err = np.abs(F_gpu[:, :3] - F_ref[:, :3]).max() / scale
    assert err <= tol, "force error %.3e relative to max|F|" % err
    assert np.all(F_gpu[:, 3] == 0.0)  # force.w = 0 (LamellarOrderParameter.cc:108)


@pytest.mark.parametrize("fast", [False, True])
def test_config0b_cv_and_forces(abi, ref, fast):
    """BASELINE.json configs[0] wording: 4096 particles, 1 lamellar CV [(0,0,4)], lamellae of period 4a."""
    pos, types, L = util.snapshot_config0b()
    pos = pos.astype(np.float32)
    box, rbox = _box(abi, ref, L)
    lat = [(0, 0, 4)]
    s_ref = ref.lamellar_cv(lat, util.oracle_postype(pos, types), util.MODE_AB, rbox)
    assert abs(s_ref) > 0.4  # strongly ordered by construction
    s = gpu_cv(abi, [(lat, util.MODE_AB)], util.pack_postype(pos, types, np.float32), box, fast=fast)
    assert abs(s[0] - s_ref) <= 1e-6 * abs(s_ref)
    F_ref = ref.lamellar_forces(lat, util.oracle_postype(pos, types), util.MODE_AB, rbox, bias=-3.25)
    F = gpu_forces(abi, [(lat, util.MODE_AB)], util.pack_postype(pos, types, np.float32), box, [-3.25], fast=fast)
    check_forces(F[0], F_ref)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("fast", [False, True])
def test_two_cvs_modulated(abi, ref, dtype, fast):
    """two fused 8-mode CVs (config 2's vectors) on a modulated snapshot, ragged N (not a multiple of anything)"""
    N, L = 50021, 100.0
    pos, types = util.snapshot_random(N, L, seed=99, modulated=True, dtype=dtype)
    box, rbox = _box(abi, ref, L)
    opt = util.oracle_postype(pos, types)
    cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]
    s = gpu_cv(abi, cvs, util.pack_postype(pos, types, dtype), box, fast=fast)
    s1 = ref.lamellar_cv(util.CV1_VECTORS, opt, util.MODE_AB, rbox)
    s2 = ref.lamellar_cv(util.CV2_VECTORS, opt, util.MODE_AB, rbox)
    assert abs(s1) > 0.05
    assert abs(s[0] - s1) <= 1e-6 * abs(s1)
    assert abs(s[1] - s2) <= 1e-6 * 8 / np.sqrt(N)  # cancellation-dominated CV: absolute floor
    bias = [0.8, -1.7]
    F = gpu_forces(abi, cvs, util.pack_postype(pos, types, dtype), box, bias, fast=fast)
    check_forces(F[0], ref.lamellar_forces(util.CV1_VECTORS, opt, util.MODE_AB, rbox, bias[0]))
    check_forces(F[1], ref.lamellar_forces(util.CV2_VECTORS, opt, util.MODE_AB, rbox, bias[1]))


def test_triclinic_three_types(abi, ref):
    """general (tilted, non-cubic, off-centre) box, 3 particle types, 3 CVs with 1/3/5 modes"""
    N = 7001
    rng = np.random.default_rng(5)
    Ls = (11.0, 13.5, 9.25)
    tilt = dict(xy=0.3, xz=-0.2, yz=0.15)
    lo = [-4.0, -7.0, -3.0]
    box = abi.Box.make(Ls, lo=lo, **tilt)
    rbox = ref.Box.make(Ls, lo=lo, **tilt)
    f = rng.random((N, 3))
    a1 = np.array([Ls[0], 0, 0]); a2 = np.array([tilt["xy"] * Ls[1], Ls[1], 0])
    a3 = np.array([tilt["xz"] * Ls[2], tilt["yz"] * Ls[2], Ls[2]])
    pos = (np.array(lo) + f[:, :1] * a1 + f[:, 1:2] * a2 + f[:, 2:3] * a3).astype(np.float32)
    types = rng.integers(0, 3, N).astype(np.int32)
    coeff = [[1.0, -0.5, 0.25], [0.0, 2.0, -1.0], [0.75, 0.75, -1.5]]
    lats = [[(1, 0, 2)], [(0, 2, -1), (3, 1, 0), (-2, 0, 1)], [(1, 1, 1), (2, 0, 0), (0, 0, 5), (-1, 4, 0), (2, -3, 1)]]
    cvs = list(zip(lats, coeff))
    opt = util.oracle_postype(pos, types)
    s = gpu_cv(abi, cvs, util.pack_postype(pos, types, np.float32), box)
    bias = [1.0, -2.0, 0.5]
    F = gpu_forces(abi, cvs, util.pack_postype(pos, types, np.float32), box, bias)
    for c in range(3):
        s_ref = ref.lamellar_cv(lats[c], opt, coeff[c], rbox)
        n_wave = len(lats[c])
        assert abs(s[c] - s_ref) <= max(1e-6 * abs(s_ref), 1e-6 * n_wave / np.sqrt(N))
        check_forces(F[c], ref.lamellar_forces(lats[c], opt, coeff[c], rbox, bias[c]))


def test_known_answers_and_edges(abi, ref):
    """single particle at the origin => s = n_wave * a / N_global; N = 0 => s = 0 and no crash;
    N_global != N (domain-decomposed shard) divides by N_global"""
    box, rbox = _box(abi, ref, 10.0)
    pt = util.pack_postype(np.zeros((1, 3), dtype=np.float32), np.array([1]), np.float32)
    s = gpu_cv(abi, [([(0, 0, 3), (1, 2, 3), (4, 0, 0)], [0.5, 1.5])], pt, box)
    assert s[0] == pytest.approx(3 * 1.5, rel=1e-12)
    s = gpu_cv(abi, [([(0, 0, 3), (1, 2, 3), (4, 0, 0)], [0.5, 1.5])], pt, box, n_global=8)
    assert s[0] == pytest.approx(3 * 1.5 / 8, rel=1e-12)
    empty = np.zeros((0, 4), dtype=np.float32)
    s = gpu_cv(abi, [([(0, 0, 3)], [1.0, -1.0]), ([(1, 0, 0)], [1.0, -1.0])], empty, box, n_global=5)
    assert np.all(s == 0.0)


def test_invalid_arguments(abi):
    lib = abi.load()
    box = abi.Box.make(10.0)
    lset = abi.LamellarSet.make([([(0, 0, 1)], [1.0])])
    lset.n_modes = 0  # empty lattice-vector list is an error in the reference (cv.py:232-234)
    n = C.c_uint(0)
    buf = torch.zeros(16, dtype=torch.float64, device="cuda")
    rc = lib.mtd_lamellar_cv_partials(C.byref(lset), 0, None, abi.MTD_F32, C.byref(box), abi.ptr(buf), C.byref(n), None)
    assert rc == -1
    with pytest.raises(abi.MtdError):
        abi.check(rc)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_dropin_fourier_modes_and_sq_forces(abi, ref, dtype):
    """gpu_calculate_fourier_modes / gpu_compute_sq_forces replacements (one CV, host bias scalar),
    12 modes so the 8-mode chunking is exercised"""
    lib = abi.load()
    N, L = 30011, 40.0
    pos, types = util.snapshot_random(N, L, seed=3, modulated=True, dtype=dtype)
    box, rbox = _box(abi, ref, L)
    lat = util.CV1_VECTORS + [(1, 1, 1), (2, -1, 0), (0, 1, -2), (5, 0, 1)]
    mode = [1.0, -0.5]
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda()
    scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
    out = torch.zeros(2 * len(lat), dtype=torch.float64, device="cuda")
    abi.check(lib.mtd_calculate_fourier_modes(len(lat), util.flat_lattice(lat), N, abi.ptr(d_pos), dt,
                                              util.dbl_array(mode), 2, abi.ptr(out), abi.ptr(scratch),
                                              C.byref(box), None))
    torch.cuda.synchronize()
    modes = out.cpu().numpy().reshape(-1, 2)
    modes_ref = ref.lamellar_fourier_modes(lat, util.oracle_postype(pos, types), mode, rbox)
    # each mode is a sum of N unit-magnitude terms: compare on the scale of the largest mode
    assert np.abs(modes - modes_ref).max() <= 1e-6 * max(np.abs(modes_ref).max(), np.sqrt(N))
    force = torch.zeros((N, 4), dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda")
    abi.check(lib.mtd_compute_sq_forces(N, abi.ptr(d_pos), abi.ptr(force), dt, len(lat), util.flat_lattice(lat),
                                        util.dbl_array(mode), 2, N, 2.5, C.byref(box), None))
    torch.cuda.synchronize()
    check_forces(force.cpu().numpy().astype(np.float64),
                 ref.lamellar_forces(lat, util.oracle_postype(pos, types), mode, rbox, 2.5))


def test_full_size_config2(abi, ref):
    """BASELINE.json configs[1] size: 10^6 particles, 2 lamellar CVs x 8 modes, modulated parity snapshot"""
    N, L = 1_000_000, 100.0
    pos, types = util.snapshot_random(N, L, seed=12345, modulated=True, dtype=np.float32)
    box, rbox = _box(abi, ref, L)
    opt = util.oracle_postype(pos, types)
    cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]
    packed = util.pack_postype(pos, types, np.float32)
    for fast in (False, True):
        s = gpu_cv(abi, cvs, packed, box, fast=fast)
        s1 = ref.lamellar_cv(util.CV1_VECTORS, opt, util.MODE_AB, rbox)
        s2 = ref.lamellar_cv(util.CV2_VECTORS, opt, util.MODE_AB, rbox)
        assert abs(s1) > 0.05
        assert abs(s[0] - s1) <= 1e-6 * abs(s1), (fast, s[0], s1)
        assert abs(s[1] - s2) <= 1e-6 * 8 / np.sqrt(N), (fast, s[1], s2)
    # forces on a 100k-particle slice of the same snapshot (the oracle force loop is the slow part)
    sl = slice(0, 100_000)
    F = gpu_forces(abi, cvs, packed, box, [1.0, 1.0])
    check_forces(F[0][sl], ref.lamellar_forces(util.CV1_VECTORS, opt[sl], util.MODE_AB, rbox, 1.0, n_global=N))
    check_forces(F[1][sl], ref.lamellar_forces(util.CV2_VECTORS, opt[sl], util.MODE_AB, rbox, 1.0, n_global=N))
    # size-independent property: the force kernel is linear in the bias factor
    F2 = gpu_forces(abi, cvs, packed, box, [-2.0, 0.0])
    assert np.allclose(F2[0][:, :3], -2.0 * F[0][:, :3], rtol=2e-6, atol=1e-12)
    assert np.all(F2[1] == 0.0)


def test_fast_trig_only_inside_the_instructions_domain(abi, ref):
    """the hardware sine / cosine take the phase in turns and are only defined on [-256, 256] (beyond it they return 1 / 0):
    a mode set whose phases can leave that range runs the accurate path even with fast trigonometry switched on
    (lamellar.hip::lam_fast_trig: |h| + |k| + |l| <= 100 per mode)"""
    N, L = 20000, 20.0
    pos, types = util.snapshot_random(N, L, seed=77, modulated=True, dtype=np.float32)
    box, rbox = _box(abi, ref, L)
    cvs = [([(600, 0, 0), (0, 599, 1)], util.MODE_AB)]                 # phases up to 300 turns
    pt = util.pack_postype(pos, types, np.float32)
    opt = util.oracle_postype(pos, types)
    s_ref = ref.lamellar_cv(cvs[0][0], opt, cvs[0][1], rbox)
    s_fast = gpu_cv(abi, cvs, pt, box, fast=True)[0]
    s_acc = gpu_cv(abi, cvs, pt, box, fast=False)[0]
    assert s_fast == s_acc                                             # the same kernels ran
    # (the phase itself is a float: 300 turns carry 2e-5 turns of rounding, whatever evaluates the cosine)
    assert abs(s_fast - s_ref) <= 2e-3 * 2 / np.sqrt(N)
    F_fast = gpu_forces(abi, cvs, pt, box, [0.7], fast=True)[0]
    F_acc = gpu_forces(abi, cvs, pt, box, [0.7], fast=False)[0]
    assert np.array_equal(F_fast, F_acc)


def test_trig_mode_per_set(abi, ref):
    """The trigonometry mode is a property of the CV SET (mtd_lamellar_set::trig_mode); the process switch is only the default of
    sets that leave it open.  Two sets with different modes evaluated side by side in one process, under BOTH values of the process
    default: each gives bit for bit what the process-wide switch gave in its mode, and both meet the oracle; the host classes carry
    the mode per variable (cv.lamellar.set_trig_mode) and a fused launch runs accurately if any of its variables asks to."""
    lib = abi.load()
    N, L = 50000, 30.0
    pos, types = util.snapshot_random(N, L, seed=11, modulated=True, dtype=np.float32)
    pt = util.pack_postype(pos, types, np.float32)
    box, rbox = _box(abi, ref, L)
    cvs = [(util.CV1_VECTORS, util.MODE_AB)]
    d_pos = torch.from_numpy(pt).cuda()

    def run(set_mode, process_default):
        abi.check(lib.mtd_lamellar_set_fast_trig(int(process_default)))
        lset = abi.LamellarSet.make(cvs, trig_mode=set_mode)
        scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
        n_part = C.c_uint(0)
        abi.check(lib.mtd_lamellar_cv_partials(C.byref(lset), N, abi.ptr(d_pos), abi.MTD_F32, C.byref(box), abi.ptr(scratch), C.byref(n_part), None))
        out = torch.zeros(1, dtype=torch.float64, device="cuda")
        abi.check(lib.mtd_reduce_partials(abi.ptr(scratch), n_part.value, 1, 1, 1.0 / N, 0.0, abi.ptr(out), None))
        force = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
        fptr = (C.c_void_p * 1)(force.data_ptr())
        d_bias = torch.tensor([0.7], dtype=torch.float64, device="cuda")
        abi.check(lib.mtd_lamellar_forces(C.byref(lset), N, abi.ptr(d_pos), fptr, abi.MTD_F32, N, abi.ptr(d_bias), C.byref(box), None))
        torch.cuda.synchronize()
        abi.check(lib.mtd_lamellar_set_fast_trig(0))
        return out.item(), force.cpu().numpy()

    hw_global, acc_global = run(0, 1), run(0, 0)                       # the process-wide switch, as before
    assert hw_global[0] != acc_global[0] or not np.array_equal(hw_global[1], acc_global[1])      # (the two modes are distinguishable)
    for default in (0, 1):
        hw, acc = run(1, default), run(2, default)                      # per-set modes, whatever the default says
        assert hw[0] == hw_global[0] and np.array_equal(hw[1], hw_global[1])
        assert acc[0] == acc_global[0] and np.array_equal(acc[1], acc_global[1])
    opt = util.oracle_postype(pos, types)
    s_ref = ref.lamellar_cv(util.CV1_VECTORS, opt, util.MODE_AB, rbox)
    F_ref = ref.lamellar_forces(util.CV1_VECTORS, opt, util.MODE_AB, rbox, 0.7)
    for s, F in (hw_global, acc_global):
        assert s == pytest.approx(s_ref, rel=1e-6)
        check_forces(F.astype(np.float64), F_ref)
    bad = abi.LamellarSet.make(cvs, trig_mode=7)
    n_part = C.c_uint(0)
    assert lib.mtd_lamellar_cv_partials(C.byref(bad), N, abi.ptr(d_pos), abi.MTD_F32, C.byref(box), abi.ptr(d_pos), C.byref(n_part), None) == -1

    # host classes: two lamellar CVs with different modes in one fused step = the accurate functions for the launch
    from metadynamics import context, cv, integrate
    try:
        res = {}
        for modes in (("hardware", "accurate"), ("accurate", "accurate"), ("hardware", "hardware")):
            abi.check(lib.mtd_lamellar_set_fast_trig(1))
            context.initialize(pos, types, ["A", "B"], L, dtype=np.float32)
            meta = integrate.mode_metadynamics(dt=0.005, stride=1, mode="well_tempered", W=1.0, deltaT=7.0, T=1.0)
            cs = []
            for i, (vecs, mode) in enumerate(zip((util.CV1_VECTORS, util.CV2_VECTORS), modes)):
                c = cv.lamellar(sigma=0.02, mode=dict(A=1.0, B=-1.0), lattice_vectors=vecs, name="c%d" % i)
                c.set_grid(-1.0, 1.0, 48)
                c.set_trig_mode(mode)
                cs.append(c)
            context.run(2)
            assert meta.cpp_integrator.usedFusedPath()
            res[modes] = list(meta.cpp_integrator.getCurrentValues())
            context.current = None
        assert res[("hardware", "accurate")] == res[("accurate", "accurate")]
        assert res[("hardware", "hardware")] != res[("accurate", "accurate")]
    finally:
        context.current = None
        abi.check(lib.mtd_lamellar_set_fast_trig(0))
