"""GPU parity: Steinhardt Q_l kernels (C-ABI) vs the oracle restatement of SteinhardtQl.cc (+ fsph).
Double precision on both sides: Q_lm to 1e-11 of its scale, the CV to 1e-10 relative (stated tolerance 1e-6),
forces to 1e-9 of max|F| in fp64 (stated 1e-5)."""
import ctypes as C

import numpy as np
import pytest

import util

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


def run_gpu(abi, pos, types, L, nl, rcut, ron, lmax, type_id, Ql_ref, dtype, half=False, n_global=None, bias=0.9, tilt=None):
    lib = abi.load()
    N = len(pos)
    n_global = N if n_global is None else n_global
    box = abi.Box.make(L, **(tilt or {}))
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    d_pos = torch.from_numpy(util.pack_postype(pos.astype(dtype), types, dtype)).cuda()
    d_head, d_nn, d_nl = (torch.from_numpy(x.astype(np.int32)).cuda() for x in nl)
    scratch = torch.zeros(lib.mtd_ql_scratch_doubles(lmax), dtype=torch.float64, device="cuda")
    p_val, p_ql, p_qlm = C.c_void_p(), C.c_void_p(), C.c_void_p()
    abi.check(lib.mtd_ql_accumulate(N, abi.ptr(d_pos), dt, C.byref(box), abi.ptr(d_head), abi.ptr(d_nn), abi.ptr(d_nl), int(half),
                                    rcut, ron, lmax, type_id, util.dbl_array(Ql_ref), n_global, abi.ptr(scratch),
                                    C.byref(p_val), C.byref(p_ql), C.byref(p_qlm), None))
    torch.cuda.synchronize()
    base = scratch.data_ptr()
    s = scratch.cpu().numpy()
    off = lambda p: (p.value - base) // 8
    val = s[off(p_val)]
    Ql = s[off(p_ql):off(p_ql) + lmax + 1].copy()
    q = s[off(p_qlm):off(p_qlm) + 2 * (lmax + 1) ** 2]
    Qlm = q[0::2] + 1j * q[1::2]
    force = torch.full((N, 4), 3.0, dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda")
    d_bias = torch.tensor([bias], dtype=torch.float64, device="cuda")
    abi.check(lib.mtd_ql_forces(N, abi.ptr(d_pos), abi.ptr(force), dt, C.byref(box), abi.ptr(d_head), abi.ptr(d_nn), abi.ptr(d_nl),
                                int(half), rcut, ron, lmax, type_id, util.dbl_array(Ql_ref), n_global, abi.ptr(scratch),
                                abi.ptr(d_bias), 0.0, None))
    torch.cuda.synchronize()
    return val, Ql, Qlm, force.cpu().numpy().astype(np.float64)


def run_ref(ref, pos, types, L, nl, rcut, ron, lmax, type_id, Ql_ref, half=False, n_global=None, bias=0.9, tilt=None):
    box = ref.Box.make(L, **(tilt or {}))
    pt = util.oracle_postype(pos, types)
    val, Qlm, Ql = ref.ql_compute_cv(pt, box, *nl, rcut, ron, lmax, type_id, Ql_ref, half=half, n_global=n_global)
    F = ref.ql_compute_forces(pt, box, *nl, rcut, ron, lmax, type_id, Ql_ref, Qlm, bias, half=half, n_global=n_global)
    return val, Ql, Qlm, F


def noisy_fcc(n, sigma=0.05, seed=777):
    pos, L = util.fcc_lattice(n)
    rng = np.random.default_rng(seed)
    return pos + rng.normal(0, sigma, pos.shape), L


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("lmax,Ql_ref", [(6, [0, 0, 0, 0, 1, 0, 1]), (4, [0.2, 0, 1.0, 0.5, 1.0]), (8, [0, 0, 0, 0, 1, 0, 1, 0, 0.5]),
                                         # run-time lmax below the compiled table size (2 -> 4, 5 -> 6, 10 -> 12) and the largest one
                                         (2, [0.5, 0.3, 1.0]), (5, [0, 0, 0.2, 0, 1, 0.7]), (10, [0] * 10 + [1.0]),
                                         (12, [0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0.5, 0, 0.25])])
@pytest.mark.parametrize("half", [False, True])
def test_ql_parity(abi, ref, dtype, lmax, Ql_ref, half):
    pos, L = noisy_fcc(5)
    pos = pos.astype(dtype)                      # the snapshot is the rounded array
    N = len(pos)
    types = np.zeros(N, dtype=np.int32)
    rcut, ron = 1.4, 1.2
    nl = util.build_nlist(pos.astype(np.float64), L, rcut + 0.15, half=half)
    g = run_gpu(abi, pos, types, L, nl, rcut, ron, lmax, 0, Ql_ref, dtype, half=half)
    r = run_ref(ref, pos.astype(np.float64), types, L, nl, rcut, ron, lmax, 0, Ql_ref, half=half)
    qs = np.abs(r[2]).max()
    assert np.abs(g[2] - r[2]).max() <= 1e-11 * qs
    assert np.allclose(g[1], r[1], rtol=1e-10, atol=1e-13 * np.abs(r[1]).max())
    assert g[0] == pytest.approx(r[0], rel=1e-10)
    fs = np.abs(r[3][:, :3]).max()
    tol = 1e-9 if dtype == np.float64 else 2e-7    # fp32 force array: one rounding on store, half lists too (exact integer sums)
    assert np.abs(g[3][:, :3] - r[3][:, :3]).max() <= tol * fs
    assert np.all(g[3][:, 3] == 0.0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ql_half_list_forces_bitwise_reproducible(abi, dtype):
    """half lists: the reaction forces of a pair go to the other particle from whichever block holds the pair — summed as exact
    integers (steinhardt.hip: ql_exact_add), so two runs give the same bits, and a list with its rows permuted (another
    order of the adds) too"""
    pos, L = noisy_fcc(6, seed=9)
    pos = pos.astype(dtype)
    N = len(pos)
    types = np.zeros(N, dtype=np.int32)
    nl = util.build_nlist(pos.astype(np.float64), L, 1.55, half=True)
    args = (1.4, 1.2, 6, 0, [0, 0, 0, 0, 1, 0, 1])
    runs = [run_gpu(abi, pos, types, L, nl, *args, dtype, half=True)[3] for _ in range(3)]
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])
    assert np.abs(runs[0][:, :3]).max() > 0
    # the same pairs listed in another order inside every row
    head, nn, lst = [np.array(a).copy() for a in nl]
    rng = np.random.default_rng(2)
    for i in range(N):
        seg = lst[head[i]:head[i] + nn[i]]
        lst[head[i]:head[i] + nn[i]] = rng.permutation(seg)
    shuffled = run_gpu(abi, pos, types, L, (head, nn, lst), *args, dtype, half=True)[3]
    # own-role sums follow the list order (like the reference's loop), so only agreement to rounding is asked of them
    assert np.abs(shuffled[:, :3] - runs[0][:, :3]).max() <= (1e-13 if dtype == np.float64 else 2e-7) * np.abs(runs[0][:, :3]).max()
    # the floating-point atomics stay available (mtd_ql_set_half_list_exact(0)): same forces to rounding
    lib = abi.load()
    abi.check(lib.mtd_ql_set_half_list_exact(0))
    try:
        fast = run_gpu(abi, pos, types, L, nl, *args, dtype, half=True)[3]
    finally:
        abi.check(lib.mtd_ql_set_half_list_exact(1))
    assert np.abs(fast[:, :3] - runs[0][:, :3]).max() <= (1e-12 if dtype == np.float64 else 2e-6) * np.abs(runs[0][:, :3]).max()


def test_ql_two_types_and_shard(abi, ref):
    """only particles of `type` take part (SteinhardtQl.cc:105, 126); N_global != N; triclinic box"""
    pos, L = noisy_fcc(4, seed=5)
    N = len(pos)
    rng = np.random.default_rng(1)
    types = (rng.random(N) < 0.3).astype(np.int32)
    nl = util.build_nlist(pos, L, 1.6)
    args = (1.45, 1.1, 6, 0, [0.5, 0, 0.25, 0, 1, 0, 1])
    g = run_gpu(abi, pos, types, L, nl, *args, np.float64, n_global=3 * N)
    r = run_ref(ref, pos, types, L, nl, *args, n_global=3 * N)
    assert g[0] == pytest.approx(r[0], rel=1e-10)
    assert np.abs(g[3] - r[3]).max() <= 1e-9 * np.abs(r[3]).max()
    assert np.all(g[3][types == 1] == 0.0)          # other types carry no force (memset at :236)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ql_config5_size(abi, ref, dtype):
    """BASELINE.json configs[4]: 256 000 particles (40^3 fcc cells), lmax = 6, r_cut 1.4, r_on 1.2, Ql_ref [0,0,0,0,1,0,1]:
    CV and Q_l against the oracle on the whole system (SteinhardtQl.cc:62-201), forces against the oracle's
    computeBiasForces (SteinhardtQl.cc:203-339) on the first 30 000 central particles (the oracle takes a subset of the
    head list and the full position array), stated tolerance 1e-5 of max|F|"""
    pos, L = noisy_fcc(40)
    N = len(pos)
    assert N == 256000
    pos = pos.astype(dtype)
    types = np.zeros(N, dtype=np.int32)
    nl = util.build_nlist(pos.astype(np.float64), L, 1.4)
    Ql_ref = [0, 0, 0, 0, 1, 0, 1]
    g = run_gpu(abi, pos, types, L, nl, 1.4, 1.2, 6, 0, Ql_ref, dtype)
    box = ref.Box.make(L)
    pt = util.oracle_postype(pos, types)
    val, Qlm, Ql = ref.ql_compute_cv(pt, box, *nl, 1.4, 1.2, 6, 0, Ql_ref)
    assert g[0] == pytest.approx(val, rel=1e-6)
    assert np.allclose(g[1], Ql, rtol=1e-6, atol=1e-9 * np.abs(Ql).max())
    assert np.abs(g[2] - Qlm).max() <= 1e-9 * np.abs(Qlm).max()
    # noisy fcc keeps most of the ideal order: Q6(code) below but near 144 * 0.57452^2
    assert 0.5 * 144 * 0.57452 ** 2 < Ql[6] < 144 * 0.57452 ** 2
    n_sl = 30000
    F_ref = ref.ql_compute_forces(pt, box, nl[0][:n_sl], nl[1][:n_sl], nl[2], 1.4, 1.2, 6, 0, Ql_ref, Qlm, 0.9, n_global=N)[:n_sl]
    fs = np.abs(F_ref[:, :3]).max()
    assert fs > 0
    err = np.abs(g[3][:n_sl, :3] - F_ref[:, :3]).max() / fs
    assert err <= (1e-6 if dtype == np.float32 else 1e-8), err
    assert np.isfinite(g[3]).all() and np.all(g[3][:, 3] == 0.0)
    # size-independent property over all particles: a symmetric full list => the pair forces cancel, sum F = 0
    assert np.abs(g[3][:, :3].sum(axis=0)).max() <= 1e-4 * fs * np.sqrt(N)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ql_symmetric_full_list_visits_each_pair_once(abi, ref, dtype):
    """half_nlist = 2: a symmetric full list without ghosts — the CV pass visits each pair from its lower index and scales
    like a half list; the reference's result for the FULL list is reproduced (odd degrees cancel exactly instead of to
    rounding), the force pass is the full-list one"""
    pos, L = noisy_fcc(6, seed=3)
    pos = pos.astype(dtype)
    N = len(pos)
    types = (np.random.default_rng(2).random(N) < 0.15).astype(np.int32)      # two types: only type 0 takes part
    nl = util.build_nlist(pos.astype(np.float64), L, 1.55)                    # list with a buffer beyond r_cut
    Ql_ref = [0.3, 0.2, 0.1, 0.4, 1.0, 0.6, 1.0]
    lib = abi.load()
    box = abi.Box.make(L)
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda()
    d_head, d_nn, d_nl = (torch.from_numpy(x.astype(np.int32)).cuda() for x in nl)
    out = {}
    for mode in (0, 2):
        scratch = torch.zeros(lib.mtd_ql_scratch_doubles(6), dtype=torch.float64, device="cuda")
        p_val, p_ql, p_qlm = C.c_void_p(), C.c_void_p(), C.c_void_p()
        abi.check(lib.mtd_ql_accumulate(N, abi.ptr(d_pos), dt, C.byref(box), abi.ptr(d_head), abi.ptr(d_nn), abi.ptr(d_nl), mode,
                                        1.4, 1.2, 6, 0, util.dbl_array(Ql_ref), N, abi.ptr(scratch),
                                        C.byref(p_val), C.byref(p_ql), C.byref(p_qlm), None))
        force = torch.zeros((N, 4), dtype=d_pos.dtype, device="cuda")
        d_bias = torch.tensor([0.7], dtype=torch.float64, device="cuda")
        abi.check(lib.mtd_ql_forces(N, abi.ptr(d_pos), abi.ptr(force), dt, C.byref(box), abi.ptr(d_head), abi.ptr(d_nn), abi.ptr(d_nl),
                                    mode, 1.4, 1.2, 6, 0, util.dbl_array(Ql_ref), N, abi.ptr(scratch), abi.ptr(d_bias), 0.0, None))
        torch.cuda.synchronize()
        s = scratch.cpu().numpy()
        off = lambda p: (p.value - scratch.data_ptr()) // 8
        q = s[off(p_qlm):off(p_qlm) + 2 * 49]
        out[mode] = (s[off(p_val)], s[off(p_ql):off(p_ql) + 7].copy(), q[0::2] + 1j * q[1::2], force.cpu().numpy().astype(np.float64))
    r = run_ref(ref, pos.astype(np.float64), types, L, nl, 1.4, 1.2, 6, 0, Ql_ref, bias=0.7)
    qs = np.abs(r[2]).max()
    for mode in (0, 2):
        val, Ql, Qlm, F = out[mode]
        assert np.abs(Qlm - r[2]).max() <= 1e-11 * qs
        assert val == pytest.approx(r[0], rel=1e-10)
        assert np.allclose(Ql, r[1], rtol=1e-10, atol=1e-13 * np.abs(r[1]).max())
    # odd degrees: exact zeros in the once-per-pair pass
    odd = [l * l + p for l in (1, 3, 5) for p in range(2 * l + 1)]
    assert np.all(out[2][2][odd] == 0.0)
    fs = np.abs(r[3][:, :3]).max()
    tol = 1e-9 if dtype == np.float64 else 2e-7
    assert np.abs(out[2][3][:, :3] - r[3][:, :3]).max() <= tol * fs


def _symmetrize(abi, N, nl, capacity=None):
    lib = abi.load()
    d_head, d_nn, d_nl = (torch.from_numpy(np.ascontiguousarray(x).astype(np.int32)).cuda() for x in nl)
    cap = 2 * len(nl[2]) if capacity is None else capacity
    f_head = torch.zeros(max(N, 1), dtype=torch.int32, device="cuda")
    f_nn = torch.zeros(max(N, 1), dtype=torch.int32, device="cuda")
    f_nl = torch.zeros(max(cap, 1), dtype=torch.int32, device="cuda")
    n_full = C.c_size_t()
    rc = lib.mtd_ql_symmetrize_half_list(N, abi.ptr(d_head), abi.ptr(d_nn), abi.ptr(d_nl), abi.ptr(f_head), abi.ptr(f_nn), abi.ptr(f_nl), cap,
                                         C.byref(n_full), None)
    torch.cuda.synchronize()
    return rc, n_full.value, (f_head.cpu().numpy().astype(np.uint32), f_nn.cpu().numpy().astype(np.uint32), f_nl.cpu().numpy().astype(np.uint32))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ql_half_list_symmetrized(abi, ref, dtype):
    """mtd_ql_symmetrize_half_list: the symmetric full list a half list stands for, built on the device once per list update; the
    passes in the symmetric-full-list mode (half_nlist = 2) then give the reference's HALF-list result (SteinhardtQl.cc:80,
    173-179, 328-333) — the force pass gathers, no atomics: bitwise reproducible, and independent of which end of a pair the
    half list happens to store it at"""
    pos, L = noisy_fcc(6, seed=9)
    pos = pos.astype(dtype)
    N = len(pos)
    types = np.zeros(N, dtype=np.int32)
    p64 = pos.astype(np.float64)
    half = util.build_nlist(p64, L, 1.55, half=True)
    full = util.build_nlist(p64, L, 1.55, half=False)
    rc, n_full, sym = _symmetrize(abi, N, half)
    assert rc == 0 and n_full == 2 * len(half[2]) == len(full[2])
    # exactly the full list, every particle's partners ascending
    assert np.array_equal(sym[0], full[0]) and np.array_equal(sym[1], full[1]) and np.array_equal(sym[2][:n_full], full[2])
    args = (1.4, 1.2, 6, 0, [0, 0, 0, 0, 1, 0, 1])
    g = run_gpu(abi, pos, types, L, sym, *args, dtype, half=2)
    r = run_ref(ref, p64 if dtype == np.float64 else pos.astype(np.float64), types, L, half, *args, half=True)
    assert g[0] == pytest.approx(r[0], rel=1e-10 if dtype == np.float64 else 1e-6)
    tol = 1e-9 if dtype == np.float64 else 2e-7
    scale = np.abs(r[3][:, :3]).max()
    assert scale > 0 and np.abs(g[3][:, :3] - r[3][:, :3]).max() <= tol * scale
    # the same pairs stored at their OTHER end, rows in another order: the same symmetric list, so the same bits
    rng = np.random.default_rng(4)
    i_of = np.repeat(np.arange(N), half[1])
    flip = rng.random(len(i_of)) < 0.5
    a = np.where(flip, half[2], i_of)
    b = np.where(flip, i_of, half[2])
    order = rng.permutation(len(a))
    a, b = a[order], b[order]
    order = np.argsort(a, kind="stable")
    a, b = a[order], b[order]
    nn2 = np.bincount(a, minlength=N).astype(np.uint32)
    head2 = np.zeros(N, dtype=np.uint32)
    head2[1:] = np.cumsum(nn2)[:-1]
    rc2, n2, sym2 = _symmetrize(abi, N, (head2, nn2, b.astype(np.uint32)))
    assert rc2 == 0 and np.array_equal(sym2[2][:n2], full[2])
    g2 = run_gpu(abi, pos, types, L, sym2, *args, dtype, half=2)
    assert np.array_equal(g2[3], g[3]) and g2[0] == g[0]
    # too little room: nothing written, the size needed comes back; a ghost particle in the half list: refused
    rc3, n3, _ = _symmetrize(abi, N, half, capacity=len(half[2]))
    assert rc3 == -1 and n3 == n_full
    ghost = [np.array(x).copy() for x in half]
    ghost[2][3] = N + 5
    rc4, _, _ = _symmetrize(abi, N, ghost)
    assert rc4 == -2
    # an empty list
    rc5, n5, sym5 = _symmetrize(abi, N, (np.zeros(N, dtype=np.uint32), np.zeros(N, dtype=np.uint32), np.zeros(0, dtype=np.uint32)))
    assert rc5 == 0 and n5 == 0 and not sym5[1].any()


def test_deferred_grid_pass_rides_in_the_finalize_launch(abi, ref):
    """the bias-grid engine's deferred second pass (updateReweightedEstimator's second loop + accumulate, IntegratorMetaDynamics.cc:
    1077-1087, 426-437) is announced by the generic grid launch and taken along by the next mtd_ql_accumulate on the same stream
    (extra blocks of its finalize launch) instead of a 5 us launch of its own: the grid arrays are the oracle's either way"""
    from test_gpu_metad import GpuMetad, compare
    pos, L = noisy_fcc(5, seed=3)
    N = len(pos)
    types = np.zeros(N, dtype=np.int32)
    nl = util.build_nlist(pos, L, 1.5)
    kw = dict(sigma=[0.4], cv_min=[0.0], cv_max=[80.0], num_points=[700], W=1.0, T_shift=7.0, T=1.0, stride=1, mode="well_tempered")
    g, r = GpuMetad(abi, **kw), ref.Metad(**kw)
    try:
        vals = []
        for t in range(4):
            val = run_gpu(abi, pos, types, L, nl, 1.4, 1.2, 6, 0, [0, 0, 0, 0, 1, 0, 1], np.float64)[0]      # accumulate (+ passenger), forces
            vals.append(val)
            g.step(t, [val])                                          # generic grid launch: leaves a pass pending, announces it
            compare(g, r, r.update_bias(t, [val]), label="step %d" % t)      # (get_array flushes what nobody took)
        # without a flush in between: three steps whose deferred passes can only have run as passengers
        for t in range(4, 7):
            val = run_gpu(abi, pos, types, L, nl, 1.4, 1.2, 6, 0, [0, 0, 0, 0, 1, 0, 1], np.float64)[0]
            g.abi.check(g.lib.mtd_metad_set_cv_value(g.h, 0, float(val)))
            g.abi.check(g.lib.mtd_metad_update_bias(g.h, t, None))
            b = r.update_bias(t, [val])
        compare(g, r, b, label="after three unflushed steps")
        assert vals[0] == pytest.approx(vals[-1], rel=1e-14)
    finally:
        g.close()


@pytest.mark.parametrize("dtype,stride,mode,off_grid", [(np.float64, 1, "well_tempered", False), (np.float32, 2, "standard", False),
                                                       (np.float64, 3, "well_tempered", True)])
def test_ql_merged_launch_matches_separate_launches(abi, ref, dtype, stride, mode, off_grid):
    """mtd_ql_finalize_update_bias (finalize step + scalar chain + first grid pass in ONE launch, the engine's deferred pass riding
    in the force pass) against mtd_ql_accumulate + mtd_metad_update_bias + mtd_ql_forces on a moving snapshot: every grid array,
    V, w, dV/ds, the hill count and the forces the same BITS (the grid-pass code of k_ql_finalize_chain is the twin of
    k_fused_force's), deposit and non-deposit steps, a value that leaves the grid; and the separate path against the oracle"""
    from test_gpu_metad import GpuMetad
    lib = abi.load()
    pos0, L = noisy_fcc(5, seed=3)
    N = len(pos0)
    types = np.zeros(N, dtype=np.int32)
    rcut, ron, lmax, Ql_ref = 1.4, 1.2, 6, [0, 0, 0, 0, 1, 0, 1]
    box, dt = abi.Box.make(L), abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    rng = np.random.default_rng(8)
    snaps = [(pos0 + rng.normal(0, 0.01 * k, pos0.shape)).astype(dtype) for k in range(6)]
    val0 = run_ref(ref, snaps[0].astype(np.float64), types, L, util.build_nlist(snaps[0].astype(np.float64), L, rcut + 0.15), rcut, ron, lmax, 0, Ql_ref)[0]
    lo, hi = (0.6 * val0, 1.2 * val0) if not off_grid else (1.0005 * val0, 1.4 * val0)       # off_grid: the first values lie below the grid
    kw = dict(sigma=[0.01 * val0], cv_min=[lo], cv_max=[hi], num_points=[96], W=1.3, T_shift=5.0, T=1.0, stride=stride, mode=mode)
    ql = util.dbl_array(Ql_ref)

    def run(merged):
        g = GpuMetad(abi, **kw)
        scratch = torch.zeros(lib.mtd_ql_scratch_doubles(lmax), dtype=torch.float64, device="cuda")
        force = torch.zeros((N, 4), dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda")
        out = []
        try:
            for t, p in enumerate(snaps):
                nl = util.build_nlist(p.astype(np.float64), L, rcut + 0.15)
                d_pos = torch.from_numpy(util.pack_postype(p, types, dtype)).cuda()
                d_head, d_nn, d_nl = (torch.from_numpy(x.astype(np.int32)).cuda() for x in nl)
                geo = (N, abi.ptr(d_pos), dt, C.byref(box), abi.ptr(d_head), abi.ptr(d_nn), abi.ptr(d_nl), 0, rcut, ron, lmax, 0)
                p_val = C.c_void_p()
                if merged:
                    sums, n_sums = C.c_void_p(), C.c_uint()
                    abi.check(lib.mtd_ql_accumulate_local(*geo, N, abi.ptr(scratch), C.byref(sums), C.byref(n_sums), None))
                    abi.check(lib.mtd_ql_finalize_update_bias(g.h, 0, lmax, ql, N, abi.ptr(scratch), t, C.byref(p_val), None, None, None))
                else:
                    abi.check(lib.mtd_ql_accumulate(*geo, ql, N, abi.ptr(scratch), C.byref(p_val), None, None, None))
                    abi.check(lib.mtd_metad_set_cv_source(g.h, 0, p_val.value, 1, 1, 0, 1.0, 0.0))
                    abi.check(lib.mtd_metad_update_bias(g.h, t, None))
                abi.check(lib.mtd_ql_forces(N, abi.ptr(d_pos), abi.ptr(force), dt, C.byref(box), abi.ptr(d_head), abi.ptr(d_nn), abi.ptr(d_nl), 0,
                                            rcut, ron, lmax, 0, ql, N, abi.ptr(scratch), lib.mtd_metad_bias_device(g.h), 0.0, None))
                torch.cuda.synchronize()
                st = g.state()
                out.append(dict(st=st, F=force.cpu().numpy().copy(), arrays={n: g.array(n) for n in abi.ARRAY_NAMES}))
        finally:
            g.close()
        return out

    a, b = run(True), run(False)
    for t, (x, y) in enumerate(zip(a, b)):
        for k in ("cv", "bias"):
            assert np.array_equal(x["st"][k], y["st"][k], equal_nan=True), (t, k, x["st"][k], y["st"][k])
        for k in ("V", "w", "num_gaussians", "oob"):
            assert x["st"][k] == y["st"][k] or (np.isnan(x["st"][k]) and np.isnan(y["st"][k])), (t, k, x["st"][k], y["st"][k])
        for n in abi.ARRAY_NAMES:
            assert np.array_equal(x["arrays"][n], y["arrays"][n], equal_nan=True), (t, n)
        assert np.array_equal(x["F"], y["F"]), t
    assert a[-1]["st"]["num_gaussians"] == len([t for t in range(len(snaps)) if t % stride == 0])
    if off_grid:
        assert a[0]["st"]["oob"] > 0                               # (the first values lie below the grid: V = 0 there, :677-683)
    else:
        assert abs(a[-1]["st"]["bias"][0]) > 0 and np.abs(a[-1]["F"][:, :3]).max() > 0
    # the separate path against the oracle (grid driven with the device's values)
    r = ref.Metad(**kw)
    for t, y in enumerate(b):
        bias = r.update_bias(t, y["st"]["cv"])
        assert np.allclose(y["st"]["bias"], bias, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(r.array("grid")).max())), t
        assert y["st"]["V"] == pytest.approx(r.curr_bias, rel=1e-10, abs=1e-300)


def test_ql_merged_launch_refuses_other_grids(abi):
    """more than one variable on the grid: MTD_ERR_UNSUPPORTED (the caller then finalizes and updates the grid separately)"""
    from test_gpu_metad import GpuMetad
    lib = abi.load()
    g = GpuMetad(abi, sigma=[0.1, 0.1], cv_min=[0.0, 0.0], cv_max=[1.0, 1.0], num_points=[8, 8])
    scratch = torch.zeros(lib.mtd_ql_scratch_doubles(6), dtype=torch.float64, device="cuda")
    try:
        assert lib.mtd_ql_finalize_update_bias(g.h, 0, 6, util.dbl_array([0, 0, 0, 0, 1, 0, 1]), 10, abi.ptr(scratch), 0, None, None, None, None) == -2
    finally:
        g.close()
