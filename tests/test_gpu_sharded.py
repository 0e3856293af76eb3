"""GPU: the particle-sharded and multiple-walker product backends (metadynamics.sharded.HipCvSetBackend) with two
shards / walkers emulated in one process: the exchange buffers of the two backends are summed by hand exactly where
ShardedBiasStep / WalkerBiasStep all-reduce them.  Checked against the single-shard run and the oracle."""
import numpy as np
import pytest

import util

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

KW = dict(W=1.0, T_shift=7.0, T=1.0, stride=1, mode="well_tempered")


def _allreduce_by_hand(bufs_a, bufs_b):
    for a, b in zip(bufs_a, bufs_b):
        tot = a + b
        a.copy_(tot)
        b.copy_(tot)


def _dev_postype(pos, types, dtype):
    return torch.from_numpy(util.pack_postype(pos.astype(dtype), types, dtype)).cuda()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_two_shards_lamellar_mesh_energy(abi, ref, dtype):
    """config-3-like CV set (lamellar + mesh) plus a wrapped energy, particles split over two shards"""
    from metadynamics import sharded
    N, L = 20000, 20.0
    pos, types = util.snapshot_random(N, L, seed=31, modulated=True, dtype=np.float64)
    lv, mode = [(0, 0, 3), (0, 3, 0)], [1.0, -1.0]
    rbox = ref.Box.make(L)
    opt = util.oracle_postype(pos.astype(dtype).astype(np.float64), types)
    s_lam = ref.lamellar_cv(lv, opt, mode, rbox)
    mesh_ref = ref.Mesh(16, 16, 16, mode)
    s_mesh = mesh_ref.cv(opt, rbox)
    rng = np.random.default_rng(4)
    frc = rng.normal(size=(N, 4)).astype(dtype)
    frc[:, 3] = rng.normal(-0.5, 0.1, N)
    e_tot = float(frc[:, 3].astype(np.float64).sum()) + 2.5 + 1.5
    grid = dict(sigma=[0.02, 0.05 * abs(s_mesh), 5.0], cv_min=[-0.6, 0.25 * s_mesh, e_tot - 40.0], cv_max=[0.4, 1.6 * s_mesh, e_tot + 55.0],
                num_points=[24, 20, 16])

    def make(sl, ext):
        dp = _dev_postype(pos[sl], types[sl], dtype)
        f = torch.from_numpy(frc[sl].copy()).cuda()
        n = f.shape[0]
        parts = [sharded.LamellarPart(lv, mode, dp, N, L), sharded.MeshPart(16, 16, 16, mode, dp, N, L),
                 sharded.EnergyPart(f, torch.zeros_like(f), torch.zeros((6, n), dtype=f.dtype, device="cuda"), n, ext, wrapper=True)]
        return sharded.HipCvSetBackend(parts, grid, **KW), f

    single, f_s = make(slice(0, N), 4.0)
    a, f_a = make(slice(0, N // 2 + 37), 2.5)
    b, f_b = make(slice(N // 2 + 37, N), 1.5)
    g = ref.Metad(grid["sigma"], grid["cv_min"], grid["cv_max"], grid["num_points"], **dict(KW))
    for t in range(3):
        single.force_pass(single.cv_pass(), t)
        ba, bb = a.cv_pass(), b.cv_pass()
        _allreduce_by_hand(ba, bb)
        a.force_pass(ba, t)
        b.force_pass(bb, t)
        torch.cuda.synchronize()
        st_s, st_a, st_b = single.state(), a.state(), b.state()
        assert st_a["cv"] == st_b["cv"] and st_a["bias"] == st_b["bias"]            # replicated, bit for bit
        # lamellar: fp32 per-thread partial sums, so the summation order (= the sharding) shows at 1e-9; mesh and energy
        # are double throughout
        assert st_a["cv"][0] == pytest.approx(st_s["cv"][0], rel=1e-7)
        assert np.allclose(st_a["cv"][1:], st_s["cv"][1:], rtol=1e-11, atol=1e-14)
        assert np.allclose(st_a["bias"], st_s["bias"], rtol=1e-5, atol=1e-7 * np.abs(st_s["bias"]).max())
        bias_ref = g.update_bias(t, st_s["cv"])
        assert np.allclose(st_s["bias"], bias_ref, rtol=1e-7, atol=1e-9 * np.abs(bias_ref).max())
    assert st_s["cv"][0] == pytest.approx(s_lam, rel=1e-6)
    assert st_s["cv"][1] == pytest.approx(s_mesh, rel=1e-9)
    assert st_s["cv"][2] == pytest.approx(e_tot, rel=1e-12)
    assert np.array_equal(a.grid_array(0), b.grid_array(0))
    assert np.allclose(a.grid_array(0), single.grid_array(0), rtol=1e-6, atol=1e-9)
    cut = N // 2 + 37
    for p in range(2):
        fs = single.parts[p].force.cpu().numpy().astype(np.float64)
        fa, fb = a.parts[p].force.cpu().numpy(), b.parts[p].force.cpu().numpy()
        scale = np.abs(fs[:, :3]).max()
        assert scale > 0
        # a force is (bias factor) x (a sum over the particle's own terms): the bias factors of the sharded and the single run
        # agree to 1e-5 only (asserted above: a finite difference of the grid amplifies the 1e-9 of the fp32 partial sums), so
        # the forces are compared PER UNIT BIAS FACTOR — what is left is the rounding of the fp32 weight a_type * bias * 2 / N the
        # lamellar kernels scale with (6e-8) and of the stored components
        ba, bs = st_a["bias"][p], st_s["bias"][p]
        tol = (1e-6 if dtype == np.float32 else 3e-7) * scale / abs(bs)
        assert np.abs(fa[:, :3] / ba - fs[:cut, :3] / bs).max() <= tol and np.abs(fb[:, :3] / ba - fs[cut:, :3] / bs).max() <= tol
        assert np.abs(fa[:, :3] - fs[:cut, :3]).max() <= 2e-5 * scale and np.abs(fb[:, :3] - fs[cut:, :3]).max() <= 2e-5 * scale
    # the wrapped arrays were scaled by the same bias factor three times on every shard
    assert np.allclose(torch.cat([f_a, f_b]).cpu().numpy(), f_s.cpu().numpy(), rtol=1e-5 if dtype == np.float32 else 1e-7)
    for be in (single, a, b):
        be.close()


@pytest.mark.parametrize("reduce_locally", [False, True])
def test_two_shards_fused_lamellar(abi, ref, reduce_locally):
    """the fused two-launch step sharded over two backends (what bench.py --gpus N runs): launch A per shard, the block
    partial sums (or the n_cv local sums) summed across the shards, launch B per shard with N_global"""
    from metadynamics import sharded
    N, L = 30000, 24.0
    pos, types = util.snapshot_random(N, L, seed=41, modulated=True, dtype=np.float32)
    cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]
    grid = dict(sigma=[0.05, 0.05], cv_min=[-1.0, -1.0], cv_max=[1.0, 1.0], num_points=[64, 48])
    cut = N // 2 if not reduce_locally else 17000          # exchanging block partials needs equal shards

    def make(sl):
        dp = _dev_postype(pos[sl], types[sl], np.float32)
        return sharded.HipLamellarBackend(cvs, dp, N, L, grid, 1.0, 7.0, 1.0, 1, "well_tempered", fast_trig=True, fused=True)

    single, a, b = make(slice(0, N)), make(slice(0, cut)), make(slice(cut, N))
    rbox = ref.Box.make(L)
    opt = util.oracle_postype(pos, types)
    s_ref = [ref.lamellar_cv(v, opt, m, rbox) for v, m in cvs]
    g = ref.Metad(grid["sigma"], grid["cv_min"], grid["cv_max"], grid["num_points"], **dict(KW))
    for t in range(4):
        single.step_single(t)
        sa, sb = a.cv_pass(reduce_locally), b.cv_pass(reduce_locally)
        assert sa.numel() == sb.numel()                       # same block count on every shard (equal launch geometry)
        _allreduce_by_hand([sa], [sb])
        a.force_pass(sa, t)
        b.force_pass(sb, t)
        bias = g.update_bias(t, s_ref)
    torch.cuda.synchronize()
    st, sta, stb = single.state(), a.state(), b.state()
    assert sta["cv"] == stb["cv"] and sta["bias"] == stb["bias"]
    assert np.allclose(sta["cv"], s_ref, rtol=1e-6) and np.allclose(st["cv"], s_ref, rtol=1e-6)
    assert np.allclose(sta["bias"], bias, rtol=1e-4, atol=1e-6 * np.abs(bias).max())
    assert sta["num_gaussians"] == st["num_gaussians"] == 4
    for c in range(2):
        fs = single.forces[c].cpu().numpy()
        f2 = torch.cat([a.forces[c], b.forces[c]]).cpu().numpy()
        assert np.abs(f2[:, :3] - fs[:, :3]).max() <= 2e-5 * np.abs(fs[:, :3]).max()
    for be in (single, a, b):
        be.close()


def test_two_shards_steinhardt_with_ghosts(abi, ref):
    """Q_l over two spatial slabs: each shard stores its own particles first and the other slab's particles as ghosts;
    the exchange is the Q'_lm sums"""
    from metadynamics import sharded
    half = False           # third-law lists drop the reaction on ghosts (SteinhardtQl.cc:328): full lists when sharded
    pos, L = util.fcc_lattice(6)
    rng = np.random.default_rng(9)
    pos = pos + rng.normal(0, 0.04, pos.shape)
    N = len(pos)
    types = np.zeros(N, dtype=np.int32)
    Ql_ref = [0, 0, 0, 0, 1, 0, 1]
    rcut, ron, lmax = 1.4, 1.2, 6
    head, nn, nl = util.build_nlist(pos, L, 1.5, half=half)
    rbox = ref.Box.make(L)
    pt = util.oracle_postype(pos, types)
    val, Qlm, Ql = ref.ql_compute_cv(pt, rbox, head, nn, nl, rcut, ron, lmax, 0, Ql_ref, half=half)
    grid = dict(sigma=[0.02 * val], cv_min=[0.55 * val], cv_max=[1.3 * val], num_points=[64])

    def shard(ids):
        other = np.setdiff1d(np.arange(N), ids)
        perm = np.concatenate([ids, other])
        inv = np.empty(N, dtype=np.int64)
        inv[perm] = np.arange(N)
        h2, n2, l2 = [], [], []
        for i in ids:
            h2.append(len(l2))
            n2.append(nn[i])
            l2.extend(inv[nl[head[i]:head[i] + nn[i]]])
        lists = tuple(torch.from_numpy(np.asarray(x, dtype=np.uint32).view(np.int32)).cuda() for x in (h2, n2, l2 if l2 else [0]))
        dp = _dev_postype(pos[perm], types[perm], np.float64)
        part = sharded.SteinhardtPart(rcut, ron, lmax, Ql_ref, 0, dp, len(ids), lists, N, L, half=half)
        return sharded.HipCvSetBackend([part], grid, **KW), perm

    left = np.where(pos[:, 0] < 0)[0]
    right = np.where(pos[:, 0] >= 0)[0]
    a, perm_a = shard(left)
    b, perm_b = shard(right)
    g = ref.Metad(grid["sigma"], grid["cv_min"], grid["cv_max"], grid["num_points"], **dict(KW))
    for t in range(3):
        ba, bb = a.cv_pass(), b.cv_pass()
        _allreduce_by_hand(ba, bb)
        a.force_pass(ba, t)
        b.force_pass(bb, t)
        bias = g.update_bias(t, [val])
    torch.cuda.synchronize()
    sa, sb = a.state(), b.state()
    assert sa["cv"] == sb["cv"] and sa["bias"] == sb["bias"]
    assert sa["cv"][0] == pytest.approx(val, rel=1e-10)
    assert sa["bias"][0] == pytest.approx(bias[0], rel=1e-7)
    F_ref = ref.ql_compute_forces(pt, rbox, head, nn, nl, rcut, ron, lmax, 0, Ql_ref, Qlm, bias[0], half=half)
    F = np.zeros((N, 4))
    for be, perm, ids in ((a, perm_a, left), (b, perm_b, right)):
        f = be.parts[0].force.cpu().numpy()
        assert len(f) == len(ids)
        F[perm[:len(f)]] = f
    assert np.abs(F[:, :3] - F_ref[:, :3]).max() <= 1e-7 * np.abs(F_ref[:, :3]).max()
    with pytest.raises(ValueError):
        sharded.SteinhardtPart(rcut, ron, lmax, Ql_ref, 0, a.parts[0].d_pos, len(left), (a.parts[0].head, a.parts[0].nneigh, a.parts[0].nlist),
                               N, L, half=True)
    a.close()
    b.close()


def test_two_walkers_share_the_grid(abi, ref):
    """multiple walkers: phase A on each walker, the packed delta arrays summed, phase B — vs two oracle engines whose
    delta arrays are summed the same way (IntegratorMetaDynamics.cc:393-409)"""
    from metadynamics import sharded
    N, L = 6000, 12.0
    lv1, lv2, mode = [(0, 0, 3), (0, 3, 0), (1, 1, 1)], [(2, 0, 0), (0, 0, 6)], [1.0, -1.0]
    grid = dict(sigma=[0.05, 0.05], cv_min=[-1.0, -1.0], cv_max=[1.0, 1.0], num_points=[48, 40])
    rbox = ref.Box.make(L)

    def snapshot(w, t):
        pos, types = util.snapshot_random(N, L, seed=5 + w, modulated=False, dtype=np.float64)
        amp = np.where(types == 0, 1.0, -1.0)
        pos[:, 2] += (0.3 + 0.1 * w + 0.05 * t) * amp * np.sin(2 * np.pi * 3 * pos[:, 2] / L)
        return pos, types

    walkers, d_pos = [], []
    for w in range(2):
        pos, types = snapshot(w, 0)
        dp = _dev_postype(pos, types, np.float64)
        d_pos.append(dp)
        walkers.append(sharded.HipCvSetBackend([sharded.LamellarPart(lv1, mode, dp, N, L), sharded.LamellarPart(lv2, mode, dp, N, L)],
                                               grid, **dict(KW, stride=2)))
    refs = [ref.Metad(grid["sigma"], grid["cv_min"], grid["cv_max"], grid["num_points"], **dict(KW, stride=2)) for _ in range(2)]
    names = ("grid_delta", "sigma_grid_delta", "hist_delta", "hist_gauss_delta")
    for t in range(5):
        vals = []
        for w in range(2):
            pos, types = snapshot(w, t)
            d_pos[w].copy_(_dev_postype(pos, types, np.float64))
            opt = util.oracle_postype(pos, types)
            vals.append([ref.lamellar_cv(lv1, opt, mode, rbox), ref.lamellar_cv(lv2, opt, mode, rbox)])
        deps = [wk.phase_a(t) for wk in walkers]
        assert deps[0] == deps[1] == (t % 2 == 0)
        if deps[0]:
            bufs = [wk.delta_buffers() for wk in walkers]
            _allreduce_by_hand(bufs[0], bufs[1])
        for wk, d in zip(walkers, deps):
            wk.phase_b(d)
        # oracle: same flow
        rdeps = [r.phase_a(t, v) for r, v in zip(refs, vals)]
        if rdeps[0]:
            for name in names:
                tot = refs[0].array(name) + refs[1].array(name)
                for r in refs:
                    r.array(name)[:] = tot
        rb = [r.phase_b(int(d), v) for r, d, v in zip(refs, rdeps, vals)]
        torch.cuda.synchronize()
        for wk, r, v, b in zip(walkers, refs, vals, rb):
            st = wk.state()
            assert np.allclose(st["cv"], v, rtol=1e-6)
            assert np.allclose(st["bias"], b, rtol=2e-5, atol=1e-6 * np.abs(b).max())
    g0, g1 = walkers[0].grid_array(0), walkers[1].grid_array(0)
    assert np.array_equal(g0, g1) and np.abs(g0).max() > 0
    assert np.allclose(g0, refs[0].array("grid"), rtol=1e-4, atol=1e-6 * np.abs(g0).max())
    assert np.array_equal(walkers[0].grid_array(6), refs[0].array("hist"))
    assert walkers[0].state()["num_gaussians"] == refs[0].num_gaussians == 3
    for wk in walkers:
        wk.close()
