"""CPU: the oracle against analytic known answers and against the reference's own standalone sources.

The reference ships no golden vectors; these closed forms plus oracle/_ref
(IndexGrid.cc, spherical_harmonics.hpp compiled from /root/reference) are what pins the oracle.
"""
import json
import os

import numpy as np
import pytest

import util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------- IndexGrid

def test_index_grid_golden(ref):
    """committed fixture generated from the reference's IndexGrid.cc (tests/golden/make_golden.py)"""
    data = json.load(open(os.path.join(GOLDEN, "index_grid.json")))
    for case in data["cases"]:
        lengths = case["lengths"]
        for idx, coords in zip(case["indices"], case["coords"]):
            assert ref.index_get(lengths, coords) == idx
            assert list(ref.index_coords(lengths, idx)) == coords


def test_index_grid_against_reference_source(ref):
    """live cross-check with oracle/_ref (only where /root/reference exists, i.e. not on the GPU box)"""
    src = ref.refsrc()
    if src is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    import ctypes as C
    rng = np.random.default_rng(0)
    for dim in (1, 2, 3, 4):
        lengths = rng.integers(2, 9, dim).astype(np.uint32)
        n = int(np.prod(lengths))
        assert src.refsrc_index_num_elements(dim, lengths.ctypes.data_as(C.POINTER(C.c_uint))) == n
        for idx in range(n):
            c = np.zeros(dim, dtype=np.uint32)
            src.refsrc_index_coords(dim, lengths.ctypes.data_as(C.POINTER(C.c_uint)), idx,
                                    c.ctypes.data_as(C.POINTER(C.c_uint)))
            assert list(ref.index_coords(lengths, idx)) == list(c)
            assert ref.index_get(lengths, c) == idx
            assert src.refsrc_index_get(dim, lengths.ctypes.data_as(C.POINTER(C.c_uint)),
                                        c.ctypes.data_as(C.POINTER(C.c_uint))) == idx


def test_first_cv_fastest(ref):
    assert ref.index_get([20, 30], [3, 0]) == 3
    assert ref.index_get([20, 30], [0, 1]) == 20
    assert list(ref.index_coords([20, 30], 20 * 7 + 5)) == [5, 7]


# ---------------------------------------------------------------- lamellar

def test_lamellar_single_particle(ref):
    """particle at the origin: every mode contributes a*cos(0) => s = n_wave * a / N_global"""
    box = ref.Box.make(10.0)
    pt = ref.as_postype([[0, 0, 0]], [1])
    assert ref.lamellar_cv([(0, 0, 3), (1, 2, 3)], pt, [0.5, 1.5], box) == pytest.approx(3.0)
    assert ref.lamellar_cv([(0, 0, 3), (1, 2, 3)], pt, [0.5, 1.5], box, n_global=4) == pytest.approx(0.75)
    F = ref.lamellar_forces([(0, 0, 3)], pt, [0.5, 1.5], box, 2.0)
    assert np.all(F == 0.0)  # sin(0)


def test_lamellar_perfect_lattice(ref):
    """perfect lamellae: a_j = sign(cos(q z_j)) on a fine uniform z-grid => s -> (2/pi) for one mode"""
    n = 4000
    L = 8.0
    z = (np.arange(n) + 0.5) / n * L - L / 2
    pos = np.stack([np.zeros(n), np.zeros(n), z], axis=1)
    q = 2 * np.pi * 2 / L
    types = (np.cos(q * z) < 0).astype(np.int32)
    box = ref.Box.make(L)
    s = ref.lamellar_cv([(0, 0, 2)], ref.as_postype(pos, types), [1.0, -1.0], box)
    assert s == pytest.approx(2 / np.pi, rel=1e-5)


def test_lamellar_matches_numpy_and_factor_two(ref):
    """value vs a direct numpy evaluation; force = bias * 2 * (-grad s) (the reference's factor 2, Q1),
    checked against a numerical gradient of the oracle's own CV"""
    rng = np.random.default_rng(1)
    N, L = 200, 7.0
    pos = rng.random((N, 3)) * L - L / 2
    types = rng.integers(0, 2, N)
    box = ref.Box.make(L)
    lat = [(0, 0, 3), (1, -1, 2), (2, 0, 0)]
    pt = ref.as_postype(pos, types)
    q = 2 * np.pi * np.array(lat) / L
    a = np.where(types == 0, 1.0, -0.5)
    assert ref.lamellar_cv(lat, pt, [1.0, -0.5], box) == pytest.approx((a[:, None] * np.cos(pos @ q.T)).sum() / N, rel=1e-12)
    bias = 0.7
    F = ref.lamellar_forces(lat, pt, [1.0, -0.5], box, bias)
    eps = 1e-6
    for j in (0, 17, 199):
        for d in range(3):
            p1, p2 = pos.copy(), pos.copy()
            p1[j, d] -= eps
            p2[j, d] += eps
            grad = (ref.lamellar_cv(lat, ref.as_postype(p2, types), [1.0, -0.5], box)
                    - ref.lamellar_cv(lat, ref.as_postype(p1, types), [1.0, -0.5], box)) / (2 * eps)
            assert F[j, d] == pytest.approx(-2.0 * bias * grad, rel=1e-5, abs=1e-9)


def test_lamellar_triclinic_reciprocal(ref):
    """q.a_i = 2 pi (h,k,l)_i for the tilted lattice: a particle displaced by a lattice vector keeps its phase"""
    box = ref.Box.make([5.0, 6.0, 7.0], xy=0.2, xz=-0.1, yz=0.3)
    a2 = np.array([0.2 * 6.0, 6.0, 0.0])
    a3 = np.array([-0.1 * 7.0, 0.3 * 7.0, 7.0])
    p = np.array([[0.3, -1.1, 2.2]])
    lat = [(1, 2, -1), (0, 3, 2)]
    m0 = ref.lamellar_fourier_modes(lat, ref.as_postype(p, [0]), [1.0], box)
    m1 = ref.lamellar_fourier_modes(lat, ref.as_postype(p + a2 - 2 * a3, [0]), [1.0], box)
    assert np.allclose(m0, m1, atol=1e-12)


# ---------------------------------------------------------------- bias grid

def test_gaussian_on_nodes_and_interpolation(ref):
    """standard metadynamics, one hill exactly on a node: grid = W exp(-d^2/2 sigma^2) on every node,
    interpolation reproduces node values, derivative = central difference of the interpolant (Q14)"""
    m = ref.Metad([0.1], [0.0], [1.0], [11], W=2.0, stride=1, mode="standard")
    b = m.update_bias(0, [0.5])
    nodes = np.linspace(0, 1, 11)
    assert np.allclose(m.array("grid"), 2.0 * np.exp(-(nodes - 0.5) ** 2 / (2 * 0.01)), rtol=1e-13)
    assert m.interpolate([0.3]) == pytest.approx(m.array("grid")[3], rel=1e-12)
    assert m.curr_bias == pytest.approx(2.0)
    assert b[0] == pytest.approx(0.0, abs=1e-12)  # symmetric hill, central difference
    assert m.array("hist")[5] == 1 and m.array("hist").sum() == 1
    assert m.array("hist_gauss")[5] == 1
    assert m.array("sigma_grid")[5] == pytest.approx(1 / 0.1)  # det(sigma^-1), diagonal 1/sigma (:177)
    assert m.num_gaussians == 1


def test_well_tempered_scale_and_reweight(ref):
    """second hill is scaled by exp(-V(s)/dT) (:374-379); reweighting: R and w evolve by exp(-(dV-<dV>)/T)"""
    kw = dict(W=1.0, T_shift=2.0, T=0.5, stride=1)
    m = ref.Metad([0.2], [0.0], [1.0], [6], mode="well_tempered", **kw)
    m.update_bias(0, [0.4])
    g1 = m.array("grid").copy()
    nodes = np.linspace(0, 1, 6)
    assert np.allclose(g1, np.exp(-(nodes - 0.4) ** 2 / 0.08))  # V = 0 before the first hill
    # after step 0: R = onehot(bin 2) * fac, w = 1/fac with <dV> = dV[bin] => fac[bin] = 1
    assert m.array("reweighted")[2] == pytest.approx(1.0)
    assert m.array("weight")[2] == pytest.approx(1.0)
    assert m.array("weight")[0] == pytest.approx(np.exp((g1[0] - g1[2]) / 0.5))
    V = m.interpolate([0.4])
    m.update_bias(1, [0.4])
    assert np.allclose(m.array("grid") - g1, np.exp(-V / 2.0) * np.exp(-(nodes - 0.4) ** 2 / 0.08))


def test_off_grid_and_edges(ref):
    m = ref.Metad([0.1], [0.0], [1.0], [11], mode="standard")
    assert m.interpolate([1.0]) == 0.0 and m.interpolate([-1e-9]) == 0.0  # s >= max or s < min -> 0 (:677-683)
    assert m.num_oob_warnings == 2
    m.update_bias(0, [1.5])  # off-grid histogram, deposit still happens (Q13); norm = 0 -> NaN (Q15)
    assert m.array("hist").sum() == 0
    assert np.isnan(m.array("weight")).all()


def test_dump_restart_round_trip(ref, tmp_path):
    """test_2d.py semantics (config 0a): N=1, V=10, density + aspect ratio on a 20 x 30 grid, well-tempered,
    stride 1, box volume x0.125 between the two steps; restarting from the step-1 dump and repeating step 2
    gives the same dump 'up to rounding errors' (test/test_2d.py:1-2)"""
    kw = dict(sigma=[0.25, 0.1], cv_min=[0.0, 0.0], cv_max=[1.0, 2.0], num_points=[20, 30], W=1.0, T_shift=1.0, T=1.0,
              stride=1, mode="well_tempered")
    names = ["cv_density_all", "cv_aspect_ratio"]
    L1 = 10 ** (1.0 / 3.0)
    box1 = ref.Box.make(L1)
    box2 = ref.Box.make(L1 * 0.125 ** (1.0 / 3.0))
    v1 = [ref.density(box1, 1), ref.aspect_ratio(box1, 0, 1)]
    v2 = [ref.density(box2, 1), ref.aspect_ratio(box2, 0, 1)]
    assert v1 == pytest.approx([0.1, 1.0]) and v2 == pytest.approx([0.8, 1.0])
    a = ref.Metad(**kw)
    # run(1): prepRun deposits at t=0 (Q17), update(0) works on t=1; run(1) again: prepRun at t=1, update at t=2
    a.update_bias(0, v1)
    a.update_bias(1, v1)
    a.write_grid(str(tmp_path / "bias.dat"), 1, names)
    a.update_bias(1, v2)
    a.update_bias(2, v2)
    a.write_grid(str(tmp_path / "bias.dat"), 2, names)
    b = ref.Metad(**kw)
    b.read_grid(str(tmp_path / "bias.dat_1"))
    b.update_bias(1, v2)
    b.update_bias(2, v2)
    b.write_grid(str(tmp_path / "bias_restart.dat"), 2, names)
    A = np.loadtxt(tmp_path / "bias.dat_2", skiprows=4)
    B = np.loadtxt(tmp_path / "bias_restart.dat_2", skiprows=4)
    assert A.shape == (600, 8)
    assert np.allclose(A, B, rtol=1e-8, atol=1e-12)
    head = open(tmp_path / "bias.dat_2").read().splitlines()[:4]
    assert head[0] == "#n_cv: 2" and head[1] == "#dim:  20 30" and head[2] == "#num_gaussians: 4"
    assert head[3].split("\t") == names + ["grid_value", "det_sigma", "num_gaussians", "hist", "hist_reweight", "weight"]


def test_umbrella(ref):
    assert ref.umbrella_bias("harmonic", 1.3, 0.5, 1.0, 10.0, 0.2, 1.0) == pytest.approx(0.5 + 10.0 * (1.3 - 1.0 - 0.1))
    assert ref.umbrella_bias("harmonic", 1.05, 0.5, 1.0, 10.0, 0.2, 1.0) == 0.5  # flat region
    assert ref.umbrella_energy("harmonic", 0.7, 1.0, 10.0, 0.2, 1.0) == pytest.approx(0.5 * 10.0 * 0.2 ** 2)
    assert ref.umbrella_bias("linear", 2.0, 0.0, 1.0, 1.0, 0.0, 3.0) == 3.0
    assert ref.umbrella_energy("wall", 1.5, 1.0, 0.5, 0.0, 2.0) == pytest.approx(2.0)
    assert ref.umbrella_bias("gaussian", 1.0, 0.0, 1.0, 0.3, 0.0, 2.0) == 0.0


def test_wte(ref):
    rng = np.random.default_rng(2)
    N, pitch = 50, 64
    nf, nt = rng.normal(size=(N, 4)), rng.normal(size=(N, 4))
    nv = rng.normal(size=(6, pitch))
    ev = rng.normal(size=6)
    assert ref.wte_potential_energy(nf, 1.25) == pytest.approx(nf[:, 3].sum() + 1.25)
    f2, t2, v2, e2 = ref.wte_scale(nf, nt, nv.reshape(-1), pitch, ev, 0.3)
    assert np.allclose(f2[:, :3], 1.3 * nf[:, :3]) and np.array_equal(f2[:, 3], nf[:, 3])
    assert np.allclose(t2, 1.3 * nt)  # CPU path scales torque.w too (Q18)
    assert np.allclose(v2.reshape(6, pitch)[:, :N], 1.3 * nv[:, :N]) and np.array_equal(v2.reshape(6, pitch)[:, N:], nv[:, N:])
    assert np.allclose(e2, 1.3 * ev)


def test_compute_sigma_kat(ref):
    """computeSigma (IntegratorMetaDynamics.cc:1205-1294): closed forms — orthogonal derivative fields give a diagonal
    sigma = sigma_g*|grad s|, sigma_inv its reciprocal; a CV without derivatives keeps its registered sigma; the
    matrix inverse agrees with numpy"""
    rng = np.random.default_rng(5)
    N = 1000
    f1 = np.zeros((N, 4)); f1[:, 0] = rng.normal(size=N)
    f2 = np.zeros((N, 4)); f2[:, 1] = rng.normal(size=N)
    sq, inv = ref.compute_sigma([f1, f2], [1, 1], [0.1, 0.2], 0.5)
    assert sq[0, 1] == 0 and sq[1, 0] == 0
    assert sq[0, 0] == pytest.approx(0.25 * (f1[:, 0] ** 2).sum(), rel=1e-12)
    assert inv[0, 0] == pytest.approx(1 / np.sqrt(sq[0, 0]), rel=1e-12) and inv[0, 1] == 0
    # correlated fields with positive products + one CV that cannot compute derivatives
    g1 = np.abs(rng.normal(size=(N, 4)))
    g2 = np.abs(rng.normal(size=(N, 4)))
    g3 = rng.normal(size=(N, 4))
    sq, inv = ref.compute_sigma([g1, g2, g3], [1, 1, 0], [0.1, 0.2, 0.3], 0.7)
    assert sq[2, 2] == pytest.approx(0.09) and sq[0, 2] == 0 and sq[2, 1] == 0
    assert sq[0, 1] == pytest.approx(0.49 * (g1[:, :3] * g2[:, :3]).sum(), rel=1e-12)
    assert np.allclose(inv, np.linalg.inv(np.sqrt(sq)), rtol=1e-10)
    # a negative off-diagonal product makes the element-wise sqrt NaN, like the reference
    sq, inv = ref.compute_sigma([f1, -f1 + f2], [1, 1], [0.1, 0.2], 1.0)
    assert sq[0, 1] < 0 and np.isnan(inv).all()


def test_wrapper_kat(ref):
    """CollectiveWrapper.cc: energy = sum force.w + external; scaling by the bias itself incl. torque.w"""
    rng = np.random.default_rng(6)
    N, pitch = 17, 20
    f, t, v = rng.normal(size=(N, 4)), rng.normal(size=(N, 4)), rng.normal(size=6 * pitch)
    assert ref.wrapper_energy(f, 1.5) == pytest.approx(f[:, 3].sum() + 1.5, rel=1e-14)
    f2, t2, v2 = ref.wrapper_scale(f, t, v, pitch, -0.3)
    assert np.allclose(f2[:, :3], -0.3 * f[:, :3]) and np.array_equal(f2[:, 3], f[:, 3])
    assert np.allclose(t2, -0.3 * t)
    vv = v.reshape(6, pitch)
    assert np.allclose(v2.reshape(6, pitch)[:, :N], -0.3 * vv[:, :N]) and np.array_equal(v2.reshape(6, pitch)[:, N:], vv[:, N:])
