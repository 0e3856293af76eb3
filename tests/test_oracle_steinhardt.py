"""CPU: Steinhardt oracle — the spherical harmonics are pinned against the reference's own header (golden fixture
generated from /root/reference/metadynamics/spherical_harmonics.hpp via oracle/_ref), SteinhardtQl.cc by closed forms."""
import json
import os

import numpy as np
import pytest
from scipy.special import sph_harm_y

import util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sph_golden_from_reference_header(ref):
    d = json.load(open(os.path.join(GOLDEN, "sph_lmax6.json")))
    want = np.array(d["values"])
    got = ref.sph_evaluate(d["lmax"], d["polar"], d["azimuth"], full_m=True)
    assert np.allclose(got.real, want[..., 0], rtol=1e-13, atol=1e-15)
    assert np.allclose(got.imag, want[..., 1], rtol=1e-13, atol=1e-15)


def test_sph_live_against_reference_header(ref):
    src = ref.refsrc()
    if src is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    import ctypes as C
    rng = np.random.default_rng(1)
    n, lmax = 50, 8
    polar, azim = rng.uniform(0.01, np.pi - 0.01, n), rng.uniform(-np.pi, np.pi, n)
    out = np.zeros(2 * (lmax + 1) ** 2 * n)
    DP = C.POINTER(C.c_double)
    src.refsrc_evaluate_sph(out.ctypes.data_as(DP), lmax, polar.ctypes.data_as(DP), azim.ctypes.data_as(DP), n, 1)
    want = out.reshape(n, -1, 2)
    got = ref.sph_evaluate(lmax, polar, azim)
    assert np.allclose(got.real, want[..., 0], rtol=1e-13, atol=1e-15) and np.allclose(got.imag, want[..., 1], rtol=1e-13, atol=1e-15)


def test_sph_matches_scipy_up_to_condon_shortley(ref):
    """fsph has no Condon-Shortley phase; SteinhardtQl.cc:150-153 adds (-1)^m for odd m > 0 (Q21)"""
    rng = np.random.default_rng(2)
    polar, azim = rng.uniform(0.1, 3.0, 10), rng.uniform(-3, 3, 10)
    Y = ref.sph_evaluate(6, polar, azim)
    n = 0
    for l in range(7):
        for p in range(2 * l + 1):
            m = p if p <= l else l - p
            sp = sph_harm_y(l, m, polar, azim)
            if m >= 0:
                assert np.allclose(Y[:, n] * (-1) ** m, sp, atol=1e-12)
            else:
                # fsph's negative-m entry is the plain conjugate of +|m| (no (-1)^m)
                assert np.allclose(Y[:, n], np.conj(sph_harm_y(l, -m, polar, azim)) * (-1) ** m, atol=1e-12)
            n += 1


def test_fcc_known_values(ref):
    """ideal fcc, full neighbour list of the 12 nearest neighbours: Q_l(code) = 144 Q_l(Steinhardt)^2 (Q19);
    Q6 = 0.57452, Q4 = 0.19094, odd l vanish"""
    pos, L = util.fcc_lattice(4)
    N = len(pos)
    head, nn, nl = util.build_nlist(pos, L, 1.1)
    assert np.all(nn == 12)
    box = ref.Box.make(L)
    pt = ref.as_postype(pos, np.zeros(N, dtype=int))
    val, Qlm, Ql = ref.ql_compute_cv(pt, box, head, nn, nl, 1.4, 1.2, 6, 0, [0, 0, 0, 0, 1, 0, 1])
    assert Ql[6] == pytest.approx(144 * 0.57452 ** 2, rel=1e-4)
    assert Ql[4] == pytest.approx(144 * 0.19094 ** 2, rel=1e-4)
    assert abs(Ql[1]) < 1e-20 and abs(Ql[3]) < 1e-20 and abs(Ql[5]) < 1e-20
    assert Ql[0] == pytest.approx(144.0, rel=1e-12)          # Y00 = 1/sqrt(4 pi): 4 pi |12 N Y00|^2 / N^2 = 144
    assert val == pytest.approx(Ql[4] + Ql[6])
    # half neighbour list: even l doubled, odd l zeroed (:173-179) => same Q_l
    h2, n2, l2 = util.build_nlist(pos, L, 1.1, half=True)
    val2, _, Ql2 = ref.ql_compute_cv(pt, box, h2, n2, l2, 1.4, 1.2, 6, 0, [0, 0, 0, 0, 1, 0, 1], half=True)
    assert np.allclose(Ql2, Ql, rtol=1e-12, atol=1e-20)


def test_force_is_minus_bias_gradient_of_cv_central_terms(ref):
    """SteinhardtQl's force on i keeps only the pair terms in which i is the central particle (Q20): it equals
    -bias * d/dr_i of  sum_l Ql_ref[l] 4pi/(2l+1)/N^2 * 2 Re( conj(Q_lm) * sum_{j in nb(i)} f Y_lm(r_i - r_j) )
    with Q_lm held fixed — checked by finite differences of exactly that function"""
    rng = np.random.default_rng(3)
    pos, L = util.fcc_lattice(3)
    pos = pos + rng.normal(0, 0.05, pos.shape)
    N = len(pos)
    types = np.zeros(N, dtype=int)
    rcut, ron, lmax = 1.4, 1.2, 6
    Ql_ref = [0.3, 0, 0.5, 0, 1.0, 0, 1.0]
    head, nn, nl = util.build_nlist(pos, L, rcut + 0.2)
    box = ref.Box.make(L)
    pt = ref.as_postype(pos, types)
    val, Qlm, Ql = ref.ql_compute_cv(pt, box, head, nn, nl, rcut, ron, lmax, 0, Ql_ref)
    bias = 0.8
    F = ref.ql_compute_forces(pt, box, head, nn, nl, rcut, ron, lmax, 0, Ql_ref, Qlm, bias)

    def central(i, p):
        """sum_l Ql_ref 4pi/(2l+1)/N^2 sum_m 2 Re(conj(Qlm) q_lm(i)), q_lm(i) = sum_j f Y (with the code's phases)"""
        # evaluate through the oracle itself on a 1-centre neighbour list
        hl = np.zeros(N, dtype=np.uint32); n1 = np.zeros(N, dtype=np.uint32)
        n1[i] = nn[i]; hl[i] = 0
        lst = nl[head[i]:head[i] + nn[i]]
        _, q, _ = ref.ql_compute_cv(ref.as_postype(p, types), box, hl, n1, lst, rcut, ron, lmax, 0, Ql_ref)
        tot, n = 0.0, 0
        for l in range(lmax + 1):
            for _ in range(2 * l + 1):
                tot += Ql_ref[l] * 4 * np.pi / (2 * l + 1) / N ** 2 * 2 * np.real(np.conj(Qlm[n]) * q[n])
                n += 1
        return tot

    eps = 1e-6
    for i in (0, 5, 40):
        for d in range(3):
            p1, p2 = pos.copy(), pos.copy()
            p1[i, d] -= eps; p2[i, d] += eps
            g = (central(i, p2) - central(i, p1)) / (2 * eps)
            assert F[i, d] == pytest.approx(-bias * g, rel=1e-5, abs=1e-10)
