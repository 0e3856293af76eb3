"""CPU: the mesh-CV oracle (OrderParameterMesh.cc restatement) against closed forms and numpy.fft."""
import numpy as np
import pytest


def _snapshot(N, L, seed=3, ntypes=2):
    rng = np.random.default_rng(seed)
    pos = rng.random((N, 3)) * L - L / 2
    types = rng.integers(0, ntypes, N)
    return pos, types


def test_mesh_assignment_conserves_charge_and_fft_matches_numpy(ref):
    """TSC weights sum to one per particle => sum(mesh) = sum_j mode_j; the oracle's DFT = numpy.fft.fftn (unnormalised)"""
    N, L = 500, 9.0
    pos, types = _snapshot(N, L)
    mode = [1.0, -0.5]
    m = ref.Mesh(12, 8, 16, mode)           # mixed: 12 is not a power of two (O(n^2) branch), 8 and 16 are
    box = ref.Box.make(L)
    m.cv(ref.as_postype(pos, types), box)
    a = np.where(types == 0, 1.0, -0.5)
    mesh = m.array("mesh")
    assert mesh.imag.max() == 0.0
    assert mesh.real.sum() == pytest.approx(a.sum(), rel=1e-12)
    assert m.mode_sq == pytest.approx((a * a).sum())
    f = np.fft.fftn(mesh.real) / N
    assert np.allclose(m.array("fourier_mesh"), f, atol=1e-13)
    inv = np.fft.ifftn(m.array("fourier_mesh_G")) * m.M      # unnormalised inverse
    assert np.allclose(m.array("inv_fourier_mesh"), inv, atol=1e-13)


def test_interpolation_function_bug_compat(ref):
    """Q6: with the reference's unsigned division I(k) = 1 where all Miller indices are >= 0 and ~0 elsewhere"""
    m = ref.Mesh(8, 8, 8, [1.0])
    box = ref.Box.make(5.0)
    m.cv(ref.as_postype([[0.1, 0.2, 0.3]], [0]), box)
    I = m.array("interpolation_f")
    assert np.all(I[:4, :4, :4] == 1.0)
    assert np.abs(I[4:, :, :]).max() < 1e-20 and np.abs(I[:, 4:, :]).max() < 1e-20 and np.abs(I[:, :, 4:]).max() < 1e-20
    m.set_bug_compat(False)
    m.cv(ref.as_postype([[0.1, 0.2, 0.3]], [0]), box)
    I2 = m.array("interpolation_f")
    kH = 2 * np.pi * np.fft.fftfreq(8)
    sinc = np.where(kH == 0, 1.0, np.sin(kH) / np.where(kH == 0, 1, kH))
    expect = (sinc[:, None, None] * sinc[None, :, None] * sinc[None, None, :]) ** 3
    assert np.allclose(I2, expect, rtol=1e-6)


def test_mesh_cv_quartic_formula(ref):
    """Q8: s = 1/2 sum_{k != 0} [ |f|^4 - I^2 (sum mode^2 / N^2) |f|^2 ], f = FFT(rho)/N"""
    N, L = 300, 7.0
    pos, types = _snapshot(N, L, seed=9)
    m = ref.Mesh(8, 8, 8, [1.0, -1.0])
    box = ref.Box.make(L)
    s = m.cv(ref.as_postype(pos, types), box)
    f = m.array("fourier_mesh")
    I = m.array("interpolation_f")
    t = np.abs(f) ** 4 - I ** 2 * m.mode_sq / N ** 2 * np.abs(f) ** 2
    t[0, 0, 0] = 0.0
    assert s == pytest.approx(0.5 * t.sum(), rel=1e-12)


@pytest.mark.parametrize("tilt", [dict(), dict(xy=0.2, xz=-0.1, yz=0.15)])
def test_mesh_force_is_minus_bias_gradient(ref, tilt):
    """force = -bias * grad s (the energy-conservation criterion of test/test_mesh.py as a direct check)"""
    N = 60
    Ls = (6.0, 7.0, 5.0)
    rng = np.random.default_rng(2)
    box = ref.Box.make(Ls, **tilt)
    f = rng.random((N, 3))
    a1 = np.array([Ls[0], 0, 0]); a2 = np.array([tilt.get("xy", 0) * Ls[1], Ls[1], 0])
    a3 = np.array([tilt.get("xz", 0) * Ls[2], tilt.get("yz", 0) * Ls[2], Ls[2]])
    lo = -0.5 * np.array(Ls)
    pos = lo + f[:, :1] * a1 + f[:, 1:2] * a2 + f[:, 2:3] * a3
    types = rng.integers(0, 2, N)
    m = ref.Mesh(8, 8, 8, [1.0, -0.7])
    bias = 1.3
    m.cv(ref.as_postype(pos, types), box)
    F = m.forces(ref.as_postype(pos, types), box, bias)
    eps = 1e-5
    for j in (0, 7, 33):
        for d in range(3):
            p1, p2 = pos.copy(), pos.copy()
            p1[j, d] -= eps
            p2[j, d] += eps
            g = (m.cv(ref.as_postype(p2, types), box) - m.cv(ref.as_postype(p1, types), box)) / (2 * eps)
            assert F[j, d] == pytest.approx(-bias * g, rel=2e-4, abs=1e-9)


def test_mesh_uniform_lattice_has_zero_cv_modes(ref):
    """one particle per cell centre, equal modes: rho is uniform => only the DC mode is populated => s = 0"""
    n, L = 8, 8.0
    g = (np.arange(n) + 0.5) / n * L - L / 2
    pos = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    m = ref.Mesh(n, n, n, [1.0])
    s = m.cv(ref.as_postype(pos, np.zeros(len(pos), dtype=int)), ref.Box.make(L))
    assert abs(s) < 1e-25
    assert np.allclose(m.array("mesh").real, 1.0)


def test_qmax_and_virial_kat(ref):
    """computeQmax / computeVirial (OrderParameterMesh.cc:1108-1179, 970-1050): a density wave along (2,0,1) puts the
    largest amplitude at that wave vector (or its mirror image, whichever cell comes first); the virial vanishes without a
    table and equals the k-space sum with K'(k) = const when a table is in use"""
    N, L = 4000, 10.0
    rng = np.random.default_rng(8)
    pos = rng.random((N, 3)) * L - L / 2
    q = 2 * np.pi / L * np.array([2, 0, 1])
    keep = rng.random(N) < 0.5 * (1 + 0.9 * np.cos(pos @ q))
    pos = pos[keep]
    n = len(pos)
    pt = np.column_stack([pos, np.zeros(n)])
    box = ref.Box.make(L)
    m = ref.Mesh(16, 16, 16, [1.0])
    m.set_bug_compat(False)
    m.cv(pt, box)
    qx, qy, qz, sq = m.qmax(n)
    # the DC bin (k = 0) holds (sum a / N)^2 = 1, the largest amplitude of all: the reference does not exclude it
    assert (qx, qy, qz) == (0.0, 0.0, 0.0) and sq == pytest.approx(n, rel=1e-12)
    # with a symmetric mode set the DC bin is empty and the modulation wins
    types = (np.arange(n) % 2).astype(np.float64)
    wave = np.cos(pos @ q) > 0
    pt2 = np.column_stack([pos, np.where(wave, 0.0, 1.0)])
    m2 = ref.Mesh(16, 16, 16, [1.0, -1.0])
    m2.set_bug_compat(False)
    m2.cv(pt2, box)
    got = np.array(m2.qmax(n)[:3])
    assert np.allclose(np.abs(got), np.abs(q), atol=1e-12) and np.allclose(got, q) | np.allclose(got, -q)
    f = m2.array("fourier_mesh")
    assert m2.qmax(n)[3] == pytest.approx((np.abs(f) ** 2).max() * n, rel=1e-12)

    assert np.all(m2.virial(n, 0.7) == 0.0)                      # no table: K' = 0 (:1015-1032)
    kk = m2.array("k")
    knorm = np.sqrt((kk ** 2).sum(-1))
    kmax = knorm.max() * 1.01
    npts = 64
    ktab = np.linspace(0.0, kmax, npts)
    m2.set_table(np.ones(npts), 0.3 * ktab, 0.0, kmax)          # K' = 0.3 k, linear: the table interpolation is exact
    m2.set_use_table(True)
    m2.cv(pt2, box)
    v = m2.virial(n, 0.7)
    a = np.abs(m2.array("fourier_mesh")) ** 2
    rhog = a * a / n / n
    w = np.where(knorm > 0, rhog * 0.3 * 0.5, 0.0)               # kfac = K'/(2k) = 0.15
    w.flat[0] = 0.0
    expect = 0.7 * np.array([(w * kk[..., 0] * kk[..., 0]).sum(), (w * kk[..., 0] * kk[..., 1]).sum(), (w * kk[..., 0] * kk[..., 2]).sum(),
                             (w * kk[..., 1] * kk[..., 1]).sum(), (w * kk[..., 1] * kk[..., 2]).sum(), (w * kk[..., 2] * kk[..., 2]).sum()])
    assert np.allclose(v, expect, rtol=1e-10, atol=1e-14 * np.abs(expect).max())
    assert v[0] > 0 and v[5] > 0
    with pytest.raises(RuntimeError):
        m2.set_table(np.ones(4), np.ones(4), 2.0, 1.0)           # kmax <= kmin (:153-158)
