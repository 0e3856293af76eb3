"""GPU parity: CollectiveWrapper scale kernel and the adaptive-Gaussian reduction (computeSigma) through the C-ABI
vs the oracle (CollectiveWrapper.cc:136-179, IntegratorMetaDynamics.cc:1205-1294)."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("N", [1, 777, 300_007])
def test_wrapper_scale(abi, ref, dtype, N):
    lib = abi.load()
    rng = np.random.default_rng(N)
    pitch = N
    f, t, v = (rng.normal(size=s).astype(dtype) for s in ((N, 4), (N, 4), (6, pitch)))
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    a, b, c = (torch.from_numpy(x.copy()).cuda() for x in (f, t, v))
    d_bias = torch.tensor([-0.42], dtype=torch.float64, device="cuda")
    abi.check(lib.mtd_wrapper_scale_forces(N, abi.ptr(a), abi.ptr(b), abi.ptr(c), pitch, dt, abi.ptr(d_bias), 0.0, 1, None))
    torch.cuda.synchronize()
    f2, t2, v2 = ref.wrapper_scale(f.astype(np.float64), t.astype(np.float64), v.astype(np.float64).reshape(-1), pitch, -0.42)
    tol = 2e-7 if dtype == np.float32 else 1e-15
    assert np.allclose(a.cpu().numpy(), f2, rtol=tol, atol=0)
    assert np.array_equal(a.cpu().numpy()[:, 3], f[:, 3])
    assert np.allclose(b.cpu().numpy(), t2, rtol=tol, atol=0)
    assert np.allclose(c.cpu().numpy().reshape(-1), v2, rtol=tol, atol=0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n_cv,N", [(1, 1000), (2, 100_003), (3, 1_000_000), (6, 20_011)])
def test_sigma_products_and_inverse(abi, ref, dtype, n_cv, N):
    lib = abi.load()
    rng = np.random.default_rng(n_cv * 1000 + 7)
    # positive-ish correlated fields so the element-wise sqrt stays real (a common base field plus noise)
    base = np.abs(rng.normal(size=(N, 4)))
    forces = [(base + 0.5 * np.abs(rng.normal(size=(N, 4)))).astype(dtype) for _ in range(n_cv)]
    can = [1] * n_cv
    if n_cv >= 3:
        can[1] = 0                                     # a box CV in the middle: no derivatives
    sigma = [0.1 * (c + 1) for c in range(n_cv)]
    sigma_g = 0.3
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    d_forces = [torch.from_numpy(f).cuda() for f in forces]
    ptrs = (C.c_void_p * n_cv)(*[d_forces[c].data_ptr() if can[c] else None for c in range(n_cv)])
    scratch = torch.zeros(lib.mtd_sigma_scratch_doubles(), dtype=torch.float64, device="cuda")
    sq = np.zeros(n_cv * n_cv)
    abi.check(lib.mtd_sigma_products(n_cv, ptrs, N, dt, sigma_g, abi.ptr(scratch),
                                     sq.ctypes.data_as(C.POINTER(C.c_double)), None))
    for c in range(n_cv):
        if not can[c]:
            assert sq[c * n_cv + c] == 0.0
            sq[c * n_cv + c] = sigma[c] ** 2          # the caller's part (:1249)
    inv = np.zeros(n_cv * n_cv)
    abi.check(lib.mtd_sigma_inverse(n_cv, sq.ctypes.data_as(C.POINTER(C.c_double)), inv.ctypes.data_as(C.POINTER(C.c_double))))
    sq_ref, inv_ref = ref.compute_sigma([f.astype(np.float64) for f in forces], can, sigma, sigma_g)
    assert np.allclose(sq.reshape(n_cv, n_cv), sq_ref, rtol=1e-11, atol=0)
    # sqrt(sigmasq) of strongly correlated fields is ill-conditioned: compare through the condition number
    cond = np.linalg.cond(np.sqrt(sq_ref))
    assert np.allclose(inv.reshape(n_cv, n_cv), inv_ref, rtol=1e-10 * cond, atol=1e-12 * cond * np.abs(inv_ref).max())
    # reproducible run to run (fixed summation order, no atomics)
    sq2 = np.zeros(n_cv * n_cv)
    abi.check(lib.mtd_sigma_products(n_cv, ptrs, N, dt, sigma_g, abi.ptr(scratch),
                                     sq2.ctypes.data_as(C.POINTER(C.c_double)), None))
    for c in range(n_cv):
        if not can[c]:
            sq2[c * n_cv + c] = sigma[c] ** 2
    assert np.array_equal(sq, sq2)


def test_sigma_nan_on_negative_product(abi):
    """anticorrelated derivative fields: sqrt of a negative product is NaN in the reference, and so here"""
    lib = abi.load()
    N = 1000
    f1 = torch.randn((N, 4), dtype=torch.float64, device="cuda")
    f2 = -f1
    ptrs = (C.c_void_p * 2)(f1.data_ptr(), f2.data_ptr())
    scratch = torch.zeros(lib.mtd_sigma_scratch_doubles(), dtype=torch.float64, device="cuda")
    sq, inv = np.zeros(4), np.zeros(4)
    abi.check(lib.mtd_sigma_products(2, ptrs, N, abi.MTD_F64, 1.0, abi.ptr(scratch), sq.ctypes.data_as(C.POINTER(C.c_double)), None))
    assert sq[1] < 0 and sq[1] == sq[2]
    abi.check(lib.mtd_sigma_inverse(2, sq.ctypes.data_as(C.POINTER(C.c_double)), inv.ctypes.data_as(C.POINTER(C.c_double))))
    assert np.isnan(inv).all()
