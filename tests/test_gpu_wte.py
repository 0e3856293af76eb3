"""GPU parity: WellTemperedEnsemble kernels (C-ABI) vs the oracle (WellTemperedEnsemble.cc:30-68, 135-188)."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("N", [0, 1, 777, 1_000_003])
def test_energy_and_scale(abi, ref, dtype, N):
    lib = abi.load()
    rng = np.random.default_rng(N + 1)
    pitch = N + 5
    nf = rng.normal(size=(N, 4)).astype(dtype)
    nt = rng.normal(size=(N, 4)).astype(dtype)
    nv = rng.normal(size=(6, pitch)).astype(dtype)
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    d_nf, d_nt, d_nv = (torch.from_numpy(x.copy()).cuda() for x in (nf, nt, nv))
    parts = torch.zeros(lib.mtd_wte_scratch_doubles(N), dtype=torch.float64, device="cuda")
    n_part = C.c_uint()
    abi.check(lib.mtd_wte_energy_partials(N, abi.ptr(d_nf), dt, abi.ptr(parts), C.byref(n_part), None))
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    abi.check(lib.mtd_reduce_partials(abi.ptr(parts), n_part.value, 1, 1, 1.0, 3.5, abi.ptr(out), None))
    torch.cuda.synchronize()
    pe_ref = ref.wte_potential_energy(nf.astype(np.float64), 3.5)
    # the oracle sums the same (rounded) energies in double: only the summation order differs
    assert out.item() == pytest.approx(pe_ref, rel=1e-12, abs=1e-9)

    bias = 0.37
    d_bias = torch.tensor([bias], dtype=torch.float64, device="cuda")
    for use_device_bias in (True, False):
        a, b, c = d_nf.clone(), d_nt.clone(), d_nv.clone()
        abi.check(lib.mtd_wte_scale_netforce(N, abi.ptr(a), abi.ptr(b), abi.ptr(c), pitch, dt,
                                             abi.ptr(d_bias) if use_device_bias else None, bias, 1, None))
        torch.cuda.synchronize()
        f2, t2, v2, _ = ref.wte_scale(nf.astype(np.float64), nt.astype(np.float64), nv.astype(np.float64).reshape(-1), pitch,
                                      np.zeros(6), bias)
        tol = 2e-7 if dtype == np.float32 else 1e-15
        assert np.allclose(a.cpu().numpy(), f2, rtol=tol, atol=0)
        assert np.array_equal(a.cpu().numpy()[:, 3], nf[:, 3])                 # energies untouched
        assert np.allclose(b.cpu().numpy(), t2, rtol=tol, atol=0)              # torque.w scaled (CPU path, Q18)
        assert np.allclose(c.cpu().numpy().reshape(-1), v2, rtol=tol, atol=0)
        assert np.array_equal(c.cpu().numpy()[:, N:], nv[:, N:])               # padding beyond N untouched
    # reference GPU behaviour (torque.w not scaled) is available too
    b = d_nt.clone()
    abi.check(lib.mtd_wte_scale_netforce(N, abi.ptr(d_nf.clone()), abi.ptr(b), None, pitch, dt, None, bias, 0, None))
    torch.cuda.synchronize()
    assert np.array_equal(b.cpu().numpy()[:, 3], nt[:, 3])


def test_wte_idempotence_and_linearity(abi):
    """size-independent properties at 10^6 particles: scaling by (1+b1) then (1+b2) == scaling once by the product"""
    lib = abi.load()
    N = 1_000_000
    x = torch.randn((N, 4), dtype=torch.float64, device="cuda")
    a, b = x.clone(), x.clone()
    for bias in (0.25, -0.4):
        abi.check(lib.mtd_wte_scale_netforce(N, abi.ptr(a), None, None, N, abi.MTD_F64, None, bias, 0, None))
    abi.check(lib.mtd_wte_scale_netforce(N, abi.ptr(b), None, None, N, abi.MTD_F64, None, 1.25 * 0.6 - 1.0, 0, None))
    torch.cuda.synchronize()
    assert torch.allclose(a, b, rtol=1e-14, atol=0)
