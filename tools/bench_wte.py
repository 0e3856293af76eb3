#!/usr/bin/env python3
"""Developer tool: the WellTemperedEnsemble kernels and the adaptive-Gaussian products (N2) at 10^6
particles: energy partial sums + reduce, and the in-place scaling of net force / torque / virial by (1 + bias).
Algorithmic bytes (fp32): energy pass 16 B/particle read; scale pass 2 x (16 + 16 + 24) = 112 B/particle."""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
from metadynamics import _abi
lib = _abi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for name, tdt, dt, S in (("f32", torch.float32, _abi.MTD_F32, 4), ("f64", torch.float64, _abi.MTD_F64, 8)):
    nf = torch.randn((N, 4), dtype=tdt, device="cuda")
    nt = torch.randn((N, 4), dtype=tdt, device="cuda")
    nv = torch.randn((6, N), dtype=tdt, device="cuda")
    parts = torch.zeros(lib.mtd_wte_scratch_doubles(N), dtype=torch.float64, device="cuda")
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    bias = torch.tensor([1e-9], dtype=torch.float64, device="cuda")
    n_part = C.c_uint()

    def energy():
        _abi.check(lib.mtd_wte_energy_partials(N, nf.data_ptr(), dt, parts.data_ptr(), C.byref(n_part), None))
        _abi.check(lib.mtd_reduce_partials(parts.data_ptr(), n_part.value, 1, 1, 1.0, 0.0, out.data_ptr(), None))

    def scale():
        _abi.check(lib.mtd_wte_scale_netforce(N, nf.data_ptr(), nt.data_ptr(), nv.data_ptr(), N, dt, bias.data_ptr(), 0.0, 0, None))

    for fn, label, nbytes in ((energy, "energy (partials + reduce)", N * 4 * S), (scale, "scale net force/torque/virial", N * 2 * 14 * S)):
        for _ in range(20):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 200
        print("WTE %s %-32s %7.1f us  %6.0f GB/s of algorithmic traffic (%d B/particle)" % (name, label, us, nbytes / us / 1e3, nbytes // N))
