#!/usr/bin/env python3
"""Developer tool (GPU box): the Steinhardt force pass at config-5 size (256 000 particles, noisy fcc, l <= 6) with a full and
with a half neighbour list through the C ABI (the half-list pass adds the reaction forces as exact integers: steinhardt.hip).
usage: bench_ql_half.py [repeats]"""
import ctypes as C, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import util
from metadynamics import _abi
lib = _abi.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pos, L = util.fcc_lattice(40)
rng = np.random.default_rng(777)
pos = pos + rng.normal(0.0, 0.05, pos.shape)
N = len(pos)
types = np.zeros(N, dtype=np.int32)
box = _abi.Box.make(L)
d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float64)).cuda()
d_f = torch.zeros((N, 4), dtype=torch.float64, device="cuda")
Ql_ref = util.dbl_array([0, 0, 0, 0, 1, 0, 1])
scratch = torch.zeros(lib.mtd_ql_scratch_doubles(6), dtype=torch.float64, device="cuda")
p_val, p_ql, p_qlm = C.c_void_p(), C.c_void_p(), C.c_void_p()
d_bias = torch.tensor([-0.8], dtype=torch.float64, device="cuda")
for half in (0, 1):
    head, nn, lst = util.build_nlist(pos, L, 1.4, half=bool(half))
    d_head, d_nn, d_l = [torch.from_numpy(a.astype(np.int32)).cuda() for a in (head, nn, lst)]
    _abi.check(lib.mtd_ql_accumulate(N, d_pos.data_ptr(), _abi.MTD_F64, C.byref(box), d_head.data_ptr(), d_nn.data_ptr(), d_l.data_ptr(), half, 1.4, 1.2, 6, 0,
                                     Ql_ref, N, scratch.data_ptr(), C.byref(p_val), C.byref(p_ql), C.byref(p_qlm), None))
    def forces():
        _abi.check(lib.mtd_ql_forces(N, d_pos.data_ptr(), d_f.data_ptr(), _abi.MTD_F64, C.byref(box), d_head.data_ptr(), d_nn.data_ptr(), d_l.data_ptr(),
                                     half, 1.4, 1.2, 6, 0, Ql_ref, N, scratch.data_ptr(), d_bias.data_ptr(), 0.0, None))
    for _ in range(5):
        forces()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        forces()
    torch.cuda.synchronize()
    print("%s list (%d entries): force pass %.1f us" % ("half" if half else "full", len(lst), (time.perf_counter() - t0) / reps * 1e6))
    if half:
        # the same half list turned into the symmetric full list it stands for (once per list update), passes in mode 2
        f_head, f_nn = torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda")
        f_l = torch.zeros(2 * len(lst), dtype=torch.int32, device="cuda")
        n_full = C.c_size_t()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _abi.check(lib.mtd_ql_symmetrize_half_list(N, d_head.data_ptr(), d_nn.data_ptr(), d_l.data_ptr(), f_head.data_ptr(), f_nn.data_ptr(), f_l.data_ptr(),
                                                   2 * len(lst), C.byref(n_full), None))
        t_sym = time.perf_counter() - t0
        _abi.check(lib.mtd_ql_accumulate(N, d_pos.data_ptr(), _abi.MTD_F64, C.byref(box), f_head.data_ptr(), f_nn.data_ptr(), f_l.data_ptr(), 2, 1.4, 1.2, 6, 0,
                                         Ql_ref, N, scratch.data_ptr(), C.byref(p_val), C.byref(p_ql), C.byref(p_qlm), None))
        def forces2():
            _abi.check(lib.mtd_ql_forces(N, d_pos.data_ptr(), d_f.data_ptr(), _abi.MTD_F64, C.byref(box), f_head.data_ptr(), f_nn.data_ptr(), f_l.data_ptr(),
                                         2, 1.4, 1.2, 6, 0, Ql_ref, N, scratch.data_ptr(), d_bias.data_ptr(), 0.0, None))
        for _ in range(5):
            forces2()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            forces2()
        torch.cuda.synchronize()
        print("half list symmetrized (%d entries, built in %.0f us, once per list update): force pass %.1f us" % (n_full.value, t_sym * 1e6, (time.perf_counter() - t0) / reps * 1e6))
