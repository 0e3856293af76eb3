#!/usr/bin/env python3
"""Developer tool (GPU box): the slab-decomposed mesh path (mtd_mesh_slab_*) with ONE rank (all pulls local: the slab layouts,
the pencil layout, the gathered z pass, barriers) against the whole-mesh path over random mesh sizes / boxes / particles; the
multi-rank indexing is covered by tests/test_gpu_comm.py.  usage: fuzz_slab.py [seconds] [seed]"""
import ctypes as C, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import util
from metadynamics import _abi, xgmi
from test_gpu_mesh import GpuMesh

lib = _abi.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DIMS = [4, 5, 6, 8, 9, 12, 16, 20, 24, 32, 48]
t0, t_print, it, worst = time.time(), time.time(), 0, dict(cv=0.0, f=0.0)
while time.time() - t0 < budget:
    it += 1
    if time.time() - t_print > 30.0:
        t_print = time.time()
        print("fuzz_slab: %d cases so far" % it, flush=True)
    dims = tuple(int(rng.choice(DIMS)) for _ in range(3))
    N = int(rng.choice([0, 1, 37, 500, 4000]))
    Ls = tuple(float(x) for x in rng.uniform(3.0, 15.0, 3))
    tilt = dict(xy=float(rng.uniform(-0.3, 0.3)), xz=float(rng.uniform(-0.3, 0.3)), yz=float(rng.uniform(-0.3, 0.3))) if rng.random() < 0.5 else {}
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    mode = [float(x) for x in rng.uniform(-1.5, 1.5, 2)]
    f = rng.random((N, 3))
    a1 = np.array([Ls[0], 0, 0]); a2 = np.array([tilt.get("xy", 0) * Ls[1], Ls[1], 0])
    a3 = np.array([tilt.get("xz", 0) * Ls[2], tilt.get("yz", 0) * Ls[2], Ls[2]])
    pos = (-0.5 * np.array(Ls) + f[:, :1] * a1 + f[:, 1:2] * a2 + f[:, 2:3] * a3).astype(dtype)
    types = rng.integers(0, 2, N).astype(np.int32)
    box = _abi.Box.make(Ls, **tilt)
    dt = _abi.MTD_F32 if dtype == np.float32 else _abi.MTD_F64
    d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda()
    n_global = max(N, 1)
    whole = GpuMesh(_abi, dims, mode, max(N, 1))
    s_whole = whole.cv(d_pos, dt, box, n_global)
    F_whole = whole.forces(d_pos, dt, box, n_global, 0.8) if N else None
    whole.close()
    h = C.c_void_p()
    _abi.check(lib.mtd_comm_create(C.byref(h), 0, 1, 8))
    slab = GpuMesh(_abi, dims, mode, max(N, 1))
    sizes = (C.c_size_t * 4)()
    _abi.check(lib.mtd_mesh_slab_bytes(slab.h, 1, sizes))
    peers = []
    for k in range(4):
        local, slot, hd = C.c_void_p(), C.c_uint(), (C.c_ubyte * 64)()
        _abi.check(lib.mtd_comm_share(h, sizes[k], C.byref(local), C.byref(slot), hd))
        pp = (C.c_void_p * 1)()
        _abi.check(lib.mtd_comm_open(h, slot.value, None, pp))
        peers.append(pp)
    _abi.check(lib.mtd_mesh_slab_attach(slab.h, h, peers[0], peers[1], peers[2], peers[3]))
    for rep in range(2):                                                  # twice: buffer re-use between steps
        cv_sum = C.c_void_p()
        _abi.check(lib.mtd_mesh_slab_compute_cv(slab.h, N, _abi.ptr(d_pos), dt, C.byref(box), n_global, C.byref(cv_sum), None))
        out = torch.zeros(1, dtype=torch.float64, device="cuda")
        _abi.check(lib.mtd_reduce_partials(cv_sum.value, 1, 1, 1, 0.5, 0.0, out.data_ptr(), None))
        torch.cuda.synchronize()
        s_slab = out.item()
        if abs(s_slab - s_whole) > 1e-10 * abs(s_whole) + 1e-300:
            print("DEBUG rep", rep, "dims", dims, "N", N, "dtype", np.dtype(dtype).name, "tilt", bool(tilt), "slab", s_slab, "whole", s_whole, flush=True)
    if s_whole != 0.0: worst["cv"] = max(worst["cv"], abs(s_slab - s_whole) / abs(s_whole))
    assert abs(s_slab - s_whole) <= 1e-10 * abs(s_whole) + 1e-300, ("cv", dims, N, tilt, s_slab, s_whole)
    if N:
        F_slab = slab.forces(d_pos, dt, box, n_global, 0.8)
        fm = max(np.abs(F_whole).max(), 1e-300)
        worst["f"] = max(worst["f"], np.abs(F_slab - F_whole).max() / fm)
        assert np.abs(F_slab - F_whole).max() <= (1e-9 if dtype == np.float64 else 5e-7) * fm, ("force", dims, N, tilt)
    slab.close()
    _abi.check(lib.mtd_comm_destroy(h))
print("fuzz_slab: %d random cases in %.0f s, worst relative deviations %s" % (it, time.time() - t0, {k: float("%.2e" % v) for k, v in worst.items()}))
