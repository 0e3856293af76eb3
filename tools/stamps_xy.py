#!/usr/bin/env python3
"""Developer tool: where the blocks of k_fft_xy_forward / k_fft_xy_inverse (x and y passes of a mesh plane in one launch) spend
their time; needs the -DMTD_STAMPS diagnostic library (tools/build_stamps.sh).  s_memrealtime, 10 ns ticks."""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MTD_LIB_OVERRIDE"] = os.path.join(root, "tools", "bin", "libmtd_hip_stamps.so")
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import util
from metadynamics import _abi
lib = _abi.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N, L = 1_000_000, 100.0
pos, types = util.snapshot_random(N, L, seed=12345, modulated=True, dtype=np.float32)
pos = (np.mod(pos.astype(np.float64) + L / 2, L) - L / 2).astype(np.float32)
pos[pos >= L / 2] = -L / 2
d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
box = _abi.Box.make(L)
h = C.c_void_p()
mode = (C.c_double * 2)(1.0, -1.0)
_abi.check(lib.mtd_mesh_create(C.byref(h), n, n, n, mode, 2, N))
part, npart = C.c_void_p(), C.c_uint()
names = [["entry", "x batch 0 staged", "x sweeps", "columns kept", "x batch 1 staged", "x sweeps", "columns kept", "y sweeps", "stored"],
         ["entry", "y batch 0 staged", "y sweeps", "rows kept", "y batch 1 staged", "y sweeps", "rows kept", "x sweeps", "stored"]]
for rep in range(3):
    for t in range(30):
        _abi.check(lib.mtd_mesh_compute_cv(h, N, d_pos.data_ptr(), _abi.MTD_F32, C.byref(box), N, C.byref(part), C.byref(npart), None))
    torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 16 * 256))()
lib.mtd_debug_read_xy_stamps.argtypes = [C.c_void_p]
lib.mtd_debug_read_xy_stamps(buf)
a = np.array(buf[:], dtype=np.float64).reshape(2, 16, 256) * 0.01
for d, kn in enumerate(["k_fft_xy_forward", "k_fft_xy_inverse"]):
    used = a[d, 0] > 0
    t0 = a[d, 0][used].min()
    print(kn, "(%d blocks; us after the first block's entry: min / median / max)" % used.sum())
    for row, name in enumerate(names[d]):
        x = a[d, row][used] - t0
        print("  %-18s %6.2f %6.2f %6.2f" % (name, x.min(), np.median(x), x.max()))
