#!/usr/bin/env python3
"""Developer tool: where the 1024 blocks of k_tile_scatter / k_tile_forces (mesh.hip) spend their time; needs the -DMTD_STAMPS
diagnostic library (tools/build_stamps.sh).  s_memrealtime, 10 ns ticks; thread 0 of every block."""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MTD_LIB_OVERRIDE"] = os.path.join(root, "tools", "bin", "libmtd_hip_stamps.so")
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import util
from metadynamics import _abi
lib = _abi.load()
n, N, L = 128, 1_000_000, 100.0
pos, types = util.snapshot_random(N, L, seed=12345, modulated=True, dtype=np.float32)
pos = (np.mod(pos.astype(np.float64) + L / 2, L) - L / 2).astype(np.float32)
pos[pos >= L / 2] = -L / 2
d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
d_f = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
box = _abi.Box.make(L)
h = C.c_void_p()
_abi.check(lib.mtd_mesh_create(C.byref(h), n, n, n, (C.c_double * 2)(1.0, -1.0), 2, N))
part, npart = C.c_void_p(), C.c_uint()
# usage: stamps_tiles.py [rider]   rider: a lamellar CV's partial sums ride in the binning kernel (as in a mixed set of the host classes)
rider = len(sys.argv) > 1 and sys.argv[1] == "rider"
if rider:
    _abi.check(lib.mtd_lamellar_set_fast_trig(1))
    lset = _abi.LamellarSet.make([(util.CV1_VECTORS, util.MODE_AB)])
    partials = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
    n_part = C.c_uint()
for t in range(60):
    if rider:
        _abi.check(lib.mtd_mesh_set_lamellar_rider(h, None, C.byref(lset), C.byref(box), N, partials.data_ptr(), C.byref(n_part), None))
    _abi.check(lib.mtd_mesh_compute_cv(h, N, d_pos.data_ptr(), _abi.MTD_F32, C.byref(box), N, C.byref(part), C.byref(npart), None))
    _abi.check(lib.mtd_mesh_forces(h, N, d_pos.data_ptr(), d_f.data_ptr(), _abi.MTD_F32, C.byref(box), N, None, C.c_double(-2.5), None))
torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 8 * 1024))()
lib.mtd_debug_read_tile_stamps.argtypes = [C.c_void_p]
lib.mtd_debug_read_tile_stamps(buf)
a = np.array(buf[:], dtype=np.float64).reshape(2, 8, 1024) * 0.01
names = [["entry", "image cleared, first particle requested", "first particle added", "thread 0 out of the loop", "block out of the loop", "image stored (issued)"],
         ["entry", "Re(inv) staged", "first particle done", "thread 0 out of the loop"]]
for k, kn in enumerate(["k_tile_scatter", "k_tile_forces"]):
    last = len(names[k]) - 1
    ok = (a[k, 0] > 0) & (a[k, last] > 0)                  # blocks that ran the whole kernel (the planning block and surplus blocks leave early)
    t0 = a[k, 0][ok].min()
    print(kn, "(%d blocks; us after the first block's entry: min / median / max)" % ok.sum())
    for row, name in enumerate(names[k]):
        x = a[k, row][ok] - t0
        print("  %-42s %6.2f %6.2f %6.2f" % (name, x.min(), np.median(x), x.max()))

cb = (C.c_ulonglong * (8 * 1024))()
lib.mtd_debug_read_count_stamps.argtypes = [C.c_void_p]
lib.mtd_debug_read_count_stamps(cb)
c = np.array(cb[:], dtype=np.float64).reshape(8, 1024) * 0.01
used = c[0] > 0
t0 = c[0][used].min()
print("k_tile_count / k_tile_bin (%d blocks; us after the first block's entry: min / median / max)" % used.sum())
rows_count = ["entry", "histogram cleared, first position in", "out of the particle loop (thread 0)", "block out of the loop", "histogram row and block sum written"]
rows_bin = ["entry", "histogram cleared, first position in", "out of the particle loop (thread 0)", "block out of the loop", "atomics requested, prefix written",
            "chunk sorted in LDS, atomics back", "thread 0 out of the store loop", "block sum written"]
for row, name in enumerate(rows_bin if c[7][used].max() > 0 else rows_count):
    x = c[row][used] - t0
    print("  %-42s %6.2f %6.2f %6.2f" % (name, x.min(), np.median(x), x.max()))
