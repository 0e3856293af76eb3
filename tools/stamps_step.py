#!/usr/bin/env python3
"""Developer tool: where block 0 / the last block of k_fused_step (one-launch bias step) spend their time; needs the -DMTD_STAMPS
diagnostic library (tools/build_stamps.sh).  s_memrealtime, 10 ns ticks."""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MTD_LIB_OVERRIDE"] = os.path.join(root, "tools", "bin", "libmtd_hip_stamps.so")
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import numpy as np, torch
import util
from metadynamics import _abi
lib = _abi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = 100.0
pos, types = util.snapshot_random(N, L, seed=12345, dtype=np.float32)
cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]
lset, box = _abi.LamellarSet.make(cvs), _abi.Box.make(L)
d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
forces = [torch.zeros((N, 4), dtype=torch.float32, device="cuda") for _ in cvs]
fptr = (C.c_void_p * 2)(*[f.data_ptr() for f in forces])
lib.mtd_lamellar_set_fast_trig(1)
dbl = lambda v: (C.c_double * len(v))(*v)
h = C.c_void_p()
_abi.check(lib.mtd_metad_create(C.byref(h), 2, dbl([1e-3, 1e-3]), dbl([-0.02, -0.02]), dbl([0.02, 0.02]), (C.c_uint * 2)(256, 256), 1.0, 7.0, 1.0, 1, 1, 1))
_abi.check(lib.mtd_fused_step_set_mode(h, 1))
names = ["entry", "tables staged", "cv sums", "posted 1", "collected", "chain done", "unscaled forces", "sync", "forces stored(issued)", "grid pass + posted 2", "avg known", "applied"]
for rep in range(3):
    for t in range(60):
        _abi.check(lib.mtd_fused_step(h, C.byref(lset), N, d_pos.data_ptr(), fptr, _abi.MTD_F32, N, C.byref(box), scratch.data_ptr(), t, None))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.mtd_debug_read_step_stamps(buf)
    t = [x * 0.01 for x in buf]
    for who, off in (("block 0 wave 0", 0), ("last block wave 0", 16), ("block 0 wave 1", 32)):
        line = "%-18s" % who
        for i, n in enumerate(names):
            if t[off + i] == 0:
                continue
            line += " | %s +%.2f" % (n, t[off + i] - t[0])
        print(line)
    print()

blk = (C.c_ulonglong * (6 * 256))()
lib.mtd_debug_read_step_blocks(blk)
a = np.array(blk[:], dtype=np.float64).reshape(6, 256) * 0.01
used = a[0] > 0
t0 = a[0][used].min()
for row, name in enumerate(["entry", "posted 1", "collected", "chain done", "posted 2", "end"]):
    x = a[row][used] - t0
    print("%-10s over the 256 blocks: min %.2f  median %.2f  p90 %.2f  max %.2f (block %d)" % (name, x.min(), np.median(x), np.percentile(x, 90), x.max(), int(x.argmax())))
