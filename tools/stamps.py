#!/usr/bin/env python3
"""Developer tool: run a few bias steps with the -DMTD_STAMPS diagnostic library and print where one grid
block and one particle block of each fused kernel spend their time (s_memrealtime, 10 ns ticks)."""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MTD_LIB_OVERRIDE"] = os.path.join(root, "tools", "bin", "libmtd_hip_stamps.so")
sys.path[:0] = [root, os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests")]
import torch
import bench
from metadynamics import _abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
eng = bench.Engine(n, n, 0, seed=12345, stride=1, fast_trig=1)
box = None
if len(sys.argv) > 2 and sys.argv[2] == "mailbox":          # one-rank xGMI mailbox: the sharded step's send / receive code
    from metadynamics import xgmi
    h = C.c_void_p()
    _abi.check(_abi.load().mtd_comm_create(C.byref(h), 0, 1, 8))
    box = xgmi.Mailbox(h, 0, 1)
    eng.be.attach_mailbox(box)
for _ in range(50):
    eng.step()
torch.cuda.synchronize()
lib = _abi.load()
buf = (C.c_ulonglong * 64)()
lib.mtd_debug_read_stamps(buf)
t = [x * 0.01 for x in buf]   # us
def rel(a, b): return "%6.2f" % (t[b] - t[a])
print("N=%d" % n)
print("k_fused_cv   apply block0: entry->end %s" % rel(0, 1))
print("k_fused_cv   cv block0   : entry->tables %s  ->accumulate %s  ->reduce/store %s | cv entry - apply entry %s" % (rel(3, 4), rel(4, 5), rel(5, 6), rel(0, 3)))
print("k_fused_force grid block0 : entry->chain %s  ->sync %s  ->cells %s  ->block sums %s | kernel A start -> B start %s" % (rel(16, 17), rel(17, 18), rel(18, 19), rel(19, 20), rel(0, 16)))
print("k_fused_force force block : entry->tables %s  wave0 chain %s | wave1: tables->unscaled %s ->sync %s ->stored %s | grid entry -> force entry %s" % (rel(24, 25), rel(25, 26), rel(25, 27), rel(27, 28), rel(28, 29), rel(16, 24)))
print("A entry -> next... B end(grid blk) %s ; B force store end %s (both from B grid-block entry)" % (rel(16, 20), rel(16, 29)))
print("chain (grid block0): entry->cv sums %s  ->geometry %s  ->pairs+grid loads %s  ->V_old,scal %s  ->closed form %s  ->res,bias %s  ->bin(end) %s" % (rel(16, 40), rel(40, 41), rel(41, 42), rel(42, 43), rel(43, 44), rel(44, 45), rel(45, 17)))
if box is not None:
    print("mailbox: cv block0 entry -> all block sums collected %s  ->totals %s  ->sent %s | collected -> B grid-block entry %s" % (rel(3, 7), rel(7, 8), rel(8, 9), rel(7, 16)))
