#!/usr/bin/env python3
"""Developer tool (GPU box): the Q9 conditioning case of tests/test_gpu_mesh.py — mesh (17, 33, 16), float32 positions: forces of
both assignment pipelines and of fp64 arrays against the oracle, the worst particles and their shifts."""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(root, "metadynamics-plugin_amd"), os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np, torch
import util, mtd_ref as ref
from metadynamics import _abi as abi
from test_gpu_mesh import GpuMesh
dims, tilt, dtype = (17, 33, 16), {}, np.float32
N = 6007; Ls = (9.0, 7.5, 11.0)
rng = np.random.default_rng(11); f = rng.random((N, 3))
a1 = np.array([Ls[0], 0, 0]); a2 = np.array([0, Ls[1], 0]); a3 = np.array([0, 0, Ls[2]])
pos = (-0.5 * np.array(Ls) + f[:, :1] * a1 + f[:, 1:2] * a2 + f[:, 2:3] * a3).astype(dtype)
types = (np.sin(2 * np.pi * 2 * f[:, 2]) > 0).astype(np.int32); mode = [1.0, -0.6]
box, rbox = abi.Box.make(Ls), ref.Box.make(Ls)
d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda(); opt = util.oracle_postype(pos, types)
r = ref.Mesh(*dims, mode); r.cv(opt, rbox); F_ref = r.forces(opt, rbox, -2.5)
out = {}
for path in ("tiles", "cells"):
    os.environ["MTD_MESH_ASSIGN"] = path
    g = GpuMesh(abi, dims, mode, N); g.cv(d_pos, abi.MTD_F32, box, N)
    out[path] = g.forces(d_pos, abi.MTD_F32, box, N, -2.5, True); g.close()
os.environ["MTD_MESH_ASSIGN"] = "tiles"
pos64 = pos.astype(np.float64)
d_pos64 = torch.from_numpy(util.pack_postype(pos64, types, np.float64)).cuda()
g = GpuMesh(abi, dims, mode, N); g.cv(d_pos64, abi.MTD_F64, box, N)
inv_dev = np.abs(g.array(3) - r.array("inv_fourier_mesh").real).max() / np.abs(r.array("inv_fourier_mesh")).max()
print("inv rel dev", inv_dev, "rho dev", np.abs(g.array(0) - r.array("mesh").real).max())
out["tiles_f64_arrays"] = g.forces(d_pos64, abi.MTD_F64, box, N, -2.5, True); g.close()
fm = np.abs(F_ref[:, :3]).max()
for path, F in out.items():
    per = np.abs(F[:, :3] - F_ref[:, :3]).max(axis=1) / fm
    k = np.argsort(per)[-3:][::-1]
    print(path, "worst", [(int(i), float("%.3g" % per[i])) for i in k], "n > 2e-7:", int((per > 2e-7).sum()))
    i = int(k[0]); print("  particle", i, "frac", f[i], "pos", pos[i], "F", F[i, :3], "F_ref", F_ref[i, :3])
    fr = (pos[i].astype(np.float64) + 0.5 * np.array(Ls)) / np.array(Ls) * np.array(dims)
    print("  mesh coordinate", fr, "shift", fr - np.floor(fr + 0.5))
print("tiles vs cells", np.abs(out["tiles"][:, :3] - out["cells"][:, :3]).max() / fm)
