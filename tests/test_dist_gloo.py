"""CPU, world_size 2, gloo: the N>1 orchestration of the bias step.

The product classes metadynamics.sharded.ShardedBiasStep / WalkerBiasStep are exercised with a checker
backend built on the CPU oracle (the product backend needs a GPU).  What is verified:
  * particle sharding: after the all-reduce every rank holds the single-rank CV values, its replicated
    bias grid equals the single-rank grid bit for bit, and its shard's forces equal the matching slice of the
    single-rank forces;
  * multiple walkers: after the packed-delta all-reduce both walkers hold the same grid, equal to an
    oracle run that sums the walkers' deltas.
"""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CVS = [([(0, 0, 3), (0, 3, 0), (1, 1, 1)], [1.0, -1.0]), ([(2, 0, 0), (0, 0, 6)], [1.0, -1.0])]
GRID = dict(sigma=[0.05, 0.05], cv_min=[-1.0, -1.0], cv_max=[1.0, 1.0], num_points=[48, 40])
KW = dict(W=1.0, T_shift=7.0, T=1.0, stride=2, mode="well_tempered")
N, L, STEPS = 4000, 12.0, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup_paths():
    for p in (os.path.join(ROOT, "metadynamics-plugin_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def _snapshot(t):
    import util
    pos, types = util.snapshot_random(N, L, seed=5, modulated=False, dtype=np.float64)
    a = np.where(types == 0, 1.0, -1.0)
    pos[:, 2] += (0.3 + 0.05 * t) * a * np.sin(2 * np.pi * 3 * pos[:, 2] / L)
    return pos, types


class OracleLamellarBackend:
    """checker backend: the CPU oracle behind the ShardedBiasStep protocol"""

    def __init__(self, n_global, sl):
        import mtd_ref
        self.ref, self.sl, self.n_global = mtd_ref, sl, n_global
        self.box = mtd_ref.Box.make(L)
        self.metad = mtd_ref.Metad(**GRID, **KW)
        self.t = 0
        self.forces = None
        self.cv = None
        self.bias = None

    def cv_pass(self):
        pos, types = _snapshot(self.t)
        self.opt = self.ref.as_postype(pos[self.sl], types[self.sl])
        sums = [self.ref.lamellar_fourier_modes(v, self.opt, m, self.box)[:, 0].sum() for v, m in CVS]
        return torch.tensor(sums, dtype=torch.float64)

    def force_pass(self, sums, timestep):
        self.cv = (sums / self.n_global).numpy().copy()
        self.bias = self.metad.update_bias(timestep, self.cv)
        self.forces = [self.ref.lamellar_forces(v, self.opt, m, self.box, self.bias[c], n_global=self.n_global)
                       for c, (v, m) in enumerate(CVS)]
        self.t += 1


def _worker_sharded(rank, world, port, out):
    _setup_paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from metadynamics.sharded import ShardedBiasStep
        n_local = N // world
        be = OracleLamellarBackend(N, slice(rank * n_local, (rank + 1) * n_local))
        step = ShardedBiasStep(be, dist)
        single = OracleLamellarBackend(N, slice(0, N)) if rank == 0 else None
        ok = True
        for t in range(STEPS):
            step.step(t)
            if single is not None:
                single.force_pass(single.cv_pass(), t)
                ok &= bool(np.allclose(be.cv, single.cv, rtol=1e-12, atol=1e-15))
                ok &= bool(np.allclose(be.bias, single.bias, rtol=1e-9, atol=1e-12))
                for c in range(len(CVS)):
                    ok &= bool(np.allclose(be.forces[c], single.forces[c][be.sl], rtol=1e-9, atol=1e-15))
        # replicated grids: gather rank 1's grid on rank 0 and compare bit for bit
        g = torch.from_numpy(be.metad.array("grid").copy())
        gl = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(gl, g)
        ok &= all(bool(torch.equal(gl[0], x)) for x in gl[1:])
        if single is not None:
            ok &= bool(np.allclose(be.metad.array("grid"), single.metad.array("grid"), rtol=1e-10, atol=1e-14))
            ok &= be.metad.num_gaussians == single.metad.num_gaussians == 3
        out[rank] = ok
    finally:
        dist.destroy_process_group()


class OracleWalkerBackend:
    def __init__(self, rank):
        import mtd_ref
        self.metad = mtd_ref.Metad(**GRID, **dict(KW, stride=1))
        self.rank = rank
        self.val = None

    def phase_a(self, t):
        self.val = [0.2 + 0.05 * t * (1 if self.rank == 0 else -1), -0.3 + 0.04 * t + 0.1 * self.rank]
        return bool(self.metad.phase_a(t, self.val))

    def delta_buffers(self):
        m = self.metad
        # views aliasing the engine's arrays, like the device pointers mtd_metad_delta_buffers returns
        self._real = [torch.from_numpy(m.array("grid_delta")), torch.from_numpy(m.array("sigma_grid_delta"))]
        self._cnt = [torch.from_numpy(m.array("hist_delta").view(np.int32)), torch.from_numpy(m.array("hist_gauss_delta").view(np.int32))]
        real, cnt = torch.cat(self._real), torch.cat(self._cnt)
        self._packed = (real, cnt)
        return real, cnt

    def phase_b(self, dep):
        if dep:
            real, cnt = self._packed
            G = self.metad.len
            self._real[0].copy_(real[:G]); self._real[1].copy_(real[G:])
            self._cnt[0].copy_(cnt[:G]); self._cnt[1].copy_(cnt[G:])
        self.bias = self.metad.phase_b(int(dep), self.val)


def _worker_walkers(rank, world, port, out):
    _setup_paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mtd_ref
        from metadynamics.sharded import WalkerBiasStep
        be = OracleWalkerBackend(rank)
        step = WalkerBiasStep(be, dist)
        # reference: both walkers in one process, deltas summed by hand
        refs = [OracleWalkerBackend(r) for r in range(world)] if rank == 0 else None
        ok = True
        for t in range(4):
            step.step(t)
            if refs is not None:
                deps = [r.phase_a(t) for r in refs]
                for name in ("grid_delta", "sigma_grid_delta", "hist_delta", "hist_gauss_delta"):
                    tot = sum(r.metad.array(name) for r in refs)
                    for r in refs:
                        r.metad.array(name)[:] = tot
                for r, d in zip(refs, deps):
                    r.metad.phase_b(int(d), r.val)
                for name in mtd_ref.ARRAY_NAMES:
                    ok &= bool(np.array_equal(be.metad.array(name), refs[0].metad.array(name)))
        g = torch.from_numpy(be.metad.array("grid").copy())
        gl = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(gl, g)
        ok &= all(bool(torch.equal(gl[0], x)) for x in gl[1:])
        ok &= float(g.abs().max()) > 0
        out[rank] = ok
    finally:
        dist.destroy_process_group()


class OracleCvSetBackend:
    """checker backend for a mixed CV set (lamellar + mesh + Steinhardt + wrapped energy): one exchange buffer per CV,
    exactly the buffers metadynamics.sharded.HipCvSetBackend hands to the all-reduce"""

    LV, MODE = [(0, 0, 3), (0, 3, 0)], [1.0, -1.0]
    QL = dict(rcut=1.4, ron=1.2, lmax=6, Ql_ref=[0, 0, 0, 0, 1, 0, 1])

    def __init__(self, pos, types, L, ids, energy, ext, grid):
        import mtd_ref
        import util
        self.ref = mtd_ref
        self.N = len(pos)
        self.ids = ids
        self.box = mtd_ref.Box.make(L)
        self.opt = mtd_ref.as_postype(pos[ids], types[ids])
        self.mesh = mtd_ref.Mesh(8, 8, 8, self.MODE)
        # Steinhardt shard: own particles first, every other particle as ghost; neighbour list re-indexed
        head, nn, nl = util.build_nlist(pos, L, 1.5)
        other = np.setdiff1d(np.arange(self.N), ids)
        perm = np.concatenate([ids, other])
        inv = np.empty(self.N, dtype=np.int64)
        inv[perm] = np.arange(self.N)
        h2, n2, l2 = [], [], []
        for i in ids:
            h2.append(len(l2))
            n2.append(nn[i])
            l2.extend(inv[nl[head[i]:head[i] + nn[i]]])
        self.nlist = (np.asarray(h2, dtype=np.uint32), np.asarray(n2, dtype=np.uint32), np.asarray(l2, dtype=np.uint32))
        self.opt_ql = mtd_ref.as_postype(pos[perm], np.zeros(self.N, dtype=np.int32))
        self.energy, self.ext = energy[ids], ext
        self.metad = mtd_ref.Metad(**grid, **dict(KW, stride=1))

    def cv_pass(self):
        r = self.ref
        lam = torch.tensor([r.lamellar_fourier_modes(self.LV, self.opt, self.MODE, self.box)[:, 0].sum()], dtype=torch.float64)
        self.mesh.assign(self.opt, self.box)
        mesh = torch.from_numpy(np.concatenate([self.mesh.raw_mesh()[:, 0].copy(), [self.mesh.mode_sq]]))
        q = self.QL
        _, qlm, _ = r.ql_compute_cv(self.opt_ql, self.box, *self.nlist, q["rcut"], q["ron"], q["lmax"], 0, q["Ql_ref"], n_global=self.N)
        ql = torch.from_numpy(np.concatenate([qlm.real, qlm.imag]))
        en = torch.tensor([r.wrapper_energy(np.column_stack([np.zeros((len(self.ids), 3)), self.energy]), self.ext)], dtype=torch.float64)
        return [lam, mesh, ql, en]

    def force_pass(self, bufs, timestep):
        r, q = self.ref, self.QL
        lam, mesh, ql, en = [b.numpy() for b in bufs]
        self.mesh.raw_mesh()[:, 0] = mesh[:-1]
        self.mesh.set_mode_sq(mesh[-1])
        n = len(ql) // 2
        self.qlm = ql[:n] + 1j * ql[n:]
        s_ql, _ = r.ql_from_qlm(q["lmax"], self.qlm, q["Ql_ref"], self.N)
        self.cv = np.array([lam[0] / self.N, self.mesh.spectral(self.N), s_ql, en[0]])
        self.bias = self.metad.update_bias(timestep, self.cv)
        self.forces = [r.lamellar_forces(self.LV, self.opt, self.MODE, self.box, self.bias[0], n_global=self.N),
                       self.mesh.forces(self.opt, self.box, self.bias[1], n_global=self.N),
                       r.ql_compute_forces(self.opt_ql, self.box, *self.nlist, q["rcut"], q["ron"], q["lmax"], 0, q["Ql_ref"], self.qlm,
                                           self.bias[2], n_global=self.N)[:len(self.ids)]]


def _worker_cvset(rank, world, port, out):
    _setup_paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mtd_ref
        import util
        from metadynamics.sharded import ShardedBiasStep
        pos, Lc = util.fcc_lattice(4)
        rng = np.random.default_rng(17)
        pos = pos + rng.normal(0, 0.05, pos.shape)
        n = len(pos)
        types = (np.arange(n) % 2).astype(np.int32)
        energy = rng.normal(-0.5, 0.1, n)
        # single-rank values fix the grid ranges (identical on both ranks: same seeds)
        full = np.arange(n)
        probe = OracleCvSetBackend(pos, types, Lc, full, energy, 3.0, dict(sigma=[1, 1, 1, 1], cv_min=[0] * 4, cv_max=[1] * 4, num_points=[2] * 4))
        probe.force_pass(probe.cv_pass(), 0)
        s = probe.cv
        grid = dict(sigma=[0.05, 0.1 * abs(s[1]), 0.02 * s[2], 2.0], cv_min=[s[0] - 0.5, 0.3 * s[1], 0.6 * s[2], s[3] - 20.0],
                    cv_max=[s[0] + 0.6, 1.8 * s[1], 1.3 * s[2], s[3] + 27.0], num_points=[12, 10, 12, 10])
        ids = np.where(pos[:, 0] < 0)[0] if rank == 0 else np.where(pos[:, 0] >= 0)[0]
        be = OracleCvSetBackend(pos, types, Lc, ids, energy, 1.0 if rank == 0 else 2.0, grid)
        step = ShardedBiasStep(be, dist)
        single = OracleCvSetBackend(pos, types, Lc, full, energy, 3.0, grid)
        ok = True
        for t in range(3):
            step.step(t)
            single.force_pass(single.cv_pass(), t)
            ok &= bool(np.allclose(be.cv, single.cv, rtol=1e-10, atol=1e-14))
            ok &= bool(np.allclose(be.bias, single.bias, rtol=1e-6, atol=1e-9 * np.abs(single.bias).max()))
            for c in range(3):
                ref_f = single.forces[c][ids]
                ok &= bool(np.abs(be.forces[c][:, :3] - ref_f[:, :3]).max() <= 1e-6 * np.abs(single.forces[c][:, :3]).max() + 1e-16)
        ok &= float(np.abs(single.bias).max()) > 0
        g = torch.from_numpy(be.metad.array("grid").copy())
        gl = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(gl, g)
        ok &= all(bool(torch.equal(gl[0], x)) for x in gl[1:])
        out[rank] = ok
    finally:
        dist.destroy_process_group()


def _run(worker):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_sharded_particles_two_ranks():
    _run(_worker_sharded)


def test_multiple_walkers_two_ranks():
    _run(_worker_walkers)


def test_sharded_cv_set_two_ranks():
    """lamellar + mesh + Steinhardt (with ghosts) + wrapped energy: one all-reduce per CV buffer"""
    _run(_worker_cvset)
