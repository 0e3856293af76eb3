"""GPU parity: particle-mesh order parameter (C-ABI) vs the oracle restatement of OrderParameterMesh.cc.
Everything is double precision on both sides: meshes agree to 1e-11, the CV to 1e-9 relative (tolerance stated by
BASELINE.json: 1e-6), forces to 1e-8 of max|F| (stated: 1e-5)."""
import ctypes as C

import numpy as np
import pytest

import util

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


class GpuMesh:
    def __init__(self, abi, dims, mode, max_particles):
        self.abi, self.lib = abi, abi.load()
        self.dims = dims
        self.h = C.c_void_p()
        abi.check(self.lib.mtd_mesh_create(C.byref(self.h), dims[0], dims[1], dims[2], util.dbl_array(mode), len(mode), max_particles))
        self.M = self.lib.mtd_mesh_num_cells(self.h)

    def close(self):
        if self.h:
            self.abi.check(self.lib.mtd_mesh_destroy(self.h))
            self.h = None

    def cv(self, d_pos, dt, box, n_global):
        parts, n = C.c_void_p(), C.c_uint()
        self.abi.check(self.lib.mtd_mesh_compute_cv(self.h, d_pos.shape[0], self.abi.ptr(d_pos), dt, C.byref(box), n_global,
                                                    C.byref(parts), C.byref(n), None))
        out = torch.zeros(1, dtype=torch.float64, device="cuda")
        self.abi.check(self.lib.mtd_reduce_partials(parts.value, n.value, 1, 1, 0.5, 0.0, self.abi.ptr(out), None))
        torch.cuda.synchronize()
        return out.item()

    def forces(self, d_pos, dt, box, n_global, bias, device_bias=True):
        f = torch.zeros_like(d_pos)
        d_bias = torch.tensor([bias], dtype=torch.float64, device="cuda")
        self.abi.check(self.lib.mtd_mesh_forces(self.h, d_pos.shape[0], self.abi.ptr(d_pos), self.abi.ptr(f), dt, C.byref(box),
                                                n_global, self.abi.ptr(d_bias) if device_bias else None, bias, None))
        torch.cuda.synchronize()
        return f.cpu().numpy().astype(np.float64)

    def array(self, which):
        nz, ny, nx = self.dims[2], self.dims[1], self.dims[0]
        if which in (0, 3):
            out = np.zeros(self.M)
        elif which == 7:
            out = np.zeros(1)
        else:
            out = np.zeros(2 * self.M)
        self.abi.check(self.lib.mtd_mesh_get_array(self.h, which, out.ctypes.data, None))
        if which in (0, 3):
            return out.reshape(nz, ny, nx)
        if which == 7:
            return out[0]
        return (out[0::2] + 1j * out[1::2]).reshape(nz, ny, nx)


def assign_info(abi, g):
    """(pipeline, particles through the overflow list) of the mesh's last assignment: mtd_mesh_assign_info"""
    pl, ovf = C.c_int(-1), C.c_uint(0)
    abi.check(abi.load().mtd_mesh_assign_info(g.h, C.byref(pl), C.byref(ovf), None))
    return pl.value, ovf.value


@pytest.fixture(params=["tiles", "tiles-ids", "cells"])
def assign_path(request, monkeypatch):
    """the assignment / force pipelines of mesh.hip: by tiles with the sorted place kernel (default: position records travel into
    tile order in runs), by tiles with the one-store-per-particle place kernel (its fallback: the scatter pass gathers by id), and
    the cell-level one (meshes > 256^3)"""
    monkeypatch.setenv("MTD_MESH_ASSIGN", request.param.split("-")[0])
    monkeypatch.setenv("MTD_MESH_PLACE", "ids" if request.param.endswith("-ids") else "sorted")
    return request.param


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("N,dims", [(0, (16, 16, 16)), (1, (16, 16, 16)), (4097, (32, 16, 24)), (70001, (64, 32, 16)), (12289, (128, 8, 8))])
def test_sorted_place_against_place_by_id(abi, monkeypatch, dtype, N, dims):
    """k_tile_place_sorted against k_tile_place: the same particles in every tile (in another order inside a tile, which the
    fixed-point sums do not see), so the mesh, the CV and every particle's force are the same bits.  Chunk boundaries: empty
    system, one particle, one particle beyond a chunk, many chunks."""
    L = 14.0
    pos, types = util.snapshot_random(max(N, 1), L, seed=5 + N, modulated=True, dtype=dtype)
    pos, types = pos[:N], types[:N]
    box = abi.Box.make(L)
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda() if N else torch.zeros((0, 4), dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda")
    out = {}
    for place in ("sorted", "ids"):
        monkeypatch.setenv("MTD_MESH_PLACE", place)
        g = GpuMesh(abi, dims, [1.0, -0.7], max(N, 1))
        try:
            s = g.cv(d_pos, dt, box, max(N, 1))
            F = g.forces(d_pos, dt, box, max(N, 1), 0.8) if N else np.zeros((0, 4))
            out[place] = (s, g.array(0).copy(), F)
        finally:
            g.close()
    assert out["sorted"][0] == out["ids"][0]
    assert np.array_equal(out["sorted"][1], out["ids"][1])
    assert np.array_equal(out["sorted"][2], out["ids"][2])
    if N > 1:
        assert np.abs(out["sorted"][1]).max() > 0 and np.abs(out["sorted"][2]).max() > 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("N,dims", [(1, (16, 16, 16)), (5000, (32, 16, 24)), (70001, (128, 32, 16)), (200003, (128, 128, 128))])
def test_bin_pipeline_against_counting_pipeline(abi, monkeypatch, dtype, N, dims):
    """k_tile_bin (one launch: segments with slack planned from the previous snapshot's exact counts, runs reserved with atomics,
    overflow list) against count -> row scan -> sorted place on a SEQUENCE of snapshots through one mesh: the first assignment
    (counting pipeline, plans), the same snapshot again (bin, everything fits), a snapshot with every particle in a corner of the
    box (most of it overflows the old plan), that one again (re-planned), a uniform one again (the corner's segments overflow the
    other way), an empty... no: a shuffled one.  The same particles end up in every tile, so mesh, sum of mode^2, CV and every
    particle's force are the same bits as with MTD_MESH_BIN=0."""
    L = 14.0
    rng = np.random.default_rng(100 + N)
    pos_u, types = util.snapshot_random(N, L, seed=9 + N, modulated=True, dtype=dtype)
    pos_c = (pos_u.astype(np.float64) * 0.12 - 0.3 * L).astype(dtype)              # every particle in one corner
    perm = rng.permutation(N)
    seq = [(pos_u, types), (pos_u, types), (pos_c, types), (pos_c, types), (pos_u, types), (pos_u[perm], types[perm])]
    box = abi.Box.make(L)
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MTD_MESH_BIN", mode)
        g = GpuMesh(abi, dims, [1.0, -0.7], N)
        res = []
        try:
            for step, (pos, ty) in enumerate(seq):
                d_pos = torch.from_numpy(util.pack_postype(pos, ty, dtype)).cuda()
                s = g.cv(d_pos, dt, box, N)
                pl, n_ovf = C.c_int(-1), C.c_uint(0)
                abi.check(g.lib.mtd_mesh_assign_info(g.h, C.byref(pl), C.byref(n_ovf), None))
                assert pl.value == (2 if mode == "1" and step > 0 else 1)
                if mode == "1" and N >= 5000:
                    # the plan fits a repeated snapshot; the corner snapshot overflows the uniform plan and vice versa
                    assert (n_ovf.value > 0) == (step in (2, 4)), (step, n_ovf.value)
                F = g.forces(d_pos, dt, box, N, 0.8)
                res.append((s, g.array(0).copy(), g.array(7), F))
        finally:
            g.close()
        out[mode] = res
    for a, b in zip(out["1"], out["0"]):
        assert a[0] == b[0] and a[2] == b[2]
        assert np.array_equal(a[1], b[1])
        assert np.array_equal(a[3], b[3])
    if N > 1:
        assert np.abs(out["1"][2][1]).max() > 0 and np.abs(out["1"][2][3]).max() > 0


def test_bin_pipeline_with_a_changing_particle_number(abi, monkeypatch):
    """a domain-decomposed run's local particle number changes from step to step: the plan of the previous snapshot still serves (what
    does not fit overflows), only a count that differs by a factor of two or more is counted and planned afresh.  Mesh, CV and
    forces are the same bits as with the counting pipeline for every N of the sequence."""
    L, dims, n_max = 11.0, (32, 32, 16), 30000
    pos, types = util.snapshot_random(n_max, L, seed=41, modulated=True, dtype=np.float32)
    box = abi.Box.make(L)
    seq = [5000, 5200, 4000, 7500, 30000, 29000, 0, 1, 17]
    expect = [1, 2, 2, 2, 1, 2, 1, 2, 1]          # 1 counting, 2 bin: 30000 > 2 x 7500; 0 < 29000 / 2; 1 / 2 <= 0 (a plan of empty tiles: all overflow); 17 / 2 > 1
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MTD_MESH_BIN", mode)
        g = GpuMesh(abi, dims, [1.0, -0.7], n_max)
        res = []
        try:
            for n, want in zip(seq, expect):
                d_pos = torch.from_numpy(util.pack_postype(pos[:n], types[:n], np.float32)).cuda() if n else torch.zeros((0, 4), dtype=torch.float32, device="cuda")
                s = g.cv(d_pos, abi.MTD_F32, box, max(n, 1))
                pl = C.c_int(-1)
                abi.check(g.lib.mtd_mesh_assign_info(g.h, C.byref(pl), None, None))
                assert pl.value == (want if mode == "1" else 1), (n, pl.value)
                F = g.forces(d_pos, abi.MTD_F32, box, max(n, 1), 0.8) if n else np.zeros((0, 4))
                res.append((s, g.array(0).copy(), F))
        finally:
            g.close()
        out[mode] = res
    for a, b in zip(out["1"], out["0"]):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_bin_pipeline_over_a_random_walk(abi, monkeypatch):
    """sixty snapshots of a random walk (rms step 0.3 mesh cells, a drift on top: the tiles' populations shift steadily) through one
    mesh: every step plans the next one's segments from its own exact counts, the two sets of segments alternate.  CV and mesh of
    every step are the same bits as with the counting pipeline, whatever overflowed."""
    L, dims, N = 16.0, (32, 32, 32), 20000
    rng = np.random.default_rng(7)
    pos, types = util.snapshot_random(N, L, seed=43, modulated=True, dtype=np.float32)
    box = abi.Box.make(L)
    walk = [pos]
    for step in range(59):
        p = walk[-1].astype(np.float64) + rng.normal(0.0, 0.3 * L / 32, size=pos.shape) + np.array([0.05, 0.0, -0.03])
        p = (np.mod(p + L / 2, L) - L / 2).astype(np.float32)
        p[p >= L / 2] = -L / 2
        walk.append(p)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MTD_MESH_BIN", mode)
        g = GpuMesh(abi, dims, [1.0, -0.7], N)
        res, n_overflow = [], 0
        try:
            for p in walk:
                d_pos = torch.from_numpy(util.pack_postype(p, types, np.float32)).cuda()
                s = g.cv(d_pos, abi.MTD_F32, box, N)
                pl, n_ovf = C.c_int(-1), C.c_uint(0)
                abi.check(g.lib.mtd_mesh_assign_info(g.h, C.byref(pl), C.byref(n_ovf), None))
                n_overflow += n_ovf.value
                res.append((s, g.array(0).copy()))
            F = g.forces(d_pos, abi.MTD_F32, box, N, 0.8)
        finally:
            g.close()
        out[mode] = (res, F, n_overflow)
    for a, b in zip(out["1"][0], out["0"][0]):
        assert a[0] == b[0] and np.array_equal(a[1], b[1])
    assert np.array_equal(out["1"][1], out["0"][1])
    # this walk is ~50 thermal MD steps per snapshot with tiles of only 8^3 cells: a few particles per step do not fit their tile's
    # planned segment (slack max(count / 8, 32)) and travel through the overflow list — a small fraction, and the same bits
    assert out["1"][2] < 0.01 * N * len(walk) and out["0"][2] == 0


def test_mesh_bitwise_independent_of_particle_order(abi):
    """tile path: the weights are summed as 64-bit fixed point, so the mesh does not depend on the order of the adds"""
    N, L = 40011, 12.0
    pos, types = util.snapshot_random(N, L, seed=3, dtype=np.float64)
    box = abi.Box.make(L)
    perm = np.random.default_rng(0).permutation(N)
    rho = []
    for order in (np.arange(N), perm):
        m = GpuMesh(abi, (32, 16, 24), [1.0, -0.7], N)
        d = torch.from_numpy(util.pack_postype(pos[order], types[order], np.float64)).cuda()
        m.cv(d, abi.MTD_F64, box, N)
        rho.append(m.array(0).copy())
        m.close()
    assert np.array_equal(rho[0], rho[1])
    assert np.abs(rho[0]).max() > 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dims,tilt", [((8, 8, 8), {}), ((16, 8, 32), dict(xy=0.2, xz=-0.1, yz=0.15)), ((32, 32, 32), {}),
                                       # sizes that are not powers of two (direct transforms, partial gather tiles, odd lengths)
                                       ((12, 20, 6), dict(xy=0.1, xz=0.05, yz=-0.2)), ((5, 7, 9), {}), ((48, 16, 36), {}),
                                       # one-cell edge tiles in x and y (the loop form of the combine pass; the others take the row form)
                                       ((17, 33, 16), {}),
                                       # corners of the one-launch x/y transform (k_fft_xy_*): one k_x column in a part, one line
                                       # pair per batch, the longest lines whose planes still fit the LDS, planes that do not
                                       ((4, 4, 4), {}), ((256, 8, 4), {}), ((4, 64, 8), dict(xy=-0.15, xz=0.1, yz=0.05)), ((64, 128, 16), {}),
                                       ((512, 4, 8), {}), ((256, 256, 4), {})])
def test_mesh_cv_and_forces(abi, ref, dtype, dims, tilt, assign_path):
    N = 6007
    Ls = (9.0, 7.5, 11.0)
    rng = np.random.default_rng(11)
    f = rng.random((N, 3))
    a1 = np.array([Ls[0], 0, 0]); a2 = np.array([tilt.get("xy", 0) * Ls[1], Ls[1], 0])
    a3 = np.array([tilt.get("xz", 0) * Ls[2], tilt.get("yz", 0) * Ls[2], Ls[2]])
    pos = (-0.5 * np.array(Ls) + f[:, :1] * a1 + f[:, 1:2] * a2 + f[:, 2:3] * a3).astype(dtype)
    # a density wave so the CV is not pure noise
    types = (np.sin(2 * np.pi * 2 * f[:, 2]) > 0).astype(np.int32)
    mode = [1.0, -0.6]
    box, rbox = abi.Box.make(Ls, **tilt), ref.Box.make(Ls, **tilt)
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda()
    opt = util.oracle_postype(pos, types)
    g = GpuMesh(abi, dims, mode, N)
    r = ref.Mesh(*dims, mode)
    try:
        for bug_compat in (True, False):
            abi.check(g.lib.mtd_mesh_set_bug_compat(g.h, int(bug_compat)))
            r.set_bug_compat(bug_compat)
            s = g.cv(d_pos, dt, box, N)
            s_ref = r.cv(opt, rbox)
            assert np.allclose(g.array(0), r.array("mesh").real, rtol=1e-12, atol=1e-13)
            assert g.array(7) == pytest.approx(r.mode_sq, rel=1e-13)
            scale = np.abs(r.array("fourier_mesh")).max()
            assert np.abs(g.array(1) - r.array("fourier_mesh")).max() <= 1e-12 * scale
            scale = np.abs(r.array("inv_fourier_mesh")).max()
            assert np.abs(g.array(3) - r.array("inv_fourier_mesh").real).max() <= 1e-11 * scale
            assert s == pytest.approx(s_ref, rel=1e-9)
            for device_bias in (True, False):
                F = g.forces(d_pos, dt, box, N, -2.5, device_bias)
                F_ref = r.forces(opt, rbox, -2.5)
                tol = 1e-8 if dtype == np.float64 else 2e-7    # fp32 force array: one rounding on store
                # the in-cell shift is formed with the reference's operations in the reference's order (mesh.hip locate): the
                # float rounding inside the TSC derivative (Q9) resolves as in the oracle for every particle, any box / mesh ratio
                assert np.abs(F[:, :3] - F_ref[:, :3]).max() <= tol * np.abs(F_ref[:, :3]).max()
                assert np.all(F[:, 3] == 0.0)
    finally:
        g.close()


def test_mesh_edge_cases(abi, ref, assign_path):
    """crowded cells (many particles per cell: per-cell sort), particles exactly on the box boundary (ix == nx -> 0,
    :556-561), N_global != N, empty system"""
    lib = abi.load()
    L = 4.0
    rng = np.random.default_rng(5)
    pos = np.concatenate([rng.normal(0.3, 0.05, size=(3000, 3)),          # ~all in a handful of cells
                          np.array([[2.0, -2.0, 2.0], [-2.0, 2.0, 0.0], [1.999999, 0.0, -2.0]])])
    pos = np.clip(pos, -2.0, 2.0)
    types = rng.integers(0, 2, len(pos)).astype(np.int32)
    N = len(pos)
    box, rbox = abi.Box.make(L), ref.Box.make(L)
    d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float64)).cuda()
    g = GpuMesh(abi, (8, 8, 8), [1.0, -1.0], N)
    r = ref.Mesh(8, 8, 8, [1.0, -1.0])
    try:
        s = g.cv(d_pos, abi.MTD_F64, box, 5 * N)
        s_ref = r.cv(util.oracle_postype(pos, types), rbox, n_global=5 * N)
        assert np.allclose(g.array(0), r.array("mesh").real, rtol=1e-12, atol=1e-13)
        assert s == pytest.approx(s_ref, rel=1e-9)
        F = g.forces(d_pos, abi.MTD_F64, box, 5 * N, 1.0)
        F_ref = r.forces(util.oracle_postype(pos, types), rbox, 1.0, n_global=5 * N)
        assert np.abs(F - F_ref).max() <= 1e-8 * np.abs(F_ref).max()
        # deterministic: a second evaluation gives the same bits
        s2 = g.cv(d_pos, abi.MTD_F64, box, 5 * N)
        assert s2 == s
        empty = torch.zeros((0, 4), dtype=torch.float64, device="cuda")
        assert g.cv(empty, abi.MTD_F64, box, 10) == 0.0
    finally:
        g.close()
    h = C.c_void_p()
    assert lib.mtd_mesh_create(C.byref(h), 300, 8, 8, util.dbl_array([1.0]), 1, 10) == -2  # not a power of two and > 256
    assert lib.mtd_mesh_create(C.byref(h), 2048, 8, 8, util.dbl_array([1.0]), 1, 10) == -2
    assert lib.mtd_mesh_create(C.byref(h), 3, 8, 8, util.dbl_array([1.0]), 1, 10) == -2    # 3x3x3 stencils would alias
    assert lib.mtd_mesh_create(C.byref(h), 8, 8, 0, util.dbl_array([1.0]), 1, 10) == -1


def test_mesh_config3_size(abi, ref):
    """BASELINE.json configs[2] size: 10^6 particles on a 128^3 mesh (bug-compatible), checked through properties that
    do not need the oracle at full size (charge conservation, determinism) plus the oracle on the CV itself"""
    N, L = 1_000_000, 100.0
    pos, types = util.snapshot_random(N, L, seed=12345, modulated=True, dtype=np.float32)
    box, rbox = abi.Box.make(L), ref.Box.make(L)
    d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
    g = GpuMesh(abi, (128, 128, 128), util.MODE_AB, N)
    try:
        s = g.cv(d_pos, abi.MTD_F32, box, N)
        rho = g.array(0)
        a = np.where(types == 0, 1.0, -1.0)
        assert rho.sum() == pytest.approx(a.sum(), abs=1e-6)            # TSC weights sum to one per particle
        assert g.array(7) == float(N)
        r = ref.Mesh(128, 128, 128, util.MODE_AB)
        s_ref = r.cv(util.oracle_postype(pos, types), rbox)
        assert s == pytest.approx(s_ref, rel=1e-6)
        F = g.forces(d_pos, abi.MTD_F32, box, N, 1.0)
        sl = slice(0, 50_000)
        F_ref = r.forces(util.oracle_postype(pos, types)[sl], rbox, 1.0, n_global=N)
        assert np.abs(F[sl, :3] - F_ref[:, :3]).max() <= 1e-5 * np.abs(F_ref[:, :3]).max()
        assert assign_info(abi, g) == (1, 0)                             # the first assignment of a mesh: counting pipeline

        # The BENCHMARKED assignment is the bin pipeline (every assignment of a mesh but its first: one launch bins on tile segments
        # planned from the previous snapshot) — it meets the oracle directly here, at 10^6 particles / 128^3: the same
        # snapshot again (planned from its own exact counts), once more, then a DISPLACED snapshot binned on the plan of the old one
        for rep in range(2):
            s2 = g.cv(d_pos, abi.MTD_F32, box, N)
            pl, ovf = assign_info(abi, g)
            assert pl == 2, "bin pipeline expected"
            assert s2 == pytest.approx(s_ref, rel=1e-6)
            assert np.array_equal(g.array(0), rho)                      # integer sums: the mesh does not depend on the pipeline
            F2 = g.forces(d_pos, abi.MTD_F32, box, N, 1.0)
            assert np.abs(F2[sl, :3] - F_ref[:, :3]).max() <= 1e-5 * np.abs(F_ref[:, :3]).max()
        rng = np.random.default_rng(99)
        pos2 = pos.astype(np.float64) + rng.normal(0.0, 0.1 * L / 128, size=pos.shape)      # rms 0.1 mesh cells per coordinate
        pos2 = (np.mod(pos2 + L / 2, L) - L / 2).astype(np.float32)
        pos2[pos2 >= L / 2] = -L / 2
        d_pos2 = torch.from_numpy(util.pack_postype(pos2, types, np.float32)).cuda()
        s3 = g.cv(d_pos2, abi.MTD_F32, box, N)
        assert assign_info(abi, g)[0] == 2
        opt2 = util.oracle_postype(pos2, types)
        s3_ref = r.cv(opt2, rbox)
        assert s3 == pytest.approx(s3_ref, rel=1e-6) and abs(s3_ref - s_ref) > 1e-9 * abs(s_ref)
        F3 = g.forces(d_pos2, abi.MTD_F32, box, N, 1.0)
        F3_ref = r.forces(opt2[sl], rbox, 1.0, n_global=N)
        assert np.abs(F3[sl, :3] - F3_ref[:, :3]).max() <= 1e-5 * np.abs(F3_ref[:, :3]).max()
    finally:
        g.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("N,dims", [(3, (128, 16, 16)), (30011, (128, 32, 64)), (400009, (128, 128, 128)), (20011, (128, 64, 8)), (3, (128, 128, 16))])
def test_forward_transform_from_tile_images_against_combined_mesh(abi, monkeypatch, dtype, N, dims):
    """k_fft_xy_forward<true> (meshes 128 cells wide: the transform sums the tile images' halo entries itself, two cells per lane
    with 16-byte loads, no combine launch) against k_tile_combine_rows + k_fft_xy_forward<false> (MTD_FFT_FROM_TILES=0), both with
    the unsplit x/y kernels (MTD_FFT_SPLIT=0): the sums are integers, so the CV, the Fourier mesh, the combined mesh read back
    afterwards and every particle's force are the same BITS.  And the default — the x/y transforms split by the parity of their
    output (k_fft_xy_forward_split on 128 x 128 planes, k_fft_xy_inverse_split) — against them: another factorisation of the same
    transform, equal to rounding (CV 1e-12, Fourier mesh 1e-12 of its largest element, forces 1e-10 / one fp32 rounding).  Two
    snapshots each (counting pipeline, then the bin pipeline's other cursor set)."""
    L = 21.0
    box = abi.Box.make(L)
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    snaps = []
    for seed in (3, 4):
        pos, types = util.snapshot_random(N, L, seed=seed + N, modulated=True, dtype=dtype)
        snaps.append(torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda())
    out = {}
    for mode, (tiles, split) in dict(default=("1", "1"), tiles=("1", "0"), combine=("0", "0")).items():
        monkeypatch.setenv("MTD_FFT_FROM_TILES", tiles)
        monkeypatch.setenv("MTD_FFT_SPLIT", split)
        g = GpuMesh(abi, dims, [1.0, -0.6], N)
        try:
            res = []
            for d_pos in snaps:
                s = g.cv(d_pos, dt, box, N)
                fwd = C.c_int(-1)
                abi.check(g.lib.mtd_mesh_transform_info(g.h, C.byref(fwd)))
                # (the tile form needs batches of 32 line pairs: at least 64 rows; smaller planes keep the combine launch)
                assert fwd.value == (2 if tiles == "1" and dims[1] >= 64 else 1)
                F = g.forces(d_pos, dt, box, N, 0.7)
                res.append((s, F.copy(), g.array(0).copy(), g.array(1).copy()))
            out[mode] = res
        finally:
            g.close()
    for a, b in zip(out["tiles"], out["combine"]):
        assert a[0] == b[0] and a[0] != 0.0
        for x, y in zip(a[1:], b[1:]):
            assert np.array_equal(x, y)
        assert np.abs(a[1]).max() > 0
    for a, b in zip(out["default"], out["tiles"]):
        assert a[0] == pytest.approx(b[0], rel=1e-12)
        assert np.array_equal(a[2], b[2])                               # the real mesh: integer sums, no transform in it
        assert np.abs(a[3] - b[3]).max() <= 1e-12 * np.abs(b[3]).max()
        assert np.abs(a[1] - b[1]).max() <= (2e-6 if dtype == np.float32 else 1e-10) * np.abs(b[1]).max()


@pytest.mark.parametrize("tilt", [{}, dict(xy=0.1, xz=-0.05, yz=0.2)])
def test_qmax_virial_table(abi, ref, tilt):
    """q_max / sq_max log quantities (OrderParameterMesh.cc:1108-1179), convolution-kernel table
    (:148-189) and the virial (:970-1050) vs the oracle, incl. a triclinic box"""
    lib = abi.load()
    N, L = 20000, 16.0
    rng = np.random.default_rng(77)
    f = rng.random((N, 3))                                            # fractional coordinates: inside the (tilted) box
    a1 = np.array([L, 0, 0]); a2 = np.array([tilt.get("xy", 0) * L, L, 0])
    a3 = np.array([tilt.get("xz", 0) * L, tilt.get("yz", 0) * L, L])
    pos = -0.5 * L + f[:, :1] * a1 + f[:, 1:2] * a2 + f[:, 2:3] * a3
    types = (np.sin(2 * np.pi * 3 * f[:, 2]) > 0).astype(np.int32)    # a density wave along the third reciprocal vector
    box, rbox = abi.Box.make(L, **tilt), ref.Box.make(L, **tilt)
    mode = [1.0, -1.0]
    m = GpuMesh(abi, (32, 16, 32), mode, N)
    r = ref.Mesh(32, 16, 32, mode)
    d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float64)).cuda()
    pt = util.oracle_postype(pos, types)
    s_gpu, s_ref = m.cv(d_pos, abi.MTD_F64, box, N), r.cv(pt, rbox)
    assert s_gpu == pytest.approx(s_ref, rel=1e-9)
    out = np.zeros(4)
    abi.check(lib.mtd_mesh_qmax(m.h, C.byref(box), N, out.ctypes.data_as(C.POINTER(C.c_double)), None))
    q_ref = r.qmax(N)
    # |f(k)| = |f(-k)| for a real mesh: which of the two cells holds the (rounding-level) larger value depends on the FFT's
    # summation order, in the reference as well — the wave vector is defined up to its sign
    assert np.abs(q_ref[:3]).max() > 0
    assert np.allclose(out[:3], q_ref[:3], rtol=1e-12, atol=1e-13) or np.allclose(out[:3], -q_ref[:3], rtol=1e-12, atol=1e-13)
    assert out[3] == pytest.approx(q_ref[3], rel=1e-10)

    vir = np.ones(6)
    abi.check(lib.mtd_mesh_virial(m.h, C.byref(box), N, 0.8, vir.ctypes.data_as(C.POINTER(C.c_double)), None))
    assert np.all(vir == 0.0) and np.all(r.virial(N, 0.8) == 0.0)          # no table in use
    # a smooth kernel on a window that cuts through the populated k range (some cells outside [kmin, kmax))
    npts, kmin, kmax = 48, 0.5, 9.0
    kt = np.linspace(kmin, kmax, npts)
    K, dK = np.exp(-0.1 * kt ** 2), -0.2 * kt * np.exp(-0.1 * kt ** 2)
    abi.check(lib.mtd_mesh_set_table(m.h, util.dbl_array(K), util.dbl_array(dK), npts, kmin, kmax))
    abi.check(lib.mtd_mesh_set_use_table(m.h, 1))
    r.set_table(K, dK, kmin, kmax)
    r.set_use_table(True)
    # K is stored and never applied (Q7): the CV does not change
    assert m.cv(d_pos, abi.MTD_F64, box, N) == pytest.approx(s_gpu, rel=1e-14)
    assert r.cv(pt, rbox) == pytest.approx(s_ref, rel=1e-14)
    abi.check(lib.mtd_mesh_virial(m.h, C.byref(box), N, 0.8, vir.ctypes.data_as(C.POINTER(C.c_double)), None))
    v_ref = r.virial(N, 0.8)
    assert np.abs(v_ref).max() > 0
    # components that vanish by symmetry are sums of cancelling terms (rounding noise, summation-order dependent): the
    # absolute tolerance is set by the largest component
    assert np.allclose(vir, v_ref, rtol=1e-9, atol=1e-9 * np.abs(v_ref).max())
    assert lib.mtd_mesh_set_table(m.h, util.dbl_array(K), util.dbl_array(dK), npts, 2.0, 1.0) == -1   # MTD_ERR_INVALID_ARGUMENT
    m.close()


def test_mesh_particles_on_cell_boundaries(abi, ref, monkeypatch):
    """particles sitting exactly on cell faces / corners (in-cell shift = -1/2 up to the rounding of an fp32 position): the
    force sums differences of neighbouring mesh rows, which magnifies any O(1e-7) liberty in the TSC derivative there a
    thousandfold (regression: closed forms applied a hair beyond |shift| = 1/2, found by tools/fuzz_mesh.py).  The two
    pipelines share their shifts and must agree to the rounding of the fp32 force array, and so must the oracle: the shift is
    formed with the reference's operation order (OrderParameterMesh.cc:565-573), so Q9's float rounding of |x| inside the
    derivative resolves identically."""
    dims, Ls = (64, 20, 24), (12.3082284, 9.7, 6.6019954)
    rng = np.random.default_rng(3)
    N = 20000
    f = rng.random((N, 3))
    f[:2000] = np.round(f[:2000] * np.array(dims)) / np.array(dims)          # on cell corners
    f[2000:4000, 2] = np.round(f[2000:4000, 2] * dims[2]) / dims[2]          # on z faces
    f[0] = [0.0, 0.0, 0.0]; f[1] = [0.999999999, 0.5, 0.0]; f[2] = [0.5, 0.0, 0.999999999]
    pos = ((f - 0.5) * np.array(Ls)).astype(np.float32)
    types = rng.integers(0, 2, N).astype(np.int32)
    mode = [1.0, -0.6]
    box, rbox = abi.Box.make(Ls), ref.Box.make(Ls)
    d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
    opt = util.oracle_postype(pos, types)
    r = ref.Mesh(*dims, mode)
    s_ref = r.cv(opt, rbox)
    F_ref = r.forces(opt, rbox, 0.8)
    F = {}
    for path in ("tiles", "cells"):
        monkeypatch.setenv("MTD_MESH_ASSIGN", path)
        g = GpuMesh(abi, dims, mode, N)
        try:
            assert g.cv(d_pos, abi.MTD_F32, box, N) == pytest.approx(s_ref, rel=1e-9)
            F[path] = g.forces(d_pos, abi.MTD_F32, box, N, 0.8)
        finally:
            g.close()
    fm = np.abs(F_ref).max()
    assert np.abs(F["tiles"] - F["cells"]).max() <= 5e-7 * fm
    # every particle, no exemption: the shift is bit-identical to the reference's (same operations, same order), so the
    # float rounding of |x| in the derivative (Q9) falls the same way
    per = np.abs(F["tiles"][:, :3] - F_ref[:, :3]).max(axis=1) / fm
    assert per.max() <= 5e-7, (per.max(), int(per.argmax()))


def test_mesh_forces_every_particle_in_random_triclinic_boxes(abi, ref):
    """Q9 regression (round 2): 40 seeded random meshes (sizes with and without powers of two) in random TRICLINIC boxes, float32 and
    float64 positions — every particle's force against the oracle within the tolerance, no counted exemption.  A randomised
    campaign found one particle off by 1.2e-3 of max|F| in such a box while `-ffp-contract=fast` let the back end fuse the
    multiply-adds of mesh.hip::locate in spite of its `fp contract(off)` pragma (csrc/Makefile now builds mesh.hip with
    -ffp-contract=on): the in-cell shift must be the reference's double, bit for bit (OrderParameterMesh.cc:540-573)."""
    rng = np.random.default_rng(2024)
    DIMS = [4, 5, 6, 8, 9, 12, 16, 20, 24, 32]
    worst = 0.0
    for case in range(40):
        dims = tuple(int(rng.choice(DIMS)) for _ in range(3))
        N = 4000
        Ls = tuple(float(x) for x in rng.uniform(3.0, 15.0, 3))
        tilt = dict(xy=float(rng.uniform(-0.3, 0.3)), xz=float(rng.uniform(-0.3, 0.3)), yz=float(rng.uniform(-0.3, 0.3)))
        dtype = np.float32 if case % 2 else np.float64
        mode = [float(x) for x in rng.uniform(-1.5, 1.5, 2)]
        f = rng.random((N, 3))
        a1 = np.array([Ls[0], 0, 0]); a2 = np.array([tilt["xy"] * Ls[1], Ls[1], 0])
        a3 = np.array([tilt["xz"] * Ls[2], tilt["yz"] * Ls[2], Ls[2]])
        pos = (-0.5 * np.array(Ls) + f[:, :1] * a1 + f[:, 1:2] * a2 + f[:, 2:3] * a3).astype(dtype)
        types = rng.integers(0, 2, N).astype(np.int32)
        box, rbox = abi.Box.make(Ls, **tilt), ref.Box.make(Ls, **tilt)
        dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
        d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda()
        opt = util.oracle_postype(pos, types)
        g = GpuMesh(abi, dims, mode, N)
        r = ref.Mesh(*dims, mode)
        try:
            s = g.cv(d_pos, dt, box, N)
            s_ref = r.cv(opt, rbox)
            assert s == pytest.approx(s_ref, rel=1e-8), (case, dims)
            F = g.forces(d_pos, dt, box, N, 0.8)
            F_ref = r.forces(opt, rbox, 0.8)
            fm = np.abs(F_ref[:, :3]).max()
            per = np.abs(F[:, :3] - F_ref[:, :3]).max(axis=1) / fm
            worst = max(worst, per.max())
            assert per.max() <= (5e-7 if dtype == np.float32 else 1e-8), (case, dims, tilt, per.max(), int(per.argmax()))
        finally:
            g.close()


def test_mesh_256_cubed_tile_path_against_cell_path(abi, monkeypatch):
    """the largest mesh the tile path takes (256^3: 8192 tiles, the count kernel's two LDS arrays of 8192 counters need the raised
    dynamic-LDS limit) against the cell-level pipeline: same mesh sums, CV and forces (no oracle at this size: its DFT is O(n^2)
    per line)"""
    dims, L, N = (256, 256, 256), 64.0, 200_000
    pos, types = util.snapshot_random(N, L, seed=11, modulated=True, dtype=np.float32)
    pos = (np.mod(pos.astype(np.float64) + L / 2, L) - L / 2).astype(np.float32)
    pos[pos >= L / 2] = -L / 2
    box = abi.Box.make(L)
    d_pos = torch.from_numpy(util.pack_postype(pos, types, np.float32)).cuda()
    out = {}
    for path in ("tiles", "cells"):
        monkeypatch.setenv("MTD_MESH_ASSIGN", path)
        g = GpuMesh(abi, dims, [1.0, -0.7], N)
        try:
            s = g.cv(d_pos, abi.MTD_F32, box, N)
            F = g.forces(d_pos, abi.MTD_F32, box, N, 0.8)
            out[path] = (s, g.array(0).copy(), F)
        finally:
            g.close()
    assert out["tiles"][0] == pytest.approx(out["cells"][0], rel=1e-9)
    scale = np.abs(out["cells"][1]).max()
    assert np.abs(out["tiles"][1] - out["cells"][1]).max() <= 1e-11 * scale
    fm = np.abs(out["cells"][2]).max()
    assert fm > 0 and np.abs(out["tiles"][2] - out["cells"][2]).max() <= 5e-7 * fm


@pytest.mark.parametrize("pipeline", ["bin", "counting"])
@pytest.mark.parametrize("dtype,fast", [(np.float32, 1), (np.float32, 0), (np.float64, 1)])
def test_lamellar_and_deferred_grid_pass_ride_in_the_binning_kernel(abi, ref, dtype, fast, pipeline):
    """mtd_mesh_set_lamellar_rider: the kernel that bins the particles for the mesh also forms the block partial sums of a set of
    lamellar CVs (LamellarOrderParameter.cc:143-179 beside OrderParameterMesh.cc:517-640: one pass over the positions) and the
    assignment carries the bias-grid engine's deferred second pass — in the bin pipeline (k_tile_bin while it waits for its atomics;
    the grid pass as extra blocks of the scatter launch) and in the counting pipeline of a mesh's first assignment (k_tile_count's
    particle loop; the row-scan launch).  The lamellar sums against the oracle (1e-6) and against the stand-alone CV pass (another
    grouping of the same fp32 terms: 1e-7), the mesh bit for bit what it is without riders, the grid arrays after the ridden
    deferred pass against the oracle's."""
    from test_gpu_metad import GpuMetad, compare
    lib = abi.load()
    N, L = 30011, 28.0
    pos, types = util.snapshot_random(N, L, seed=77, modulated=True, dtype=np.float32)
    pos = (np.mod(pos.astype(np.float64) + L / 2, L) - L / 2).astype(np.float32)
    pos[pos >= L / 2] = -L / 2
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    d_pos = torch.from_numpy(util.pack_postype(pos.astype(dtype), types, dtype)).cuda()
    box, rbox = abi.Box.make(L), ref.Box.make(L)
    cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)]
    lset = abi.LamellarSet.make(cvs)
    abi.check(lib.mtd_lamellar_set_fast_trig(fast))
    plain, ridden = GpuMesh(abi, (32, 32, 32), [1.0, -1.0], N), GpuMesh(abi, (32, 32, 32), [1.0, -1.0], N)
    kw = dict(sigma=[0.05], cv_min=[-1.0], cv_max=[1.0], num_points=[300], W=1.0, T_shift=7.0, T=1.0, stride=1, mode="well_tempered")
    g, r = GpuMetad(abi, **kw), ref.Metad(**kw)
    try:
        s_plain = plain.cv(d_pos, dt, box, N)
        rho_plain = plain.array(0).copy()
        if pipeline == "bin":
            assert ridden.cv(d_pos, dt, box, N) == s_plain       # the first assignment of a mesh counts and plans; the ridden one bins
        # a deposit through the generic entry point leaves its second grid pass pending ...
        g.step(0, [0.3])
        b_ref = r.update_bias(0, [0.3])
        # ... which rides, with the lamellar sums, in the mesh's binning kernel
        partials = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
        n_part = C.c_uint()
        abi.check(lib.mtd_mesh_set_lamellar_rider(ridden.h, g.h, C.byref(lset), C.byref(box), N, abi.ptr(partials), C.byref(n_part), None))
        s_ridden = ridden.cv(d_pos, dt, box, N)
        was = C.c_int(-1)
        abi.check(lib.mtd_mesh_clear_rider(ridden.h, C.byref(was)))
        assert was.value == 0                                        # consumed by the assignment
        assert s_ridden == s_plain and np.array_equal(ridden.array(0), rho_plain)
        pl = C.c_int(-1)
        abi.check(lib.mtd_mesh_assign_info(ridden.h, C.byref(pl), None, None))
        assert pl.value == (2 if pipeline == "bin" else 1)
        sums = partials[: n_part.value * 2].cpu().numpy().reshape(n_part.value, 2).sum(axis=0) / N
        opt = util.oracle_postype(pos.astype(np.float64) if dtype == np.float64 else pos, types)
        s_ref = [ref.lamellar_cv(v, opt, m, rbox) for v, m in cvs]
        tol = [1e-6 * max(abs(s), 8 / np.sqrt(N)) for s in s_ref]
        assert abs(sums[0] - s_ref[0]) <= tol[0] and abs(sums[1] - s_ref[1]) <= tol[1], (sums, s_ref)
        scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
        n2 = C.c_uint()
        abi.check(lib.mtd_lamellar_cv_partials(C.byref(lset), N, abi.ptr(d_pos), dt, C.byref(box), abi.ptr(scratch), C.byref(n2), None))
        alone = scratch[: n2.value * 2].cpu().numpy().reshape(n2.value, 2).sum(axis=0) / N
        assert np.allclose(sums, alone, rtol=1e-7, atol=1e-7 * 8 / np.sqrt(N))
        # the deferred pass ran as a passenger: the arrays are final WITHOUT another launch (nothing is pending any more)
        compare(g, r, b_ref, label="after the ridden deferred pass")
        # riders are one-shot: the next assignment carries nothing, and an armed rider nobody consumes is reported
        partials.zero_()
        assert ridden.cv(d_pos, dt, box, N) == s_plain and float(partials.abs().sum()) == 0.0
        abi.check(lib.mtd_mesh_set_lamellar_rider(ridden.h, None, C.byref(lset), C.byref(box), N, abi.ptr(partials), C.byref(n_part), None))
        abi.check(lib.mtd_mesh_clear_rider(ridden.h, C.byref(was)))
        assert was.value == 1
        assert ridden.cv(d_pos, dt, box, N) == s_plain and float(partials.abs().sum()) == 0.0
        # a second deposit + ridden pass, this time with an engine only partly busy (no pending pass: no apply blocks)
        abi.check(lib.mtd_mesh_set_lamellar_rider(ridden.h, g.h, C.byref(lset), C.byref(box), N, abi.ptr(partials), C.byref(n_part), None))
        assert ridden.cv(d_pos, dt, box, N) == s_plain
        sums2 = partials[: n_part.value * 2].cpu().numpy().reshape(n_part.value, 2).sum(axis=0) / N
        assert np.array_equal(sums2, sums)
        g.step(1, [0.31])
        compare(g, r, r.update_bias(1, [0.31]), label="second deposit")
    finally:
        abi.check(lib.mtd_lamellar_set_fast_trig(0))
        for x in (plain, ridden, g):
            x.close()


@pytest.mark.parametrize("dtype,fast", [(np.float32, 1), (np.float32, 0), (np.float64, 1)])
@pytest.mark.parametrize("n_lam,stride", [(1, 1), (2, 3), (0, 2)])
def test_force_pass_with_the_bias_update_inside(abi, ref, dtype, fast, n_lam, stride):
    """mtd_mesh_forces_update_bias (k_tile_forces_chain: the bias-grid engine's launch — scalar chain, first grid pass, lamellar
    forces — inside the mesh's force pass) against mtd_fused_force_pass_slots + mtd_mesh_forces on an identical engine, over a
    trajectory with deposit and non-deposit steps: CV values the same bits; bias factors, V and the grid arrays to 1e-12 (the two
    forms are compiled with different contraction settings); lamellar forces to one fp32 rounding, mesh forces to 1e-12 of the
    largest (their bias factor differs by as much); and the mesh forces of the last step against the ORACLE with the oracle's own
    bias factor from the oracle's grid driven with the device's CV values.  n_lam = 0: the mesh variable alone on the grid (no
    lamellar set: mtd_metad_update_bias + mtd_mesh_forces in one launch)."""
    from test_gpu_metad import compare
    lib = abi.load()
    N, L, T = 60013, 30.0, 6
    dt = abi.MTD_F32 if dtype == np.float32 else abi.MTD_F64
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    box, rbox = abi.Box.make(L), ref.Box.make(L)
    cvs = [(util.CV1_VECTORS, util.MODE_AB), (util.CV2_VECTORS, util.MODE_AB)][:n_lam]
    lset = abi.LamellarSet.make(cvs) if n_lam else None
    n_var = n_lam + 1
    mesh_slot = 1 if n_lam == 2 else 0                                   # the mesh BETWEEN the lamellar variables / in front of it
    lam_slots = [s for s in range(n_var) if s != mesh_slot]
    slots = (C.c_uint * n_lam)(*lam_slots) if n_lam else None
    dbl = lambda v: (C.c_double * len(v))(*[float(x) for x in v])
    abi.check(lib.mtd_lamellar_set_fast_trig(fast))
    rng = np.random.default_rng(5)
    base, types = util.snapshot_random(N, L, seed=321, modulated=True, dtype=np.float64)
    meshes = [GpuMesh(abi, (128, 32, 16), util.MODE_AB, N) for _ in range(2)]       # 2 x 4 x 2 tiles of 64 x 8 x 8
    # grids: the mesh CV's value is not known in advance — one untimed evaluation supplies it
    d0 = torch.from_numpy(util.pack_postype(base.astype(dtype), types, dtype)).cuda()
    s_mesh0 = meshes[0].cv(d0, dt, box, N)
    sig = [0.05] * n_var
    lo, hi, npts = [-1.0] * n_var, [1.0] * n_var, [16, 12, 10][:n_var]          # (at most as many 256-cell grid blocks as the mesh has tiles)
    sig[mesh_slot], lo[mesh_slot], hi[mesh_slot] = 0.05 * s_mesh0, 0.0, 2.0 * s_mesh0

    def engine():
        h = C.c_void_p()
        abi.check(lib.mtd_metad_create(C.byref(h), n_var, dbl(sig), dbl(lo), dbl(hi), (C.c_uint * n_var)(*npts), 1.0, 7.0, 1.0, stride, 1, 1))
        return h

    ha, hb = engine(), engine()
    r = ref.Metad(sigma=sig, cv_min=lo, cv_max=hi, num_points=npts, W=1.0, T_shift=7.0, T=1.0, stride=stride, mode="well_tempered")
    scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(N), dtype=torch.float64, device="cuda")
    fa = [torch.zeros((N, 4), dtype=tdt, device="cuda") for _ in range(n_lam + 1)]
    fb = [torch.zeros((N, 4), dtype=tdt, device="cuda") for _ in range(n_lam + 1)]
    pa = (C.c_void_p * n_lam)(*[f.data_ptr() for f in fa[:n_lam]]) if n_lam else None
    pb = (C.c_void_p * n_lam)(*[f.data_ptr() for f in fb[:n_lam]]) if n_lam else None
    p_set = C.byref(lset) if n_lam else None
    try:
        for t in range(T):
            pos = base + rng.normal(0.0, 0.02 * t, size=base.shape)
            if t == T - 2: pos = base * 0.3 - 0.2 * L                    # everything in a corner: most of it overflows the plan (overflow list)
            pos = np.mod(pos + L / 2, L) - L / 2
            pos = pos.astype(dtype)
            pos[pos >= L / 2] = -L / 2
            d_pos = torch.from_numpy(util.pack_postype(pos, types, dtype)).cuda()
            n_part = C.c_uint()
            if n_lam:
                abi.check(lib.mtd_lamellar_cv_partials(p_set, N, d_pos.data_ptr(), dt, C.byref(box), scratch.data_ptr(), C.byref(n_part), None))
            for h, mesh in ((ha, meshes[0]), (hb, meshes[1])):
                parts, n = C.c_void_p(), C.c_uint()
                abi.check(lib.mtd_mesh_compute_cv(mesh.h, N, d_pos.data_ptr(), dt, C.byref(box), N, C.byref(parts), C.byref(n), None))
                abi.check(lib.mtd_metad_set_cv_source(h, mesh_slot, parts.value, n.value, 1, 0, 0.5, 0.0))
                for c, slot in enumerate(lam_slots):
                    abi.check(lib.mtd_metad_set_cv_source(h, slot, scratch.data_ptr(), n_part.value, n_lam, c, 1.0 / N, 0.0))
            if t == T - 2:
                assert assign_info(abi, meshes[0])[1] > 0 and assign_info(abi, meshes[1])[1] > 0
            # A: one launch
            abi.check(lib.mtd_mesh_forces_update_bias(meshes[0].h, ha, mesh_slot, p_set, slots, N, d_pos.data_ptr(), fa[n_lam].data_ptr(), pa,
                                                      dt, N, C.byref(box), t, None))
            # B: the engine's launch with the lamellar forces, then the mesh's force pass with the device bias factor
            if n_lam:
                abi.check(lib.mtd_fused_force_pass_slots(hb, p_set, slots, N, d_pos.data_ptr(), pb, dt, N, C.byref(box), t, None))
            else:
                abi.check(lib.mtd_metad_update_bias(hb, t, None))
            d_bias = lib.mtd_metad_bias_device(hb)
            abi.check(lib.mtd_mesh_forces(meshes[1].h, N, d_pos.data_ptr(), fb[n_lam].data_ptr(), dt, C.byref(box), N, d_bias + 8 * mesh_slot, 0.0, None))
            torch.cuda.synchronize()
            cv_a, cv_b, ba, bb = (C.c_double * n_var)(), (C.c_double * n_var)(), (C.c_double * n_var)(), (C.c_double * n_var)()
            Va, Vb = C.c_double(), C.c_double()
            abi.check(lib.mtd_metad_get_state(ha, cv_a, ba, C.byref(Va), None, None, None, None))
            abi.check(lib.mtd_metad_get_state(hb, cv_b, bb, C.byref(Vb), None, None, None, None))
            assert list(cv_a) == list(cv_b)
            assert np.allclose(list(ba), list(bb), rtol=1e-11, atol=1e-300) and Va.value == pytest.approx(Vb.value, rel=1e-12, abs=1e-300)
            b_ref = r.update_bias(t, list(cv_a))
            assert np.allclose(list(ba), b_ref, rtol=1e-7, atol=1e-9 * np.abs(b_ref).max())
            for c in range(n_lam + 1):
                A, B = fa[c].cpu().numpy().astype(np.float64), fb[c].cpu().numpy().astype(np.float64)
                if t > 0 and t != T - 2: assert np.abs(B).max() > 0        # (the corner snapshot may leave the grid: zero bias is legitimate)
                tol = 1e-11 if c == n_lam and dtype == np.float64 else 2e-6
                assert np.abs(A - B).max() <= tol * max(np.abs(B).max(), 1e-300), (t, c)
        for name in ("grid", "reweighted", "hist", "sigma_grid"):
            which = abi.ARRAY_NAMES.index(name)
            n_cells = int(np.prod(npts))
            oa = np.zeros(n_cells, dtype=np.float64 if which < 6 else np.uint32)
            ob = oa.copy()
            abi.check(lib.mtd_metad_get_array(ha, which, oa.ctypes.data, None))
            abi.check(lib.mtd_metad_get_array(hb, which, ob.ctypes.data, None))
            assert np.allclose(oa, ob, rtol=1e-11, atol=1e-14 * max(np.abs(ob).max(), 1e-300), equal_nan=True), name
            assert np.abs(ob.astype(np.float64)).max() > 0
        # the mesh forces of the last step against the oracle (its own bias factor from its own grid)
        rm = ref.Mesh(128, 32, 16, util.MODE_AB)
        opt = util.oracle_postype(pos, types)
        rm.cv(opt, rbox)
        sl = slice(0, 20000)
        F_ref = rm.forces(opt[sl], rbox, b_ref[mesh_slot], n_global=N)
        F = fa[n_lam].cpu().numpy().astype(np.float64)
        assert np.abs(F_ref[:, :3]).max() > 0
        assert np.abs(F[sl, :3] - F_ref[:, :3]).max() <= (1e-5 if dtype == np.float32 else 1e-7) * np.abs(F_ref[:, :3]).max()
        # refusal: a mesh slot that is a lamellar slot
        if n_lam:
            assert lib.mtd_mesh_forces_update_bias(meshes[0].h, ha, lam_slots[0], p_set, slots, N, d_pos.data_ptr(), fa[n_lam].data_ptr(), pa,
                                                   dt, N, C.byref(box), T, None) == -1
    finally:
        abi.check(lib.mtd_lamellar_set_fast_trig(0))
        for m in meshes:
            m.close()
        for h in (ha, hb):
            abi.check(lib.mtd_metad_destroy(h))
