"""Shared test helpers: seeded particle snapshots and tensor packing."""
import ctypes as C

import numpy as np

# lattice vectors of the headline config
CV1_VECTORS = [(0, 0, 3), (0, 3, 0), (3, 0, 0), (0, 0, 6), (0, 6, 0), (6, 0, 0), (3, 3, 0), (0, 3, 3)]
CV2_VECTORS = [(1, 1, 1), (1, -1, 1), (1, 1, -1), (-1, 1, 1), (2, 2, 2), (2, -2, 2), (2, 2, -2), (-2, 2, 2)]
MODE_AB = [1.0, -1.0]


def snapshot_config0b():
    """N=4096: 16^3 simple-cubic lattice + U(-0.1,0.1) noise, lamellae of period 4a along z."""
    rng = np.random.default_rng(2024)
    n = 16
    ix, iy, iz = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    pos = np.stack([ix, iy, iz], axis=-1).reshape(-1, 3).astype(np.float64) - n / 2  # lattice sites at lo + i*a
    pos += rng.uniform(-0.1, 0.1, size=pos.shape)
    types = (iz.reshape(-1) % 4 >= 2).astype(np.int32)  # A (0) if iz mod 4 < 2 else B (1)
    return pos, types, float(n)


def snapshot_random(N, L, seed=12345, modulated=False, dtype=np.float32):
    """config 2 style: uniform random positions, types = index parity; optional lamellar modulation
    z <- z + 1.5 sign(a) sin(2 pi 3 z / L) so that |s_1| = O(0.1)."""
    rng = np.random.default_rng(seed)
    pos = rng.random((N, 3)) * L - L / 2
    types = (np.arange(N) % 2).astype(np.int32)
    if modulated:
        a = np.where(types == 0, 1.0, -1.0)
        pos[:, 2] = pos[:, 2] + 1.5 * a * np.sin(2 * np.pi * 3 * pos[:, 2] / L)
    pos = pos.astype(dtype)  # the snapshot IS the rounded array; the oracle sees the same values
    return pos, types


def pack_postype(pos, types, dtype):
    """HOOMD Scalar4 postype as a numpy array: float32 (N,4) with the int32 type bit-cast into w,
    or float64 (N,4) with the type in the LOW 32 bits of w."""
    N = pos.shape[0]
    if dtype == np.float32:
        out = np.empty((N, 4), dtype=np.float32)
        out[:, :3] = pos
        out[:, 3] = np.asarray(types, dtype=np.int32).view(np.float32)
    else:
        out = np.empty((N, 4), dtype=np.float64)
        out[:, :3] = pos
        w = np.zeros(N, dtype=np.int64)
        w[:] = np.asarray(types, dtype=np.int64) & 0xFFFFFFFF
        out[:, 3] = w.view(np.float64)
    return out


def oracle_postype(pos, types):
    out = np.empty((pos.shape[0], 4), dtype=np.float64)
    out[:, :3] = pos.astype(np.float64)
    out[:, 3] = types
    return out


def flat_lattice(vectors):
    return (C.c_int * (3 * len(vectors)))(*[int(x) for v in vectors for x in v])


def dbl_array(values):
    return (C.c_double * len(values))(*[float(v) for v in values])


def uint_array(values):
    return (C.c_uint * len(values))(*[int(v) for v in values])


def build_nlist(pos, L, r_cut, half=False):
    """HOOMD-layout neighbour list (head_list, n_neigh, nlist) of a cubic periodic box with scipy's periodic KD-tree:
    the stand-in for hoomd.md.nlist in tests (HOOMD's NeighborList is not part of the plugin)."""
    from scipy.spatial import cKDTree
    p = np.mod(np.asarray(pos, dtype=np.float64) + L / 2, L)
    p[p >= L] -= L
    tree = cKDTree(p, boxsize=L)
    pairs = tree.query_pairs(r_cut, output_type="ndarray")
    N = len(p)
    if half:
        i, j = pairs[:, 0], pairs[:, 1]
    else:
        i = np.concatenate([pairs[:, 0], pairs[:, 1]])
        j = np.concatenate([pairs[:, 1], pairs[:, 0]])
    order = np.lexsort((j, i))
    i, j = i[order], j[order]
    n_neigh = np.bincount(i, minlength=N).astype(np.uint32)
    head = np.zeros(N, dtype=np.uint32)
    head[1:] = np.cumsum(n_neigh)[:-1]
    return head, n_neigh, j.astype(np.uint32)


def fcc_lattice(n, nn_dist=1.0):
    """n^3 fcc cells x 4 particles, nearest-neighbour distance nn_dist; returns positions centred on 0 and box length"""
    a = nn_dist * np.sqrt(2.0)
    basis = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
    cells = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + basis[None, :, :]).reshape(-1, 3) * a
    L = n * a
    return pos - L / 2 + 0.25 * a, L
