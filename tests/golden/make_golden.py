#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference's OWN sources.

Runs only where /root/reference exists: oracle/_ref/libmtd_refsrc.so is IndexGrid.cc and
spherical_harmonics.hpp compiled where they lie (oracle/Makefile target `_ref`).  The fixtures are
data (inputs + expected outputs); no reference source text is stored.

    python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mtd_ref  # noqa: E402

UP = C.POINTER(C.c_uint)
DP = C.POINTER(C.c_double)


def index_grid(src):
    rng = np.random.default_rng(20240)
    cases = []
    for lengths in ([128], [20, 30], [256, 256], [9, 7, 8], [3, 4, 5, 6]):
        l = np.array(lengths, dtype=np.uint32)
        n = int(np.prod(l))
        idxs = sorted(set([0, n - 1] + [int(x) for x in rng.integers(0, n, 12)]))
        coords = []
        for idx in idxs:
            c = np.zeros(len(l), dtype=np.uint32)
            src.refsrc_index_coords(len(l), l.ctypes.data_as(UP), idx, c.ctypes.data_as(UP))
            assert src.refsrc_index_get(len(l), l.ctypes.data_as(UP), c.ctypes.data_as(UP)) == idx
            coords.append([int(x) for x in c])
        cases.append(dict(lengths=lengths, indices=idxs, coords=coords))
    return dict(source="IndexGrid.cc:20-58 via oracle/_ref", cases=cases)


def sph(src):
    """fsph::evaluate_SPH<double>(lmax=6, full_m=true) at seeded angles; argument order as at the
    reference call site SteinhardtQl.cc:143 (phi = polar angle, theta = azimuth)."""
    rng = np.random.default_rng(606)
    n = 24
    polar = np.concatenate([[0.3, 1e-3, np.pi - 1e-3, np.pi / 2], rng.uniform(0.05, np.pi - 0.05, n - 4)])
    azim = np.concatenate([[0.0, 1.0, -2.0, np.pi], rng.uniform(-np.pi, np.pi, n - 4)])
    lmax = 6
    per = (lmax + 1) ** 2
    out = np.zeros(2 * per * n)
    src.refsrc_evaluate_sph(out.ctypes.data_as(DP), lmax, polar.ctypes.data_as(DP), azim.ctypes.data_as(DP), n, 1)
    return dict(source="spherical_harmonics.hpp:229-246 via oracle/_ref", lmax=lmax, full_m=True,
                polar=polar.tolist(), azimuth=azim.tolist(), values=out.reshape(n, per, 2).tolist())


def main():
    mtd_ref.build()
    src = mtd_ref.refsrc()
    if src is None:
        raise SystemExit("needs /root/reference (oracle/_ref)")
    json.dump(index_grid(src), open(os.path.join(HERE, "index_grid.json"), "w"))
    json.dump(sph(src), open(os.path.join(HERE, "sph_lmax6.json"), "w"))
    print("wrote index_grid.json, sph_lmax6.json")


if __name__ == "__main__":
    main()
