"""GPU: the kernels' own arithmetic against the only numbers in this repository that came out of reference code — the
fixtures tests/golden/*.json, produced by the reference's spherical_harmonics.hpp and IndexGrid.cc compiled where they lie
(oracle/_ref, tests/golden/make_golden.py).  The oracle is pinned to them on the CPU (test_oracle_steinhardt.py,
test_oracle_kat.py); here the DEVICE code meets them directly, not by transitivity through the oracle:
  * the Y_lm evaluation of the Steinhardt pair kernels (unit-vector trigonometry from a separation, Jacobi recurrence with folded
    prefactors, e^{i m phi} by repeated multiplication) through mtd_debug_sph_harmonics — 1e-12;
  * IndexGrid::getCoordinates / getIndex as the grid kernels compute them through mtd_debug_index_decode — bit-exact."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_device_spherical_harmonics_against_reference_vectors(abi):
    lib = abi.load()
    d = json.load(open(os.path.join(GOLDEN, "sph_lmax6.json")))
    lmax = d["lmax"]
    polar, azim = np.array(d["polar"]), np.array(d["azimuth"])
    want = np.array(d["values"])                                  # [n][(lmax+1)^2][re, im], fsph's order
    n = len(polar)
    for r in (1.0, 0.37):                                         # the kernels divide the separation by its length themselves
        sep = np.ascontiguousarray(r * np.stack([np.sin(polar) * np.cos(azim), np.sin(polar) * np.sin(azim), np.cos(polar)], axis=1))
        out = np.zeros((n, (lmax + 1) ** 2, 2))
        abi.check(lib.mtd_debug_sph_harmonics(lmax, n, sep.ctypes.data, out.ctypes.data))
        # absolute 1e-12 on values of order one (sin(theta)^m amplitudes of 1e-19 at theta = 1e-3 are compared relatively too)
        assert np.allclose(out, want, rtol=1e-9, atol=1e-12), np.abs(out - want).max()
        big = np.abs(want) > 1e-3
        assert np.abs(out[big] / want[big] - 1.0).max() < 1e-12
    # degrees below the template's capacity (lmax 5 runs in the LMAX = 6 instantiation, 3 in LMAX = 4) are the same numbers
    for lm in (3, 5):
        out = np.zeros((n, (lm + 1) ** 2, 2))
        abi.check(lib.mtd_debug_sph_harmonics(lm, n, sep.ctypes.data, out.ctypes.data))
        assert np.allclose(out, want[:, :(lm + 1) ** 2], rtol=1e-9, atol=1e-12)


def test_device_index_grid_against_reference_tables(abi):
    lib = abi.load()
    d = json.load(open(os.path.join(GOLDEN, "index_grid.json")))
    assert d["cases"]
    for case in d["cases"]:
        lengths = np.array(case["lengths"], dtype=np.uint32)
        idx = np.array(case["indices"], dtype=np.uint32)
        want = np.array(case["coords"], dtype=np.uint32)
        coords = np.zeros((len(idx), len(lengths)), dtype=np.uint32)
        back = np.zeros(len(idx), dtype=np.uint32)
        abi.check(lib.mtd_debug_index_decode(len(lengths), lengths.ctypes.data, len(idx), idx.ctypes.data, coords.ctypes.data, back.ctypes.data))
        assert np.array_equal(coords, want), (case["lengths"], coords, want)          # getCoordinates (IndexGrid.cc:46-58)
        assert np.array_equal(back, idx)                                              # getIndex (:20-44)
