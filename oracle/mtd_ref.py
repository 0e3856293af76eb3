"""ctypes front-end of the CPU oracle (oracle/_build/libmtd_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package.  See oracle/mtd_ref.h for the parity-pinning
statement and the reference file:line each function restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmtd_ref.so")
_REFSRC_PATH = os.path.join(_HERE, "_ref", "libmtd_refsrc.so")


def build(force=False):
    """Compile the restatement (and, when /root/reference is present, oracle/_ref)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    if os.path.isdir("/root/reference/metadynamics") and (force or not os.path.exists(_REFSRC_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_ref"])


class Box(C.Structure):
    _fields_ = [("L", C.c_double * 3), ("lo", C.c_double * 3),
                ("xy", C.c_double), ("xz", C.c_double), ("yz", C.c_double)]

    @classmethod
    def make(cls, L, lo=None, xy=0.0, xz=0.0, yz=0.0):
        L = [float(L)] * 3 if np.isscalar(L) else [float(x) for x in L]
        if lo is None:
            lo = [-0.5 * (L[0] + 0.0), -0.5 * L[1], -0.5 * L[2]]
        b = cls()
        b.L[:] = L
        b.lo[:] = [float(x) for x in lo]
        b.xy, b.xz, b.yz = float(xy), float(xz), float(yz)
        return b


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_uint)


def _d(a):
    return a.ctypes.data_as(_dp)


def _u(a):
    return a.ctypes.data_as(_up)


def _i(a):
    return a.ctypes.data_as(_ip)


_lib = None
_OMP_LIB_PATH = os.path.join(_HERE, "_build", "libmtd_ref_omp.so")
_use_omp = False


def use_openmp(enable):
    """bench.py only: switch to the build with OpenMP over the particles of the lamellar loops (the "all cores" CPU
    baseline).  The checker used by the tests is always the serial build."""
    global _lib, _use_omp
    if bool(enable) != _use_omp:
        _use_omp = bool(enable)
        _lib = None


def lib():
    global _lib
    if _lib is None:
        path = _OMP_LIB_PATH if _use_omp else _LIB_PATH
        if not os.path.exists(path):
            build(force=_use_omp)
        L = C.CDLL(path)
        L.ref_index_get.restype = C.c_uint
        L.ref_index_get.argtypes = [C.c_uint, _up, _up]
        L.ref_index_coords.restype = None
        L.ref_index_coords.argtypes = [C.c_uint, _up, C.c_uint, _up]
        L.ref_lamellar_fourier_modes.restype = None
        L.ref_lamellar_fourier_modes.argtypes = [C.c_uint, _ip, C.c_uint, _dp, _dp, C.POINTER(Box), _dp]
        L.ref_lamellar_cv.restype = C.c_double
        L.ref_lamellar_cv.argtypes = [C.c_uint, _dp, C.c_uint]
        L.ref_lamellar_forces.restype = None
        L.ref_lamellar_forces.argtypes = [C.c_uint, _ip, C.c_uint, _dp, _dp, C.POINTER(Box), C.c_uint, C.c_double, _dp]
        L.ref_metad_create.restype = C.c_void_p
        L.ref_metad_create.argtypes = [C.c_uint, _dp, _dp, _dp, _up, C.c_double, C.c_double, C.c_double,
                                       C.c_uint, C.c_int, C.c_int]
        L.ref_metad_destroy.argtypes = [C.c_void_p]
        L.ref_metad_num_elements.restype = C.c_uint
        L.ref_metad_num_elements.argtypes = [C.c_void_p]
        for name, extra in (("ref_metad_set_stride", [C.c_uint]), ("ref_metad_set_add_bias", [C.c_int]),
                            ("ref_metad_set_mode", [C.c_int]), ("ref_metad_set_sigma_inv", [_dp]),
                            ("ref_metad_reset_histogram", [])):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_void_p] + extra
        L.ref_metad_update_bias.restype = None
        L.ref_metad_update_bias.argtypes = [C.c_void_p, C.c_uint, _dp, _dp]
        L.ref_metad_update_phase_a.restype = C.c_int
        L.ref_metad_update_phase_a.argtypes = [C.c_void_p, C.c_uint, _dp]
        L.ref_metad_update_phase_b.restype = None
        L.ref_metad_update_phase_b.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.ref_metad_interpolate.restype = C.c_double
        L.ref_metad_interpolate.argtypes = [C.c_void_p, _dp, C.c_int]
        L.ref_metad_derivative.restype = C.c_double
        L.ref_metad_derivative.argtypes = [C.c_void_p, C.c_uint, _dp]
        for name in ("ref_metad_sigma_determinant", "ref_metad_curr_bias", "ref_metad_curr_weight"):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("ref_metad_num_gaussians", "ref_metad_num_oob_warnings"):
            getattr(L, name).restype = C.c_uint
            getattr(L, name).argtypes = [C.c_void_p]
        L.ref_metad_array.restype = C.c_void_p
        L.ref_metad_array.argtypes = [C.c_void_p, C.c_int]
        L.ref_metad_write_grid.restype = C.c_int
        L.ref_metad_write_grid.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.POINTER(C.c_char_p)]
        L.ref_metad_read_grid.restype = C.c_int
        L.ref_metad_read_grid.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_update_grid.restype = None
        L.ref_update_grid.argtypes = [C.c_uint, _up, _dp, _dp, _dp, _dp, C.c_double, C.c_double, _dp]
        L.ref_umbrella_bias.restype = C.c_double
        L.ref_umbrella_bias.argtypes = [C.c_int] + [C.c_double] * 6
        L.ref_umbrella_energy.restype = C.c_double
        L.ref_umbrella_energy.argtypes = [C.c_int] + [C.c_double] * 5
        L.ref_aspect_ratio.restype = C.c_double
        L.ref_aspect_ratio.argtypes = [C.POINTER(Box), C.c_uint, C.c_uint]
        L.ref_density.restype = C.c_double
        L.ref_density.argtypes = [C.POINTER(Box), C.c_uint]
        L.ref_wte_potential_energy.restype = C.c_double
        L.ref_wte_potential_energy.argtypes = [C.c_uint, _dp, C.c_double]
        L.ref_wte_scale.restype = None
        L.ref_wte_scale.argtypes = [C.c_uint, _dp, _dp, _dp, C.c_uint, _dp, C.c_double]
        L.ref_wrapper_energy.restype = C.c_double
        L.ref_wrapper_energy.argtypes = [C.c_uint, _dp, C.c_double]
        L.ref_wrapper_scale.restype = None
        L.ref_wrapper_scale.argtypes = [C.c_uint, _dp, _dp, _dp, C.c_uint, C.c_double]
        L.ref_compute_sigma.restype = None
        L.ref_compute_sigma.argtypes = [C.c_uint, C.c_uint, C.POINTER(_dp), C.POINTER(C.c_int), _dp, C.c_double, _dp, _dp]
        _bind_optional(L)
        _lib = L
    return _lib


def _bind_optional(L):
    """mesh / steinhardt entry points (bound when the library has them)."""
    if hasattr(L, "ref_mesh_create"):
        L.ref_mesh_create.restype = C.c_void_p
        L.ref_mesh_create.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, _dp]
        L.ref_mesh_destroy.argtypes = [C.c_void_p]
        L.ref_mesh_cv.restype = C.c_double
        L.ref_mesh_cv.argtypes = [C.c_void_p, C.c_uint, _dp, C.POINTER(Box), C.c_uint]
        L.ref_mesh_forces.restype = None
        L.ref_mesh_forces.argtypes = [C.c_void_p, C.c_uint, _dp, C.POINTER(Box), C.c_uint, C.c_double, _dp]
        L.ref_mesh_array.restype = C.c_void_p
        L.ref_mesh_array.argtypes = [C.c_void_p, C.c_int]
        L.ref_mesh_mode_sq.restype = C.c_double
        L.ref_mesh_mode_sq.argtypes = [C.c_void_p]
        L.ref_mesh_set_table.restype = C.c_int
        L.ref_mesh_set_table.argtypes = [C.c_void_p, _dp, _dp, C.c_uint, C.c_double, C.c_double]
        L.ref_mesh_set_use_table.restype = None
        L.ref_mesh_set_use_table.argtypes = [C.c_void_p, C.c_int]
        L.ref_mesh_qmax.restype = None
        L.ref_mesh_qmax.argtypes = [C.c_void_p, C.c_uint, _dp]
        L.ref_mesh_virial.restype = None
        L.ref_mesh_virial.argtypes = [C.c_void_p, C.c_uint, C.c_double, _dp]
        L.ref_mesh_assign.restype = None
        L.ref_mesh_assign.argtypes = [C.c_void_p, C.c_uint, _dp, C.POINTER(Box)]
        L.ref_mesh_set_mode_sq.restype = None
        L.ref_mesh_set_mode_sq.argtypes = [C.c_void_p, C.c_double]
        L.ref_mesh_spectral.restype = C.c_double
        L.ref_mesh_spectral.argtypes = [C.c_void_p, C.c_uint]
        L.ref_mesh_set_bug_compat.restype = None
        L.ref_mesh_set_bug_compat.argtypes = [C.c_void_p, C.c_int]
    if hasattr(L, "ref_sph_evaluate"):
        L.ref_sph_evaluate.restype = None
        L.ref_sph_evaluate.argtypes = [_dp, C.c_uint, _dp, _dp, C.c_uint, C.c_int]
    if hasattr(L, "ref_ql_compute_cv"):
        L.ref_ql_compute_cv.restype = C.c_double
        L.ref_ql_compute_cv.argtypes = [C.c_uint, _dp, C.POINTER(Box), _up, _up, _up, C.c_int,
                                        C.c_double, C.c_double, C.c_uint, C.c_uint, _dp, C.c_uint,
                                        _dp, _dp]
        L.ref_ql_from_qlm.restype = C.c_double
        L.ref_ql_from_qlm.argtypes = [C.c_uint, _dp, _dp, C.c_uint, _dp]
        L.ref_ql_compute_forces.restype = None
        L.ref_ql_compute_forces.argtypes = [C.c_uint, _dp, C.POINTER(Box), _up, _up, _up, C.c_int,
                                            C.c_double, C.c_double, C.c_uint, C.c_uint, _dp, C.c_uint,
                                            _dp, C.c_double, _dp]


_refsrc = None


def refsrc():
    """The reference's own IndexGrid.cc / spherical_harmonics.hpp (oracle/_ref), or None."""
    global _refsrc
    if _refsrc is None:
        if not os.path.exists(_REFSRC_PATH):
            if os.path.isdir("/root/reference/metadynamics"):
                build()
            else:
                return None
        L = C.CDLL(_REFSRC_PATH)
        L.refsrc_index_get.restype = C.c_uint
        L.refsrc_index_get.argtypes = [C.c_uint, _up, _up]
        L.refsrc_index_coords.restype = None
        L.refsrc_index_coords.argtypes = [C.c_uint, _up, C.c_uint, _up]
        L.refsrc_index_num_elements.restype = C.c_uint
        L.refsrc_index_num_elements.argtypes = [C.c_uint, _up]
        L.refsrc_evaluate_sph.restype = None
        L.refsrc_evaluate_sph.argtypes = [_dp, C.c_uint, _dp, _dp, C.c_uint, C.c_int]
        _refsrc = L
    return _refsrc


# ----------------------------------------------------------------------------- numpy helpers

def as_postype(pos, types):
    """(N,3) positions + (N,) integer types -> contiguous double (N,4) 'postype'."""
    pos = np.asarray(pos, dtype=np.float64)
    out = np.empty((pos.shape[0], 4), dtype=np.float64)
    out[:, :3] = pos
    out[:, 3] = np.asarray(types, dtype=np.float64)
    return out


def index_get(lengths, coords):
    l = np.ascontiguousarray(lengths, dtype=np.uint32)
    c = np.ascontiguousarray(coords, dtype=np.uint32)
    return lib().ref_index_get(len(l), _u(l), _u(c))


def index_coords(lengths, idx):
    l = np.ascontiguousarray(lengths, dtype=np.uint32)
    c = np.zeros(len(l), dtype=np.uint32)
    lib().ref_index_coords(len(l), _u(l), int(idx), _u(c))
    return c


def lamellar_fourier_modes(lattice, postype, mode, box):
    lat = np.ascontiguousarray(lattice, dtype=np.int32).reshape(-1, 3)
    pt = np.ascontiguousarray(postype, dtype=np.float64)
    md = np.ascontiguousarray(mode, dtype=np.float64)
    out = np.zeros((lat.shape[0], 2), dtype=np.float64)
    lib().ref_lamellar_fourier_modes(lat.shape[0], _i(lat), pt.shape[0], _d(pt), _d(md), C.byref(box), _d(out))
    return out


def lamellar_cv(lattice, postype, mode, box, n_global=None):
    modes = lamellar_fourier_modes(lattice, postype, mode, box)
    n_global = postype.shape[0] if n_global is None else n_global
    return lib().ref_lamellar_cv(modes.shape[0], _d(modes), int(n_global))


def lamellar_forces(lattice, postype, mode, box, bias, n_global=None):
    lat = np.ascontiguousarray(lattice, dtype=np.int32).reshape(-1, 3)
    pt = np.ascontiguousarray(postype, dtype=np.float64)
    md = np.ascontiguousarray(mode, dtype=np.float64)
    n_global = pt.shape[0] if n_global is None else n_global
    out = np.zeros((pt.shape[0], 4), dtype=np.float64)
    lib().ref_lamellar_forces(lat.shape[0], _i(lat), pt.shape[0], _d(pt), _d(md), C.byref(box),
                              int(n_global), float(bias), _d(out))
    return out


ARRAY_NAMES = ["grid", "grid_delta", "reweighted", "weight", "sigma_grid", "sigma_grid_delta",
               "hist", "hist_delta", "hist_gauss", "hist_gauss_delta"]


class Metad:
    """The bias-grid engine of IntegratorMetaDynamics.cc (grid mode, single rank)."""

    def __init__(self, sigma, cv_min, cv_max, num_points, W=1.0, T_shift=1.0, T=1.0, stride=1,
                 mode="standard", add_bias=True):
        self.n_cv = len(sigma)
        s = np.ascontiguousarray(sigma, dtype=np.float64)
        lo = np.ascontiguousarray(cv_min, dtype=np.float64)
        hi = np.ascontiguousarray(cv_max, dtype=np.float64)
        n = np.ascontiguousarray(num_points, dtype=np.uint32)
        self.num_points = n.copy()
        m = {"standard": 0, "well_tempered": 1}[mode]
        self._h = lib().ref_metad_create(self.n_cv, _d(s), _d(lo), _d(hi), _u(n), W, T_shift, T, stride, m,
                                         int(bool(add_bias)))
        if not self._h:
            raise RuntimeError("Error creating collective variable.")
        self.len = lib().ref_metad_num_elements(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ref_metad_destroy(self._h)
            self._h = None

    def update_bias(self, timestep, current_val):
        v = np.ascontiguousarray(current_val, dtype=np.float64)
        b = np.zeros(self.n_cv, dtype=np.float64)
        lib().ref_metad_update_bias(self._h, int(timestep), _d(v), _d(b))
        return b

    def phase_a(self, timestep, current_val):
        v = np.ascontiguousarray(current_val, dtype=np.float64)
        return lib().ref_metad_update_phase_a(self._h, int(timestep), _d(v))

    def phase_b(self, deposited, current_val):
        v = np.ascontiguousarray(current_val, dtype=np.float64)
        b = np.zeros(self.n_cv, dtype=np.float64)
        lib().ref_metad_update_phase_b(self._h, int(deposited), _d(v), _d(b))
        return b

    def interpolate(self, val, reweight=False):
        v = np.ascontiguousarray(val, dtype=np.float64)
        return lib().ref_metad_interpolate(self._h, _d(v), int(reweight))

    def derivative(self, cv, val):
        v = np.ascontiguousarray(val, dtype=np.float64)
        return lib().ref_metad_derivative(self._h, int(cv), _d(v))

    def array(self, name):
        """numpy *view* of an engine array (writes go through)."""
        which = ARRAY_NAMES.index(name)
        ptr = lib().ref_metad_array(self._h, which)
        ctype = C.c_double if which < 6 else C.c_uint
        buf = (ctype * self.len).from_address(ptr)
        return np.frombuffer(buf, dtype=np.float64 if which < 6 else np.uint32)

    def set_stride(self, s):
        lib().ref_metad_set_stride(self._h, int(s))

    def set_add_bias(self, b):
        lib().ref_metad_set_add_bias(self._h, int(bool(b)))

    def set_mode(self, mode):
        lib().ref_metad_set_mode(self._h, {"standard": 0, "well_tempered": 1}[mode])

    def set_sigma_inv(self, mat):
        a = np.ascontiguousarray(mat, dtype=np.float64).reshape(self.n_cv * self.n_cv)
        lib().ref_metad_set_sigma_inv(self._h, _d(a))

    def reset_histogram(self):
        lib().ref_metad_reset_histogram(self._h)

    @property
    def sigma_determinant(self):
        return lib().ref_metad_sigma_determinant(self._h)

    @property
    def curr_bias(self):
        return lib().ref_metad_curr_bias(self._h)

    @property
    def curr_weight(self):
        return lib().ref_metad_curr_weight(self._h)

    @property
    def num_gaussians(self):
        return lib().ref_metad_num_gaussians(self._h)

    @property
    def num_oob_warnings(self):
        return lib().ref_metad_num_oob_warnings(self._h)

    def write_grid(self, filename, timestep, names):
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        rc = lib().ref_metad_write_grid(self._h, filename.encode(), int(timestep), arr)
        if rc:
            raise RuntimeError("Error dumping grid.")

    def read_grid(self, filename):
        rc = lib().ref_metad_read_grid(self._h, filename.encode())
        if rc:
            raise RuntimeError("Error reading grid.")


def update_grid(lengths, cv_min, cv_max, sigma_inv, current_val, scal, W):
    l = np.ascontiguousarray(lengths, dtype=np.uint32)
    lo = np.ascontiguousarray(cv_min, dtype=np.float64)
    hi = np.ascontiguousarray(cv_max, dtype=np.float64)
    si = np.ascontiguousarray(sigma_inv, dtype=np.float64).reshape(-1)
    v = np.ascontiguousarray(current_val, dtype=np.float64)
    out = np.zeros(int(np.prod(l)), dtype=np.float64)
    lib().ref_update_grid(len(l), _u(l), _d(lo), _d(hi), _d(si), _d(v), float(scal), float(W), _d(out))
    return out


UMBRELLA = {"no_umbrella": 0, "linear": 1, "harmonic": 2, "wall": 3, "gaussian": 4}


def umbrella_bias(kind, val, bias_in, cv0, kappa, width_flat, scale):
    return lib().ref_umbrella_bias(UMBRELLA[kind], val, bias_in, cv0, kappa, width_flat, scale)


def umbrella_energy(kind, val, cv0, kappa, width_flat, scale):
    return lib().ref_umbrella_energy(UMBRELLA[kind], val, cv0, kappa, width_flat, scale)


def aspect_ratio(box, dir1, dir2):
    return lib().ref_aspect_ratio(C.byref(box), dir1, dir2)


def density(box, n_group):
    return lib().ref_density(C.byref(box), n_group)


def wte_potential_energy(net_force, external_energy=0.0):
    nf = np.ascontiguousarray(net_force, dtype=np.float64)
    return lib().ref_wte_potential_energy(nf.shape[0], _d(nf), float(external_energy))


def wte_scale(net_force, net_torque, net_virial, pitch, external_virial, bias):
    """In-place on copies; returns (net_force, net_torque, net_virial, external_virial)."""
    nf = np.array(net_force, dtype=np.float64, order="C")
    nt = np.array(net_torque, dtype=np.float64, order="C")
    nv = np.array(net_virial, dtype=np.float64, order="C")
    ev = np.array(external_virial, dtype=np.float64, order="C")
    lib().ref_wte_scale(nf.shape[0], _d(nf), _d(nt), _d(nv), int(pitch), _d(ev), float(bias))
    return nf, nt, nv, ev


def wrapper_energy(force, external_energy=0.0):
    f = np.ascontiguousarray(force, dtype=np.float64)
    return lib().ref_wrapper_energy(f.shape[0], _d(f), float(external_energy))


def wrapper_scale(force, torque, virial, pitch, bias):
    """In-place on copies; returns (force, torque, virial)."""
    f = np.array(force, dtype=np.float64, order="C")
    t = np.array(torque, dtype=np.float64, order="C")
    v = np.array(virial, dtype=np.float64, order="C")
    lib().ref_wrapper_scale(f.shape[0], _d(f), _d(t), _d(v), int(pitch), float(bias))
    return f, t, v


def compute_sigma(forces, can_derive, sigma, sigma_g):
    """IntegratorMetaDynamics::computeSigma; forces = list of (N,4) arrays; returns (sigmasq, sigma_inv) n_cv x n_cv"""
    n_cv = len(forces)
    arrs = [np.ascontiguousarray(f, dtype=np.float64) for f in forces]
    ptrs = (_dp * n_cv)(*[_d(a) for a in arrs])
    cd = (C.c_int * n_cv)(*[int(bool(c)) for c in can_derive])
    sg = np.ascontiguousarray(sigma, dtype=np.float64)
    sq = np.zeros(n_cv * n_cv)
    inv = np.zeros(n_cv * n_cv)
    lib().ref_compute_sigma(n_cv, arrs[0].shape[0], ptrs, cd, _d(sg), float(sigma_g), _d(sq), _d(inv))
    return sq.reshape(n_cv, n_cv), inv.reshape(n_cv, n_cv)


class Mesh:
    """OrderParameterMesh.cc, single rank (no ghost cells)."""

    def __init__(self, nx, ny, nz, mode):
        self.dims = (int(nx), int(ny), int(nz))
        self.M = self.dims[0] * self.dims[1] * self.dims[2]
        md = np.ascontiguousarray(mode, dtype=np.float64)
        self._h = lib().ref_mesh_create(self.dims[0], self.dims[1], self.dims[2], len(md), _d(md))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ref_mesh_destroy(self._h)
            self._h = None

    def set_bug_compat(self, on):
        lib().ref_mesh_set_bug_compat(self._h, int(bool(on)))

    def cv(self, postype, box, n_global=None):
        pt = np.ascontiguousarray(postype, dtype=np.float64)
        return lib().ref_mesh_cv(self._h, pt.shape[0], _d(pt), C.byref(box), pt.shape[0] if n_global is None else int(n_global))

    def forces(self, postype, box, bias, n_global=None):
        pt = np.ascontiguousarray(postype, dtype=np.float64)
        out = np.zeros((pt.shape[0], 4), dtype=np.float64)
        lib().ref_mesh_forces(self._h, pt.shape[0], _d(pt), C.byref(box), pt.shape[0] if n_global is None else int(n_global),
                              float(bias), _d(out))
        return out

    def set_table(self, K, dK, kmin, kmax):
        k = np.ascontiguousarray(K, dtype=np.float64)
        d = np.ascontiguousarray(dK, dtype=np.float64)
        if lib().ref_mesh_set_table(self._h, _d(k), _d(d), len(k), float(kmin), float(kmax)):
            raise RuntimeError("Error setting up OrderParameterMesh")

    def set_use_table(self, on):
        lib().ref_mesh_set_use_table(self._h, int(bool(on)))

    def qmax(self, n_global):
        out = np.zeros(4)
        lib().ref_mesh_qmax(self._h, int(n_global), _d(out))
        return out

    def virial(self, n_global, bias):
        out = np.zeros(6)
        lib().ref_mesh_virial(self._h, int(n_global), float(bias), _d(out))
        return out

    def assign(self, postype, box):
        """spread one shard: fills array("mesh") and mode_sq (sharded checks sum both over the shards)"""
        pt = np.ascontiguousarray(postype, dtype=np.float64)
        lib().ref_mesh_assign(self._h, pt.shape[0], _d(pt), C.byref(box))

    def raw_mesh(self):
        """writable (M, 2) view of the real-space mesh (re, im)"""
        ptr = lib().ref_mesh_array(self._h, 0)
        buf = (C.c_double * (2 * self.M)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.float64).reshape(self.M, 2)

    def set_mode_sq(self, v):
        lib().ref_mesh_set_mode_sq(self._h, float(v))

    def spectral(self, n_global):
        return lib().ref_mesh_spectral(self._h, int(n_global))

    @property
    def mode_sq(self):
        return lib().ref_mesh_mode_sq(self._h)

    def array(self, name):
        names = ["mesh", "fourier_mesh", "fourier_mesh_G", "inv_fourier_mesh", "interpolation_f", "inf_f", "k"]
        which = names.index(name)
        ptr = lib().ref_mesh_array(self._h, which)
        nz, ny, nx = self.dims[2], self.dims[1], self.dims[0]
        if which < 4:
            buf = (C.c_double * (2 * self.M)).from_address(ptr)
            a = np.frombuffer(buf, dtype=np.float64).reshape(nz, ny, nx, 2)
            return a[..., 0] + 1j * a[..., 1]
        n = self.M * (3 if which == 6 else 1)
        buf = (C.c_double * n).from_address(ptr)
        a = np.frombuffer(buf, dtype=np.float64)
        return a.reshape(nz, ny, nx, 3) if which == 6 else a.reshape(nz, ny, nx)


def sph_evaluate(lmax, polar, azimuth, full_m=True):
    """fsph::evaluate_SPH: complex array (N, per)"""
    ph = np.ascontiguousarray(polar, dtype=np.float64)
    th = np.ascontiguousarray(azimuth, dtype=np.float64)
    per = (lmax + 1) ** 2 if full_m else (lmax + 1) * (lmax + 2) // 2
    out = np.zeros(2 * per * len(ph))
    lib().ref_sph_evaluate(_d(out), int(lmax), _d(ph), _d(th), len(ph), int(full_m))
    out = out.reshape(len(ph), per, 2)
    return out[..., 0] + 1j * out[..., 1]


def ql_compute_cv(postype, box, head_list, n_neigh, nlist, rcut, ron, lmax, type_id, Ql_ref, half=False, n_global=None):
    """returns (value, Qlm complex[(lmax+1)^2], Ql[lmax+1])"""
    pt = np.ascontiguousarray(postype, dtype=np.float64)
    hl = np.ascontiguousarray(head_list, dtype=np.uint32)
    nn = np.ascontiguousarray(n_neigh, dtype=np.uint32)
    nl = np.ascontiguousarray(nlist, dtype=np.uint32)
    qr = np.ascontiguousarray(Ql_ref, dtype=np.float64)
    cnt = (lmax + 1) ** 2
    qlm = np.zeros(2 * cnt)
    ql = np.zeros(lmax + 1)
    # central particles = entries of head_list; postype may hold ghost particles behind them (sharded checks)
    v = lib().ref_ql_compute_cv(hl.shape[0], _d(pt), C.byref(box), _u(hl), _u(nn), _u(nl), int(half), float(rcut), float(ron),
                                int(lmax), int(type_id), _d(qr), pt.shape[0] if n_global is None else int(n_global), _d(qlm), _d(ql))
    return v, qlm[0::2] + 1j * qlm[1::2], ql


def ql_from_qlm(lmax, Qlm, Ql_ref, n_global):
    """tail of computeCV from a summed Q_lm table: returns (value, Ql)"""
    q = np.zeros(2 * len(Qlm))
    q[0::2], q[1::2] = np.real(Qlm), np.imag(Qlm)
    qr = np.ascontiguousarray(Ql_ref, dtype=np.float64)
    ql = np.zeros(lmax + 1)
    v = lib().ref_ql_from_qlm(int(lmax), _d(q), _d(qr), int(n_global), _d(ql))
    return v, ql


def ql_compute_forces(postype, box, head_list, n_neigh, nlist, rcut, ron, lmax, type_id, Ql_ref, Qlm, bias, half=False,
                      n_global=None):
    pt = np.ascontiguousarray(postype, dtype=np.float64)
    hl = np.ascontiguousarray(head_list, dtype=np.uint32)
    nn = np.ascontiguousarray(n_neigh, dtype=np.uint32)
    nl = np.ascontiguousarray(nlist, dtype=np.uint32)
    qr = np.ascontiguousarray(Ql_ref, dtype=np.float64)
    q = np.zeros(2 * len(Qlm))
    q[0::2], q[1::2] = np.real(Qlm), np.imag(Qlm)
    out = np.zeros((pt.shape[0], 4))             # half lists also write reaction forces on neighbours (ghosts included)
    lib().ref_ql_compute_forces(hl.shape[0], _d(pt), C.byref(box), _u(hl), _u(nn), _u(nl), int(half), float(rcut), float(ron),
                                int(lmax), int(type_id), _d(qr), pt.shape[0] if n_global is None else int(n_global), _d(q),
                                float(bias), _d(out))
    return out
