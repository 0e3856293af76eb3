"""ctypes binding of libmtd_hip.so (include/mtd_abi.h) — the C-ABI drop-in boundary.

The library is the product: if it is missing or does not load this module raises, it never falls
back to a CPU path.  Device buffers are passed as raw addresses (``tensor.data_ptr()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
REPO_ROOT = os.path.dirname(PKG_ROOT)
LIB_PATH = os.environ.get("MTD_LIB_OVERRIDE") or os.path.join(PKG_ROOT, "lib", "libmtd_hip.so")  # override: diagnostic builds only
HEADER_PATH = os.path.join(REPO_ROOT, "include", "mtd_abi.h")

MTD_MAX_CV = 8
MTD_MAX_MODES = 64
MTD_MAX_TYPES = 16
MTD_METAD_MAX_CV = 6
MTD_F32, MTD_F64 = 0, 1
MODE_STANDARD, MODE_WELL_TEMPERED = 0, 1

ARRAY_NAMES = ["grid", "grid_delta", "reweighted", "weight", "sigma_grid", "sigma_grid_delta",
               "hist", "hist_delta", "hist_gauss", "hist_gauss_delta"]


class MtdError(RuntimeError):
    pass


class Box(C.Structure):
    _fields_ = [("L", C.c_double * 3), ("lo", C.c_double * 3),
                ("xy", C.c_double), ("xz", C.c_double), ("yz", C.c_double),
                ("periodic", C.c_ubyte * 3), ("_pad", C.c_ubyte * 5)]

    @classmethod
    def make(cls, L, lo=None, xy=0.0, xz=0.0, yz=0.0):
        if isinstance(L, (int, float)):
            L = [float(L)] * 3
        L = [float(x) for x in L]
        if lo is None:
            lo = [-0.5 * x for x in L]
        b = cls()
        b.L[:] = L
        b.lo[:] = [float(x) for x in lo]
        b.xy, b.xz, b.yz = float(xy), float(xz), float(yz)
        b.periodic[:] = [1, 1, 1]
        return b


class LamellarSet(C.Structure):
    _fields_ = [("n_cv", C.c_uint), ("n_types", C.c_uint), ("n_modes", C.c_uint),
                ("first", C.c_uint * (MTD_MAX_CV + 1)),
                ("hkl", (C.c_int * 3) * MTD_MAX_MODES),
                ("coeff", (C.c_double * MTD_MAX_TYPES) * MTD_MAX_CV),
                ("trig_mode", C.c_int)]

    @classmethod
    def make(cls, cvs, trig_mode=0):
        """cvs: list of (lattice_vectors [(h,k,l)...], mode coefficients per type [a_0, a_1, ...]);
        trig_mode: 0 the process default, 1 hardware sine / cosine, 2 ocml sinpi / cospi (mtd_abi.h)."""
        s = cls()
        s.trig_mode = int(trig_mode)
        if not 1 <= len(cvs) <= MTD_MAX_CV:
            raise MtdError("between 1 and %d lamellar CVs can be fused" % MTD_MAX_CV)
        n_types = len(cvs[0][1])
        k = 0
        s.first[0] = 0
        for c, (lattice, mode) in enumerate(cvs):
            if len(mode) != n_types:
                raise MtdError("cv.lamellar: Number of mode parameters has to equal the number of particle types!")
            if len(lattice) == 0:
                raise MtdError("cv.lamellar: List of supplied latice vectors is empty.")
            for hkl in lattice:
                if len(hkl) != 3:
                    raise MtdError("cv.lamellar: List of input lattice vectors not a list of triples.")
                if k >= MTD_MAX_MODES:
                    raise MtdError("too many Fourier modes (max %d)" % MTD_MAX_MODES)
                s.hkl[k][:] = [int(x) for x in hkl]
                k += 1
            s.first[c + 1] = k
            for t, a in enumerate(mode):
                s.coeff[c][t] = float(a)
        s.n_cv, s.n_types, s.n_modes = len(cvs), n_types, k
        return s


_vp = C.c_void_p
_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint)
_ip = C.POINTER(C.c_int)

_SIGNATURES = {
    "mtd_abi_version": (C.c_int, []),
    "mtd_status_string": (C.c_char_p, [C.c_int]),
    "mtd_device_count": (C.c_int, []),
    "mtd_lamellar_scratch_doubles": (C.c_size_t, [C.c_uint]),
    "mtd_calculate_fourier_modes": (C.c_int, [C.c_uint, _ip, C.c_uint, _vp, C.c_int, _dp, C.c_uint, _vp, _vp,
                                               C.POINTER(Box), _vp]),
    "mtd_lamellar_cv_partials": (C.c_int, [C.POINTER(LamellarSet), C.c_uint, _vp, C.c_int, C.POINTER(Box), _vp,
                                            _up, _vp]),
    "mtd_reduce_partials": (C.c_int, [_vp, C.c_uint, C.c_uint, C.c_uint, C.c_double, C.c_double, _vp, _vp]),
    "mtd_compute_sq_forces": (C.c_int, [C.c_uint, _vp, _vp, C.c_int, C.c_uint, _ip, _dp, C.c_uint, C.c_uint,
                                         C.c_double, C.POINTER(Box), _vp]),
    "mtd_lamellar_forces": (C.c_int, [C.POINTER(LamellarSet), C.c_uint, _vp, C.POINTER(_vp), C.c_int, C.c_uint,
                                       _vp, C.POINTER(Box), _vp]),
    "mtd_lamellar_set_fast_trig": (C.c_int, [C.c_int]),
    "mtd_lamellar_get_fast_trig": (C.c_int, []),
    "mtd_debug_sph_harmonics": (C.c_int, [C.c_uint, C.c_uint, _vp, _vp]),
    "mtd_rccl_last_error": (C.c_char_p, []),
    "mtd_mesh_clear_rider": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "mtd_mesh_assign_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_uint), _vp]),
    "mtd_mesh_transform_info": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "mtd_mesh_forces_update_bias": (C.c_int, [_vp, _vp, C.c_uint, C.POINTER(LamellarSet), _up, C.c_uint, _vp, _vp, C.POINTER(_vp), C.c_int, C.c_uint,
                                              C.POINTER(Box), C.c_uint, _vp]),
    "mtd_mesh_set_lamellar_rider": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint, _vp, C.POINTER(C.c_uint), _vp]),
    "mtd_ql_symmetrize_half_list": (C.c_int, [C.c_uint, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t), _vp]),
    "mtd_debug_index_decode": (C.c_int, [C.c_uint, _vp, C.c_uint, _vp, _vp, _vp]),
    "mtd_update_grid": (C.c_int, [C.c_uint, _up, C.c_uint, _vp, _vp, _dp, _dp, _dp, C.c_double, C.c_double, _vp]),
    "mtd_metad_create": (C.c_int, [C.POINTER(_vp), C.c_uint, _dp, _dp, _dp, _up, C.c_double, C.c_double,
                                    C.c_double, C.c_uint, C.c_int, C.c_int]),
    "mtd_metad_destroy": (C.c_int, [_vp]),
    "mtd_metad_set_stride": (C.c_int, [_vp, C.c_uint]),
    "mtd_metad_set_add_hills": (C.c_int, [_vp, C.c_int]),
    "mtd_metad_set_mode": (C.c_int, [_vp, C.c_int]),
    "mtd_metad_set_sigma_inv": (C.c_int, [_vp, _dp]),
    "mtd_metad_reset_histogram": (C.c_int, [_vp, _vp]),
    "mtd_metad_set_cv_source": (C.c_int, [_vp, C.c_uint, _vp, C.c_uint, C.c_uint, C.c_uint, C.c_double, C.c_double]),
    "mtd_metad_set_cv_value": (C.c_int, [_vp, C.c_uint, C.c_double]),
    "mtd_metad_bias_device": (_vp, [_vp]),
    "mtd_metad_cv_device": (_vp, [_vp]),
    "mtd_metad_update_bias": (C.c_int, [_vp, C.c_uint, _vp]),
    "mtd_metad_update_phase_a": (C.c_int, [_vp, C.c_uint, _ip, _vp]),
    "mtd_metad_update_phase_b": (C.c_int, [_vp, C.c_int, _vp]),
    "mtd_metad_delta_buffers": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), _up]),
    "mtd_metad_get_state": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _up, _up, _vp]),
    "mtd_metad_sigma_determinant": (C.c_double, [_vp]),
    "mtd_metad_num_elements": (C.c_uint, [_vp]),
    "mtd_metad_get_array": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "mtd_metad_set_array": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "mtd_metad_set_num_gaussians": (C.c_int, [_vp, C.c_uint, _vp]),
    "mtd_metad_device_array": (_vp, [_vp, C.c_int]),
    "mtd_fused_cv_pass": (C.c_int, [_vp, C.POINTER(LamellarSet), C.c_uint, _vp, C.c_int, C.POINTER(Box), _vp, _up, _vp]),
    "mtd_fused_force_pass": (C.c_int, [_vp, C.POINTER(LamellarSet), C.c_uint, _vp, C.POINTER(_vp), C.c_int, C.c_uint,
                                        C.POINTER(Box), C.c_uint, _vp]),
    "mtd_fused_step": (C.c_int, [_vp, C.POINTER(LamellarSet), C.c_uint, _vp, C.POINTER(_vp), C.c_int, C.c_uint, C.POINTER(Box), _vp,
                                  C.c_uint, _vp]),
    "mtd_fused_step_launches": (C.c_uint, [_vp]),
    "mtd_fused_step_set_mode": (C.c_int, [_vp, C.c_int]),
    "mtd_profile_force_begin": (C.c_int, [C.c_uint]),
    "mtd_profile_force_end": (C.c_int, [_dp, C.c_uint, _up]),
    "mtd_comm_create": (C.c_int, [C.POINTER(_vp), C.c_uint, C.c_uint, C.c_uint]),
    "mtd_comm_handle": (C.c_int, [_vp, _vp]),
    "mtd_comm_connect": (C.c_int, [_vp, _vp]),
    "mtd_comm_allreduce_small": (C.c_int, [_vp, _vp, C.c_uint, _vp]),
    "mtd_comm_status": (C.c_int, [_vp, _up, _vp]),
    "mtd_comm_share": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp), _up, _vp]),
    "mtd_comm_open": (C.c_int, [_vp, C.c_uint, _vp, C.POINTER(_vp)]),
    "mtd_comm_world": (C.c_uint, [_vp]),
    "mtd_comm_rank": (C.c_uint, [_vp]),
    "mtd_comm_destroy": (C.c_int, [_vp]),
    "mtd_metad_set_comm": (C.c_int, [_vp, _vp]),
    "mtd_rccl_unique_id": (C.c_int, [_vp]),
    "mtd_rccl_create": (C.c_int, [C.POINTER(_vp), _vp, C.c_uint, C.c_uint]),
    "mtd_comm_allreduce_large": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp]),
    "mtd_rccl_world": (C.c_uint, [_vp]),
    "mtd_rccl_rank": (C.c_uint, [_vp]),
    "mtd_rccl_destroy": (C.c_int, [_vp]),
    "mtd_metad_update_bias_walkers": (C.c_int, [_vp, _vp, C.c_uint, _vp]),
    "mtd_fused_force_pass_slots": (C.c_int, [_vp, C.POINTER(LamellarSet), _up, C.c_uint, _vp, C.POINTER(_vp), C.c_int, C.c_uint,
                                              C.POINTER(Box), C.c_uint, _vp]),
    "mtd_mesh_create": (C.c_int, [C.POINTER(_vp), C.c_uint, C.c_uint, C.c_uint, _dp, C.c_uint, C.c_uint]),
    "mtd_mesh_destroy": (C.c_int, [_vp]),
    "mtd_mesh_set_bug_compat": (C.c_int, [_vp, C.c_int]),
    "mtd_mesh_set_keep_fourier": (C.c_int, [_vp, C.c_int]),
    "mtd_mesh_set_cv_event": (C.c_int, [_vp, _vp]),
    "mtd_mesh_num_cells": (C.c_uint, [_vp]),
    "mtd_mesh_set_table": (C.c_int, [_vp, _dp, _dp, C.c_uint, C.c_double, C.c_double]),
    "mtd_mesh_set_use_table": (C.c_int, [_vp, C.c_int]),
    "mtd_mesh_qmax": (C.c_int, [_vp, C.POINTER(Box), C.c_uint, _dp, _vp]),
    "mtd_mesh_virial": (C.c_int, [_vp, C.POINTER(Box), C.c_uint, C.c_double, _dp, _vp]),
    "mtd_mesh_assign": (C.c_int, [_vp, C.c_uint, _vp, C.c_int, C.POINTER(Box), _vp]),
    "mtd_mesh_exchange_buffer": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "mtd_mesh_spectral": (C.c_int, [_vp, C.POINTER(Box), C.c_uint, C.POINTER(_vp), _up, _vp]),
    "mtd_mesh_compute_cv": (C.c_int, [_vp, C.c_uint, _vp, C.c_int, C.POINTER(Box), C.c_uint, C.POINTER(_vp), _up, _vp]),
    "mtd_mesh_forces": (C.c_int, [_vp, C.c_uint, _vp, _vp, C.c_int, C.POINTER(Box), C.c_uint, _vp, C.c_double, _vp]),
    "mtd_mesh_slab_bytes": (C.c_int, [_vp, C.c_uint, C.POINTER(C.c_size_t)]),
    "mtd_mesh_slab_attach": (C.c_int, [_vp, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "mtd_mesh_slab_compute_cv": (C.c_int, [_vp, C.c_uint, _vp, C.c_int, C.POINTER(Box), C.c_uint, C.POINTER(_vp), _vp]),
    "mtd_mesh_get_array": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "mtd_ql_scratch_doubles": (C.c_size_t, [C.c_uint]),
    "mtd_ql_accumulate_local": (C.c_int, [C.c_uint, _vp, C.c_int, C.POINTER(Box), _vp, _vp, _vp, C.c_int, C.c_double, C.c_double,
                                           C.c_uint, C.c_uint, C.c_uint, _vp, C.POINTER(_vp), _up, _vp]),
    "mtd_ql_finalize": (C.c_int, [C.c_int, C.c_uint, _dp, C.c_uint, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "mtd_ql_finalize_update_bias": (C.c_int, [_vp, C.c_int, C.c_uint, _dp, C.c_uint, _vp, C.c_uint, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "mtd_metad_grid_touched": (C.c_int, [_vp, _vp]),
    "mtd_comm_pull_bytes": (C.c_size_t, [C.c_size_t]),
    "mtd_comm_pull_attach": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp), C.POINTER(_vp)]),
    "mtd_comm_allreduce_pull": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "mtd_ql_accumulate": (C.c_int, [C.c_uint, _vp, C.c_int, C.POINTER(Box), _vp, _vp, _vp, C.c_int, C.c_double, C.c_double, C.c_uint,
                                     C.c_uint, _dp, C.c_uint, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "mtd_ql_set_half_list_exact": (C.c_int, [C.c_int]),
    "mtd_ql_forces": (C.c_int, [C.c_uint, _vp, _vp, C.c_int, C.POINTER(Box), _vp, _vp, _vp, C.c_int, C.c_double, C.c_double, C.c_uint,
                                 C.c_uint, _dp, C.c_uint, _vp, _vp, C.c_double, _vp]),
    "mtd_wte_scratch_doubles": (C.c_size_t, [C.c_uint]),
    "mtd_wte_energy_partials": (C.c_int, [C.c_uint, _vp, C.c_int, _vp, _up, _vp]),
    "mtd_wte_scale_netforce": (C.c_int, [C.c_uint, _vp, _vp, _vp, C.c_uint, C.c_int, _vp, C.c_double, C.c_int, _vp]),
    "mtd_wrapper_scale_forces": (C.c_int, [C.c_uint, _vp, _vp, _vp, C.c_uint, C.c_int, _vp, C.c_double, C.c_int, _vp]),
    "mtd_sigma_scratch_doubles": (C.c_size_t, []),
    "mtd_sigma_products": (C.c_int, [C.c_uint, C.POINTER(C.c_void_p), C.c_uint, C.c_int, C.c_double, _vp, _dp, _vp]),
    "mtd_sigma_inverse": (C.c_int, [C.c_uint, _dp, _dp]),
}

_lib = None


def load():
    """Load libmtd_hip.so; raises MtdError when the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MtdError("libmtd_hip.so not found at %s — build it with __graft_entry__.build() "
                           "(make -C metadynamics-plugin_amd/csrc); there is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (SONAME
        # libamdhip64.so.7).  A process that uses torch as well (the test / bench harness) imports it
        # FIRST: the dynamic linker then satisfies this library's DT_NEEDED with that already-loaded copy,
        # so torch tensors, streams and RCCL buffers and our kernels share one runtime; loaded the other
        # way round two runtimes would coexist.
        # (torch is never imported from here: a process without it loads the HIP runtime libmtd_hip.so links against)
        # The xGMI mailbox and the slab mesh share device buffers between the processes of a node with hipIpcGetMemHandle; on
        # hosts whose driver only supports dmabuf IPC the runtime must be told BEFORE it initialises (it reads the variable
        # once), or the export fails with "invalid argument" and every rank falls back to the RCCL all-reduce.  Set here when
        # nobody chose a value (a process that initialised HIP before importing this module has to export it itself:
        # xgmi.connect reports the value it ran with).
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            if not hasattr(lib, name):
                continue  # optional blocks are bound by their own modules
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def bind(name, restype, argtypes):
    """Bind an additional entry point (used by the mesh / steinhardt modules)."""
    fn = getattr(load(), name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


def check(rc):
    if rc != 0:
        msg = load().mtd_status_string(int(rc))
        raise MtdError("libmtd_hip: %s (status %d)" % (msg.decode() if msg else "?", rc))


def declared_symbols():
    """Every function name include/mtd_abi.h declares (for the export test)."""
    import re
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mtd_[a-z0-9_]+)\s*\(", text)))


def ptr(t):
    """Device (or host) address of a torch tensor / numpy array / int / None."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data
