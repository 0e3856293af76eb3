"""Particle-sharded and multiple-walker orchestration of the bias step.

The reference reduces the Fourier modes with a host-staged ``MPI_Allreduce``
(LamellarOrderParameterGPU.cc:69-77), computes the bias on the root rank and broadcasts it
(IntegratorMetaDynamics.cc:346-351, 571-575).  Here every rank keeps a replicated, deterministic
bias grid: the only per-step exchange is an all-reduce of n_cv doubles between the two launches of
the fused step, so the broadcast disappears.  Multiple walkers exchange the packed delta arrays
(IntegratorMetaDynamics.cc:393-409) with one all-reduce per element type.

The classes are backend-agnostic: the product backend is :class:`HipLamellarBackend` (libmtd_hip.so
through the C ABI, no CPU fallback); the CPU gloo tests inject a checker backend of their own.
"""
import ctypes as C


class ShardedBiasStep:
    """One metadynamics bias step with the particles sharded over the ranks of ``dist``.

    backend protocol:
        cv_pass()            -> tensor of n_cv local per-CV sums (on the backend's device)
        force_pass(sums, t)  -> runs updateBiasPotential(t) for s_c = sums[c] / N_global and writes the
                                bias forces of the local particles
    """

    def __init__(self, backend, dist=None, group=None):
        self.backend = backend
        self.dist = dist
        self.group = group

    def step(self, timestep):
        sums = self.backend.cv_pass()
        if self.dist is not None:
            # Q3 of SURVEY §2.3: the reference reduces only half of its Scalar2 buffer; every sum is reduced here
            self.dist.all_reduce(sums, group=self.group)
        self.backend.force_pass(sums, timestep)


class WalkerBiasStep:
    """Multiple walkers: one full simulation per rank sharing one bias grid.

    backend protocol:
        phase_a(t)        -> bool deposited; fills the delta arrays
        delta_buffers()   -> (real tensor view [2G], count tensor view [2G]) aliasing the engine's arrays
        phase_b(deposited)
    """

    def __init__(self, backend, dist, group=None):
        self.backend = backend
        self.dist = dist
        self.group = group

    def step(self, timestep):
        dep = self.backend.phase_a(timestep)
        if dep:
            real, count = self.backend.delta_buffers()
            self.dist.all_reduce(real, group=self.group)
            self.dist.all_reduce(count, group=self.group)
        self.backend.phase_b(dep)


class HipLamellarBackend:
    """The fused two-launch lamellar bias step on one GPU (C ABI of libmtd_hip.so)."""

    def __init__(self, cvs, d_postype, n_global, box_L, grid, W, T_shift, T, stride, mode="well_tempered",
                 fast_trig=True, fused=True):
        import torch
        from . import _abi
        self._abi, self._torch = _abi, torch
        self.lib = lib = _abi.load()
        self.fused = fused
        self.n_cv = len(cvs)
        self.N = int(d_postype.shape[0])
        self.N_global = int(n_global)
        self.d_pos = d_postype
        self.dt = _abi.MTD_F32 if d_postype.dtype == torch.float32 else _abi.MTD_F64
        self.box = _abi.Box.make(box_L)
        self.lset = _abi.LamellarSet.make(cvs)
        self.scratch = torch.zeros(lib.mtd_lamellar_scratch_doubles(self.N), dtype=torch.float64, device=d_postype.device)
        self.cv_sum = torch.zeros(self.n_cv, dtype=torch.float64, device=d_postype.device)
        self.forces = [torch.zeros((self.N, 4), dtype=d_postype.dtype, device=d_postype.device) for _ in cvs]
        self.fptr = (C.c_void_p * self.n_cv)(*[f.data_ptr() for f in self.forces])
        dbl = lambda v: (C.c_double * len(v))(*[float(x) for x in v])
        self.h = C.c_void_p()
        _abi.check(lib.mtd_metad_create(C.byref(self.h), self.n_cv, dbl(grid["sigma"]), dbl(grid["cv_min"]),
                                        dbl(grid["cv_max"]), (C.c_uint * self.n_cv)(*grid["num_points"]), W, T_shift, T,
                                        stride, {"standard": 0, "well_tempered": 1}[mode], 1))
        _abi.check(lib.mtd_lamellar_set_fast_trig(int(fast_trig)))
        self.d_bias = lib.mtd_metad_bias_device(self.h)
        self.n_part = C.c_uint()
        self._sources = None

    def close(self):
        if self.h:
            self._abi.check(self.lib.mtd_metad_destroy(self.h))
            self.h = None

    def _set_sources(self, ptr, n_partials):
        key = (ptr, n_partials)
        if self._sources != key:
            for c in range(self.n_cv):
                self._abi.check(self.lib.mtd_metad_set_cv_source(self.h, c, ptr, n_partials, self.n_cv, c,
                                                                 1.0 / self.N_global, 0.0))
            self._sources = key

    # ---- launch A
    def cv_partials(self):
        abi, lib = self._abi, self.lib
        if self.fused:
            abi.check(lib.mtd_fused_cv_pass(self.h, C.byref(self.lset), self.N, self.d_pos.data_ptr(), self.dt,
                                            C.byref(self.box), self.scratch.data_ptr(), C.byref(self.n_part), None))
        else:
            abi.check(lib.mtd_lamellar_cv_partials(C.byref(self.lset), self.N, self.d_pos.data_ptr(), self.dt,
                                                   C.byref(self.box), self.scratch.data_ptr(), C.byref(self.n_part), None))

    def cv_pass(self):
        """local per-CV sums as a device tensor (what the ranks all-reduce)"""
        self.cv_partials()
        self._abi.check(self.lib.mtd_reduce_partials(self.scratch.data_ptr(), self.n_part.value, self.n_cv, self.n_cv,
                                                     1.0, 0.0, self.cv_sum.data_ptr(), None))
        return self.cv_sum

    # ---- launch B
    def force_pass(self, sums, timestep):
        """sums: None -> single GPU, read the block partial sums directly; else the (all-reduced) n_cv sums"""
        abi, lib = self._abi, self.lib
        if sums is None:
            self._set_sources(self.scratch.data_ptr(), self.n_part.value)
        else:
            self._set_sources(sums.data_ptr(), 1)
        if self.fused:
            abi.check(lib.mtd_fused_force_pass(self.h, C.byref(self.lset), self.N, self.d_pos.data_ptr(), self.fptr,
                                               self.dt, self.N_global, C.byref(self.box), int(timestep), None))
        else:
            abi.check(lib.mtd_metad_update_bias(self.h, int(timestep), None))
            abi.check(lib.mtd_lamellar_forces(C.byref(self.lset), self.N, self.d_pos.data_ptr(), self.fptr, self.dt,
                                              self.N_global, self.d_bias, C.byref(self.box), None))

    def step_single(self, timestep):
        """single-GPU step: no reduce kernel, launch B reads the block partial sums itself"""
        self.cv_partials()
        self.force_pass(None, timestep)

    def state(self):
        n = self.n_cv
        cv, bias = (C.c_double * n)(), (C.c_double * n)()
        V, w, ng = C.c_double(), C.c_double(), C.c_uint()
        self._abi.check(self.lib.mtd_metad_get_state(self.h, cv, bias, C.byref(V), C.byref(w), C.byref(ng), None, None))
        return dict(cv=list(cv), bias=list(bias), V=V.value, w=w.value, num_gaussians=ng.value)
