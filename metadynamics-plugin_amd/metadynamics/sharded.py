"""Particle-sharded and multiple-walker orchestration of the bias step.

The reference reduces the Fourier modes with a host-staged ``MPI_Allreduce``
(LamellarOrderParameterGPU.cc:69-77), computes the bias on the root rank and broadcasts it
(IntegratorMetaDynamics.cc:346-351, 571-575).  Here every rank keeps a replicated, deterministic
bias grid: the only per-step exchange is an all-reduce of n_cv doubles between the two launches of
the fused step, so the broadcast disappears.  Multiple walkers exchange the packed delta arrays
(IntegratorMetaDynamics.cc:393-409) with one all-reduce per element type.

The classes are backend-agnostic: the product backends are :class:`HipLamellarBackend` (the fused two-launch
lamellar step) and :class:`HipCvSetBackend` (any mix of lamellar / mesh / Steinhardt / energy CVs; also the walker
protocol) — libmtd_hip.so through the C ABI, no CPU fallback; the CPU gloo tests inject checker backends of their own.

Per-CV exchange of a particle-sharded step: lamellar — its sum; mesh — the replicated real mesh and
sum(mode^2) (M + 1 doubles); Steinhardt — the (lmax+1)(lmax+2) Q'_lm sums; potential energy / wrapper — one double.

These classes are LAUNCHER-SIDE orchestration for Python drivers (tests, bench.py): device buffers arrive as tensors and the
control plane is ``torch.distributed``.  A C++ caller needs none of it — the library exports the same exchanges itself:
``mtd_comm_allreduce_small`` (xGMI mailbox), ``mtd_comm_allreduce_large`` / ``mtd_rccl_*`` (RCCL, bound at run time) and
``mtd_metad_update_bias_walkers``; pass ``large=RcclAllReduce(...)`` to route the large buffers through them here as well.
"""
import ctypes as C


class RcclAllReduce:
    """``mtd_comm_allreduce_large`` over an ``mtd_rccl`` communicator built from an id the process group broadcasts:
    the large exchanges (replicated mesh, walker deltas) without torch's collective — what a C++ host does"""

    def __init__(self, dist):
        import torch
        from . import _abi
        self._abi, self.lib = _abi, _abi.load()
        rank, world = dist.get_rank(), dist.get_world_size()
        uid = (C.c_ubyte * 128)()
        if rank == 0:
            _abi.check(self.lib.mtd_rccl_unique_id(uid))
        t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone()
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = t.to(dev)
        dist.broadcast(t, 0)
        blob = (C.c_ubyte * 128)(*t.cpu().numpy().tolist())
        self.handle = C.c_void_p()
        _abi.check(self.lib.mtd_rccl_create(C.byref(self.handle), blob, rank, world))

    def all_reduce(self, tensor):
        elem = 1 if str(tensor.dtype).endswith("float64") else 2
        self._abi.check(self.lib.mtd_comm_allreduce_large(self.handle, tensor.data_ptr(), int(tensor.numel()), elem, None))
        return tensor

    def close(self):
        if self.handle:
            self._abi.check(self.lib.mtd_rccl_destroy(self.handle))
            self.handle = None


class ShardedBiasStep:
    """One metadynamics bias step with the particles sharded over the ranks of ``dist``.

    backend protocol:
        cv_pass()            -> tensor of n_cv local per-CV sums (on the backend's device)
        force_pass(sums, t)  -> runs updateBiasPotential(t) for s_c = sums[c] / N_global and writes the
                                bias forces of the local particles
    """

    def __init__(self, backend, dist=None, group=None, mailbox=None, mailbox_max=64, large=None):
        self.backend = backend
        self.dist = dist
        self.group = group
        self.large = large            # RcclAllReduce: the large buffers through the library's own RCCL binding
        # xGMI mailbox (metadynamics.xgmi.Mailbox) for the small exchange buffers of a CV set (lamellar sums, Q_lm sums,
        # energies: a few doubles each); buffers above mailbox_max doubles (the replicated mesh) stay on the collective
        self.small = mailbox
        self.small_max = int(mailbox_max)

    def step(self, timestep):
        if getattr(self.backend, "mailbox", None) is not None:
            # xGMI mailbox attached (metadynamics.xgmi): launch A's last block sends this rank's sums to every rank,
            # launch B's scalar chain polls the local mailbox — two launches, no collective call
            self.backend.step_single(timestep)
            return
        sums = self.backend.cv_pass()
        if self.dist is not None:
            # Q3 of SURVEY §2.3: the reference reduces only half of its Scalar2 buffer; every sum is reduced here.
            # A CV set hands over one buffer per collective variable (n_cv doubles for lamellar sums, the real mesh,
            # the Q_lm sums, one energy): one all-reduce each, in the order of the CVs.
            for buf in (sums if isinstance(sums, (list, tuple)) else [sums]):
                if buf is None:
                    continue                    # the part exchanged what it needed itself (MeshSlabPart)
                if self.small is not None and buf.numel() <= self.small_max and str(buf.dtype).endswith("float64"):
                    self.small.all_reduce(buf)
                elif self.large is not None and (str(buf.dtype).endswith("float64") or str(buf.dtype).endswith("int32")):
                    self.large.all_reduce(buf)
                else:
                    self.dist.all_reduce(buf, group=self.group)
        self.backend.force_pass(sums, timestep)


class WalkerBiasStep:
    """Multiple walkers: one full simulation per rank sharing one bias grid.

    backend protocol:
        phase_a(t)        -> bool deposited; fills the delta arrays
        delta_buffers()   -> (real tensor view [2G], count tensor view [2G]) aliasing the engine's arrays
        phase_b(deposited)
    """

    def __init__(self, backend, dist, group=None, large=None):
        self.backend = backend
        self.dist = dist
        self.group = group
        self.large = large            # RcclAllReduce: mtd_comm_allreduce_large instead of the process group's collective

    def step(self, timestep):
        dep = self.backend.phase_a(timestep)
        if dep:
            real, count = self.backend.delta_buffers()
            if self.large is not None:
                self.large.all_reduce(real)
                self.large.all_reduce(count)
            else:
                self.dist.all_reduce(real, group=self.group)
                self.dist.all_reduce(count, group=self.group)
        self.backend.phase_b(dep)


class _DeviceView:
    """``__cuda_array_interface__`` carrier: lets torch alias device memory owned by libmtd_hip.so without a copy"""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(count),), typestr=typestr, data=(int(ptr), False), version=2)


def device_view(ptr, count, typestr="<f8"):
    """torch tensor over ``count`` elements of device memory at ``ptr`` (typestr '<f8' double, '<i4' int32)"""
    import torch
    return torch.as_tensor(_DeviceView(ptr, count, typestr), device="cuda")


class LamellarPart:
    """one cv.lamellar in a CV set (generic kernels: mtd_lamellar_cv_partials / mtd_lamellar_forces)"""

    def __init__(self, lattice_vectors, mode, d_postype, n_global, box_L):
        import torch
        from . import _abi
        self._abi, self.lib = _abi, _abi.load()
        self.N, self.N_global, self.d_pos = int(d_postype.shape[0]), int(n_global), d_postype
        self.dt = _abi.MTD_F32 if d_postype.dtype == torch.float32 else _abi.MTD_F64
        self.box = _abi.Box.make(box_L)
        self.lset = _abi.LamellarSet.make([(lattice_vectors, mode)])
        self.scratch = torch.zeros(self.lib.mtd_lamellar_scratch_doubles(self.N), dtype=torch.float64, device=d_postype.device)
        self.sum = torch.zeros(1, dtype=torch.float64, device=d_postype.device)
        self.force = torch.zeros((self.N, 4), dtype=d_postype.dtype, device=d_postype.device)
        self.fptr = (C.c_void_p * 1)(self.force.data_ptr())

    def local_pass(self):
        n = C.c_uint()
        self._abi.check(self.lib.mtd_lamellar_cv_partials(C.byref(self.lset), self.N, self.d_pos.data_ptr(), self.dt, C.byref(self.box),
                                                          self.scratch.data_ptr(), C.byref(n), None))
        self._abi.check(self.lib.mtd_reduce_partials(self.scratch.data_ptr(), n.value, 1, 1, 1.0, 0.0, self.sum.data_ptr(), None))
        return self.sum

    def finish(self, engine, slot):
        self._abi.check(self.lib.mtd_metad_set_cv_source(engine, slot, self.sum.data_ptr(), 1, 1, 0, 1.0 / self.N_global, 0.0))

    def forces(self, d_bias):
        self._abi.check(self.lib.mtd_lamellar_forces(C.byref(self.lset), self.N, self.d_pos.data_ptr(), self.fptr, self.dt,
                                                     self.N_global, d_bias, C.byref(self.box), None))


class MeshPart:
    """cv.mesh with a replicated mesh: spread the local particles, all-reduce mesh + sum(mode^2), FFTs on every rank"""

    def __init__(self, nx, ny, nz, mode, d_postype, n_global, box_L, bug_compat=True):
        import torch
        from . import _abi
        self._abi, self.lib = _abi, _abi.load()
        self.N, self.N_global, self.d_pos = int(d_postype.shape[0]), int(n_global), d_postype
        self.dt = _abi.MTD_F32 if d_postype.dtype == torch.float32 else _abi.MTD_F64
        self.box = _abi.Box.make(box_L)
        self.h = C.c_void_p()
        md = (C.c_double * len(mode))(*[float(x) for x in mode])
        _abi.check(self.lib.mtd_mesh_create(C.byref(self.h), nx, ny, nz, md, len(mode), max(self.N, 1)))
        _abi.check(self.lib.mtd_mesh_set_bug_compat(self.h, int(bug_compat)))
        ptr, cnt = C.c_void_p(), C.c_size_t()
        _abi.check(self.lib.mtd_mesh_exchange_buffer(self.h, C.byref(ptr), C.byref(cnt)))
        self.exchange = device_view(ptr.value, cnt.value)
        self.force = torch.zeros((self.N, 4), dtype=d_postype.dtype, device=d_postype.device)

    def close(self):
        if self.h:
            self._abi.check(self.lib.mtd_mesh_destroy(self.h))
            self.h = None

    def local_pass(self):
        self._abi.check(self.lib.mtd_mesh_assign(self.h, self.N, self.d_pos.data_ptr(), self.dt, C.byref(self.box), None))
        return self.exchange

    def finish(self, engine, slot):
        part, n = C.c_void_p(), C.c_uint()
        self._abi.check(self.lib.mtd_mesh_spectral(self.h, C.byref(self.box), self.N_global, C.byref(part), C.byref(n), None))
        self._abi.check(self.lib.mtd_metad_set_cv_source(engine, slot, part.value, n.value, 1, 0, 0.5, 0.0))

    def forces(self, d_bias):
        self._abi.check(self.lib.mtd_mesh_forces(self.h, self.N, self.d_pos.data_ptr(), self.force.data_ptr(), self.dt,
                                                 C.byref(self.box), self.N_global, d_bias, 0.0, None))


class MeshSlabPart(MeshPart):
    """cv.mesh with the mesh DECOMPOSED over the ranks of an xGMI mailbox instead of replicated: z slabs
    for the x / y transforms, y rows for the z transform, the transposes as remote loads out of exported buffers
    (``mtd_mesh_slab_*``).  nz and ny must be multiples of the number of ranks.  The whole forward / spectral / inverse
    sequence, its four barriers included, runs inside ``local_pass``; there is nothing left for the caller to all-reduce."""

    def __init__(self, nx, ny, nz, mode, d_postype, n_global, box_L, mailbox, dist, bug_compat=True):
        super().__init__(nx, ny, nz, mode, d_postype, n_global, box_L, bug_compat=bug_compat)
        self.box_mail = mailbox
        sizes = (C.c_size_t * 4)()
        self._abi.check(self.lib.mtd_mesh_slab_bytes(self.h, mailbox.world, sizes))
        peers = []
        for k in range(4):
            _, addr = mailbox.share(dist, sizes[k])
            peers.append((C.c_void_p * mailbox.world)(*addr))
        self._abi.check(self.lib.mtd_mesh_slab_attach(self.h, mailbox.handle, peers[0], peers[1], peers[2], peers[3]))
        self.cv_sum = C.c_void_p()

    def local_pass(self):
        self._abi.check(self.lib.mtd_mesh_slab_compute_cv(self.h, self.N, self.d_pos.data_ptr(), self.dt, C.byref(self.box),
                                                          self.N_global, C.byref(self.cv_sum), None))
        return None

    def finish(self, engine, slot):
        self._abi.check(self.lib.mtd_metad_set_cv_source(engine, slot, self.cv_sum.value, 1, 1, 0, 0.5, 0.0))


class SteinhardtPart:
    """cv.steinhardt over a shard: d_postype holds the n_local central particles followed by their ghost particles;
    the neighbour list (HOOMD layout, n_local heads) may index the ghosts.  Exchange: the Q'_lm sums."""

    def __init__(self, r_cut, r_on, lmax, Ql_ref, type_id, d_postype, n_local, nlist, n_global, box_L, half=False):
        import torch
        from . import _abi
        self._abi, self.lib = _abi, _abi.load()
        self.N, self.N_global, self.d_pos = int(n_local), int(n_global), d_postype
        self.dt = _abi.MTD_F32 if d_postype.dtype == torch.float32 else _abi.MTD_F64
        self.box = _abi.Box.make(box_L)
        self.r_cut, self.r_on, self.lmax, self.type_id, self.half = float(r_cut), float(r_on), int(lmax), int(type_id), int(half)
        self.ql_ref = (C.c_double * (self.lmax + 1))(*[float(x) for x in Ql_ref])
        self.head, self.nneigh, self.nlist = nlist
        self.scratch = torch.zeros(self.lib.mtd_ql_scratch_doubles(self.lmax), dtype=torch.float64, device=d_postype.device)
        if half and int(d_postype.shape[0]) > self.N:
            # the third-law path only adds the reaction force to LOCAL partners (SteinhardtQl.cc:328: j < N), so a pair that
            # straddles two shards would lose it; like HOOMD's domain decomposition, sharded runs take full lists
            raise ValueError("SteinhardtPart: half neighbour lists cannot be combined with ghost particles")
        self.force = torch.zeros((self.N, 4), dtype=d_postype.dtype, device=d_postype.device)
        self.sums = None

    def local_pass(self):
        ptr, n = C.c_void_p(), C.c_uint()
        self._abi.check(self.lib.mtd_ql_accumulate_local(self.N, self.d_pos.data_ptr(), self.dt, C.byref(self.box), self.head.data_ptr(),
                                                         self.nneigh.data_ptr(), self.nlist.data_ptr(), self.half, self.r_cut, self.r_on,
                                                         self.lmax, self.type_id, self.N_global, self.scratch.data_ptr(), C.byref(ptr),
                                                         C.byref(n), None))
        off = (ptr.value - self.scratch.data_ptr()) // 8
        self.sums = self.scratch[off:off + n.value]
        return self.sums

    def finish(self, engine, slot):
        val = C.c_void_p()
        self._abi.check(self.lib.mtd_ql_finalize(self.half, self.lmax, self.ql_ref, self.N_global, self.scratch.data_ptr(), C.byref(val),
                                                 None, None, None))
        self._abi.check(self.lib.mtd_metad_set_cv_source(engine, slot, val.value, 1, 1, 0, 1.0, 0.0))

    def forces(self, d_bias):
        self._abi.check(self.lib.mtd_ql_forces(self.N, self.d_pos.data_ptr(), self.force.data_ptr(), self.dt, C.byref(self.box),
                                               self.head.data_ptr(), self.nneigh.data_ptr(), self.nlist.data_ptr(), self.half, self.r_cut,
                                               self.r_on, self.lmax, self.type_id, self.ql_ref, self.N_global, self.scratch.data_ptr(),
                                               d_bias, 0.0, None))


class EnergyPart:
    """an energy as CV: the net force arrays (cv.potential_energy, factor 1 + bias) or one wrapped compute's own arrays
    (cv.wrap, factor bias).  Exchange: one double (local sum of force.w + this rank's external energy)."""

    def __init__(self, d_force, d_torque, d_virial, pitch, external_energy=0.0, wrapper=False):
        import torch
        from . import _abi
        self._abi, self.lib = _abi, _abi.load()
        self.f, self.t, self.v, self.pitch = d_force, d_torque, d_virial, int(pitch)
        self.N = int(d_force.shape[0])
        self.dt = _abi.MTD_F32 if d_force.dtype == torch.float32 else _abi.MTD_F64
        self.ext, self.wrapper = float(external_energy), bool(wrapper)
        self.partials = torch.zeros(self.lib.mtd_wte_scratch_doubles(self.N), dtype=torch.float64, device=d_force.device)
        self.sum = torch.zeros(1, dtype=torch.float64, device=d_force.device)

    def local_pass(self):
        n = C.c_uint()
        self._abi.check(self.lib.mtd_wte_energy_partials(self.N, self.f.data_ptr(), self.dt, self.partials.data_ptr(), C.byref(n), None))
        self._abi.check(self.lib.mtd_reduce_partials(self.partials.data_ptr(), n.value, 1, 1, 1.0, self.ext, self.sum.data_ptr(), None))
        return self.sum

    def finish(self, engine, slot):
        self._abi.check(self.lib.mtd_metad_set_cv_source(engine, slot, self.sum.data_ptr(), 1, 1, 0, 1.0, 0.0))

    def forces(self, d_bias):
        fn = self.lib.mtd_wrapper_scale_forces if self.wrapper else self.lib.mtd_wte_scale_netforce
        self._abi.check(fn(self.N, self.f.data_ptr(), self.t.data_ptr() if self.t is not None else None,
                           self.v.data_ptr() if self.v is not None else None, self.pitch, self.dt, d_bias, 0.0, 1, None))


class HipCvSetBackend:
    """Any mix of collective variables sharing one device-resident bias grid (generic path), particle sharded:
    every part contributes one exchange buffer; after the all-reduce every rank evaluates the replicated grid."""

    def __init__(self, parts, grid, W, T_shift, T, stride, mode="well_tempered", add_hills=True):
        from . import _abi
        self._abi, self.lib = _abi, _abi.load()
        self.parts = list(parts)
        self.n_cv = len(self.parts)
        dbl = lambda v: (C.c_double * len(v))(*[float(x) for x in v])
        self.h = C.c_void_p()
        _abi.check(self.lib.mtd_metad_create(C.byref(self.h), self.n_cv, dbl(grid["sigma"]), dbl(grid["cv_min"]), dbl(grid["cv_max"]),
                                             (C.c_uint * self.n_cv)(*grid["num_points"]), W, T_shift, T, stride,
                                             {"standard": 0, "well_tempered": 1}[mode], int(add_hills)))
        self.d_bias = self.lib.mtd_metad_bias_device(self.h)

    def close(self):
        if self.h:
            self._abi.check(self.lib.mtd_metad_destroy(self.h))
            self.h = None
        for p in self.parts:
            if hasattr(p, "close"):
                p.close()

    def cv_pass(self):
        return [p.local_pass() for p in self.parts]

    def finish(self):
        for slot, p in enumerate(self.parts):
            p.finish(self.h, slot)

    def write_forces(self):
        for slot, p in enumerate(self.parts):
            p.forces(self.d_bias + 8 * slot)

    def force_pass(self, sums, timestep):
        self.finish()
        self._abi.check(self.lib.mtd_metad_update_bias(self.h, int(timestep), None))
        self.write_forces()

    # ---- multiple walkers (WalkerBiasStep protocol): each walker is a whole simulation, the grid deltas are exchanged
    def phase_a(self, timestep):
        self.cv_pass()
        self.finish()
        dep = C.c_int()
        self._abi.check(self.lib.mtd_metad_update_phase_a(self.h, int(timestep), C.byref(dep), None))
        return bool(dep.value)

    def delta_buffers(self):
        real, cnt, n = C.c_void_p(), C.c_void_p(), C.c_uint()
        self._abi.check(self.lib.mtd_metad_delta_buffers(self.h, C.byref(real), C.byref(cnt), C.byref(n)))
        # counts travel as int32 (RCCL has no uint32 sum in torch); histogram counts stay far below 2^31
        return device_view(real.value, 2 * n.value, "<f8"), device_view(cnt.value, 2 * n.value, "<i4")

    def phase_b(self, deposited):
        self._abi.check(self.lib.mtd_metad_update_phase_b(self.h, int(bool(deposited)), None))
        self.write_forces()

    def state(self):
        n = self.n_cv
        cv, bias = (C.c_double * n)(), (C.c_double * n)()
        V, w, ng = C.c_double(), C.c_double(), C.c_uint()
        self._abi.check(self.lib.mtd_metad_get_state(self.h, cv, bias, C.byref(V), C.byref(w), C.byref(ng), None, None))
        return dict(cv=list(cv), bias=list(bias), V=V.value, w=w.value, num_gaussians=ng.value)

    def grid_array(self, which=0):
        import numpy as np
        G = self.lib.mtd_metad_num_elements(self.h)
        out = np.zeros(G, dtype=np.float64 if which < 6 else np.uint32)
        self._abi.check(self.lib.mtd_metad_get_array(self.h, which, out.ctypes.data, None))
        return out


class HipLamellarBackend:
    """The fused two-launch lamellar bias step on one GPU (C ABI of libmtd_hip.so)."""

    def __init__(self, cvs, d_postype, n_global, box_L, grid, W, T_shift, T, stride, mode="well_tempered",
                 fast_trig=True, fused=True, exchange="sums"):
        import torch
        self.exchange = exchange
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
        self.mailbox = None

    def attach_mailbox(self, box):
        """route the per-step exchange of the fused path through the xGMI mailbox ``box`` (metadynamics.xgmi.Mailbox);
        ``None`` detaches.  Every rank must attach (or not) alike."""
        if box is not None and not self.fused:
            raise ValueError("the mailbox exchange belongs to the fused path")
        self._abi.check(self.lib.mtd_metad_set_comm(self.h, box.handle if box is not None else None))
        self.mailbox = box

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

    def cv_pass(self, reduce_locally=None):
        """what the ranks all-reduce, as a device tensor: the n_cv local sums (``exchange="sums"``, any shard sizes) or the
        block partial sums themselves (``exchange="partials"``: n_blocks x n_cv doubles, 4 KB at the headline config — as
        cheap to all-reduce as 16 bytes and it saves the reduce launch; launch B then adds up the reduced rows exactly as
        on one GPU.  Every rank must hold the same number of particles, so that the launch geometry — the row count — agrees)"""
        self.cv_partials()
        if reduce_locally is None:
            reduce_locally = self.exchange != "partials"
        if not reduce_locally:
            return self.scratch[: self.n_part.value * self.n_cv]
        self._abi.check(self.lib.mtd_reduce_partials(self.scratch.data_ptr(), self.n_part.value, self.n_cv, self.n_cv,
                                                     1.0, 0.0, self.cv_sum.data_ptr(), None))
        return self.cv_sum

    # ---- launch B
    def force_pass(self, sums, timestep):
        """sums: None -> single GPU, read the block partial sums directly; else the (all-reduced) n_cv sums"""
        abi, lib = self._abi, self.lib
        if sums is None or sums.data_ptr() == self.scratch.data_ptr():
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
