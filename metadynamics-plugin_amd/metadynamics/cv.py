"""Collective variables — the reference's ``hoomd.metadynamics.cv`` API (metadynamics/cv.py) over the MI355X host classes.

Class names, constructor arguments, ``set_grid`` / ``set_params`` and the error behaviour follow the reference
(file:line citations refer to /root/reference/metadynamics/cv.py).  There is no CPU class to choose
(cv.py:260-265 picks LamellarOrderParameter vs ...GPU by ``exec_conf.isCUDAEnabled()``): the GPU class is the only one.
"""
from . import _metadynamics
from . import context


class _collective_variable(object):
    """Base class (cv.py:11-170): a collective variable is a force with grid / umbrella parameters."""

    def __init__(self, sigma, name=None):
        self.name = name
        self.force_name = "cv" if name is None else str(name)
        self.enabled = True
        self.log = True
        self.sigma = sigma
        self.cv_min = 0.0
        self.cv_max = 0.0
        self.num_points = 0
        self.grid_set = False
        self.ftm_min = 0.0
        self.ftm_max = 0.0
        self.ftm_parameters_set = False
        self.umbrella = False
        self.reweight = False
        self.cpp_force = None
        context.current.forces.append(self)

    def set_grid(self, cv_min, cv_max, num_points):                 # cv.py:70-86
        self.cv_min = cv_min
        self.cv_max = cv_max
        self.num_points = int(num_points)
        self.grid_set = True

    def enable_histograms(self, ftm_min, ftm_max):                  # cv.py:88-103
        self.ftm_min = ftm_min
        self.ftm_max = ftm_max
        self.ftm_parameters_set = True

    def set_params(self, sigma=None, kappa=None, cv0=None, umbrella=None, width_flat=None, scale=None, reweight=None):
        """cv.py:105-170"""
        if sigma is not None:
            self.sigma = sigma
        if umbrella is not None:
            modes = {"no_umbrella": (self.cpp_force.umbrella.no_umbrella, False),
                     "linear": (self.cpp_force.umbrella.linear, True),
                     "harmonic": (self.cpp_force.umbrella.harmonic, True),
                     "wall": (self.cpp_force.umbrella.wall, True),
                     "gaussian": (self.cpp_force.umbrella.gaussian, True)}
            if umbrella not in modes:
                raise RuntimeError("Error setting parameters of collective variable.")   # cv.py:150-152
            cpp_umbrella, flag = modes[umbrella]
            self.reweight = flag
            self.umbrella = flag
            self.cpp_force.setUmbrella(cpp_umbrella)
        if kappa is not None:
            self.cpp_force.setKappa(kappa)
        if width_flat is not None:
            self.cpp_force.setWidthFlat(width_flat)
        if cv0 is not None:
            self.cpp_force.setMinimum(cv0)
        if scale is not None:
            self.cpp_force.setScale(scale)
        if reweight is not None:
            self.reweight = reweight

    def update_coeffs(self):
        pass


def _mode_vector(mode, what):
    if type(mode) != type(dict()):
        raise RuntimeError("Error creating collective variable.")    # cv.py:236-238
    pdata = context.current.system_definition.getParticleData()
    out = []
    for i in range(pdata.getNTypes()):
        t = pdata.getNameByType(i)
        if t not in mode.keys():
            raise RuntimeError("Error creating collective variable.")   # cv.py:245-247: missing mode amplitude
        out.append(float(mode[t]))
    return out


class lamellar(_collective_variable):
    """Lamellar order parameter (cv.py:173-272): s = (1/N) sum_i sum_j a(type_j) cos(q_i . r_j)."""

    def __init__(self, mode, lattice_vectors, name=None, sigma=1.0):
        if name is not None:
            name = "_" + name
            suffix = name
        else:
            suffix = ""
        _collective_variable.__init__(self, sigma, name)
        if len(lattice_vectors) == 0:
            raise RuntimeError("Error creating collective variable.")   # cv.py:232-234
        cpp_mode = _mode_vector(mode, "cv.lamellar")
        cpp_lattice_vectors = _metadynamics.std_vector_int3()
        for l in lattice_vectors:
            if len(l) != 3:
                raise RuntimeError("Error creating collective variable.")   # cv.py:252-254
            cpp_lattice_vectors.append(_metadynamics.make_int3(int(l[0]), int(l[1]), int(l[2])))
        self.cpp_force = _metadynamics.LamellarOrderParameterGPU(context.current.system_definition, cpp_mode,
                                                                 cpp_lattice_vectors, suffix)

    def set_trig_mode(self, mode):
        """this build: the trigonometry of this variable's kernels — "hardware" (v_sin_f32 / v_cos_f32, like the reference's
        fast::sin / fast::cos), "accurate" (sinpi / cospi; also the mode for unwrapped coordinates far outside the box) or
        "default" (the process-wide default).  Variables evaluated in one fused launch run accurately if any of them asks to."""
        codes = {"default": 0, "hardware": 1, "accurate": 2}
        if mode not in codes:
            raise RuntimeError("Error setting parameters of collective variable.")
        self.cpp_force.setTrigMode(codes[mode])


class aspect_ratio(_collective_variable):
    """cv.py:275-305"""

    def __init__(self, dir1, dir2, name="", sigma=1.0):
        _collective_variable.__init__(self, sigma, name)
        self.cpp_force = _metadynamics.AspectRatio(context.current.system_definition, int(dir1), int(dir2))


class density(_collective_variable):
    """cv.py:308-338 (group = all particles)"""

    def __init__(self, group=None, sigma=1.0):
        name = "all" if group is None else str(group)
        _collective_variable.__init__(self, sigma, name)
        self.cpp_force = _metadynamics.Density(context.current.system_definition, name)


class potential_energy(_collective_variable):
    """Well-tempered ensemble: the potential energy as collective variable (cv.py:469-497)."""

    def __init__(self, sigma=1.0):
        name = "cv_potential_energy"
        _collective_variable.__init__(self, sigma, name)
        self.enabled = False                                          # cv.py:486-487: not a regular ForceCompute
        self.cpp_force = _metadynamics.WellTemperedEnsemble(context.current.system_definition, name)


class mesh(_collective_variable):
    """Particle-mesh order parameter (cv.py:350-466)."""

    def __init__(self, mode, nx, ny=None, nz=None, name=None, sigma=1.0, zero_modes=None):
        if name is not None:
            name = "_" + name
        if ny is None:
            ny = nx
        if nz is None:
            nz = nx
        _collective_variable.__init__(self, sigma, name)
        cpp_mode = _mode_vector(mode, "cv.mesh")
        cpp_zero_modes = _metadynamics.std_vector_int3()
        if zero_modes is not None:
            for l in zero_modes:
                if len(l) != 3:
                    raise RuntimeError("Error creating collective variable.")   # cv.py:403-405
                cpp_zero_modes.append(_metadynamics.make_int3(int(l[0]), int(l[1]), int(l[2])))
        self.cpp_force = _metadynamics.OrderParameterMeshGPU(context.current.system_definition, int(nx), int(ny), int(nz),
                                                             cpp_mode, cpp_zero_modes)

    def set_decomposition(self, kind):
        """this build, domain-decomposed runs: "replicated" (default: every rank keeps the whole mesh, the ranks sum their
        assignments) or "slab" (the mesh itself is decomposed over the ranks; ny and nz multiples of the number of ranks)"""
        if kind not in ("replicated", "slab"):
            raise RuntimeError("Error setting parameters of collective variable.")
        self.cpp_force.setSlabDecomposition(kind == "slab")

    def set_params(self, use_table=None, **args):                  # cv.py:423-436
        if use_table is not None:
            self.cpp_force.setUseTable(use_table)
        _collective_variable.set_params(self, **args)

    def set_kernel(self, func, kmin, kmax, width, coeff=dict()):   # cv.py:438-466
        Ktable, dKtable = [], []
        dk = (kmax - kmin) / float(width - 1)
        for i in range(0, width):
            k = kmin + dk * i
            (K, dK) = func(k, kmin, kmax, **coeff)
            Ktable.append(K)
            dKtable.append(dK)
        self.cpp_force.setTable(Ktable, dKtable, kmin, kmax)


class steinhardt(_collective_variable):
    """Steinhardt Ql (cv.py:540-617): CV = sum_l Ql_ref[l] * Q_l over particles of one type within r_cut."""

    def __init__(self, r_cut, r_on, lmax, Ql_ref, nlist, type, name=None, sigma=1.0):
        suffix = ""
        if name is not None:
            suffix = "_" + name
        _collective_variable.__init__(self, sigma, name)
        self.type = type
        self.nlist = nlist
        self.r_cut = r_cut
        # cv.py:584-585: the reference forces full storage when HOOMD runs on the GPU — this build always does
        self.nlist.cpp_nlist.setStorageMode(_metadynamics.NeighborList.storageMode.full)
        type_list = context.current.type_names
        if type not in type_list:
            raise RuntimeError("Error creating collective variable.")     # cv.py:591-593
        self.cpp_force = _metadynamics.SteinhardtQl(context.current.system_definition, float(r_cut), float(r_on), int(lmax),
                                                    nlist.cpp_nlist, type_list.index(type), [float(q) for q in Ql_ref], suffix)

    def get_rcut(self):
        """cv.py:603-617: the cut-off this CV asks of the neighbour list, by type pair — only (type, type) interacts.
        (The reference body refers to an undefined ``nl``; this returns the plain dict it was building.)"""
        names = context.current.type_names
        return {(a, b): (self.r_cut if a == b == self.type else -1.0) for i, a in enumerate(names) for b in names[i:]}


class nlist_cell(object):
    """Stand-in for ``hoomd.md.nlist.cell``: HOOMD's NeighborList is not part of the plugin.  The list is built on the host
    with a periodic KD-tree (cubic boxes) whenever ``update`` is called and handed to the device in HOOMD's layout."""

    def __init__(self, r_cut):
        self.r_cut = float(r_cut)
        self.cpp_nlist = _metadynamics.NeighborList(context.current.system_definition)

    def set_lists(self, head_list, n_neigh, nlist):
        """hand over a list in HOOMD's layout: neighbours of i are nlist[head_list[i] : head_list[i] + n_neigh[i]]"""
        self.cpp_nlist.setLists(head_list, n_neigh, nlist)

    def update(self):
        """rebuild from the current positions (cubic boxes; host-side periodic KD-tree)"""
        import numpy as np
        from scipy.spatial import cKDTree
        pdata = context.current.system_definition.getParticleData()
        L = np.asarray(pdata.getGlobalBox().getL())
        n_local = pdata.getN()
        # (a shard: the ghost particles sit behind the local ones; heads for the local particles only, entries may index ghosts)
        p = np.mod(np.asarray(pdata.getPositions()[:, :3], dtype=np.float64) + L / 2, L)
        p[p >= L] = 0.0
        pairs = cKDTree(p, boxsize=L).query_pairs(self.r_cut, output_type="ndarray")
        i = np.concatenate([pairs[:, 0], pairs[:, 1]])
        j = np.concatenate([pairs[:, 1], pairs[:, 0]])
        keep = i < n_local
        i, j = i[keep], j[keep]
        order = np.lexsort((j, i))
        i, j = i[order], j[order]
        p = p[:n_local]
        n_neigh = np.bincount(i, minlength=len(p)).astype(np.uint32)
        head = np.zeros(len(p), dtype=np.uint32)
        head[1:] = np.cumsum(n_neigh)[:-1]
        lists = (head, n_neigh, j.astype(np.uint32))
        self.set_lists(*lists)
        return lists


class wrap(_collective_variable):
    """Force wrapper (cv.py:500-537): the energy of an arbitrary force as collective variable."""

    def __init__(self, force, sigma=1.0):
        from . import force as _force_module
        if not isinstance(force, _force_module._force):
            raise RuntimeError("cv.wrap needs a md._force instance as argument.")          # cv.py:518-519
        name = "cv_" + force.name
        _collective_variable.__init__(self, sigma, name)
        self.force = force
        self.cpp_force = _metadynamics.CollectiveWrapper(context.current.system_definition, force.cpp_force, name)
        self.log = force.log

    # cv.py:531-537 call themselves recursively and name an undefined ``force``; the evident intent is kept
    def disable(self, log=False):
        self.enabled = False
        self.log = log
        self.force.enabled = False
        self.force.log = log

    def enable(self):
        self.enabled = True
        self.force.enabled = True
