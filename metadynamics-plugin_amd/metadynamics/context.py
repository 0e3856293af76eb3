"""Stand-alone replacement for the slice of ``hoomd.context`` / ``hoomd.init`` / ``hoomd.run`` the plugin's Python API
touches (cv.py:242-267 and integrate.py:204-269 of the reference read ``hoomd.context.current.system_definition``,
``.system``, ``.forces`` and ``hoomd.context.exec_conf``).  Inside a real HOOMD-ROCm build these come from HOOMD.
"""
import numpy as np

from . import _metadynamics

current = None
exec_conf = None


class _context:
    def __init__(self, system_definition, system):
        self.system_definition = system_definition
        self.system = system
        self.forces = []          # every md.force._force created in this context (integrate.py:247 iterates it)
        self.integrator = None
        self.type_names = []


def initialize(positions, types, type_names, box, dtype=np.float32, n_global=None, timestep=0, ghost_positions=None, ghost_types=None):
    """Create a SystemDefinition from a particle snapshot (the role of ``init.read_snapshot``).

    positions (N,3), types (N,) int, type_names list[str], box = L | (Lx,Ly,Lz) | _metadynamics.BoxDim.
    A shard of a domain-decomposed system: ``n_global`` = particles of the whole system, ``ghost_positions`` / ``ghost_types`` =
    the ghost particles of this rank (stored behind the local ones, as HOOMD does; neighbour lists index them from N on).
    """
    global current, exec_conf
    positions = np.asarray(positions)
    N = positions.shape[0]
    if not isinstance(box, _metadynamics.BoxDim):
        L = [float(box)] * 3 if np.isscalar(box) else [float(x) for x in box]
        box = _metadynamics.BoxDim(*L)
    code = _metadynamics.MTD_F32 if np.dtype(dtype) == np.float32 else _metadynamics.MTD_F64
    exec_conf = _metadynamics.ExecutionConfiguration()
    pdata = _metadynamics.ParticleData(N, code, list(type_names), box)
    positions, types = np.asarray(positions, dtype=np.float64).reshape(-1, 3), np.asarray(types, dtype=np.int32)
    if ghost_positions is not None and len(ghost_positions):
        pdata.setNGhosts(len(ghost_positions))
        positions = np.concatenate([positions, np.asarray(ghost_positions, dtype=np.float64).reshape(-1, 3)])
        types = np.concatenate([types, np.asarray(ghost_types, dtype=np.int32)])
    pdata.setPositions(_metadynamics.pack_postype(positions, types, code))
    if n_global is not None:
        pdata.setNGlobal(int(n_global))
    sysdef = _metadynamics.SystemDefinition(pdata, exec_conf)
    current = _context(sysdef, _metadynamics.System(sysdef, int(timestep)))
    current.type_names = list(type_names)
    return current


def run(nsteps):
    """``hoomd.run``: (re-)register the collective variables, then advance nsteps."""
    if current is None or current.integrator is None:
        raise RuntimeError("no integrator defined")
    current.integrator.update_forces()
    current.system.run(int(nsteps))


def set_positions(positions, types):
    pdata = current.system_definition.getParticleData()
    pdata.setPositions(_metadynamics.pack_postype(np.asarray(positions, dtype=np.float64), np.asarray(types, dtype=np.int32),
                                                  pdata.getDtype()))
