"""Stand-in for ``hoomd.md.force``: the forces ``cv.wrap`` can wrap.

In HOOMD any ``md.force._force`` (pair, bond, external ...) can be wrapped (cv.py:500-530).  Those computes are HOOMD
core, not part of the plugin; the stand-alone system offers one whose per-particle force / energy, torque and virial
arrays are prescribed from outside and re-established on every ``compute()``.
"""
import numpy as np

from . import _metadynamics
from . import context


class _force(object):
    def __init__(self, name):
        self.name = name
        self.enabled = True
        self.log = True
        self.cpp_force = None
        context.current.forces.append(self)


class prescribed(_force):
    """force (N,4: xyz + energy), torque (N,4), virial (6,pitch) given as arrays; external_energy a scalar."""

    def __init__(self, force, torque=None, virial=None, external_energy=0.0, name="prescribed"):
        _force.__init__(self, name)
        sysdef = context.current.system_definition
        pdata = sysdef.getParticleData()
        dt = np.float32 if pdata.getDtype() == _metadynamics.MTD_F32 else np.float64
        self.cpp_force = _metadynamics.PrescribedForceCompute(sysdef)
        N = pdata.getN()
        pitch = self.cpp_force.getVirialPitch()
        force = np.ascontiguousarray(force, dtype=dt).reshape(N, 4)
        torque = np.zeros((N, 4), dtype=dt) if torque is None else np.ascontiguousarray(torque, dtype=dt).reshape(N, 4)
        virial = np.zeros((6, pitch), dtype=dt) if virial is None else np.ascontiguousarray(virial, dtype=dt).reshape(6, pitch)
        self.cpp_force.setArrays(force, torque, virial)
        self.cpp_force.setExternalEnergy(float(external_energy))
