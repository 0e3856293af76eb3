"""``hoomd.metadynamics.integrate`` API (metadynamics/integrate.py:204-357 of the reference)."""
from . import _metadynamics
from . import context
from . import cv


class mode_metadynamics(object):
    """integrate.py:27-226: metadynamics integration mode (grid mode, standard / well-tempered)."""

    def __init__(self, dt, stride, mode="standard", W=1.0, deltaT=1.0, T=1.0, filename="", overwrite=False, add_hills=True):
        if mode == "standard":
            cpp_mode = _metadynamics.IntegratorMetaDynamics.mode.standard
        elif mode == "well_tempered":
            cpp_mode = _metadynamics.IntegratorMetaDynamics.mode.well_tempered
        else:
            raise RuntimeError("Error setting up Metadynamics.")     # integrate.py:214-216
        self.cpp_integrator = _metadynamics.IntegratorMetaDynamics(context.current.system_definition, dt, W, deltaT, T,
                                                                   int(stride), add_hills, filename, overwrite, cpp_mode)
        self.supports_methods = True
        context.current.system.setIntegrator(self.cpp_integrator)
        context.current.integrator = self
        self.cv_names = []

    def update_forces(self):
        """Registers the collective variables with the C++ integration class (integrate.py:228-269)."""
        forces = context.current.forces
        if self.cpp_integrator.isInitialized():
            notfound = False
            num_cv = 0
            for f in forces:
                if isinstance(f, cv._collective_variable) and f.grid_set:
                    if num_cv >= len(self.cv_names) or f.name != self.cv_names[num_cv]:
                        notfound = True
                    num_cv += 1
            if (len(self.cv_names) != num_cv) or notfound:
                raise RuntimeError("Error setting up Metadynamics.")  # integrate.py:239-242
        self.cv_names = []
        self.cpp_integrator.removeAllVariables()
        self.cpp_integrator.removeForceComputes()
        for f in forces:
            if isinstance(f, cv._collective_variable) and f.cpp_force is not None:
                if f.grid_set is True:
                    self.cpp_integrator.registerCollectiveVariable(f.cpp_force, f.sigma, f.cv_min, f.cv_max, f.num_points)
                    self.cv_names.append(f.name)
                # every enabled force is computed by the integrator's computeNetForce (HOOMD: system.addCompute)
                if f.enabled:
                    self.cpp_integrator.addForceCompute(f.cpp_force)
        if not self.cpp_integrator.isInitialized():
            self.cpp_integrator.setGrid(True)                         # integrate.py:266-267

    def dump_grid(self, filename1, filename2="", period=0):          # integrate.py:271-292
        self.cpp_integrator.dumpGrid(filename1, filename2, int(period))

    def restart_from_grid(self, filename):                           # integrate.py:294-306
        self.cpp_integrator.restartFromGridFile(filename)

    def reset_histogram(self):                                       # integrate.py:308-315
        self.cpp_integrator.resetHistogram()

    def set_params(self, add_hills=None, mode=None, stride=None, adaptive=None, sigma_g=None, multiple_walkers=None):
        """integrate.py:317-357"""
        if add_hills is not None:
            self.cpp_integrator.setAddHills(add_hills)
        if mode is not None:
            if mode == "standard":
                cpp_mode = _metadynamics.IntegratorMetaDynamics.mode.standard
            elif mode == "well_tempered":
                cpp_mode = _metadynamics.IntegratorMetaDynamics.mode.well_tempered
            else:
                raise RuntimeError("Error setting up Metadynamics.")
            self.cpp_integrator.setMode(cpp_mode)
        if stride is not None:
            self.cpp_integrator.setStride(int(stride))
        if adaptive is not None:
            self.cpp_integrator.setAdaptive(adaptive)
        if sigma_g is not None:
            self.cpp_integrator.setSigmaG(sigma_g)
        if multiple_walkers is not None:
            self.cpp_integrator.setMultipleWalkers(multiple_walkers)
