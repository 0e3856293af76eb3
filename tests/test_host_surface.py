"""CPU: the pybind11 module and the Python API present the reference's surface — class names, methods,
enums, constructor arity — and the pure host logic that needs no device (box maths, argument validation in Python).
Nothing here touches the GPU: objects that allocate device memory are not constructed."""
import inspect
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mod():
    from metadynamics import _metadynamics
    return _metadynamics


def test_module_classes_and_methods(mod):
    # module `_metadynamics` (module.cc:23-41 of the reference)
    for cls in ("CollectiveVariable", "IntegratorMetaDynamics", "LamellarOrderParameterGPU", "OrderParameterMeshGPU",
                "WellTemperedEnsemble", "CollectiveWrapper", "SteinhardtQl", "AspectRatio", "Density", "std_vector_int3"):
        assert hasattr(mod, cls), cls
    # host-path names of the reference (module.cc:29-31) resolve to the device classes: this build has no CPU path
    assert mod.LamellarOrderParameter is mod.LamellarOrderParameterGPU and mod.OrderParameterMesh is mod.OrderParameterMeshGPU
    # IntegratorMetaDynamics.cc:1315-1349
    for meth in ("registerCollectiveVariable", "removeAllVariables", "isInitialized", "setGrid", "dumpGrid", "restartFromGridFile",
                 "setAddHills", "setMode", "setStride", "setAdaptive", "setSigmaG", "resetHistogram", "setMultipleWalkers"):
        assert hasattr(mod.IntegratorMetaDynamics, meth), meth
    assert {"standard", "well_tempered"} <= set(mod.IntegratorMetaDynamics.mode.__members__)
    # CollectiveVariable.cc:109-130
    for meth in ("getCurrentValue", "setUmbrella", "setKappa", "setWidthFlat", "setMinimum", "setScale", "requiresNetForce"):
        assert hasattr(mod.CollectiveVariable, meth), meth
    assert {"no_umbrella", "linear", "harmonic", "wall", "gaussian"} <= set(mod.CollectiveVariable.umbrella.__members__)
    # OrderParameterMesh.cc:1181-1193
    for meth in ("setTable", "setUseTable"):
        assert hasattr(mod.OrderParameterMeshGPU, meth), meth
    # every CV class derives from CollectiveVariable, which derives from ForceCompute (CollectiveVariable.h:32)
    for cls in ("LamellarOrderParameterGPU", "OrderParameterMeshGPU", "WellTemperedEnsemble", "CollectiveWrapper", "SteinhardtQl",
                "AspectRatio", "Density"):
        assert issubclass(getattr(mod, cls), mod.CollectiveVariable)
    assert issubclass(mod.CollectiveVariable, mod.ForceCompute)


def test_python_api_signatures():
    """metadynamics.cv / metadynamics.integrate: argument names and defaults of the reference (cv.py, integrate.py)"""
    from metadynamics import cv, integrate

    def params(f):
        return [(n, p.default) for n, p in inspect.signature(f).parameters.items() if n != "self"]

    E = inspect.Parameter.empty
    assert params(cv.lamellar.__init__) == [("mode", E), ("lattice_vectors", E), ("name", None), ("sigma", 1.0)]
    assert params(cv.mesh.__init__) == [("mode", E), ("nx", E), ("ny", None), ("nz", None), ("name", None), ("sigma", 1.0),
                                         ("zero_modes", None)]
    assert params(cv.steinhardt.__init__) == [("r_cut", E), ("r_on", E), ("lmax", E), ("Ql_ref", E), ("nlist", E), ("type", E),
                                               ("name", None), ("sigma", 1.0)]
    assert params(cv.potential_energy.__init__) == [("sigma", 1.0)]
    assert params(cv.wrap.__init__) == [("force", E), ("sigma", 1.0)]
    assert params(cv.aspect_ratio.__init__) == [("dir1", E), ("dir2", E), ("name", ""), ("sigma", 1.0)]
    assert params(cv.density.__init__) == [("group", None), ("sigma", 1.0)]
    assert params(cv._collective_variable.set_grid) == [("cv_min", E), ("cv_max", E), ("num_points", E)]
    assert [n for n, _ in params(cv._collective_variable.set_params)] == ["sigma", "kappa", "cv0", "umbrella", "width_flat", "scale",
                                                                          "reweight"]
    assert params(integrate.mode_metadynamics.__init__) == [("dt", E), ("stride", E), ("mode", "standard"), ("W", 1.0), ("deltaT", 1.0),
                                                            ("T", 1.0), ("filename", ""), ("overwrite", False), ("add_hills", True)]
    assert params(integrate.mode_metadynamics.dump_grid) == [("filename1", E), ("filename2", ""), ("period", 0)]
    assert [n for n, _ in params(integrate.mode_metadynamics.set_params)] == ["add_hills", "mode", "stride", "adaptive", "sigma_g",
                                                                              "multiple_walkers"]
    for meth in ("restart_from_grid", "reset_histogram", "update_forces"):
        assert hasattr(integrate.mode_metadynamics, meth)


def test_boxdim_host_maths(mod, ref):
    """BoxDim stand-in vs the oracle's box CVs (AspectRatio.cc:24-57, Density.cc:20-27): plain host arithmetic"""
    b = mod.BoxDim(3.0, 4.0, 6.0)
    assert tuple(b.getL()) == (3.0, 4.0, 6.0)
    s = b.scale(0.5)
    assert tuple(s.getL()) == (1.5, 2.0, 3.0)
    rbox = ref.Box.make([3.0, 4.0, 6.0])
    assert ref.aspect_ratio(rbox, 0, 1) == pytest.approx(3.0 / 4.0)
    assert ref.density(rbox, 36) == pytest.approx(36 / 72.0)
    # pack_postype: type id bit-cast into w like HOOMD's __scalar_as_int
    p = mod.pack_postype(np.array([[1.0, 2.0, 3.0]]), np.array([5], dtype=np.int32), mod.MTD_F32)
    assert p.dtype == np.float32 and p.view(np.int32)[0, 3] == 5
    p = mod.pack_postype(np.array([[1.0, 2.0, 3.0]]), np.array([7], dtype=np.int32), mod.MTD_F64)
    assert p.dtype == np.float64 and p.view(np.int32)[0, 6] == 7       # low word of w


def test_sharded_step_routes_buffers_between_mailbox_and_collective():
    """ShardedBiasStep: float64 buffers up to mailbox_max elements go through the xGMI mailbox, larger ones (the replicated
    mesh) and other dtypes through the process group, None (a part that exchanged for itself: MeshSlabPart) is skipped;
    a backend with a mailbox attached runs its own two-launch step"""
    torch = pytest.importorskip("torch")
    from metadynamics.sharded import ShardedBiasStep

    class Rec:
        def __init__(self):
            self.calls = []

        def all_reduce(self, buf, group=None):
            self.calls.append(int(buf.numel()))

    class Backend:
        mailbox = None

        def __init__(self, bufs):
            self.bufs, self.done = bufs, []

        def cv_pass(self):
            return self.bufs

        def force_pass(self, sums, t):
            self.done.append(t)

        def step_single(self, t):
            self.done.append(("single", t))

    bufs = [torch.zeros(2, dtype=torch.float64), None, torch.zeros(5000, dtype=torch.float64), torch.zeros(3, dtype=torch.int32)]
    box, pg = Rec(), Rec()
    be = Backend(bufs)
    ShardedBiasStep(be, pg, mailbox=box, mailbox_max=64).step(7)
    assert box.calls == [2] and pg.calls == [5000, 3] and be.done == [7]
    pg2 = Rec()
    ShardedBiasStep(Backend(bufs[:1]), pg2).step(0)
    assert pg2.calls == [2]                                # no mailbox: everything through the collective
    be3 = Backend(bufs)
    be3.mailbox = object()
    ShardedBiasStep(be3, pg).step(3)
    assert be3.done == [("single", 3)]


def test_product_package_imports_without_torch():
    """the product (metadynamics package: ctypes view of the C ABI + pybind11 host module + Python API) does not depend on
    PyTorch — torch is the test / bench harness's tool for device buffers"""
    import subprocess
    import sys
    code = """
import sys
sys.path.insert(0, %r)
class Block:
    def find_spec(self, name, path=None, target=None):
        if name == "torch" or name.startswith("torch."):
            raise ImportError("torch blocked")
sys.meta_path.insert(0, Block())
import metadynamics
from metadynamics import _abi, cv, integrate, context
assert _abi.load().mtd_abi_version() >= 1
assert "torch" not in sys.modules
print("ok")
""" % os.path.join(ROOT, "metadynamics-plugin_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]
