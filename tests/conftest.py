"""pytest configuration: `gpu` marker, import paths, shared fixtures.

Import-path policy: the product package lives in metadynamics-plugin_amd/ (imported as
`metadynamics`), the CPU oracle in oracle/ (imported as `mtd_ref`, tests only).
"""
import os
import sys

import pytest

try:  # the harness uses torch for device buffers: it must be loaded BEFORE libmtd_hip.so so that both share one HIP runtime
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "metadynamics-plugin_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass on a box without a GPU: skip them loudly instead.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu-marked tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def _ensure_built():
    """The shared objects are build products (git-ignored): build them when a fresh checkout has none yet."""
    lib = os.path.join(ROOT, "metadynamics-plugin_amd", "lib", "libmtd_hip.so")
    mod_dir = os.path.join(ROOT, "metadynamics-plugin_amd", "metadynamics")
    have_mod = any(f.startswith("_metadynamics") and f.endswith(".so") for f in os.listdir(mod_dir))
    if not (os.path.exists(lib) and have_mod):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session", autouse=True)
def _built():
    _ensure_built()


@pytest.fixture(scope="session")
def abi():
    from metadynamics import _abi
    _abi.load()
    return _abi


@pytest.fixture(scope="session")
def ref():
    import mtd_ref
    mtd_ref.lib()
    return mtd_ref


@pytest.fixture(autouse=True)
def _accurate_trig_by_default(request):
    """The process-wide DEFAULT of the lamellar kernels' trigonometry (mtd_lamellar_set_fast_trig; the library starts with 1,
    the hardware sine / cosine) applies to every CV set that does not choose a mode of its own: every GPU test starts from
    the accurate functions (0) so that results do not depend on the order the tests run in; tests of the hardware mode
    switch it on themselves or set mtd_lamellar_set::trig_mode."""
    if "gpu" in request.keywords:
        try:
            from metadynamics import _abi
            _abi.load().mtd_lamellar_set_fast_trig(0)
        except Exception:
            pass
    yield
