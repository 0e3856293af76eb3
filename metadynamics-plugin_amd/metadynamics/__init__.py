"""MI355X-native metadynamics hot path behind the reference's ``hoomd.metadynamics`` API.

Sub-modules mirror the reference package (metadynamics/__init__.py:1-2): ``cv``, ``integrate``; ``context`` stands in
for the slice of HOOMD the API touches; ``_abi`` is the ctypes view of the C-ABI library (include/mtd_abi.h);
``_metadynamics`` is the pybind11 module of the C++ host classes (imported lazily: it needs libmtd_hip.so).
"""
# No dependency on PyTorch: the package is numpy + the two shared objects.  (A process that ALSO uses torch must import torch
# first, so that both share torch's bundled HIP runtime — see _abi.load.)
