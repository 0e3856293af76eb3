"""MI355X-native metadynamics hot path behind the reference's `hoomd.metadynamics` API.

Sub-modules mirror the reference package (metadynamics/__init__.py:1-2): `cv`, `integrate`.
`_abi` is the ctypes view of the C-ABI library (include/mtd_abi.h).
"""
