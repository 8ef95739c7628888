"""mgcfd — Python plumbing around libmgcfd_hip.so, the MI355X-native MG-CFD hot path.

``mgcfd.api``      ctypes binding of include/mgcfd.h (Solver, Mesh)
``mgcfd.meshgen``  synthetic meshes in the reference's file formats
"""
from . import meshgen  # noqa: F401
from .api import (EXPORTED_SYMBOLS, LIB_PATH, LOOPS, Group, MgcfdError, Mesh, Solver,  # noqa: F401
                  generated_to_levels, load_library, plan_audit, rccl_unique_id)
