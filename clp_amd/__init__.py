"""clp_amd -- MI355X-native dual-simplex iteration engine behind coin-or/Clp's plug-in surface.

Only the hot path of the revised dual simplex lives here; the product compute path
is the HIP library ``libclpgpu.so`` (clp_amd/csrc), reached through the C ABI in include/clpgpu.h.
"""
from .mps import read_mps, LpData  # noqa: F401
