"""MI355X-native C2-Ray evolve hot path (short-characteristics sweep + photo-ionization rates +
doric chemistry + convergence loop) behind the reference's evolve3D / do_source / global_pass
call surface.  The compute lives in csrc/ (hand-written HIP for gfx950, C ABI in
include/c2ray_hip.h); this package is the Python host mirror used by tests and bench.py.
The Fortran drop-in shim is fortran/evolve_hip.F90.
"""
from ._capi import (load_library, default_params, build_tables, Params, Report, SedParams,  # noqa: F401
                    C2RayHipError, LIB_PATH)
from .evolve import Evolve, HipBackend, static_source_share, balanced_source_shares, box_cost  # noqa: F401
from .testproblem import TestProblem, seeded_sources  # noqa: F401
from . import testproblem  # noqa: F401
from . import fileio  # noqa: F401
from . import _capi  # noqa: F401
