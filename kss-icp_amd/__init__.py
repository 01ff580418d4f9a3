"""kss-icp_amd -- MI355X-native KSS-ICP registration core (host-side Python plumbing).

The product is libkssicp.so (hand-written HIP kernels + C-ABI, include/kssicp.h) and the C++
mirror of the reference class surface (include/KSS_ICP.hpp ...).  This package only binds the
C-ABI with ctypes for the tests and bench.py.  It never falls back to a CPU implementation:
every compute call raises KssError when the library or a GPU is missing.

The directory name contains a hyphen, so import it through `__graft_entry__.load_package()`
(module name `kss_icp_amd`).
"""
from .binding import (KssError, Context, IcpParams, IcpResult, RegisterResult, Pose, lib_path, load_library,
                      exported_symbols, NSUMS, K_NN_SWEEP, K_CORR_REDUCE, K_PRESHAPE, K_ROT_SEARCH, K_POSE_APPLY, K_GRID_NN, K_GRID_BUILD, K_GRID_CHAIN, K_GRID_CHAIN_PASS, K_RESIDENT, K_RESIDENT_PASS, NN_AUTO, NN_BRUTE, NN_GRID,
                      grid_angles, rotation_candidates, rigid_from_sums, build_library)
from . import synth
from . import shard

__all__ = ["KssError", "Context", "IcpParams", "IcpResult", "RegisterResult", "Pose", "lib_path", "load_library",
           "exported_symbols", "NSUMS", "grid_angles", "rotation_candidates", "rigid_from_sums", "build_library", "synth"]
