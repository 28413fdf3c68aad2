"""povu_amd -- MI355X-native `povu decompose` hot path.

Python here is plumbing only (ctypes over the C ABI in include/povu_hip.h, workload
generators, torch.distributed sharding glue).  The product is the HIP library
(povu_amd/lib/libpovu_hip.so), the C++ `povu` CLI and libpovu_ffi.so.
"""
from .hip import HipDecomposer, HipUnavailable, lib_path  # noqa: F401

__version__ = "0.1.0"
