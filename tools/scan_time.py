"""Time of the device-wide exclusive scans alone (GPU box): ms per scan and effective GB/s (one read + one write of n words)
for a few sizes, through the timing hook povu_hip_debug_scan_time.  Environment switches of primitives.hip apply."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povu_amd import HipDecomposer
d = HipDecomposer(0)
lib = d._lib
lib.povu_hip_debug_scan_time.restype = C.c_double
lib.povu_hip_debug_scan_time.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]
for n in (1 << 20, 10_000_000, 100_000_000, 200_000_000):
    ms = lib.povu_hip_debug_scan_time(d._ctx, n, 20, 0)
    print(f"n={n:>11} {ms:8.4f} ms  {8 * n / ms / 1e6:8.1f} GB/s (2 x 4 bytes per element)", flush=True)
d.close()
