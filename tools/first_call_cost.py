"""What the first decompose on a fresh context pays beyond the pass itself (arena allocation, pinned result block)."""
import sys, time
sys.path.insert(0, '.')
from povu_amd import HipDecomposer, workloads as W
g = W.hprc_whole_genome(1e8)
d = HipDecomposer(0)
t = time.perf_counter(); d.upload(g); print('upload ms', round((time.perf_counter() - t) * 1e3, 1))
for k in range(3):
    t = time.perf_counter(); f = d.decompose(); dt = (time.perf_counter() - t) * 1e3
    print('decompose', k, 'ms', round(dt, 1)); del f
t = time.perf_counter(); d.upload(g); print('upload again ms', round((time.perf_counter() - t) * 1e3, 1))
