"""Worst case for the bridge-decomposed spanning tree: one giant 2-edge-connected class."""
import sys, time, hashlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_lib as O
from povu_amd import HipDecomposer, workloads as W
md5 = lambda s: hashlib.md5(s.encode()).hexdigest()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
g = W.random_bidirected(n, int(n * 1.6), 5, connected=True)
h = HipDecomposer(0)
h.upload(g)
for i in range(2):
    t = time.time(); f = h.decompose(); dt = time.time() - t
    print('decompose', round(dt, 4), 'links/s', round(g.n_links / dt), flush=True)
print({s['name']: round(s['ms'], 3) for s in h.stage_times()}, flush=True)
t = time.time(); want = O.decompose(g); print('oracle', round(time.time() - t, 3))
print('MATCH', {k: md5(v) for k, v in f.texts().items()} == {k: md5(v) for k, v in want.items()})
