"""Many small components (default 200 000 of < 50 segments each) + one backbone: HIP path vs the oracle."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib as O
from povu_amd import HipDecomposer, workloads as W
md5 = lambda s: hashlib.md5(s.encode()).hexdigest()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
t = time.time(); g = W.hprc_shaped([50000], seed=5, tiny=n); print('gen', g.n_vtx, g.n_links, round(time.time() - t, 1), flush=True)
h = HipDecomposer(0); h.upload(g)
for i in range(3):
    t = time.time(); f = h.decompose(); dt = time.time() - t
    print(f'decompose {dt*1e3:.1f} ms  {g.n_links/dt:.3e} links/s  components {f.total_components} trees {len(f)}', flush=True)
print({s['name']: round(s['ms'], 3) for s in h.stage_times()}, flush=True)
t = time.time(); got = {k: md5(v) for k, v in f.texts().items()}; print('format', round(time.time() - t, 1), flush=True)
t = time.time(); want = O.decompose(g); print('oracle', round(time.time() - t, 1), flush=True)
print('MATCH', got == {k: md5(v) for k, v in want.items()}, len(got))
