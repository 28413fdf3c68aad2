"""BASELINE config 4 shape at full size: 24 components proportional to chr1..22,X,Y summing to
`total` segments (default 1e8) + 2000 tiny components.  HIP path vs the CPU oracle (md5 per tree)."""
import os, sys, time, hashlib, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from povu_amd import HipDecomposer, workloads as W

CHR_MBP = list(W.CHR_MBP)
total = float(sys.argv[1]) if len(sys.argv) > 1 else 1e8
check = (sys.argv[2] if len(sys.argv) > 2 else "oracle")
# backbone -> segments ratio of the generator is ~1.675
sizes = [max(8, int(total / 1.675 * m / sum(CHR_MBP))) for m in CHR_MBP]
t = time.time(); g = W.hprc_shaped(sizes, seed=20260612, tiny=2000)
print(f'gen {g.n_vtx} segments {g.n_links} links in {time.time()-t:.1f}s  rss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss/1e6:.1f} GB', flush=True)
h = HipDecomposer(0)
print('workspace estimate GB', __import__('povu_amd.hip', fromlist=['x']).load_lib().povu_hip_workspace_estimate(g.n_vtx, g.n_links, 2024) / 1e9, flush=True)
t = time.time(); h.upload(g); print(f'upload {time.time()-t:.2f}s', flush=True)
for i in range(3):
    t = time.time(); f = h.decompose(); dt = time.time() - t
    print(f'decompose {dt*1e3:.1f} ms  {g.n_links/dt:.3e} edges/s  trees {len(f)} redo {h.seq_redo_count()}', flush=True)
print({s['name']: round(s['ms'], 3) for s in h.stage_times()}, flush=True)
md5 = lambda s: hashlib.md5(s.encode()).hexdigest()
t = time.time(); got = {k: md5(v) for k, v in f.texts().items()}
print(f'format+hash {time.time()-t:.1f}s  trees {len(got)}', flush=True)
if check == "oracle":
    import oracle_lib as O
    t = time.time(); want, info = O.decompose(g, timings=True)
    print(f'oracle {time.time()-t:.1f}s', {k: round(v, 2) for k, v in info.items() if k.startswith("t_")}, flush=True)
    want = {k: md5(v) for k, v in want.items()}
    print('MATCH', got == want, len(got), len(want), flush=True)
    sys.exit(0 if got == want else 1)
