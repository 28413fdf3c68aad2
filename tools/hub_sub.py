import sys, time, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from povu_amd import HipDecomposer, workloads as W
from povu_amd.hip import F_SUBFLUBBLES
rng=np.random.default_rng(1)
base = W.chain_of_bubbles(100000)
hv1 = np.zeros(200000, dtype=np.int64); hv2 = rng.integers(1, base.n_vtx, size=200000)
g=W._mk(base.vid, np.concatenate([base.v1, hv1]), np.concatenate([base.s1, np.ones(200000, np.uint8)]), np.concatenate([base.v2, hv2]), np.concatenate([base.s2, np.zeros(200000, np.uint8)]))
hip=HipDecomposer(0); hip.upload(g)
for flags in (0, F_SUBFLUBBLES, F_SUBFLUBBLES):
    t0=time.time(); f=hip.decompose(flags=flags); print('flags',flags,round(time.time()-t0,3),'s', {s['name']:round(s['ms'],1) for s in hip.stage_times() if s['ms']>20}, flush=True)
