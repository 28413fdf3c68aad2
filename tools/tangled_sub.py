import sys, time, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import bench
from povu_amd import HipDecomposer
from povu_amd.hip import F_SUBFLUBBLES
g, wl = bench.build_workload("tangled", 1.0)
print(wl, flush=True)
hip = HipDecomposer(0); hip.upload(g)
for rep in range(2):
    t0=time.time()
    try:
        f = hip.decompose(flags=F_SUBFLUBBLES); print('ok', round(time.time()-t0,2), 's', sum(f.subtree(i)['n_concealed'] for i in range(len(f))), 'concealed', flush=True)
    except RuntimeError as e:
        print('refused:', str(e)[:300], round(time.time()-t0,2), 's', flush=True)
