import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, oracle_lib as O
from povu_amd import HipDecomposer, workloads as W
from povu_amd.hip import F_SUBFLUBBLES
hip=HipDecomposer(0)
for nv,ne in [(2000,5000),(20000,26000)]:
    g=W.random_bidirected(nv,ne,5,self_loops=False)
    hip.upload(g)
    t0=time.time(); f=hip.decompose(flags=F_SUBFLUBBLES); t1=time.time()
    print(nv,ne,'hip s',round(t1-t0,2), {k:round(v,1) for k,v in hip.last_stage_ms().items() if 'sub' in k or k=='total'} if hasattr(hip,'last_stage_ms') else '')
    t0=time.time(); w=O.decompose(g,leaf=2); t1=time.time(); print('oracle s',round(t1-t0,2))
    t0=time.time(); w=O.decompose(g,leaf=1); t1=time.time(); print('oracle leaf only s',round(t1-t0,2))
