import sys, time, hashlib
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import oracle_lib as O
from povu_amd import HipDecomposer, workloads as W
md5=lambda s: hashlib.md5(s.encode()).hexdigest()
n=int(sys.argv[1])
t=time.time(); g=W.hprc_shaped([n, n//3, n//7], seed=99, tiny=1000); print('gen',g.n_vtx,g.n_links,time.time()-t,flush=True)
h=HipDecomposer(0)
t=time.time(); h.upload(g); print('upload',time.time()-t,flush=True)
for i in range(3):
    t=time.time(); f=h.decompose(); dt=time.time()-t
    print('decompose',dt,'edges/s',g.n_links/dt, 'trees',len(f), 'redo',h.seq_redo_count(),flush=True)
print({s['name']:round(s['ms'],3) for s in h.stage_times()},flush=True)
t=time.time(); got={k:md5(v) for k,v in f.texts().items()}; print('format',time.time()-t,flush=True)
t=time.time(); want,info=O.decompose(g,timings=True); print('oracle',time.time()-t,{k:v for k,v in info.items() if k.startswith('t_')},flush=True)
want={k:md5(v) for k,v in want.items()}
print('MATCH',got==want, len(got))
