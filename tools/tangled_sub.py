"""-s on bench.py's tangled workload, whose bracket table has 1.07e10 entries: the device against the oracle (64-bit table
offsets: ~45 GB of host memory).  python tools/tangled_sub.py [compare]"""
import sys, time, os, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench
from povu_amd import HipDecomposer
from povu_amd.hip import F_SUBFLUBBLES
g, wl = bench.build_workload("tangled", 1.0)
print(wl, flush=True)
hip = HipDecomposer(0); hip.upload(g)
for rep in range(2):
    t0 = time.time()
    f = hip.decompose(flags=F_SUBFLUBBLES)
    print('device', round(time.time() - t0, 2), 's', sum(f.subtree(i)['n_concealed'] for i in range(len(f))), 'concealed', flush=True)
if len(sys.argv) > 1:
    import oracle_lib as O
    t0 = time.time()
    want = O.decompose(g, threads=os.cpu_count(), lpt=True, leaf=2)
    print(f"oracle {time.time() - t0:.1f} s", flush=True)
    got = f.texts()
    bad = [c for c in want if got.get(c) != want[c]]
    print("components", len(want), "mismatching", len(bad), bad[:5])
    sys.exit(1 if bad or got.keys() != want.keys() else 0)
