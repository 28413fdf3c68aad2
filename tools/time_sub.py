"""Times the -s path (POVU_HIP_F_SUBFLUBBLES) on whole-genome-shaped graphs: python tools/time_sub.py <segments> [compare]"""
import sys, time, hashlib, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from povu_amd import HipDecomposer, workloads as W
from povu_amd.hip import F_SUBFLUBBLES
n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
g = W.hprc_whole_genome(n)
hip = HipDecomposer(0)
hip.upload(g)
for rep in range(4):
    t0 = time.time(); f = hip.decompose(flags=F_SUBFLUBBLES); t1 = time.time()
    ms = hip.last_stage_times() if hasattr(hip, "last_stage_times") else {}
    print(f"segments {g.n_vtx} links {g.n_links}: decompose with -s {t1 - t0:.2f} s", {k: round(v, 1) for k, v in dict(ms).items() if "sub" in k or k == "total"}, flush=True)
kinds = {}
for i in range(len(f)):
    st = f.subtree(i)
    for k in ("n_concealed", "n_midi", "n_smothered"):
        kinds[k] = kinds.get(k, 0) + int(st[k])
print(len(f), "trees", kinds, flush=True)
if len(sys.argv) > 2:
    import oracle_lib as O
    t0 = time.time()
    want = O.decompose(g, threads=os.cpu_count(), lpt=True, leaf=2)
    print(f"oracle {time.time() - t0:.1f} s", flush=True)
    got = f.texts()
    bad = [c for c in want if got.get(c) != want[c]]
    print("components", len(want), "mismatching", len(bad), bad[:5])
    sys.exit(1 if bad or got.keys() != want.keys() else 0)
