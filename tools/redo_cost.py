"""What the guard path costs at size (run on the GPU box): BASELINE config 2 (one component, 10^6 segments) through
(a) the parallel pass, (b) the redo the pass takes when a candidate stack is not laminar (add_flubbles' stack machine on one
lane, on top of the parallel stages), (c) every stage on one lane (what round 2's redo did)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povu_amd import HipDecomposer, workloads as W
from povu_amd.hip import F_FORCE_REDO, F_SEQUENTIAL
g = W.chain_of_bubbles(333333)
hip = HipDecomposer(0); hip.upload(g)
for name, fl in (("parallel pass", 0), ("redo: stack machine only", F_FORCE_REDO), ("all stages on one lane", F_SEQUENTIAL)):
    hip.decompose(flags=fl)
    t = time.perf_counter(); f = hip.decompose(flags=fl); dt = time.perf_counter() - t
    print(f"{name}: {dt * 1e3:.1f} ms", {s["name"]: round(s["ms"], 1) for s in hip.stage_times() if s["ms"] > 1})
