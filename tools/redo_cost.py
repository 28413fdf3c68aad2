"""What a candidate stack with crossing intervals costs at size (run on the GPU box): BASELINE config 2 (one component,
10^6 segments) through (a) the parallel pass, (b) the parallel pass with the laminarity check and the crossing resolution
switched on (what a graph on which the literal hi_2 rule deviates pays; until round 4 such a component was redone on one
lane), (c) that one-lane stack machine (test mode POVU_HIP_F_FORCE_REDO), (d) every stage on one lane."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povu_amd import HipDecomposer, workloads as W
from povu_amd.hip import F_CHECK_LAMINAR, F_FORCE_REDO, F_SEQUENTIAL
g = W.chain_of_bubbles(333333)
hip = HipDecomposer(0); hip.upload(g)
for name, fl in (("parallel pass", 0), ("parallel pass + laminarity check + crossing resolution", F_CHECK_LAMINAR),
                 ("test mode: stack machine on one lane", F_FORCE_REDO), ("test mode: all stages on one lane", F_SEQUENTIAL)):
    hip.decompose(flags=fl)
    t = time.perf_counter(); f = hip.decompose(flags=fl); dt = time.perf_counter() - t
    print(f"{name}: {dt * 1e3:.1f} ms", {s["name"]: round(s["ms"], 2) for s in hip.stage_times() if s["ms"] > 0.2})
