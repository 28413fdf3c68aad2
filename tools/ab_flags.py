"""A/B timing of decompose flags on the config-2 chain: python tools/ab_flags.py [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from povu_amd import HipDecomposer, workloads as W
from povu_amd import hip as H
K = int(sys.argv[1]) if len(sys.argv) > 1 else 333333
g = W.chain_of_bubbles(K)
h = HipDecomposer(0); h.upload(g)
for name, fl in [("default", 0), ("no_stage_times", H.F_NO_STAGE_TIMES), ("default", 0), ("no_stage_times", H.F_NO_STAGE_TIMES)]:
    for _ in range(3): h.decompose(flags=fl)
    t = time.perf_counter()
    for _ in range(30): f = h.decompose(flags=fl)
    dt = (time.perf_counter() - t) / 30
    tot = [s for s in h.stage_times() if s["name"] == "total"][0]["ms"]
    print(f"{name:16s} wall {dt*1e3:.3f} ms  event-total {tot:.3f} ms", flush=True)
