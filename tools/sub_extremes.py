"""-s (all five passes) on the shapes that stress its tables: deep nests, a million tiny components, hub segments, a dense
random component.  Against the oracle where the oracle can do it in reasonable time.  python tools/sub_extremes.py"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import bench, oracle_lib as O
from povu_amd import HipDecomposer, workloads as W
from povu_amd.hip import F_SUBFLUBBLES
hip = HipDecomposer(0)
COMPARE = len(sys.argv) > 1 and sys.argv[1] == "compare"
LOG = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "sub_extremes.log"), "a")
def say(m):
    print(m, flush=True); LOG.write(m + "\n"); LOG.flush()
def run(name, g, compare=True):
    compare = compare and COMPARE
    say(f"{name}: {g.n_vtx} segments / {g.n_links} links ...")
    hip.upload(g)
    t0 = time.time(); f = hip.decompose(flags=F_SUBFLUBBLES); dt = time.time() - t0
    kinds = [sum(f.subtree(i)[k] for i in range(len(f))) for k in ("n_concealed", "n_midi", "n_smothered")]
    msg = f"{name}: {g.n_vtx} segments / {g.n_links} links, {len(f)} trees, device {dt:.2f} s, C/M/S {kinds}"
    if compare:
        t0 = time.time(); want = O.decompose(g, threads=os.cpu_count(), lpt=True, leaf=2); got = f.texts()
        bad = [c for c in want if got.get(c) != want[c]]
        msg += f", oracle {time.time() - t0:.1f} s, mismatching {len(bad)}"
        assert not bad and got.keys() == want.keys(), msg
    say(msg)
run("nested towers (config 5)", bench.build_workload("nest", 1.0)[0], compare=False)  # (the oracle materialises the bracket table: minutes)
run("deep nest", W.nested_towers(2000, 3))
rng = np.random.default_rng(1)
# a million tiny components
k = 1000000
vid = np.arange(1, 3 * k + 1, dtype=np.uint32)
v1 = np.repeat(np.arange(k) * 3, 3) + np.tile([0, 0, 1], k); v2 = np.repeat(np.arange(k) * 3, 3) + np.tile([1, 2, 2], k)
run("a million triangles", W._mk(vid, v1, np.ones(3 * k, np.uint8), v2, np.zeros(3 * k, np.uint8)))
# hub segments: a star of 200 000 links into one segment, on top of a chain
base = W.chain_of_bubbles(100000)
hv1 = np.zeros(200000, dtype=np.int64); hv2 = rng.integers(1, base.n_vtx, size=200000)
run("hub segment", compare=False, g=W._mk(base.vid, np.concatenate([base.v1, hv1]), np.concatenate([base.s1, np.ones(200000, np.uint8)]),
                         np.concatenate([base.v2, hv2]), np.concatenate([base.s2, np.zeros(200000, np.uint8)])))
run("dense random component", W.random_bidirected(30000, 120000, 7, self_loops=True, connected=True))
run("sparse random, 10^6 segments", W.random_bidirected(1000000, 1100000, 8, self_loops=True), compare=False)  # (bracket table: 1.85e11 entries)
say("extremes ok")
