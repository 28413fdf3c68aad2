"""Differential fuzz of all five passes of -s (POVU_HIP_F_SUBFLUBBLES) against the oracle, on the graph families that
reach the rules of the inserting passes (tests/test_oracle_subflubbles.py: RULES).  python tools/fuzz_sub.py <seconds> [stream]"""
import sys, time, collections
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as O
from povu_amd import HipDecomposer, workloads as W
from povu_amd.hip import F_SUBFLUBBLES, F_BIG_CLASS_DFS, F_CHECK_LAMINAR, F_SORTED_ADJ, F_HAIRPINS
from test_oracle_subflubbles import sub_stats, _with_extra

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
hip = HipDecomposer(0)
t0 = time.time(); last = t0; n = 0; kinds = collections.Counter(); rules = collections.Counter()
sub_stats()
while time.time() - t0 < budget:
    if time.time() - last > 60:
        last = time.time(); print('...', n, 'graphs', flush=True)
    seed = int(rng.integers(1 << 30)); fam = n % 6
    if fam == 0:
        nv = int(rng.integers(4, 26)); g = W.random_bidirected(nv, int(nv * rng.uniform(1.0, 2.6)), seed, self_loops=bool(n % 3 == 0), connected=bool(n % 5 == 0))
    elif fam == 1:
        nv = int(rng.integers(20, 120)); g = W.random_bidirected(nv, int(nv * rng.uniform(1.05, 1.6)), seed, self_loops=False, connected=True)
    elif fam == 2:
        g = _with_extra(W.chain_of_bubbles(int(rng.integers(3, 60))), float(rng.uniform(0.05, 0.4)), seed)
    elif fam == 3:
        g = _with_extra(W.bubble_zoo(int(rng.integers(1, 4)), int(rng.integers(2, 12)), seed), float(rng.uniform(0.02, 0.3)), seed + 1)
    elif fam == 4:
        g = _with_extra(W.nested_towers(int(rng.integers(2, 6)), int(rng.integers(1, 4))), float(rng.uniform(0.02, 0.2)), seed)
    else:
        g = W.hprc_tangled(int(rng.integers(300, 6000)), seed=seed, tangle_every=int(rng.integers(100, 2000)), max_tangle=int(rng.integers(20, 800)))
    tips = np.zeros(g.n_vtx, dtype=np.uint8) if n % 13 == 0 else None
    want = O.decompose(g, tips=tips, leaf=2)
    hip.upload(g, tips)
    fl = F_SUBFLUBBLES | [0, F_BIG_CLASS_DFS, F_CHECK_LAMINAR, F_SORTED_ADJ, F_HAIRPINS][n % 5]
    try:
        got = hip.decompose(flags=fl).texts()
    except RuntimeError as e:
        if 'from scratch' in str(e) or 'sequential redo' in str(e):  # (a component that needs the redo: refused by design)
            n += 1; continue
        raise
    if got != want:
        print('SUBFLUBBLE MISMATCH family', fam, 'seed', seed, 'n', g.n_vtx, g.n_links, 'flags', fl, 'tips', tips is not None)
        np.savez('gpurun_out/fuzz_sub_fail.npz', vid=g.vid, v1=g.v1, s1=g.s1, v2=g.v2, s2=g.s2)
        sys.exit(4)
    for t in want.values():
        kinds.update(l[0] for l in t.splitlines() if l[0] in 'CMS')
    n += 1
st = sub_stats()
print('fuzz_sub ok:', n, 'graphs in', round(time.time() - t0, 1), 's; inserted vertices', dict(kinds), '; rules reached', {k: v for k, v in st.items() if v})
