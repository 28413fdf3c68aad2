"""Differential fuzz of the HIP path against the oracle (run on the GPU box)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as O
from povu_amd import HipDecomposer, workloads as W
from povu_amd.hip import (F_HAIRPINS, F_SEQ_TREE, F_SORTED_ADJ, F_NO_STAGE_TIMES, F_BIG_CLASS_DFS, F_SPARSE_SPLITTERS,
                          F_ALL_VERTEX_CLASSES, F_CHECK_LAMINAR, F_LEAF_SUBFLUBBLES, F_SUBFLUBBLES)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
big = len(sys.argv) > 2 and sys.argv[2] == "big"  # only the mid-size kinds, 10x larger
hip = HipDecomposer(0)
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 12345)  # (argv[3]: another stream of graphs)
t0 = time.time(); n_graphs = 0; n_links = 0; n_black_only = 0; n_redo = 0; n_leaf = 0; n_leaf_redo = 0; n_flagged = 0; n_crossed = 0; n_cross_graphs = 0; n_sub = 0; sub_kinds = {'C': 0, 'M': 0, 'S': 0}
last = t0
while time.time() - t0 < budget:
    if time.time() - last > 60:
        last = time.time(); print('...', n_graphs, 'graphs', flush=True)
    seed = int(rng.integers(1 << 30))
    kind = n_graphs % 10
    if big:
        kind = 6 + n_graphs % 2
    if kind == 0:
        n = int(rng.integers(3, 40)); g = W.random_bidirected(n, int(n * rng.uniform(0.8, 3.5)), seed)
    elif kind == 1:
        n = int(rng.integers(40, 400)); g = W.random_bidirected(n, int(n * rng.uniform(1.0, 2.2)), seed, connected=True)
    elif kind == 2:
        n = int(rng.integers(400, 4000)); g = W.random_bidirected(n, int(n * rng.uniform(1.0, 1.6)), seed)
    elif kind == 3:
        g = W.hprc_shaped([int(rng.integers(50, 3000)) for _ in range(int(rng.integers(1, 5)))], seed=seed, tiny=int(rng.integers(0, 20)))
    elif kind == 4:
        n = int(rng.integers(10, 200)); g = W.random_bidirected(n, int(n * rng.uniform(1.5, 4.0)), seed, self_loops=True, connected=True)
    elif kind == 9:
        g = W.bubble_zoo(int(rng.integers(1, 60)), int(rng.integers(1, 40)), seed)
    elif kind == 8:
        g = W.hprc_tangled(int(rng.integers(500, 20000)), seed=seed, tangle_every=int(rng.integers(200, 3000)), max_tangle=int(rng.integers(50, 3000)))
    elif kind == 6:
        # several union-find tiles (8192 vertices each), multi-block scans, many components
        n = int(rng.integers(9000, 60000)) * (10 if big else 1); g = W.random_bidirected(n, int(n * rng.uniform(0.9, 1.2 if big else 1.5)), seed)
    elif kind == 7:
        m = 10 if big else 1
        g = W.hprc_shaped([int(rng.integers(9000, 40000)) * m, int(rng.integers(50, 9000)) * m], seed=seed, tiny=int(rng.integers(0, 300)) * m)
    else:
        # chains with random extra links (long bridge chains + local tangles)
        base = W.chain_of_bubbles(int(rng.integers(5, 3000)))
        extra = W.random_bidirected(base.n_vtx, int(base.n_vtx * rng.uniform(0.0, 0.3)), seed)
        g = W._mk(base.vid, np.concatenate([base.v1, extra.v1]), np.concatenate([base.s1, extra.s1]),
                  np.concatenate([base.v2, extra.v2]), np.concatenate([base.s2, extra.s2]))
    tips = None
    if n_graphs % 11 == 0:
        tips = np.zeros(g.n_vtx, dtype=np.uint8)  # builder-style graphs without tips
    want = O.decompose(g, tips=tips)
    hip.upload(g, tips)
    flags = [0, F_SEQ_TREE, F_HAIRPINS, F_SORTED_ADJ, F_NO_STAGE_TIMES, F_BIG_CLASS_DFS, F_BIG_CLASS_DFS | F_HAIRPINS, F_SPARSE_SPLITTERS, F_ALL_VERTEX_CLASSES, F_CHECK_LAMINAR][(n_graphs + n_graphs // 10) % 10]
    got = hip.decompose(flags=flags).texts()
    if got != want:
        print('MISMATCH kind', kind, 'seed', seed, 'n', g.n_vtx, g.n_links, 'flags', flags, 'tips', tips is not None)
        np.savez('gpurun_out/fuzz_fail.npz', vid=g.vid, v1=g.v1, s1=g.s1, v2=g.v2, s2=g.s2)
        sys.exit(1)
    if hip.seq_redo_count():  # (no mode of this fuzz forces a redo, and crossings are resolved in place)
        print('UNEXPECTED REDO: kind', kind, 'seed', seed, 'flags', flags)
        sys.exit(2)
    fl_x, cr_x = hip.last_crossings()
    if fl_x:
        # a non-laminar candidate stack.  Expected only where the literal hi_2 rule deviated (then the pass numbered all
        # tree vertices: black_only is off); with exact classes it would contradict DESIGN.md section 4, "Row G"
        n_flagged += fl_x; n_crossed += cr_x; n_cross_graphs += 1
        if hip.last_black_only_classes():
            print('CROSSING INTERVALS WITH EXACT CLASSES: kind', kind, 'seed', seed, 'flags', flags)
            np.savez('gpurun_out/fuzz_redo_exact.npz', vid=g.vid, v1=g.v1, s1=g.s1, v2=g.v2, s2=g.s2)
            sys.exit(2)
    n_black_only += int(hip.last_black_only_classes())
    # the two relabelling passes of -s on top (the oracle materialises the reference's bracket table, quadratic on deep
    # trees: small and mid-size graphs only)
    if not big and kind in (0, 1, 3, 4, 5, 9) and g.n_vtx <= 6000 and n_graphs % 2 == 0:
        want_leaf = O.decompose(g, tips=tips, leaf=True)
        lf = F_LEAF_SUBFLUBBLES | [0, F_HAIRPINS, F_BIG_CLASS_DFS, F_CHECK_LAMINAR][(n_graphs // 2) % 4]
        try:
            got_leaf = hip.decompose(flags=lf).texts()
        except RuntimeError as e:
            if 'from scratch' not in str(e):  # (--hairpins + a component that needs the redo: refused by design)
                raise
            got_leaf = want_leaf
        if got_leaf != want_leaf:
            print('LEAF MISMATCH kind', kind, 'seed', seed, 'n', g.n_vtx, g.n_links, 'flags', lf, 'tips', tips is not None)
            np.savez('gpurun_out/fuzz_leaf_fail.npz', vid=g.vid, v1=g.v1, s1=g.s1, v2=g.v2, s2=g.s2)
            sys.exit(3)
        n_leaf += 1; n_leaf_redo += int(hip.seq_redo_count() > 0)
        # ... and all five passes (find_concealed, find_midi, find_smothered insert vertices)
        want_sub = O.decompose(g, tips=tips, leaf=2)
        sf = F_SUBFLUBBLES | [0, F_BIG_CLASS_DFS, F_CHECK_LAMINAR, F_SORTED_ADJ][(n_graphs // 2) % 4]
        got_sub = hip.decompose(flags=sf).texts()
        if got_sub != want_sub:
            print('SUBFLUBBLE MISMATCH kind', kind, 'seed', seed, 'n', g.n_vtx, g.n_links, 'flags', sf, 'tips', tips is not None)
            np.savez('gpurun_out/fuzz_sub_fail.npz', vid=g.vid, v1=g.v1, s1=g.s1, v2=g.v2, s2=g.s2)
            sys.exit(4)
        n_sub += 1
        for t in want_sub.values():
            for line in t.splitlines():
                if line[0] in sub_kinds:
                    sub_kinds[line[0]] += 1
    n_graphs += 1; n_links += g.n_links
print('fuzz ok:', n_graphs, 'graphs,', n_links, 'links in', round(time.time() - t0, 1), 's;', n_black_only, 'passes numbered black edges only;', n_cross_graphs, 'passes met crossing candidate-stack intervals (all with the literal hi_2 rule deviating):', n_flagged, 'entries flagged,', n_crossed, 'resolved as popped, none redone;', n_leaf, 'graphs also through the leaf subflubble passes (', n_leaf_redo, 'of them with a redone component);', n_sub, 'through all five passes of -s:', sub_kinds)
