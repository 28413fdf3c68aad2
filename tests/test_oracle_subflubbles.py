"""The oracle's restatement of the three INSERTING passes of `povu decompose -s` (oracle/povu_oracle_sub.inc:
find_concealed, find_midi, find_smothered).  PARITY UNPINNED: the reference holds no C / M / S line anywhere, so what
can be checked is (a) a case traced by hand through the reference's code, (b) the shape of the output on the
reference's own input graphs, (c) that seeded random graphs reach all three kinds of vertices."""
import collections
import glob
import os

import numpy as np

import oracle_lib as O
from povu_amd import workloads as W
from test_oracle import _load_gfa_links


def _lines(text):
    rows = [l.split("\t") for l in text.splitlines()]
    assert rows[0] == ["H", "0.0.3", ".", ".", "."]
    return rows[1:]


def test_hand_traced_concealed_vertex_under_a_tiny_flubble(golden_dir):
    """tests/golden/gfa/pvst_tests_graph.gfa, traced through the reference by hand.  Spanning tree (vertex: segment side):
    0 D, 1 1L, 2 1R, 3 3L, 4 3R, 5 4L, 6 4R, 7 5L, 8 5R, 9 6L, 10 6R, 11 7L, 12 7R on one path, 13 2R and 14 2L below 5;
    back edges 12->0, 11->5, 9->6, 5->2, 14->3.  Flubble 1 = >1>7 has ai 2, zi 11; flubble 2 = >4>6 (ai 6, zi 9) is tiny.
      compute_m (concealed.cpp:234-283): IBE(2) = {5->2}, lca(5, 11) = 5 above zi -> m = 5;  compute_n (:285-328): the
      only back edge of 11 ends at 5 -> n = 5;  can_contain (:348-384): lo[11] = 3 is not above ai, m != ai -> yes.
      ai_trunk (:388-522): the candidate {5, 5} is erased: bracket 14->3 of vertex 5 ends below ai and 5 has a back edge.
      ai_branches (:525-583): zi has one child.  ji_trunk (:699-760): the child 12 of zi is j_x (hi = 0); the back edge
      11->5 is not above n -> slubble at 5.  gen_zi_slubble (:154-206): z = 7L -> >7, s = 4L over a gray edge -> >4:
      "C >4>7", route s2e = L.
      add_concealed (:1183-1196): tree vertex 1 has a child, so not "leaf"; nest_trunk_zi (:1054-1081): flubble 2 starts
      below n = 5 and does not end below zi -> it leaves flubble 1 and (sic) gets the concealed vertex as ITS child."""
    L = _load_gfa_links(os.path.join(golden_dir, "gfa", "pvst_tests_graph.gfa"))
    text = O.decompose(L, leaf=2)[1]
    assert _lines(text) == [["D", "0", ".", "1", "."], ["F", "1", ">1>7", "3", "L"], ["T", "2", ">4>6", "3", "L"],
                            ["C", "3", ">4>7", ".", "L"]]


def _check_shape(plain, full):
    p, f = _lines(plain), _lines(full)
    n0 = len(p)
    assert len(f) >= n0
    for a, b in zip(p, f[:n0]):  # the flubble-like vertices keep their letter, idx, boundaries and route
        assert a[:3] == b[:3] and a[4] == b[4]
    kinds = [r[0] for r in f[n0:]]
    assert set(kinds) <= set("CMS")
    assert kinds == sorted(kinds, key="CMS".index)  # the passes run one after the other and only append
    listed = collections.Counter()
    for i, r in enumerate(f):
        assert int(r[1]) == i
        if r[3] != ".":
            for c in r[3].split(", "):
                assert 0 < int(c) < len(f)
                listed[int(c)] += 1
        assert r[4] in (".", "L", "R")
    for i, r in enumerate(f[n0:], start=n0):
        assert listed[i] >= 1  # every inserted vertex hangs somewhere
        if r[0] == "M":
            assert r[4] == "L"
    return collections.Counter(kinds)


def test_inserting_passes_on_the_references_own_graphs(golden_dir):
    seen = collections.Counter()
    for path in sorted(glob.glob(os.path.join(golden_dir, "gfa", "**", "*.gfa"), recursive=True)):
        L = _load_gfa_links(path)
        plain, full = O.decompose(L, leaf=1), O.decompose(L, leaf=2)
        assert plain.keys() == full.keys()
        for c in plain:
            seen += _check_shape(plain[c], full[c])
    assert seen["C"] >= 10 and seen["M"] >= 1


def test_seeded_random_graphs_reach_all_three_kinds():
    rng = np.random.default_rng(3)
    seen = collections.Counter()
    for it in range(1500):
        nv = int(rng.integers(5, 30))
        L = W.random_bidirected(nv, int(rng.integers(nv, 3 * nv)), int(rng.integers(1 << 30)), self_loops=bool(it % 2))
        plain, full = O.decompose(L, leaf=1), O.decompose(L, leaf=2)
        for c in plain:
            seen += _check_shape(plain[c], full[c])
    assert seen["C"] and seen["M"] and seen["S"], seen
