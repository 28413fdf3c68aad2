"""The oracle's restatement of the three INSERTING passes of `povu decompose -s` (oracle/povu_oracle_sub.inc:
find_concealed, find_midi, find_smothered).  PARITY UNPINNED: the reference holds no C / M / S line anywhere, so what
can be checked is (a) a case traced by hand through the reference's code, (b) the shape of the output on the
reference's own input graphs, (c) that seeded random graphs reach all three kinds of vertices."""
import collections
import ctypes as C
import glob
import os

import numpy as np

import oracle_lib as O
from povu_amd import workloads as W
from test_oracle import _load_gfa_links


RULES = ("ai_trunk ai_branch zi_trunk zi_branch nest_trunk_ai nest_branch_ai nest_trunk_zi zi_branch_unnested leaf_by_tree_idx midi "
         "midi_same_kind midi_nest smo_g_trunk_tgt smo_g_trunk_src smo_g_branch smo_s_trunk smo_s_branch_src smo_s_branch_tgt smo_nest "
         "smo_stale_read depth_of_invalid self_loop_in_loa override_ji").split()
# small random graphs (n_vtx, n_links, seed, self_loops, connected of workloads.random_bidirected) that reach a rule of the
# three inserting passes, found by search (the oracle counts which rule fires: orc_sub_stats)
RULE_SEEDS = {
    "ai_trunk": (6, 8, 320817378, False, False), "ai_branch": (4, 6, 171298024, True, False),
    "zi_trunk": (17, 22, 445721484, True, False), "zi_branch": (7, 8, 225429374, False, False),
    "nest_trunk_ai": (9, 14, 263055830, False, False), "nest_trunk_zi": (12, 18, 733972094, False, False),
    "zi_branch_unnested": (9, 22, 72618221, False, False), "midi": (20, 48, 779247219, False, False),
    "midi_same_kind": (14, 27, 70283665, False, False), "smo_g_trunk_tgt": (7, 10, 556213187, False, False),
    "smo_g_trunk_src": (9, 18, 141043583, True, False), "smo_g_branch": (8, 12, 302624343, False, True),
    "smo_s_trunk": (13, 20, 573077957, False, True), "smo_s_branch_src": (15, 20, 600991250, True, False),
    "smo_s_branch_tgt": (10, 12, 521325456, True, True), "depth_of_invalid": (12, 13, 528241921, False, True),
    "override_ji": (7, 13, 811069680, True, True),
    # a chain of bubbles / a bubble zoo with a few random extra links: (family, size..., fraction of extra links, seed)
    "smo_nest": ("chain", 11, 0.2333583231116823, 996436063), "smo_stale_read": ("chain", 32, 0.053941270117384026, 485877343),
    "leaf_by_tree_idx": ("zoo", 1, 9, 0.09662316694210007, 282500235),
    # a graph uploaded without tips (builder style): the root of a tip-less component gets a back edge to itself, which
    # compute_LoA pushes and never pops -- the case where the closed form of LoA does not hold
    "self_loop_in_loa": ("tipless", 9, 16, 1069469600),
}
# Never reached by 10^7 searched graphs, and for a reason: nest_branch_ai wants a bracket of a child's zi that STARTS at the
# flubble's ai (concealed.cpp:1006 reads get_src where get_tgt is meant: a bracket's source lies below, ai above);
# midi_nest wants spanning-tree depths at two PVST indices a few apart to differ by more than the indices do.
UNREACHED_RULES = ("nest_branch_ai", "midi_nest")


def _with_extra(base, frac, seed):
    extra = W.random_bidirected(base.n_vtx, max(1, int(base.n_vtx * frac)), seed, self_loops=False)
    return W._mk(base.vid, np.concatenate([base.v1, extra.v1]), np.concatenate([base.s1, extra.s1]),
                 np.concatenate([base.v2, extra.v2]), np.concatenate([base.s2, extra.s2]))


def rule_tips(rule, g):
    """explicit tips of the rule's graph: None = as the GFA loader infers them"""
    return np.zeros(g.n_vtx, dtype=np.uint8) if RULE_SEEDS[rule][0] == "tipless" else None


def rule_graph(rule):
    p = RULE_SEEDS[rule]
    if p[0] == "tipless":
        return W.random_bidirected(p[1], p[2], p[3], self_loops=False)
    if p[0] == "chain":
        return _with_extra(W.chain_of_bubbles(p[1]), p[2], p[3])
    if p[0] == "zoo":
        return _with_extra(W.bubble_zoo(p[1], p[2], p[4]), p[3], p[4] + 1)
    nv, ne, seed, sl, conn = p
    return W.random_bidirected(nv, ne, seed, self_loops=sl, connected=conn)


def sub_stats(reset=True):
    lib = O.lib()
    lib.orc_sub_stats.argtypes = [C.c_void_p, C.c_int]
    out = (C.c_uint64 * len(RULES))()
    lib.orc_sub_stats(out, 1 if reset else 0)
    return dict(zip(RULES, list(out)))


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


def test_rule_seeds_reach_their_rules():
    assert set(RULE_SEEDS) | set(UNREACHED_RULES) == set(RULES)
    for rule in RULE_SEEDS:
        sub_stats()
        g = rule_graph(rule)
        O.decompose(g, tips=rule_tips(rule, g), leaf=2)
        assert sub_stats()[rule] >= 1, rule
