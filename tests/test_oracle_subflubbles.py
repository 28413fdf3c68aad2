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


# ---- three more cases traced by hand through the reference's code (one per family the first case does not reach): a TRUNK
# midi bubble, an a-side nesting, a smothered vertex.  The graphs are workloads.random_bidirected(n_vtx, n_links, seed,
# self_loops=False, connected) -- found by a search for the smallest graphs whose oracle run reaches the rule; what is
# asserted is the text derived by hand below, for the oracle here and for the HIP path in tests/test_gpu_subflubbles.py.
# Notation: tree vertex: segment side; "brackets(v)" = the ordinary back edges whose walk from parent(source) up to (not
# including) the target passes v (collect_backedges_by_vertex, tree_utils.cpp:167-216, B = ordinary edges in creation
# order, :640-648); lo[] = compute_LoA (tree_utils.cpp:224-273) run by hand (descending sweep, deepest queued target).
HAND_TRACED = {
    # segments 6 10 11 12 13; links 6l-10l 10r-11l 11r-12l 12l-13r 10r-6l 11l-13l.  Tips 6r, 12r; from_bd starts at (6, r).
    # Tree: 0 D, 1 6r, 2 6l, 3 10l, 4 10r, 5 11l, 6 11r, 7 12l on one path, 8 12r (black) and 9 13r (gray) below 7, 10 13l below
    # 9.  Ordinary back edges 4->2, 8->0 (12r has no link), 10->5; capping 7->5.  Flubble 1 = <6>12: ai 2, zi 7.
    #   compute_m (concealed.cpp:206-253): IBE(2) = {4->2}, lca(4, 7) = 4 above zi -> m = 4.  compute_n (:255-300): 7's only
    #   outgoing edge is the capping one -> n = zi = 7.  can_contain (:318-363): lo[7] = 5 (10->5 is queued when the sweep
    #   reaches 7) is not above ai; m != ai -> yes.
    #   ai_trunk (:367-522): {src 4, lca 4} passes cond i (depth 4 <= depth m); brackets(4) = {8->0} all end at or above ai ->
    #   kept.  gen_ai_slubble (:69-151): a = 6l -> <6; g = vertex 4 = 10r below a black edge -> >10: "C <6>10", route e2s = R.
    #   ai_branches (:525-583): neither child of 7 has hi == lo == ai.
    #   ji_trunk (:702-781) -> override_ji_trunk (:617-700): child 8 of zi is j_x (hi 0 above ai); child 9 has ONE bracket
    #   10->5 with m < 5 < n -> candidate 5; brackets(5) = {8->0}: source not above zi, target not below ai -> 5 wins.
    #   gen_zi_slubble (:153-204): z = 12l -> >12, s = vertex 5 = 11l below a gray edge -> >11: "C >11>12", route s2e = L.
    #   ji_branches (:783-917): zi == n -> none.  add_concealed (:1183-1196): PVST 2 and 3 under flubble 1, nothing to nest.
    #   find_midi (midi.cpp:223-268): flubble 1 has two concealed children; handle_fl (:130-221): both slubble vertices (4, 5)
    #   lie above zi -> the TRUNK case (:195-205); gen_midi_bub (:68-127): g from the ai_trunk vertex (>10), s from the
    #   zi_trunk vertex (>11), route s2e: "M >10>11 L"; add_midi (:18-65): child of flubble 1, no flubble child to nest.
    "midi_trunk": ((5, 6, 526540656, True),
                   [["D", "0", ".", "1", "."], ["F", "1", "<6>12", "2, 3, 4", "L"], ["C", "2", "<6>10", ".", "R"],
                    ["C", "3", ">11>12", ".", "L"], ["M", "4", ">10>11", ".", "L"]]),
    # segments 2 8 10 15 17; links 2l-8r 8r-10l 10l-15r 15l-17r 2r-15l 2r-17l 15r-2l 15r-8r.  Tips 8l, 10r; start (8, l).
    # Local link order (componetize, bidirected.cpp:558-569): 2l-8r, 15r-2l, 2r-15l, 2r-17l, 8r-10l, 15r-8r, 10l-15r, 15l-17r.
    # Tree: 0 D, 1 8l, 2 8r, 3 2l, 4 2r, 5 15l, 6 15r, 7 10l, 8 10r on one path, 9 17r below 5, 10 17l below 9.  Ordinary back
    # edges 6->3, 6->2, 8->0, 7->2, 10->4; capping 5->4.  Flubbles: 1 = >8>10 (ai 2, zi 7), 2 = >2>15 below it (ai 4, zi 5),
    # which find_tiny turns into T (not a flubble any more: find_concealed skips it).
    #   Flubble 1: compute_m: IBE(2) = {6->2, 7->2}; lca(6, 7) = 6 above zi -> candidate 6; lca(7, 7) = 7 is not above zi -> m = 6.
    #   compute_n: 7->2 ends AT ai -> n = zi = 7.  can_contain: lo[7] = 4 (10->4 is queued: the sweep does not ask whether its
    #   source lies below 7) is not above ai -> yes.  ai_trunk: {src 6, lca 6}; brackets(6) = {8->0, 7->2} end at or above ai ->
    #   kept; gen_ai_slubble: a = 8r -> >8, g = vertex 6 = 15r below a black edge -> >15: "C >8>15 R".  zi has one child:
    #   no branches; ji_trunk: child 8 is j_x, the edge 7->2 ends above n -> none.
    #   add_conc_ai (:1046-1075) -> nest_trunk_ai (:953-987): child 2 (T: still fl_like) has zi 5 above g (depth 6 > 5) -> it
    #   moves from flubble 1 to the concealed vertex.
    "nest_trunk_ai": ((5, 8, 239188140, True),
                      [["D", "0", ".", "1", "."], ["F", "1", ">8>10", "3", "L"], ["T", "2", ">2>15", ".", "L"], ["C", "3", ">8>15", "2", "R"]]),
    # segments 3 8 9 11 18; links 18l-9r 8r-9l 3r-9l 9r-11l 8l-9l 8r-3r 9r-11l 18r-9l.  Tips 3l, 11r; start (3, l).
    # Local link order: 3r-9l, 8r-3r, 8l-9l, 8r-9l, 18r-9l, 18l-9r, 9r-11l, 9r-11l.  Tree: 0 D, 1 3l, 2 3r, 3 9l, 4 9r, 5 18l, 6 18r;
    # 7 11l, 8 11r below 4; 9 8l, 10 8r below 3.  Ordinary back edges 6->3, 8->0, 10->2, 10->3 (the second 9r-11l link leads to
    # the tree parent: none); capping 4->3, 3->2.  Flubble 1 = >3>11: ai 2, zi 7.
    #   compute_m: IBE(2) = {10->2} (the capping edge is skipped), lca(10, 7) = 3 above zi -> m = 3; n = zi = 7 (no outgoing
    #   edge); can_contain: lo[7] = 3 not above ai.  ai_trunk: {src 10, lca 3}; brackets(3) = {8->0, 10->2} end at or above ai ->
    #   kept; gen_ai_slubble: a = 3r -> >3, g = vertex 3 = 9l below a GRAY edge -> >9: "C >3>9 R".  Nothing on the z side.
    #   find_smothered (smothered.cpp:385-431): the concealed vertex is an ai_trunk one -> g::trunk (:61-134) over the
    #   children of vertex 3: child 4 has brackets {6->3, 8->0} from two sources -> no; child 9 has {10->2, 10->3}, one
    #   source; the LAST one, 10->3, ends below ai; brackets(10) is empty -> e = comp_e(target 3 = 9l) = <9, g = >9, cn_b is no
    #   ancestor: "S <9>9", route s2e = L.  add_smothered (:352-383): child of the concealed vertex; nest (:332-350) finds
    #   nothing with bounds among its children.
    "smothered_g_trunk": ((5, 8, 219224739, False),
                          [["D", "0", ".", "1", "."], ["F", "1", ">3>11", "2", "L"], ["C", "2", ">3>9", "3", "R"], ["S", "3", "<9>9", ".", "L"]]),
}


def hand_traced_graph(name):
    (nv, ne, seed, conn), _ = HAND_TRACED[name]
    return W.random_bidirected(nv, ne, seed, self_loops=False, connected=conn)


def test_three_more_hand_traced_cases():
    """A trunk midi bubble (midi.cpp:195-205), an a-side nesting (concealed.cpp:953-987) and a smothered vertex
    (smothered.cpp:61-134): the PVST text derived by hand from the reference's code (comments of HAND_TRACED), line by line."""
    for name, (_, want) in HAND_TRACED.items():
        text = O.decompose(hand_traced_graph(name), leaf=2)
        assert list(text) == [1], name
        assert _lines(text[1]) == want, name


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
