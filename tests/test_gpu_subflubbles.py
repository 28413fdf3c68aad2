"""GPU parity of ALL five passes of `povu decompose -s` (app/subcommand/decompose.cpp:63-70): POVU_HIP_F_SUBFLUBBLES through
the C ABI against the oracle's sequential restatement (oracle/povu_oracle_sub.inc) -- the PVST text write_pvst makes of
every tree after find_tiny, find_parallel, find_concealed, find_midi and find_smothered.  PARITY UNPINNED: the reference
holds no T / O / C / M / S line anywhere; tests/test_oracle_subflubbles.py has the hand-traced case."""
import collections
import glob
import hashlib
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from povu_amd import HipDecomposer, workloads as W
from povu_amd.hip import F_SUBFLUBBLES
from test_oracle import _load_gfa_links
from test_oracle_subflubbles import HAND_TRACED, RULE_SEEDS, hand_traced_graph, rule_graph, rule_tips

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POVU = os.path.join(ROOT, "povu_amd", "bin", "povu")


@pytest.fixture(scope="module")
def hip():
    d = HipDecomposer(0)
    yield d
    d.close()


def check(hip, g, tips=None):
    hip.upload(g, tips)
    f = hip.decompose(flags=F_SUBFLUBBLES)
    got, want = f.texts(), O.decompose(g, tips=tips, leaf=2)
    assert got.keys() == want.keys()
    seen = collections.Counter()
    for c in want:
        assert got[c] == want[c], f"component {c}:\n--- HIP\n{got[c]}\n--- oracle\n{want[c]}"
        seen.update(l[0] for l in want[c].splitlines())
    return seen


def test_the_references_own_graphs(hip, golden_dir):
    seen = collections.Counter()
    for path in sorted(glob.glob(os.path.join(golden_dir, "gfa", "**", "*.gfa"), recursive=True)):
        seen += check(hip, _load_gfa_links(path))
    assert seen["C"] >= 10 and seen["M"] >= 1


@pytest.mark.parametrize("seed", range(6))
def test_small_random_graphs(hip, seed):
    rng = np.random.default_rng(1000 + seed)
    seen = collections.Counter()
    for it in range(250):
        nv = int(rng.integers(5, 30))
        seen += check(hip, W.random_bidirected(nv, int(rng.integers(nv, 3 * nv)), int(rng.integers(1 << 30)), self_loops=bool(it % 2)))
    assert seen["C"], seen


@pytest.mark.parametrize("seed", range(4))
def test_bubble_zoo_and_hprc_shapes(hip, seed):
    check(hip, W.bubble_zoo(20, 8, seed))
    check(hip, W.hprc_shaped([3000, 1500], seed=seed))
    check(hip, W.nested_towers(6, 3))
    check(hip, W.chain_of_bubbles(200))


def test_all_three_kinds_are_reached(hip):
    """5 000 seeded random graphs: concealed, midi and smothered vertices all occur (and match)."""
    rng = np.random.default_rng(3)
    seen = collections.Counter()
    for it in range(5000):
        nv = int(rng.integers(5, 30))
        seen += check(hip, W.random_bidirected(nv, int(rng.integers(nv, 3 * nv)), int(rng.integers(1 << 30)), self_loops=bool(it % 2)))
    assert seen["C"] and seen["M"] and seen["S"], seen


@pytest.mark.parametrize("seed", range(4))
def test_larger_random_graphs_with_many_components(hip, seed):
    rng = np.random.default_rng(50 + seed)
    for nv, ne in [(300, 420), (2000, 2600), (2000, 5000), (20000, 26000)]:
        check(hip, W.random_bidirected(nv, ne, int(rng.integers(1 << 30)), self_loops=bool(seed % 2)))


def test_config_sized_graphs(hip):
    """BASELINE configs 2 and 3 (chain of 10^6 segments; HPRC-shaped chromosome) and a tangled one: text of every PVST."""
    check(hip, W.chain_of_bubbles(250000))
    check(hip, W.hprc_shaped([600000, 200000], seed=5))
    check(hip, W.hprc_tangled(100000, seed=7, tangle_every=5000, max_tangle=2000))


def test_subtree_arrays_and_the_plain_forest(hip):
    """povu_hip_forest_get_subtree: counts and letters agree with the text; without the flag the forest carries none."""
    g = _load_gfa_links(os.path.join(ROOT, "tests", "golden", "gfa", "downstream_repetitive", "vcfwave-complex-decomposition.gfa"))
    hip.upload(g)
    f = hip.decompose(flags=F_SUBFLUBBLES)
    st = f.subtree(0)
    lines = [l.split("\t") for l in f.texts()[1].splitlines()[1:]]
    assert st["n_total"] == len(lines) and bytes(st["fam"]).decode() == "".join(l[0] for l in lines)
    assert (st["n_concealed"], st["n_midi"], st["n_smothered"]) == tuple(sum(l[0] == k for l in lines) for k in "CMS")
    assert st["n_midi"] == 1  # (the reference's own graph with a midi bubble)
    f0 = hip.decompose()
    with pytest.raises(RuntimeError):
        f0.subtree(0)


def test_cli_subflubbles(tmp_path):
    g = W.random_bidirected(400, 560, 11)
    gfa = tmp_path / "g.gfa"
    gfa.write_text(g.to_gfa())
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([POVU, "decompose", "-i", str(gfa), "-o", str(out), "-s"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = O.decompose(g, leaf=2)
    got = {int(p.stem): p.read_text() for p in out.glob("*.pvst")}
    assert got == want
    out2 = tmp_path / "out2"
    out2.mkdir()
    r = subprocess.run([POVU, "decompose", "-i", str(gfa), "-o", str(out2), "-s", "--gpus", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_config4_whole_genome_full_size_all_five_passes(hip):
    """BASELINE config 4 at full size (10^8 segments, 2 024 components): the text of every PVST after all five passes, by
    md5, against the oracle on all host cores."""
    g = W.hprc_whole_genome(1e8)
    hip.upload(g)
    f = hip.decompose(flags=F_SUBFLUBBLES)
    got = {k: hashlib.md5(v.encode()).hexdigest() for k, v in f.texts().items()}
    n_c = sum(f.subtree(i)["n_concealed"] for i in range(len(f)))
    del f
    want = {k: hashlib.md5(v.encode()).hexdigest() for k, v in O.decompose(g, threads=os.cpu_count() or 1, lpt=True, leaf=2).items()}
    assert got == want
    assert n_c > 100000


@pytest.mark.parametrize("name", sorted(HAND_TRACED))
def test_hand_traced_cases_on_the_device(hip, name):
    """The three cases traced by hand through the reference (tests/test_oracle_subflubbles.py: HAND_TRACED) and the first one
    (pvst_tests_graph.gfa): the HIP path writes the hand-derived lines -- evidence for C / M / S that does not pass through
    the oracle."""
    hip.upload(hand_traced_graph(name))
    text = hip.decompose(flags=F_SUBFLUBBLES).texts()
    assert list(text) == [1]
    rows = [l.split("\t") for l in text[1].splitlines()]
    assert rows[0] == ["H", "0.0.3", ".", ".", "."] and rows[1:] == HAND_TRACED[name][1]


def test_first_hand_traced_case_on_the_device(hip, golden_dir):
    hip.upload(_load_gfa_links(os.path.join(golden_dir, "gfa", "pvst_tests_graph.gfa")))
    rows = [l.split("\t") for l in hip.decompose(flags=F_SUBFLUBBLES).texts()[1].splitlines()]
    assert rows[1:] == [["D", "0", ".", "1", "."], ["F", "1", ">1>7", "3", "L"], ["T", "2", ">4>6", "3", "L"], ["C", "3", ">4>7", ".", "L"]]


@pytest.mark.parametrize("rule", sorted(RULE_SEEDS))
def test_every_rule_the_search_found(hip, rule):
    """One small graph per rule of the three passes (which slubble, which nesting, which smothered vertex ...: the oracle
    counts them, tests/test_oracle_subflubbles.py checks that these graphs reach them)."""
    g = rule_graph(rule)
    check(hip, g, rule_tips(rule, g))


def test_the_literal_heap_of_lo_gives_the_closed_form(hip, monkeypatch):
    """LoA: the device uses a closed form (max-tree over the edges' intervals); the reference's heap, one lane per
    component, is kept for a back edge from a vertex to itself anywhere but at the root (none can arise).
    POVU_HIP_SUB_LITERAL_LOA sends every component through the heap: same PVSTs."""
    monkeypatch.setenv("POVU_HIP_SUB_LITERAL_LOA", "1")
    rng = np.random.default_rng(77)
    for it in range(300):
        nv = int(rng.integers(5, 40))
        check(hip, W.random_bidirected(nv, int(rng.integers(nv, 3 * nv)), int(rng.integers(1 << 30)), self_loops=bool(it % 2)))
    check(hip, W.hprc_tangled(20000, seed=3, tangle_every=2000, max_tangle=500))
    check(hip, W.bubble_zoo(20, 8, 5))


@pytest.mark.parametrize("seed", range(4))
def test_builder_style_graphs_without_tips(hip, seed, monkeypatch):
    """Uploaded with explicit all-zero tips (the library's builder API): the root of every component gets a back edge to
    itself (spanning_tree.cpp:433-438) -- in nobody's bracket table (oracle: "leaf subflubble passes"), and the one entry
    compute_LoA's heap never pops.  It is pushed at tree vertex 0, which the descending sweep visits last (tree_utils.cpp:
    250-271): nobody reads the heap after it, so the closed form holds for tip-less (circular) components too -- until round 5
    they took the reference's heap on one lane, 4 s per 10^6 segments.  POVU_HIP_SUB_FORBID_HEAP makes the pass fail if any
    component would still take the heap; the oracle runs the literal heap."""
    monkeypatch.setenv("POVU_HIP_SUB_FORBID_HEAP", "1")
    for g in (W.bubble_zoo(12, 8, 100 + seed), W.hprc_shaped([2000, 700], seed=seed), W.random_bidirected(300, 420, 900 + seed, self_loops=True)):
        check(hip, g, np.zeros(g.n_vtx, dtype=np.uint8))


def test_closed_form_lo_on_many_tip_less_components(hip, monkeypatch):
    """The closed form of LoA against the oracle's literal heap on 1 500 random tip-less components (every one carries the
    0 -> 0 back edge at its root), self loops and parallel links included, and on a circular chromosome-shaped component of
    2 * 10^5 segments; no component may take the one-lane heap."""
    monkeypatch.setenv("POVU_HIP_SUB_FORBID_HEAP", "1")
    rng = np.random.default_rng(2025)
    for it in range(1500):
        nv = int(rng.integers(3, 60))
        g = W.random_bidirected(nv, int(rng.integers(nv, 3 * nv)), int(rng.integers(1 << 30)), self_loops=bool(it % 3 == 0))
        check(hip, g, np.zeros(g.n_vtx, dtype=np.uint8))
    g = W.hprc_shaped([200000], seed=11)
    check(hip, g, np.zeros(g.n_vtx, dtype=np.uint8))


def test_reader_round_trip_on_the_extended_trees(hip):
    """povu_pvst_parse (read_pvst + comp_heights, from_pvst.cpp:162-302, pvst.hpp:807-836) on what -s wrote: letters,
    boundaries and routes come back, the parent of a vertex is its LAST lister (add_edge), a vertex nobody lists has none."""
    import ctypes as C
    from povu_amd import hip as H
    from test_cabi_and_host import _Doc
    hl = H.load_lib()
    hl.povu_pvst_parse.restype = C.POINTER(_Doc)
    hl.povu_pvst_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    hl.povu_pvst_doc_free.argtypes = [C.POINTER(_Doc)]
    NIL = 0xFFFFFFFF
    seen = collections.Counter()
    for rule in ("nest_trunk_zi", "midi", "smo_g_trunk_src", "smo_nest", "zi_branch"):
        g = rule_graph(rule)
        hip.upload(g, rule_tips(rule, g))
        for text in hip.decompose(flags=F_SUBFLUBBLES).texts().values():
            rows = [l.split("\t") for l in text.splitlines()[1:]]
            err = C.create_string_buffer(256)
            raw = text.encode()
            d = hl.povu_pvst_parse(raw, len(raw), err, 256)
            assert d, err.value
            doc = d.contents
            assert doc.n == len(rows)
            want_parent = [NIL] * len(rows)
            for i, r in enumerate(rows):
                if r[3] != ".":
                    for c in r[3].split(", "):
                        want_parent[int(c)] = i
            assert [doc.parent[i] for i in range(doc.n)] == want_parent
            assert b"".join(doc.type[i] for i in range(doc.n)).decode() == "".join(r[0] for r in rows)
            assert [doc.route[i] for i in range(1, doc.n)] == [0 if r[4] == "L" else 1 for r in rows[1:]]
            seen.update(r[0] for r in rows)
            hl.povu_pvst_doc_free(d)
    assert seen["C"] and seen["M"] and seen["S"]


def test_tangled_workload_whose_bracket_table_would_not_fit(hip):
    """bench.py's tangled workload (5.6 M segments, tangles of up to 3e5): gen_tree_meta's bracket table would hold 1.07e10
    entries there (tree_utils.cpp:167-216).  The oracle materialises it like the reference: 45 GB and 95 s, too much for this
    suite -- `tools/tangled_sub.py compare` is that comparison (profiles/r04_sub_tangled_vs_oracle.log: equal).  The device
    enumerates rows on demand; checked here: the pass completes and is consistent -- the flubble-like vertices are those of
    the leaf passes alone, every inserted vertex hangs somewhere."""
    import bench
    from povu_amd.hip import F_LEAF_SUBFLUBBLES
    g, _ = bench.build_workload("tangled", 1.0)
    hip.upload(g)
    f = hip.decompose(flags=F_SUBFLUBBLES)
    leaf = hip.decompose(flags=F_LEAF_SUBFLUBBLES)
    n_c = 0
    for i in range(len(f)):
        st = f.subtree(i)
        _, _, fam = leaf.sub(i)
        n0 = st["n_flubble_like"]
        assert n0 == len(fam) and bytes(st["fam"][:n0]) == bytes(fam)
        assert set(bytes(st["fam"][n0:]).decode()) <= set("CMS")
        listed = np.zeros(st["n_total"], dtype=np.int64)
        np.add.at(listed, st["child"], 1)
        assert (listed[n0:] >= 1).all()
        n_c += st["n_concealed"]
    assert n_c > 1000
