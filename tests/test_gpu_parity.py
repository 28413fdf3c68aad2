"""GPU parity tests: the HIP path (through the C ABI, include/povu_hip.h) against the CPU oracle,
bit-exact on the PVST text, on golden fixtures, seeded random graphs and BASELINE-sized inputs."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from povu_amd import HipDecomposer, workloads as W
from test_oracle import _load_gfa_links, dump_component

pytestmark = pytest.mark.gpu


def md5(s):
    return hashlib.md5(s.encode()).hexdigest()


@pytest.fixture(scope="module")
def hip():
    d = HipDecomposer(0)
    yield d
    d.close()


def gpu_texts(hip, links, tips=None, **kw):
    hip.upload(links, tips)
    return hip.decompose(**kw).texts()


def test_golden_fixtures(hip, golden_dir):
    for p in sorted(glob.glob(os.path.join(golden_dir, "pvst", "*.pvst"))):
        name = os.path.basename(p)[:-5]
        g = _load_gfa_links(os.path.join(golden_dir, "gfa", name + ".gfa"))
        assert gpu_texts(hip, g) == {1: open(p).read()}, name


def test_downstream_repetitive_inputs_match_oracle(hip, golden_dir):
    """the reference's downstream_repetitive input graphs (no PVST answers there: HIP against the oracle)"""
    files = sorted(glob.glob(os.path.join(golden_dir, "gfa", "downstream_repetitive", "*.gfa")))
    assert len(files) == 5
    for p in files:
        g = _load_gfa_links(p)
        want = O.decompose(g)
        assert gpu_texts(hip, g) == want, p
        from povu_amd.hip import F_SEQUENTIAL
        assert gpu_texts(hip, g, flags=F_SEQUENTIAL) == want, p


def test_overlapped_passes_are_bit_exact(hip, golden_dir):
    """POVU_HIP_F_ASYNC: decompose returns when the forest is laid out; the last kernels and the copies of the PVST arrays
    run under the next pass.  Forests of several passes in flight, uploads between them, a result large enough for the
    staged copies, and a graph whose pass runs the laminarity check and resolves a crossing all equal the oracle / the
    one-pass-at-a-time result."""
    from povu_amd.hip import F_ASYNC, F_NO_STAGE_TIMES
    fl = F_ASYNC | F_NO_STAGE_TIMES
    # small results (the emit kernels write straight into the page-locked block), a new graph every pass
    graphs = [W.random_bidirected(400 + 90 * k, 700 + 130 * k, 300 + k) for k in range(6)] + [W.chain_of_bubbles(3000), W.nested_towers(40, 7)]
    forests = []
    for g in graphs:
        hip.upload(g)  # (waits for the tail of the pass before: the graph arena is about to change)
        forests.append(hip.decompose(flags=fl))
    for g, f in zip(graphs, forests):
        assert f.texts() == O.decompose(g)
    # the same graph, passes back to back, nobody waits in between
    g = W.hprc_shaped([30000, 9000, 200], seed=12, tiny=30)
    want = O.decompose(g)
    hip.upload(g)
    forests = [hip.decompose(flags=fl) for _ in range(5)]
    assert forests[-1].pass_ms() > 0 and forests[0].span_ms(forests[-1]) >= forests[-1].pass_ms()
    for f in forests:
        assert f.texts() == want
    # a result of more than 2^20 PVST vertices: the arrays go through the device block and the copy engine
    big = W.chain_of_bubbles(1_150_000)
    hip.upload(big)
    sync = hip.decompose(flags=F_NO_STAGE_TIMES)
    a = json.load(open(os.path.join(golden_dir, "anchors.json")))
    forests = [hip.decompose(flags=fl) for _ in range(3)]
    other = hip.decompose()  # a plain pass behind overlapped ones
    ref = sync.tree(0)
    assert ref.a_id.size == 1_150_001
    for f in forests + [other]:
        t = f.tree(0)
        for k in ("a_id", "z_id", "a_or", "z_or", "parent"):
            assert np.array_equal(getattr(t, k), getattr(ref, k)), k
    assert md5(sync.text(0)) == md5(O.decompose(big)[1])
    del forests, sync, other
    # a pass that runs the laminarity check (and resolves a crossing pair in place)
    z = np.load(os.path.join(golden_dir, "literal_hi2_crossing_stack.npz"))
    bad = W.Links(z["vid"], z["v1"], z["s1"], z["v2"], z["s2"])
    hip.upload(bad)
    f = hip.decompose(flags=fl)
    assert hip.last_laminar_check_ran() and hip.seq_redo_count() == 0 and hip.last_crossings()[1] >= 1
    assert f.texts() == O.decompose(bad)
    assert a  # (anchors loaded: keeps the fixture dir in use)


def test_kernels_started_ahead_of_the_hosts_reads(hip):
    """Without stage timers povu_hip_decompose gives the stream its next kernel BEFORE it waits for the words it needs from the
    device: the re-index's adjacency kernel ahead of the component count (on the assumption: vertices grouped by component, no
    hub, no self loop), the tree stage's first kernel ahead of the component sizes, the pre-order's event kernel ahead of the
    count of overflowed classes (on the assumption that there are none: with large 2-edge-connected classes it runs again behind
    the wave walks), the bracket placement ahead of the class stage's counts, the level kernels ahead of the PVST count.  Graphs
    that keep the assumptions and graphs that break them in every way, each twice on a warm context (the early start needs the
    arena of the pass before) and once with timers: all equal the oracle."""
    from povu_amd.hip import F_NO_STAGE_TIMES
    base = W.hprc_shaped([4000, 1500, 300], seed=5, tiny=20)
    rng = np.random.default_rng(9)
    perm = rng.permutation(base.n_vtx)  # the components interleaved in vertex order: the re-index has to renumber
    inv = np.empty_like(perm)
    inv[perm] = np.arange(base.n_vtx)
    mixed = W._mk(base.vid[perm], inv[base.v1], base.s1, inv[base.v2], base.s2)
    graphs = [base, W.random_bidirected(3000, 5200, 41, self_loops=True), mixed, W.hub_on_chain(2000, 3000),
              W.chain_of_bubbles(5000), W.random_bidirected(2500, 2600, 43, self_loops=False), base,
              W.nested_towers(200, 6), W.hprc_circular(3000), W.random_bidirected(4000, 9000, 47, connected=True), base]
    for g in graphs:
        want = O.decompose(g)
        hip.upload(g)
        for fl in (F_NO_STAGE_TIMES, F_NO_STAGE_TIMES, 0):
            assert hip.decompose(flags=fl).texts() == want


def test_both_forms_of_the_wave_walk_on_every_kind_of_large_class(hip, monkeypatch):
    """Large 2-edge-connected classes are walked by one of two kernels -- with a hot loop for the steps the LDS window answers
    alone, or without window at all --, picked per class by a probe of its records.  Forced either way (POVU_HIP_WALK_ROUTE)
    both kernels have to produce the reference's tree on classes they would never be given: towers and a ring without the
    hot loop, tangles with links all over the index space with it."""
    graphs = [W.nested_towers(300, 5), W.hprc_circular(4000), W.hprc_tangled(3000, seed=7, tangle_every=700, max_tangle=2500),
              W.random_bidirected(5000, 11000, 53, connected=True), W.hub_on_chain(1500, 2500)]
    for g in graphs:
        want = O.decompose(g)
        hip.upload(g)
        for route in ("1", "2", "0"):
            monkeypatch.setenv("POVU_HIP_WALK_ROUTE", route)
            assert hip.decompose().texts() == want, f"route {route}"
    monkeypatch.delenv("POVU_HIP_WALK_ROUTE", raising=False)


def test_lpa_md5(hip, golden_dir):
    a = json.load(open(os.path.join(golden_dir, "anchors.json")))
    g = _load_gfa_links(os.path.join(golden_dir, "gfa", "LPA.gfa"))
    t = gpu_texts(hip, g)
    assert list(t) == [1] and md5(t[1]) == a["md5"]["LPA.gfa"]


@pytest.mark.parametrize("seed", range(40))
def test_random_graphs_match_oracle(hip, seed):
    n = 30 + 17 * seed
    g = W.random_bidirected(n, int(n * (1.1 + 0.05 * (seed % 7))), seed)
    assert gpu_texts(hip, g) == O.decompose(g)


@pytest.mark.parametrize("seed", range(10))
def test_random_connected_stage_parity(hip, seed):
    g = W.random_bidirected(200 + 31 * seed, 330 + 50 * seed, 1000 + seed, connected=True)
    hip.upload(g)
    f = hip.decompose()
    assert f.texts() == O.decompose(g)
    d = dump_component(g, 0)
    t = hip.debug_tree(0)
    assert np.array_equal(t["gid"], d["gid"]) and np.array_equal(t["par"], d["par"])
    assert np.array_equal(t["typ"], d["typ"]) and np.array_equal(t["black"], d["pe_black"])
    s = hip.debug_stack(0)
    assert np.array_equal(s["tree_vtx"], d["s_st_idx"] + 1)
    assert np.array_equal(s["next_seen"], d["next_seen"])
    # classes: equal as partitions of the candidate stack
    def canon(x):
        m = {}
        return [m.setdefault(v, len(m)) for v in x.tolist()]
    assert canon(s["cls"]) == canon(d["s_cls"])
    assert np.array_equal(hip.debug_edge_ids(0), d["pe_id"])


@pytest.mark.parametrize("seed", range(12))
def test_tree_edge_ids_shared_counter(hip, seed):
    """Tree::add_tree_edge / add_be draw their ids from one counter in creation order (spanning_tree.cpp:784-805); the
    conformance export needs the tree-edge ids, which the GPU derives from per-side back-edge counts."""
    n = 40 + 23 * seed
    g = W.random_bidirected(n, int(n * (1.0 + 0.08 * (seed % 6))), 9000 + seed)
    hip.upload(g)
    hip.decompose()
    c = 0
    checked = 0
    while True:
        d = dump_component(g, c)
        if d is None:
            break
        if len(d["gid"]):
            assert np.array_equal(hip.debug_edge_ids(c), d["pe_id"]), (seed, c)
            checked += 1
        c += 1
    assert checked


def test_components_and_skips(hip):
    g = W.hprc_shaped([300, 40, 1200], seed=7, tiny=25)
    hip.upload(g)
    f = hip.decompose()
    want = O.decompose(g)
    assert f.texts() == want
    comp, loc = hip.debug_components(g.n_vtx)
    # first-appearance order of component ids must be 0,1,2,... (ordered by min vertex idx)
    seen = []
    for c in comp.tolist():
        if c not in seen:
            seen.append(c)
    assert seen == list(range(len(seen)))
    for c in range(len(seen)):
        assert np.array_equal(loc[comp == c], np.arange((comp == c).sum()))


def test_in_memory_graph_without_tips(hip, golden_dir):
    g = _load_gfa_links(os.path.join(golden_dir, "gfa", "pvst_tests_graph.gfa"))
    tips = np.zeros(g.n_vtx, dtype=np.uint8)
    assert gpu_texts(hip, g, tips) == O.decompose(g, tips=tips)


def test_sharded_decompose_union_equals_whole(hip):
    g = W.hprc_shaped([500, 800, 200, 350], seed=11, tiny=40)
    want = O.decompose(g)
    hip.upload(g)
    got = {}
    for r in range(3):
        part = hip.decompose(rank=r, world=3).texts()
        assert not (set(part) & set(got))
        got.update(part)
    assert got == want


def test_chain_3333_md5(hip, golden_dir):
    a = json.load(open(os.path.join(golden_dir, "anchors.json")))
    t = gpu_texts(hip, W.chain_of_bubbles(3333))
    assert md5(t[1]) == a["md5"]["chain_of_bubbles:3333"]


def test_nested_towers_md5(hip, golden_dir):
    a = json.load(open(os.path.join(golden_dir, "anchors.json")))
    assert md5(gpu_texts(hip, W.nested_towers(5, 1))[1]) == a["md5"]["nested_towers:5x1"]
    assert md5(gpu_texts(hip, W.nested_towers(1000, 100))[1]) == a["md5"]["nested_towers:1000x100"]


@pytest.mark.parametrize("seed", range(12))
def test_sequential_kernels_match_oracle(hip, seed):
    """POVU_HIP_F_SEQUENTIAL: the one-lane-per-component fallback for every stage."""
    from povu_amd.hip import F_SEQUENTIAL
    n = 50 + 23 * seed
    g = W.random_bidirected(n, int(n * 1.4), 500 + seed)
    hip.upload(g)
    assert hip.decompose(flags=F_SEQUENTIAL).texts() == O.decompose(g)


@pytest.mark.parametrize("seed", range(30))
def test_dense_random_graphs(hip, seed):
    """Higher link density: many capping edges, >= 3 children with hi < v (the hi_2 quirk)."""
    n = 40 + 11 * seed
    g = W.random_bidirected(n, int(n * (2.0 + 0.1 * (seed % 10))), 9000 + seed, connected=True)
    assert gpu_texts(hip, g) == O.decompose(g)


@pytest.mark.parametrize("seed", range(12))
def test_sequential_tree_parallel_rest(hip, seed):
    """POVU_HIP_F_SEQ_TREE: one-lane DFS feeding the parallel class / stack / PVST kernels."""
    from povu_amd.hip import F_SEQ_TREE
    n = 60 + 19 * seed
    g = W.random_bidirected(n, int(n * 1.5), 700 + seed)
    hip.upload(g)
    assert hip.decompose(flags=F_SEQ_TREE).texts() == O.decompose(g)


@pytest.mark.parametrize("seed", range(6))
def test_forced_sequential_redo(hip, seed):
    """The guard path: components the parallel PVST stage would flag are redone sequentially."""
    from povu_amd.hip import F_FORCE_REDO, F_SEQ_TREE
    g = W.hprc_shaped([200 + 50 * seed, 90], seed=seed, tiny=8)
    hip.upload(g)
    want = O.decompose(g)
    assert hip.decompose(flags=F_FORCE_REDO).texts() == want
    assert hip.seq_redo_count() > 0
    assert hip.decompose(flags=F_FORCE_REDO | F_SEQ_TREE).texts() == want
    assert hip.decompose().texts() == want and hip.seq_redo_count() == 0


@pytest.mark.parametrize("seed", range(4))
def test_partial_sequential_redo_keeps_the_parallel_results_of_the_rest(hip, seed):
    """Per-component redo: every second component is flagged as if its candidate stack were not laminar; those go
    through the sequential kernels (a block of their own in the forest), the others keep the dense parallel result."""
    from povu_amd.hip import F_REDO_ODD, F_SEQ_TREE
    g = W.hprc_shaped([300 + 40 * seed, 120, 75, 210], seed=seed, tiny=9 + seed)
    hip.upload(g)
    want = O.decompose(g)
    f = hip.decompose(flags=F_REDO_ODD)
    n_comp = f.total_components
    assert hip.seq_redo_count() == n_comp // 2
    assert f.texts() == want
    packed = f.pack()  # the wire format flattens the two blocks
    assert hip.merge_forests([packed]).texts() == want
    assert hip.decompose(flags=F_REDO_ODD | F_SEQ_TREE).texts() == want
    assert hip.decompose().texts() == want and hip.seq_redo_count() == 0
    assert f.texts() == want  # the earlier forest still owns its blocks


def test_debug_hooks_refuse_the_state_of_a_mixed_pass(hip):
    """After a pass that redid SOME components sequentially the classes and candidate stacks sit in two layouts: the hooks
    return 4 instead of half-valid arrays (and work again after an all-parallel pass)."""
    from povu_amd.hip import F_REDO_ODD
    g = W.hprc_shaped([300, 120, 75, 210], seed=5, tiny=9)
    hip.upload(g)
    hip.decompose(flags=F_REDO_ODD)
    assert 0 < hip.seq_redo_count()
    import ctypes as C
    n = C.c_uint32(0)
    lib, ctx = hip._lib, hip._ctx
    assert lib.povu_hip_debug_stack(ctx, 0, C.byref(n), None, None, None) == 4
    assert lib.povu_hip_debug_edge_ids(ctx, 0, C.byref(n), None) == 4
    cls = np.zeros(4 * g.n_vtx + 8, dtype=np.uint32)
    assert lib.povu_hip_debug_tree(ctx, 0, C.byref(n), None, None, None, cls.ctypes.data) == 4
    assert lib.povu_hip_debug_tree(ctx, 0, C.byref(n), None, None, None, None) == 0 and n.value > 0
    hip.decompose()
    assert lib.povu_hip_debug_stack(ctx, 0, C.byref(n), None, None, None) == 0 and n.value > 0


@pytest.mark.parametrize("seed", range(12))
def test_laminar_check_on_demand_never_fires(hip, seed):
    """The range-min check of the candidate stack's (prev, i) intervals runs only when the literal hi_2 rule capped
    differently from the second-highest reach (Row G has the proof for the other case);
    POVU_HIP_F_CHECK_LAMINAR forces it.  With and without it: same PVSTs, no component sent to the sequential redo."""
    from povu_amd.hip import F_CHECK_LAMINAR
    if seed % 3 == 0:
        g = W.hprc_shaped([900 + 200 * seed, 70], seed=300 + seed, tiny=5)
    elif seed % 3 == 1:
        n = 80 + 31 * seed
        g = W.random_bidirected(n, int(n * (1.3 + 0.2 * (seed % 4))), 9100 + seed, connected=True, self_loops=(seed % 2 == 0))
    else:
        g = W.hprc_tangled(4000, seed=seed, tangle_every=900, max_tangle=500)
    want = O.decompose(g)
    hip.upload(g)
    assert hip.decompose(flags=F_CHECK_LAMINAR).texts() == want and hip.seq_redo_count() == 0
    assert hip.decompose().texts() == want and hip.seq_redo_count() == 0


def test_non_laminar_stack_is_resolved_in_place(hip, golden_dir):
    """A real crossing pair (tests/test_laminar_fuzz.py::test_the_literal_hi2_rule_can_cross_intervals): the literal hi_2 rule
    deviates, the pass numbers all tree vertices, the laminarity check runs by itself, flags the entry whose interval is
    crossed, and k_resolve_crossings finds that its class had been popped off add_flubbles' stack: its U event is dropped and
    the closed form goes on -- bit-exact, and no component is redone by the one-lane kernels (until round 4 this one was)."""
    from povu_amd.hip import F_CHECK_LAMINAR, F_FORCE_REDO
    d = np.load(os.path.join(golden_dir, "literal_hi2_crossing_stack.npz"))
    g = W._mk(d["vid"], d["v1"], d["s1"], d["v2"], d["s2"])
    want = O.decompose(g)
    hip.upload(g)
    for flags in (0, F_CHECK_LAMINAR):
        assert hip.decompose(flags=flags).texts() == want
        assert hip.seq_redo_count() == 0 and not hip.last_black_only_classes() and hip.last_laminar_check_ran()
        assert hip.last_crossings() == (1, 1)
    assert hip.decompose(flags=F_FORCE_REDO).texts() == want and hip.seq_redo_count() == 1  # (the one-lane machine agrees)


@pytest.mark.parametrize("seed", range(6))
def test_heavily_crossing_candidate_stacks(hip, seed):
    """Tangled graphs on which the literal hi_2 rule deviates in many places: whatever the laminarity check flags is resolved
    in place; the PVSTs equal the oracle's (which runs the literal stack machine) and nothing is redone."""
    g = W.hprc_tangled(6000 + 900 * seed, seed=100 + seed, tangle_every=400, max_tangle=350)
    hip.upload(g)
    f = hip.decompose()
    assert f.texts() == O.decompose(g)
    assert hip.seq_redo_count() == 0


def test_sequential_redo_of_a_million_segment_component(hip, golden_dir):
    """What the guard path costs at size: BASELINE config 2 (one component of 10^6 segments) forced through the one-lane
    kernels.  Bit-exact against the reference md5; the time is printed (`pytest -s`) and bounded loosely -- the redo is a
    single lane walking the whole component, seconds where the parallel pass takes 1.3 ms."""
    import time
    from povu_amd.hip import F_FORCE_REDO
    a = json.load(open(os.path.join(golden_dir, "anchors.json")))
    g = W.chain_of_bubbles(333333)
    hip.upload(g)
    t0 = time.perf_counter()
    f = hip.decompose(flags=F_FORCE_REDO)
    dt = time.perf_counter() - t0
    assert hip.seq_redo_count() == 1
    assert md5(f.text(0)) == a["md5"]["chain_of_bubbles:333333"]
    print(f"sequential redo of 10^6 segments / 2*10^6 links: {dt:.2f} s")
    assert dt < 300


@pytest.mark.parametrize("seed", range(10))
def test_both_class_walks(hip, seed):
    """The per-class DFS has two walks: one lane per class (small classes) and the wave-cooperative walk over
    class-filtered candidate records that a class gets once it turns out larger than the lane's budget;
    POVU_HIP_F_BIG_CLASS_DFS sends every class through the latter.  Dense random graphs = few large 2-edge-connected
    classes with many in-class back edges."""
    from povu_amd.hip import F_BIG_CLASS_DFS
    n = 60 + 37 * seed
    g = W.random_bidirected(n, int(n * (1.2 + 0.25 * (seed % 5))), 8800 + seed, connected=(seed % 2 == 0), self_loops=(seed % 3 == 0))
    want = O.decompose(g)
    hip.upload(g)
    assert hip.decompose(flags=F_BIG_CLASS_DFS).texts() == want
    assert hip.decompose().texts() == want


@pytest.mark.parametrize("seed", range(6))
def test_sparse_list_ranking_splitters(hip, seed):
    """Lists of 2^26+ elements are ranked with 1-in-16 instead of 1-in-8 random splitters; the flag forces that
    density on small inputs (many components = many lists, long chains = long lists)."""
    from povu_amd.hip import F_SPARSE_SPLITTERS
    g = W.hprc_shaped([3000 + 700 * seed, 40 + seed], seed=60 + seed, tiny=30) if seed % 2 else \
        W.random_bidirected(500 + 300 * seed, 700 + 400 * seed, 6100 + seed)
    hip.upload(g)
    assert hip.decompose(flags=F_SPARSE_SPLITTERS).texts() == O.decompose(g)


def test_large_class_takes_the_filtered_walk(hip):
    """One 2-edge-connected class of ~6000 sides (a random connected multigraph): the size sample must pick
    the filtered walk by itself, and the result equals the oracle's."""
    g = W.random_bidirected(3000, 4200, 77, connected=True)
    hip.upload(g)
    assert hip.decompose().texts() == O.decompose(g)


@pytest.mark.parametrize("seed", range(10))
def test_both_local_adjacency_builders(hip, seed):
    """Row B's per-side adjacency: insertion sort per side (default) and the radix-sort builder hub graphs
    take (POVU_HIP_F_SORTED_ADJ) give the same forest; graphs here are rich in self loops of every kind
    (l-l, r-r, l-r) and repeated links, which componetize stores as (ve, complement(ve)) (bidirected.cpp:529-531)."""
    from povu_amd.hip import F_SORTED_ADJ
    rng = np.random.default_rng(4200 + seed)
    n = 30 + 17 * seed
    base = W.random_bidirected(n, int(n * 1.6), 4300 + seed, connected=(seed % 2 == 0))
    k = n // 2
    lv = rng.integers(0, n, k).astype(np.uint32)
    g = W.Links(vid=base.vid,
                v1=np.concatenate([base.v1, lv, base.v1[:k]]), s1=np.concatenate([base.s1, rng.integers(0, 2, k).astype(np.uint8), base.s1[:k]]),
                v2=np.concatenate([base.v2, lv, base.v2[:k]]), s2=np.concatenate([base.s2, rng.integers(0, 2, k).astype(np.uint8), base.s2[:k]]))
    # interleave so that loops and repeats are not all at the end of the L-line order
    order = rng.permutation(g.n_links)
    g = W.Links(vid=g.vid, v1=g.v1[order], s1=g.s1[order], v2=g.v2[order], s2=g.s2[order])
    want = O.decompose(g)
    hip.upload(g)
    assert hip.decompose().texts() == want
    assert hip.decompose(flags=F_SORTED_ADJ).texts() == want


@pytest.mark.parametrize("n", [1, 2, 63, 4095, 4096, 4097, 8191, 8192, 8193, 16385, 100003, 64 * 8192, 64 * 8192 + 1, 65 * 8192 + 7, 4 * 1024 * 1024 + 5, 30_000_001,
                               48 * 1024 * 1024 + 12_345])  # (from 48 * 2^20 elements: one launch with decoupled look-back)
def test_single_pass_scans(hip, n):
    """The device-wide exclusive scans against numpy: sum mod 2^32, running maximum, and two independent
    scans sharing their launches."""
    rng = np.random.default_rng(n)
    a = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 7, max(1, n // 3), dtype=np.uint64).astype(np.uint32)
    z = np.zeros(1, dtype=np.uint64)  # (a python 0 would promote the concatenation to float64)
    want_sum = np.concatenate([z, np.cumsum(a.astype(np.uint64))[:-1]]) & np.uint64(0xFFFFFFFF)
    want_max = np.concatenate([z.astype(np.uint32), np.maximum.accumulate(a)[:-1]])
    want_b = np.concatenate([z, np.cumsum(b.astype(np.uint64))[:-1]])
    for _ in range(2):
        assert np.array_equal(hip.debug_scan(a, 0), want_sum.astype(np.uint32))
        assert np.array_equal(hip.debug_scan(a, 1), want_max.astype(np.uint32))
        got_a, got_b = hip.debug_scan(a, 0, b)
        assert np.array_equal(got_a, want_sum.astype(np.uint32)) and np.array_equal(got_b, want_b.astype(np.uint32))


# ---- BASELINE.json full-size configurations
def test_config2_chain_1m_nodes_bit_exact_vs_reference_md5(hip, golden_dir):
    a = json.load(open(os.path.join(golden_dir, "anchors.json")))
    t = gpu_texts(hip, W.chain_of_bubbles(333333))
    assert list(t) == [1]
    assert len(t[1].encode()) == a["bytes"]["chain_of_bubbles:333333"]
    assert md5(t[1]) == a["md5"]["chain_of_bubbles:333333"]


def test_config5_deep_nest_10m_links_vs_oracle(hip):
    g = W.nested_towers(1000, 3333)
    assert g.n_links > 9_900_000
    got = gpu_texts(hip, g)
    want = O.decompose(g)
    assert list(got) == [1] and md5(got[1]) == md5(want[1])
    # PVST recursion depth 1000: the deepest flubble chain
    depth = {0: 0}
    hip.upload(g)
    t = hip.decompose().tree(0)
    par = t.parent
    d = np.zeros(len(par), dtype=np.int64)
    for i in range(1, len(par)):
        d[i] = d[par[i]] + 1
    assert d.max() == 1000


def test_config3_hprc_shaped_component_vs_oracle(hip):
    g = W.hprc_shaped([4_000_000], seed=20260612)  # one component, backbone N = 4e6
    assert g.n_links > 8_000_000
    got = gpu_texts(hip, g)
    want = O.decompose(g)
    assert {k: md5(v) for k, v in got.items()} == {k: md5(v) for k, v in want.items()}


def test_config4_many_components_sharded_vs_oracle(hip):
    sizes = [int(x) for x in np.linspace(60_000, 8_000, 24)]
    g = W.hprc_shaped(sizes, seed=4, tiny=500)
    want = {k: md5(v) for k, v in O.decompose(g).items()}
    hip.upload(g)
    got = {}
    for r in range(8):
        part = hip.decompose(rank=r, world=8).texts()
        got.update({k: md5(v) for k, v in part.items()})
    assert got == want


def test_config4_whole_genome_full_size_vs_oracle():
    """BASELINE config 4 at FULL size (99.9 M segments / 122.4 M links / 2 024 components): every PVST of the HIP path
    against the oracle (md5 per tree), on one GPU and through the sharded path (partition for 8 ranks, every shard
    decomposed in turn, forests merged).  The oracle runs its components on the host's cores (LPT threads)."""
    from povu_amd import HipDecomposer
    g = W.hprc_whole_genome(1e8)
    assert g.n_links == 122435438 and g.n_vtx == 99860187
    cores = min(32, len(os.sched_getaffinity(0)))
    want = {k: md5(v) for k, v in O.decompose(g, threads=cores, lpt=True).items()}
    assert len(want) == 2024
    d = HipDecomposer(0)
    d.upload(g)
    f = d.decompose()
    assert d.seq_redo_count() == 0
    got = {k: md5(v) for k, v in f.texts().items()}
    del f
    assert got == want
    # the strong-scaling path of bench.py --gpus 8 on the same graph
    work = HipDecomposer(0)
    sh = d.partition(8)
    loads = [sh.info(r)["weight"] for r in range(8)]
    assert max(loads) <= 1.05 * (sum(loads) / 8)
    packed = []
    for r in range(8):
        i = sh.info(r)
        work.upload_shard(i["device_ptr"], i["bytes"], on_device=True)
        packed.append(work.decompose_shard().pack())
    merged = work.merge_forests(packed)
    del packed
    assert {k: md5(v) for k, v in merged.texts().items()} == want
    del merged, sh
    work.close()
    d.close()


@pytest.mark.parametrize("seed", range(3))
def test_tangled_hprc_shape_vs_oracle(hip, seed):
    """HPRC-shaped chain of bubbles with heavy-tailed TANGLES (random 2-edge-connected blocks of 10^2 .. 10^4.8 segments,
    links in random order, random sides): large classes next to millions of small ones, both class walks."""
    from povu_amd.hip import F_BIG_CLASS_DFS
    g = W.hprc_tangled(120000, seed=seed, tangle_every=12000, max_tangle=60000)
    want = {k: md5(v) for k, v in O.decompose(g).items()}
    hip.upload(g)
    assert {k: md5(v) for k, v in hip.decompose().texts().items()} == want
    assert hip.seq_redo_count() == 0
    assert {k: md5(v) for k, v in hip.decompose(flags=F_BIG_CLASS_DFS).texts().items()} == want


def test_circular_component_and_hub_workloads_vs_oracle(hip):
    """The two cliff workloads of bench.py's `secondary` at a size the oracle does in seconds: a tip-less (circular) HPRC-shaped
    component -- ONE 2-edge-connected class, rooted at (l, vertex 0) with the 0 -> 0 back edge, walked by one wave with the
    black follow-through -- and a chain of bubbles with a hub segment (one side with 4 * 10^4 links, the dense re-index);
    plain passes and all five passes of -s."""
    from povu_amd.hip import F_SUBFLUBBLES
    for g in (W.hprc_circular(60000), W.hub_on_chain(20000, 40000)):
        want = {k: md5(v) for k, v in O.decompose(g).items()}
        hip.upload(g)
        assert {k: md5(v) for k, v in hip.decompose().texts().items()} == want
        assert hip.seq_redo_count() == 0
    g = W.hprc_circular(20000)
    hip.upload(g)
    assert hip.decompose(flags=F_SUBFLUBBLES).texts() == O.decompose(g, leaf=2)


def test_extracted_reference_structures_on_gpu(hip, golden_dir):
    """The structure expectations of the reference's conformance suite (tests/golden/reference_vectors.json)."""
    from test_oracle import check_structure, links_of_vector, reference_vectors
    for fid, want in sorted(reference_vectors(golden_dir).items()):
        t = gpu_texts(hip, links_of_vector(want))
        assert list(t) == [1], fid
        check_structure(t[1], want)


def test_cli_and_ffi_on_gpu(tmp_path, golden_dir):
    """The two drop-in surfaces end to end: `povu decompose` files and povu_graph_find_flubbles."""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    povu = os.path.join(root, "povu_amd", "bin", "povu")
    # chain 1>2>3>4 with a bubble 2>6>3, a hairpin 3+ -> 5+/5- and a self loop on 7: one hairpin boundary
    hl = [(0, 1, 1, 0), (1, 1, 2, 0), (2, 1, 3, 0), (1, 1, 5, 0), (5, 1, 2, 0), (2, 1, 4, 0), (2, 1, 4, 1), (3, 1, 6, 0),
          (6, 1, 6, 1)]
    h = W._mk(np.arange(1, 8), [a for a, _, _, _ in hl], [x for _, x, _, _ in hl], [b for _, _, b, _ in hl],
              [x for _, _, _, x in hl])
    r0 = W.random_bidirected(120, 170, 31)
    off = h.n_vtx
    g = W._mk(np.concatenate([h.vid, r0.vid + 100]), np.concatenate([h.v1, r0.v1 + off]), np.concatenate([h.s1, r0.s1]),
              np.concatenate([h.v2, r0.v2 + off]), np.concatenate([h.s2, r0.s2]))
    gfa = tmp_path / "g.gfa"
    gfa.write_text(g.to_gfa())
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ, POVU_STAGE_COST_TRACE="1")
    r = subprocess.run([povu, "-t", "2", "decompose", "-i", str(gfa), "-o", str(out), "-h"], capture_output=True,
                       text=True, env=env)
    assert r.returncode == 0, r.stderr
    want = O.decompose(g)
    got = {int(p.name[:-5]): p.read_text() for p in out.glob("*.pvst")}
    assert got == want
    assert "povu-stage-cost contract=hip:" in r.stderr
    exp_b = []
    c = 0
    while True:
        d = dump_component(g, c)
        if d is None:
            break
        exp_b += [f"Boundary: {int(b1)} {int(b2)}" for b1, b2 in d["bry"].tolist()]
        c += 1
    assert [l for l in r.stderr.splitlines() if l.startswith("Boundary: ")] == exp_b
    assert "Boundary: 7 1" in exp_b
    # LPA through the CLI: md5 anchor of the reference output
    out2 = tmp_path / "lpa"
    out2.mkdir()
    r = subprocess.run([povu, "decompose", "-i", os.path.join(golden_dir, "gfa", "LPA.gfa"), "-o", str(out2)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr == ""
    a = json.load(open(os.path.join(golden_dir, "anchors.json")))
    assert md5((out2 / "1.pvst").read_text()) == a["md5"]["LPA.gfa"]
    # FFI
    from test_cabi_and_host import _Err, _ffi
    lib = _ffi()
    lib.povu_graph_decompose.restype = C.c_void_p
    lib.povu_graph_decompose.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(_Err)]
    lib.povu_forest_tree_count.restype = C.c_size_t
    lib.povu_forest_tree_count.argtypes = [C.c_void_p]
    lib.povu_forest_component_id.restype = C.c_uint32
    lib.povu_forest_component_id.argtypes = [C.c_void_p, C.c_size_t]
    lib.povu_forest_pvst_text.restype = C.c_void_p
    lib.povu_forest_pvst_text.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.povu_string_free.argtypes = [C.c_void_p]
    lib.povu_forest_free.argtypes = [C.c_void_p]
    lib.povu_flubbles_free.argtypes = [C.c_void_p]
    err = _Err(0, None)
    gh = lib.povu_graph_from_gfa(str(gfa).encode(), C.byref(err))
    assert gh
    fl = lib.povu_graph_find_flubbles(gh, C.byref(err))
    assert fl, err.message
    n_flub = sum(t.count("\nF\t") for t in want.values())
    assert lib.povu_flubbles_count(fl) == n_flub + 1
    assert lib.povu_flubbles_get(fl, 0) is None
    lib.povu_flubbles_free(fl)
    fo = lib.povu_graph_decompose(gh, 0, 0, C.byref(err))
    assert fo
    texts = {}
    for i in range(lib.povu_forest_tree_count(fo)):
        ln = C.c_size_t(0)
        p = lib.povu_forest_pvst_text(fo, i, C.byref(ln))
        texts[lib.povu_forest_component_id(fo, i)] = C.string_at(p, ln.value).decode()
        lib.povu_string_free(p)
    assert texts == want
    lib.povu_forest_free(fo)
    lib.povu_graph_free(gh)
    # builder graph (no tips, like tests/integration_tests/pvst_tests.cc): PVST = . -> >1>7 -> >4>6
    b = lib.povu_graph_new(7, 10, 0)
    for i in range(1, 8):
        lib.povu_graph_add_vertex(b, i, b"A")
    for a_, ao, b_, bo in [(1, 1, 3, 0), (1, 1, 4, 0), (2, 1, 4, 0), (3, 1, 4, 0), (3, 0, 2, 0), (4, 1, 5, 0), (4, 1, 6, 0),
                           (4, 0, 7, 0), (5, 1, 6, 0), (6, 1, 7, 0)]:
        lib.povu_graph_add_edge(b, a_, ao, b_, bo)
    fo = lib.povu_graph_decompose(b, 0, 0, C.byref(err))
    assert fo and lib.povu_forest_tree_count(fo) == 1
    ln = C.c_size_t(0)
    p = lib.povu_forest_pvst_text(fo, 0, C.byref(ln))
    text = C.string_at(p, ln.value).decode()
    lib.povu_string_free(p)
    assert sorted(l.split("\t")[2] for l in text.splitlines()[1:]) == sorted([".", ">1>7", ">4>6"])
    lib.povu_forest_free(fo)
    lib.povu_graph_free(b)


def _mk(vid, links):
    return W._mk(np.array(vid), [a for a, _, _, _ in links], [x for _, x, _, _ in links], [b for _, _, b, _ in links],
                 [x for _, _, _, x in links])


def test_edge_cases_match_oracle(hip):
    R, L = W.R, W.L
    cases = {
        "single vertex": _mk([7], []),
        "two vertices": _mk([1, 2], [(0, R, 1, L)]),
        "no links, five vertices": _mk([1, 2, 3, 4, 5], []),
        "three-vertex chain": _mk([1, 2, 3], [(0, R, 1, L), (1, R, 2, L)]),
        "only self loops": _mk([1, 2, 3], [(0, R, 0, L), (1, L, 1, L), (2, R, 2, R)]),
        "triangle without tips": _mk([1, 2, 3], [(0, R, 1, L), (1, R, 2, L), (2, R, 0, L)]),
        "parallel links both orientations": _mk([1, 2, 3, 4], [(0, R, 1, L), (1, L, 0, R), (1, R, 2, L), (1, R, 2, L),
                                                                (2, R, 3, L), (0, R, 3, L)]),
        "ids not ascending in idx": _mk([50, 10, 40, 20, 30], [(0, R, 1, L), (1, R, 2, L), (0, R, 2, L), (2, R, 3, L),
                                                                 (3, R, 4, L), (2, R, 4, L)]),
        "huge ids": _mk([1, 4294967000, 4294967294], [(0, R, 1, L), (1, R, 2, L), (0, R, 2, L)]),
        "hub": _mk(list(range(1, 42)), [(0, R, i, L) for i in range(1, 40)] + [(i, R, 40, L) for i in range(1, 40)]),
        "circular with gray children at the root": _mk([1, 2, 3, 4, 5], [(0, R, 1, L), (1, R, 0, L), (0, L, 2, L),
                                                                         (2, R, 3, L), (3, R, 4, L), (4, R, 2, L)]),
    }
    for name, g in cases.items():
        want = O.decompose(g)
        assert gpu_texts(hip, g) == want, name
        from povu_amd.hip import F_SEQUENTIAL
        hip.upload(g)
        assert hip.decompose(flags=F_SEQUENTIAL).texts() == want, name + " (sequential)"


def test_invalid_operands_are_rejected_on_the_host(hip):
    g = _mk([1, 2, 3], [(0, W.R, 1, W.L)])
    g.v2[0] = 9  # unknown vertex: would index out of bounds on the device
    with pytest.raises(RuntimeError, match="unknown vertex"):
        hip.upload(g)
    g = _mk([1, 2, 3], [(0, W.R, 1, W.L)])
    with pytest.raises(RuntimeError, match="bad tip"):
        hip.upload(g, tips=np.array([0, 5, 0], dtype=np.uint8))


# ---- size-independent properties at BASELINE sizes (no oracle involved)
def _forest_arrays(hip, g):
    hip.upload(g)
    f = hip.decompose()
    return [f.tree(i) for i in range(len(f))]


def test_full_size_structural_properties_and_determinism(hip):
    k = 333333
    g = W.chain_of_bubbles(k)
    t1 = _forest_arrays(hip, g)[0]
    t2 = _forest_arrays(hip, g)[0]
    for a, b in ((t1.a_id, t2.a_id), (t1.z_id, t2.z_id), (t1.parent, t2.parent), (t1.a_or, t2.a_or), (t1.z_or, t2.z_or)):
        assert np.array_equal(a, b)  # same input, same bits
    n = len(t1.parent)
    assert n == k + 1 and t1.parent[0] == 0xFFFFFFFF
    assert np.all(t1.parent[1:] < np.arange(1, n))          # a PVST parent is emitted before its children
    assert np.all(t1.parent[1:] == 0)                        # K sibling flubbles under the dummy root
    a = 3 * np.arange(k, dtype=np.int64) + 1
    assert set(zip(t1.a_id[1:].tolist(), t1.z_id[1:].tolist())) == set(zip(a.tolist(), (a + 3).tolist()))
    assert not t1.a_or[1:].any() and not t1.z_or[1:].any()
    # checksum of checksums: same value from the all-sequential kernels on a 1/50 slice of the same shape
    from povu_amd.hip import F_SEQUENTIAL
    small = W.chain_of_bubbles(k // 50)
    hip.upload(small)
    assert hip.decompose().texts() == hip.decompose(flags=F_SEQUENTIAL).texts()


def test_deep_nest_properties(hip):
    d, towers = 1000, 400
    t = _forest_arrays(hip, W.nested_towers(d, towers))[0]
    n = len(t.parent)
    assert n == d * towers + 1
    depth = np.zeros(n, dtype=np.int64)
    for i in range(1, n):
        depth[i] = depth[t.parent[i]] + 1
    assert depth.max() == d and np.count_nonzero(depth == 1) == towers
    # every flubble >a>b of a tower nests exactly one child except the innermost one
    kids = np.bincount(t.parent[1:], minlength=n)
    assert kids[0] == towers and set(np.unique(kids[1:]).tolist()) == {0, 1}
    assert np.count_nonzero(kids[1:] == 0) == towers


def test_hairpin_boundaries_parallel_and_sequential(hip):
    """--hairpins (flubbles.cpp:531-535, 621-656, 712-717): Boundary pairs in reporting order."""
    from povu_amd.hip import F_HAIRPINS, F_SEQUENTIAL
    n_with = 0
    for seed in range(60):
        n = 30 + 7 * seed
        g = W.random_bidirected(n, int(n * (1.0 + 0.04 * (seed % 9))), 31337 + seed)
        want = {}
        c = 0
        while True:
            d = dump_component(g, c)
            if d is None:
                break
            if len(d["gid"]):
                want[c + 1] = d["bry"].tolist()
            c += 1
        n_with += sum(1 for v in want.values() if v)
        hip.upload(g)
        for flags in (F_HAIRPINS, F_HAIRPINS | F_SEQUENTIAL):
            f = hip.decompose(flags=flags)
            got = {}
            for i in range(len(f)):
                t = f.tree(i)
                got[t.component_id] = t.hairpins.tolist()
            assert got == want, (seed, flags)
        assert hip.decompose(flags=F_HAIRPINS).texts() == O.decompose(g)
    assert n_with >= 10  # the sample does exercise hairpins


def test_cli_info_and_prune(tmp_path):
    """`povu info` / `povu prune` reuse row B: component order, local vertex / link order, tips."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    povu = os.path.join(root, "povu_amd", "bin", "povu")
    g = W.random_bidirected(90, 120, 4711)
    gfa = tmp_path / "g.gfa"
    gfa.write_text(g.to_gfa())
    out = tmp_path / "pruned"
    out.mkdir()
    comps = []
    c = 0
    while True:
        d = dump_component(g, c)
        if d is None:
            break
        comps.append(d)
        c += 1
    # loader tips of the whole graph
    deg = np.zeros((g.n_vtx, 2), dtype=np.int64)
    for a, sa, b, sb in zip(g.v1.tolist(), g.s1.tolist(), g.v2.tolist(), g.s2.tolist()):
        deg[a, sa] += 1
        if not (a == b and sa == sb):
            deg[b, sb] += 1
    tip = {int(g.vid[v]): ("+" if deg[v, 0] == 0 else "-") for v in range(g.n_vtx) if deg[v, 0] == 0 or deg[v, 1] == 0}
    r = subprocess.run([povu, "info", "-i", str(gfa), "-t"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stderr.startswith(f"[povu::main::do_info] Component count {len(comps)}\n")
    exp = ""
    for d in comps:
        ids = sorted(int(g.vid[v]) for v in d["gidx"].tolist())
        tl = [f"{i}{tip[i]}" for i in ids if i in tip]
        exp += f"Bidirected Graph: \n\tvertex count: {len(ids)}\n\tedge count: {len(d['ev1'])}\n\tTip count {len(tl)}\n"
        exp += "\t" + ", ".join(tl) + "\n"
    assert r.stdout == exp
    r = subprocess.run([povu, "prune", "-i", str(gfa), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for k, d in enumerate(comps):
        ids = [int(g.vid[v]) for v in d["gidx"].tolist()]
        text = "H\tVN:Z:1.0\n" + "".join(f"S\t{i}\tA\n" for i in ids)
        for a, sa, b, sb in zip(d["ev1"].tolist(), d["es1"].tolist(), d["ev2"].tolist(), d["es2"].tolist()):
            ea, eb = ("+", "+") if a == b else ("+" if sa == 1 else "-", "+" if sb == 0 else "-")
            text += f"L\t{ids[a]}\t{ea}\t{ids[b]}\t{eb}\t0M\n"
        assert (out / f"component_{k + 1}.gfa").read_text() == text, k
    assert len(list(out.glob("*.gfa"))) == len(comps)


def test_independent_contexts_run_concurrently():
    """povu-rs marks handles Send and uses independent ones from several threads (builder_tests.rs:106-137):
    two contexts on the same GPU decomposing different graphs at the same time."""
    import threading
    graphs = [W.hprc_shaped([3000 + 500 * i, 700], seed=50 + i, tiny=10) for i in range(4)]
    want = [O.decompose(g) for g in graphs]
    got = [None] * len(graphs)
    errs = []

    def work(k):
        try:
            d = HipDecomposer(0)
            for rep in range(3):
                d.upload(graphs[k])
                got[k] = d.decompose().texts()
                assert got[k] == want[k]
            d.close()
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(len(graphs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert got == want


def test_hub_with_many_parallel_links(hip):
    """A segment with thousands of links, many of them repeated: the duplicate-link rule of process_edge
    (spanning_tree.cpp:379-386) goes through the sort-based flags instead of the per-side look-back."""
    rng = np.random.default_rng(3)
    n = 1500
    spokes = rng.integers(1, n, size=6000)
    links = [(0, W.R, int(v), W.L) for v in spokes] + [(int(v), W.R, int(v) % (n - 1) + 1, W.L) for v in spokes[:2000]]
    links += [(int(v), W.R, 0, W.L) for v in spokes[:500]]
    g = _mk(list(range(1, n + 1)), links)
    want = O.decompose(g)
    assert gpu_texts(hip, g) == want
    assert np.array_equal(hip.debug_edge_ids(0), dump_component(g, 0)["pe_id"])
    from povu_amd.hip import F_SEQ_TREE, F_BIG_CLASS_DFS
    hip.upload(g)
    # (the hub side has thousands of candidates inside its class: the wave walk's overflow lists and 64-candidate windows)
    assert hip.decompose(flags=F_BIG_CLASS_DFS).texts() == want
    assert hip.decompose(flags=F_SEQ_TREE).texts() == want
    with pytest.raises(RuntimeError):
        hip.debug_edge_ids(0)


@pytest.mark.parametrize("fan", [2, 3, 17, 63, 64, 65, 66, 200])
def test_side_degrees_around_the_csr_builders_switch(hip, fan):
    """The adjacency at upload is filled through atomic cursors and sorted per side while no side has more than 64 links
    (k_slot_fill / k_side_sort), and by the stable radix sort of (side, link) pairs beyond: the same graph either way.  A
    fan of `fan` links on one side of a segment -- links of that side given in DESCENDING order of their far ends, some
    twice -- with bubbles behind the fan."""
    n = fan + 4
    links = [(0, W.R, k, W.L) for k in range(fan, 0, -1)]
    links += [(0, W.R, k, W.L) for k in range(1, fan, 7)]  # repeated links: same sides, later link index
    links += [(k, W.R, fan + 1, W.L) for k in range(1, fan + 1)]
    links += [(fan + 1, W.R, fan + 2, W.L), (fan + 1, W.R, fan + 3, W.L), (fan + 2, W.R, fan + 3, W.L)]
    g = _mk(list(range(1, n + 1)), links)
    want = O.decompose(g)
    assert gpu_texts(hip, g) == want
    assert np.array_equal(hip.debug_edge_ids(0), dump_component(g, 0)["pe_id"])


def _read_sidecar(path):
    with open(path) as fh:
        return [json.loads(l) for l in fh if l.strip()]


def _harness_checks(frame):
    """validate_flubble_debug_export, tests/lean4_conformance/src/main.rs:1288-1403 (per frame)."""
    assert frame["schema"] == "povu.flubble-debug.frame.v1"
    assert "tree_vertex_count" in frame and "tree_edge_count" in frame
    st, tb = frame["stack_entries"], frame["next_seen_table"]
    assert len(st) == len(tb)
    for e in st:
        for f in ["order", "tree_edge_index", "tree_edge_id", "boundary_vertex_id", "orientation", "provenance", "color",
                  "class_id", "parent_tree_vertex", "child_tree_vertex", "next_seen", "expected_next_seen",
                  "next_seen_in_range", "next_seen_same_class", "diagnostic"]:
            assert f in e
        assert e["diagnostic"] == "ok" and e["next_seen_in_range"] is True and e["next_seen_same_class"] is True
    for r in tb:
        for f in ["stack_order", "class_id", "next_seen", "expected_next_seen", "diagnostic"]:
            assert f in r
        assert r["diagnostic"] == "ok"


def test_structure_export_sidecar_and_gfa2vcf_glue(tmp_path, golden_dir):
    """--structure-export: the flubble debug sidecar (flubbles.cpp:108-231) frame by frame against the oracle's stack,
    the conformance harness's own checks, its cycle-class oracles by tree_edge_id (main.rs:1594-1622), and the
    gfa2vcf glue (gfa2vcf.cpp:18-87) handing the forest to the `call` of an external povu binary."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    povu = os.path.join(root, "povu_amd", "bin", "povu")
    r0 = W.random_bidirected(160, 230, 77)
    r1 = W.hprc_shaped([120, 60], seed=5, tiny=6)
    off = r0.n_vtx
    g = W._mk(np.concatenate([r0.vid, r1.vid + 1000]), np.concatenate([r0.v1, r1.v1 + off]), np.concatenate([r0.s1, r1.s1]),
              np.concatenate([r0.v2, r1.v2 + off]), np.concatenate([r0.s2, r1.s2]))
    gfa = tmp_path / "g.gfa"
    gfa.write_text(g.to_gfa())
    out = tmp_path / "out"
    out.mkdir()
    sx = tmp_path / "structure.json"
    side = str(sx) + ".flubble-debug.jsonl"
    with open(side, "w") as fh:
        fh.write("stale\n")  # reset_debug_sidecar removes an older file
    r = subprocess.run([povu, "decompose", "-i", str(gfa), "-o", str(out), "--structure-export", str(sx)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = O.decompose(g)
    assert {int(p.name[:-5]): p.read_text() for p in out.glob("*.pvst")} == want
    frames = _read_sidecar(side)
    assert len(frames) == len(want)
    def canon(x):
        m = {}
        return [m.setdefault(v, len(m)) for v in x]
    for frame, cid in zip(frames, sorted(want)):
        _harness_checks(frame)
        d = dump_component(g, cid - 1)
        assert frame["tree_vertex_count"] == len(d["gid"]) and frame["tree_edge_count"] == len(d["gid"]) - 1
        st = frame["stack_entries"]
        assert [e["order"] for e in st] == list(range(len(d["s_id"])))
        assert [e["tree_edge_index"] for e in st] == d["s_st_idx"].tolist()
        assert [e["tree_edge_id"] for e in st] == d["s_edge_id"].tolist()
        assert [e["boundary_vertex_id"] for e in st] == d["s_id"].tolist()
        assert [e["orientation"] for e in st] == [">" if o == 0 else "<" for o in d["s_orient"].tolist()]
        assert all(e["color"] == "black" and e["provenance"] == "real" for e in st)
        assert [e["child_tree_vertex"] for e in st] == (d["s_st_idx"] + 1).tolist()
        assert [e["parent_tree_vertex"] for e in st] == d["par"][d["s_st_idx"] + 1].tolist()
        assert [e["next_seen"] for e in st] == d["next_seen"].tolist()
        assert [e["expected_next_seen"] for e in st] == d["next_seen"].tolist()
        assert canon([e["class_id"] for e in st]) == canon(d["s_cls"].tolist())
        assert [(x["stack_order"], x["class_id"], x["next_seen"]) for x in frame["next_seen_table"]] == \
            [(e["order"], e["class_id"], e["next_seen"]) for e in st]
    # the harness's cycle-class oracles: groups of tree_edge_ids that must share a class, and only those
    vec = json.load(open(os.path.join(golden_dir, "reference_vectors.json")))["fixtures"]
    n_checked = 0
    for name, fx in vec.items():
        groups = fx.get("cycle_classes_by_tree_edge_id")
        if not groups:
            continue
        o2 = tmp_path / ("o_" + name)
        o2.mkdir()
        s2 = tmp_path / (name + ".json")
        r = subprocess.run([povu, "decompose", "-i", os.path.join(golden_dir, "gfa", fx["gfa"]), "-o", str(o2),
                            "--structure-export=" + str(s2)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        fr = _read_sidecar(str(s2) + ".flubble-debug.jsonl")
        assert len(fr) == 1
        _harness_checks(fr[0])
        st = fr[0]["stack_entries"]
        group_of = {e: k for k, grp in enumerate(groups) for e in grp}
        for e in st:
            assert e["tree_edge_id"] in group_of, (name, e)
        for a in st:
            for b in st:
                assert (a["class_id"] == b["class_id"]) == (group_of[a["tree_edge_id"]] == group_of[b["tree_edge_id"]]), name
        n_checked += 1
    assert n_checked == 6
    # gfa2vcf: without a `call` provider it refuses; with one, the child sees the forest and the pass-through options
    env = {k: v for k, v in os.environ.items() if k != "POVU_CALL_EXE"}
    r = subprocess.run([povu, "gfa2vcf", "-i", str(gfa), "-p", "HG"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "POVU_CALL_EXE" in r.stderr
    fake = tmp_path / "fake_povu.sh"
    log = tmp_path / "call.log"
    fake.write_text("#!/bin/bash\necho \"$@\" > %s\nwhile [ $# -gt 0 ]; do if [ \"$1\" = -f ]; then ls \"$2\" | sort -n >> %s; fi; shift; done\n"
                    "echo '##fileformat=VCFv4.2'\nexit 0\n" % (log, log))
    fake.chmod(0o755)
    os.remove(side)
    r = subprocess.run([povu, "-t", "2", "gfa2vcf", "-i", str(gfa), "--structure-export", str(sx), "--stdout", "-p", "HG", "-c", "50"],
                       capture_output=True, text=True, env=dict(env, POVU_CALL_EXE=str(fake)))
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("##fileformat=VCF")
    lines = log.read_text().splitlines()
    args = lines[0].split()
    assert args[:3] == ["-t", "2", "call"] and args[3:5] == ["-i", str(gfa)] and args[5] == "-f"
    assert args[7:] == ["--structure-export", str(sx), "--stdout", "-p", "HG", "-c", "50"]
    assert lines[1:] == ["%d.pvst" % c for c in sorted(want)]
    assert not os.path.exists(args[6])  # the temporary forest is removed afterwards
    assert len(_read_sidecar(side)) == len(want)
    # a failing `call` fails gfa2vcf
    fake.write_text("#!/bin/bash\nexit 7\n")
    r = subprocess.run([povu, "gfa2vcf", "-i", str(gfa)], capture_output=True, text=True, env=dict(env, POVU_CALL_EXE=str(fake)))
    assert r.returncode == 7


def test_reader_round_trip_on_gpu_output(hip):
    """SURVEY 8f item 2 on the product path: the PVST text the GPU pass writes goes back through the reader
    (read_pvst, src/mto/from_pvst.cpp:162-302) and comp_heights (pvst.hpp:807-836) -- what `povu call` does first
    (call.cpp:36-53) -- and must give the arrays the forest holds."""
    import ctypes as C
    from test_cabi_and_host import _Doc
    from povu_amd import hip as H
    hl = H.load_lib()
    hl.povu_pvst_parse.restype = C.POINTER(_Doc)
    hl.povu_pvst_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    hl.povu_pvst_doc_free.argtypes = [C.POINTER(_Doc)]
    g = W.hprc_shaped([3000, 800], seed=21, tiny=5)
    g2 = W.nested_towers(40, 6)
    for links in (g, g2):
        hip.upload(links)
        f = hip.decompose()
        texts = f.texts()
        assert texts == O.decompose(links)
        arrays = {}
        for i in range(len(f)):
            pt = f.tree(i)
            arrays[pt.component_id] = dict(parent=pt.parent, a_id=pt.a_id, z_id=pt.z_id, a_or=pt.a_or, z_or=pt.z_or)
        for cid, text in texts.items():
            raw = text.encode()
            err = C.create_string_buffer(256)
            d = hl.povu_pvst_parse(raw, len(raw), err, 256)
            assert d, err.value
            doc = d.contents
            t = arrays[cid]
            n = doc.n
            assert n == len(t["parent"])
            assert np.array_equal(np.ctypeslib.as_array(doc.parent, (n,))[1:], t["parent"][1:])
            assert np.array_equal(np.ctypeslib.as_array(doc.a_id, (n,))[1:], t["a_id"][1:])
            assert np.array_equal(np.ctypeslib.as_array(doc.z_id, (n,))[1:], t["z_id"][1:])
            assert np.array_equal(np.ctypeslib.as_array(doc.a_or, (n,))[1:], t["a_or"][1:])
            assert np.array_equal(np.ctypeslib.as_array(doc.z_or, (n,))[1:], t["z_or"][1:])
            assert [doc.type[i] for i in range(n)] == [b"D"] + [b"F"] * (n - 1)
            depth = np.zeros(n, dtype=np.int64)
            par = t["parent"].astype(np.int64)
            for i in range(1, n):  # parents precede children (emission order, flubbles.cpp:345-361)
                depth[i] = depth[par[i]] + 1
            assert np.array_equal(np.ctypeslib.as_array(doc.height, (n,)), depth)
            hl.povu_pvst_doc_free(d)


def test_black_edge_only_class_pass(hip, golden_dir):
    """The class pass numbers the black tree edges only (half the vertices to sort) unless the literal hi_2 rule
    (flubbles.cpp:566-574) capped somewhere else than at the second-highest reach, or hairpins are reported; both
    passes must give the reference's result."""
    from povu_amd.hip import F_ALL_VERTEX_CLASSES, F_HAIRPINS
    fast = 0
    for g in [W.chain_of_bubbles(4000), W.hprc_shaped([5000, 700], seed=3, tiny=11), W.nested_towers(30, 5),
              W.hprc_tangled(3000, seed=4, tangle_every=700, max_tangle=400)]:
        want = O.decompose(g)
        hip.upload(g)
        assert hip.decompose().texts() == want
        fast += hip.last_black_only_classes()
        assert hip.decompose(flags=F_ALL_VERTEX_CLASSES).texts() == want
        assert not hip.last_black_only_classes()
        assert hip.decompose(flags=F_HAIRPINS).texts() == want
        assert not hip.last_black_only_classes()
    assert fast >= 3
    # the hand-derived vector where hi_2 is NOT the second-highest reach: the fast pass must stand down
    g = _load_gfa_links(os.path.join(golden_dir, "gfa", "hi2_literal_rule.gfa"))
    hip.upload(g)
    got = hip.decompose().texts()
    assert not hip.last_black_only_classes()
    assert got[1] == open(os.path.join(golden_dir, "pvst", "hi2_literal_rule.pvst")).read()
    used = [0, 0]
    for seed in range(40):
        n = 25 + 3 * seed
        g = W.random_bidirected(n, int(n * (1.2 + 0.1 * (seed % 8))), 4242 + seed, connected=True)
        want = O.decompose(g)
        hip.upload(g)
        assert hip.decompose().texts() == want, seed
        used[int(hip.last_black_only_classes())] += 1
        assert hip.decompose(flags=F_ALL_VERTEX_CLASSES).texts() == want, seed
    assert used[0] > 0 and used[1] > 0, used


def test_prewarm_reserves_without_changing_results():
    """povu_hip_prewarm on a fresh context (what the CLI does while it parses): same PVSTs afterwards, a second call on a
    context that holds a graph is a no-op, and a graph larger than the one announced still works (the arenas grow)."""
    d = HipDecomposer(0)
    g = W.hprc_shaped([3000, 500], seed=5, tiny=20)
    d.prewarm(g.n_vtx, g.n_links)
    d.upload(g)
    assert d.decompose().texts() == O.decompose(g)
    d.prewarm(10, 10)  # holds a graph: nothing happens
    assert d.decompose().texts() == O.decompose(g)
    big = W.hprc_shaped([20000, 900], seed=6, tiny=50)
    d.upload(big)
    assert d.decompose().texts() == O.decompose(big)
    d.close()
    with pytest.raises(RuntimeError):
        e = HipDecomposer(0)
        try:
            e.prewarm(0, 0)  # no vertices: refused like an upload
        finally:
            e.close()
