"""GPU parity of the two relabelling passes of `povu decompose -s` (find_tiny, tiny.cpp:100-129; find_parallel,
parallel.cpp:263-287): POVU_HIP_F_LEAF_SUBFLUBBLES through the C ABI against the oracle's literal restatement -- PVST text
with T / O lines, ai / zi (compute_ai_zi, flubbles.cpp:264-290) and the line letter of every PVST vertex.  PARITY UNPINNED:
the reference holds no T or O line anywhere; tests/test_leaf_subflubbles_oracle.py has the one hand-derived vector and the
rule coverage of these inputs."""
import collections
import glob
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from povu_amd import HipDecomposer, workloads as W
from povu_amd.hip import F_LEAF_SUBFLUBBLES
from test_leaf_subflubbles_oracle import ACCIDENT_SEEDS, components
from test_oracle import _load_gfa_links

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POVU = os.path.join(ROOT, "povu_amd", "bin", "povu")


@pytest.fixture(scope="module")
def hip():
    d = HipDecomposer(0)
    yield d
    d.close()


def check(hip, g, arrays=True):
    """Text of every PVST and (arrays) ai / zi / letters of every tree against the oracle; returns the letter counts."""
    hip.upload(g)
    f = hip.decompose(flags=F_LEAF_SUBFLUBBLES)
    assert f.texts() == O.decompose(g, leaf=True)
    seen = collections.Counter()
    if arrays:
        want = list(components(g))
        assert len(want) == len(f)
        for i, d in enumerate(want):
            ai, zi, fam = f.sub(i)
            assert bytes(fam) == bytes(d["p_fam"])
            assert ai.tolist() == d["p_ai"].tolist() and zi.tolist() == d["p_zi"].tolist()
            seen.update(bytes(fam).decode())
    # the same graph without the flag: F lines only, and the forest says it carries no labels
    f0 = hip.decompose()
    assert f0.texts() == O.decompose(g)
    if len(f0):
        with pytest.raises(RuntimeError):
            f0.sub(0)
    return seen


@pytest.mark.parametrize("seed", range(24))
def test_bubble_zoo(hip, seed):
    seen = check(hip, W.bubble_zoo(30, 6, seed))
    assert seen["T"] and seen["F"]


@pytest.mark.parametrize("seed", ACCIDENT_SEEDS)
def test_back_edge_index_equal_to_ai(hip, seed):
    """tiny.cpp:52-56 compares back-edge INDICES with the vertex idx ai; on these graphs the comparison decides a label
    (oracle rule counter idx_ord), so the kernels' index ranges (back edges created before a vertex was discovered) are
    what is being checked."""
    check(hip, W.bubble_zoo(40, 5, seed))


@pytest.mark.parametrize("seed", range(4))
def test_long_backbones(hip, seed):
    check(hip, W.bubble_zoo(3, 60, 100 + seed))


@pytest.mark.parametrize("seed", range(16))
def test_random_graphs(hip, seed):
    n = 30 + 11 * (seed % 17)
    check(hip, W.random_bidirected(n, int(n * (1.0 + 0.2 * (seed % 9))), 4200 + seed, connected=(seed % 2 == 0),
                                   self_loops=(seed % 3 == 0)))


def test_golden_fixture_graphs(hip, golden_dir):
    for p in sorted(glob.glob(os.path.join(golden_dir, "gfa", "*.gfa"))):
        check(hip, _load_gfa_links(p))


def test_shapes_of_the_baseline_configs(hip):
    assert check(hip, W.chain_of_bubbles(400))["O"] == 400
    check(hip, W.nested_towers(40, 6))
    check(hip, W.hprc_shaped([1500, 60, 700], seed=900, tiny=40))


def test_config2_at_full_size(hip):
    """BASELINE config 2 (10^6 segments): every flubble of the chain is a leaf, all of them 'parallel' by in_trunk."""
    g = W.chain_of_bubbles(333333)
    hip.upload(g)
    f = hip.decompose(flags=F_LEAF_SUBFLUBBLES)
    want = O.decompose(g, leaf=True)
    assert f.texts() == want and want[1].count("\nO\t") == 333333


def test_hprc_shaped_chromosome(hip):
    g = W.hprc_shaped([400000, 9000], seed=11, tiny=100)
    hip.upload(g)
    f = hip.decompose(flags=F_LEAF_SUBFLUBBLES)
    assert f.texts() == O.decompose(g, leaf=True)


def test_deep_nests(hip):
    """BASELINE config 5's shape at a depth the reference's own bracket table still fits in memory (it materialises, per
    back edge, every tree vertex the edge spans: tree_utils.cpp:169-216 -- quadratic in the depth; the HIP path's prefix
    sums are not)."""
    check(hip, W.nested_towers(300, 40), arrays=False)


def test_config4_whole_genome_full_size(hip):
    """BASELINE config 4 at full size with the two passes: md5 of every PVST text against the oracle (its components on
    the host's cores), and what the extra stage costs."""
    import hashlib
    g = W.hprc_whole_genome(1e8)
    cores = min(32, len(os.sched_getaffinity(0)))
    want = {k: hashlib.md5(v.encode()).hexdigest() for k, v in O.decompose(g, threads=cores, lpt=True, leaf=True).items()}
    hip.upload(g)
    f = hip.decompose(flags=F_LEAF_SUBFLUBBLES)
    assert hip.seq_redo_count() == 0
    ms = {t["name"]: t["ms"] for t in hip.stage_times()}
    got = {k: hashlib.md5(v.encode()).hexdigest() for k, v in f.texts().items()}
    assert got == want and len(want) == 2024
    letters = collections.Counter()
    for i in range(0, len(f), 97):
        letters.update(bytes(f.sub(i)[2]).decode())
    print(f"leaf_subflubbles stage on config 4 at full size: {ms.get('leaf_subflubbles', -1):.2f} ms of {ms.get('total', -1):.2f}; "
          f"letters of a sample of trees {dict(letters)}")
    assert letters["T"] and letters["F"]
    hip.upload(W.chain_of_bubbles(3))  # (frees the big graph for the tests that follow)


def test_components_that_went_through_the_redo(hip, golden_dir):
    """A candidate stack with a crossing pair (tests/golden/literal_hi2_crossing_stack.npz): resolved in place by the parallel
    stage since round 4 (no redo), labels as the oracle's; the redo path itself (seq_pvst leaves ai / zi in the one-lane
    kernels' layout, the passes label the PVST in place) is covered by the forced modes below."""
    d = np.load(os.path.join(golden_dir, "literal_hi2_crossing_stack.npz"))
    g = W._mk(d["vid"], d["v1"], d["s1"], d["v2"], d["s2"])
    hip.upload(g)
    f = hip.decompose(flags=F_LEAF_SUBFLUBBLES)
    assert hip.seq_redo_count() == 0 and hip.last_crossings() == (1, 1)
    assert f.texts() == O.decompose(g, leaf=True)
    want = list(components(g))
    ai, zi, fam = f.sub(0)
    assert bytes(fam) == bytes(want[0]["p_fam"]) and ai.tolist() == want[0]["p_ai"].tolist() and zi.tolist() == want[0]["p_zi"].tolist()


@pytest.mark.parametrize("seed", range(6))
def test_forced_and_partial_redo(hip, seed):
    """Every component / every second component through the redo of add_flubbles: the labels of the redone PVSTs come from
    the per-component layout, the others from the dense output -- same answers either way."""
    from povu_amd.hip import F_FORCE_REDO, F_REDO_ODD
    g = W.bubble_zoo(25, 6, 300 + seed)
    want_text = O.decompose(g, leaf=True)
    want = list(components(g))
    hip.upload(g)
    for fl in (F_FORCE_REDO, F_REDO_ODD):
        f = hip.decompose(flags=F_LEAF_SUBFLUBBLES | fl)
        assert hip.seq_redo_count() > 0
        assert f.texts() == want_text
        assert len(f) == len(want)
        for i, dd in enumerate(want):
            ai, zi, fam = f.sub(i)
            assert bytes(fam) == bytes(dd["p_fam"]) and ai.tolist() == dd["p_ai"].tolist() and zi.tolist() == dd["p_zi"].tolist()


def test_not_with_the_sequential_tree_modes(hip):
    from povu_amd.hip import F_FORCE_REDO, F_HAIRPINS, F_SEQ_TREE, F_SEQUENTIAL
    hip.upload(W.chain_of_bubbles(5))
    for fl in (F_SEQUENTIAL, F_SEQ_TREE):
        with pytest.raises(RuntimeError, match="parallel stages"):
            hip.decompose(flags=F_LEAF_SUBFLUBBLES | fl)
    with pytest.raises(RuntimeError, match="from scratch"):  # with hairpins the redo rebuilds the tree on one lane
        hip.decompose(flags=F_LEAF_SUBFLUBBLES | F_FORCE_REDO | F_HAIRPINS)
    f = hip.decompose(flags=F_LEAF_SUBFLUBBLES | F_HAIRPINS)  # (without a redo the two go together)
    assert f.texts() == O.decompose(W.chain_of_bubbles(5), leaf=True)


def test_labels_do_not_travel_in_the_wire_format(hip):
    hip.upload(W.chain_of_bubbles(5))
    f = hip.decompose(flags=F_LEAF_SUBFLUBBLES)
    with pytest.raises(RuntimeError):
        f.pack()


def test_cli_leaf_subflubbles(tmp_path):
    g = W.bubble_zoo(12, 6, 77)
    gfa = tmp_path / "zoo.gfa"
    gfa.write_text(g.to_gfa())
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([POVU, "decompose", "-i", str(gfa), "-o", str(out), "--leaf-subflubbles"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = O.decompose(g, leaf=True)
    got = {int(p.stem): p.read_text() for p in out.glob("*.pvst")}
    assert got == want and any("\nT\t" in t for t in got.values())
    # (-s itself = all five passes: tests/test_gpu_subflubbles.py)
