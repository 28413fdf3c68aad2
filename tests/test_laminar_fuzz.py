"""CPU hunt for a crossing pair of candidate-stack intervals (row G).

add_flubbles (flubbles.cpp:295-367) is a stack machine; the HIP path evaluates it in closed form, which is exact when
the intervals (previous occurrence of a class, this occurrence) over the candidate stack are LAMINAR: no two of them
cross.  DESIGN.md section 4 ("Row G") proves that for exact cycle-equivalence classes; the GPU kernels only run their
range-min check when the literal hi_2 rule (flubbles.cpp:555-574) capped differently from the second-highest reach,
i.e. when the classes might not be the exact ones.  This test looks for a counter-example on the oracle's candidate
stacks: random multigraphs (sparse to dense, with self loops and parallel links), chains of bubbles, nested towers,
HPRC-shaped graphs with tangles -- shapes that do trigger the literal rule.  No GPU minutes."""
import numpy as np
import pytest

import oracle_lib as O
from povu_amd import workloads as W
from test_oracle import dump_component


def crossing_pairs(s_cls):
    """Number of entries i whose interval (prev[i], i) is crossed by the interval of an entry inside it (the test the
    GPU's k_laminar_walk makes: some k in (prev[i], i) has prev[k] < prev[i])."""
    n = len(s_cls)
    last = {}
    prev = np.full(n, -1, dtype=np.int64)
    for i, c in enumerate(s_cls.tolist()):
        prev[i] = last.get(c, -1)
        last[c] = i
    bad = 0
    # stack of open intervals, innermost on top: entry i with prev p >= 0 must find p as the start of ... simply check
    # with a running structure: for every i with p >= 0, min over prev[p+1 .. i-1] of the entries that HAVE a prev
    for i in range(n):
        p = prev[i]
        if p < 0 or p + 1 >= i:
            continue
        inner = prev[p + 1:i]
        inner = inner[inner >= 0]
        if inner.size and inner.min() < p:
            bad += 1
    return bad


def graphs():
    k = 0
    for seed in range(120):
        n = 30 + 11 * (seed % 17)
        dens = 1.0 + 0.2 * (seed % 9)
        yield f"random n={n} dens={dens:.1f} seed={seed}", W.random_bidirected(n, int(n * dens), 4200 + seed,
                                                                              connected=(seed % 2 == 0), self_loops=(seed % 3 == 0))
    for seed in range(12):
        yield f"hprc {seed}", W.hprc_shaped([1500 + 300 * seed, 60], seed=900 + seed, tiny=4)
    for seed in range(8):
        yield f"tangled {seed}", W.hprc_tangled(2500, seed=seed, tangle_every=500, max_tangle=300)
    yield "chain", W.chain_of_bubbles(400)
    yield "towers", W.nested_towers(40, 6)


def test_candidate_stack_intervals_are_laminar_on_every_fuzzed_graph():
    comps = entries = 0
    for name, g in graphs():
        c = 0
        while True:
            d = dump_component(g, c)
            if d is None:
                break
            c += 1
            if len(d["s_cls"]) == 0:
                continue
            comps += 1
            entries += len(d["s_cls"])
            assert crossing_pairs(d["s_cls"]) == 0, f"crossing candidate-stack intervals in {name}, component {c}"
    assert comps > 150 and entries > 30000  # (the hunt did look at something)


def test_the_literal_hi2_rule_can_cross_intervals(golden_dir):
    """A graph the GPU fuzz found (kept as data: an HPRC-shaped backbone of 8 299 segments with random tangles,
    `workloads.hprc_tangled`): the literal hi_2 rule caps below the second-highest reach there, the reference's classes are
    not the exact cycle-equivalence classes, and its OWN candidate stack has one crossing pair of intervals.  This is the
    case the HIP path's range-min check exists for (it runs exactly when `k_capping` saw the rule deviate) and the
    component is then redone by the sequential kernels: tests/test_gpu_parity.py::test_non_laminar_stack_takes_the_redo."""
    import os
    d = np.load(os.path.join(golden_dir, "literal_hi2_crossing_stack.npz"))
    g = W._mk(d["vid"], d["v1"], d["s1"], d["v2"], d["s2"])
    dd = dump_component(g, 0)
    assert len(dd["s_cls"]) == g.n_vtx and crossing_pairs(dd["s_cls"]) == 1


def test_crossing_detector_sees_a_crossing():
    assert crossing_pairs(np.array([1, 2, 1, 2])) == 1
    assert crossing_pairs(np.array([1, 2, 2, 1, 3, 3])) == 0
