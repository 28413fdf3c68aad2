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


# ---- the closed form WITH crossings (what k_resolve_crossings + the walk kernels evaluate): model against the machine
def machine_pops(cls):
    """add_flubbles' stack machine (flubbles.cpp:316-365), the part that does not depend on the graph: U[i] = the class of
    entry i was on the auxiliary stack (it is popped through and the PVST parent climbs one level)."""
    s, in_s, U = [], set(), []
    for c in cls:
        if c in in_s:
            U.append(1)
            while s:
                t = s.pop()
                in_s.discard(t)
                if t == c:
                    break
        else:
            U.append(0)
        s.append(c)
        in_s.add(c)
    return np.array(U, dtype=np.int64)


def closed_form_pops(cls):
    """U[i] = has a previous occurrence p AND no entry k in (p, i) whose class was open at k AND reaches back beyond p
    (prev[k] < p): such a k popped through its own class and took the entry pushed at p with it.  `open` is evaluated for
    the FLAGGED entries only -- those whose interval holds an entry with a smaller prev (the range-min test of
    k_laminar_walk) --, in stack order, each looking at the flagged entries before it; an unflagged entry is open iff it
    has a previous occurrence.  Returns (U, number of flagged entries)."""
    n = len(cls)
    last, prev = {}, np.full(n, -1, dtype=np.int64)
    for i, c in enumerate(cls):
        prev[i] = last.get(c, -1)
        last[c] = i
    flagged = np.zeros(n, dtype=bool)
    for i in range(n):
        p = prev[i]
        if p >= 0 and p + 1 < i:
            inner = prev[p + 1:i]
            inner = inner[inner >= 0]
            flagged[i] = inner.size > 0 and inner.min() < p
    crossed = np.zeros(n, dtype=bool)
    for i in np.flatnonzero(flagged):  # ascending
        p = prev[i]
        for k in range(p + 1, i):
            if 0 <= prev[k] < p and not crossed[k]:  # (an entry with a previous occurrence is open unless it was crossed)
                crossed[i] = True
                break
    return ((prev >= 0) & ~crossed).astype(np.int64), int(flagged.sum())


def test_closed_form_with_crossings_equals_the_stack_machine():
    """Arbitrary label sequences (far more tangled than any candidate stack of a graph): the closed form with resolved
    crossings answers "was the class open" exactly as the machine does, so the U / D events -- all the PVST construction
    depends on -- are the machine's."""
    rng = np.random.default_rng(11)
    flagged_total = differ_from_naive = 0
    for t in range(3000):
        n = int(rng.integers(2, 60))
        k = int(rng.integers(1, max(2, n // 2 + 1)))
        cls = rng.integers(0, k, size=n).tolist()
        want = machine_pops(cls)
        got, nf = closed_form_pops(cls)
        assert np.array_equal(got, want), (cls, got.tolist(), want.tolist())
        flagged_total += nf
        naive, _seen = [], set()
        for c in cls:
            naive.append(1 if c in _seen else 0)
            _seen.add(c)
        differ_from_naive += int((np.array(naive) != want).any())
    assert flagged_total > 5000 and differ_from_naive > 1000  # (the sequences did cross, and crossing did matter)
    # the graph the GPU fuzz found: one crossing pair, one entry whose class is NOT open although it occurred before
    assert closed_form_pops([1, 2, 1, 2])[0].tolist() == [0, 0, 1, 0]
