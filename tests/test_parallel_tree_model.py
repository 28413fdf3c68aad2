"""Design check (CPU): the bridge-decomposed formulation of the reference's lexicographic biedged
DFS (row C) reproduces pst::Tree::from_bd exactly.

Model (what the parallel tree kernels compute, DESIGN.md "Parallel spanning tree"):
  H = biedged graph on segment sides: black edge (S, S^1) per segment, one gray edge per local link.
  from_bd is the lexicographic DFS of H where every side scans [black edge, gray links in local-edge
  order].  Bridges of H split it into 2-edge-connected classes; the DFS can only enter a class
  through its unique bridge towards the root, and what it does inside a class does not depend on the
  rest of the graph.  So: (1) bridges + classes (any spanning tree), (2) per class: entry side and its
  DFS parent = the far end of that bridge, (3) an independent sequential DFS inside every class,
  (4) pre-order numbering of the union tree with children ordered by the parent's scan position,
  (5) back edges from the non-tree slots with the reference's de-duplication rules.
The oracle's tree / back-edge dump is the expected value."""
import sys

import numpy as np
import pytest

import oracle_lib as O
from povu_amd import workloads as W
from test_oracle import dump_component

sys.setrecursionlimit(100000)


def local_graph(d):
    """per-side adjacency (other side, local edge) in ascending local edge idx, from the oracle's
    component dump (local edge arrays)."""
    nv = len(d["gidx"])
    adj = [[] for _ in range(2 * nv)]
    for le, (a, sa, b, sb) in enumerate(zip(d["ev1"].tolist(), d["es1"].tolist(), d["ev2"].tolist(), d["es2"].tolist())):
        A, B = 2 * a + sa, 2 * b + sb
        adj[A].append((B, le))
        if B != A:
            adj[B].append((A, le))
    return nv, adj


def bridges_and_classes(n, edges):
    """edges: list of (u, w, eid) of H (multigraph).  Returns set of bridge eids and class labels."""
    g = [[] for _ in range(n)]
    for u, w, e in edges:
        g[u].append((w, e))
        g[w].append((u, e))
    disc = [-1] * n
    low = [0] * n
    bridges = set()
    t = 0
    for r in range(n):
        if disc[r] != -1:
            continue
        stack = [(r, -1, 0)]
        disc[r] = low[r] = t
        t += 1
        while stack:
            u, pe, i = stack.pop()
            if i < len(g[u]):
                stack.append((u, pe, i + 1))
                w, e = g[u][i]
                if e == pe:
                    continue
                if disc[w] == -1:
                    disc[w] = low[w] = t
                    t += 1
                    stack.append((w, e, 0))
                else:
                    low[u] = min(low[u], disc[w])
            else:
                if stack:
                    p = stack[-1][0]
                    low[p] = min(low[p], low[u])
                    if low[u] > disc[p]:
                        bridges.add(pe)
    lab = list(range(n))

    def find(x):
        while lab[x] != x:
            lab[x] = lab[lab[x]]
            x = lab[x]
        return x

    for u, w, e in edges:
        if e not in bridges:
            a, b = find(u), find(w)
            if a != b:
                lab[max(a, b)] = min(a, b)
    return bridges, [find(x) for x in range(n)]


def model_tree(d, tips):
    nv, adj = local_graph(d)
    n = 2 * nv
    has_tips = any(tips)
    vid = d["vid_local"]
    if has_tips:
        best = min((vid[v], v) for v in range(nv) if tips[v])[1]
        start = 2 * best + (0 if tips[best] == 1 else 1)
    else:
        start = 0
    # H edges: black ids 0..nv-1, gray ids nv+le
    edges = [(2 * v, 2 * v + 1, v) for v in range(nv)]
    ne = len(d["ev1"])
    for le in range(ne):
        a, b = 2 * int(d["ev1"][le]) + int(d["es1"][le]), 2 * int(d["ev2"][le]) + int(d["es2"][le])
        edges.append((a, b, nv + le))
    bridges, ecc = bridges_and_classes(n, edges)
    # scan list of a side: black first, then gray slots
    def scan(S):
        return [(S ^ 1, S >> 1, 0)] + [(o, nv + le, k + 1) for k, (o, le) in enumerate(adj[S])]
    # (2) entry + DFS parent of each class: walk any spanning tree from `start`
    seen = [False] * n
    par0 = [-1] * n
    pe0 = [-1] * n
    order = [start]
    seen[start] = True
    for u in order:
        for o, e, _ in scan(u):
            if not seen[o]:
                seen[o] = True
                par0[o] = u
                pe0[o] = e
                order.append(o)
    dpar = [-1] * n
    cslot = [0] * n
    entry = {ecc[start]: start}
    for u in order[1:]:
        if pe0[u] in bridges:
            entry[ecc[u]] = u
            dpar[u] = par0[u]
            cslot[u] = next(k for o, e, k in scan(par0[u]) if e == pe0[u])
    # (3) independent DFS inside each class
    for cls, s in entry.items():
        vis = {s}
        cur = {s: 0}
        u = s
        while True:
            lst = scan(u)
            k = cur[u]
            adv = False
            while k < len(lst):
                o, e, slot = lst[k]
                k += 1
                if ecc[o] == cls and o not in vis:
                    vis.add(o)
                    dpar[o] = u
                    cslot[o] = slot
                    cur[u] = k
                    cur[o] = 0
                    u = o
                    adv = True
                    break
            if adv:
                continue
            cur[u] = k
            if u == s:
                break
            u = dpar[u]
    # (4) pre-order with children ordered by the parent's scan position
    kids = [[] for _ in range(n)]
    for u in range(n):
        if dpar[u] >= 0:
            kids[dpar[u]].append((cslot[u], u))
    for k in kids:
        k.sort()
    base = 1 if has_tips else 0
    pre = [-1] * n
    stack = [start]
    cnt = base
    while stack:
        u = stack.pop()
        pre[u] = cnt
        cnt += 1
        for _, c in reversed(kids[u]):
            stack.append(c)
    N = n + base
    par = [0xFFFFFFFF] * N
    gid = [0xFFFFFFFF] * N
    typ = [2] * N
    black = [0] * N
    for S in range(n):
        t = pre[S]
        gid[t] = vid[S >> 1]
        typ[t] = S & 1
        if dpar[S] >= 0:
            par[t] = pre[dpar[S]]
            black[t] = 1 if dpar[S] == (S ^ 1) else 0
        else:
            par[t] = 0 if has_tips else 0xFFFFFFFF
    # (5) back edges
    bes = []
    for S in range(n):
        p = pre[S]
        if not adj[S]:
            if p == 0 or par[p] != 0:
                bes.append((p, -1, p, 0))
            continue
        seen_loop = False
        for k, (o, le) in enumerate(adj[S]):
            x = pre[o]
            if o == (S ^ 1):
                if dpar[S] == o and not seen_loop:
                    bes.append((p, k, p, x))
                seen_loop = True
                continue
            if x > p or o == dpar[S]:
                continue
            if any(adj[S][j][0] == o for j in range(k)):
                continue
            bes.append((p, k, p, x))
    return dict(par=par, gid=gid, typ=typ, black=black, bes=sorted(bes))


def check(g, tips=None):
    info_tips = None
    for comp in range(64):
        d = dump_component(g, comp, tips)
        if d is None:
            break
        if len(d["gid"]) == 0:
            continue
        gl = d["gidx"]
        d["vid_local"] = [int(g.vid[v]) for v in gl.tolist()]
        if tips is None:
            # infer tips the way the loader does, on the global graph
            deg = np.zeros((g.n_vtx, 2), dtype=np.int64)
            for a, sa, b, sb in zip(g.v1.tolist(), g.s1.tolist(), g.v2.tolist(), g.s2.tolist()):
                deg[a, sa] += 1
                if not (a == b and sa == sb):
                    deg[b, sb] += 1
            info_tips = [1 if deg[v, 0] == 0 else (2 if deg[v, 1] == 0 else 0) for v in gl.tolist()]
        else:
            info_tips = [int(tips[v]) for v in gl.tolist()]
        m = model_tree(d, info_tips)
        assert m["par"] == d["par"].tolist()
        assert m["gid"] == d["gid"].tolist()
        assert m["typ"] == d["typ"].tolist()
        assert m["black"] == d["pe_black"].tolist()
        # back edges of from_bd: same set, same order per source
        n0 = d["n_be0"]
        want = {}
        for s, t in zip(d["be_src"][:n0].tolist(), d["be_tgt"][:n0].tolist()):
            want.setdefault(s, []).append(t)
        got = {}
        for p, k, s, t in m["bes"]:
            got.setdefault(s, []).append(t)
        assert got == want


@pytest.mark.parametrize("seed", range(40))
def test_model_matches_from_bd_random(seed):
    n = 12 + 5 * seed
    check(W.random_bidirected(n, int(n * (1.1 + 0.08 * (seed % 9))), 4242 + seed))


def test_model_matches_from_bd_shapes():
    check(W.chain_of_bubbles(40))
    check(W.nested_towers(6, 3))
    check(W.hprc_shaped([120, 60], seed=5, tiny=6))


def test_model_without_tips():
    g = W.random_bidirected(30, 60, 77)
    check(g, tips=np.zeros(g.n_vtx, dtype=np.uint8))
