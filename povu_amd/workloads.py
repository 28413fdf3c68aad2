"""Deterministic synthetic bidirected graphs for parity tests and bench.py.

The shapes are the ones SURVEY.md section 8(d) fixes for BASELINE.json's
configs.  A graph is returned as a :class:`Links` record: segment ids in
ascending order plus one row per GFA L-line, already translated to vertex
*indices* and sides the way the loader does (`+` on the source = right side,
`+` on the sink = left side; reference src/mto/from_gfa.cpp:223-243).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

L, R = 0, 1


@dataclass
class Links:
    vid: np.ndarray  # uint32 [V] ascending segment ids
    v1: np.ndarray   # uint32 [E] vertex idx
    s1: np.ndarray   # uint8  [E] side (0 = l, 1 = r)
    v2: np.ndarray   # uint32 [E]
    s2: np.ndarray   # uint8  [E]

    @property
    def n_vtx(self) -> int:
        return int(self.vid.shape[0])

    @property
    def n_links(self) -> int:
        return int(self.v1.shape[0])

    def to_gfa(self) -> str:
        """GFA text: S lines first (sequence `A`), then L lines, all `0M`."""
        out = ["H\tVN:Z:1.0"]
        out += [f"S\t{i}\tA" for i in self.vid.tolist()]
        vid = self.vid
        for a, sa, b, sb in zip(self.v1.tolist(), self.s1.tolist(), self.v2.tolist(), self.s2.tolist()):
            out.append(f"L\t{vid[a]}\t{'+' if sa == R else '-'}\t{vid[b]}\t{'+' if sb == L else '-'}\t0M")
        return "\n".join(out) + "\n"


def _mk(vid, v1, s1, v2, s2) -> Links:
    return Links(np.ascontiguousarray(vid, dtype=np.uint32), np.ascontiguousarray(v1, dtype=np.uint32),
                 np.ascontiguousarray(s1, dtype=np.uint8), np.ascontiguousarray(v2, dtype=np.uint32),
                 np.ascontiguousarray(s2, dtype=np.uint8))


def from_plus_links(vid, src_idx, dst_idx) -> Links:
    """All links `a + b +`."""
    e = len(src_idx)
    return _mk(vid, src_idx, np.full(e, R), dst_idx, np.full(e, L))


def chain_of_bubbles(k: int) -> Links:
    """BASELINE config 2: K units, ids 1..3K+1; unit i has a=3i+1, x=a+1,
    y=a+2, b=a+3 and links a>x a>y x>y x>b y>b a>b.  K=333333 -> 1 000 000 segments,
    1 999 998 links, one component, K flubbles."""
    vid = np.arange(1, 3 * k + 2, dtype=np.uint32)
    a = 3 * np.arange(k, dtype=np.int64)  # vertex idx of `a`
    src = np.stack([a, a, a + 1, a + 1, a + 2, a], axis=1).reshape(-1)
    dst = np.stack([a + 1, a + 2, a + 2, a + 3, a + 3, a + 3], axis=1).reshape(-1)
    return from_plus_links(vid, src, dst)


def nested_towers(depth: int, towers: int) -> Links:
    """BASELINE config 5 shape: per tower a_0..a_{d-1}, c, b_{d-1}..b_0 (ids in
    that order), links a_i>a_{i+1} and b_{i+1}>b_i interleaved, a_{d-1}>c, c>b_{d-1}, bypasses
    a_i>b_i, towers chained b_0(t)>a_0(t+1)."""
    d = depth
    per = 2 * d + 1
    vid = np.arange(1, per * towers + 1, dtype=np.uint32)
    src, dst = [], []
    i = np.arange(d - 1, dtype=np.int64)
    a = lambda j: j            # noqa: E731  idx inside tower
    b = lambda j: 2 * d - j    # noqa: E731  b_j : b_{d-1} = d+1 ... b_0 = 2d
    c = d
    inter_s = np.stack([a(i), b(i + 1)], axis=1).reshape(-1)
    inter_d = np.stack([a(i + 1), b(i)], axis=1).reshape(-1)
    j = np.arange(d, dtype=np.int64)
    ts = np.concatenate([inter_s, [a(d - 1), c], a(j)])
    td = np.concatenate([inter_d, [c, b(d - 1)], b(j)])
    for t in range(towers):
        base = t * per
        src.append(ts + base)
        dst.append(td + base)
        if t + 1 < towers:
            src.append(np.array([b(0) + base]))
            dst.append(np.array([a(0) + base + per]))
    return from_plus_links(vid, np.concatenate(src), np.concatenate(dst))


class _PCG32:
    """PCG32 (XSH-RR) so the HPRC-shaped generator is reproducible everywhere."""

    def __init__(self, seed: int, seq: int = 54):
        self.m = (1 << 64) - 1
        self.state = 0
        self.inc = ((seq << 1) | 1) & self.m
        self.next()
        self.state = (self.state + seed) & self.m
        self.next()

    def next(self) -> int:
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & self.m
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF


def hprc_shaped(backbone_sizes, seed: int = 20260612, tiny: int = 0) -> Links:
    """HPRC-shaped multi-component graph: per component a backbone of
    N segments with bubble density 0.35/segment; bubble mix SNP 70 % / indel 20 % / nested 8 % /
    inversion 2 %; plus `tiny` small (<50 segment) components.  Uses numpy's PCG64 for the bulk
    draws (seeded) -- the graph is deterministic for a given numpy version and is always fed
    to both the oracle and the HIP path in the same process."""
    rng = np.random.Generator(np.random.PCG64(seed))
    vids, s_all, d_all, s1_all, s2_all = [], [], [], [], []
    next_id = 1
    base_idx = 0

    def emit_component(n_backbone: int):
        nonlocal next_id, base_idx
        kinds = rng.random(n_backbone - 1)
        has = rng.random(n_backbone - 1) < 0.35
        # extra segments per backbone gap: SNP 2 (two alleles), indel 1, nested 4, inversion 0
        k_snp = has & (kinds < 0.70)
        k_indel = has & (kinds >= 0.70) & (kinds < 0.90)
        k_nest = has & (kinds >= 0.90) & (kinds < 0.98)
        k_inv = has & (kinds >= 0.98)
        extra = np.zeros(n_backbone - 1, dtype=np.int64)
        extra[k_snp] = 2
        extra[k_indel] = 1
        extra[k_nest] = 4
        # layout: backbone segment i followed by its gap's extra segments
        stride = np.concatenate([[0], np.cumsum(1 + extra)])
        bb = stride[:-1] if len(stride) == n_backbone else stride[:n_backbone]
        bb = np.concatenate([bb, [stride[-1]]])[:n_backbone]
        n_local = int(stride[-1] + 1)
        a = bb[:-1]
        b = bb[1:]
        S, D, S1, S2 = [], [], [], []

        def plus(s, d):
            S.append(s)
            D.append(d)
            S1.append(np.full(len(s), R, dtype=np.uint8))
            S2.append(np.full(len(s), L, dtype=np.uint8))

        plain = ~has
        plus(a[plain], b[plain])
        # SNP: a>x a>y x>b y>b
        m = k_snp
        x, y = a[m] + 1, a[m] + 2
        plus(a[m], x); plus(a[m], y); plus(x, b[m]); plus(y, b[m])
        # indel: a>x x>b a>b
        m = k_indel
        x = a[m] + 1
        plus(a[m], x); plus(x, b[m]); plus(a[m], b[m])
        # nested: a>p p>q p>r q>s r>s s>b a>b   (p..s = a+1..a+4)
        m = k_nest
        p_, q_, r_, s_ = a[m] + 1, a[m] + 2, a[m] + 3, a[m] + 4
        plus(a[m], p_); plus(p_, q_); plus(p_, r_); plus(q_, s_); plus(r_, s_); plus(s_, b[m]); plus(a[m], b[m])
        # inversion: a>b plus the reverse link `b - a -` written as a `+ -` style link a+ -> b-
        m = k_inv
        plus(a[m], b[m])
        S.append(a[m]); D.append(b[m])
        S1.append(np.full(int(m.sum()), R, dtype=np.uint8)); S2.append(np.full(int(m.sum()), R, dtype=np.uint8))
        s = np.concatenate(S); d = np.concatenate(D)
        s1 = np.concatenate(S1); s2 = np.concatenate(S2)
        # L-line order: by source backbone position then as emitted (stable)
        order = np.argsort(np.minimum(s, d), kind="stable")
        vids.append(np.arange(next_id, next_id + n_local, dtype=np.uint32))
        s_all.append(s[order] + base_idx); d_all.append(d[order] + base_idx)
        s1_all.append(s1[order]); s2_all.append(s2[order])
        next_id += n_local
        base_idx += n_local

    for n in backbone_sizes:
        emit_component(int(n))
    for _ in range(tiny):
        emit_component(int(rng.integers(3, 20)))
    return _mk(np.concatenate(vids), np.concatenate(s_all), np.concatenate(s1_all), np.concatenate(d_all),
               np.concatenate(s2_all))


def hprc_circular(n_backbone: int, seed: int = 20260612) -> Links:
    """One HPRC-shaped component closed into a ring (the last backbone segment links back to the first): no side is left
    without a link, so the component has NO tip -- from_bd starts at (l, vertex 0) without a dummy root and gives that root a
    back edge to itself (spanning_tree.cpp:433-438).  The shape of a circular chromosome / plasmid / mitochondrial genome."""
    g = hprc_shaped([n_backbone], seed=seed)
    last = g.n_vtx - 1
    return _mk(g.vid, np.concatenate([g.v1, [last]]), np.concatenate([g.s1, np.array([R], dtype=np.uint8)]),
               np.concatenate([g.v2, [0]]), np.concatenate([g.s2, np.array([L], dtype=np.uint8)]))


def hub_on_chain(k_units: int = 100000, hub_links: int = 200000, seed: int = 1) -> Links:
    """A chain of bubbles with ONE hub segment: `hub_links` links from the r side of segment 0 to the l sides of segments drawn
    uniformly from the chain (repeats included): one side with 2 * 10^5 links, one 2-edge-connected class that holds most of
    the graph.  Nothing like a pangenome; the shape that finds quadratic corners."""
    base = chain_of_bubbles(k_units)
    rng = np.random.default_rng(seed)
    hv2 = rng.integers(1, base.n_vtx, size=hub_links)
    return _mk(base.vid, np.concatenate([base.v1, np.zeros(hub_links, dtype=np.int64)]),
               np.concatenate([base.s1, np.full(hub_links, R, dtype=np.uint8)]), np.concatenate([base.v2, hv2]),
               np.concatenate([base.s2, np.full(hub_links, L, dtype=np.uint8)]))


# chr1..22, X, Y lengths in Mbp: the relative sizes of the 24 large components of a whole-genome pangenome
CHR_MBP = (248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57)


def hprc_whole_genome(total_segments: float = 1e8, tiny: int = 2000, seed: int = 20260612) -> Links:
    """BASELINE config 4: 24 HPRC-shaped components with sizes proportional to chr1..22,X,Y
    summing to ~`total_segments` segments, plus `tiny` components of < 50 segments.  At the default size:
    99 860 187 segments / 122 435 438 links / 2 024 components."""
    # the generator emits ~1.675 segments per backbone segment
    sizes = [max(8, int(total_segments / 1.675 * m / sum(CHR_MBP))) for m in CHR_MBP]
    return hprc_shaped(sizes, seed=seed, tiny=tiny)


def hprc_tangled(n_backbone: int, seed: int = 20260612, tangle_every: int = 20000, max_tangle: int = 200000,
                 shuffle_links: bool = True) -> Links:
    """HPRC-shaped component with TANGLES: the chain of small bubbles of `hprc_shaped`, and every ~`tangle_every`
    backbone segments a tangle replaces a bubble -- `n` extra segments (heavy-tailed: n = 100 * 10^(3u), u uniform,
    capped at `max_tangle`, i.e. 10^2 .. 10^5+ segments) wired by a random spanning path plus 0.6 n random links with
    random sides (inversions, self loops, parallel links), entered from the backbone segment before it and left to the
    one behind it.  Each tangle is one large 2-edge-connected class (the case a chain of bubbles never produces); with
    `shuffle_links` the L lines of a tangle come in random order instead of sorted along the backbone."""
    rng = np.random.Generator(np.random.PCG64(seed))
    base = hprc_shaped([n_backbone], seed=seed)
    nv = base.n_vtx
    # backbone vertices of `base`: recover them as the articulation-free spine is not needed -- tangles hang between
    # two consecutive vertex indices chosen at random positions (a link a -> b of the spine is rerouted through the tangle)
    v1, v2, s1, s2 = [base.v1.astype(np.int64)], [base.v2.astype(np.int64)], [base.s1], [base.s2]
    n_t = max(1, n_backbone // tangle_every)
    # candidate attachment links: plain backbone links (+ +) between consecutive vertices
    plain = np.flatnonzero((base.v2.astype(np.int64) - base.v1.astype(np.int64) == 1) & (base.s1 == R) & (base.s2 == L))
    pick = np.sort(rng.choice(plain, size=min(n_t, len(plain)), replace=False))
    keep = np.ones(base.n_links, dtype=bool)
    keep[pick] = False
    nxt = nv
    ex_v1, ex_v2, ex_s1, ex_s2 = [], [], [], []
    for e in pick.tolist():
        n = int(min(max_tangle, 100 * 10 ** (3 * rng.random())))
        a, b = int(base.v1[e]), int(base.v2[e])
        ids = np.arange(nxt, nxt + n, dtype=np.int64)
        nxt += n
        perm = rng.permutation(ids)
        # spanning path through the tangle, entered from a and left to b
        p1 = np.concatenate([[a], perm])
        p2 = np.concatenate([perm, [b]])
        m = int(0.6 * n)
        r1 = rng.choice(ids, size=m)
        r2 = rng.choice(ids, size=m)
        t1 = np.concatenate([p1, r1]); t2 = np.concatenate([p2, r2])
        ts1 = np.concatenate([np.full(len(p1), R), rng.integers(0, 2, size=m)]).astype(np.uint8)
        ts2 = np.concatenate([np.full(len(p2), L), rng.integers(0, 2, size=m)]).astype(np.uint8)
        if shuffle_links:
            o = rng.permutation(len(t1))
            t1, t2, ts1, ts2 = t1[o], t2[o], ts1[o], ts2[o]
        ex_v1.append(t1); ex_v2.append(t2); ex_s1.append(ts1); ex_s2.append(ts2)
    n_all = nxt
    vid = np.arange(1, n_all + 1, dtype=np.uint32)
    V1 = np.concatenate([base.v1.astype(np.int64)[keep]] + ex_v1)
    V2 = np.concatenate([base.v2.astype(np.int64)[keep]] + ex_v2)
    S1 = np.concatenate([base.s1[keep]] + ex_s1)
    S2 = np.concatenate([base.s2[keep]] + ex_s2)
    return _mk(vid, V1, S1, V2, S2)


def random_bidirected(n_vtx: int, n_links: int, seed: int, self_loops: bool = True,
                      connected: bool = False) -> Links:
    """Differential-fuzz input: random sides, parallel links, self loops, several components."""
    rng = np.random.Generator(np.random.PCG64(seed))
    vid = np.sort(rng.choice(np.arange(1, 4 * n_vtx + 1), size=n_vtx, replace=False)).astype(np.uint32)
    v1 = rng.integers(0, n_vtx, size=n_links)
    v2 = rng.integers(0, n_vtx, size=n_links)
    if connected and n_vtx > 1:
        k = min(n_links, n_vtx - 1)
        v1[:k] = np.arange(k)
        v2[:k] = np.arange(1, k + 1)
    if not self_loops:
        same = v1 == v2
        v2[same] = (v2[same] + 1) % n_vtx
    s1 = rng.integers(0, 2, size=n_links)
    s2 = rng.integers(0, 2, size=n_links)
    # mostly "forward" links so that long chains / bubbles appear
    fwd = rng.random(n_links) < 0.7
    s1[fwd] = R
    s2[fwd] = L
    return _mk(vid, v1, s1, v2, s2)


def bubble_zoo(n_comp: int, sites: int, seed: int, shuffle_ids: bool = True) -> Links:
    """Many small components, each a backbone whose sites are small motifs drawn at random: SNPs, multi-allelic sites,
    indels, the chain-of-bubbles unit, repeated links, inversions, self loops (both kinds), a bubble nested in an allele,
    small random tangles, dangling hairpins, flipped interior segments.  The leaf flubbles of such graphs come in every
    shape the two relabelling passes of `-s` (find_tiny / find_parallel) distinguish, and the components are small enough
    that a back-edge INDEX often equals a tree vertex index (tiny.cpp:52-56 compares the two)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    links = []  # (a, side_a, b, side_b) over provisional vertex numbers
    nv = 0

    def new():
        nonlocal nv
        nv += 1
        return nv - 1

    def fwd(a, b, fa=False, fb=False):  # a -> b; a flipped segment presents its other side
        links.append((a, L if fa else R, b, R if fb else L))

    for _ in range(n_comp):
        prev = new()
        anchors = []
        for _ in range(int(rng.integers(1, sites + 1))):
            nxt = new()
            m = int(rng.integers(0, 16))
            flip = bool(rng.random() < 0.25)
            if m == 0:  # SNP
                for _ in range(2):
                    x = new()
                    fwd(prev, x, fb=flip)
                    fwd(x, nxt, fa=flip)
            elif m == 1:  # multi-allelic
                for _ in range(int(rng.integers(3, 7))):
                    x = new()
                    fwd(prev, x)
                    fwd(x, nxt)
            elif m == 2:  # indel
                x = new()
                fwd(prev, x, fb=flip)
                fwd(x, nxt, fa=flip)
                fwd(prev, nxt)
            elif m == 3:  # the unit of chain_of_bubbles
                x, y = new(), new()
                fwd(prev, x), fwd(prev, y), fwd(x, y), fwd(x, nxt), fwd(y, nxt), fwd(prev, nxt)
            elif m == 4:  # repeated links
                x = new()
                for _ in range(int(rng.integers(2, 4))):
                    fwd(prev, x)
                fwd(x, nxt)
                if rng.random() < 0.5:
                    fwd(x, nxt)
            elif m == 5:  # inversion: the allele is entered from both ends
                x = new()
                fwd(prev, x), fwd(x, nxt)
                fwd(prev, x, fb=True), fwd(x, nxt, fa=True)
            elif m == 6:  # self loops
                x = new()
                fwd(prev, x), fwd(x, nxt)
                if rng.random() < 0.5:
                    links.append((x, R, x, L))
                else:
                    s = int(rng.integers(0, 2))
                    links.append((x, s, x, s))
                if rng.random() < 0.5:
                    fwd(prev, nxt)
            elif m == 7:  # a SNP nested in one allele
                x, y, a, b = new(), new(), new(), new()
                fwd(prev, x), fwd(x, a), fwd(x, b), fwd(a, y), fwd(b, y), fwd(y, nxt)
                z = new()
                fwd(prev, z), fwd(z, nxt)
            elif m == 8:  # small random tangle between the anchors
                k = int(rng.integers(2, 6))
                xs = [new() for _ in range(k)]
                fwd(prev, xs[0]), fwd(xs[-1], nxt)
                for _ in range(int(rng.integers(k, 3 * k))):
                    a, b = int(rng.integers(0, k)), int(rng.integers(0, k))
                    links.append((xs[a], int(rng.integers(0, 2)), xs[b], int(rng.integers(0, 2))))
                for i in range(k - 1):
                    fwd(xs[i], xs[i + 1])
            elif m == 9:  # dangling hairpin and a plain step
                x = new()
                fwd(prev, x)
                links.append((x, R, x, R))
                fwd(prev, nxt)
            elif m == 10:  # two alleles of different length
                x, y, z = new(), new(), new()
                fwd(prev, x), fwd(x, nxt), fwd(prev, y), fwd(y, z), fwd(z, nxt)
            elif m == 11:  # funnel: several alleles merge into one segment before the site ends; plus the direct step
                k = int(rng.integers(3, 6))
                x = new()
                for _ in range(k):
                    y = new()
                    fwd(prev, y), fwd(y, x)
                fwd(x, nxt), fwd(prev, nxt)
            elif m == 12:  # the funnel the other way round
                k = int(rng.integers(3, 6))
                x = new()
                fwd(prev, x)
                for _ in range(k):
                    y = new()
                    fwd(x, y), fwd(y, nxt)
                fwd(prev, nxt)
            elif m == 13:  # a segment hanging off the site, reached from earlier anchors as well
                x = new()
                fwd(prev, x)
                fwd(prev, nxt)
                for a in anchors[-3:]:
                    if rng.random() < 0.6:
                        fwd(a, x)
                if rng.random() < 0.5 and anchors:
                    fwd(x, anchors[int(rng.integers(0, len(anchors)))], fb=True)
            elif m == 14:  # a tip hanging off the far end of the site
                x = new()
                fwd(prev, nxt)
                links.append((nxt, L, x, int(rng.integers(0, 2))))
            else:  # plain step, sometimes doubled
                fwd(prev, nxt)
                if rng.random() < 0.3:
                    fwd(prev, nxt)
            anchors.append(prev)
            prev = nxt
    perm = rng.permutation(nv) if shuffle_ids else np.arange(nv)
    arr = np.array(links, dtype=np.int64).reshape(-1, 4)
    order = rng.permutation(len(arr)) if shuffle_ids else np.arange(len(arr))
    arr = arr[order]
    vid = np.arange(1, nv + 1, dtype=np.uint32)
    return _mk(vid, perm[arr[:, 0]], arr[:, 1], perm[arr[:, 2]], arr[:, 3])


def hanger_family(prefix: int, suffix: int, variant: int, direct_first: bool = True) -> Links:
    """One component: `prefix` plain steps, an indel site (prev -> y -> nxt and prev -> nxt) with one more segment x
    hanging off the inner side of nxt, then `suffix` plain steps.  variant 0: x carries a loop between its two sides
    (its tree edge is a bridge: x gets a simplifying back edge); variant 1: x's far side is a tip and its near side is also
    reached from the first segment of the component (an ordinary back edge out of x); variant 2: both.  As the prefix
    grows, the tree vertex idx of the site's boundary sweeps across the back-edge indices of x's edges -- the
    coincidence tiny.cpp:52-56 turns into a 'tiny' label."""
    links = []
    n = prefix + 4 + suffix + 1
    first = 0
    for i in range(prefix):
        links.append((i, R, i + 1, L))
    prev, y, nxt, x = prefix, prefix + 1, prefix + 2, prefix + 3
    site = [(prev, R, nxt, L), (prev, R, y, L), (y, R, nxt, L)]
    if not direct_first:
        site = site[1:] + site[:1]
    links += site
    links.append((nxt, L, x, L))
    if variant in (0, 2):
        links.append((x, R, x, L))
    if variant in (1, 2) and prefix > 0:
        links.append((first, R, x, L))
    last = nxt
    for i in range(suffix):
        links.append((last, R, prefix + 4 + i, L))
        last = prefix + 4 + i
    arr = np.array(links, dtype=np.int64)
    return _mk(np.arange(1, n + 1, dtype=np.uint32), arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3])
