// par_kernels.hip -- rows D-G as data-parallel kernels over ALL components at once.
//
// The reference computes these stages with sequential sweeps (reverse pre-order bracket lists,
// a list-splicing tree walk, a hash-map sweep and a stack machine).  Because the tree vertex idx
// IS the DFS pre-order number, every one of them has an exact closed form over pre-order
// intervals [v, v+size(v)); the kernels below evaluate those closed forms with scans, radix
// sorts and min-segment-tree descents.  DESIGN.md ("Parallel formulations") derives each one;
// tests/ differential-test them against the oracle (and the sequential kernels) bit for bit.
//
//   D  cycle classes      flubbles.cpp:503-719 (handle_vertex), bracket_list.cpp:61-100
//   E  candidate stack    tree_utils.cpp:19-155, flubbles.cpp:412-501
//   F  next_seen          flubbles.cpp:375-410
//   G  PVST               flubbles.cpp:295-367
//
// Index spaces: T-space = global tree vertex idx (component c starts at 2*voff[c]+c, holes are
// marked gsize == 0); bracket space (dense, all components); stack space (one entry per black
// tree edge, component-major); emitted-flubble space.
#include "par_kernels.hpp"
#include "segtree.hpp"

#include <algorithm>
#include <cstdlib>

namespace povu_hip
{

#define NIL POVU_NIL
static constexpr int TPB = 256;
static constexpr size_t PVST_STAGE_MIN = size_t(1) << 20; // PVST vertices from which the result goes through a device block
static inline unsigned nblk(size_t n) { return (unsigned)((n + TPB - 1) / TPB); }
#define LAUNCH(k, n, s, ...)                                                                     \
	do {                                                                                     \
		if ((n) > 0) {                                                                   \
			hipLaunchKernelGGL(k, dim3(nblk(n)), dim3(TPB), 0, s, __VA_ARGS__);      \
			HIP_CHECK(hipGetLastError());                                            \
		}                                                                                \
	} while (0)

// ------------------------------------------------------------- T-space setup
// component c owns the tree-vertex slots [2 voff[c] + c, 2 voff[c+1] + c]: two per segment and one more
// (the dummy root, when it exists, shifts the sides up by one); t_comp[T] = NIL closes the array
__global__ void k_tcomp_vertices(uint32_t V, uint32_t T, const uint32_t *__restrict__ ckey, const uint32_t *__restrict__ voff,
				 uint32_t *__restrict__ t_comp)
{
	uint32_t i = BIDX * blockDim.x + threadIdx.x;
	if (i >= V)
		return;
	uint32_t c = ckey[i];
	t_comp[2 * i + c] = c;
	t_comp[2 * i + 1 + c] = c;
	if (i + 1 == voff[c + 1])
		t_comp[2 * i + 2 + c] = c;
	if (i == V - 1)
		t_comp[T] = NIL;
}
__global__ void k_globalize(uint32_t T, const uint32_t *__restrict__ t_comp, const uint32_t *__restrict__ voff,
			    const uint32_t *__restrict__ c_ntree, const uint32_t *__restrict__ t_par,
			    const uint32_t *__restrict__ t_size, uint32_t *__restrict__ gpar, uint32_t *__restrict__ gsize,
			    uint32_t *__restrict__ t_root, uint32_t *__restrict__ hi0, uint32_t *__restrict__ cov,
			    const uint32_t *__restrict__ depth, uint32_t *__restrict__ mpre, uint32_t *__restrict__ incnt,
			    uint32_t *__restrict__ srccnt)
{
	uint32_t t = BIDX * blockDim.x + threadIdx.x;
	if (t >= T)
		return;
	srccnt[t] = 0; // brackets per source, [T+2]
	if (t == T - 1)
		srccnt[T] = srccnt[T + 1] = 0;
	if (cov) { // (not on the dense path: the tree stage left hi0 and the count of bracket ends, and needs no coverage array)
		hi0[t] = NIL; // k_hi0 takes minima into it
		cov[t] = 0;   // ... and counts back-edge ends here
		incnt[t] = 0; // brackets that end at a vertex, [T+2]
		if (t == T - 1) {
			hi0[T] = NIL;
			cov[T] = 0;
			incnt[T] = incnt[T + 1] = 0;
		}
	} else if (t == T - 1) {
		hi0[T] = NIL;
	}
	uint32_t c = t_comp[t];
	uint32_t base = 2 * voff[c] + c, l = t - base, n = c_ntree[c];
	t_root[t] = base;
	if (l < n) {
		uint32_t p = t_par[t], sz = t_size[t];
		gpar[t] = p == NIL ? NIL : base + p;
		gsize[t] = sz;
		// mirror pre-order (children visited in DESCENDING idx): the order brackets sit in a bracket list, because each
		// child's list is spliced in front of its earlier siblings' (flubbles.cpp:586-588).  With q(v) = mpre(v) + v + size(v)
		// one gets q(child) = q(parent) + 1, hence mpre(v) = depth(v) + N - v - size(v) (local indices, N = tree vertices
		// of the component)
		mpre[t] = base + depth[t] + n - l - sz;
	} else {
		gpar[t] = NIL;
		gsize[t] = 0;
		mpre[t] = NIL;
	}
}
__global__ void k_dense_be(uint32_t NB0, uint32_t C, const uint32_t *__restrict__ dbo, const uint32_t *__restrict__ voff,
			   const uint32_t *__restrict__ eoff, const uint32_t *__restrict__ be_src,
			   const uint32_t *__restrict__ be_tgt, uint32_t *__restrict__ b_src, uint32_t *__restrict__ b_tgt)
{
	uint32_t j = BIDX * blockDim.x + threadIdx.x;
	if (j >= NB0)
		return;
	uint32_t lo = 0, hi = C; // last c with dbo[c] <= j
	while (hi - lo > 1) {
		uint32_t mid = (lo + hi) >> 1;
		if (dbo[mid] <= j)
			lo = mid;
		else
			hi = mid;
	}
	uint32_t c = lo, b = j - dbo[c];
	uint64_t tb = 2ull * voff[c] + c, bb = (uint64_t)eoff[c] + voff[c] + 2 * tb;
	b_src[j] = (uint32_t)tb + be_src[bb + b];
	b_tgt[j] = (uint32_t)tb + be_tgt[bb + b];
}

// ------------------------------------------------------------- row D
// hi0(v) = min target of v's own back edges; cov(v) = back edges leaving v minus back edges arriving at v.  A back
// edge runs from a descendant to an ancestor, so the sum of cov over subtree(v) counts exactly the back edges that
// leave subtree(v) upwards past v.
// (incnt, when given: brackets that end at a vertex -- the ordinary ones are counted here, while their ends are in hand)
__global__ void k_hi0(uint32_t NB0, const uint32_t *__restrict__ b_src, const uint32_t *__restrict__ b_tgt,
		      uint32_t *__restrict__ hi0, uint32_t *__restrict__ cov, uint32_t *__restrict__ incnt)
{
	uint32_t j = BIDX * blockDim.x + threadIdx.x;
	if (j >= NB0)
		return;
	const uint32_t sv = b_src[j], tv = b_tgt[j];
	atomicMin(&hi0[sv], tv);
	if (incnt) { // (dense path: cov = edges leaving - edges arriving is read off the two counts, see run_parallel_dg)
		atomicAdd(&incnt[tv], 1u);
	} else if (sv != tv) {
		atomicAdd(&cov[sv], 1u);
		atomicSub(&cov[tv], 1u);
	}
}
// bridge(v): no ordinary back edge out of subtree(v) reaches a proper ancestor of v (the bracket list of v would be
// empty but for simplifying edges): the subtree sum of cov is zero
// (four vertices a lane, 16-byte loads: see k_entry_list)
__global__ void __launch_bounds__(TPB) k_bridge_flags(uint32_t T, const uint32_t *__restrict__ gsize, const uint32_t *__restrict__ gpar,
						       const uint32_t *__restrict__ pscov, uint4 *__restrict__ brec)
{
	// the flags go out as BITS, 64 to a record of the bit-rank directory (common.hpp): "how many bridge vertices in front of x"
	// is then one 16-byte look-up into a 50 MB table -- until round 5 a byte per vertex, scanned into a 4-byte prefix per vertex
	const uint32_t t0 = (BIDX * blockDim.x + threadIdx.x) * 4u;
	uint32_t f = 0;
	if (t0 + 4 <= T) {
		const uint4 sz = *reinterpret_cast<const uint4 *>(gsize + t0), gp = *reinterpret_cast<const uint4 *>(gpar + t0);
		const uint4 pc = *reinterpret_cast<const uint4 *>(pscov + t0);
		f = ((sz.x && gp.x != NIL && pscov[t0 + sz.x] == pc.x) ? 1u : 0u) | ((sz.y && gp.y != NIL && pscov[t0 + 1 + sz.y] == pc.y) ? 2u : 0u) |
		    ((sz.z && gp.z != NIL && pscov[t0 + 2 + sz.z] == pc.z) ? 4u : 0u) | ((sz.w && gp.w != NIL && pscov[t0 + 3 + sz.w] == pc.w) ? 8u : 0u);
	} else {
		for (uint32_t t = t0; t < T; t++) {
			const uint32_t sz = gsize[t];
			f |= ((sz && gpar[t] != NIL && pscov[t + sz] == pscov[t]) ? 1u : 0u) << (t - t0);
		}
	}
	const uint32_t w0 = (BIDX * blockDim.x + (threadIdx.x & ~63u)) / 16u; // first record of this wave's 256 vertices
	bitrank_store_wave(brec + w0, f, w0 + (threadIdx.x & 63u) <= T / 64u);
}
// capping back edge v -> hi_2 when hi_2 < hi_0 (flubbles.cpp:555-574, 613-619).  Children of v in
// ascending idx are v+1, then each next sibling at c + size(c).  hi(c) = min target of the back edges leaving
// subtree(c) -- the root as soon as subtree(c) holds a simplifying edge -- is only ever compared between siblings,
// so it is evaluated on demand, for the children of branching vertices alone (range-min over hi0).
// Root of the tree that holds tree vertex t.  The all-parallel pass keeps no per-vertex table of it (it is needed at
// few vertices): component c owns [2 voff[c] + c, 2 voff[c+1] + c], so the component is found by bisection over voff.
struct RootOf {
	const uint32_t *t_root; // per-vertex table, or null: bisect
	const uint32_t *voff;
	uint32_t C;
	__device__ __forceinline__ uint32_t operator()(uint32_t t) const
	{
		if (t_root)
			return t_root[t];
		uint32_t lo = 0, hi = C; // last c with 2 voff[c] + c <= t
		while (hi - lo > 1) {
			const uint32_t mid = (lo + hi) >> 1;
			if (2 * voff[mid] + mid <= t)
				lo = mid;
			else
				hi = mid;
		}
		return 2 * voff[lo] + lo;
	}
};
__device__ __forceinline__ void capping_of(uint32_t v, const uint32_t *__restrict__ gsize, const uint32_t *__restrict__ hi0,
					   const uint4 *__restrict__ brec, const RootOf &root_of,
					   const SegTree &segA, uint32_t *__restrict__ cap_tgt,
					   uint8_t *__restrict__ capf, uint32_t *__restrict__ literal_rule_seen)
{
	const uint32_t end = v + gsize[v];
	uint32_t root = NIL; // of v's tree, looked up at most once
	auto hi_of = [&](uint32_t c) {
		const uint32_t cs = max(gsize[c], 1u);
		if (bitrank(brec, c + cs) != bitrank(brec, c)) { // a bridge vertex in subtree(c): a simplifying edge sits at or below it
			if (root == NIL)
				root = root_of(v);
			return root;
		}
		if (cs <= 8) { // a small branch: its hi0 values share a cache line or two
			uint32_t m = NIL;
			for (uint32_t k = c; k < c + cs; k++)
				m = min(m, hi0[k]);
			return m;
		}
		return seg_min(segA, c, c + cs);
	};
	// one sweep over the children: the first child that attains the minimum (hi_child), and the first two children
	// whose hi lies above v in the tree -- hi_2 belongs to the first of them that is not hi_child
	uint32_t hi_1 = NIL, hi_child = NIL, lt_c[2] = {NIL, NIL}, lt_h[2] = {NIL, NIL};
	uint32_t second = NIL; // the second smallest hi over the children, with multiplicity: what hi_2 "should" be
	for (uint32_t c = v + 1; c < end; c += max(gsize[c], 1u)) {
		const uint32_t h = hi_of(c);
		if (h < hi_1 || hi_child == NIL) {
			second = hi_1;
			hi_1 = h;
			hi_child = c;
		} else {
			second = min(second, h);
		}
		if (h < v) {
			if (lt_c[0] == NIL)
				lt_c[0] = c, lt_h[0] = h;
			else if (lt_c[1] == NIL)
				lt_c[1] = c, lt_h[1] = h;
		}
	}
	const uint32_t hi_2 = lt_c[0] != hi_child ? lt_h[0] : lt_h[1];
	// The literal rule (first other child in idx order with hi < v, not the one reaching highest) can leave a capping
	// edge that ends too early; only then can a bracket come back on top with MORE brackets under it than before.
	// The black-edge-only class pass relies on that never happening (see run_parallel_dg), so say when it might.
	if (hi_2 != (second < v ? second : NIL))
		*literal_rule_seen = 1u;
	if (hi_2 < hi0[v]) {
		cap_tgt[v] = hi_2;
		capf[v] = 1;
	}
}
// simplifying(v): bridge(v) and no bridge vertex below it (the deepest ones get the back edge to
// the root, flubbles.cpp:621-643)
// ... and which vertices BRANCH (two or more children): only they can get a capping edge, about one vertex in nine.
// The sweep over a branching vertex's children is a chain of dependent gathers (child, its size, the flags behind its
// subtree, the next child ...): it runs in a kernel of its own over the compacted list of branching vertices, every lane
// busy with one of them -- inside this kernel the few lanes that had work left the rest of their workgroup waiting.
static constexpr uint32_t HS_ITER = LIST_ITER, HS_VERTS = LIST_SPAN; // vertices a workgroup of k_hi_simp takes: one atomic add for 16 384 of them
static_assert(TPB == (int)LIST_TPB, "append_in_order is written for workgroups of 256");
__global__ void __launch_bounds__(TPB) k_hi_simp(uint32_t T, const uint32_t *__restrict__ gsize, const uint4 *__restrict__ brec,
						  uint8_t *__restrict__ simp, uint8_t *__restrict__ hpf,
						  uint8_t *__restrict__ capf, uint32_t *__restrict__ br_list, uint32_t *__restrict__ n_br)
{
	const uint32_t B0 = BIDX * HS_VERTS;
	// p0 = bridge vertices in front of t: simplifying iff t is the only one of its subtree
	auto one = [&](uint32_t t, uint32_t sz, uint32_t br, uint32_t p0, uint32_t sz_next, uint32_t &sm, uint32_t &bch) {
		sm = (sz && br && bitrank(brec, t + sz) - p0 == 1) ? 1u : 0u;
		bch = (sz > 2 && t + 1 + max(sz_next, 1u) < t + sz) ? 1u : 0u; // the first child does not fill the subtree
	};
	unsigned long long bw = 0; // bit 4 it + j: vertex B0 + it * 1024 + 4 tid + j branches
#pragma unroll 4
	for (uint32_t it = 0; it < HS_ITER; it++) {
		const uint32_t t0 = B0 + it * (TPB * 4u) + threadIdx.x * 4u; // (four vertices a lane, 16-byte loads: see k_entry_list)
		uint32_t b4 = 0;
		if (t0 + 4 < T) { // (strictly: the last of the four looks at its successor's size)
			const uint4 sz = *reinterpret_cast<const uint4 *>(gsize + t0);
			const uint32_t sz4 = gsize[t0 + 4];
			// the four flags and the count in front of them out of one record (t0 is a multiple of four: one record holds all four)
			const uint4 r = brec[t0 >> 6];
			const unsigned long long bits = (unsigned long long)r.x | ((unsigned long long)r.y << 32);
			const uint32_t br = (uint32_t)(bits >> (t0 & 63u)) & 15u;
			const uint32_t p0 = r.z + (uint32_t)__popcll(bits & ((1ull << (t0 & 63u)) - 1ull));
			const uint32_t p1 = p0 + (br & 1u), p2 = p1 + ((br >> 1) & 1u), p3 = p2 + ((br >> 2) & 1u);
			uint32_t s0, s1, s2, s3, c0, c1, c2, c3;
			one(t0, sz.x, br & 1u, p0, sz.y, s0, c0);
			one(t0 + 1, sz.y, br & 2u, p1, sz.z, s1, c1);
			one(t0 + 2, sz.z, br & 4u, p2, sz.w, s2, c2);
			one(t0 + 3, sz.w, br & 8u, p3, sz4, s3, c3);
			const uint32_t sw = s0 | (s1 << 8) | (s2 << 16) | (s3 << 24);
			*reinterpret_cast<uint32_t *>(simp + t0) = sw;
			if (hpf)
				*reinterpret_cast<uint32_t *>(hpf + t0) = sw;
			*reinterpret_cast<uint32_t *>(capf + t0) = 0u; // (cap_tgt is only read where capf says so)
			b4 = c0 | (c1 << 1) | (c2 << 2) | (c3 << 3);
		} else {
			for (uint32_t t = t0; t < T; t++) {
				uint32_t sm, bch;
				one(t, gsize[t], bitrank_test(brec, t) ? 1u : 0u, bitrank(brec, t), gsize[t + 1], sm, bch);
				simp[t] = (uint8_t)sm;
				if (t == T - 1)
					simp[T] = 0;
				if (hpf)
					hpf[t] = (uint8_t)sm;
				capf[t] = 0;
				b4 |= bch << (t - t0);
			}
		}
		bw |= (unsigned long long)b4 << (4 * it);
	}
	// the branching vertices go straight onto the list k_capping works through, in vertex order (append_in_order, common.hpp)
	append_in_order(bw, B0, br_list, n_br);
}
__global__ void k_capping(const uint32_t *__restrict__ n_list, const uint32_t *__restrict__ list, const uint32_t *__restrict__ gsize,
			  const uint32_t *__restrict__ hi0, const uint4 *__restrict__ brec, const RootOf root_of, const SegTree segA,
			  uint32_t *__restrict__ cap_tgt, uint8_t *__restrict__ capf, uint32_t *__restrict__ literal_rule_seen)
{
	const uint32_t n = *n_list; // (the count only exists on the device: grid-stride)
	for (uint32_t i = BIDX * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		capping_of(list[i], gsize, hi0, brec, root_of, segA, cap_tgt, capf, literal_rule_seen);
}
// Bracket list order = (mirror pre-order of the source) and, inside one source, the later pushed
// first: simplifying, capping, ordinary edges by descending creation idx (flubbles.cpp:608-643).
// The brackets are laid out in an initial order that already satisfies the second criterion
// (simplifying | capping | ordinary by DESCENDING dense idx), so ONE stable 32-bit radix sort by the
// source's mirror pre-order yields the list order.
// Where a capping / simplifying bracket goes needs the rank of its vertex among the flagged ones.  No prefix arrays: the
// flags are counted per workgroup of k_bracket_extra (k_flag_tile_counts, 256 vertices a tile), the two count arrays
// scanned, and k_bracket_extra ranks its own 256 flags again with ballots.
static constexpr uint32_t BX_TILE = TPB;
__global__ void __launch_bounds__(TPB) k_flag_tile_counts(uint32_t T, const uint8_t *__restrict__ capf, const uint8_t *__restrict__ simp,
							   uint32_t *__restrict__ tcap, uint32_t *__restrict__ tsimp, uint32_t ntiles)
{
	// a workgroup counts 16 tiles: every lane takes 16 consecutive vertices (one 16-byte load per flag array), sixteen
	// consecutive lanes make a tile
	const uint32_t e0 = (blockIdx.x * TPB + threadIdx.x) * 16u;
	uint32_t nc = 0, ns = 0;
	if (e0 + 16 <= T) {
		const uint4 a = *reinterpret_cast<const uint4 *>(capf + e0), b = *reinterpret_cast<const uint4 *>(simp + e0);
		const uint32_t wa[4] = {a.x, a.y, a.z, a.w}, wb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
		for (int k = 0; k < 4; k++)
#pragma unroll
			for (int q = 0; q < 4; q++) {
				nc += ((wa[k] >> (8 * q)) & 0xFFu) ? 1u : 0u;
				ns += ((wb[k] >> (8 * q)) & 0xFFu) ? 1u : 0u;
			}
	} else {
		for (uint32_t k = e0; k < T; k++) {
			nc += capf[k] ? 1u : 0u;
			ns += simp[k] ? 1u : 0u;
		}
	}
	for (int off = 8; off; off >>= 1) {
		nc += __shfl_down(nc, off, 16);
		ns += __shfl_down(ns, off, 16);
	}
	const uint32_t tile = (blockIdx.x * TPB + threadIdx.x) / 16u;
	if ((threadIdx.x & 15u) == 0 && tile <= ntiles) { // (tile == ntiles closes the arrays: the scan leaves the totals there)
		tcap[tile] = tile < ntiles ? nc : 0u;
		tsimp[tile] = tile < ntiles ? ns : 0u;
	}
}
// Four vertices a lane (16-byte loads: a kernel of a few loads per element is bound by its memory INSTRUCTIONS), so a
// wave covers exactly one tile of 256 vertices and ranks its flags with eight ballots; no LDS, no barrier.
// (the counts it places by -- ordinary back edges, capping vertices -- are read from the device: the kernel is launched
// before the host has them, see the driver)
__global__ void __launch_bounds__(TPB) k_bracket_extra(uint32_t T, uint32_t NB0_host, const uint32_t *__restrict__ nb0_dev, uint32_t ntiles,
							uint32_t nb_cap, const uint8_t *__restrict__ capf,
							const uint32_t *__restrict__ tcap, const uint8_t *__restrict__ simp,
							const uint32_t *__restrict__ tsimp, const uint32_t *__restrict__ cap_tgt,
							const RootOf root_of, uint32_t *__restrict__ b_src, uint32_t *__restrict__ b_tgt,
							const uint32_t *__restrict__ ordcnt, const uint32_t *__restrict__ gsize,
							const uint32_t *__restrict__ mpre, uint32_t *__restrict__ incnt,
							uint32_t *__restrict__ srccnt)
{
	const uint32_t NB0 = nb0_dev ? *nb0_dev : NB0_host, ncap = tcap[ntiles];
	if ((uint64_t)NB0 + ncap + tsimp[ntiles] > nb_cap)
		return; // (an internal sizing bug: the host throws when it reads the counts; nothing is written out of bounds)
	const uint32_t v0 = (BIDX * blockDim.x + threadIdx.x) * 4u, lane = threadIdx.x & 63u, tile = v0 / BX_TILE;
	uint32_t cw = 0, sw = 0; // the four flag bytes of this lane
	if (v0 + 4 <= T) {
		cw = *reinterpret_cast<const uint32_t *>(capf + v0);
		sw = *reinterpret_cast<const uint32_t *>(simp + v0);
	} else {
		for (uint32_t j = 0; v0 + j < T && j < 4; j++) {
			cw |= (capf[v0 + j] ? 1u : 0u) << (8 * j);
			sw |= (simp[v0 + j] ? 1u : 0u) << (8 * j);
		}
	}
	// rank among the flagged vertices: the tiles before (scanned counts) + the lanes before + this lane's own earlier ones
	const unsigned long long lt = (1ull << lane) - 1ull;
	uint32_t rc = 0, rs = 0;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		rc += (uint32_t)__popcll(__ballot(((cw >> (8 * j)) & 0xFFu) != 0) & lt);
		rs += (uint32_t)__popcll(__ballot(((sw >> (8 * j)) & 0xFFu) != 0) & lt);
	}
	if (v0 >= T)
		return;
	if (cw)
		rc += tcap[tile];
	if (sw)
		rs += tsimp[tile];
	uint4 oc = make_uint4(0u, 0u, 0u, 0u), gs = oc, mp = oc;
	if (ordcnt) {
		if (v0 + 4 <= T) {
			oc = *reinterpret_cast<const uint4 *>(ordcnt + v0);
			gs = *reinterpret_cast<const uint4 *>(gsize + v0);
			mp = *reinterpret_cast<const uint4 *>(mpre + v0);
		} else {
			uint32_t *o = &oc.x, *g = &gs.x, *m = &mp.x;
			for (uint32_t j = 0; v0 + j < T && j < 4; j++)
				o[j] = ordcnt[v0 + j], g[j] = gsize[v0 + j], m[j] = mpre[v0 + j];
		}
	}
	const uint32_t ocv[4] = {oc.x, oc.y, oc.z, oc.w}, gsv[4] = {gs.x, gs.y, gs.z, gs.w}, mpv[4] = {mp.x, mp.y, mp.z, mp.w};
#pragma unroll
	for (uint32_t j = 0; j < 4; j++) {
		const uint32_t v = v0 + j;
		if (v >= T)
			break;
		const uint32_t c = (cw >> (8 * j)) & 0xFFu, sm = (sw >> (8 * j)) & 0xFFu;
		if (c) {
			const uint32_t q = NB0 + rc++;
			const uint32_t tg = cap_tgt[v];
			b_src[q] = v;
			b_tgt[q] = tg;
			if (ordcnt)
				atomicAdd(&incnt[tg], 1u);
		}
		if (sm) {
			const uint32_t q = NB0 + ncap + rs++;
			const uint32_t root = root_of(v);
			b_src[q] = v;
			b_tgt[q] = root;
			if (ordcnt)
				atomicAdd(&incnt[root], 1u);
		}
		// (dense path) brackets per source, at its place in the list order: known per vertex, no counting pass over the brackets
		if (ordcnt && gsv[j])
			srccnt[mpv[j]] = ocv[j] + (c ? 1u : 0u) + (sm ? 1u : 0u);
	}
}
__global__ void k_bracket_order(uint32_t NB, uint32_t NB0, uint32_t ncap, uint32_t nsimp, const uint32_t *__restrict__ b_src,
				const uint32_t *__restrict__ b_tgt, const uint32_t *__restrict__ mpre,
				uint32_t *__restrict__ key, uint32_t *__restrict__ val, uint32_t *__restrict__ incnt,
				uint32_t *__restrict__ srccnt)
{
	uint32_t q = BIDX * blockDim.x + threadIdx.x;
	if (q >= NB)
		return;
	uint32_t j;
	if (q < nsimp)
		j = NB0 + ncap + q;
	else if (q < nsimp + ncap)
		j = NB0 + (q - nsimp);
	else
		j = NB0 - 1 - (q - nsimp - ncap);
	uint32_t m = mpre[b_src[j]];
	key[q] = m;
	val[q] = j;
	atomicAdd(&incnt[b_tgt[j]], 1u);
	atomicAdd(&srccnt[m], 1u); // brackets per mirror pre-order position -> range starts
}
// With the per-source rank of every ordinary edge known (parallel tree stage), the list order needs no
// sort: a bracket's position = (first bracket of its source's mirror pre-order position) + its rank
// inside the source (simplifying, capping, ordinary edges top first).
__global__ void k_bracket_place(uint32_t NB, uint32_t NB0, uint32_t ncap, const uint32_t *__restrict__ b_src,
				const uint32_t *__restrict__ b_tgt, const uint32_t *__restrict__ b_ord,
				const uint32_t *__restrict__ mpre, const uint32_t *__restrict__ bstart,
				const uint8_t *__restrict__ capf, const uint8_t *__restrict__ simp,
				uint32_t *__restrict__ tgtR, uint32_t *__restrict__ rid)
{
	uint32_t j = BIDX * blockDim.x + threadIdx.x;
	if (j >= NB)
		return;
	const uint32_t v = b_src[j];
	uint32_t rank;
	if (j < NB0) // b_ord counts from the bottom of the source's ordinary edges; the list has the later pushed on top
		rank = bstart[mpre[v] + 1] - bstart[mpre[v]] - 1 - b_ord[j];
	else if (j < NB0 + ncap)
		rank = simp[v];
	else
		rank = 0;
	const uint32_t pos = bstart[mpre[v]] + rank;
	tgtR[pos] = b_tgt[j];
	if (rid) // (which bracket sits at a list position: only the hairpin report asks, k_top_bracket)
		rid[pos] = j;
}
__global__ void k_gather_u32(uint32_t n, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ src,
			     uint32_t *__restrict__ dst)
{
	uint32_t i = BIDX * blockDim.x + threadIdx.x;
	if (i < n)
		dst[i] = src[idx[i]];
}
// top bracket and bracket-list size at v when the class of tree edge (parent(v), v) is decided
// (flubbles.cpp:664-686): the live brackets are those with source in subtree(v) and target a
// proper ancestor of v; in list order they are the contiguous range [bstart[mpre(v)],
// bstart[mpre(v)+size(v)]), the top is the first live one.  Output in DESCENDING v order so that a
// stable sort by top bracket leaves every group ordered from the deepest vertex up.
// where the candidate-stack entry of a black vertex goes (row E worked out before the class pass)
struct StackPlace {
	const uint32_t *voff, *soff, *dlt, *dlt_ps;
	uint32_t *s_vtx, *s_comp;
};
// BLACK: only the child ends of black tree edges take part (n = V of them: segment slot g of component c is tree
// vertex 2g + c + [dummy root] + 1, the opposite side follows the entered side in pre-order) -- the candidate stack
// holds no other edge, see run_parallel_dg for when that is enough.
template <bool BLACK>
__global__ void k_top_bracket(uint32_t n, const uint32_t *__restrict__ gsize, const uint32_t *__restrict__ gpar,
			      const uint32_t *__restrict__ mpre, const uint32_t *__restrict__ bstart,
			      const SegTree segB, const uint32_t *__restrict__ tgtR,
			      const uint32_t *__restrict__ psin, uint32_t *__restrict__ ckey, uint32_t *__restrict__ cval,
			      uint32_t *__restrict__ lsz, uint32_t *__restrict__ err, const uint32_t *__restrict__ rid,
			      uint32_t first_simp_id, uint8_t *__restrict__ hpf, const uint32_t *__restrict__ seg_comp,
			      const uint32_t *__restrict__ c_ntree, const StackPlace sp)
{
	const uint32_t q = BIDX * blockDim.x + threadIdx.x;
	if (q >= n)
		return;
	uint32_t v = n - 1 - q, c = 0;
	const uint32_t g = v;
	if (BLACK) {
		c = seg_comp[g];
		v = 2 * g + c + (c_ntree[c] & 1u) + 1;
	}
	uint32_t sz = gsize[v];
	if (sz == 0 || (!BLACK && gpar[v] == NIL)) { // (the far end of a black edge always has a parent: the entered side)
		cval[q] = v;
		ckey[q] = NIL;
		if (!BLACK)
			lsz[v] = 0;
		return;
	}
	// BLACK: everything downstream lives in candidate-stack order, so the sort carries the entry's stack index and the
	// entry itself (its tree vertex, its component) is written here; lsz is indexed by stack position too
	uint32_t at = v;
	if (BLACK) {
		at = sp.soff[c] + (g - sp.voff[c]) + sp.dlt_ps[g] + sp.dlt[g];
		sp.s_vtx[at] = v;
		sp.s_comp[at] = c;
	}
	cval[q] = at;
	uint32_t m = mpre[v];
	uint32_t lo = bstart[m], hi = bstart[m + sz];
	// Number of live brackets first (two prefix sums): in a chain of bubbles the brackets of everything below v are closed
	// inside their bubbles and the few live ones -- a simplifying edge, a tip's edge to the root -- come from the
	// deepest sources, i.e. sit at the END of the range.  When the last `live` entries are all live they are THE live
	// ones, and the top bracket is the first of them: no search at all.
	const uint32_t live = (hi - lo) - (psin[v + sz] - psin[v]);
	uint32_t i = NIL;
	if (live >= 1 && live <= 4) { // (the four words in one 16-byte load; the array carries slack behind its last entry)
		const uint4 a = load4_unaligned(tgtR + (hi - live));
		const bool all = a.x < v && (live < 2 || a.y < v) && (live < 3 || a.z < v) && (live < 4 || a.w < v);
		if (all)
			i = hi - live;
	}
	// otherwise the top bracket often is one of the first few of the range: probe them (one load)
	// before falling back to the O(log n) descent
	if (i == NIL) {
		const uint32_t np = min(hi - lo, 4u);
		const uint4 a = load4_unaligned(tgtR + lo);
		if (np > 0 && a.x < v)
			i = lo;
		else if (np > 1 && a.y < v)
			i = lo + 1;
		else if (np > 2 && a.z < v)
			i = lo + 2;
		else if (np > 3 && a.w < v)
			i = lo + 3;
		if (i == NIL && lo + np < hi)
			i = seg_first_less(segB, lo + np, hi, v);
	}
	if (i == NIL) {
		atomicAdd(&err[0], 1u); // cannot happen: every list holds at least a simplifying bracket
		ckey[q] = NIL;
		lsz[at] = 0;
		return;
	}
	lsz[at] = live;
	ckey[q] = i;
	if (hpf && rid[i] >= first_simp_id) // the top bracket is a simplifying edge (flubbles.cpp:644-656)
		hpf[v] |= 2;
}
// (Tried in round 5 and dropped: the black-edge-only form with FOUR segment slots a lane -- component, shift, size and list
// position words in 16-byte loads, keys and stack indices in 16-byte stores.  3.43 ms against 1.86: the look-ups into the
// bracket list are chains of dependent gathers, and four of them one after the other in a lane cost more than the narrow
// loads of four lanes did; the wide form only pays in kernels that stream, gpurun_out r5l.)
// a bracket hands out a new class whenever the list size differs from the size it saw last
// (recent_size / recent_class, flubbles.cpp:668-676)
// (also row F's marks: q + 1 where the vertex at sorted position q ends a black edge, see k_next_from_runs)
// BLACK: n = V sorted entries, all of them black edges (no marks).
template <bool BLACK>
__global__ void k_class_flags(uint32_t n, uint32_t V, const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval,
			      const uint32_t *__restrict__ lsz, const uint8_t *__restrict__ tf, uint8_t *__restrict__ flag,
			      uint32_t *__restrict__ dlt, uint32_t *__restrict__ mark)
{
	uint32_t q = BIDX * blockDim.x + threadIdx.x;
	if (!BLACK && q < V + 2)
		dlt[q] = 0; // row E's difference array, [V+2] (launched with max(n, V + 2) threads; BLACK: row E ran already)
	if (q >= n)
		return;
	uint32_t k = skey[q];
	if (k == NIL) {
		flag[q] = 0;
		if (!BLACK)
			mark[q] = 0;
		return;
	}
	const uint32_t v = sval[q];
	bool fresh = q == 0 || skey[q - 1] != k || lsz[sval[q - 1]] != lsz[v];
	flag[q] = fresh ? 1 : 0;
	if (!BLACK)
		mark[q] = (tf[v] & TF_BLACK) ? q + 1 : 0;
}
__global__ void k_class_scatter(uint32_t n, const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval,
				const uint8_t *__restrict__ flag, const uint32_t *__restrict__ ps,
				uint32_t *__restrict__ gcls)
{
	uint32_t q = BIDX * blockDim.x + threadIdx.x;
	if (q >= n)
		return;
	const uint32_t cls = skey[q] == NIL ? NIL : ps[q] + flag[q] - 1; // inclusive scan - 1
	gcls[sval[q]] = cls; // (T-space = the per-component layout the debug hook reads)
}

// ------------------------------------------------------------- row E
// candidate-stack order = pre-order, except that under a branching entered side `a` the black
// child's subtree comes after the gray children's (tree_utils.cpp:47-76, flubbles.cpp:446-457).
// Only black edges enter the stack, and pre-order puts exactly one black vertex on every other position (segment slot
// g of component c <-> black vertex b = 2g + c + [dummy root] + 1, its parent a = b - 1): the whole permutation is
// worked out in units of black vertices, over the V segment slots.  subtree(b) holds (size + 1) / 2 black vertices,
// the gray children of a (size(a) - 1 - size(b)) / 2.
__global__ void k_shift_delta(uint32_t V, const uint32_t *__restrict__ seg_comp, const uint32_t *__restrict__ c_ntree,
			      const uint32_t *__restrict__ gsize, uint32_t *__restrict__ dlt)
{
	uint32_t g = BIDX * blockDim.x + threadIdx.x;
	if (g >= V)
		return;
	const uint32_t c = seg_comp[g], b = 2 * g + c + (c_ntree[c] & 1u) + 1, sb = gsize[b];
	if (sb == 0)
		return;
	const uint32_t sa = gsize[b - 1], gray = (sa - 1 - sb) / 2, black = (sb + 1) / 2;
	if (gray == 0)
		return;
	atomicAdd(&dlt[g], gray);		       // the black subtree moves behind the gray ones
	atomicAdd(&dlt[g + black], 0u - gray - black); // the gray subtrees move forward by the black one's entries
	atomicAdd(&dlt[g + sa / 2], black);
}
// the entry of every black edge, straight at its stack position (no flags, no compaction: a processed component's
// entries start at soff[c], a host-built table)
__global__ void k_stack_emit(uint32_t V, const uint32_t *__restrict__ seg_comp, const uint32_t *__restrict__ c_ntree,
			     const uint32_t *__restrict__ voff, const uint32_t *__restrict__ soff,
			     const uint32_t *__restrict__ dlt, const uint32_t *__restrict__ dlt_ps,
			     const uint32_t *__restrict__ gsize, const uint32_t *__restrict__ gcls,
			     uint32_t *__restrict__ s_vtx, uint32_t *__restrict__ s_cls, uint32_t *__restrict__ s_comp,
			     uint32_t *__restrict__ sidx, uint32_t *__restrict__ ns, uint32_t *__restrict__ prev)
{
	uint32_t g = BIDX * blockDim.x + threadIdx.x;
	if (g >= V)
		return;
	const uint32_t c = seg_comp[g], b = 2 * g + c + (c_ntree[c] & 1u) + 1;
	if (gsize[b] == 0)
		return;
	const uint32_t i = soff[c] + (g - voff[c]) + dlt_ps[g] + dlt[g];
	s_vtx[i] = b;
	s_cls[i] = gcls[b];
	s_comp[i] = c;
	sidx[b] = i;
	ns[i] = i; // "no later occurrence" until k_next_from_runs says otherwise (flubbles.cpp:391-399)
	prev[i] = NIL;
}
// ------------------------------------------------------------- row F
// next_seen without another sort.  The class stage already grouped the tree vertices by top bracket
// with every group ordered from the deepest vertex up, and a class is a run inside a group.  The
// members of a class lie on one root-to-leaf path, so their order in the candidate stack is their
// order by depth: for consecutive BLACK members (deeper d, shallower u) of a run, next_seen[u] = d.
__global__ void k_next_from_runs(uint32_t T, const uint32_t *__restrict__ mark, const uint32_t *__restrict__ lastb,
				 const uint32_t *__restrict__ sval, const uint32_t *__restrict__ gcls,
				 const uint32_t *__restrict__ sidx, uint32_t *__restrict__ ns, uint32_t *__restrict__ prev)
{
	uint32_t q = BIDX * blockDim.x + threadIdx.x;
	if (q >= T || !mark[q])
		return;
	const uint32_t p = lastb[q]; // 1 + position of the previous black entry of the sorted order
	if (!p)
		return;
	const uint32_t u = sval[q], d = sval[p - 1];
	if (gcls[u] != gcls[d])
		return;
	const uint32_t iu = sidx[u], id = sidx[d];
	ns[iu] = id;
	prev[id] = iu;
}

// BLACK pass, all in one: next_seen / prev and the "opens a flubble" flag of the entry at sorted position q, written at
// its stack index (the sort's payload).  Consecutive entries of a run are consecutive members of a class (deeper
// first), so next_seen[u] = the entry before it in the run, prev[u] = the entry after it.  Nothing downstream needs the
// classes as numbers; the debug hooks number them on demand (k_class_ids_black).
__global__ void k_class_finish_black(uint32_t n, uint32_t S, const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval,
				     const uint32_t *__restrict__ lsz, uint8_t *__restrict__ flag, uint32_t *__restrict__ ns,
				     uint32_t *__restrict__ prev, uint8_t *__restrict__ dflag)
{
	// Four sorted positions a lane (the sorted keys, the stack indices and the flags move in 16-byte words), and the
	// "opens a new class" flags (k_class_flags<true>: a bracket hands out a new class whenever the list size differs from
	// the one it saw last, flubbles.cpp:668-676) are worked out right here, for the four and the one behind them.
	const uint32_t q0 = (BIDX * blockDim.x + threadIdx.x) * 4u;
	if (q0 == 0)
		dflag[S] = 0;
	if (q0 >= n)
		return;
	uint32_t K[6], U[6], L[6]; // positions q0 - 1 .. q0 + 4
	if (q0 + 4 <= n) {
		const uint4 k4 = *reinterpret_cast<const uint4 *>(skey + q0), u4 = *reinterpret_cast<const uint4 *>(sval + q0);
		K[1] = k4.x, K[2] = k4.y, K[3] = k4.z, K[4] = k4.w;
		U[1] = u4.x, U[2] = u4.y, U[3] = u4.z, U[4] = u4.w;
	} else {
		for (uint32_t j = 1; j <= 4; j++) {
			K[j] = q0 + j - 1 < n ? skey[q0 + j - 1] : NIL;
			U[j] = q0 + j - 1 < n ? sval[q0 + j - 1] : 0u;
		}
	}
	K[0] = q0 ? skey[q0 - 1] : NIL, U[0] = q0 ? sval[q0 - 1] : 0u;
	K[5] = q0 + 4 < n ? skey[q0 + 4] : NIL, U[5] = q0 + 4 < n ? sval[q0 + 4] : 0u;
#pragma unroll
	for (int j = 0; j < 6; j++)
		L[j] = K[j] != NIL ? lsz[U[j]] : 0u;
	bool fresh[6];
	fresh[0] = true;
#pragma unroll
	for (int j = 1; j < 6; j++)
		fresh[j] = (q0 + j - 1 == 0) || K[j - 1] != K[j] || L[j - 1] != L[j];
	uint32_t fw = 0;
#pragma unroll
	for (int j = 1; j <= 4; j++) {
		const uint32_t q = q0 + j - 1;
		if (q >= n || K[j] == NIL)
			continue; // (invalid entries carry NIL and sit behind every valid key: flag 0)
		const uint32_t u = U[j];
		fw |= (fresh[j] ? 1u : 0u) << (8 * (j - 1));
		const uint32_t nx = fresh[j] ? u : U[j - 1];
		ns[u] = nx;
		prev[u] = (K[j + 1] != NIL && !fresh[j + 1]) ? U[j + 1] : NIL;
		dflag[u] = (u + 1 < nx) ? 1 : 0; // entry u opens a flubble iff its class comes back later than at the next entry
	}
	if (q0 + 4 <= n) {
		*reinterpret_cast<uint32_t *>(flag + q0) = fw;
	} else {
		for (uint32_t j = 0; q0 + j < n; j++)
			flag[q0 + j] = (uint8_t)((fw >> (8 * j)) & 0xFFu);
	}
}
__global__ void k_class_ids_black(uint32_t n, const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval,
				  const uint8_t *__restrict__ fresh, const uint32_t *__restrict__ ps, uint32_t *__restrict__ s_cls)
{
	uint32_t q = BIDX * blockDim.x + threadIdx.x;
	if (q < n && skey[q] != NIL)
		s_cls[sval[q]] = ps[q] + fresh[q] - 1; // inclusive scan - 1
}
// classes of the black tree vertices back in T-space (debug hook after a BLACK pass)
__global__ void k_cls_from_stack(uint32_t S, const uint32_t *__restrict__ s_vtx, const uint32_t *__restrict__ s_cls,
				 uint32_t *__restrict__ gcls)
{
	uint32_t i = BIDX * blockDim.x + threadIdx.x;
	if (i < S)
		gcls[s_vtx[i]] = s_cls[i];
}
// ------------------------------------------------------------- row G
// add_flubbles as a stack machine: at entry i, "class already open" pops through it and moves the
// PVST parent one level up (U, saturating at the root); a non-adjacent next occurrence emits a
// flubble and descends (D).  When the (prev, i) intervals are laminar the class is open iff it
// occurred before, so U/D are known per entry; a crossing pair sends the component to the
// sequential kernel instead.
// one kernel: the laminarity check of the (prev, i) intervals and the +-1 walk of the stack machine.
// Entry i takes two steps: U (-1 if its class was seen before) and then D (+1 if it opens a flubble).  The level of a
// new flubble is the walk's value after its D less the running minimum, and as D never goes down the minimum is
// always attained right after a U: the walk is kept at those points only, one word per entry, step(i) = D(i-1) + U(i).
// The walks of all components sit back to back in one array.  So that ONE unsegmented running minimum serves every
// component, the first step of component c also drops by B_c = 2 * (entries of the component before it) + 2: further
// than that component's walk can have climbed or fallen, so nothing in front of c ever is the minimum again (and the
// zero of component c is simply the walk's value at its first entry, whose own U is 0).
//
// CROSSING intervals (only possible when the literal hi_2 rule made the classes inexact, DESIGN.md section 4 "Row G").  The
// machine's U at entry i is "class(i) is on its stack", i.e. the entry pushed at p = prev[i] is still there: no entry k in
// (p, i) popped through a class that was pushed before p -- no k in (p, i) with prev[k] < p whose own class was open at k.
// With laminar intervals no such k exists and U = "has a previous occurrence".  Otherwise the entries whose interval holds
// a smaller prev are FLAGGED here (the range-min test), and k_resolve_crossings decides them, in stack order, one lane per
// component: U(i) = has-prev and not crossed(i), crossed(i) = some k in (p, i) with prev[k] < p that is not itself crossed
// (an unflagged entry with a previous occurrence is always open).  The U / D events are then the machine's own -- and they
// are all the PVST depends on --, so such a component needs no sequential redo (until round 4 it took one: 0.66 s per
// 10^6 segments).  tests/test_laminar_fuzz.py checks the rule against the machine on arbitrary label sequences.
__device__ __forceinline__ void laminar_check_one(uint32_t i, uint32_t p, const uint32_t *__restrict__ prev, const SegTree &segP,
						   uint8_t *__restrict__ xflag)
{
	if (p == NIL || p + 1 >= i)
		return;
	uint32_t lowest;
	if (i - p <= 9) { // a class that comes back within a few entries: its neighbours' words sit next to prev[i]
		lowest = NIL;
		for (uint32_t k = p + 1; k < i; k++)
			lowest = min(lowest, prev[k]);
	} else {
		lowest = seg_min(segP, p + 1, i);
	}
	if (lowest < p)
		xflag[i] = 1; // (cleared by the caller)
}
__global__ void k_laminar_walk(uint32_t S, const uint32_t *__restrict__ prev, const SegTree segP, bool check,
			       const uint32_t *__restrict__ s_comp, const uint32_t *__restrict__ soff, uint8_t *__restrict__ xflag,
			       const uint8_t *__restrict__ dflag, uint32_t *__restrict__ walk)
{
	// four entries a lane (16-byte loads and one 16-byte store: a kernel of a few loads per element is bound by the memory
	// instructions it issues)
	const uint32_t i0 = (BIDX * blockDim.x + threadIdx.x) * 4u;
	if (i0 >= S)
		return;
	if (i0 + 4 <= S) {
		const uint4 p4 = *reinterpret_cast<const uint4 *>(prev + i0), c4 = *reinterpret_cast<const uint4 *>(s_comp + i0);
		const uint32_t d4 = *reinterpret_cast<const uint32_t *>(dflag + i0); // D of the entries i0 .. i0 + 3
		const uint32_t cm = i0 ? s_comp[i0 - 1] : c4.x, dm = i0 ? dflag[i0 - 1] : 0u;
		const uint32_t ps[4] = {p4.x, p4.y, p4.z, p4.w}, cs[5] = {cm, c4.x, c4.y, c4.z, c4.w};
		const uint32_t ds[4] = {dm, d4 & 0xFFu, (d4 >> 8) & 0xFFu, (d4 >> 16) & 0xFFu}; // D of the entry before each
		uint32_t w[4];
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			const uint32_t i = i0 + j;
			uint32_t step = ps[j] != NIL ? 0xFFFFFFFFu : 0u; // U: -1
			if (i > 0) {
				step += ds[j]; // D of the entry before
				if (cs[j] != cs[j + 1])
					step -= 2 * (i - soff[cs[j]]) + 2;
			}
			w[j] = step;
		}
		*reinterpret_cast<uint4 *>(walk + i0) = make_uint4(w[0], w[1], w[2], w[3]);
		if (check) // (!check: the class stage was exact, the intervals are laminar by construction)
#pragma unroll
			for (uint32_t j = 0; j < 4; j++)
				laminar_check_one(i0 + j, ps[j], prev, segP, xflag);
		return;
	}
	for (uint32_t i = i0; i < S; i++) {
		const uint32_t p = prev[i];
		uint32_t step = p != NIL ? 0xFFFFFFFFu : 0u; // U: -1
		if (i > 0) {
			step += dflag[i - 1]; // D of the entry before
			const uint32_t cp = s_comp[i - 1];
			if (cp != s_comp[i])
				step -= 2 * (i - soff[cp]) + 2;
		}
		walk[i] = step;
		if (check)
			laminar_check_one(i, p, prev, segP, xflag);
	}
}
// the flagged entries of every component, in stack order (xlist is ascending): crossed ones get xflag 2 and their U undone
__global__ void k_resolve_crossings(uint32_t C, const uint32_t *__restrict__ soff, const uint32_t *__restrict__ n_x,
				    const uint32_t *__restrict__ xlist, const uint32_t *__restrict__ prev, const SegTree segP,
				    uint8_t *xflag, uint32_t *__restrict__ walk, uint32_t *__restrict__ n_crossed)
{
	const uint32_t nx = *n_x;
	for (uint32_t c = BIDX * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
		// this component's stretch of the list
		uint32_t lo = 0, hi = nx;
		const uint32_t b = soff[c], e = soff[c + 1];
		while (lo < hi) {
			const uint32_t mid = (lo + hi) >> 1;
			if (xlist[mid] < b)
				lo = mid + 1;
			else
				hi = mid;
		}
		uint32_t crossed = 0;
		for (uint32_t t = lo; t < nx; t++) {
			const uint32_t i = xlist[t];
			if (i >= e)
				break;
			const uint32_t p = prev[i];
			for (uint32_t k = p + 1;;) {
				k = seg_first_less(segP, k, i, p); // next entry inside (p, i) that reaches back beyond p
				if (k == NIL)
					break;
				if (xflag[k] != 2) { // it was open at k (prev[k] < p: it has a previous occurrence): it popped p's entry
					xflag[i] = 2;
					walk[i] += 1u; // the U of entry i does not happen
					crossed++;
					break;
				}
				k++;
			}
		}
		if (crossed)
			atomicAdd(n_crossed, crossed);
	}
}
// inclusive prefix sums of the walk, biased so that u32 order = int order; neg = their complement (a running
// maximum of the complement is the running minimum of the walk)
__global__ void k_walk_bias(uint32_t n, const uint32_t *walk, const uint32_t *ps, uint32_t *out, uint32_t *neg)
{
	const uint32_t k0 = (BIDX * blockDim.x + threadIdx.x) * 4u; // (in place on both arrays: a lane reads its four, then writes them)
	if (k0 >= n)
		return;
	if (k0 + 4 <= n) {
		const uint4 a = *reinterpret_cast<const uint4 *>(ps + k0), b = *reinterpret_cast<const uint4 *>(walk + k0);
		const uint4 w = make_uint4(a.x + b.x + 0x80000000u, a.y + b.y + 0x80000000u, a.z + b.z + 0x80000000u, a.w + b.w + 0x80000000u);
		*reinterpret_cast<uint4 *>(out + k0) = w;
		*reinterpret_cast<uint4 *>(neg + k0) = make_uint4(~w.x, ~w.y, ~w.z, ~w.w);
		return;
	}
	for (uint32_t k = k0; k < n; k++) {
		const uint32_t w = ps[k] + walk[k] + 0x80000000u;
		out[k] = w;
		neg[k] = ~w;
	}
}
// entry i opens a flubble iff its class comes back later than at the next entry (flubbles.cpp:344)
__global__ void k_dflag(uint32_t S, const uint32_t *__restrict__ ns, uint8_t *__restrict__ dflag)
{
	uint32_t i = BIDX * blockDim.x + threadIdx.x;
	if (i < S)
		dflag[i] = (i + 1 < ns[i]) ? 1 : 0;
	if (i == S)
		dflag[S] = 0;
}
// Endpoints and orientations of every flubble straight into the (page-locked host) PVST arrays.  They only depend on
// the candidate stack and next_seen, so this kernel runs on the context's side stream while the main stream still
// computes levels and parents: the PCIe writes (10 of the 14 bytes per PVST vertex) hide behind that work.  A small
// grid-stride launch: a few ten thousand lanes keep the link busy and leave the CUs to the main stream.
__global__ void __launch_bounds__(TPB) k_emit_endpoints(uint32_t S, const uint8_t *__restrict__ dflag, const uint32_t *__restrict__ erank,
							 const uint32_t *__restrict__ s_comp, const uint32_t *__restrict__ ns,
							 const uint32_t *__restrict__ s_vtx, const uint8_t *__restrict__ tf,
							 const uint32_t *__restrict__ t_gid, const uint32_t *__restrict__ cproc_ps,
							 uint32_t *__restrict__ p_a, uint32_t *__restrict__ p_z, uint8_t *__restrict__ p_aor,
							 uint8_t *__restrict__ p_zor)
{
	// Only one entry in four or five opens a flubble.  A wave reads the flags of 256 entries (four a lane, one load),
	// lists the ones that do in LDS, and then every lane takes one of THOSE: the ten gathers behind an emitted flubble run
	// with full waves instead of waves that are four fifths idle.
	__shared__ uint32_t list[TPB / 64][256];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	for (uint32_t b0 = blockIdx.x * (4u * TPB); b0 < S; b0 += gridDim.x * (4u * TPB)) { // (uniform per workgroup: the barriers below are safe)
		const uint32_t i0 = b0 + wave * 256u + lane * 4u;
		uint32_t fl = 0;
		if (i0 + 4 <= S) {
			fl = *reinterpret_cast<const uint32_t *>(dflag + i0);
		} else {
			for (uint32_t k = 0; i0 + k < S && k < 4; k++)
				fl |= (dflag[i0 + k] ? 1u : 0u) << (8 * k);
		}
		const uint32_t cnt = ((fl & 0xFFu) ? 1u : 0u) + ((fl & 0xFF00u) ? 1u : 0u) + ((fl & 0xFF0000u) ? 1u : 0u) + ((fl & 0xFF000000u) ? 1u : 0u);
		uint32_t inc = cnt;
		for (int off = 1; off < 64; off <<= 1) {
			const uint32_t y = __shfl_up(inc, off);
			if ((int)lane >= off)
				inc += y;
		}
		const uint32_t total = __shfl(inc, 63);
		uint32_t at = inc - cnt;
#pragma unroll
		for (uint32_t k = 0; k < 4; k++)
			if ((fl >> (8 * k)) & 0xFFu)
				list[wave][at++] = i0 + k;
		__syncthreads();
		for (uint32_t k = lane; k < total; k += 64) {
			const uint32_t i = list[wave][k];
			// dense output slot: flubbles emitted before + one root per earlier component, + 1 for this component's root
			const uint64_t q = (uint64_t)erank[i] + cproc_ps[s_comp[i]] + 1;
			uint32_t va = s_vtx[i], vz = s_vtx[ns[i]];
			uint32_t ra = ((tf[va] & TF_TYPE_MASK) == 1) ? 0u : 1u, rz = ((tf[vz] & TF_TYPE_MASK) == 1) ? 0u : 1u;
			if (ra && rz) { // normalize_endpoints, flubbles.cpp:233-244
				p_a[q] = t_gid[vz];
				p_z[q] = t_gid[va];
				p_aor[q] = 0;
				p_zor[q] = 0;
			} else {
				p_a[q] = t_gid[va];
				p_z[q] = t_gid[vz];
				p_aor[q] = (uint8_t)ra;
				p_zor[q] = (uint8_t)rz;
			}
		}
		__syncthreads();
	}
}
// level of every emitted flubble
__global__ void k_levels(uint32_t S, const uint8_t *__restrict__ dflag, const uint32_t *__restrict__ erank,
			 const uint32_t *__restrict__ s_comp, const uint32_t *__restrict__ soff,
			 const uint32_t *__restrict__ wb, const uint32_t *__restrict__ negmax,
			 uint32_t *__restrict__ lev, uint32_t *__restrict__ e_i)
{
	// four entries a lane: one load says which of them open a flubble (one in four or five does)
	const uint32_t i0 = (BIDX * blockDim.x + threadIdx.x) * 4u;
	if (i0 >= S)
		return;
	uint32_t fl = 0;
	if (i0 + 4 <= S) {
		fl = *reinterpret_cast<const uint32_t *>(dflag + i0);
	} else {
		for (uint32_t k = 0; i0 + k < S; k++)
			fl |= (dflag[i0 + k] ? 1u : 0u) << (8 * k);
	}
#pragma unroll
	for (uint32_t k = 0; k < 4; k++) {
		if (!((fl >> (8 * k)) & 0xFFu))
			continue;
		const uint32_t i = i0 + k, c = s_comp[i], f0 = soff[c];
		// zero of this component's walk = its value at the component's first entry (after the drop B_c, before any U or D)
		const uint32_t zero = wb[f0];
		// running minimum up to i (negmax = exclusive running maximum of ~wb): everything in front of f0 lies above `zero`
		const uint32_t cur = wb[i], run = min(cur, ~negmax[i]);
		const uint32_t j = erank[i];
		lev[j] = cur + 1 - min(zero, run); // depth of the new flubble (>= 1): the walk after this entry's own D
		e_i[j] = i;
	}
}
// PVST parent of every flubble = nearest earlier flubble of its component with a smaller level
__global__ void k_pvst_emit(uint32_t NE, const uint32_t *__restrict__ lev, const uint32_t *__restrict__ e_i,
			    const SegTree segL, const uint32_t *__restrict__ s_comp,
			    const uint32_t *__restrict__ soff, const uint32_t *__restrict__ erank,
			    const uint32_t *__restrict__ cproc_ps, uint32_t *__restrict__ p_parent)
{
	uint32_t j = BIDX * blockDim.x + threadIdx.x;
	if (j >= NE)
		return;
	uint32_t i = e_i[j], c = s_comp[i], jb = erank[soff[c]];
	uint32_t jp = seg_last_less(segL, jb, j, lev[j]);
	uint64_t pb = (uint64_t)jb + cproc_ps[c]; // dense: flubbles emitted before + one root per earlier component
	p_parent[pb + 1 + (j - jb)] = jp == NIL ? 0u : 1 + (jp - jb);
}
// per component: where its PVST starts in the dense output, its size, its stack entries -- everything the host needs for
// the forest's tree table, known as soon as the "opens a flubble" flags are ranked (before the result block exists)
__global__ void k_pvst_counts(uint32_t C, const uint32_t *__restrict__ c_ntree, const uint32_t *__restrict__ soff,
			      const uint32_t *__restrict__ erank, const uint32_t *__restrict__ cproc_ps, uint32_t *__restrict__ doff,
			      uint32_t *__restrict__ c_npvst, uint32_t *__restrict__ c_nstack)
{
	uint32_t c = BIDX * blockDim.x + threadIdx.x;
	if (c > C)
		return;
	doff[c] = erank[soff[c]] + cproc_ps[c];
	if (c == C)
		return;
	if (c_ntree[c] == 0) {
		c_npvst[c] = 0;
		return;
	}
	c_npvst[c] = 1 + (erank[soff[c + 1]] - erank[soff[c]]);
	c_nstack[c] = soff[c + 1] - soff[c];
}
// the root vertex of every PVST (flubbles.cpp:736-741)
__global__ void k_pvst_roots(uint32_t C, const uint32_t *__restrict__ c_ntree, const uint32_t *__restrict__ doff,
			     uint32_t *__restrict__ p_parent, uint32_t *__restrict__ p_a, uint32_t *__restrict__ p_z,
			     uint8_t *__restrict__ p_aor, uint8_t *__restrict__ p_zor)
{
	uint32_t c = BIDX * blockDim.x + threadIdx.x;
	if (c >= C || c_ntree[c] == 0)
		return;
	const uint64_t pb = doff[c];
	p_parent[pb] = NIL;
	p_a[pb] = p_z[pb] = NIL;
	p_aor[pb] = p_zor[pb] = 0;
}
// copies of the parallel results into the per-component layout the debug hooks read
__global__ void k_export_stack(uint32_t S, const uint32_t *__restrict__ s_comp, const uint32_t *__restrict__ soff,
			       const uint32_t *__restrict__ voff, const uint32_t *__restrict__ s_vtx,
			       const uint32_t *__restrict__ s_cls, const uint32_t *__restrict__ ns,
			       uint32_t *__restrict__ o_vtx, uint32_t *__restrict__ o_cls, uint32_t *__restrict__ o_ns)
{
	uint32_t i = BIDX * blockDim.x + threadIdx.x;
	if (i >= S)
		return;
	uint32_t c = s_comp[i], l = i - soff[c], base = 2 * voff[c] + c;
	o_vtx[voff[c] + l] = s_vtx[i] - base;
	o_cls[voff[c] + l] = s_cls[i];
	o_ns[voff[c] + l] = ns[i] - soff[c];
}
// ------------------------------------------------------------- hairpin boundaries
// The reverse pre-order sweep opens a hairpin at every simplifying vertex (b1 = its segment), extends
// b2 while the top bracket is a simplifying edge, and closes it at the next leaf (or the root)
// (flubbles.cpp:531-535, 621-656).  Per closing vertex c the window of vertices processed since the
// previous closer is (c, prev closer]; everything is a "nearest flagged vertex" query.
__global__ void k_hp_inputs(uint32_t T, const uint32_t *__restrict__ gsize, const uint32_t *__restrict__ gpar,
			    const uint8_t *__restrict__ hpf, uint32_t *__restrict__ a_simp, uint32_t *__restrict__ a_q,
			    uint32_t *__restrict__ a_close)
{
	uint32_t v = BIDX * blockDim.x + threadIdx.x;
	if (v >= T)
		return;
	const uint32_t sz = gsize[v];
	const uint8_t f = sz ? hpf[v] : 0;
	const bool root = sz && gpar[v] == NIL;
	a_simp[v] = (sz && !root && (f & 1)) ? 0u : 1u;
	a_q[v] = (sz && !root && !(f & 1) && (f & 2)) ? 0u : 1u;
	a_close[v] = (sz && (root || sz == 1)) ? 0u : 1u;
}
__global__ void k_hp_close(uint32_t T, const uint32_t *__restrict__ gsize, const uint32_t *__restrict__ t_root,
			   const uint32_t *__restrict__ t_comp, const uint32_t *__restrict__ c_ntree,
			   const uint32_t *__restrict__ t_gid, const uint32_t *__restrict__ a_close,
			   const SegTree s1, const SegTree s2, const SegTree s3, uint8_t *__restrict__ push,
			   unsigned long long *__restrict__ b12)
{
	uint32_t c = BIDX * blockDim.x + threadIdx.x;
	if (c >= T)
		return;
	push[c] = 0;
	if (a_close[c])
		return;
	const uint32_t base = t_root[c], end = base + c_ntree[t_comp[c]];
	// previous closer in processing order = nearest closer with a larger idx
	const uint32_t cp = seg_first_less(s3, c + 1, end, 1u);
	const uint32_t hi = cp == NIL ? end : cp + 1; // window (c, hi)
	const uint32_t smin = seg_first_less(s1, c + 1, hi, 1u);
	if (smin == NIL)
		return; // no simplifying vertex since the previous closer: not in a hairpin
	const uint32_t smax = seg_last_less(s1, c + 1, hi, 1u);
	const uint32_t q = seg_first_less(s2, c + 1, smax, 1u);
	push[c] = 1;
	b12[2 * (size_t)c] = t_gid[smin];
	b12[2 * (size_t)c + 1] = q == NIL ? (unsigned long long)NIL : (unsigned long long)t_gid[q];
	(void)gsize;
}
// boundaries are reported in processing order: closers by descending idx inside a component
__global__ void k_hp_emit(uint32_t T, const uint8_t *__restrict__ push, const uint32_t *__restrict__ ps,
			  const uint32_t *__restrict__ t_root, const uint32_t *__restrict__ t_comp,
			  const uint32_t *__restrict__ c_ntree, const uint32_t *__restrict__ voff,
			  const unsigned long long *__restrict__ b12, unsigned long long *__restrict__ out,
			  uint32_t *__restrict__ c_nbry)
{
	uint32_t c = BIDX * blockDim.x + threadIdx.x;
	if (c >= T)
		return;
	const uint32_t ci = t_comp[c];
	if (ci == NIL)
		return;
	const uint32_t base = t_root[c], end = base + c_ntree[ci];
	if (c == base && c_ntree[ci])
		c_nbry[ci] = ps[end] - ps[base];
	if (!push[c])
		return;
	const uint32_t after = ps[end] - ps[c + 1]; // pushes of this component with a larger idx come first
	const uint64_t pb = (uint64_t)voff[ci] + ci;
	out[2 * (pb + after)] = b12[2 * (size_t)c];
	out[2 * (pb + after) + 1] = b12[2 * (size_t)c + 1];
}

void run_parallel_hairpins(const CompState &cs, SeqWs &sw, ParWs &pw, uint32_t C, StageTimer &tm, hipStream_t s)
{
	const uint32_t T = 2 * sw.V + C;
	tm.begin("par_hairpins");
	LAUNCH(k_hp_inputs, T, s, T, pw.gsize, pw.gpar, pw.hpf, pw.hp1, pw.hp2, pw.hp3);
	seg_build(pw.segH1, pw.hp1, T, s);
	seg_build(pw.segH2, pw.hp2, T, s);
	seg_build(pw.segH3, pw.hp3, T, s);
	uint8_t *push = pw.f8a;
	uint32_t *pps = pw.psA;
	unsigned long long *b12 = pw.hp_b12; // [2T] boundary pairs
	if (!b12 || !pw.hpf || !pw.t_comp)
		throw HipError("hairpin report: its arrays were not carved (internal)");
	LAUNCH(k_hp_close, T, s, T, pw.gsize, pw.t_root, pw.t_comp, sw.c_ntree, sw.t_gid, pw.hp3, pw.segH1, pw.segH2, pw.segH3, push,
	       b12);
	scan_exclusive_u8(push, pps, (size_t)T + 1, nullptr, nullptr, 0, pw.scan_tmp, pw.scan_tmp_bytes, s);
	LAUNCH(k_hp_emit, T, s, T, push, pps, pw.t_root, pw.t_comp, sw.c_ntree, cs.voff, b12, (unsigned long long *)sw.hairpins,
	       sw.c_nbry);
	tm.end(10);
}

// ------------------------------------------------------------- workspace
template <typename F>
static void for_each_span(ParWs &pw, size_t V, size_t E, size_t Cmax, int groups, const StageWsOpts &o, F &&take_any)
{
	// Group 1: what the TREE stage already writes (its tree arrays in the layout the class stage reads, the back edges, the
	// scratch of the primitives).  Group 2: everything first written by the class stage and later -- by then the tree
	// stage's own workspace is dead but for a few arrays (tree_spans, group 1), and group 2 lies over the rest of it
	// (stage_workspace_carve).
	auto take1 = [&](void **p, size_t bytes) {
		if (groups & 1)
			take_any(p, bytes);
	};
	auto take = [&](void **p, size_t bytes) {
		if (groups & 2)
			take_any(p, bytes);
	};
	// brackets: the back edges of from_bd (links outside the tree, one per side without links) and, per tree vertex, at most
	// ONE edge of the class stage -- a capping edge needs a non-empty bracket list, a simplifying edge an empty one
	const size_t T = 2 * V + Cmax, NB = o.nb_cap ? std::min(o.nb_cap, E + V + T) : E + V + T, S = V + 1;
	pw.nb_cap = NB;
	auto skip = [](auto **p) { *p = nullptr; };
	for (uint32_t **p : {&pw.hi0, &pw.mpre, &pw.dlt, &pw.incnt, &pw.lsz})
		take1((void **)p, (T + 2) * 4);
	for (uint32_t **p : {&pw.b_src, &pw.b_tgt, &pw.b_ord})
		take1((void **)p, (NB + 2) * 4);
	take1((void **)&pw.sdl, (V + 4) * 4);
	take1((void **)&pw.err, 64);
	take1((void **)&pw.comp_bad, (Cmax + 2) * 4); // (zeroed when the pass starts, ahead of the tree stage)
	pw.scan_tmp_bytes = std::max(scan_tmp_bytes(std::max(T, NB) + 4), 2 * compact_tmp_bytes(std::max(T, NB) + 4));
	pw.sort_tmp_bytes = sort_tmp_bytes(std::max(std::max(T, NB), std::max(4 * V, 2 * E) + 8) + 4);
	take1(&pw.scan_tmp, pw.scan_tmp_bytes);
	take1(&pw.sort_tmp, pw.sort_tmp_bytes);
	for (uint32_t **p : {&pw.t_comp, &pw.t_root, &pw.gpar, &pw.gsize, &pw.cov}) {
		if (o.full_t)
			take((void **)p, (T + 2) * 4);
		else if (groups & 2)
			skip(p);
	}
	for (uint32_t **p : {&pw.psA, &pw.psB, &pw.flagC, &pw.psC, &pw.cap_tgt, &pw.dlt_ps, &pw.psin, &pw.topi, &pw.gcls, &pw.vals_t,
			     &pw.vals_t2})
		take((void **)p, (T + 2) * 4);
	for (uint8_t **p : {&pw.f8a, &pw.f8b, &pw.f8c, &pw.f8d})
		take((void **)p, T + 32);
	take((void **)&pw.keys_t, (T + 2) * 4);
	take((void **)&pw.keys_t2, (T + 2) * 4);
	take((void **)&pw.dbo, (Cmax + 2) * 4);
	for (uint32_t **p : {&pw.b_val2, &pw.tgtR})
		take((void **)p, (NB + 2) * 4);
	for (uint32_t **p : {&pw.b_val, &pw.b_key, &pw.b_key2}) { // (the bracket sort: only behind a sequential tree stage)
		if (o.sorted_brackets)
			take((void **)p, (NB + 2) * 4);
		else if (groups & 2)
			skip(p);
	}
	for (uint32_t **p : {&pw.s_vtx, &pw.s_cls, &pw.s_comp, &pw.ns, &pw.prev, &pw.erank, &pw.lev, &pw.e_i})
		take((void **)p, (S + 2) * 4);
	take((void **)&pw.walk, (S + 4) * 4);
	take((void **)&pw.walk_ps, (S + 4) * 4);
	take((void **)&pw.wrun, (S + 4) * 4);
	take((void **)&pw.cproc_ps, (Cmax + 2) * 4);
	take((void **)&pw.doff, (Cmax + 2) * 4);
	take((void **)&pw.segA.tree, SegTree::tree_words(T + 1) * 4);
	take((void **)&pw.segB.tree, SegTree::tree_words(NB + 1) * 4);
	take((void **)&pw.segP.tree, SegTree::tree_words(S + 1) * 4);
	take((void **)&pw.segL.tree, SegTree::tree_words(S + 1) * 4);
	take((void **)&pw.stage, ((S + Cmax + 2) * 4 + 64) * 3 + ((S + Cmax + 2) + 64) * 2 + 256);
	if (o.hairpins) {
		take((void **)&pw.hp_b12, (2 * T + 4) * 8);
		take((void **)&pw.hpf, T + 2);
		for (uint32_t **p : {&pw.hp1, &pw.hp2, &pw.hp3})
			take((void **)p, (T + 2) * 4);
		take((void **)&pw.segH1.tree, SegTree::tree_words(T + 1) * 4);
		take((void **)&pw.segH2.tree, SegTree::tree_words(T + 1) * 4);
		take((void **)&pw.segH3.tree, SegTree::tree_words(T + 1) * 4);
	} else if (groups & 2) {
		pw.hpf = nullptr;
		pw.hp_b12 = nullptr;
		pw.hp1 = pw.hp2 = pw.hp3 = nullptr;
		pw.segH1.tree = pw.segH2.tree = pw.segH3.tree = nullptr;
	}
}

size_t par_workspace_bytes(size_t V, size_t E, size_t Cmax, int groups, const StageWsOpts &o)
{
	ParWs tmp{};
	size_t total = 0;
	for_each_span(tmp, V, E, Cmax, groups, o, [&](void **, size_t bytes) { total += ((bytes + 255) & ~size_t(255)) + 256; });
	return total + (1 << 20);
}

void par_carve(Arena &ar, ParWs &pw, size_t V, size_t E, size_t Cmax, int groups, const StageWsOpts &o)
{
	for_each_span(pw, V, E, Cmax, groups, o, [&](void **dst, size_t bytes) { *dst = ar.take<char>(bytes); });
}

// ------------------------------------------------------------- driver
// Everything the host needs to know about a finished pass, written straight into page-locked host
// memory: [0..2] error words, then bad[C], status[C], npvst[C], nbry[C], doff[C+1].
__global__ void k_summary(uint32_t C, const uint32_t *__restrict__ err, const uint32_t *__restrict__ bad,
			  const uint32_t *__restrict__ status, const uint32_t *__restrict__ npvst,
			  const uint32_t *__restrict__ nbry, const uint32_t *__restrict__ doff, uint32_t *__restrict__ out)
{
	uint32_t i = BIDX * blockDim.x + threadIdx.x;
	if (i > C)
		return;
	if (i == 0) {
		out[0] = err ? err[0] : 0;
		out[1] = err ? (err[1] | (err[8] << 1)) : 0; // bit 0: list ranking, bit 1: stack pool of the class walk
		out[2] = err ? err[2] : 0;
		out[3] = err ? err[3] : 0;
	}
	uint32_t *o = out + 4;
	if (i < C) {
		o[i] = bad ? bad[i] : 0;
		o[(size_t)C + i] = status[i];
		o[2 * (size_t)C + i] = npvst[i];
		o[3 * (size_t)C + i] = nbry[i];
	}
	o[4 * (size_t)C + i] = doff ? doff[i] : 0;
}

void pass_summary(const SeqWs &sw, const ParWs *pw, uint32_t C, uint32_t *host_out, hipStream_t s)
{
	uint32_t *dev_out = nullptr;
	HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&dev_out), host_out, 0));
	LAUNCH(k_summary, (size_t)C + 1, s, C, pw ? pw->err : nullptr, pw ? pw->comp_bad : nullptr, sw.c_status, sw.c_npvst,
	       sw.c_nbry, pw ? pw->doff : nullptr, dev_out);
}

void run_parallel_dg(const CompState &cs, SeqWs &sw, ParWs &pw, uint32_t C, uint32_t n_processed, uint32_t n_stack,
		     int64_t dense_nb0, const std::function<void *(size_t)> &alloc_result_block, StageTimer &tm, hipStream_t s,
		     const SideStream &side, PassTail &tail)
{
	const uint32_t V = sw.V, T = 2 * V + C;
	const bool want_hp = sw.hairpins != nullptr;
	pw.V = V;
	pw.E = sw.E;
	pw.C = C;
	pw.T = T;
	auto scan = [&](const uint32_t *in, uint32_t *out, size_t n) {
		scan_exclusive_u32(in, out, n, pw.scan_tmp, pw.scan_tmp_bytes, s);
	};
	auto scan8 = [&](const uint8_t *in, uint32_t *out, size_t n) {
		scan_exclusive_u8(in, out, n, nullptr, nullptr, 0, pw.scan_tmp, pw.scan_tmp_bytes, s);
	};
	auto scan2 = [&](const uint32_t *in0, uint32_t *out0, size_t n0, const uint32_t *in1, uint32_t *out1, size_t n1) {
		scan_exclusive_u32_pair(in0, out0, n0, in1, out1, n1, pw.scan_tmp, pw.scan_tmp_bytes, s);
	};

	// ---- T-space + dense back edges
	tm.begin("par_setup");
	// All-parallel pass: the tree stage wrote sizes (0 = no vertex in this slot), parents (only ever tested against NIL), the
	// mirror pre-order and the cleared bracket counts in T-space itself; the root of a vertex's tree is looked up where it
	// is needed.  A sequential tree stage works per component: its arrays are brought into that form here, and the
	// hairpin report wants the per-vertex tables too.
	const bool lean = dense_nb0 >= 0 && !want_hp;
	if (lean) {
		pw.gsize = sw.t_size;
		pw.gpar = sw.t_par;
	} else {
		LAUNCH(k_tcomp_vertices, V, s, V, T, cs.ckey, cs.voff, pw.t_comp);
		LAUNCH(k_globalize, T, s, T, pw.t_comp, cs.voff, sw.c_ntree, sw.t_par, sw.t_size, pw.gpar, pw.gsize, pw.t_root, pw.hi0,
		       dense_nb0 >= 0 ? nullptr : pw.cov, sw.t_depth, pw.mpre, pw.incnt, pw.dlt);
	}
	const RootOf root_of{lean ? nullptr : pw.t_root, cs.voff, C};
	// pw.comp_bad and pw.err are zeroed by the caller (zero_component_counters)
	uint32_t NB0;
	if (dense_nb0 >= 0) { // the parallel tree stage already left the back edges in b_src / b_tgt
		NB0 = dense_nb0 == NB0_ON_DEVICE ? 0u : (uint32_t)dense_nb0; // (on the device: read below, nothing before that needs it)
	} else {
		scan(sw.c_nbe0, pw.dbo, (size_t)C + 1);
		NB0 = pw.host->read_u32(pw.dbo + C, s);
		LAUNCH(k_dense_be, NB0, s, NB0, C, pw.dbo, cs.voff, cs.eoff, sw.be_src, sw.be_tgt, pw.b_src, pw.b_tgt);
	}
	tm.end(7);

	// ---- row D
	tm.begin("par_classes");
	if (dense_nb0 < 0)
		LAUNCH(k_hi0, NB0, s, NB0, pw.b_src, pw.b_tgt, pw.hi0, pw.cov, nullptr);
	seg_build(pw.segA, pw.hi0, T, s);
	uint8_t *simp = pw.f8b, *capf = pw.f8c; // [T+1] flags, one byte each
	uint32_t *pssimp = pw.psB, *pscap = pw.psC;
	// bridge flags: a bit-rank directory (common.hpp) over T + 1 positions in psA's space: (T / 64 + 2) records of 16 bytes
	// = T / 4 + 32 bytes, the counts its build scans in f8a's ((T / 64 + 2) words)
	uint4 *brec = reinterpret_cast<uint4 *>(pw.psA);
	uint32_t *brec_cnt = reinterpret_cast<uint32_t *>(pw.f8a);
	uint32_t *pscov = pw.psB; // (free until the simplifying flags are scanned)
	if (dense_nb0 >= 0) // back edges leaving a vertex (counted by the tree stage) minus those arriving (counted by k_hi0)
		scan_exclusive_diff_u32(pw.lsz, pw.incnt, pscov, (size_t)T + 1, pw.scan_tmp, pw.scan_tmp_bytes, s);
	else
		scan(pw.cov, pscov, (size_t)T + 1);
	KLAUNCH(k_bridge_flags, dim3(nblk(((size_t)T / 256 + 1) * 64)), dim3(TPB), 0, s, T, pw.gsize, pw.gpar, pscov, brec); // (whole waves: every record of [0, T] is written)
	bitrank_build(brec, (size_t)T / 64 + 1, brec_cnt, pw.scan_tmp, pw.scan_tmp_bytes, s);
	uint32_t *br_list = pw.vals_t, *n_br = pw.err + 10; // (the sort's value buffer is free until the class pass; the count was cleared with the other counters)
	KLAUNCH(k_hi_simp, dim3((unsigned)(((size_t)T + HS_VERTS - 1) / HS_VERTS)), dim3(TPB), 0, s, T, pw.gsize, brec, simp, want_hp ? pw.hpf : nullptr, capf, br_list,
		n_br);
	KLAUNCH(k_capping, dim3(std::min<unsigned>(nblk(T / 8 + 1), 16384)), dim3(TPB), 0, s, n_br, br_list, pw.gsize, pw.hi0, brec, root_of,
		pw.segA, pw.cap_tgt, capf, pw.err + 5);
	// capping / simplifying vertices per tile of k_bracket_extra, scanned: [ntiles + 1] each, the totals in the last word
	const uint32_t ntiles = (T + BX_TILE - 1) / BX_TILE;
	uint32_t *tsimp = pssimp, *tcap = pscap;
	KLAUNCH(k_flag_tile_counts, dim3((ntiles + 1 + 15) / 16 + 1), dim3(TPB), 0, s, T, capf, simp, tcap, tsimp, ntiles);
	scan_exclusive_u32_pair(tsimp, tsimp, (size_t)ntiles + 1, tcap, tcap, (size_t)ntiles + 1, pw.scan_tmp, pw.scan_tmp_bytes, s);
	uint32_t *srccnt = pw.dlt, *bstart = pw.dlt_ps; // free until row E
	uint32_t *extra = pw.host->take<uint32_t>(4);
	publish_words(extra, WordSrc{{tcap + ntiles, tsimp + ntiles, pw.err + 5, pw.err + 6}}, 4, s); // counts | literal-rule flag | back edges of the tree stage
	// The stream is not left idle while the host reads them (HostScratch::mark): the kernel that places the capping and
	// simplifying brackets takes the counts from the device, and the scan behind it does not need them at all.
	pw.host->mark(s);
	LAUNCH(k_bracket_extra, ((size_t)T + 3) / 4, s, T, NB0, dense_nb0 == NB0_ON_DEVICE ? pw.err + 6 : nullptr, ntiles,
	       (uint32_t)std::min<size_t>(pw.nb_cap, 0xFFFFFFFFu), capf, tcap, simp, tsimp, pw.cap_tgt, root_of, pw.b_src, pw.b_tgt,
	       dense_nb0 >= 0 ? pw.lsz : nullptr, pw.gsize, pw.mpre, pw.incnt, srccnt);
	if (dense_nb0 >= 0) // ranks inside every source and the counts per source are known: place directly
		scan2(pw.incnt, pw.psin, (size_t)T + 1, srccnt, bstart, (size_t)T + 1);
	pw.host->wait();
	if (dense_nb0 == NB0_ON_DEVICE) {
		NB0 = extra[3];
		if (NB0 > pw.nb_cap)
			throw HipError("spanning tree: more back edges than links outside the tree and sides without links (internal sizing bug)");
	}
	const uint32_t ncap = extra[0], nsimp = extra[1], NB = NB0 + ncap + nsimp;
	pw.nb0 = NB0;
	pw.ncap = ncap;
	pw.nsimp = nsimp;
	if ((size_t)NB > pw.nb_cap) // (the bracket arrays are carved for that many entries, see for_each_span)
		throw HipError("class stage: more brackets than the workspace was sized for (internal sizing bug)");
	if (dense_nb0 < 0 && !pw.b_key)
		throw HipError("class stage: the bracket sort's buffers were not carved for a sequential tree stage (internal)");
	if (!lean && !pw.t_comp)
		throw HipError("class stage: the per-vertex tables were not carved (internal)");
	// Classes of the black tree edges only (the candidate stack holds no others: half the vertices to look up, sort and
	// number).  Per top bracket the reference walks ALL vertices that have it on top, deepest first, and opens a class
	// whenever the list size differs from the one before (recent_size, flubbles.cpp:668-676); leaving the gray edges'
	// vertices out of that walk gives the same classes on the black ones as long as the sizes seen by one bracket never
	// grow again on the way up -- which is what capping edges are for (a bracket only resurfaces once every list spliced
	// under it has ended).  The literal hi_2 rule (flubbles.cpp:566-574) can cap too low; k_capping reports when it
	// picked another target than the second-highest reach, and hairpin reports want the top bracket of every vertex:
	// both take the all-vertices pass.
	const bool black_only = !want_hp && !pw.all_vertex_classes && extra[2] == 0;
	pw.black_only_used = black_only;
	if (dense_nb0 >= 0) {
		LAUNCH(k_bracket_place, NB, s, NB, NB0, ncap, pw.b_src, pw.b_tgt, pw.b_ord, pw.mpre, bstart, capf, simp, pw.tgtR,
		       want_hp ? pw.b_val2 : nullptr);
	} else {
		uint32_t *bk = pw.b_key, *bk2 = pw.b_key2;
		LAUNCH(k_bracket_order, NB, s, NB, NB0, ncap, nsimp, pw.b_src, pw.b_tgt, pw.mpre, bk, pw.b_val, pw.incnt, srccnt);
		scan2(pw.incnt, pw.psin, (size_t)T + 1, srccnt, bstart, (size_t)T + 1);
		sort_pairs_u32(bk, bk2, pw.b_val, pw.b_val2, NB, bits_for(T), pw.sort_tmp, pw.sort_tmp_bytes, s);
		LAUNCH(k_gather_u32, NB, s, NB, pw.b_val2, pw.b_tgt, pw.tgtR);
	}
	seg_build(pw.segB, pw.tgtR, NB, s);
	uint32_t *ck = pw.keys_t, *ck2 = pw.keys_t2;
	const uint32_t NC = black_only ? V : T; // vertices that get a class
	const uint32_t S = n_stack;
	uint8_t *cflag = pw.f8a; // bridge flags are dead by now
	uint32_t *cps = pw.psA;
	uint8_t *dflag = pw.f8c; // [S+1] <= [T+1]; capping flags are dead by now
	if (black_only) {
		// row E first (it only needs the tree): the stack index of every black vertex, so that the class pass can work
		// in stack order from the start -- the sort carries stack indices, and classes, next_seen and prev are written
		// where rows F/G read them
		uint32_t *sdl = pw.dlt;
		if (pw.sdl_filled && dense_nb0 >= 0) {
			sdl = pw.sdl; // the tree stage's emit kernel had both sizes of every segment in hand and left the differences there
		} else {
			HIP_CHECK(hipMemsetAsync(pw.dlt, 0, ((size_t)V + 2) * 4, s)); // (the bracket counts that lived here are dead)
			LAUNCH(k_shift_delta, V, s, V, cs.ckey, sw.c_ntree, pw.gsize, pw.dlt);
		}
		uint32_t *shift_ps = pw.topi; // (dlt_ps still holds the bracket range starts; nobody needs vertex -> stack index here)
		scan(sdl, shift_ps, (size_t)V + 1);
		const StackPlace sp{cs.voff, pw.soff, sdl, shift_ps, pw.s_vtx, pw.s_comp};
		LAUNCH(k_top_bracket<true>, NC, s, NC, pw.gsize, pw.gpar, pw.mpre, bstart, pw.segB, pw.tgtR, pw.psin, ck,
		       pw.vals_t, pw.lsz, pw.err, pw.b_val2, NB0 + ncap, nullptr, cs.ckey, sw.c_ntree, sp);
		sort_pairs_u32(ck, ck2, pw.vals_t, pw.vals_t2, NC, bits_for((uint64_t)NB + 1), pw.sort_tmp, pw.sort_tmp_bytes, s);
		// invalid entries carry NIL; after the sort on the low bits they sit behind every valid key
		tm.end(30 + 2 * 22);
		tm.begin("par_stack");
		// (the class flags are worked out by the same kernel: invalid entries carry NIL and sort behind every valid key)
		LAUNCH(k_class_finish_black, std::max<size_t>(((size_t)NC + 3) / 4, 1), s, NC, S, ck2, pw.vals_t2, pw.lsz, cflag, pw.ns, pw.prev,
		       dflag);
		tm.end(1);
		tm.begin("par_next_seen"); // (folded into the kernel above)
		tm.end(0);
		pw.gcls_valid = false;
		pw.s_cls_valid = false; // (cflag, the sorted keys and their stack indices stay where they are for stack_class_ids)
	} else {
		const StackPlace none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
		LAUNCH(k_top_bracket<false>, NC, s, NC, pw.gsize, pw.gpar, pw.mpre, bstart, pw.segB, pw.tgtR, pw.psin, ck,
		       pw.vals_t, pw.lsz, pw.err, pw.b_val2, NB0 + ncap, want_hp ? pw.hpf : nullptr, nullptr, nullptr, none);
		sort_pairs_u32(ck, ck2, pw.vals_t, pw.vals_t2, NC, bits_for((uint64_t)NB + 1), pw.sort_tmp, pw.sort_tmp_bytes, s);
		LAUNCH(k_class_flags<false>, std::max<size_t>(NC, (size_t)V + 2), s, NC, V, ck2, pw.vals_t2, pw.lsz, sw.t_flags, cflag, pw.dlt,
		       pw.flagC);
		scan8(cflag, cps, (size_t)NC + 1);
		LAUNCH(k_class_scatter, NC, s, NC, ck2, pw.vals_t2, cflag, cps, pw.gcls);
		pw.gcls_valid = true;
		pw.s_cls_valid = true;
		tm.end(30 + 2 * 22);

		// ---- row E
		tm.begin("par_stack");
		LAUNCH(k_shift_delta, V, s, V, cs.ckey, sw.c_ntree, pw.gsize, pw.dlt);
		scan(pw.dlt, pw.dlt_ps, (size_t)V + 1);
		// one candidate-stack entry per black tree edge = per segment of a processed component: the host knows where
		// every component's entries start (pw.soff) and the total
		LAUNCH(k_stack_emit, V, s, V, cs.ckey, sw.c_ntree, cs.voff, pw.soff, pw.dlt, pw.dlt_ps, pw.gsize, pw.gcls, pw.s_vtx, pw.s_cls,
		       pw.s_comp, pw.topi, pw.ns, pw.prev);
		tm.end(9);

		// ---- row F
		tm.begin("par_next_seen");
		uint32_t *mark = pw.flagC, *lastb = pw.psC; // (the marks were written with the class flags)
		scan_exclusive_max_u32(mark, lastb, T, pw.scan_tmp, pw.scan_tmp_bytes, s);
		LAUNCH(k_next_from_runs, T, s, T, mark, lastb, pw.vals_t2, pw.gcls, pw.topi, pw.ns, pw.prev);
		tm.end(4);
		LAUNCH(k_dflag, (size_t)S + 1, s, S, pw.ns, dflag);
	}

	// ---- row G
	tm.begin("par_pvst");
	scan8(dflag, pw.erank, (size_t)S + 1);
	// What the host needs to lay out the forest -- PVST size and offset of every component, the error words -- goes back in
	// ONE read together with the PVST count: after this synchronisation only the PVST arrays themselves are still to come.
	LAUNCH(k_pvst_counts, (size_t)C + 1, s, C, sw.c_ntree, pw.soff, pw.erank, pw.cproc_ps, pw.doff, sw.c_npvst, sw.c_nstack);
	uint32_t *early = pw.host->take<uint32_t>(5 * (size_t)C + 8);
	count_kernel_d2h((5 * (size_t)C + 8) * 4);
	pass_summary(sw, &pw, C, early, s);
	const bool use_side = S && side.stream;
	if (use_side) // (everything the endpoints need is computed: the side stream starts from here)
		HIP_CHECK(hipEventRecord(side.fork, s));
	// The stream is not left idle while the host reads the PVST count (HostScratch::mark): levels and parents of the
	// flubbles need the stack only, their kernels go out first.
	pw.host->mark(s);
	// The (prev, i) intervals of exact cycle-equivalence classes never cross ("Row G"): the laminarity
	// check is only needed when the literal hi_2 rule capped differently from the second-highest reach (extra[2]), i.e. when
	// the classes may not be the exact ones -- or when a caller asks for it.  Without it nothing that follows can flag a
	// component: the summary the host is about to read is the final one.
	pw.laminar_checked = pw.check_laminar || extra[2] != 0;
	uint8_t *xflag = pw.f8d; // [T + 32] >= S (the branching flags are long compacted)
	if (pw.laminar_checked) {
		seg_build(pw.segP, pw.prev, S, s); // NIL (= +inf) where a class has no earlier occurrence
		HIP_CHECK(hipMemsetAsync(xflag, 0, (size_t)S + 1, s));
	}
	LAUNCH(k_laminar_walk, ((size_t)S + 3) / 4, s, S, pw.prev, pw.segP, pw.laminar_checked, pw.s_comp, pw.soff, xflag, dflag, pw.walk);
	if (pw.laminar_checked && S) { // the entries whose interval is crossed: decided in stack order, U undone where the class was popped
		uint32_t *xlist = pw.wrun, *n_x = pw.err + 11, *n_crossed = pw.err + 12; // (wrun is written by the max-scan below; err words cleared at the start of the pass)
		compact_flagged_u8(xflag, S, xlist, n_x, pw.scan_tmp, pw.scan_tmp_bytes, s);
		KLAUNCH(k_resolve_crossings, dim3(std::min<unsigned>(nblk(C), 1024)), dim3(TPB), 0, s, C, pw.soff, n_x, xlist, pw.prev, pw.segP, xflag,
			pw.walk, n_crossed);
	}
	scan(pw.walk, pw.walk_ps, (size_t)S);
	uint32_t *wb = pw.walk_ps; // in place: exclusive -> biased inclusive
	uint32_t *wneg = pw.walk, *wrun = pw.wrun; // the steps themselves are dead after the bias kernel read them
	LAUNCH(k_walk_bias, ((size_t)S + 3) / 4, s, S, pw.walk, pw.walk_ps, wb, wneg);
	scan_exclusive_max_u32(wneg, wrun, (size_t)S, pw.scan_tmp, pw.scan_tmp_bytes, s);
	LAUNCH(k_levels, ((size_t)S + 3) / 4, s, S, dflag, pw.erank, pw.s_comp, pw.soff, wb, wrun, pw.lev, pw.e_i);
	pw.host->wait();
	const size_t total = early[4 + 4 * (size_t)C + C]; // doff[C] = flubbles + one root per processed component
	if (total < n_processed || total - n_processed > S)
		throw HipError("internal error: PVST size out of range");
	const uint32_t NE = (uint32_t)(total - n_processed);
	pw.n_emitted = NE;
	tail.early_summary = early;
	tail.summary_final = true; // (nothing after the count flags a component any more: crossings are resolved in place)
	tail.overlapped = tail.want_overlap && tail.summary_final;
	// The five PVST arrays back to back (povu_hip_forest::alloc has the same layout).  Small results are written by the
	// emit kernels straight into the forest's page-locked host block (no copy, no extra launch).  Large ones go
	// through a device block of the same layout: the scattered 4- and 1-byte stores of the emit kernels make poor PCIe
	// packets (~25 GB/s measured), the copy engine moves the finished arrays at link speed -- all copies on the side
	// stream: endpoints and orientations while this one still works out levels and parents, the parents behind them.
	const size_t p4 = (total * 4 + 63) & ~size_t(63), p1 = (total + 63) & ~size_t(63);
	char *host_blk = static_cast<char *>(alloc_result_block(total));
	const bool staged = total >= PVST_STAGE_MIN && side.stream != nullptr;
	if (!staged)
		count_kernel_d2h(total * 14); // (the emit kernels write the five arrays straight into the page-locked block)
	char *blk = staged ? reinterpret_cast<char *>(pw.stage) : host_blk;
	pw.d_a = reinterpret_cast<uint32_t *>(blk);
	pw.d_z = reinterpret_cast<uint32_t *>(blk + p4);
	pw.d_parent = reinterpret_cast<uint32_t *>(blk + 2 * p4);
	pw.d_aor = reinterpret_cast<uint8_t *>(blk + 3 * p4);
	pw.d_zor = reinterpret_cast<uint8_t *>(blk + 3 * p4 + p1);
	pw.d_total = total;
	if (use_side) {
		// (the roots' slots with the endpoints: the copies behind them take whole arrays)
		HIP_CHECK(hipStreamWaitEvent(side.stream, side.fork, 0));
		KLAUNCH(k_pvst_roots, dim3(nblk((size_t)C)), dim3(TPB), 0, side.stream, C, sw.c_ntree, pw.doff, pw.d_parent, pw.d_a, pw.d_z, pw.d_aor, pw.d_zor);
		KLAUNCH(k_emit_endpoints, dim3(staged ? nblk((S + 3) / 4) : std::min<unsigned>(nblk((S + 3) / 4), 160)), dim3(TPB), 0, side.stream, S, dflag, pw.erank,
			pw.s_comp, pw.ns, pw.s_vtx, sw.t_flags, sw.t_gid, pw.cproc_ps, pw.d_a, pw.d_z, pw.d_aor, pw.d_zor);
		if (staged) {
			HIP_CHECK(copy_async(host_blk, blk, 2 * p4, hipMemcpyDeviceToHost, side.stream));
			HIP_CHECK(copy_async(host_blk + 3 * p4, blk + 3 * p4, 2 * p1, hipMemcpyDeviceToHost, side.stream));
		}
	} else {
		LAUNCH(k_pvst_roots, (size_t)C, s, C, sw.c_ntree, pw.doff, pw.d_parent, pw.d_a, pw.d_z, pw.d_aor, pw.d_zor);
		if (S)
			KLAUNCH(k_emit_endpoints, dim3(nblk((S + 3) / 4)), dim3(TPB), 0, s, S, dflag, pw.erank, pw.s_comp, pw.ns, pw.s_vtx, sw.t_flags,
				sw.t_gid, pw.cproc_ps, pw.d_a, pw.d_z, pw.d_aor, pw.d_zor);
	}
	seg_build(pw.segL, pw.lev, NE, s);
	LAUNCH(k_pvst_emit, NE, s, NE, pw.lev, pw.e_i, pw.segL, pw.s_comp, pw.soff, pw.erank, pw.cproc_ps,
	       pw.d_parent);
	// the parents follow the endpoints on the side stream (the copy engine takes one array after the other anyway): the main
	// stream ends with its last kernel
	hipStream_t last = s;
	if (use_side) {
		if (staged) {
			HIP_CHECK(hipEventRecord(side.fork2, s));
			HIP_CHECK(hipStreamWaitEvent(side.stream, side.fork2, 0));
			HIP_CHECK(copy_async(host_blk + 2 * p4, blk + 2 * p4, p4, hipMemcpyDeviceToHost, side.stream));
		}
		HIP_CHECK(hipEventRecord(side.join, side.stream));
		if (!tail.overlapped)
			HIP_CHECK(hipStreamWaitEvent(s, side.join, 0)); // the pass is complete when both streams are
		else
			last = side.stream; // (the side stream's last work waited for the main stream's last kernel when the result is staged;
					    //  when it is not, `done` below is recorded on both -- see the caller)
	} else if (staged) {
		HIP_CHECK(copy_async(host_blk + 2 * p4, blk + 2 * p4, p4, hipMemcpyDeviceToHost, s));
	}
	if (tail.done) {
		if (tail.overlapped && !staged && use_side) { // both streams carry kernels that write the host block: join them on the side
			HIP_CHECK(hipEventRecord(side.fork2, s));
			HIP_CHECK(hipStreamWaitEvent(side.stream, side.fork2, 0));
		}
		HIP_CHECK(hipEventRecord(tail.done, last));
		if (tail.done2)
			HIP_CHECK(hipEventRecord(tail.done2, last));
	}
	pw.n_stack = S; // export_parallel_stack copies the stack into the per-component layout when a debug hook asks
	tm.end(12 + 3 * 22);
}

// class ids of the stack entries after a black-only pass (debug hooks): the runs of the sorted order, numbered
static void stack_class_ids(ParWs &pw, hipStream_t s)
{
	if (pw.s_cls_valid)
		return;
	const uint32_t V = pw.V;
	scan_exclusive_u8(pw.f8a, pw.psA, (size_t)V + 1, nullptr, nullptr, 0, pw.scan_tmp, pw.scan_tmp_bytes, s);
	LAUNCH(k_class_ids_black, V, s, V, pw.keys_t2, pw.vals_t2, pw.f8a, pw.psA, pw.s_cls);
	pw.s_cls_valid = true;
}
void classes_to_tree_space(ParWs &pw, hipStream_t s)
{
	if (pw.gcls_valid)
		return;
	stack_class_ids(pw, s);
	HIP_CHECK(hipMemsetAsync(pw.gcls, 0xFF, (size_t)pw.T * 4, s));
	if (pw.n_stack)
		LAUNCH(k_cls_from_stack, pw.n_stack, s, pw.n_stack, pw.s_vtx, pw.s_cls, pw.gcls);
	pw.gcls_valid = true;
}

void export_parallel_stack(const CompState &cs, SeqWs &sw, ParWs &pw, hipStream_t s)
{
	stack_class_ids(pw, s);
	const uint32_t S = pw.n_stack;
	LAUNCH(k_export_stack, S, s, S, pw.s_comp, pw.soff, cs.voff, pw.s_vtx, pw.s_cls, pw.ns, sw.s_vtx, sw.s_cls,
	       sw.next_seen);
}

} // namespace povu_hip
