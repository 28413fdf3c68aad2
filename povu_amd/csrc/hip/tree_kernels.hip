// tree_kernels.hip -- row C (pst::Tree::from_bd, spanning_tree.cpp:262-463) without a sequential
// walk over the whole component.
//
// from_bd is the lexicographic DFS of the biedged graph H (vertices = segment sides, one black edge
// per segment, one gray edge per link) in which every side scans [black edge, gray links by local
// edge idx].  Lexicographic DFS is sequential in general, but it decomposes along the bridges of H:
// the DFS can enter a 2-edge-connected class only through the unique bridge that leads to the root,
// and its walk inside a class never depends on the rest of the graph.  So
//   1. spanning forest of H  (black edges + the links that won a hook in the WCC union-find)
//   2. root it at the DFS start side: Euler tour over the forest of segments + list ranking (random splitters, recursive)
//   3. bridges: a tree edge is a bridge iff the xor of TWO independent hashes of the non-tree links over its subtree's
//      stretch of the tour vanishes (running xor over the values in tour order, kept compact behind a bitmap of the
//      positions that carry one)
//   4. 2-edge-connected classes = the pieces of the rooted forest when its bridges are cut (never numbered)
//   5. per class: entry side = top of the class, DFS parent = far end of its bridge
//   6. the reference DFS inside every class: ONE LANE per small class (pangenome graphs are chains of small bubbles, so
//      there are millions of them), ONE WAVE per class that turns out large (candidates probed by ballot, one dependent
//      gather per tree edge)
//   7. pre-order / subtree size / depth of the union tree: Euler tour of (parent, scan-slot)
//      ordered children + list ranking
//   8. tree arrays in pre-order (T-space, as the class stage reads them) + the from_bd back edges (de-duplication rules
//      of :360-398)
// tests/test_parallel_tree_model.py checks this formulation against the oracle on the CPU.
//
// How the kernels are shaped ("What bounds the pass"): nearly every kernel here is bound by the
// vector-memory instructions its CU retires, so they issue few and wide ones -- one lane per SEGMENT where the two sides
// ask different questions (k_bridges, k_tree_emit, k_t0_parents), four slots of an adjacency list a round with 16-byte
// loads and the gathers they lead to issued together (k_tour_words, k_events, side_back_edges, k_class_dfs_small), four
// elements a lane where a kernel only streams (k_entry_list).
#include "tree_kernels.hpp"

#include <cstdlib>

#include <algorithm>

namespace povu_hip
{

#ifndef NIL
#define NIL POVU_NIL
#endif
static constexpr int TPB = 256;
static inline unsigned nblk(size_t n) { return (unsigned)((n + TPB - 1) / TPB); }
#define LAUNCH(k, n, s, ...)                                                                     \
	do {                                                                                     \
		if ((n) > 0) {                                                                   \
			hipLaunchKernelGGL(k, dim3(nblk(n)), dim3(TPB), 0, s, __VA_ARGS__);      \
			HIP_CHECK(hipGetLastError());                                            \
		}                                                                                \
	} while (0)

// ---- list-ranking words and splitters (section "work-efficient list ranking" below).
// Splitters cut every list into short segments.  The index space of a list level is cut into buckets of 2^b
// consecutive elements and every bucket names exactly ONE of its elements as a splitter, pseudo-randomly
// (multiplicative hash of the bucket idx).  So the splitter of bucket q is computed, not looked up; a splitter's id on
// the next level IS its bucket idx (dense, no flag array, no scan, no compaction), and lane q of a walk kernel works
// near element q * 2^b (the scattered rank stores of neighbouring lanes meet again in L2).
// b = 3 (1 in 8): short segments, whose walks are bound by their longest one; b = 4 is kept as an A/B switch.
static unsigned rank_bucket_bits(size_t n)
{
	if (const char *ev = getenv("POVU_HIP_RANK_BITS")) // (tuning hook)
		return (unsigned)std::min(6, std::max(2, atoi(ev)));
	(void)n;
	return 3u; // (1 in 16 measured 0.2 ms slower on 2.4e8 slots once the other kernels had changed; POVU_HIP_F_SPARSE_SPLITTERS still forces it)
}
static constexpr uint32_t PK_END = 0x1FFFFFFFu, PK_STOP = 0x80000000u; // words of the levels above the list (their elements number an eighth of it)
// words of the list itself (level 0): bits 0..29 successor (P0_END = none), bit 30 = the element's 0/1 weight, bit 31 = the
// element heads a list.  Whether the walk stops behind an element -- its successor is a splitter, or there is none -- is
// worked out from the successor (three integer operations), not stored: the bit it would take is the difference between
// 1.8 * 10^8 and 3.6 * 10^8 segments a graph may have.
static constexpr uint32_t P0_END = 0x3FFFFFFFu, P0_W = 0x40000000u, P0_HEAD = 0x80000000u;
static constexpr uint32_t FT_NONE = 0x7FFFFFFFu, FT_HASH = 0x80000000u; // ft words: first arc of a side | "its hash word was written"
__device__ __forceinline__ uint32_t bucket_splitter(uint32_t q, unsigned b) { return (q << b) | ((q * 0x9E3779B1u) >> (32u - b)); }
__device__ __forceinline__ bool is_splitter(uint32_t i, unsigned b) { return bucket_splitter(i >> b, b) == i; }
// One word per list element, so that a walk step is ONE dependent load: bits 0..28 successor (PK_END = none),
// bit 29 = the element's 0/1 weight, bit 30 = the element heads a list (nobody's successor; set by the kernels that
// know the heads), bit 31 = stop after this element (the successor is a splitter, or there is none).
__device__ __forceinline__ uint32_t rank_pack(uint32_t nx, uint32_t w, unsigned b)
{
	(void)b;
	return (nx == NIL ? P0_END : nx) | ((w & 1u) ? P0_W : 0u);
}
__device__ __forceinline__ bool rank_l0_stop(uint32_t nx, unsigned b) { return nx == P0_END || is_splitter(nx, b); }

// ------------------------------------------------------------------ 1. Euler tour of the spanning forest
// The forest = every black edge + the links that won a hook in the union-find over the segments.  A black edge never
// needs rooting (whichever side of a segment the tour enters first is the parent of the other), so the tour runs over
// the forest of SEGMENTS joined by the hooked links: its arcs ARE the adjacency slots of those links (slot = index
// into ladj / lle, two per link), half as many list elements as a tour over the sides.  Nothing is numbered, counted
// or stored per arc: the twin of a slot is found in the (short, ascending) list of the side at the other end, and
// the arcs around a segment are taken in slot order -- the slots of its l side, then those of its r side, which are
// one contiguous index range [loff[2w], loff[2w + 2]) -- so the subtrees hanging off ONE side are contiguous in the tour.
// slot (1-based) of the link with id `le` (LLE_ID bits of its lle words) in the list of side w (ascending by link id)
__device__ __forceinline__ uint32_t find_link_slot(const uint32_t *__restrict__ loff, const uint32_t *__restrict__ lle, uint32_t w,
						   uint32_t le)
{
	uint32_t lo = loff[w], hi = loff[w + 1];
	const uint32_t l0 = lo;
	if (hi - lo > 8) {
		while (lo < hi) {
			const uint32_t mid = (lo + hi) >> 1;
			if ((lle[mid] & LLE_ID) < le)
				lo = mid + 1;
			else
				hi = mid;
		}
		return lo - l0 + 1;
	}
	for (uint32_t j = lo; j < hi; j++)
		if ((lle[j] & LLE_ID) == le)
			return j - l0 + 1;
	return 0; // (not reached: a link sits in the lists of both its ends)
}
// two consecutive words to a 4-byte aligned address in one store instruction
__device__ __forceinline__ void store2_unaligned(uint32_t *p, uint32_t a, uint32_t b)
{
	typedef uint32_t v2a __attribute__((ext_vector_type(2), aligned(4)));
	v2a v;
	v.x = a;
	v.y = b;
	*reinterpret_cast<v2a *>(p) = v;
}
// adjacency index of the same link in the list of the side at its other end
__device__ __forceinline__ uint32_t arc_twin(const uint32_t *__restrict__ loff, const uint32_t *__restrict__ ladj,
					     const uint32_t *__restrict__ lle, uint32_t at)
{
	const uint32_t w = ladj[at];
	return loff[w] + find_link_slot(loff, lle, w, lle[at] & LLE_ID) - 1;
}
// (bridge test of step 3, see k_t0_parents) every non-tree link gets a 64-bit hash of its local edge idx
// -- TWO independent 64-bit hashes, carried side by side in one 16-byte word (the second one is what stands behind the
// first: a tree edge is called a bridge only when both running xors vanish)
__device__ __forceinline__ ulonglong2 link_hash(uint32_t le)
{
	unsigned long long z = ((unsigned long long)le + 1ull) * 0x9E3779B97F4A7C15ull; // splitmix64 finaliser
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	unsigned long long y = ((unsigned long long)le ^ 0x5851F42D4C957F2Dull) * 0xD6E8FEB86659FD93ull; // murmur3 finaliser, other constants
	y = (y ^ (y >> 33)) * 0xFF51AFD7ED558CCDull;
	y = (y ^ (y >> 33)) * 0xC4CEB9FE1A85EC53ull;
	return make_ulonglong2(z ^ (z >> 31), y ^ (y >> 33));
}
__device__ __forceinline__ ulonglong2 hx(const ulonglong2 a, const ulonglong2 b) { return make_ulonglong2(a.x ^ b.x, a.y ^ b.y); }
__device__ __forceinline__ bool hzero(const ulonglong2 a) { return (a.x | a.y) == 0ull; }
__device__ __forceinline__ bool heq(const ulonglong2 a, const ulonglong2 b) { return a.x == b.x && a.y == b.y; }
// Euler tour successor: after u->w comes the arc that follows w->u among the arcs of w's segment (cyclically; w->u
// itself when the segment has no other).  Slots that are no arcs get an inert word (no lane ever walks into them).
// Also, while the side's links are in hand: hside[S] = xor of the hashes of its non-tree links (a link is in the lists
// of both its ends, also when they are l and r of one segment), written only where it is not zero (four sides in five
// of a pangenome graph have no such link); ft[S] = its first arc (FT_NONE: none), bit 31 set when hside[S] was written.
__global__ void k_tour_words(uint32_t nS, const uint32_t *__restrict__ loff, const uint32_t *__restrict__ ladj,
			     const uint32_t *__restrict__ lle, uint32_t *__restrict__ pk, unsigned b,
			     ulonglong2 *__restrict__ hside, uint32_t *__restrict__ ft)
{
	uint32_t S = BIDX * blockDim.x + threadIdx.x;
	if (S >= nS)
		return;
	const uint32_t lo = loff[S], hi = loff[S + 1];
	ulonglong2 h = make_ulonglong2(0ull, 0ull);
	uint32_t first = NIL;
	// The slots of the far side's SEGMENT (l side, then r side: one index range) answer both questions about a forest slot
	// -- where the link sits in the far side's list (its twin) and which forest slot follows the twin around the segment.
	// A gather whose 64 lanes hit 64 lines keeps the CU's address unit busy for 64 cycles whatever it returns, and this
	// kernel is bound by exactly that: four slots are taken a round, in phases, so that the loads of one phase are all in
	// flight together -- own words (two 16-byte loads), the far segments' offsets, their slot words (a segment with at
	// most eight slots, nearly all, in one or two 16-byte loads, searched in registers).
	for (uint32_t at0 = lo; at0 < hi; at0 += 4) {
		const uint32_t rem = hi - at0;
		const uint4 lw4 = load4_unaligned(lle + at0), w4 = load4_unaligned(ladj + at0);
		const uint32_t lws[4] = {lw4.x, lw4.y, lw4.z, lw4.w}, ws[4] = {w4.x, w4.y, w4.z, w4.w};
		bool tree[4];
		uint2 o01[4];
		uint32_t se[4];
#pragma unroll
		for (uint32_t q = 0; q < 4; q++) {
			tree[q] = q < rem && (lws[q] & LLE_TREE);
			o01[q] = make_uint2(0u, 0u);
			se[q] = 0;
			if (tree[q]) {
				const uint32_t g2 = ws[q] & ~1u;
				o01[q] = *reinterpret_cast<const uint2 *>(loff + g2); // (g2 is even: 8-byte aligned)
				se[q] = loff[g2 + 2];
			}
		}
		uint4 a[4];
#pragma unroll
		for (uint32_t q = 0; q < 4; q++) {
			a[q] = make_uint4(0u, 0u, 0u, 0u);
			if (tree[q] && se[q] - o01[q].x <= 8)
				a[q] = load4_unaligned(lle + o01[q].x);
		}
#pragma unroll
		for (uint32_t q = 0; q < 4; q++) {
			if (q >= rem)
				break;
			const uint32_t at = at0 + q, lw = lws[q], le = lw & LLE_ID;
			if (!tree[q]) {
				pk[at] = P0_END;
				h = hx(h, link_hash(le));
				continue;
			}
			if (first == NIL)
				first = at;
			const uint32_t w = ws[q], sb = o01[q].x;
			const uint32_t wl = (w & 1u) ? o01[q].y : sb, wh = (w & 1u) ? se[q] : o01[q].y; // w's own slots
			uint32_t nxt;
			if (se[q] - sb <= 8) {
				const uint32_t n = se[q] - sb, tl = wl - sb, th = wh - sb;
				uint4 c = make_uint4(0u, 0u, 0u, 0u);
				if (n > 4)
					c = load4_unaligned(lle + sb + 4);
				const uint32_t r[8] = {a[q].x, a[q].y, a[q].z, a[q].w, c.x, c.y, c.z, c.w};
				uint32_t tk = 0, first_tree = NIL, after = NIL;
#pragma unroll
				for (uint32_t k = 0; k < 8; k++)
					if (k >= tl && k < th && (r[k] & LLE_ID) == le)
						tk = k;
#pragma unroll
				for (uint32_t k = 0; k < 8; k++)
					if (k < n && (r[k] & LLE_TREE)) {
						if (first_tree == NIL)
							first_tree = k;
						if (k > tk && after == NIL)
							after = k;
					}
				nxt = sb + (after != NIL ? after : (first_tree < tk ? first_tree : tk)); // (the twin itself is a forest slot: first_tree <= tk)
			} else {
				const uint32_t t = wl + find_link_slot(loff, lle, w, le) - 1;
				nxt = t;
				for (uint32_t jj = t + 1; jj < se[q] && nxt == t; jj++)
					if (lle[jj] & LLE_TREE)
						nxt = jj;
				for (uint32_t jj = sb; jj < t && nxt == t; jj++)
					if (lle[jj] & LLE_TREE)
						nxt = jj;
			}
			pk[at] = rank_pack(nxt, 1u, b); // every arc counts 1 (k_tour_ends fixes the closing arc)
		}
	}
	const bool nz = !hzero(h);
	if (nz)
		hside[S] = h;
	ft[S] = (first == NIL ? FT_NONE : first) | (nz ? FT_HASH : 0u);
}
// sorted side id of the DFS start of component c: smallest tip (types.cpp:60-68) or (l, idx 0)
__device__ __forceinline__ uint32_t comp_root_side(const unsigned long long *start_key, const uint32_t *voff, uint32_t c)
{
	unsigned long long k = start_key[c];
	return k == ~0ull ? 2 * voff[c] : (uint32_t)(k & 0xFFFFFFFFu);
}
// per component: the tour starts with the first arc of the root segment, its slots taken from the DFS start side on
// (start side, then the other side), and is cut open behind the arc that returns to the root for the last time.
// A component of one segment has no arc and heads no list.
__global__ void k_tour_ends(uint32_t C, const uint32_t *__restrict__ voff, const unsigned long long *__restrict__ start_key,
			    const uint32_t *__restrict__ loff, const uint32_t *__restrict__ ladj, const uint32_t *__restrict__ lle,
			    uint32_t *__restrict__ pk, uint32_t *__restrict__ heads)
{
	uint32_t c = BIDX * blockDim.x + threadIdx.x;
	if (c >= C)
		return;
	const uint32_t r = comp_root_side(start_key, voff, c), sb = loff[r & ~1u], se = loff[(r & ~1u) + 2], mid = loff[r];
	// cyclic order from `mid`: [mid, se) then [sb, mid)
	uint32_t a_first = NIL, a_last = NIL;
	for (uint32_t j = mid; j < se; j++)
		if (lle[j] & LLE_TREE) {
			if (a_first == NIL)
				a_first = j;
			a_last = j;
		}
	for (uint32_t j = sb; j < mid; j++)
		if (lle[j] & LLE_TREE) {
			if (a_first == NIL)
				a_first = j;
			a_last = j;
		}
	heads[c] = a_first;
	if (a_first == NIL)
		return;
	pk[arc_twin(loff, ladj, lle, a_last)] = P0_END; // no successor, weight 0
	atomicOr(&pk[a_first], P0_HEAD);		 // (k_tour_words wrote the word in an earlier launch)
}
// one launch = several rounds of pointer jumping with two accumulators (suffix sums along the list):
// HOPS = 3 covers two rounds (every pointer then spans 4x as far), HOPS = 7 three rounds (8x).  More
// hops per launch mean fewer launches but more loads in total, which pays only while the launches are
// dispatch-bound (short splitter lists).
template <int HOPS>
__global__ void k_wyllie(uint32_t n, const uint32_t *__restrict__ nxt_in, const uint32_t *__restrict__ a_in,
			 const uint32_t *__restrict__ b_in, uint32_t *__restrict__ nxt_out, uint32_t *__restrict__ a_out,
			 uint32_t *__restrict__ b_out, const uint32_t *__restrict__ n_dev)
{
	uint32_t i = BIDX * blockDim.x + threadIdx.x;
	if (i >= n || (n_dev && i >= *n_dev))
		return;
	uint32_t nx = nxt_in[i], a = a_in[i], b = b_in ? b_in[i] : 0;
#pragma unroll
	for (int hop = 0; hop < HOPS; hop++) {
		if (nx == NIL)
			break;
		a += a_in[nx];
		if (b_in)
			b += b_in[nx];
		nx = nxt_in[nx];
	}
	nxt_out[i] = nx;
	a_out[i] = a;
	if (b_out)
		b_out[i] = b;
}
// `bits` = bits_for(longest possible list); returns which buffer set (0 = A, 1 = B) holds the result
static int list_rank(uint32_t n, unsigned bits, uint32_t *nxtA, uint32_t *nxtB, uint32_t *aA, uint32_t *aB, uint32_t *bA,
		     uint32_t *bB, hipStream_t s, const uint32_t *n_dev = nullptr)
{
	int cur = 0;
	const bool small = n < (1u << 21);
	const unsigned launches = small ? (bits + 2) / 3 : (bits + 1) / 2;
	for (unsigned r = 0; r < launches; r++) {
		uint32_t *ni = cur ? nxtB : nxtA, *no = cur ? nxtA : nxtB, *ai = cur ? aB : aA, *ao = cur ? aA : aB;
		uint32_t *bi = cur ? bB : bA, *bo = cur ? bA : bB;
		if (small)
			LAUNCH(k_wyllie<7>, n, s, n, ni, ai, bi, no, ao, bo, n_dev);
		else
			LAUNCH(k_wyllie<3>, n, s, n, ni, ai, bi, no, ao, bo, n_dev);
		cur ^= 1;
	}
	return cur;
}

// ---- work-efficient list ranking, recursive.  Level 0 is the list itself (packed words, 0/1 weights).  One lane per
// splitter (= per bucket, plus one per list head) walks its segment and leaves {next splitter, segment sums} -- the
// element of the next level, whose index is the bucket idx.  Levels shrink by 2^b until one workgroup ranks the top in
// LDS (pointer jumping); then every level hands its elements their suffix sums on the way back down.  A list of n
// elements costs about n(1 + 2^-b + ...) dependent loads up and the same down.
// Heads: an element that heads a list is never reached through its bucket (its lane stays idle); head h of the caller's
// head array is element m + h of every level above 0 (m = buckets of the level below) and is walked by its own lane.
// TWO: second weight = +1 where the first is 1, -1 where it is 0 (enter / leave events).
static constexpr uint32_t RANK_TOP = 8192; // elements one workgroup ranks in LDS
struct RankLevelArgs {
	uint32_t n;	// elements of this level
	uint32_t m;	// buckets of this level = hash lanes of the walk
	uint32_t hbase; // first head element of this level (level 0: unused, heads come from the array)
	uint32_t M;	// lanes of the walk = elements of the next level = m + heads
};
// start element of lane `id`, NIL when the lane has nothing to walk
template <bool L0>
__device__ __forceinline__ uint32_t rank_lane_start(uint32_t id, const RankLevelArgs &A, unsigned b, const uint32_t *__restrict__ pk0,
						    const uint32_t *__restrict__ heads)
{
	if (id < A.m) {
		const uint32_t x = bucket_splitter(id, b);
		if (x >= A.n)
			return NIL;
		if (L0)
			return (pk0[x] & P0_HEAD) ? NIL : x;
		return x >= A.hbase ? NIL : x;
	}
	return L0 ? heads[id - A.m] : A.hbase + (id - A.m);
}
// EVT3 (level 0 only, with TWO): the elements are the events of the pre-order ranking, three per segment, and an
// element's two weights follow from its index: 3g = enter both sides of segment g (+2 entered, depth +2), 3g + 1 =
// leave the far side (depth -1, or -2 when the weight bit says the entered side is left with it), 3g + 2 = leave the
// entered side (depth -1).
template <bool EVT3>
__device__ __forceinline__ void rank_l0_weights(uint32_t x, uint32_t p, uint32_t &wa, uint32_t &wb)
{
	if (EVT3) {
		const uint32_t t = x % 3u;
		wa = t == 0 ? 2u : 0u;
		wb = t == 0 ? 2u : ((t == 1 && (p & P0_W)) ? 0u - 2u : 0u - 1u);
	} else {
		wa = (p & P0_W) ? 1u : 0u;
		wb = wa ? 1u : 0xFFFFFFFFu;
	}
}
template <bool L0, bool TWO, bool EVT3 = false>
__global__ void k_rank_up(RankLevelArgs A, unsigned b, const uint32_t *__restrict__ nx_in, const uint32_t *__restrict__ a_in,
			  const uint32_t *__restrict__ b_in, const uint32_t *__restrict__ heads, uint32_t *__restrict__ nx_out,
			  uint32_t *__restrict__ a_out, uint32_t *__restrict__ b_out)
{
	const uint32_t id = BIDX * blockDim.x + threadIdx.x;
	if (id >= A.M)
		return;
	uint32_t x = rank_lane_start<L0>(id, A, b, nx_in, heads);
	uint32_t sa = 0, sb = 0, out = PK_END | PK_STOP;
	if (x != NIL) {
		uint32_t p;
		bool stop;
		do {
			p = nx_in[x];
			if (L0) {
				uint32_t wa, wb;
				rank_l0_weights<EVT3>(x, p, wa, wb);
				sa += wa;
				if (TWO)
					sb += wb;
				x = p & P0_END;
				stop = rank_l0_stop(x, b);
			} else {
				sa += a_in[x];
				if (TWO)
					sb += b_in[x];
				x = p & PK_END;
				stop = (p & PK_STOP) != 0;
			}
		} while (!stop);
		if (x != (L0 ? P0_END : PK_END)) { // the successor is the splitter of its bucket: that bucket is its idx one level up
			const uint32_t q = x >> b;
			out = q | (is_splitter(q, b) ? PK_STOP : 0u);
		}
	}
	nx_out[id] = out;
	a_out[id] = sa;
	if (TWO)
		b_out[id] = sb;
}
// the way back: lane `id` knows the suffix sums at its splitter (ra / rb of the level above) and hands every element of
// its segment its own.  Level 0 writes the caller's arrays, the levels above overwrite their weights in place.
template <bool L0, bool TWO, bool EVT3 = false>
__global__ void k_rank_down(RankLevelArgs A, unsigned b, const uint32_t *__restrict__ nx_in, uint32_t *a_io, uint32_t *b_io,
			    const uint32_t *__restrict__ heads, const uint32_t *__restrict__ ra, const uint32_t *__restrict__ rb,
			    uint32_t *__restrict__ out1, uint2 *__restrict__ out12)
{
	const uint32_t id = BIDX * blockDim.x + threadIdx.x;
	if (id >= A.M)
		return;
	uint32_t x = rank_lane_start<L0>(id, A, b, nx_in, heads);
	if (x == NIL)
		return;
	uint32_t sa = ra[id], sb = TWO ? rb[id] : 0, p;
	do {
		p = nx_in[x];
		if (L0) {
			uint32_t wa, wb;
			rank_l0_weights<EVT3>(x, p, wa, wb);
			if (TWO) { // both sums in one 8-byte store
				out12[x] = make_uint2(sa, sb);
				sb -= wb;
			} else {
				out1[x] = sa;
			}
			sa -= wa;
		} else {
			const uint32_t wa = a_io[x];
			a_io[x] = sa;
			sa -= wa;
			if (TWO) {
				const uint32_t wb = b_io[x];
				b_io[x] = sb;
				sb -= wb;
			}
		}
		if (L0) {
			x = p & P0_END;
			if (rank_l0_stop(x, b))
				break;
		} else {
			x = p & PK_END;
			if (p & PK_STOP)
				break;
		}
	} while (true);
}
// top level: inclusive suffix sums of at most RANK_TOP elements by pointer jumping in LDS, one workgroup
template <bool TWO>
__global__ void __launch_bounds__(1024) k_rank_top(uint32_t n, const uint32_t *__restrict__ nx_in, uint32_t *a_io, uint32_t *b_io)
{
	__shared__ uint32_t nx[RANK_TOP], sa[RANK_TOP], sb[TWO ? RANK_TOP : 1];
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
		const uint32_t p = nx_in[i] & PK_END;
		nx[i] = p == PK_END ? NIL : p;
		sa[i] = a_io[i];
		if (TWO)
			sb[i] = b_io[i];
	}
	__syncthreads();
	constexpr int PER = RANK_TOP / 1024;
	for (uint32_t span = 1; span < n; span <<= 1) {
		uint32_t tn[PER], ta[PER], tb[PER];
#pragma unroll
		for (int k = 0; k < PER; k++) {
			const uint32_t i = threadIdx.x + k * 1024;
			tn[k] = NIL, ta[k] = 0, tb[k] = 0;
			if (i < n && nx[i] != NIL) {
				const uint32_t j = nx[i];
				tn[k] = nx[j];
				ta[k] = sa[j];
				if (TWO)
					tb[k] = sb[j];
			}
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < PER; k++) {
			const uint32_t i = threadIdx.x + k * 1024;
			if (i < n && nx[i] != NIL) {
				nx[i] = tn[k];
				sa[i] += ta[k];
				if (TWO)
					sb[i] += tb[k];
			}
		}
		__syncthreads();
	}
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
		a_io[i] = sa[i];
		if (TWO)
			b_io[i] = sb[i];
	}
}
// (top level too large for one workgroup -- graphs with very many tiny components: the old pointer jumping in global memory)
__global__ void k_rank_unpack_next(uint32_t n, const uint32_t *__restrict__ nx_in, uint32_t *__restrict__ out)
{
	uint32_t i = BIDX * blockDim.x + threadIdx.x;
	if (i < n) {
		const uint32_t p = nx_in[i] & PK_END;
		out[i] = p == PK_END ? NIL : p;
	}
}

struct RankBufs {
	uint32_t *pk;			 // [n+1] packed list words of level 0 (rank_pack)
	uint32_t *heads;		 // [max heads] level-0 element that heads list h, NIL for none
	uint32_t *nx, *wa, *wb;		 // pools for the levels above 0 (next words, sums), rank_pool_words() each
	uint32_t *tA, *tB, *tC;		 // three more pools: ping-pong of the global-memory fallback of the top level
};
static constexpr int RANK_MAX_LEVELS = 12;
// words every pool needs for lists of n elements in total with nh heads: the first level above the list has
// M = n / 2^b + nh elements (b >= 3), the following ones at least halve until the last, which may be as large as
// the one before it again -- < 3.2 M in all
static size_t rank_pool_words(size_t n, size_t nh) { return n / 2 + 4 * nh + 64; }

// suffix sums (inclusive) along the lists packed in rb.pk: out1 of the 0/1 weights or, when TWO, out12 = {that sum,
// the sum of the +-1 weights derived from them}
template <bool TWO, bool EVT3 = false>
static void list_rank_splitters(uint32_t n, unsigned b, uint32_t *out1, uint2 *out12, uint32_t nh, RankBufs &rb, hipStream_t s)
{
	if (n >= P0_END)
		throw HipError("list ranking: more than 2^30 elements (graph too large for the packed walk)");
	// plan: level L has lv[L].n elements; its walk has lv[L].M lanes = the elements of level L + 1
	RankLevelArgs lv[RANK_MAX_LEVELS];
	uint32_t *nxp[RANK_MAX_LEVELS + 1], *wap[RANK_MAX_LEVELS + 1], *wbp[RANK_MAX_LEVELS + 1];
	int levels = 0;
	size_t used = 0;
	const size_t pool = rank_pool_words(n, nh);
	nxp[0] = rb.pk, wap[0] = wbp[0] = nullptr;
	uint32_t cur_n = n, hbase = 0;
	for (;;) {
		RankLevelArgs &A = lv[levels];
		A.n = cur_n;
		A.m = (cur_n + (1u << b) - 1) >> b;
		A.hbase = hbase;
		A.M = A.m + nh;
		if (used + A.M > pool)
			throw HipError("list ranking: level pool exhausted (internal sizing bug)");
		nxp[levels + 1] = rb.nx + used, wap[levels + 1] = rb.wa + used, wbp[levels + 1] = rb.wb + used;
		used += A.M;
		levels++;
		hbase = A.m;
		const uint32_t next_n = A.M;
		// stop when one workgroup can finish, or when the heads keep the levels from shrinking
		if (next_n <= RANK_TOP || next_n > cur_n / 2 || levels == RANK_MAX_LEVELS) {
			cur_n = next_n;
			break;
		}
		cur_n = next_n;
	}
	for (int L = 0; L < levels; L++) {
		const RankLevelArgs &A = lv[L];
		if (L == 0)
			LAUNCH((k_rank_up<true, TWO, EVT3>), A.M, s, A, b, nxp[0], nullptr, nullptr, rb.heads, nxp[1], wap[1], wbp[1]);
		else
			LAUNCH((k_rank_up<false, TWO>), A.M, s, A, b, nxp[L], wap[L], wbp[L], nullptr, nxp[L + 1], wap[L + 1], wbp[L + 1]);
	}
	const uint32_t nt = cur_n; // elements of the top level
	if (nt <= RANK_TOP) {
		KLAUNCH((k_rank_top<TWO>), dim3(1), dim3(1024), 0, s, nt, nxp[levels], wap[levels], wbp[levels]);
	} else {
		LAUNCH(k_rank_unpack_next, nt, s, nt, nxp[levels], rb.tA);
		HIP_CHECK(copy_async(rb.tB, wap[levels], (size_t)nt * 4, hipMemcpyDeviceToDevice, s));
		if (TWO)
			HIP_CHECK(copy_async(rb.tC, wbp[levels], (size_t)nt * 4, hipMemcpyDeviceToDevice, s));
		// ping-pong partners: the level's own arrays (their contents were just copied out)
		uint32_t *nA = rb.tA, *nB = nxp[levels], *aA = rb.tB, *aB = wap[levels], *bA = rb.tC, *bB = wbp[levels];
		const int side = list_rank(nt, bits_for(nt) + 1, nA, nB, aA, aB, TWO ? bA : nullptr, TWO ? bB : nullptr, s);
		if (side == 0) { // result in the A set: bring it home
			HIP_CHECK(copy_async(wap[levels], aA, (size_t)nt * 4, hipMemcpyDeviceToDevice, s));
			if (TWO)
				HIP_CHECK(copy_async(wbp[levels], bA, (size_t)nt * 4, hipMemcpyDeviceToDevice, s));
		}
	}
	for (int L = levels - 1; L >= 0; L--) {
		const RankLevelArgs &A = lv[L];
		if (L == 0)
			LAUNCH((k_rank_down<true, TWO, EVT3>), A.M, s, A, b, nxp[0], nullptr, nullptr, rb.heads, wap[1], wbp[1], out1, out12);
		else
			LAUNCH((k_rank_down<false, TWO>), A.M, s, A, b, nxp[L], wap[L], wbp[L], nullptr, wap[L + 1], wbp[L + 1], nullptr,
			       nullptr);
	}
}

// ------------------------------------------------------------------ 2. rooted forest T0 and 3. its bridges
// dist[a] = arcs after a in its tour; position of arc a = (arcs of the components before) + L - 1 - dist[a], L = 2 (nv - 1).
// u->w is the advance arc of the hooked link {u,w} iff it comes first: then w is the side its segment is ENTERED
// through (parent u), the segment's other side is the child of w over the black edge, and the segment's subtree is the
// stretch of the tour from that arc (position tin) to its twin (position tout).  One record per segment:
// t0seg[segment] = {parent side of the entered side, its link | entered side's r/l bit << 31, tin, tout}; the root
// segment of a component is entered through the DFS start side, from nowhere.
//
// A tree edge (parent(w), w) of the rooted forest is a bridge of H iff no non-tree link has exactly one end inside
// subtree(w).  Every non-tree link gets a 64-bit hash of its local edge idx; a side's value is the xor of the hashes
// of its non-tree links, a segment's value (both sides) sits at the tour position of the arc that enters it (0 at all
// other positions); the xor over a stretch of the tour -- two look-ups into the running xor over the positions --
// cancels every link with both ends inside and keeps the ones that cross:
//   hooked link into w:   the stretch [tin, tout] of w's segment;
//   black edge w -> w^1:  hside[w^1] and the stretch of the subtrees hanging off w^1, which the slot order keeps
//                         together: from the first arc of w^1 up to the next arc of the segment behind them (the first
//                         arc of w if that is not the entering one, else the arc that leaves the segment).
// A bridge always xors to 0; a non-bridge xors to 0 only if the hashes of its crossing links cancel by accident
// (probability 2^-64 per tree edge, i.e. ~1e-11 per pass over 2e8 tree edges; the hash is a fixed function of the edge
// idx, so a result is reproducible).  This replaces two range-min queries per side over segment trees of the far ends'
// pre-order numbers, and the forest needs no pre-order numbering at all.
static constexpr uint32_t T0_RBIT = 0x80000000u;
__global__ void k_t0_parents(uint32_t V, const uint32_t *__restrict__ dist, const uint32_t *__restrict__ ckey,
			     const uint32_t *__restrict__ voff, const uint32_t *__restrict__ loff, const uint32_t *__restrict__ ladj,
			     const uint32_t *__restrict__ lle,
			     const uint32_t *__restrict__ ft, const uint32_t *__restrict__ heads,
			     uint4 *__restrict__ t0seg, uint4 *__restrict__ xrec, uint32_t C,
			     const unsigned long long *__restrict__ start_key, uint32_t *__restrict__ err, uint8_t *__restrict__ multi)
{
	uint32_t g = BIDX * blockDim.x + threadIdx.x;
	if (g < V) // (the flags k_bridges raises, cleared here on the way: no fill kernel over 2 * 10^8 bytes)
		*reinterpret_cast<uint16_t *>(multi + 2 * g) = 0;
	if (g < C) { // the DFS start of component g roots its tree (no advance arc ever enters its segment)
		const uint32_t r = comp_root_side(start_key, voff, g), L = 2 * (voff[g + 1] - voff[g] - 1);
		t0seg[r >> 1] = make_uint4(NIL, (r & 1u) ? T0_RBIT : 0u, 0u, L ? L - 1 : 0u);
		// the tour of the component covers all its arcs iff the hooks of the union-find are a spanning tree
		const uint32_t h = heads[g];
		if (h == NIL ? L != 0 : dist[h] != L - 1)
			atomicExch(err, 1u);
	}
	if (g >= V)
		return;
	// One lane per SEGMENT, from the child's end.  The tour enters a segment, takes its other forest slots in cyclic order
	// (every one followed by the whole tour of the subtree behind it) and leaves through the slot it came in by: of the
	// segment's forest slots the one back to the PARENT is the last in the tour (fewest arcs behind it), and the arc that
	// enters the segment stands right in front of the first of them (one more arc behind it than behind that one).  So the
	// distances of the segment's own slots -- one contiguous stretch -- say everything; no slot of another segment is
	// looked at (until round 5 the twin of every forest slot was stored by k_tour_words and its distance gathered here:
	// 1 GB written, 1 GB read and a scattered 4-byte gather per forest slot on the whole-genome workload).
	const uint32_t sb = loff[2 * g], sm = loff[2 * g + 1], se = loff[2 * g + 2];
	uint32_t at_e = NIL, dmin = 0xFFFFFFFFu, dmax = 0u;
	for (uint32_t at0 = sb; at0 < se; at0 += 4) {
		const uint4 lw4 = load4_unaligned(lle + at0), d4 = load4_unaligned(dist + at0);
		const uint32_t lws[4] = {lw4.x, lw4.y, lw4.z, lw4.w}, ds[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
		for (uint32_t q = 0; q < 4; q++) {
			if (at0 + q >= se || !(lws[q] & LLE_TREE))
				continue;
			if (ds[q] < dmin) {
				dmin = ds[q];
				at_e = at0 + q;
			}
			dmax = max(dmax, ds[q]);
		}
	}
	if (at_e == NIL)
		return; // a component of one segment: its record was written above
	const uint32_t c = ckey[g];
	if ((comp_root_side(start_key, voff, c) >> 1) == g)
		return; // the root segment: all its slots lead to children (record written by the lane of its component)
	const uint32_t L = 2 * (voff[c + 1] - voff[c] - 1), abase = 2 * (voff[c] - c);
	const uint32_t p_in = abase + (L - 2 - dmax), p_out = abase + (L - 1 - dmin); // tour positions of the arc in and the arc back
	t0seg[g] = make_uint4(ladj[at_e], (lle[at_e] & LLE_ID) | (at_e >= sm ? T0_RBIT : 0u), p_in, p_out);
	// Most segments have no non-tree link, so the values are kept COMPACT (in tour order, only where a side carries
	// one): here only the bit of the tour position is set; k_tour_values drops the value at its rank among the set bits.
	const uint2 f2 = *reinterpret_cast<const uint2 *>(ft + 2 * g);
	if ((f2.x | f2.y) & FT_HASH)
		atomicOr(reinterpret_cast<unsigned long long *>(&xrec[p_in >> 6]), 1ull << (p_in & 63u));
}
// The running xor is only ever read at tour positions, but only the positions of segments with non-tree links carry a
// value (about one in six of a pangenome graph): the values sit compact, in tour order, behind a bitmap of the positions
// (25 MB for 2e8 positions: cache resident) with the count of set bits in front of every word.
//   rank(p)  = set bits at positions < p;   running xor in front of position p = xps[rank(p)]
__global__ void k_bit_counts(uint32_t W, const uint4 *__restrict__ xrec, uint32_t *__restrict__ xrank)
{
	uint32_t w = BIDX * blockDim.x + threadIdx.x;
	if (w <= W)
		xrank[w] = w < W ? (uint32_t)(__popc(xrec[w].x) + __popc(xrec[w].y)) : 0u;
}
__global__ void k_bit_ranks(uint32_t W, const uint32_t *__restrict__ xrank, uint4 *__restrict__ xrec)
{
	uint32_t w = BIDX * blockDim.x + threadIdx.x;
	if (w < W)
		xrec[w].z = xrank[w];
}
// (one 16-byte load: the word of the bitmap and the count in front of it sit together)
__device__ __forceinline__ uint32_t tour_rank(const uint4 *__restrict__ xrec, uint32_t p)
{
	const uint4 r = xrec[p >> 6];
	const unsigned long long bits = (unsigned long long)r.x | ((unsigned long long)r.y << 32);
	return r.z + (uint32_t)__popcll(bits & ((1ull << (p & 63u)) - 1ull));
}
// one lane per segment (all reads in segment order): the value of a segment whose position bit is set
// (two segments a lane -- one 16-byte load of their four ft words -- measured SLOWER: 1.22 ms against 1.00)
__global__ void k_tour_values(uint32_t V, const uint4 *__restrict__ t0seg, const uint32_t *__restrict__ ft,
			      const ulonglong2 *__restrict__ hside, const uint4 *__restrict__ xrec, ulonglong2 *__restrict__ xval)
{
	uint32_t g = BIDX * blockDim.x + threadIdx.x;
	if (g >= V)
		return;
	const uint2 f2 = *reinterpret_cast<const uint2 *>(ft + 2 * g);
	if (!((f2.x | f2.y) & FT_HASH))
		return;
	const uint4 r = t0seg[g];
	if (r.x == NIL) // a root is entered by no arc: its value is in no stretch
		return;
	const ulonglong2 z = make_ulonglong2(0ull, 0ull);
	xval[tour_rank(xrec, r.z)] = hx((f2.x & FT_HASH) ? hside[2 * g] : z, (f2.y & FT_HASH) ? hside[2 * g + 1] : z);
}
// parent word of side S: its parent in the rooted forest, bit 31 set when the edge to it is a bridge (NIL: S roots its tree).
// It lives in cstate[S] (below) next to the visited bit -- a separate array of them was 0.8 GB written and read once more.
static constexpr uint32_t PB_BRIDGE = 0x80000000u;
// Also, while parent and bridge bit of a side are in hand, what the class walks start from (section 5): the side's state word
// cstate (forest parent, bridge bit, visited bit: one load tells a walk "not yet visited", which covers "not across a
// bridge" on the way down) and, for an ENTRY -- a root or the lower end of a bridge: the one side of its class the DFS
// reaches first --, its DFS record dps = {parent across the bridge, scan slot of the parent it is found through}.
static constexpr uint32_t CLASS_BUDGET = 256;  // sides a lane walks before it hands its class to the big-class walk
static constexpr uint32_t CS_VISITED = 0x40000000u; // (side ids stay below 2^29, bit 31 is PB_BRIDGE)
// One lane per SEGMENT: the side a segment is entered through and its far side ask different questions (the hooked link
// into the segment / the black edge across it), so a lane per side left half of every wave idle in either branch; a lane
// per segment answers both, loads the segment's record once and stores the words of its two sides together.
__global__ void k_bridges(uint32_t V, const uint4 *__restrict__ t0seg, const ulonglong2 *__restrict__ xps,
			  const uint4 *__restrict__ xrec,
			  const ulonglong2 *__restrict__ hside, const uint32_t *__restrict__ ft,
			  const uint32_t *__restrict__ dist, const uint32_t *__restrict__ loff, const uint32_t *__restrict__ lle,
			  const uint32_t *__restrict__ ckey, const uint32_t *__restrict__ cproc, const uint32_t *__restrict__ voff,
			  uint8_t *multi, uint32_t *__restrict__ cstate, uint2 *__restrict__ dps)
{
	const uint32_t g = BIDX * blockDim.x + threadIdx.x;
	if (g >= V)
		return;
	// multi[S] = 1: side S shares its 2-edge-connected class with another side, i.e. some tree edge at S is no bridge
	// (cleared by the caller).  Most sides of a pangenome graph sit on bridges only and are classes of their own; those
	// need no walk at all.
	// xor of the values at the tour positions [p, q): two look-ups into the running xor -- and none at all when no position in
	// between carries a value (the ranks of both ends agree: the subtree of most segments holds no non-tree link)
	auto stretch = [&](uint32_t p, uint32_t q) {
		const uint32_t rp = tour_rank(xrec, p), rq = tour_rank(xrec, q);
		return rp == rq ? make_ulonglong2(0ull, 0ull) : hx(xps[rp], xps[rq]);
	};
	const uint4 r = t0seg[g];
	const uint32_t Se = (2 * g) | (r.y >> 31), Sf = Se ^ 1u, c = ckey[g]; // entered side, far side
	const bool proc = cproc[c] != 0; // (components that are not decomposed here get inert words)
	// ---- the entered side: the hooked link into the segment is a bridge iff nothing crosses the segment's stretch of the tour
	uint32_t pvE; // parent | bridge bit
	if (r.x == NIL) {
		pvE = NIL;
	} else if (hzero(stretch(r.z, r.w + 1))) {
		pvE = r.x | PB_BRIDGE;
	} else {
		pvE = r.x;
		multi[Se] = 1;
		multi[r.x] = 1;
	}
	// ---- the far side, across the black edge: what leaves its subtree = its own non-tree links and those of the subtrees
	// hanging off it
	const uint2 f2 = *reinterpret_cast<const uint2 *>(ft + 2 * g);
	const uint32_t fF = (Sf & 1u) ? f2.y : f2.x, fE = (Sf & 1u) ? f2.x : f2.y, a1 = fF & FT_NONE;
	ulonglong2 x = (fF & FT_HASH) ? hside[Sf] : make_ulonglong2(0ull, 0ull);
	if (a1 != FT_NONE) {
		const uint32_t L = 2 * (voff[c + 1] - voff[c] - 1), abase = 2 * (voff[c] - c);
		const uint32_t a3 = fE & FT_NONE;
		uint32_t end; // position behind the last of them
		if (r.x == NIL) // root segment: the start side's arcs come first, the far side's are the rest of the tour
			end = abase + L;
		else if (a3 != FT_NONE && (lle[a3] & LLE_ID) != (r.y & ~T0_RBIT)) // arcs of the entered side in front of the entering one follow
			end = abase + (L - 1 - dist[a3]);
		else
			end = r.w;
		x = hx(x, stretch(abase + (L - 1 - dist[a1]), end));
	}
	uint32_t pvF;
	if (hzero(x)) {
		pvF = Se | PB_BRIDGE;
	} else {
		pvF = Se;
		multi[Sf] = 1;
		multi[Se] = 1;
	}
	// ---- the words of both sides, stored together
	const bool entryE = proc && (pvE & PB_BRIDGE), entryF = proc && (pvF & PB_BRIDGE); // a root (NIL) or the lower end of a bridge
	const uint32_t csE = proc ? (pvE | (entryE ? CS_VISITED : 0u)) : NIL, csF = proc ? (pvF | (entryF ? CS_VISITED : 0u)) : NIL;
	uint2 recE = make_uint2(NIL, 0u), recF = make_uint2(NIL, 0u); // {DFS parent, scan slot it was found through}
	if (entryE && pvE != NIL) { // (NIL: DFS start of the component) a gray bridge: its slot in the parent's list (ascending link id)
		const uint32_t p = pvE & ~PB_BRIDGE;
		recE.x = p;
		uint32_t le = r.y & ~T0_RBIT, lo = loff[p], hi = loff[p + 1];
		while (lo < hi) {
			const uint32_t mid = (lo + hi) >> 1;
			if ((lle[mid] & LLE_ID) < le)
				lo = mid + 1;
			else
				hi = mid;
		}
		recE.y = lo - loff[p] + 1;
	}
	if (entryF)
		recF.x = Se; // (the black edge: slot 0, scanned first)
	const bool e_first = !(Se & 1u);
	*reinterpret_cast<uint2 *>(cstate + 2 * g) = e_first ? make_uint2(csE, csF) : make_uint2(csF, csE);
	*reinterpret_cast<uint4 *>(dps + 2 * g) = e_first ? make_uint4(recE.x, recE.y, recF.x, recF.y) : make_uint4(recF.x, recF.y, recE.x, recE.y);
}

// ------------------------------------------------------------------ 4. 2-edge-connected classes
// The 2-edge-connected classes are the pieces the rooted forest falls into when its bridges are cut (every bridge of H is
// a tree edge, and a class stays connected inside any spanning tree).  They are never numbered: two neighbouring sides
// are in one class iff the edge between them is no bridge, and a bridge is recognised from pbr alone -- it is the tree
// edge between a side and its forest parent with the bridge bit set (a bridge has no parallel edge, so "the edge to my
// parent" and "a link to my parent" are the same thing when the bit is set).  Every class is walked from its ENTRY,
// the one side whose parent edge is a bridge (or the DFS start); entries are marked visited before any walk starts,
// so a walk never runs down across a bridge, and only the entry itself has to skip the way up.

// ------------------------------------------------------------------ 5. class entries
// a class is walked from its entry side (k_bridges marked the entries visited and wrote their DFS records); a side that
// is alone in its class has nothing to walk: the walks start from the entries that share their class (multi)
// (four sides a lane: one 16-byte, one 4-byte and one 8-byte load instead of twelve 4- and 1-byte ones -- a kernel of a few
// loads per element is bound by the number of memory instructions its CU can retire, not by their bytes)
static constexpr uint32_t EL_ITER = LIST_ITER, EL_SIDES = LIST_SPAN; // sides a workgroup of k_entry_list lists: one atomic add for 16 384 of them
static_assert(TPB == (int)LIST_TPB, "append_in_order is written for workgroups of 256");
__global__ void __launch_bounds__(TPB) k_entry_list(uint32_t nS, const uint32_t *__restrict__ pbr /* = cstate: only the bridge bit is read */,
						     const uint8_t *__restrict__ multi, const uint32_t *__restrict__ ckey,
						     const uint32_t *__restrict__ cproc, uint32_t *__restrict__ entry_list,
						     uint32_t *__restrict__ n_entry)
{
	// The entries go straight onto the list the walks start from, in side order (append_in_order, common.hpp: one atomic add
	// per workgroup of 16 384 sides).  Until round 5: a flag byte per side, written and then read twice by a count / scan /
	// write compaction (four launches).
	const uint32_t B0 = BIDX * EL_SIDES;
	unsigned long long fw = 0; // bit 4 it + j: side B0 + it * 1024 + 4 tid + j is an entry -- a root (NIL: all bits set) or the lower end of a bridge, in a component that is decomposed here
#pragma unroll
	for (uint32_t it = 0; it < EL_ITER; it++) {
		const uint32_t S0 = B0 + it * (TPB * 4u) + threadIdx.x * 4u;
		uint32_t f = 0;
		if (S0 + 4 <= nS) {
			const uint4 pb = *reinterpret_cast<const uint4 *>(pbr + S0);
			const uint32_t mu = *reinterpret_cast<const uint32_t *>(multi + S0);
			const uint2 ck = *reinterpret_cast<const uint2 *>(ckey + (S0 >> 1));
			const uint32_t c0 = cproc[ck.x] ? 1u : 0u, c1 = cproc[ck.y] ? 1u : 0u;
			f = (((mu & 0xFFu) && (pb.x & PB_BRIDGE)) ? c0 : 0u) | ((((mu & 0xFF00u) && (pb.y & PB_BRIDGE)) ? c0 : 0u) << 1) |
			    ((((mu & 0xFF0000u) && (pb.z & PB_BRIDGE)) ? c1 : 0u) << 2) | ((((mu & 0xFF000000u) && (pb.w & PB_BRIDGE)) ? c1 : 0u) << 3);
		} else {
			for (uint32_t S = S0; S < nS; S++)
				f |= ((multi[S] && (pbr[S] & PB_BRIDGE) && cproc[ckey[S >> 1]]) ? 1u : 0u) << (S - S0);
		}
		fw |= (unsigned long long)f << (4 * it);
	}
	append_in_order(fw, B0, entry_list, n_entry);
}

// ------------------------------------------------------------------ 6. the DFS inside every class
// Small classes (a bubble: a handful of sides) are walked by ONE LANE each, millions at a time (k_class_dfs_small).
// A class that turns out larger than CLASS_BUDGET sides is handed to the WAVE-COOPERATIVE walk below, one wave per
// class: lexicographic DFS is sequential, so the time of a large class is (sides) x (dependent load levels per side)
// and the wave exists to cut the levels, not to share the sides.
//
// k_class_recs filters every side's scan list [black edge, links by local edge idx] down to the neighbours of its own
// class (those not across a bridge) and packs a 32-byte record {count, overflow begin, first six candidates}.
//  * going DOWN costs one level: the lanes load the visited word AND the record of every candidate of the current side at
//    once, a ballot picks the first unvisited one, its record already sits in a lane.  Candidates that are visited
//    already -- most of them, deep inside a tangle -- cost nothing (the one-lane walk paid a round trip for each).
//  * a side is only remembered on the stack when the ballot showed ANOTHER unvisited candidate behind the chosen one
//    (visited is monotone: a side without one never needs a second look), so most returns never happen.
//  * coming BACK pops up to 64 stack entries at once: every lane re-checks the remaining candidates of one entry, a
//    ballot finds the nearest entry that still has an unvisited candidate.
// The top 64 entries of the stack live in LDS together with their records; older ones move out to an array in HBM
// (chunks that double in size, taken from one pool with an atomic add, so a class needs no size in advance: a stack
// that ever held d > 64 entries belongs to a class of more than d sides and takes less than 64 + 2 d entries of the pool,
// under three per side).
static constexpr uint32_t W_UNVIS = 0xFFFFFFFEu;  // wpar: side not reached yet
static constexpr uint32_t W_INLINE = 6;		  // candidates inside the record
static constexpr uint32_t W_CHUNK0 = 64;	  // first chunk of a stack's HBM array (entries); chunk c >= 1 holds W_CHUNK0 << (c - 1)
struct WalkRec { // 32 bytes
	uint32_t n, begin, c[W_INLINE];
};
__global__ void k_class_recs(uint32_t nS, const uint32_t *__restrict__ loff, const uint32_t *__restrict__ ladj,
			     const uint32_t *__restrict__ pbr, const uint32_t *__restrict__ ckey,
			     const uint32_t *__restrict__ cproc, uint32_t *__restrict__ wadj, uint4 *__restrict__ wrec,
			     uint32_t *__restrict__ wpar)
{
	uint32_t u = BIDX * blockDim.x + threadIdx.x;
	if (u >= nS)
		return;
	const uint32_t lo = loff[u], hi = loff[u + 1], base = lo + u; // deg + 1 slots per side
	uint32_t n = 0, c[W_INLINE] = {NIL, NIL, NIL, NIL, NIL, NIL};
	uint32_t vis = W_UNVIS;
	if (cproc[ckey[u >> 1]]) {
		const uint32_t mine = pbr[u] & ~CS_VISITED; // (pbr = cstate: the small walks have set visited bits by now)
		if (mine & PB_BRIDGE)
			vis = 0u; // the entry of its class (or a DFS start): never walked into
		auto same_class = [&](uint32_t o) { return o != u && mine != (o | PB_BRIDGE) && (pbr[o] & ~CS_VISITED) != (u | PB_BRIDGE); }; // no bridge between
		auto put = [&](uint32_t o) {
			if (n < W_INLINE) {
#pragma unroll
				for (uint32_t k = 0; k < W_INLINE; k++)
					if (k == n)
						c[k] = o;
			} else {
				if (n == W_INLINE) // the list spills: the overflow array holds ALL candidates, in order
					for (uint32_t k = 0; k < W_INLINE; k++)
						wadj[base + k] = c[k];
				wadj[base + n] = o;
			}
			n++;
		};
		if (same_class(u ^ 1u))
			put(u ^ 1u);
		for (uint32_t k = lo; k < hi; k++) {
			const uint32_t o = ladj[k];
			if (same_class(o))
				put(o);
		}
	}
	wrec[2 * (size_t)u] = make_uint4(n, base, c[0], c[1]);
	wrec[2 * (size_t)u + 1] = make_uint4(c[2], c[3], c[4], c[5]);
	wpar[u] = vis;
}
// the plain walk (small classes only; no filtering pass).  Dependent loads per tree edge: the candidate's adjacency
// entry and its state word on the way down; {parent, slot} of the finished side and the parent's list bounds on the way
// back -- the parent resumes its scan behind the slot the child was found through, so no cursor is stored.
static constexpr uint32_t DFS_STK = 8; // levels of a walk whose scan state stays in LDS (most classes are bubbles a few sides deep)
__global__ void __launch_bounds__(64) k_class_dfs_small(const uint32_t *__restrict__ n_entry_dev, const uint32_t *__restrict__ entry_list,
							 const uint32_t *__restrict__ loff, const uint32_t *__restrict__ ladj,
							 uint32_t *__restrict__ cstate, uint2 *__restrict__ dps, uint32_t budget,
							 uint32_t *__restrict__ n_over, uint32_t *__restrict__ over_list)
{
	// the scan state {side, next slot, list begin, list length} of the upper levels of a walk: coming back to them costs no
	// memory traffic at all (deeper levels go through dps and loff again)
	__shared__ uint4 stk[DFS_STK][64];
	const uint32_t lane = threadIdx.x;
	// grid-stride over the classes (entry_list is in side order): the lanes in flight work on one window of sides
	const uint32_t n_entry = *n_entry_dev; // (the count only exists on the device: no read-back sizes this launch)
	for (uint32_t i = BIDX * blockDim.x + threadIdx.x; i < n_entry; i += gridDim.x * blockDim.x) {
		const uint32_t s = entry_list[i];
		const uint32_t s_up = cstate[s] & ~(CS_VISITED | PB_BRIDGE); // the entry's parent, across its bridge (root: all ones)
		uint32_t u = s, k = 0, lo = loff[u], n = loff[u + 1] - lo, sides = 0, depth = 0;
		for (;;) {
			bool adv = false;
			while (k <= n) {
				// The next four candidates at once (slot 0 is the black neighbour, slots 1.. the links): their list words in
				// one 16-byte load, their state words in four independent gathers -- one round trip for all of them, where a
				// probe at a time paid two per candidate that turned out visited.  Nobody else writes the states of this
				// class, and a side that is scanned again after a return loads them afresh.
				const uint32_t rem = n + 1 - k;
				uint32_t o0, o1, o2, o3;
				if (k == 0) {
					const uint4 x = load4_unaligned(ladj + lo);
					o0 = u ^ 1u, o1 = x.x, o2 = x.y, o3 = x.z;
				} else {
					const uint4 x = load4_unaligned(ladj + lo + k - 1);
					o0 = x.x, o1 = x.y, o2 = x.z, o3 = x.w;
				}
				const uint32_t w0 = cstate[o0];
				const uint32_t w1 = rem > 1 ? cstate[o1] : CS_VISITED;
				const uint32_t w2 = rem > 2 ? cstate[o2] : CS_VISITED;
				const uint32_t w3 = rem > 3 ? cstate[o3] : CS_VISITED;
				// not visited yet (hence no entry of another class), not the way up
				auto open = [&](uint32_t o, uint32_t w) { return !(w & CS_VISITED) && !(u == s && o == s_up); };
				const uint32_t j = open(o0, w0) ? 0u : open(o1, w1) ? 1u : open(o2, w2) ? 2u : open(o3, w3) ? 3u : 4u;
				if (j == 4) {
					k += min(rem, 4u);
					continue;
				}
				const uint32_t o = j == 0 ? o0 : j == 1 ? o1 : j == 2 ? o2 : o3, w = j == 0 ? w0 : j == 1 ? w1 : j == 2 ? w2 : w3;
				const uint32_t slot = k + j;
				k = slot + 1;
				// (loads of the next step before the stores of this one: vmcnt counts both in issue order,
				// and a wait for a load behind a store waits for the store too)
				// BLACK FOLLOW-THROUGH: a step over a link is nearly always followed by the step over the black edge of the
				// segment it arrives at (slot 0 of the new side).  The list bounds of both sides of that segment are three
				// consecutive words and their state words two: fetched together on arrival, and when the partner turns out
				// unvisited both tree edges are made here -- three dependent round trips for the pair where one edge at a
				// time took six, and the partner's state word is not gathered again with the candidates.
				const uint32_t g2 = o & ~1u, odd = o & 1u;
				uint32_t l0, l1, l2 = 0, wp = CS_VISITED;
				if (slot == 0) { // came over the black edge: the partner is u, visited
					l0 = loff[o], l1 = loff[o + 1];
				} else { // (the three offsets in one load instruction: the array carries slack behind its last word)
					const uint4 l4 = load4_unaligned(loff + g2);
					l0 = l4.x, l1 = l4.y, l2 = l4.z;
					const uint2 cs2 = *reinterpret_cast<const uint2 *>(cstate + g2);
					wp = odd ? cs2.x : cs2.y;
				}
				const uint32_t nlo = (slot == 0 || !odd) ? l0 : l1, nhi = (slot == 0) ? l1 : (odd ? l2 : l1);
				if (depth < DFS_STK)
					stk[depth][lane] = make_uint4(u, k, lo, n);
				depth++;
				sides++;
				// (the black neighbour of the new side is known to be visited from here on: its scan starts at slot 1)
				if (!(wp & CS_VISITED)) { // an unvisited partner is in this class (across a bridge it would be an entry: marked)
					// both sides of the segment at once: their state words in one 8-byte store, their records in one of 16
					const uint32_t p = o ^ 1u, plo = odd ? l0 : l1, phi = odd ? l1 : l2;
					const uint32_t so = w | CS_VISITED, sp = wp | CS_VISITED;
					*reinterpret_cast<uint2 *>(cstate + g2) = odd ? make_uint2(sp, so) : make_uint2(so, sp);
					*reinterpret_cast<uint4 *>(dps + g2) = odd ? make_uint4(o, 0u, u, slot) : make_uint4(u, slot, o, 0u);
					if (depth < DFS_STK)
						stk[depth][lane] = make_uint4(o, 1u, nlo, nhi - nlo);
					depth++;
					sides++;
					u = p;
					lo = plo;
					n = phi - plo;
				} else {
					cstate[o] = w | CS_VISITED;
					dps[o] = make_uint2(u, slot);
					u = o;
					lo = nlo;
					n = nhi - nlo;
				}
				k = 1;
				adv = true;
				break;
			}
			if (adv) {
				if (sides > budget) { // a large class: the walk with the short dependent chain starts it over (k_class_dfs)
					over_list[atomicAdd(n_over, 1u)] = s;
					break;
				}
				continue;
			}
			if (u == s)
				break;
			depth--;
			if (depth < DFS_STK) {
				const uint4 r = stk[depth][lane];
				u = r.x, k = r.y, lo = r.z, n = r.w;
			} else {
				const uint2 up = dps[u];
				k = up.y + 1;
				u = up.x;
				lo = loff[u];
				n = loff[u + 1] - lo;
			}
		}
	}
}
// ---- the wave-cooperative walk (see the head of this section)
struct WalkStackEntry { // one remembered side: 40 bytes in LDS, {u, j} in HBM
	uint32_t u, j; // side, first candidate index still to look at
	uint4 r0, r1;  // its record
};
__device__ __forceinline__ uint32_t wstk_addr(uint32_t d, const uint32_t *chunk_base)
{
	const uint32_t c = d < W_CHUNK0 ? 0u : 32u - (uint32_t)__clz(d / W_CHUNK0);
	return chunk_base[c] + (d - (c ? (W_CHUNK0 << (c - 1)) : 0u));
}
__device__ __forceinline__ uint32_t wrec_cand(const uint4 &r0, const uint4 &r1, uint32_t k)
{
	return k == 0 ? r0.z : k == 1 ? r0.w : k == 2 ? r1.x : k == 3 ? r1.y : k == 4 ? r1.z : r1.w;
}
// A WINDOW of the side index space lives in LDS beside the stack: the records and the parent words of W_WIN consecutive
// sides.  The sides of a class mostly are neighbours in the index space too (a pangenome graph is numbered along its
// backbone), and a walk that moves on to a neighbouring index finds the candidate's visited word AND its record in LDS: no
// round trip to memory at all.  The window is refilled, by all 64 lanes with 16-byte loads, when the walk steps outside
// it; a class whose links lead all over the index space (the tangled workload) gets few hits per refill and backs off to
// refilling rarely.  This wave is the only writer of the parent words of its class and mirrors every write into the
// window, so the copy in LDS never goes stale for the sides it is asked about.
//
// What bounds the walk when thousands of classes are walked at once (BASELINE config 5: 3 333 classes of 4 002 sides, a
// wave each, 13 waves per CU) is neither memory latency nor bandwidth but INSTRUCTION ISSUE: three lanes of a wave have
// work, every instruction of the step is issued for all 64, and 13 waves share a CU's issue slots.  Measured there with a
// cycle counter per kind of step (build with EXTRA=-DPOVU_WALK_STATS): ~1 300 cycles a step, with or without a load in it,
// with one or three dependent LDS reads in it.  What did change the time: see the last point.  The step:
//  * what is the same in all lanes (current side and its record, stack depth, window) lives in scalar registers (readlane);
//  * a lane fetches the visited word AND the record of its candidate in one go (out of the window, or out of memory in one
//    round trip): the chosen lane then holds the next step's record -- the one dependent access of a step.  On a class that
//    is walked alone (the tangled workload: one wave, latency-bound) every further dependent LDS read cost 8 %;
//  * the stack remembers {side, next candidate} only -- eight bytes a push; a pop fetches the record again, 64 entries at once;
//  * no store in the step: on gfx950 a wait for ANY vector-memory result also waits for every older store of the wave
//    (one counter, in issue order), so a store per step makes every later wait as long as a trip to memory.  The parent
//    words written into the window go out together when the window moves on (one dirty bit per slot), and the stack
//    spills half of its LDS cache at a time.
static constexpr uint32_t W_WIN = 128;	// sides in the window: 4.6 KB of LDS per wave
static constexpr uint32_t W_WIN_BACK = 16; // sides kept behind the side the window is refilled at
__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t lane_val(uint32_t v, int l) { return __builtin_amdgcn_readlane(v, l); }
struct URec { // a side's record in scalar registers
	uint32_t n, begin, c0, c1, c2, c3, c4, c5;
};
__device__ __forceinline__ URec urec_lane(const uint4 a, const uint4 b, int l)
{
	return URec{lane_val(a.x, l), lane_val(a.y, l), lane_val(a.z, l), lane_val(a.w, l),
		    lane_val(b.x, l), lane_val(b.y, l), lane_val(b.z, l), lane_val(b.w, l)};
}
// Two kernels (HOT = with the hot loop in front of the general step / the general step alone, see the loop at the end), launched
// side by side over the same list of classes: every wave works out from the records around the entry which of the two forms
// its class gets, and the wave of the other kernel ends at once.  One kernel with both forms made the general step of a class
// that never uses the hot loop 8 - 25 % slower (the step is bound by its dependent instructions and by where the compiler
// puts them; 20 scalar registers spilled).
template <bool HOT>
__global__ void __launch_bounds__(64) k_class_walk_wave(uint32_t n_entry, const uint32_t *__restrict__ entry_list,
							 const uint4 *__restrict__ wrec, const uint32_t *__restrict__ wadj,
							 uint32_t *wpar, uint2 *wstk, uint32_t *__restrict__ pool_top, uint32_t pool_cap,
							 uint32_t *__restrict__ err, uint32_t nS, uint32_t route)
{
	__shared__ uint4 win_rec[W_WIN][2]; // {n, overflow begin, c0, c1}, {c2 .. c5}
	__shared__ uint32_t win_par[W_WIN];
	__shared__ uint2 ring[64]; // the top of the stack: {side, first candidate index still to look at}
	__shared__ uint32_t chunk_base[32];
	const uint32_t lane = threadIdx.x;
	if (blockIdx.x >= n_entry)
		return;
	uint32_t u = sgpr(entry_list[blockIdx.x]);
	URec r;
	{
		const uint4 a = wrec[2 * (size_t)u], b = wrec[2 * (size_t)u + 1];
		r = urec_lane(a, b, 0);
	}
	// Which form: do neighbours in the class sit near one another in the index space (a chain of bubbles, a tower, a ring: the
	// window answers most steps) or all over it (a tangle whose links were written in random order)?  A probe the way the walk
	// will go -- over a link to a side, across its segment, over the first link there, ... -- W_PROBE hops from the entry (a
	// dependent load each: some 40 us, once per class): "near" = the link leads within 64 sides.  Three hops in four near are enough
	// for the hot loop (it leaves by itself when it does not pay, W_TRIAL below); the entry's own neighbourhood says little (a
	// tangle is entered from the backbone, a ring from the link that closes it).  route: 0 = by the probe, 1 = every class with
	// the hot loop, 2 = every class without.
	{
		constexpr uint32_t W_PROBE = 48;
		uint32_t near = 0, tot = 0, x = u;
		uint4 a = make_uint4(r.n, r.begin, r.c0, r.c1);
		for (uint32_t h = 0; h < W_PROBE && route == 0; h++) {
			// first candidate of x that is not its segment partner (the partner, when it is in the class, is candidate 0)
			const uint32_t y = a.x == 0 ? NIL : (a.z != (x ^ 1u) ? a.z : (a.x > 1 ? a.w : NIL));
			if (y == NIL)
				break;
			tot++;
			near += (y - x + 64u) <= 128u ? 1u : 0u;
			x = y ^ 1u;
			a = wrec[2 * (size_t)x]; // (uniform: every lane reads the same record)
		}
		const bool windowy = route == 1 || (route == 0 && tot == W_PROBE && 4 * near >= 3 * tot); // (a probe that ends early -- a dead end, a small class -- says nothing: the form without)
#ifdef POVU_WALK_STATS
		if (lane == 0 && HOT)
			printf("route %u: entry %u n %u near %u tot %u -> %s\n", blockIdx.x, u, r.n, near, tot, windowy ? "hot" : "plain");
#endif
		if (windowy != HOT)
			return; // (the other kernel's wave walks this class)
	}
	uint32_t j0 = 0;
	// BLACK FOLLOW-THROUGH.  In the biedged graph nearly every step over a link is followed by the step over the black edge of the
	// segment it arrives at: the partner side c ^ 1 is the FIRST candidate of c's list.  The records of c and c ^ 1 share a
	// 64-byte line (and their parent words 8 bytes), so the step that looks at a candidate fetches its partner's words with
	// it; when the walk then moves to c and the partner is unvisited, the black step is taken straight from these registers --
	// two tree edges per dependent round trip instead of one (the tangled workload walks one class of 6 * 10^5 sides at one
	// Infinity-Cache round trip a step; BASELINE config 5 is bound by the instructions a step issues: half the steps).
	bool have_pn = false; // pn / pn_vis hold the record and the parent word of u ^ 1, as of the step that led to u
	uint32_t pn_vis = 0;
	// (the kernel with the hot loop is short of scalar registers: there the partner's record stays in the lanes' registers --
	// lane pn_lane holds it -- and becomes scalars only when the black step is taken; the other kernel keeps it as scalars)
	URec pn{};
	uint4 pq0 = make_uint4(0, 0, 0, 0), pq1 = pq0;
	int pn_lane = 0;
	uint32_t depth = 0, lds_lo = 0, n_chunks = 0; // stack entries; first entry cached in LDS; chunks taken from the pool
	uint32_t win_lo = 0xFFFFFF00u; // (no window yet: no side id comes within W_WIN of this)
	bool have_win = false;
	uint32_t win_hits = 0, win_wait = 0, win_penalty = 0; // hits since the last refill; steps until the next refill is allowed
	// A parent word the window holds but memory does not yet carries bit 31 in its LDS copy (side ids stay below 2^29; the
	// "not reached" and "no word" patterns have the bit set too and are told apart by value): one store instead of the two
	// 64-bit bitmaps the step used to update.  Every reader of the window compares with W_UNVIS only.
	constexpr uint32_t W_DIRTY = 0x80000000u;
	auto flush_window = [&]() {
		for (uint32_t k = lane; k < W_WIN; k += 64) {
			const uint32_t w = win_par[k];
			if ((w & W_DIRTY) && w < W_UNVIS) {
				wpar[win_lo + k] = w & ~W_DIRTY;
				win_par[k] = w & ~W_DIRTY;
			}
		}
	};
#ifdef POVU_WALK_STATS
	uint32_t st_fast = 0, st_slow = 0, st_refill = 0, st_pop = 0, st_ft = 0;
	const long long st_t0 = clock64();
	long long st_mark = st_t0, cy_fast = 0, cy_slow = 0, cy_refill = 0, cy_pop = 0, cy_ft = 0;
#define WSTAT(x) (x)++
#define WCYC(acc)                                                                                                             \
	do {                                                                                                                  \
		const long long now__ = clock64();                                                                            \
		acc += now__ - st_mark;                                                                                       \
		st_mark = now__;                                                                                              \
	} while (0)
#define WSTAT_DONE()                                                                                                          \
	do {                                                                                                                  \
		if (lane == 0 && (blockIdx.x < 2 || st_fast + st_slow > 50000u))                                              \
			printf("walk %u: fast %u slow %u refills %u pops %u black-follow %u cycles %lld = fast %lld slow %lld refill %lld pop %lld black-follow %lld\n", blockIdx.x, st_fast, st_slow, st_refill, st_pop, st_ft, \
			       clock64() - st_t0, cy_fast, cy_slow, cy_refill, cy_pop, cy_ft);                                          \
	} while (0)
#else
#define WSTAT(x)
#define WCYC(acc)
#define WSTAT_DONE()
#endif
	bool pool_ok = true;
	bool pend = false, pend_push = false; // bookkeeping of a black follow-through step that the next step still has to carry out
	uint32_t pend_side = 0;
	// remembers (side, first candidate index still to look at) on the stack
	auto push = [&](uint32_t side, uint32_t jn) {
		const uint32_t d = depth;
		if (d - lds_lo == 64) {
			// the cache is full: its older half moves out to the HBM array, 32 entries in one store instruction (a
			// stack that never holds more than 64 entries -- every small class -- never touches the pool); every entry
			// finds its own chunk, the 32 may straddle two
			const uint32_t ev = lds_lo;
			if (ev + 32 > (n_chunks ? (W_CHUNK0 << (n_chunks - 1)) : 0u)) { // the array grows into a new chunk (>= 64 entries: one is enough)
				uint32_t base = 0;
				if (lane == 0) {
					const uint32_t sz = n_chunks ? (W_CHUNK0 << (n_chunks - 1)) : W_CHUNK0;
					base = atomicAdd(pool_top, sz);
					if (base + sz > pool_cap || n_chunks >= 31) {
						atomicExch(err, 1u);
						base = NIL;
					}
					chunk_base[n_chunks] = base;
				}
				base = sgpr(base);
				if (base == NIL) {
					pool_ok = false; // (cannot happen: see the pool's size; the host stops the pass on err)
					return;
				}
				n_chunks++;
			}
			__syncthreads(); // (one wave: lane 0 wrote the entries and the chunk table the other lanes now read)
			if (lane < 32)
				wstk[wstk_addr(ev + lane, chunk_base)] = ring[(ev + lane) & 63u];
			lds_lo = ev + 32;
		}
		if (lane == 0)
			ring[d & 63u] = make_uint2(side, jn);
		depth = d + 1;
	};
	// the parent word of a side the walk just reached: into the window (and into memory when the window moves on), or
	// straight into memory; true when it went into the window
	auto set_parent = [&](uint32_t side, uint32_t par) {
		const uint32_t kk = side - win_lo;
		if (HOT && kk < W_WIN) {
			if (lane == 0)
				win_par[kk] = par | W_DIRTY;
			return true;
		}
		if (lane == 0)
			wpar[side] = par;
		return false;
	};
	// candidate idx of a short list, out of the scalar record: a chain of selects (as a function taking the record by reference
	// the compiler put the record into SCRATCH memory and indexed it there: a trip to memory per step)
#define W_INLINE_CAND(rr, idx, out)                                                                                            \
	do {                                                                                                                    \
		uint32_t v__ = (rr).c0;                                                                                         \
		v__ = (idx) == 1 ? (rr).c1 : v__;                                                                               \
		v__ = (idx) == 2 ? (rr).c2 : v__;                                                                               \
		v__ = (idx) == 3 ? (rr).c3 : v__;                                                                               \
		v__ = (idx) == 4 ? (rr).c4 : v__;                                                                               \
		v__ = (idx) == 5 ? (rr).c5 : v__;                                                                               \
		(out) = (idx) < (rr).n ? v__ : NIL;                                                                             \
	} while (0)
	// The loop exists TWICE: with the hot loop in front of the general step, and without it.  The mere presence of the hot
	// loop's code in the loop body made the general step a quarter slower (measured with the hot loop switched off at run
	// time: tangled 116 -> 144 ms, the general step is bound by its dependent instructions and the compiler's placement of
	// them), so a class whose steps the window rarely answers -- links all over the index space -- leaves the first form after
	// a trial of W_TRIAL general steps and continues in the second from the same state.  (The loop body is walk_step.inc,
	// included twice: as a generic lambda instantiated twice the captured state went to scratch memory.)
	constexpr uint32_t W_TRIAL = 512;
	uint32_t n_gen = 0, n_hot = 0; // steps taken by the general step / inside the hot loop
	// ---- with the hot loop, while it pays
	if (HOT) {
		for (;;) {
			if (n_gen >= W_TRIAL && n_hot < n_gen)
				break; // (the only way out of this loop short of the end of the walk)
			n_gen++;
#define W_HOT 1
#include "walk_step.inc"
#undef W_HOT
		}
	}
	// ---- the general step alone
	for (;;) {
#define W_HOT 0
#include "walk_step.inc"
#undef W_HOT
	}
}
// the wave walk leaves the DFS parent of every side it reached in wpar; the scan slot the parent found it through is
// the black edge (slot 0) or the first link of the parent that leads to it
__global__ void k_walk_finish(uint32_t nS, const uint32_t *__restrict__ wpar, const uint32_t *__restrict__ pbr,
			      const uint32_t *__restrict__ loff, const uint32_t *__restrict__ ladj, uint2 *__restrict__ dps)
{
	uint32_t S = BIDX * blockDim.x + threadIdx.x;
	if (S >= nS)
		return;
	const uint32_t p = wpar[S];
	if (p == W_UNVIS || (pbr[S] & PB_BRIDGE))
		return; // not reached by a wave walk / an entry (k_bridges wrote its record)
	uint32_t slot = 0;
	if (S != (p ^ 1u)) {
		const uint32_t lo = loff[p], hi = loff[p + 1];
		for (uint32_t k = lo; k < hi; k++)
			if (ladj[k] == S) {
				slot = k - lo + 1;
				break;
			}
	}
	dps[S] = make_uint2(p, slot);
}

// ------------------------------------------------------------------ 7. pre-order of the union tree
// Children of a side are ordered by the scan slot they were discovered through (at most one child per slot).  No child
// table: the first child of S is the first slot of S whose far side names (S, that slot) as its discovery; the next
// sibling of S is the next such slot of its parent behind S's own.  A parent's list is walked once over all its
// children, so the work stays linear in the degrees.
// The far side o of a segment always is the first child of the side e it is entered through (black edge first), so a
// segment needs three events, not four: 3g = enter e and o, 3g + 1 = leave o, 3g + 2 = leave e -- and when e has no
// other child (the common case) the two leaves are one event (bit 29 of its word) and 3g + 2 stays out of the list.
// "enter" heads the list of its component iff e is the DFS start of a processed component (segments of other
// components get inert words).  merged[g] remembers which form the segment took.
__device__ __forceinline__ uint32_t leave_event(uint32_t p, bool p_is_far) { return 3 * (p >> 1) + (p_is_far ? 1u : 2u); }
__global__ void k_events(uint32_t V, const uint2 *__restrict__ dps, const uint32_t *__restrict__ loff,
			 const uint32_t *__restrict__ ladj, const uint32_t *__restrict__ ckey, const uint32_t *__restrict__ cproc,
			 uint32_t *__restrict__ pk, uint32_t *__restrict__ heads, unsigned b, uint8_t *__restrict__ merged,
			 uint32_t *__restrict__ sdl)
{
	// One lane per SEGMENT: it needs the first child of the far side o, the next sibling of o (= the first gray child of the
	// entered side e, whose first child is o itself) and the next sibling of e -- three searches, the records of both sides
	// in one 16-byte load, the three words written by the lane that worked them out.
	const uint32_t g = BIDX * blockDim.x + threadIdx.x;
	if (g >= V)
		return;
	sdl[g] = 0; // (row E's difference array, filled by k_tree_emit with atomic adds: cleared here on the way, no fill kernel)
	if (g == 0)
		sdl[V] = sdl[V + 1] = 0;
	const uint32_t A = 3 * g;
	const uint4 d4 = *reinterpret_cast<const uint4 *>(dps + 2 * g);
	if (d4.x != 2 * g + 1 && d4.z != 2 * g) { // a segment outside the decomposed components: three inert words
		pk[A] = pk[A + 1] = pk[A + 2] = P0_END;
		merged[g] = 1;
		return;
	}
	const uint32_t o = d4.x == 2 * g + 1 ? 2 * g : 2 * g + 1, e = o ^ 1u; // far side: its DFS parent is the other side
	const uint32_t p = (e & 1u) ? d4.z : d4.x, slot_e = (e & 1u) ? d4.w : d4.y;
	// first side behind slot `from` (>= 1) of the list of `par` that names (par, its slot) as its discovery: four list words
	// in one load and their four records in independent gathers a round (a probe at a time paid two round trips per slot)
	auto child_behind = [&](uint32_t par, uint32_t lo, uint32_t n, uint32_t from) {
		for (uint32_t k = from; k <= n; k += 4) {
			const uint4 x = load4_unaligned(ladj + lo + k - 1);
			const uint32_t rem = n + 1 - k;
			const uint2 r0 = dps[x.x];
			const uint2 r1 = rem > 1 ? dps[x.y] : make_uint2(NIL, 0u);
			const uint2 r2 = rem > 2 ? dps[x.z] : make_uint2(NIL, 0u);
			const uint2 r3 = rem > 3 ? dps[x.w] : make_uint2(NIL, 0u);
			if (r0.x == par && r0.y == k)
				return x.x;
			if (r1.x == par && r1.y == k + 1)
				return x.y;
			if (r2.x == par && r2.y == k + 2)
				return x.z;
			if (r3.x == par && r3.y == k + 3)
				return x.w;
		}
		return NIL;
	};
	const uint2 s01 = *reinterpret_cast<const uint2 *>(loff + 2 * g);
	const uint32_t s2 = loff[2 * g + 2];
	const uint32_t lo_o = (o & 1u) ? s01.y : s01.x, n_o = ((o & 1u) ? s2 : s01.y) - lo_o;
	const uint32_t lo_e = (e & 1u) ? s01.y : s01.x, n_e = ((e & 1u) ? s2 : s01.y) - lo_e;
	const uint32_t c_o = child_behind(o, lo_o, n_o, 1u);  // first child of the far side
	const uint32_t ns_o = child_behind(e, lo_e, n_e, 1u); // next sibling of the far side = first gray child of the entered side
	// what follows "leave the entered side": its next sibling, else the leave of its parent, else nothing (DFS start)
	uint32_t after_e = NIL;
	if (p != NIL) {
		const uint32_t lo_p = loff[p], n_p = loff[p + 1] - lo_p;
		const uint32_t ns_e = child_behind(p, lo_p, n_p, slot_e + 1);
		after_e = ns_e != NIL ? 3 * (ns_e >> 1) : leave_event(p, dps[p].x == (p ^ 1u));
	}
	uint32_t enter = rank_pack(c_o != NIL ? 3 * (c_o >> 1) : A + 1, 0u, b);
	if (p == NIL) {
		const uint32_t comp = ckey[g];
		if (cproc[comp]) {
			enter |= P0_HEAD;
			heads[comp] = A;
		}
	}
	pk[A] = enter;
	if (ns_o != NIL) { // the entered side has more children: leave o -> the first of them; leave e on its own
		pk[A + 1] = rank_pack(3 * (ns_o >> 1), 0u, b);
		pk[A + 2] = rank_pack(after_e, 0u, b);
		merged[g] = 0;
	} else {
		pk[A + 1] = rank_pack(after_e, 1u, b); // both leaves in one
		pk[A + 2] = P0_END;
		merged[g] = 1;
	}
}

// ------------------------------------------------------------------ 8. tree arrays + back edges
__global__ void k_tree_emit(uint32_t nS, const uint2 *__restrict__ cd, const uint8_t *__restrict__ merged,
			    const uint2 *__restrict__ dps, const uint32_t *__restrict__ ckey,
			    const uint32_t *__restrict__ cproc, const uint32_t *__restrict__ voff,
			    const unsigned long long *__restrict__ start_key, const uint32_t *__restrict__ gid_s,
			    uint32_t *__restrict__ t_gid, uint8_t *__restrict__ t_flags, uint32_t *__restrict__ t_par,
			    uint32_t *__restrict__ t_size, uint32_t *__restrict__ t_depth, uint32_t *__restrict__ side_tidx,
			    uint32_t C, uint32_t *__restrict__ c_ntree, uint32_t *__restrict__ ordcnt, uint32_t *__restrict__ hi0,
			    uint32_t *__restrict__ mpre, uint32_t *__restrict__ srccnt, uint32_t *__restrict__ sdl,
			    uint32_t *__restrict__ incnt)
{
	// The class stage works on the tree in T-space (component c owns [2 voff[c] + c, 2 voff[c+1] + c]) and reads, per tree
	// vertex: t_size (0 marks a slot without a vertex), t_par (NIL = root), mpre = mirror pre-order (children visited in
	// DESCENDING idx: the order brackets sit in a bracket list, each child's list is spliced in front of its earlier
	// siblings', flubbles.cpp:586-588; with q(v) = mpre(v) + v + size(v) one gets q(child) = q(parent) + 1, hence
	// mpre(v) = depth(v) + N - v - size(v) in local indices) and srccnt (brackets per mirror pre-order position, written for
	// every vertex later: only the slots without a vertex are cleared here).  All of it is written right here.
	uint32_t S = BIDX * blockDim.x + threadIdx.x;
	// incnt (brackets that end at a vertex: k_back_edges counts into it) is cleared here on the way -- every T-space slot gets
	// its size from exactly one lane of this kernel, and its zero with it: no fill kernel over 8 * 10^8 bytes
	if (S == 0) {
		const uint32_t T = 2 * (nS >> 1) + C;
		hi0[T] = NIL;
		srccnt[T] = srccnt[T + 1] = 0;
		incnt[T] = incnt[T + 1] = 0;
	}
	if (S < C) { // tree size of component S and its dummy root (spanning_tree.cpp:397-402)
		const uint32_t Nr = 2 * (voff[S + 1] - voff[S]), tr = 2 * voff[S] + S;
		if (!cproc[S]) {
			c_ntree[S] = 0;
			t_size[tr + Nr] = 0; // the spare slot (the sides clear their own two below)
			srccnt[tr + Nr] = 0;
			incnt[tr + Nr] = 0;
		} else {
			const uint32_t hr = start_key[S] != ~0ull ? 1u : 0u;
			c_ntree[S] = Nr + hr;
			if (hr) {
				t_gid[tr] = NIL;
				t_flags[tr] = 2;
				t_par[tr] = NIL;
				t_size[tr] = Nr + 1;
				incnt[tr] = 0;
				if (t_depth)
					t_depth[tr] = 0;
				mpre[tr] = tr;
				ordcnt[tr] = 0; // (a dummy root has no back edges of its own; every other tree vertex gets these from k_back_edges)
				hi0[tr] = NIL;
			} else {
				t_size[tr + Nr] = 0;
				srccnt[tr + Nr] = 0;
				incnt[tr + Nr] = 0;
			}
		}
	}
	// One lane per SEGMENT: the entered side e and the far side o (its black child) are neighbours in pre-order, so every
	// array gets their two words in one 8-byte store, and the segment's event words are loaded once for both.
	const uint32_t g = S;
	if (g >= (nS >> 1))
		return;
	const uint32_t c = ckey[g], v0 = voff[c];
	const uint32_t tb = 2 * v0 + c;
	if (!cproc[c]) {
		const uint32_t t = tb + 2 * (g - v0);
		store2_unaligned(t_size + t, 0u, 0u);
		store2_unaligned(srccnt + t, 0u, 0u);
		store2_unaligned(incnt + t, 0u, 0u);
		*reinterpret_cast<uint2 *>(side_tidx + 2 * g) = make_uint2(NIL, NIL);
		return;
	}
	const uint32_t Nh = 2 * (voff[c + 1] - v0), hd = start_key[c] != ~0ull ? 1u : 0u;
	// cd[event] = {sides entered from this event to the end of the list, net depth change from here to the end}
	const uint4 cd01 = load4_unaligned(reinterpret_cast<const uint32_t *>(cd + 3 * g)); // cd[3g], cd[3g + 1]
	const uint2 ent = make_uint2(cd01.x, cd01.y), lv_o = make_uint2(cd01.z, cd01.w);
	const uint4 d4 = *reinterpret_cast<const uint4 *>(dps + 2 * g); // the DFS records of both sides
	const uint32_t o = d4.x == 2 * g + 1 ? 2 * g : 2 * g + 1, e = o ^ 1u; // far side: its DFS parent is the other side
	const uint32_t p = (e & 1u) ? d4.z : d4.x;			       // the entered side's parent
	const uint32_t pre_e = Nh - ent.x, depth_e = 0u - ent.y;
	const uint32_t size_o = ent.x - 1 - lv_o.x, size_e = ent.x - (merged[g] ? lv_o.x : cd[3 * g + 2].x);
	uint32_t par_e;
	if (p == NIL)
		par_e = hd ? 0u : NIL;
	else
		par_e = hd + (Nh - cd[3 * (p >> 1)].x) + (dps[p].x == (p ^ 1u) ? 1u : 0u);
	const uint32_t l = hd + pre_e, t = tb + l; // e sits at t, o at t + 1
	const uint32_t gid = gid_s[g];
	store2_unaligned(t_gid + t, gid, gid);
	t_flags[t] = (uint8_t)(e & 1u);
	t_flags[t + 1] = (uint8_t)((o & 1u) | TF_BLACK);
	store2_unaligned(t_par + t, par_e, l); // (o's parent is e)
	store2_unaligned(t_size + t, size_e, size_o);
	store2_unaligned(incnt + t, 0u, 0u);
	if (t_depth) // (only the hairpin report's T-space setup reads the depths again)
		store2_unaligned(t_depth + t, depth_e + hd, depth_e + 1 + hd);
	store2_unaligned(mpre + t, tb + (depth_e + hd) + (Nh + hd) - l - size_e, tb + (depth_e + 1 + hd) + (Nh + hd) - (l + 1) - size_o);
	*reinterpret_cast<uint2 *>(side_tidx + 2 * g) = (e & 1u) ? make_uint2(t + 1, t) : make_uint2(t, t + 1);
	// Row E's difference array (k_shift_delta, par_kernels.hip), while both sizes are in registers: under a branching entered
	// side the black child's subtree goes behind the gray ones in the candidate stack (tree_utils.cpp:47-76).  In units of
	// black vertices, over the segment slots of pre-order: this segment is slot v0 + pre_e / 2 (cleared by the caller).
	if (sdl) {
		const uint32_t gray = (size_e - 1 - size_o) / 2, black = (size_o + 1) / 2, gs = v0 + pre_e / 2;
		if (gray) {
			atomicAdd(&sdl[gs], gray);			   // the black subtree moves behind the gray ones
			atomicAdd(&sdl[gs + black], 0u - gray - black); // the gray subtrees move forward by the black one's entries
			atomicAdd(&sdl[gs + size_e / 2], black);
		}
	}
}
// back edges of from_bd out of side S, in scan order (process_edge, spanning_tree.cpp:360-398):
//  - a side without links points back at the root unless the root is its tree parent (:433-438)
//  - self-loop links: one back edge per segment, from the black child, at its first loop slot
//  - a link to a descendant was already turned into a back edge by the descendant
//  - a link to the tree parent, or a repeated link to the same side, is already "connected"
// One kernel: a workgroup takes BE_SIDES consecutive sides (BE_ITER per lane, lanes on consecutive sides), every lane
// counts the back edges of its sides, the workgroup adds the counts up (wave shuffles + LDS) and takes its stretch of the
// dense list with ONE atomic add (few enough workgroups that the one hot word does not hurt), then every lane walks its
// sides again (now from cache) and writes them.  The dense list is in no particular order -- nothing downstream needs one:
// a bracket's place in its list follows from its source and b_ord, its rank among the ordinary edges of that source
// (bottom first; k_bracket_place turns it round).
#ifndef POVU_BE_ITER
#define POVU_BE_ITER 8
#endif
static constexpr uint32_t BE_ITER = POVU_BE_ITER, BE_SIDES = TPB * BE_ITER;
template <bool EMIT>
__device__ __forceinline__ uint32_t side_back_edges(uint32_t S, uint32_t p, uint32_t at,
						    const uint32_t *__restrict__ loff, const uint32_t *__restrict__ ladj,
						    const uint2 *__restrict__ dps, const uint32_t *__restrict__ side_tidx,
						    const uint32_t *__restrict__ ckey, const uint32_t *__restrict__ voff,
						    const uint32_t *__restrict__ t_par, uint32_t *__restrict__ b_src,
						    uint32_t *__restrict__ b_tgt, uint32_t *__restrict__ b_ord,
						    const uint8_t *__restrict__ dupflag, uint32_t *__restrict__ incnt, uint32_t &highest,
						    uint32_t *first2)
{
	uint32_t n = 0;
	const uint32_t c = ckey[S >> 1], root = 2 * voff[c] + c;
	const uint32_t lo = loff[S], hi = loff[S + 1];
	auto out = [&](uint32_t tgt) {
		if (EMIT) {
			b_src[at + n] = p;
			b_tgt[at + n] = tgt;
			b_ord[at + n] = n; // later pushed = nearer the top of the bracket list
			atomicAdd(&incnt[tgt], 1u); // brackets that end at tgt
		} else {
			highest = min(highest, tgt);
			if (n < 2)
				first2[n] = tgt; // (LDS: a side with at most two back edges is written from there, without a second walk)
		}
		n++;
	};
	if (lo == hi) {
		if (p == root || t_par[p] != 0)
			out(root);
		return n;
	}
	const uint32_t dp = dps[S].x;
	bool loop_seen = false;
	// four slots a round: their far sides in one 16-byte load, the tree vertices of those in four independent gathers
	for (uint32_t k0 = lo; k0 < hi; k0 += 4) {
		const uint4 o4 = load4_unaligned(ladj + k0);
		const uint32_t rem = hi - k0;
		const uint32_t os[4] = {o4.x, o4.y, o4.z, o4.w};
		const uint32_t xs[4] = {side_tidx[o4.x], rem > 1 ? side_tidx[o4.y] : NIL, rem > 2 ? side_tidx[o4.z] : NIL,
					rem > 3 ? side_tidx[o4.w] : NIL};
#pragma unroll
		for (uint32_t q = 0; q < 4; q++) {
			if (q >= rem)
				break;
			const uint32_t k = k0 + q, o = os[q], x = xs[q];
			if (o == (S ^ 1)) {
				if (dp == o && !loop_seen)
					out(x);
				loop_seen = true;
				continue;
			}
			if (x > p || o == dp)
				continue;
			bool dup = false;
			if (dupflag) {
				dup = dupflag[k] != 0;
			} else {
				for (uint32_t j = lo; j < k; j++)
					if (ladj[j] == o) {
						dup = true;
						break;
					}
			}
			if (!dup)
				out(x);
		}
	}
	return n;
}
__global__ void __launch_bounds__(TPB) k_back_edges(uint32_t nS, const uint32_t *__restrict__ loff, const uint32_t *__restrict__ ladj,
						     const uint2 *__restrict__ dps, const uint32_t *__restrict__ side_tidx,
						     const uint32_t *__restrict__ ckey, const uint32_t *__restrict__ voff,
						     const uint32_t *__restrict__ t_par, uint32_t *__restrict__ total_out,
						     uint32_t *__restrict__ b_src, uint32_t *__restrict__ b_tgt,
						     uint32_t *__restrict__ b_ord, const uint8_t *__restrict__ dupflag,
						     uint32_t *__restrict__ ordcnt, uint32_t *__restrict__ hi0, uint32_t *__restrict__ incnt,
						     uint32_t cap)
{
	const uint32_t S0 = BIDX * BE_SIDES + threadIdx.x;
	__shared__ uint32_t first2[BE_ITER][TPB][2]; // the first two targets of every side of the chunk
	__shared__ uint8_t cnt8[BE_ITER][TPB];	     // min(its back edges, 255)
	uint32_t n = 0;
	for (uint32_t it = 0; it < BE_ITER; it++) {
		const uint32_t S = S0 + it * TPB;
		const uint32_t p = S < nS ? side_tidx[S] : NIL;
		uint32_t k = 0;
		if (p != NIL) {
			uint32_t highest = NIL;
			k = side_back_edges<false>(S, p, 0, loff, ladj, dps, side_tidx, ckey, voff, t_par, b_src, b_tgt, b_ord, dupflag, incnt, highest,
						   first2[it][threadIdx.x]);
			// what the class stage needs per tree vertex, known right here: its ordinary brackets (p's stretch of the
			// bracket list is sized with it) and the highest vertex they reach (hi_0, flubbles.cpp:515-519)
			ordcnt[p] = k;
			hi0[p] = highest;
			n += k;
		}
		cnt8[it][threadIdx.x] = (uint8_t)min(k, 255u);
	}
	// exclusive prefix of n over the workgroup
	__shared__ uint32_t wsum[TPB / 64], base;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t inc = n;
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t y = __shfl_up(inc, off);
		if ((int)lane >= off)
			inc += y;
	}
	if (lane == 63)
		wsum[wave] = inc;
	__syncthreads();
	uint32_t before = 0, all = 0;
	for (uint32_t w = 0; w < TPB / 64; w++) {
		if (w < wave)
			before += wsum[w];
		all += wsum[w];
	}
	if (threadIdx.x == 0)
		base = all ? atomicAdd(total_out, all) : 0u;
	__syncthreads();
	if (!n || base + all > cap) // (more back edges than the list was carved for cannot happen -- the host checks the total --, and must not write)
		return;
	uint32_t at = base + before + inc - n;
	for (uint32_t it = 0; it < BE_ITER; it++) {
		const uint32_t k = cnt8[it][threadIdx.x];
		if (!k)
			continue;
		const uint32_t S = S0 + it * TPB, p = side_tidx[S];
		if (k <= 2) {
			for (uint32_t j = 0; j < k; j++) {
				const uint32_t tgt = first2[it][threadIdx.x][j];
				b_src[at + j] = p;
				b_tgt[at + j] = tgt;
				b_ord[at + j] = j;
				atomicAdd(&incnt[tgt], 1u);
			}
			at += k;
		} else {
			uint32_t unused = NIL;
			at += side_back_edges<true>(S, p, at, loff, ladj, dps, side_tidx, ckey, voff, t_par, b_src, b_tgt, b_ord, dupflag, incnt, unused,
						    nullptr);
		}
	}
}

// Conformance export only (povu_hip_debug_edge_ids): the tree and back edges of from_bd share one id counter in
// creation order (spanning_tree.cpp:784-805), so the id of the tree edge into vertex t is (t - 1) plus the back
// edges created before t was discovered.  A side creates its back edges while it scans its links between two of
// its tree children: w[child] = those just before that child, tail[side] = those after its last child.
__global__ void k_edge_id_weights(uint32_t nS, const uint32_t *__restrict__ loff, const uint32_t *__restrict__ ladj,
				  const uint2 *__restrict__ dps,
				  const uint32_t *__restrict__ side_tidx, const uint32_t *__restrict__ ckey,
				  const uint32_t *__restrict__ voff, const uint32_t *__restrict__ t_par,
				  const uint8_t *__restrict__ dupflag, uint32_t *__restrict__ w, uint32_t *__restrict__ tail)
{
	uint32_t S = BIDX * blockDim.x + threadIdx.x;
	if (S >= nS)
		return;
	const uint32_t p = side_tidx[S];
	if (p == NIL)
		return;
	const uint32_t c = ckey[S >> 1], root = 2 * voff[c] + c;
	const uint32_t lo = loff[S], hi = loff[S + 1];
	uint32_t cnt = 0;
	if (lo == hi) {
		cnt = (p == root || t_par[p] != 0) ? 1u : 0u;
	} else {
		const uint32_t dp = dps[S].x;
		bool loop_seen = false;
		for (uint32_t k = lo; k < hi; k++) {
			const uint32_t o = ladj[k];
			if (o == (S ^ 1)) {
				if (dp == o && !loop_seen)
					cnt++;
				loop_seen = true;
				continue;
			}
			const uint2 ro = dps[o];
			if (ro.x == S && ro.y == k - lo + 1) { // the link that discovered o
				w[side_tidx[o]] = cnt;
				cnt = 0;
				continue;
			}
			if (side_tidx[o] > p || o == dp)
				continue;
			bool dup = false;
			if (dupflag) {
				dup = dupflag[k] != 0;
			} else {
				for (uint32_t j = lo; j < k; j++)
					if (ladj[j] == o) {
						dup = true;
						break;
					}
			}
			if (!dup)
				cnt++;
		}
	}
	tail[p] = cnt;
}
void debug_edge_id_weights(const CompState &cs, const SeqWs &sw, const TreeWs &tw, uint32_t *w, uint32_t *tail, hipStream_t s)
{
	const uint32_t nS = 2 * sw.V;
	LAUNCH(k_edge_id_weights, nS, s, nS, cs.loff, cs.ladj, tw.dps, tw.side_tidx, cs.ckey, cs.voff, sw.t_par,
	       tw.last_dupflag, w, tail);
}

// Repeated links of one side (same far side) beyond the first: only the first can become a back edge.
// Sides with few links find them by looking back over their own list; when some side has many links
// the look-back is quadratic, so the flags come from a stable two-key radix sort of all slots instead.
__global__ void k_slot_keys(uint32_t n, const uint32_t *__restrict__ ladj, uint32_t *__restrict__ key, uint32_t *__restrict__ val)
{
	uint32_t k = BIDX * blockDim.x + threadIdx.x;
	if (k < n) {
		key[k] = ladj[k];
		val[k] = k;
	}
}
__global__ void k_slot_side(uint32_t nS, const uint32_t *__restrict__ loff, uint32_t *__restrict__ side_of)
{
	uint32_t S = BIDX * blockDim.x + threadIdx.x;
	if (S >= nS)
		return;
	for (uint32_t k = loff[S]; k < loff[S + 1]; k++)
		side_of[k] = S;
}
__global__ void k_gather_keys(uint32_t n, const uint32_t *__restrict__ val, const uint32_t *__restrict__ src, uint32_t *__restrict__ key)
{
	uint32_t q = BIDX * blockDim.x + threadIdx.x;
	if (q < n)
		key[q] = src[val[q]];
}
__global__ void k_dup_flags(uint32_t n, const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval,
			    const uint32_t *__restrict__ ladj, uint8_t *__restrict__ dupflag)
{
	uint32_t q = BIDX * blockDim.x + threadIdx.x;
	if (q >= n)
		return;
	// sorted by (side, far side), slots ascending inside equal pairs (both sorts are stable)
	const uint32_t k = sval[q];
	dupflag[k] = (q > 0 && skey[q - 1] == skey[q] && ladj[sval[q - 1]] == ladj[k]) ? 1 : 0;
}

// ------------------------------------------------------------------ workspace
template <typename F>
static void tree_spans(TreeWs &tw, size_t V, size_t E, size_t Cmax, int groups, const StageWsOpts &o, F &&take_any)
{
	// Group 1: what outlives the tree stage (the debug hook that recomputes the edge-id weights after a pass reads the DFS
	// records, the tree index of every side and the duplicate-slot flags).  Group 2: the rest -- dead once the class stage
	// starts, whose own arrays lie over it (stage_workspace_carve).
	auto take1 = [&](void **p, size_t bytes) {
		if (groups & 1)
			take_any(p, bytes);
	};
	auto take = [&](void **p, size_t bytes) {
		if (groups & 2)
			take_any(p, bytes);
	};
	const size_t nS = 2 * V + 2, NA = std::max<size_t>(4 * V, 2 * E) + 8; // tour positions; the 8-byte arrays double as slot buffers
	const size_t NSL = std::max<size_t>(2 * V + 2 * E, 4 * V) + 16; // scan slots of all sides / events of the second ranking
	take1((void **)&tw.dvis_slots, std::max(2 * E, 2 * V) + 16); // per-slot duplicate flags (hub graphs); earlier: per-side class flags
	take1((void **)&tw.side_tidx, nS * 4);
	take1((void **)&tw.dps, nS * 8);
	take((void **)&tw.dist, NSL * 4);
	take((void **)&tw.xrec, (NA / 64 + 4) * 16);
	take((void **)&tw.xrank, (NA / 64 + 4) * 4);
	take((void **)&tw.evt, NA * 8);
	take((void **)&tw.t0seg, (V + 2) * 16);
	for (uint32_t **p : {&tw.entry_ps, &tw.entry_list, &tw.be_cnt})
		take((void **)p, nS * 4);
	take((void **)&tw.dvis, nS);
	take((void **)&tw.wadj, (nS + 2 * E + 8) * 4); // wave walk: class-filtered scan lists (4 bytes a slot); earlier in the pass: twin slots [2E]
	take((void **)&tw.cproc, (Cmax + 2) * 4);
	take((void **)&tw.rk_pk, NSL * 4);
	take((void **)&tw.rk_heads, (Cmax + 2) * 4);
	// One block, three tenants that are never at home together: the six level pools of the list rankings (the tour's at the
	// start, the pre-order's after the class walks); behind the pools the bridge test's values -- one 16-byte value per
	// segment that carries one (K <= V) and their running xor, live between the first ranking and the class walks, and on
	// hub graphs once more after the second ranking as the key and value buffers of the slot sort (2E four-byte words each);
	// and, over all of it, the wave walk's per-side records, stack pool and parents (only while the large classes are
	// walked).  ~12 GB instead of 21 on the whole-genome workload.
	auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
	const size_t pool_b = pad(rank_pool_words(NSL, Cmax + 1) * 4);
	const size_t xv_b = pad((std::max<size_t>(V + 2, E + 8) + 8) * 16);
	const size_t wrec_b = pad(nS * 32);	     // 32-byte record per side
	const size_t wstk_b = pad(3 * nS * 8 + 64); // stack pool: chunks double in size, a class takes less than three entries per side
	const size_t wpar_b = pad(nS * 4);
	char *blk = nullptr;
	// (the walk's third of it is 12 GB on the whole-genome workload against 9 for the other two, and most graphs never need
	// it: with !walk_inline it comes out of an arena of its own when a pass does have large classes)
	take((void **)&blk, o.walk_inline ? std::max(6 * pool_b + 2 * xv_b, wrec_b + wstk_b + wpar_b) : 6 * pool_b + 2 * xv_b);
	if (groups & 2)
		tw.walk_inline = o.walk_inline;
	if (blk) {
		uint32_t **pools[6] = {&tw.rk_nx, &tw.rk_wa, &tw.rk_wb, &tw.rk_tA, &tw.rk_tB, &tw.rk_tC};
		for (int k = 0; k < 6; k++)
			*pools[k] = reinterpret_cast<uint32_t *>(blk + (size_t)k * pool_b);
		tw.xval = reinterpret_cast<decltype(tw.xval)>(blk + 6 * pool_b);
		tw.xps = reinterpret_cast<decltype(tw.xps)>(blk + 6 * pool_b + xv_b);
		if (o.walk_inline) {
			tw.wrec = reinterpret_cast<decltype(tw.wrec)>(blk);
			tw.wstk = reinterpret_cast<decltype(tw.wstk)>(blk + wrec_b);
			tw.wpar = reinterpret_cast<decltype(tw.wpar)>(blk + wrec_b + wstk_b);
		} else {
			tw.wrec = nullptr;
			tw.wstk = nullptr;
			tw.wpar = nullptr;
		}
	}
}

size_t tree_workspace_bytes(size_t V, size_t E, size_t Cmax, int groups, const StageWsOpts &o)
{
	TreeWs tmp{};
	size_t total = 0;
	tree_spans(tmp, V, E, Cmax, groups, o, [&](void **, size_t bytes) { total += ((bytes + 255) & ~size_t(255)) + 256; });
	return total + (1 << 20);
}

void tree_carve(Arena &ar, TreeWs &tw, size_t V, size_t E, size_t Cmax, int groups, const StageWsOpts &o)
{
	tree_spans(tw, V, E, Cmax, groups, o, [&](void **dst, size_t bytes) { *dst = ar.take<char>(bytes); });
}

size_t walk_workspace_bytes(size_t V)
{
	const size_t nS = 2 * V + 2;
	auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
	return pad(nS * 32) + pad(3 * nS * 8 + 64) + pad(nS * 4) + 4096;
}

// The workspace of the parallel stages: what the tree stage hands to the class stage and what outlives the tree stage side
// by side; then ONE stretch that holds the tree stage's own arrays first and the class stage's afterwards.
size_t stage_workspace_bytes(size_t V, size_t E, size_t Cmax, const StageWsOpts &o)
{
	return par_workspace_bytes(V, E, Cmax, 1, o) + tree_workspace_bytes(V, E, Cmax, 1, o) +
	       std::max(tree_workspace_bytes(V, E, Cmax, 2, o), par_workspace_bytes(V, E, Cmax, 2, o));
}
void stage_workspace_carve(Arena &ar, ParWs &pw, TreeWs &tw, size_t V, size_t E, size_t Cmax, const StageWsOpts &o)
{
	par_carve(ar, pw, V, E, Cmax, 1, o);
	tree_carve(ar, tw, V, E, Cmax, 1, o);
	const size_t shared = ar.used();
	tree_carve(ar, tw, V, E, Cmax, 2, o);
	const size_t tree_end = ar.used();
	ar.rewind(shared);
	par_carve(ar, pw, V, E, Cmax, 2, o);
	ar.advance_to(tree_end);
}

// ------------------------------------------------------------------ driver
void tree_tour_words(const CompState &cs, uint32_t V, uint32_t E, TreeWs &tw, bool force_sparse_splitters, hipStream_t s)
{
	const uint32_t nS = 2 * V;
	const size_t n_slots = 2 * (size_t)E;
	if (n_slots >= P0_END || 3 * (size_t)V >= P0_END) // (the first ranking runs over the 2E slots, the second over 3 V events)
		throw HipError("graph too large for the packed list ranking: 2 * links and 3 * segments must stay below 2^30");
	const unsigned bitsA = force_sparse_splitters ? 4u : rank_bucket_bits(n_slots);
	ulonglong2 *hside = reinterpret_cast<ulonglong2 *>(tw.evt); // [nS] 16-byte words (evt, [max(4V, 2E)] 8-byte words, is free until the second ranking)
	uint32_t *ft = tw.be_cnt;				    // [nS] (free until the back edges are counted)
	LAUNCH(k_tour_words, nS, s, nS, cs.loff, cs.ladj, cs.lle, tw.rk_pk, bitsA, hside, ft);
	tw.tour_words_done = true;
}

int64_t run_parallel_tree(const CompState &cs, SeqWs &sw, ParWs &pw, TreeWs &tw, uint32_t C, uint32_t event_lists,
			   uint32_t max_side_links, bool force_big_class_dfs, bool force_sparse_splitters, StageTimer &tm,
			   hipStream_t s)
{
	const uint32_t V = sw.V, E = sw.E, nS = 2 * V;
	const unsigned long long *start_key = (const unsigned long long *)cs.start_key;

	// ---- 1-2. spanning forest, rooted at the DFS start by an Euler tour
	tm.begin("tree_root_forest");
	const uint32_t NA = 2 * (V - C); // arcs of the spanning forest of the segments = positions of the tours
	const size_t n_slots = 2 * (size_t)E; // list elements = adjacency slots (loff[nS] = 2E: every link has one at either end)
	RankBufs rb{tw.rk_pk, tw.rk_heads, tw.rk_nx, tw.rk_wa, tw.rk_wb, tw.rk_tA, tw.rk_tB, tw.rk_tC};
	const unsigned bitsA = force_sparse_splitters ? 4u : rank_bucket_bits(n_slots);
	ulonglong2 *hside = reinterpret_cast<ulonglong2 *>(tw.evt); // [nS] 16-byte words (evt, [max(4V, 2E)] 8-byte words, is free until the second ranking)
	uint32_t *ft = tw.be_cnt;						    // [nS] (free until the back edges are counted)
	if (!tw.tour_words_done) // (else: started by povu_hip_decompose while the host still waited for the component sizes)
		tree_tour_words(cs, V, E, tw, force_sparse_splitters, s);
	tw.tour_words_done = false;
	LAUNCH(k_tour_ends, C, s, C, cs.voff, start_key, cs.loff, cs.ladj, cs.lle, rb.pk, rb.heads);
	if (n_slots)
		list_rank_splitters<false>((uint32_t)n_slots, bitsA, tw.dist, nullptr, C, rb, s);
	const uint32_t *dist = tw.dist;
	const uint32_t XW = NA / 64 + 1; // words of the position bitmap
	HIP_CHECK(hipMemsetAsync(tw.xrec, 0, ((size_t)XW + 2) * 16, s));
	LAUNCH(k_t0_parents, std::max(V, C), s, V, dist, cs.ckey, cs.voff, cs.loff, cs.ladj, cs.lle, ft, rb.heads,
	       tw.t0seg, tw.xrec, C, start_key, pw.err + 2, tw.dvis_slots);
	tm.end(40);

	// ---- 3-4. bridges and 2-edge-connected classes
	tm.begin("tree_bridges_classes");
	LAUNCH(k_bit_counts, (size_t)XW + 1, s, XW, tw.xrec, tw.xrank);
	scan_exclusive_u32(tw.xrank, tw.xrank, (size_t)XW + 1, pw.scan_tmp, pw.scan_tmp_bytes, s); // xrank[XW] = values in all
	LAUNCH(k_bit_ranks, XW, s, XW, tw.xrank, tw.xrec);
	LAUNCH(k_tour_values, V, s, V, tw.t0seg, ft, hside, tw.xrec, tw.xval);
	// (the number of values stays on the device: the scan is launched for the most there can be, one per segment, and
	// stops at their count + 1)
	scan_exclusive_xor_u128(tw.xval, tw.xps, (size_t)V + 1, pw.scan_tmp, pw.scan_tmp_bytes, s, tw.xrank + XW);
	uint8_t *multi = tw.dvis_slots; // (see tree_spans: sized for max(2E, 2V) + 16; cleared by k_t0_parents)
	uint32_t *cstate = sw.cur; // [nS+1]
	LAUNCH(k_bridges, V, s, V, tw.t0seg, tw.xps, tw.xrec, hside, ft, dist, cs.loff, cs.lle, cs.ckey, tw.cproc, cs.voff, multi, cstate, tw.dps);
	tm.end(8 + 44);

	// ---- 5-6. entries and the per-class DFS
	tm.begin("tree_class_dfs");
	uint32_t *n_entry_dev = pw.err + 9; // (cleared with the other counters at the start of the pass)
	KLAUNCH(k_entry_list, dim3((nS + EL_SIDES - 1) / EL_SIDES), dim3(TPB), 0, s, nS, cstate, multi, cs.ckey, tw.cproc, tw.entry_list, n_entry_dev);
	// Small classes are walked by the plain walk, one lane each.  A lane that finds its class larger than CLASS_BUDGET sides
	// gives up and reports the entry; those classes are then walked from the start by the walk with the short dependent
	// chain, at the price of a filtering pass over the adjacency (it marks visits in its own bytes and rewrites the same
	// parents, so the abandoned attempt leaves nothing behind).
	const uint32_t *big_list = tw.entry_list;
	// (7.) one list per processed component, one two-event list per side of an unprocessed one
	const uint32_t n_events = 3 * V; // three per segment
	const unsigned bitsE = force_sparse_splitters ? 4u : rank_bucket_bits(n_events);
	uint8_t *merged = tw.dvis; // [V]
	bool events_done = false;
	auto enqueue_events = [&]() {
		HIP_CHECK(hipMemsetAsync(rb.heads, 0xFF, (size_t)C * 4, s)); // components that are not decomposed head no list
		LAUNCH(k_events, V, s, V, tw.dps, cs.loff, cs.ladj, cs.ckey, tw.cproc, rb.pk, rb.heads, bitsE, merged, pw.sdl);
	};
	uint32_t n_big = force_big_class_dfs ? tw.host->read_u32(n_entry_dev, s) : 0; // (A/B mode: every class through the wave walk)
	if (!force_big_class_dfs) {
		// Lanes in flight = a window of sides whose scattered stores meet again in L2.  Round 4's walk (one tree edge per
		// three dependent round trips) was best at ~3000 x 64 lanes; with the black follow-through the walk issues fewer,
		// wider accesses and more lanes in flight pay: 49 152 workgroups measured 0.2 ms faster on 2.3 * 10^7 classes
		// (gpurun_out r5m / r5n: 3072 -> 2.96 ms, 24 576 -> 2.83, 49 152 -> 2.76, 196 608 -> 2.73).  The number of classes
		// stays on the device: the launch is a grid-stride loop, one synchronisation (how many classes overflowed) serves both.
		const unsigned all_blocks = (nS + 63) / 64;
		unsigned dfs_blocks = std::min(all_blocks, 49152u);
		if (const char *ev = getenv("POVU_HIP_DFS_BLOCKS")) // (tuning hook)
			dfs_blocks = std::min(all_blocks, std::max(1u, (unsigned)atoi(ev)));
		uint32_t *n_over = pw.err + 4, *over_list = tw.entry_ps; // (a spare array of the entries' size)
		uint32_t budget = CLASS_BUDGET;
		if (const char *ev = getenv("POVU_HIP_CLASS_BUDGET")) // (tuning hook: sides a lane walks before it hands its class to the wave walk)
			budget = (uint32_t)std::max(16, atoi(ev));
		KLAUNCH(k_class_dfs_small, dim3(dfs_blocks), dim3(64), 0, s, n_entry_dev, tw.entry_list, cs.loff, cs.ladj, cstate, tw.dps,
			budget, n_over, over_list);
		// How many classes overflowed is read without leaving the stream idle (HostScratch::mark): behind the word goes the
		// next stage's first kernel, on the assumption that none did -- with large classes around it ran on half-filled
		// records and simply runs again behind the walks (it writes every word of its outputs).  Not with stage timers (the
		// kernel would be booked on this stage).
		if (!tm.enabled) {
			uint32_t *w = tw.host->take<uint32_t>(1);
			publish_words(w, WordSrc{{n_over}}, 1, s);
			tw.host->mark(s);
			enqueue_events();
			tw.host->wait();
			n_big = *w;
			events_done = n_big == 0;
		} else {
			n_big = tw.host->read_u32(n_over, s);
		}
		big_list = over_list;
	}
	if (n_big) { // large classes: one wave each (see section 6)
		if (!tw.walk_inline) { // the records of all sides, the stack pool and the parents: taken when a pass needs them
			if (!tw.walk_arena)
				throw HipError("class walk: no arena for the wave walk's arrays (internal)");
			const size_t nSw = 2 * (size_t)V + 2;
			tw.walk_arena->reserve(walk_workspace_bytes(V));
			tw.wrec = reinterpret_cast<decltype(tw.wrec)>(tw.walk_arena->take<char>(nSw * 32));
			tw.wstk = reinterpret_cast<decltype(tw.wstk)>(tw.walk_arena->take<char>(3 * nSw * 8 + 64));
			tw.wpar = reinterpret_cast<decltype(tw.wpar)>(tw.walk_arena->take<char>(nSw * 4));
		}
		uint32_t *pool_top = pw.err + 7, *walk_err = pw.err + 8; // (cleared with the other counters at the start of the pass)
		LAUNCH(k_class_recs, nS, s, nS, cs.loff, cs.ladj, cstate, cs.ckey, tw.cproc, tw.wadj, tw.wrec, tw.wpar);
		const uint32_t route = getenv("POVU_HIP_WALK_ROUTE") ? (uint32_t)atoi(getenv("POVU_HIP_WALK_ROUTE")) : 0u; // (A/B hook)
		const uint32_t pool_cap = (uint32_t)std::min<size_t>(3 * (size_t)nS, 0xFFFFFFF0u);
		// the two forms side by side (see k_class_walk_wave): the second on the context's walk stream when there is one
		hipStream_t s2 = tw.walk_stream ? tw.walk_stream : s;
		if (s2 != s) {
			HIP_CHECK(hipEventRecord(tw.walk_fork, s));
			HIP_CHECK(hipStreamWaitEvent(s2, tw.walk_fork, 0));
		}
		KLAUNCH(k_class_walk_wave<false>, dim3(n_big), dim3(64), 0, s2, n_big, big_list, tw.wrec, tw.wadj, tw.wpar, tw.wstk, pool_top, pool_cap,
			walk_err, nS, route);
		KLAUNCH(k_class_walk_wave<true>, dim3(n_big), dim3(64), 0, s, n_big, big_list, tw.wrec, tw.wadj, tw.wpar, tw.wstk, pool_top, pool_cap,
			walk_err, nS, route);
		if (s2 != s) {
			HIP_CHECK(hipEventRecord(tw.walk_join, s2));
			HIP_CHECK(hipStreamWaitEvent(s, tw.walk_join, 0));
		}
		LAUNCH(k_walk_finish, nS, s, nS, tw.wpar, cstate, cs.loff, cs.ladj, tw.dps);
		if (tw.host->read_u32(walk_err, s)) // (an unfinished walk leaves a broken tree: nothing downstream may run on it)
			throw HipError("class walk: stack pool exhausted (internal sizing bug)");
	}
	tm.end(5);

	// ---- 7. pre-order, sizes, depths
	tm.begin("tree_preorder");
	(void)max_side_links;
	if (!events_done)
		enqueue_events();
	list_rank_splitters<true, true>(n_events, bitsE, nullptr, tw.evt, C, rb, s);
	(void)event_lists;
	tm.end(40);

	// ---- 8. tree arrays in pre-order and the from_bd back edges
	tm.begin("tree_emit");
	LAUNCH(k_tree_emit, std::max(V, C), s, nS, tw.evt, merged, tw.dps, cs.ckey, tw.cproc, cs.voff, start_key, cs.gid_s, sw.t_gid, sw.t_flags,
	       sw.t_par, sw.t_size, (sw.hairpins || sw.want_depth) ? sw.t_depth : nullptr, tw.side_tidx, C, sw.c_ntree, pw.lsz, pw.hi0, pw.mpre, pw.dlt,
	       pw.sdl, pw.incnt); // (incnt: k_back_edges counts the brackets that end at a vertex into it)
	pw.sdl_filled = true;
	const uint8_t *dupflag = nullptr;
	if (max_side_links > 64 && E) { // see k_dup_flags
		const uint32_t n_slots = 2 * E;
		// the bridge test's value arrays and the event ranks are free by now ([NA] 8-byte words each, NA >= 2E)
		uint32_t *k1 = (uint32_t *)tw.xval, *k2 = k1 + n_slots, *v1 = (uint32_t *)tw.xps, *v2 = v1 + n_slots;
		uint32_t *side_of = (uint32_t *)tw.evt;
		LAUNCH(k_slot_keys, n_slots, s, n_slots, cs.ladj, k1, v1);
		sort_pairs_u32(k1, k2, v1, v2, n_slots, bits_for(nS), pw.sort_tmp, pw.sort_tmp_bytes, s);
		LAUNCH(k_slot_side, nS, s, nS, cs.loff, side_of);
		LAUNCH(k_gather_keys, n_slots, s, n_slots, v2, side_of, k1);
		sort_pairs_u32(k1, k2, v2, v1, n_slots, bits_for(nS), pw.sort_tmp, pw.sort_tmp_bytes, s);
		LAUNCH(k_dup_flags, n_slots, s, n_slots, k2, v1, cs.ladj, tw.dvis_slots);
		dupflag = tw.dvis_slots;
	}
	tw.last_dupflag = dupflag;
	uint32_t *nb0_dev = pw.err + 6; // (cleared with the other counters at the start of the pass)
	KLAUNCH(k_back_edges, dim3((nS + BE_SIDES - 1) / BE_SIDES), dim3(TPB), 0, s, nS, cs.loff, cs.ladj, tw.dps, tw.side_tidx, cs.ckey,
		cs.voff, sw.t_par, nb0_dev, pw.b_src, pw.b_tgt, pw.b_ord, dupflag, pw.lsz, pw.hi0, pw.incnt,
		(uint32_t)std::min<size_t>(pw.nb_cap, 0xFFFFFFFFu));
	// (their number stays on the device: the class stage reads it together with its own counts)
	tm.end(6);
	return NB0_ON_DEVICE;
}

} // namespace povu_hip
