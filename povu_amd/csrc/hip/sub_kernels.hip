// sub_kernels.hip -- the three INSERTING passes of `povu decompose -s` on the state of an all-parallel pass:
// find_concealed (src/povu/algorithms/concealed.cpp:1198-1243), find_midi (midi.cpp:225-268) and find_smothered
// (smothered.cpp:385-432), after find_tiny / find_parallel (leaf_kernels.hip) have relabelled the leaves.
//
// The reference walks every flubble of a PVST and asks its spanning tree and gen_tree_meta (src/povu/graph/
// tree_utils.cpp:674-688: depths, LoA, LCA, the per-vertex bracket table) a few dozen questions; what it finds it then
// splices into the PVST one vertex after the other.  Here:
//   tables    depth (the tree stage writes it), literal hi (flubbles.cpp:552,640: a simplifying edge resets it to the root,
//             which every ancestor then inherits), back edges by source and by target (all three types); the BRACKETS of a
//             vertex (count_brackets / collect_backedges_by_vertex, tree_utils.cpp:167-216, 531-574) are not materialised
//             as the reference's table is (one entry per back edge and vertex it spans: 10^10 on a graph with large
//             tangles) but enumerated where a row is asked for, through a segment tree over the targets of the edges in
//             source order; where the reference's row ORDER shows (back-edge idx = creation order in from_bd) the
//             "creation key" below decides; LoA (tree_utils.cpp:224-273) in closed form from a max-tree over the edges'
//             intervals -- by the reference's own heap, one lane per component, only where a self-loop back edge (the
//             root of a tip-less component has one) leaves stale entries in it (oracle/povu_oracle_sub.inc) --, LCA =
//             parent of the shallowest vertex of an index range (a segment tree over the depths: vertex idx = pre-order
//             rank);
//   search    one lane per flubble (concealed.cpp:234-921) and one per concealed vertex (smothered.cpp:61-318), each run
//             twice -- count, scan, emit;
//   splice    add_concealed, find_midi / add_midi and add_smothered change the children vectors of the PVST in an order
//             that shows in the output (vector::push_back / erase, pvst.hpp:860-885) and reach across families (a
//             concealed vertex is appended to the children of a CHILD flubble, concealed.cpp:1077): they run literally,
//             one WAVE per component, over vectors in a bump arena; the loops over a flubble's children are shared out
//             over the lanes.
// Accidents of the reference that are kept and behaviour it leaves undefined are those listed at the top of
// oracle/povu_oracle_sub.inc, decided the same way here.  PARITY UNPINNED: the reference holds no C / M / S line; the
// tests compare with the oracle's sequential restatement.
//
// Creation key of an ordinary back edge (the reference's rows are in back-edge idx order = the order from_bd created
// them): an edge of source u is created while u's adjacency is scanned -- between the discoveries of two of its children,
// or behind the last.  tn = the next vertex discovered after it (the child, or the vertex behind u's subtree); edges with
// the same tn were created while the recursion unwound towards parent(tn): deeper sources first, then in the source's own
// scan order (b_ord).  So (tn ascending, source descending, b_ord ascending) is the idx order.
#include "sub_kernels.hpp"

#include "context.hpp"

#include "segtree.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>

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

namespace
{
// A table of this stage.  Their sizes come out one after the other (the bracket table's from a scan, the records' from the
// searches ...), so they are taken from the context's arena for this stage while it has room -- it is reserved for what the
// call before needed -- and from hipMalloc beyond (freed when the stage returns).
struct DevBuf {
	void *p = nullptr;
	bool own = false;
	~DevBuf()
	{
		if (p && own)
			(void)hipFree(p);
	}
	template <typename T>
	T *get(size_t n, Arena *ar, size_t &need)
	{
		const size_t bytes = (n + 16) * sizeof(T);
		need += ((bytes + 255) & ~size_t(255)) + 256;
		if (ar && ar->fits(bytes)) {
			p = ar->take<char>(bytes);
			own = false;
		} else {
			HIP_CHECK(hipMalloc(&p, bytes));
			own = true;
		}
		return static_cast<T *>(p);
	}
};

enum : uint32_t { CL_AI_TRUNK = 0, CL_AI_BRANCH = 1, CL_ZI_TRUNK = 2, CL_ZI_BRANCH = 3 };
enum : uint32_t { E_POOL = 2u, E_LAYOUT = 8u };

// device view of the spanning forest (T-space: global tree vertex idx) and of the dense PVST output
struct SubT {
	uint32_t T, C, NB0, NB;
	const uint32_t *voff, *c_ntree, *doff, *c_npvst;
	const uint32_t *size, *gp, *nchild, *depth, *gid;
	const uint8_t *flags;
	const uint32_t *b_src, *b_tgt, *b_ord;
	const uint32_t *o_off, *o_adj, *i_off, *i_adj; // back edges of all types by source / by target (edge = slot of the dense list)
	const uint32_t *br_cnt;	       // |brackets(v)|
	const uint32_t *O, *slot_s;	       // the ordinary edges in source order: position O[u] + r = the r-th edge of u, its slot in the dense list
	SegTree segT;			       // over the targets of the edges in source order
	const uint32_t *lo, *hi;
	const unsigned long long *ekey; // [NB0] creation key: tn << 32 | ~source
	SegTree segD;
	const uint32_t *p_ai, *p_zi, *p_parent, *p_a, *p_z;
	const uint8_t *p_fam, *p_aor, *p_zor;
	uint32_t *err;
	__device__ __forceinline__ uint32_t base_of(uint32_t c) const { return 2 * voff[c] + c; }
	__device__ __forceinline__ bool ordinary(uint32_t j) const { return j < NB0; }
	__device__ __forceinline__ uint32_t dep(uint32_t v) const { return (v < T && size[v]) ? depth[v] : 0xFFFFFFFFu; }
	// pst::Tree::is_desc, spanning_tree.cpp:560-566: d is a proper descendant of a
	__device__ __forceinline__ bool is_desc(uint32_t a, uint32_t d) const { return a < d && d < a + size[a]; }
	__device__ __forceinline__ uint32_t n_br(uint32_t v) const { return br_cnt[v]; }
	// The brackets of v (tm.get_brackets, tree_utils.hpp:64-77): the ordinary edges whose source lies strictly below v and
	// whose target lies above it = among the edges of the sources (v, v + size) those with a target idx below v, found one
	// after the other in the segment tree over the targets.  The reference MATERIALISES the table, one entry per edge and
	// vertex it spans -- 10^10 entries on a graph with large tangles; here a row is enumerated when it is asked for, in
	// source order (what needs the reference's back-edge idx order asks created_before).  The root has none.
	struct BrRange {
		uint32_t pos, hi, v;
	};
	__device__ __forceinline__ BrRange brackets(uint32_t v) const
	{
		if (v >= T || !size[v] || gp[v] == NIL)
			return BrRange{0u, 0u, v};
		return BrRange{O[v + 1], O[v + size[v]], v};
	}
	__device__ __forceinline__ uint32_t br_next(BrRange &r) const // next bracket (slot of the dense list), NIL at the end
	{
		if (r.pos >= r.hi)
			return NIL;
		const uint32_t p = seg_first_less(segT, r.pos, r.hi, r.v);
		if (p == NIL) {
			r.pos = r.hi;
			return NIL;
		}
		r.pos = p + 1;
		return slot_s[p];
	}
	__device__ uint32_t lca(uint32_t a, uint32_t b) const
	{
		if (a == b)
			return a;
		if (a > b) {
			const uint32_t x = a;
			a = b;
			b = x;
		}
		if (is_desc(a, b))
			return a;
		const uint32_t m = seg_min(segD, a + 1, b + 1);
		const uint32_t c = seg_first_less(segD, a + 1, b + 1, m + 1);
		return gp[c];
	}
	__device__ uint32_t count_ord(const uint32_t *off, const uint32_t *adj, uint32_t v) const
	{
		uint32_t k = 0;
		for (uint32_t b = off[v]; b < off[v + 1]; b++)
			k += ordinary(adj[b]) ? 1u : 0u;
		return k;
	}
	// creation order of two ordinary back edges
	__device__ __forceinline__ bool created_before(uint32_t j1, uint32_t j2) const
	{
		const unsigned long long k1 = ekey[j1], k2 = ekey[j2];
		return k1 != k2 ? k1 < k2 : b_ord[j1] < b_ord[j2];
	}
};
struct CompAt { // component of a global tree vertex / of a dense PVST slot
	const uint32_t *voff, *doff;
	uint32_t C;
	__device__ __forceinline__ uint32_t of_tree(uint32_t t) const
	{
		uint32_t lo = 0, hi = C;
		while (hi - lo > 1) {
			const uint32_t mid = (lo + hi) >> 1;
			if (2 * voff[mid] + mid <= t)
				lo = mid;
			else
				hi = mid;
		}
		return lo;
	}
	__device__ __forceinline__ uint32_t of_slot(uint32_t q) const // last c with doff[c] <= q (components without a PVST own no slot)
	{
		uint32_t lo = 0, hi = C;
		while (hi - lo > 1) {
			const uint32_t mid = (lo + hi) >> 1;
			if (doff[mid] <= q)
				lo = mid;
			else
				hi = mid;
		}
		return lo;
	}
};

// ------------------------------------------------------------------ tables
__global__ void k_sub_depth_hi0(uint32_t T, const uint32_t *__restrict__ t_size, const uint32_t *__restrict__ t_depth,
				uint32_t *__restrict__ depth, uint32_t *__restrict__ hi0, uint32_t *__restrict__ eat)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t > T)
		return;
	depth[t] = (t < T && t_size[t]) ? t_depth[t] : 0xFFFFFFFFu;
	hi0[t] = NIL;
	eat[t] = NIL;
}
// by-source and by-target counts of all back edges; hi_0 of every vertex (ordinary edges only: the others do not exist yet
// when handle_vertex reads OBE, flubbles.cpp:515-519); first slot of every source's ordinary edges in the dense list
__global__ void k_sub_edge_counts(uint32_t NB, uint32_t NB0, const uint32_t *__restrict__ b_src, const uint32_t *__restrict__ b_tgt,
				  const uint32_t *__restrict__ b_ord, uint32_t *__restrict__ ocnt, uint32_t *__restrict__ icnt,
				  uint32_t *__restrict__ hi0, uint32_t *__restrict__ eat)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= NB)
		return;
	const uint32_t u = b_src[j], w = b_tgt[j];
	atomicAdd(&ocnt[u], 1u);
	atomicAdd(&icnt[w], 1u);
	if (j < NB0) {
		atomicMin(&hi0[u], w);
		if (b_ord[j] == 0)
			eat[u] = j;
	}
}
__global__ void k_sub_edge_fill(uint32_t NB, const uint32_t *__restrict__ b_src, const uint32_t *__restrict__ b_tgt,
				const uint32_t *__restrict__ o_off, const uint32_t *__restrict__ i_off, uint32_t *__restrict__ ocur,
				uint32_t *__restrict__ icur, uint32_t *__restrict__ o_adj, uint32_t *__restrict__ i_adj)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= NB)
		return;
	const uint32_t u = b_src[j], w = b_tgt[j];
	o_adj[o_off[u] + atomicAdd(&ocur[u], 1u)] = j;
	i_adj[i_off[w] + atomicAdd(&icur[w], 1u)] = j;
}
// in-place sort of a row by one lane: insertion sort while the row is short, heapsort beyond (rows of thousands of entries
// exist on tangled graphs; their order out of the atomics is arbitrary)
template <typename Less>
__device__ void sort_row(uint32_t *a, uint32_t n, Less less)
{
	if (n <= 24) {
		for (uint32_t i = 1; i < n; i++) {
			const uint32_t y = a[i];
			uint32_t k = i;
			while (k > 0 && less(y, a[k - 1])) {
				a[k] = a[k - 1];
				k--;
			}
			a[k] = y;
		}
		return;
	}
	auto sift = [&](uint32_t root, uint32_t end) {
		const uint32_t v = a[root];
		for (;;) {
			uint32_t c = 2 * root + 1;
			if (c >= end)
				break;
			if (c + 1 < end && less(a[c], a[c + 1]))
				c++;
			if (!less(v, a[c]))
				break;
			a[root] = a[c];
			root = c;
		}
		a[root] = v;
	};
	for (uint32_t i = n / 2; i-- > 0;)
		sift(i, n);
	for (uint32_t end = n - 1; end > 0; end--) {
		const uint32_t x = a[0];
		a[0] = a[end];
		a[end] = x;
		sift(0, end);
	}
}
// literal hi of every vertex: min over the subtree of hi_0, the root once a simplifying edge was added at or below
__global__ void k_sub_hi(uint32_t T, const uint32_t *__restrict__ size, const SegTree segH, const uint32_t *__restrict__ simp_ps,
			 const CompAt comp, uint32_t *__restrict__ hi)
{
	const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= T)
		return;
	const uint32_t sz = size[v];
	if (!sz) {
		hi[v] = NIL;
		return;
	}
	if (simp_ps[v + sz] != simp_ps[v]) {
		const uint32_t c = comp.of_tree(v);
		hi[v] = 2 * comp.voff[c] + c;
		return;
	}
	hi[v] = seg_min(segH, v, v + sz);
}
// creation keys: one lane per vertex hands its ordinary edges (a stretch of the dense list in scan order) their tn
__global__ void k_sub_edge_keys(uint32_t T, const uint32_t *__restrict__ size, const uint32_t *__restrict__ out_ord,
				const uint32_t *__restrict__ eat, const uint32_t *__restrict__ wbefore, unsigned long long *__restrict__ ekey)
{
	const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
	if (u >= T || !size[u] || !out_ord[u])
		return;
	const uint32_t at = eat[u], n = out_ord[u], end = u + size[u];
	uint32_t r = 0;
	for (uint32_t c = u + 1; c < end && r < n; c += max(size[c], 1u)) {
		const uint32_t k = min(wbefore[c], n - r); // edges of u created right before c was discovered
		for (uint32_t x = 0; x < k; x++)
			ekey[at + r + x] = ((unsigned long long)c << 32) | (0xFFFFFFFFu - u);
		r += k;
	}
	for (; r < n; r++)
		ekey[at + r] = ((unsigned long long)end << 32) | (0xFFFFFFFFu - u);
}
// |brackets(v)| (count_brackets, tree_utils.cpp:531-574): ordinary edges from strictly below v to strictly above it; the
// root has none (the fill loop stops at it), self loops are in nobody's table (oracle: "leaf subflubble passes")
__global__ void k_sub_br_counts(uint32_t T, const uint32_t *__restrict__ size, const uint32_t *__restrict__ gp, const uint32_t *__restrict__ P,
				const uint32_t *__restrict__ out_ord, const uint32_t *__restrict__ nself, uint32_t *__restrict__ cnt)
{
	const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v > T)
		return;
	uint32_t n = 0;
	if (v < T && size[v] && gp[v] != NIL)
		n = (P[v + size[v]] - P[v]) - (out_ord[v] - nself[v]);
	cnt[v] = n;
}
// the ordinary edges in source order (a source's edges are a stretch of the dense list in scan order; the stretches
// themselves lie in no order): position O[u] + r holds the r-th edge of u
__global__ void k_sub_edges_by_source(uint32_t T, const uint32_t *__restrict__ size, const uint32_t *__restrict__ out_ord,
				      const uint32_t *__restrict__ eat, const uint32_t *__restrict__ O, const uint32_t *__restrict__ b_tgt,
				      uint32_t *__restrict__ tgt_s, uint32_t *__restrict__ slot_s)
{
	const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
	if (u >= T || !size[u])
		return;
	const uint32_t n = out_ord[u], at = eat[u], o = O[u];
	for (uint32_t r = 0; r < n; r++) {
		tgt_s[o + r] = b_tgt[at + r];
		slot_s[o + r] = at + r;
	}
}
// LoA without self-loop back edges (see the proof in oracle/povu_oracle_sub.inc): lo[v] = the deepest target t < v of an
// ordinary edge whose source lies behind v -- the largest left end among the intervals (t, s) that hold v.  Every edge marks
// the canonical nodes of its interval in a max-tree over the vertex indices, every vertex reads the nodes above its leaf.
__global__ void k_sub_lo_mark(uint32_t NB0, const uint32_t *__restrict__ b_src, const uint32_t *__restrict__ b_tgt, uint32_t P2,
			      uint32_t *__restrict__ tree)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= NB0)
		return;
	const uint32_t sv = b_src[j], tv = b_tgt[j];
	if (sv <= tv + 1)
		return;
	for (uint32_t l = tv + 1 + P2, r = sv + P2; l < r; l >>= 1, r >>= 1) {
		// (the nodes near the root are everybody's: most edges find a larger mark there already and need no atomic)
		if (l & 1) {
			if (tree[l] < tv + 1)
				atomicMax(&tree[l], tv + 1);
			l++;
		}
		if (r & 1) {
			--r;
			if (tree[r] < tv + 1)
				atomicMax(&tree[r], tv + 1);
		}
	}
}
__global__ void k_sub_lo_query(uint32_t T, const uint32_t *__restrict__ size, uint32_t P2, const uint32_t *__restrict__ tree,
			       const uint32_t *__restrict__ nself, const CompAt comp, uint32_t *__restrict__ lo, uint32_t *__restrict__ has_self)
{
	const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= T)
		return;
	if (!size[v]) {
		lo[v] = NIL;
		return;
	}
	// A back edge whose source is its own target stays in the reference's heap as a stale entry (it is pushed AFTER lo[v] is
	// taken and never popped).  The only one from_bd ever makes is the 0 -> 0 edge of a tip-less root (spanning_tree.cpp:433-438;
	// a self loop of the graph is stored as (side, other side): two vertices): pushed at the first tree vertex, which the
	// descending sweep visits LAST -- nobody reads the heap after it, the closed form stands.  Only a self edge anywhere else
	// (none can arise; kept as the defined answer should one ever) sends its component through the literal heap below.
	if (nself[v]) {
		const uint32_t c = comp.of_tree(v);
		if (v != 2 * comp.voff[c] + c)
			has_self[c] = 1; // (this component's values are overwritten by the literal heap below)
	}
	uint32_t m = 0;
	for (uint32_t x = v + P2; x >= 1; x >>= 1)
		m = max(m, tree[x]);
	lo[v] = m ? m - 1 : NIL;
}
// compute_LoA, tree_utils.cpp:224-273, with the heap steps of libstdc++ (push_heap; pop_heap = __adjust_heap to the bottom,
// then __push_heap): one lane per component THAT HAS a self-loop back edge, its heap in the stretch of `heap` its ordinary
// edges number
__global__ void k_sub_lo(uint32_t C, const uint32_t *__restrict__ voff, const uint32_t *__restrict__ c_ntree,
			 const uint32_t *__restrict__ size, const uint32_t *__restrict__ depth, const uint32_t *__restrict__ out_ord,
			 const uint32_t *__restrict__ eat, const uint32_t *__restrict__ b_tgt, const uint32_t *__restrict__ O,
			 const uint32_t *__restrict__ has_self, uint32_t *__restrict__ heap_all, uint32_t *__restrict__ lo)
{
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= C || !c_ntree[c] || !has_self[c])
		return;
	const uint32_t base = 2 * voff[c] + c, N = c_ntree[c];
	uint32_t *heap = heap_all + O[base];
	uint32_t hn = 0;
	for (uint32_t v = base + N; v-- > base;) {
		if (!size[v]) {
			lo[v] = NIL;
			continue;
		}
		while (hn && heap[0] == v) {
			const uint32_t value = heap[hn - 1], len = hn - 1;
			hn = len;
			if (!len)
				break;
			uint32_t hole = 0, child = 0;
			while (child < (len - 1) / 2) {
				child = 2 * (child + 1);
				if (depth[heap[child]] < depth[heap[child - 1]])
					child--;
				heap[hole] = heap[child];
				hole = child;
			}
			if ((len & 1) == 0 && child == (len - 2) / 2) {
				child = 2 * (child + 1);
				heap[hole] = heap[child - 1];
				hole = child - 1;
			}
			while (hole > 0 && depth[heap[(hole - 1) / 2]] < depth[value]) {
				heap[hole] = heap[(hole - 1) / 2];
				hole = (hole - 1) / 2;
			}
			heap[hole] = value;
		}
		lo[v] = hn ? heap[0] : NIL;
		const uint32_t n = out_ord[v], at = eat[v];
		for (uint32_t r = 0; r < n; r++) { // OBE(v) in back-edge idx order = the source's scan order
			const uint32_t x = b_tgt[at + r];
			uint32_t i = hn++;
			while (i && depth[heap[(i - 1) / 2]] < depth[x]) {
				heap[i] = heap[(i - 1) / 2];
				i = (i - 1) / 2;
			}
			heap[i] = x;
		}
	}
}

// ------------------------------------------------------------------ find_concealed: the search (concealed.cpp:234-921)
struct Slub {
	uint32_t q, loc, sl, be_src; // flubble (dense slot), cl_e, the slubble's tree vertex, ai_trunk: the back edge's source
};
// compute_m, concealed.cpp:234-283
__device__ uint32_t cn_compute_m(const SubT &t, uint32_t ii, uint32_t ji)
{
	if (t.i_off[ii + 1] == t.i_off[ii])
		return ii;
	uint32_t best = NIL, best_h = 0;
	for (uint32_t b = t.i_off[ii]; b < t.i_off[ii + 1]; b++) {
		const uint32_t j = t.i_adj[b];
		if (!t.ordinary(j))
			continue;
		const uint32_t l = t.lca(t.b_src[j], ji);
		if (t.depth[l] >= t.depth[ji])
			continue;
		if (t.depth[l] > best_h) {
			best = l;
			best_h = t.depth[l];
		}
	}
	return best == NIL ? ii : best;
}
// compute_n, concealed.cpp:285-328
__device__ uint32_t cn_compute_n(const SubT &t, uint32_t ii, uint32_t ji)
{
	if (t.count_ord(t.o_off, t.o_adj, ji) == 0)
		return ji;
	uint32_t lowest = ii;
	for (uint32_t b = t.o_off[ji]; b < t.o_off[ji + 1]; b++) {
		const uint32_t j = t.o_adj[b];
		if (!t.ordinary(j))
			continue;
		if (t.depth[t.b_tgt[j]] > t.depth[lowest])
			lowest = t.b_tgt[j];
	}
	return lowest == ii ? ji : lowest;
}
// can_contain, concealed.cpp:348-384
__device__ bool cn_can_contain(const SubT &t, uint32_t ii, uint32_t ji, uint32_t m, uint32_t n)
{
	if (t.dep(t.lo[ji]) < t.depth[ii])
		return false;
	uint32_t ibe_ii = 0;
	for (uint32_t b = t.i_off[ii]; b < t.i_off[ii + 1]; b++) {
		const uint32_t j = t.i_adj[b];
		ibe_ii += (t.ordinary(j) && t.b_src[j] != ji) ? 1u : 0u;
	}
	if (m == ii && n == ji && ibe_ii < 2 && t.count_ord(t.i_off, t.i_adj, ji) == 0 && t.nchild[ji] < 3)
		return false;
	return true;
}
// ai_trunk, concealed.cpp:388-522
__device__ bool cn_ai_trunk(const SubT &t, uint32_t m, uint32_t n, uint32_t ai, uint32_t zi, uint32_t &be_src, uint32_t &l_out)
{
	if (t.depth[m] > t.depth[n])
		return false;
	uint32_t best_l = NIL, best_src = NIL;
	for (uint32_t b = t.i_off[ai]; b < t.i_off[ai + 1]; b++) {
		const uint32_t j = t.i_adj[b];
		if (!t.ordinary(j))
			continue;
		const uint32_t src = t.b_src[j], l = t.lca(src, zi);
		if (t.depth[l] > t.depth[m])
			continue;
		bool ell_br = true;
		{
			SubT::BrRange br = t.brackets(l);
			for (uint32_t e = t.br_next(br); e != NIL && ell_br; e = t.br_next(br))
				ell_br = t.depth[t.b_tgt[e]] <= t.depth[ai];
		}
		const bool cond_iii = t.o_off[l + 1] > t.o_off[l] || t.nchild[l] > 1;
		if (!ell_br && cond_iii)
			continue;
		// (the reference keeps the last of equal LCAs in back-edge idx order; which source survives shows nowhere -- it only
		// enters Concealed::bounds_, oracle/povu_oracle_sub.inc -- so any fixed choice does: the largest source)
		if (best_l == NIL || l > best_l || (l == best_l && src > best_src)) {
			best_l = l;
			best_src = src;
		}
	}
	if (best_l == NIL)
		return false;
	be_src = best_src;
	l_out = best_l;
	return true;
}
__device__ uint32_t lca_of_bracket_srcs(const SubT &t, uint32_t c)
{
	uint32_t d = NIL;
	SubT::BrRange br = t.brackets(c);
	for (uint32_t e = t.br_next(br); e != NIL; e = t.br_next(br)) {
		const uint32_t src = t.b_src[e];
		d = d == NIL ? src : t.lca(d, src);
	}
	return d;
}
// the searches push their slubbles through this: out == null counts
struct SlubOut {
	Slub *out;
	uint32_t n;
	__device__ __forceinline__ void push(uint32_t q, uint32_t loc, uint32_t sl, uint32_t be_src)
	{
		if (out)
			out[n] = Slub{q, loc, sl, be_src};
		n++;
	}
};
// ai_branches, concealed.cpp:525-583
__device__ void cn_ai_branches(const SubT &t, uint32_t q, uint32_t ai, uint32_t zi, SlubOut &o)
{
	if (t.nchild[zi] < 2)
		return;
	const uint32_t end = zi + t.size[zi];
	for (uint32_t c = zi + 1; c < end; c += max(t.size[c], 1u)) {
		if (!(t.hi[c] == t.lo[c] && t.hi[c] == ai))
			continue;
		if (t.n_br(c) < 2)
			continue;
		const uint32_t d = lca_of_bracket_srcs(t, c);
		if (t.n_br(d) > 0)
			o.push(q, CL_AI_BRANCH, d, NIL);
	}
}
// override_ji_trunk, concealed.cpp:617-697
__device__ uint32_t cn_override_ji_trunk(const SubT &t, uint32_t m, uint32_t n, uint32_t ii, uint32_t ji)
{
	if (t.depth[m] > t.depth[n])
		return NIL;
	uint32_t min_v = n;
	const uint32_t end = ji + t.size[ji];
	for (uint32_t c = ji + 1; c < end; c += max(t.size[c], 1u)) {
		if (t.dep(t.hi[c]) < t.depth[ii])
			continue;
		if (t.n_br(c) != 1)
			continue;
		SubT::BrRange br1 = t.brackets(c);
		const uint32_t y = t.b_tgt[t.br_next(br1)];
		if (!(t.depth[m] < t.depth[y] && t.depth[n] > t.depth[y]))
			continue;
		bool valid = true;
		SubT::BrRange bry = t.brackets(y);
		for (uint32_t j = t.br_next(bry); j != NIL && valid; j = t.br_next(bry))
			if (t.depth[t.b_src[j]] < t.depth[ji] || t.depth[t.b_tgt[j]] > t.depth[ii])
				valid = false;
		if (valid && t.depth[y] < t.depth[min_v])
			min_v = y;
	}
	return min_v == n ? NIL : min_v;
}
// ji_trunk, concealed.cpp:699-760 (the rest of the function is behind a return)
__device__ uint32_t cn_ji_trunk(const SubT &t, uint32_t m, uint32_t n, uint32_t ii, uint32_t ji)
{
	if (t.depth[m] > t.depth[n])
		return NIL;
	const uint32_t r = cn_override_ji_trunk(t, m, n, ii, ji);
	if (r != NIL)
		return r;
	uint32_t best = NIL;
	for (uint32_t b = t.o_off[ji]; b < t.o_off[ji + 1]; b++) {
		const uint32_t j = t.o_adj[b];
		if (!t.ordinary(j))
			continue;
		const uint32_t x = t.b_tgt[j];
		if (t.depth[x] < t.depth[n])
			continue;
		if (best == NIL || t.depth[x] < t.depth[best])
			best = x;
	}
	return best;
}
// ji_branches, concealed.cpp:790-921
__device__ void cn_ji_branches(const SubT &t, uint32_t q, uint32_t ii, uint32_t ji, uint32_t n, SlubOut &o)
{
	if (t.nchild[ji] < 2 || ji == n)
		return;
	bool have_main = false;
	for (uint32_t b = t.i_off[ii]; b < t.i_off[ii + 1] && !have_main; b++) {
		const uint32_t j = t.i_adj[b];
		have_main = t.ordinary(j) && t.b_src[j] == ji;
	}
	const uint32_t end = ji + t.size[ji];
	for (int pass = 0; pass < 2; pass++) {
		for (uint32_t c = ji + 1; c < end; c += max(t.size[c], 1u)) {
			uint32_t count = 0;
			bool into_ji = false;
			{
				SubT::BrRange br = t.brackets(c);
				for (uint32_t e = t.br_next(br); e != NIL; e = t.br_next(br)) {
					if (t.b_tgt[e] != ji)
						count++;
					else
						into_ji = true;
				}
			}
			if (count != 1 || into_ji != (pass == 0))
				continue;
			if (pass == 0) {
				// the deepest source of a bracket into ji; among equally deep ones the reference keeps the first in back-edge
				// idx order (depth[src] > depth[lowest], concealed.cpp:846)
				uint32_t lowest = c, low_e = NIL;
				SubT::BrRange br = t.brackets(c);
				for (uint32_t j = t.br_next(br); j != NIL; j = t.br_next(br)) {
					if (t.b_tgt[j] != ji)
						continue;
					const uint32_t sv = t.b_src[j];
					if (t.depth[sv] > t.depth[lowest] || (low_e != NIL && t.depth[sv] == t.depth[lowest] && t.created_before(j, low_e))) {
						lowest = sv;
						low_e = j;
					}
				}
				o.push(q, CL_ZI_BRANCH, lowest, NIL);
			} else {
				SubT::BrRange br = t.brackets(c);
				const uint32_t j = t.br_next(br); // (its only bracket)
				const uint32_t src = t.b_src[j], tgt = t.b_tgt[j];
				bool cond_i = false;
				uint32_t d = src;
				while (d != ji) {
					if (t.nchild[d] > 1) {
						cond_i = true;
						break;
					}
					d = t.gp[d];
				}
				const uint32_t alpha = t.n_br(tgt);
				if (alpha != 0 && have_main)
					continue;
				if (alpha != 1)
					continue;
				if (cond_i)
					o.push(q, CL_ZI_BRANCH, d, NIL);
			}
		}
	}
}
// find_concealed's loop body for the flubble in dense slot q (concealed.cpp:1203-1237); returns its slubbles
__device__ uint32_t cn_search(const SubT &t, const CompAt &comp, uint32_t q, Slub *out, uint32_t *mn)
{
	if (t.p_fam[q] != FAM_FLUBBLE)
		return 0;
	const uint32_t c = comp.of_slot(q), base = t.base_of(c);
	const uint32_t ai = base + t.p_ai[q], zi = base + t.p_zi[q];
	const uint32_t m = cn_compute_m(t, ai, zi), n = cn_compute_n(t, ai, zi);
	if (mn) {
		mn[2 * (size_t)q] = m;
		mn[2 * (size_t)q + 1] = n;
	}
	if (!cn_can_contain(t, ai, zi, m, n))
		return 0;
	SlubOut o{out, 0};
	uint32_t be_src, l;
	if (cn_ai_trunk(t, m, n, ai, zi, be_src, l))
		o.push(q, CL_AI_TRUNK, l, be_src);
	cn_ai_branches(t, q, ai, zi, o);
	const uint32_t tb = cn_ji_trunk(t, m, n, ai, zi);
	if (tb != NIL)
		o.push(q, CL_ZI_TRUNK, tb, NIL);
	cn_ji_branches(t, q, ai, zi, n, o);
	return o.n;
}
__global__ void k_sub_cn_count(uint32_t Q, const SubT t, const CompAt comp, uint32_t *__restrict__ cnt, uint32_t *__restrict__ mn)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q > Q)
		return;
	cnt[q] = q < Q ? cn_search(t, comp, q, nullptr, mn) : 0u;
}
__global__ void k_sub_cn_emit(uint32_t Q, const SubT t, const CompAt comp, const uint32_t *__restrict__ off, Slub *__restrict__ out)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= Q || off[q + 1] == off[q])
		return;
	cn_search(t, comp, q, out + off[q], nullptr);
}

// ------------------------------------------------------------------ find_smothered: the search (smothered.cpp:61-318)
struct Smo {
	uint32_t sm_st, b_up, b_lo, flags; // flags: 1 = cn_b is the ancestor, 2 = the concealed vertex is a g
};
struct SmoOut {
	Smo *out;
	uint32_t n;
	__device__ __forceinline__ void push(uint32_t sm_st, bool anc, bool is_g, uint32_t up, uint32_t lo)
	{
		if (out)
			out[n] = Smo{sm_st, up, lo, (anc ? 1u : 0u) | (is_g ? 2u : 0u)};
		n++;
	}
};
// compute_bounds, smothered.cpp:37-48
__device__ __forceinline__ void smo_bounds(const SubT &t, uint32_t cn_st, uint32_t sm_st, uint32_t &up, uint32_t &lo)
{
	if (t.is_desc(cn_st, sm_st))
		up = cn_st, lo = sm_st;
	else
		up = sm_st, lo = cn_st;
}
// the next distinct value above `last` of key[adj[..]] over a row, NIL when there is none (std::set iteration without a set)
__device__ uint32_t next_distinct(const uint32_t *off, const uint32_t *adj, const uint32_t *key, uint32_t v, uint32_t last, bool first)
{
	uint32_t best = NIL;
	for (uint32_t b = off[v]; b < off[v + 1]; b++) {
		const uint32_t x = key[adj[b]];
		if ((first || x > last) && (best == NIL || x < best))
			best = x;
	}
	return best;
}
__device__ uint32_t smo_search(const SubT &t, const CompAt &comp, const Slub &cn, Smo *out)
{
	SmoOut o{out, 0};
	const uint32_t c = comp.of_slot(cn.q), base = t.base_of(c);
	const uint32_t ai = base + t.p_ai[cn.q], zi = base + t.p_zi[cn.q], sl = cn.sl;
	switch (cn.loc) {
	case CL_AI_TRUNK: { // g::trunk, smothered.cpp:61-134
		const uint32_t end = sl + t.size[sl];
		for (uint32_t ch = sl + 1; ch < end; ch += max(t.size[ch], 1u)) {
			if (t.n_br(ch) == 0)
				continue;
			// the LAST bracket of the row in back-edge idx order (be_idx_, smothered.cpp:94-101) and whether all have its source
			uint32_t last = NIL;
			{
				SubT::BrRange br = t.brackets(ch);
				for (uint32_t e = t.br_next(br); e != NIL; e = t.br_next(br))
					if (last == NIL || t.created_before(last, e))
						last = e;
			}
			const uint32_t src = t.b_src[last], tgt = t.b_tgt[last];
			bool one_src = true;
			{
				SubT::BrRange br = t.brackets(ch);
				for (uint32_t e = t.br_next(br); e != NIL && one_src; e = t.br_next(br))
					one_src = t.b_src[e] == src;
			}
			if (!(one_src && t.depth[tgt] > t.depth[ai]))
				continue;
			const uint32_t brch = t.lca(zi, src);
			if (t.n_br(src) == 0)
				o.push(tgt, false, true, brch, src);
			else
				o.push(src, true, true, brch, src);
		}
		break;
	}
	case CL_AI_BRANCH: { // g::branch, :136-199
		const uint32_t t0 = next_distinct(t.o_off, t.o_adj, t.b_tgt, sl, 0, true);
		if (t0 == NIL || t0 != ai || next_distinct(t.o_off, t.o_adj, t.b_tgt, sl, t0, false) != NIL)
			break;
		// the brackets of sl that end at ai, in back-edge idx order (the row is enumerated in source order: the next one is
		// the earliest created behind the last); each gives one vertex per bracket of its source that ends at sl
		uint32_t prev_e = NIL;
		for (;;) {
			uint32_t best = NIL;
			SubT::BrRange br = t.brackets(sl);
			for (uint32_t j = t.br_next(br); j != NIL; j = t.br_next(br)) {
				if (t.b_tgt[j] != ai)
					continue;
				if (prev_e != NIL && !t.created_before(prev_e, j))
					continue;
				if (best == NIL || t.created_before(j, best))
					best = j;
			}
			if (best == NIL)
				break;
			prev_e = best;
			const uint32_t src = t.b_src[best];
			SubT::BrRange br2 = t.brackets(src);
			for (uint32_t r = t.br_next(br2); r != NIL; r = t.br_next(br2))
				if (t.b_tgt[r] == sl) {
					uint32_t up, lo;
					smo_bounds(t, sl, src, up, lo);
					o.push(src, true, true, up, lo);
				}
		}
		break;
	}
	case CL_ZI_TRUNK: { // s::trunk, :217-270: the sources by their LCA with zi, ascending; an LCA with one source gives a vertex
		uint32_t last_l = 0;
		bool first_l = true;
		for (;;) {
			// the next LCA value above last_l among the kept sources
			uint32_t best_l = NIL;
			for (uint32_t b = t.i_off[sl]; b < t.i_off[sl + 1]; b++) {
				const uint32_t src = t.b_src[t.i_adj[b]];
				if (t.depth[src] >= t.depth[zi])
					continue;
				const uint32_t l = t.lca(zi, src);
				if (l == src)
					continue;
				if ((first_l || l > last_l) && (best_l == NIL || l < best_l))
					best_l = l;
			}
			if (best_l == NIL)
				break;
			// its distinct sources
			uint32_t cnt = 0, only = NIL, last_s = 0;
			bool first_s = true;
			for (;;) {
				uint32_t best_s = NIL;
				for (uint32_t b = t.i_off[sl]; b < t.i_off[sl + 1]; b++) {
					const uint32_t src = t.b_src[t.i_adj[b]];
					if (t.depth[src] >= t.depth[zi])
						continue;
					const uint32_t l = t.lca(zi, src);
					if (l != best_l || l == src)
						continue;
					if ((first_s || src > last_s) && (best_s == NIL || src < best_s))
						best_s = src;
				}
				if (best_s == NIL)
					break;
				cnt++;
				only = best_s;
				last_s = best_s;
				first_s = false;
			}
			if (cnt == 1)
				o.push(only, false, false, best_l, only);
			last_l = best_l;
			first_l = false;
		}
		break;
	}
	case CL_ZI_BRANCH: { // s::branch, :272-318
		uint32_t last = 0;
		bool first = true;
		for (;;) {
			const uint32_t src = next_distinct(t.i_off, t.i_adj, t.b_src, sl, last, first);
			if (src == NIL)
				break;
			uint32_t l2 = 0;
			bool f2 = true;
			for (;;) {
				const uint32_t x = next_distinct(t.i_off, t.i_adj, t.b_src, src, l2, f2);
				if (x == NIL)
					break;
				uint32_t up, lo;
				smo_bounds(t, x, sl, up, lo);
				o.push(x, true, false, up, lo);
				l2 = x;
				f2 = false;
			}
			l2 = 0;
			f2 = true;
			for (;;) {
				const uint32_t x = next_distinct(t.o_off, t.o_adj, t.b_tgt, src, l2, f2);
				if (x == NIL)
					break;
				if (x != zi) {
					uint32_t up, lo;
					smo_bounds(t, x, sl, up, lo);
					o.push(x, false, false, up, lo);
				}
				l2 = x;
				f2 = false;
			}
			last = src;
			first = false;
		}
		break;
	}
	}
	return o.n;
}
__global__ void k_sub_smo_count(uint32_t NC, const SubT t, const CompAt comp, const Slub *__restrict__ cn, uint32_t *__restrict__ cnt)
{
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k > NC)
		return;
	cnt[k] = k < NC ? smo_search(t, comp, cn[k], nullptr) : 0u;
}
__global__ void k_sub_smo_emit(uint32_t NC, const SubT t, const CompAt comp, const Slub *__restrict__ cn, const uint32_t *__restrict__ off,
			       Smo *__restrict__ out)
{
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= NC || off[k + 1] == off[k])
		return;
	smo_search(t, comp, cn[k], out + off[k]);
}

// ------------------------------------------------------------------ the splice, one wave per component
// The extended PVST of component c lives in X-space: x = xoff[c] + (index inside the component).  Children of a vertex: a
// vector {begin, size, capacity} in the component's stretch of the pool, doubled when full (pvst.hpp:860-885).
struct XArrays {
	uint8_t *fam, *or1, *or2, *route, *loc;
	uint32_t *id1, *id2, *ai, *zi, *sl, *b_up, *b_lo; // ai / zi: global tree vertex; b_up / b_lo: bounds_t (global tree vertex, or PVST idx for a midi)
	uint32_t *vbeg, *vn, *vcap;
	uint32_t *pool;
};
// One WAVE works on the children vectors of ONE vertex and of the vertices it creates (see "the splice, in parallel"
// below).  The control flow is the reference's, followed by all 64 lanes alike (every decision is read from the same address
// by every lane); lane 0 does the stores, and the two loops that are long on a flubble with very many children -- copying a
// vector that grows, filtering the children that leave -- are shared out over the lanes.
struct Splice {
	XArrays x;
	uint32_t xb;	    // first X slot of the component
	uint32_t *pool_top; // bump pointer inside the component's stretch of the pool: shared by the waves of the component
	uint32_t pool_end;
	uint32_t *err;
	uint32_t lane;
	__device__ __forceinline__ void sync() const { __syncthreads(); } // (one wave a workgroup)
	// room for `extra` more entries in the vector of `parent`
	__device__ bool reserve(uint32_t parent, uint32_t extra)
	{
		const uint32_t p = xb + parent, n = x.vn[p], cap = x.vcap[p];
		if (n + extra <= cap)
			return true;
		uint32_t ncap = cap ? cap : 4;
		while (ncap < n + extra)
			ncap *= 2;
		uint32_t at = 0;
		if (lane == 0)
			at = ncap < n ? NIL : atomicAdd(pool_top, ncap);
		at = __shfl(at, 0);
		if (at == NIL || at + ncap > pool_end || at + ncap < at) {
			if (lane == 0)
				atomicOr(err, E_POOL);
			return false;
		}
		const uint32_t from = x.vbeg[p];
		for (uint32_t k = lane; k < n; k += 64)
			x.pool[at + k] = x.pool[from + k];
		sync();
		if (lane == 0) {
			x.vbeg[p] = at;
			x.vcap[p] = ncap;
		}
		sync();
		return true;
	}
	__device__ bool push(uint32_t parent, uint32_t child)
	{
		if (!reserve(parent, 1))
			return false;
		const uint32_t p = xb + parent;
		if (lane == 0) {
			x.pool[x.vbeg[p] + x.vn[p]] = child;
			x.vn[p] = x.vn[p] + 1;
		}
		sync();
		return true;
	}
	// del_edge: erase the first match; the slot behind the new end keeps its old value, as vector::erase leaves it
	__device__ void erase(uint32_t parent, uint32_t child)
	{
		const uint32_t p = xb + parent, b = x.vbeg[p], n = x.vn[p];
		if (lane == 0)
			for (uint32_t k = 0; k < n; k++)
				if (x.pool[b + k] == child) {
					for (uint32_t a = k; a + 1 < n; a++)
						x.pool[b + a] = x.pool[b + a + 1];
					x.vn[p] = n - 1;
					break;
				}
		sync();
	}
	__device__ __forceinline__ bool fl_like(uint32_t v) const
	{
		const uint8_t f = x.fam[xb + v];
		return f == FAM_FLUBBLE || f == FAM_TINY || f == FAM_PARALLEL;
	}
	// The nestings of add_concealed / add_midi run over a COPY of the first nch children of f and move those that `leaves`
	// says so, in order, under `dest` (to_child: the other way round -- `dest` is pushed into the vector of each of them,
	// concealed.cpp:1077: that push is the CHILD's, see k_sub_pin; here the child only leaves the vector of f).  A stable
	// filter, 64 children a round; returns false when the pool ran out.
	template <typename Leaves>
	__device__ bool filter(uint32_t f, uint32_t nch, uint32_t dest, bool to_child, Leaves &&leaves)
	{
		const uint32_t pf = xb + f;
		uint32_t w = 0;
		for (uint32_t r0 = 0; r0 < nch; r0 += 64) {
			const uint32_t r = r0 + lane, fb = x.vbeg[pf];
			const bool valid = r < nch;
			const uint32_t ch = valid ? x.pool[fb + r] : 0u;
			const bool moved = valid && leaves(ch);
			const unsigned long long mm = __ballot(moved), km = __ballot(valid && !moved);
			const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
			sync(); // (every lane holds its child: the slots may be written now)
			if (valid && !moved)
				x.pool[fb + w + (uint32_t)__popcll(km & below)] = ch;
			w += (uint32_t)__popcll(km);
			if (mm) {
				if (!to_child) {
					const uint32_t cnt = (uint32_t)__popcll(mm);
					if (!reserve(dest, cnt))
						return false;
					const uint32_t pd = xb + dest;
					if (moved)
						x.pool[x.vbeg[pd] + x.vn[pd] + (uint32_t)__popcll(mm & below)] = ch;
					sync();
					if (lane == 0)
						x.vn[pd] = x.vn[pd] + cnt;
					sync();
				}
			}
			sync();
		}
		if (lane == 0)
			x.vn[pf] = w;
		sync();
		return true;
	}
};
__device__ __forceinline__ void side_id_or(const SubT &t, uint32_t v, bool fwd_is_r, uint32_t &id, uint8_t &orr)
{
	id = t.gid[v];
	const bool is_r = (t.flags[v] & TF_TYPE_MASK) == 1u;
	orr = (is_r == fwd_is_r) ? 0 : 1;
}
// the PVST as find_flubbles left it, in X-space: fields of every vertex and the capacity of its children vector (one lane
// per dense PVST slot); then the vectors themselves, children in ascending idx (add_flubbles attaches them in that order)
__global__ void k_sub_x_init(uint32_t Q, const SubT t, const CompAt comp, const uint32_t *__restrict__ xoff, XArrays X)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= Q)
		return;
	const uint32_t c = comp.of_slot(q), v = q - t.doff[c], x = xoff[c] + v, base = t.base_of(c);
	X.fam[x] = t.p_fam[q];
	X.loc[x] = 0;
	X.sl[x] = X.b_up[x] = X.b_lo[x] = NIL;
	if (v) {
		X.id1[x] = t.p_a[q], X.or1[x] = t.p_aor[q];
		X.id2[x] = t.p_z[q], X.or2[x] = t.p_zor[q];
		X.route[x] = 'L';
		X.ai[x] = base + t.p_ai[q], X.zi[x] = base + t.p_zi[q];
	} else {
		X.id1[x] = X.id2[x] = NIL;
		X.or1[x] = X.or2[x] = 0;
		X.route[x] = 0;
		X.ai[x] = X.zi[x] = NIL;
	}
}
// where the children of X slot x begin among the sorted (parent slot, child) pairs = how many pairs have a smaller key: a
// binary search per slot (counting them with atomics put a million adds on the word of a chromosome's root)
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *__restrict__ a, uint32_t n, uint32_t x)
{
	uint32_t lo = 0, hi = n;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (a[mid] < x)
			lo = mid + 1;
		else
			hi = mid;
	}
	return lo;
}
__global__ void k_sub_x_vbeg(uint32_t Q, const SubT t, const CompAt comp, const uint32_t *__restrict__ xoff,
			     const uint32_t *__restrict__ poff, const uint32_t *__restrict__ key, uint32_t *__restrict__ cap_ps, XArrays X)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= Q)
		return;
	const uint32_t c = comp.of_slot(q), x = xoff[c] + (q - t.doff[c]);
	const uint32_t b = lower_bound_u32(key, Q, x), e = lower_bound_u32(key, Q, x + 1), b0 = lower_bound_u32(key, Q, xoff[c]);
	cap_ps[x] = b;
	X.vbeg[x] = poff[c] + (b - b0);
	X.vn[x] = X.vcap[x] = e - b;
}
// children in ascending idx: a stable sort of (X slot of the parent, child) over the dense slots, which are in ascending
// idx already (one lane sorting the million children of a chromosome's root took seconds)
__global__ void k_sub_x_keys(uint32_t Q, uint32_t NX, const SubT t, const CompAt comp, const uint32_t *__restrict__ xoff,
			     uint32_t *__restrict__ key, uint32_t *__restrict__ val)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= Q)
		return;
	const uint32_t c = comp.of_slot(q), v = q - t.doff[c];
	key[q] = v ? xoff[c] + t.p_parent[q] : NX; // (the roots have no parent: behind everything)
	val[q] = v;
}
__global__ void k_sub_x_place(uint32_t Q, uint32_t NX, const uint32_t *__restrict__ key, const uint32_t *__restrict__ val,
			      const uint32_t *__restrict__ cap_ps, XArrays X)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= Q)
		return;
	const uint32_t x = key[i];
	if (x >= NX)
		return;
	X.pool[X.vbeg[x] + (i - cap_ps[x])] = val[i];
}
// ------------------------------------------------------------------ the splice, in parallel
// Until round 5 one wave spliced a whole component, flubble after flubble (94 ms for the whole-genome workload: 10^4
// flubbles of a chromosome with a concealed bubble, some ten dependent trips to memory each).  What add_concealed does to
// one flubble f depends on nothing another flubble's turn writes:
//  * its records (cn[]: found before, in ascending idx) each become the vertex n0 + (index of the record in the
//    component) -- no counter;
//  * a record's nesting moves children of f under the new vertex, or (z-side trunk) pushes the new vertex into the vector
//    of the children that qualify and drops them from f's vector.  Whether a child qualifies is a function of the trees
//    alone (cn_leaves), so the CHILD can work out by itself which vertex its parent's records push into its vector: it
//    walks the parent's records in order until one takes it (k_sub_pin, one lane per PVST vertex; at most one vertex ever
//    arrives, because the record that pushes it also drops the child from the parent's vector).  The parent has the
//    smaller idx, its turn comes first: the pushed vertex stands in front of the child's own concealed vertices;
//  * the vectors grow out of the component's pool by an atomic bump.
// So: one wave per flubble that has records or takes a vertex in (k_sub_splice_cn); the flubbles that got a concealed child
// are flagged instead of listed.  find_midi looks at each of them (k_sub_midi_find, all before the first is added, as in the
// reference), a scan over their flags numbers the midi bubbles in ascending idx, k_sub_midi_add attaches them; add_smothered
// works on the vector of one concealed vertex (k_sub_smothered, one wave per concealed vertex that has records).
struct SpliceArgs {
	SubT t;
	CompAt comp;
	const uint32_t *xoff, *poff;
	uint32_t *ptop; // [C] bump pointers
	const uint32_t *cn_off;
	const Slub *cn;
	const uint32_t *mn;
	const uint32_t *smo_off;
	const Smo *smo;
	XArrays X;
	uint32_t *pin;	  // [Q] the concealed vertex the parent's records push into the vector of this one (NIL: none)
	uint8_t *act;	  // [Q] has records or takes a vertex in
	uint8_t *touched; // [Q] got a concealed vertex as a child
	uint8_t *md;	  // [Q] gets a midi bubble
	const uint32_t *md_ps; // [Q + 1] exclusive scan of md
};
// what the nestings of add_concealed ask of a child of the flubble (ai, zi, n_of_f) for the slubble sl
__device__ bool cn_leaves(const SubT &t, const XArrays &X, uint32_t xb, const Slub &sl, uint32_t ai, uint32_t zi, uint32_t n_of_f, uint32_t ch)
{
	const uint8_t fam = X.fam[xb + ch];
	if (!(fam == FAM_FLUBBLE || fam == FAM_TINY || fam == FAM_PARALLEL))
		return false;
	const uint32_t c_ai = X.ai[xb + ch], c_zi = X.zi[xb + ch];
	if (sl.loc == CL_AI_TRUNK) // nest_trunk_ai, :945-979
		return t.depth[sl.sl] > t.depth[c_zi] || t.is_desc(sl.sl, c_ai);
	if (sl.loc == CL_AI_BRANCH) { // nest_branch_ai, :984-1034
		bool has_br = false;  // a bracket of the child's zi that STARTS at ai (sic: get_src, :1006)
		SubT::BrRange br = t.brackets(c_zi);
		for (uint32_t e = t.br_next(br); e != NIL && !has_br; e = t.br_next(br))
			has_br = t.b_src[e] == ai;
		return t.is_desc(sl.sl, c_ai) && has_br;
	}
	// nest_trunk_zi, :1054-1081: (sic) the edge goes from the child to the slubble
	return t.is_desc(n_of_f, c_ai) && !t.is_desc(zi, c_zi);
}
// PVST vertices, concealed and smothered records of every component (what the host lays the splice out by)
__global__ void k_sub_comp_counts(uint32_t C, const uint32_t *__restrict__ doff, const uint32_t *__restrict__ c_npvst,
				  const uint32_t *__restrict__ cn_off, const uint32_t *__restrict__ smo_off, uint32_t *__restrict__ out)
{
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= C)
		return;
	const uint32_t n0 = c_npvst[c], q0 = doff[c];
	uint32_t ncn = 0, nsm = 0;
	if (n0) {
		const uint32_t b = cn_off[q0], e = cn_off[q0 + n0];
		ncn = e - b;
		nsm = smo_off[e] - smo_off[b];
	}
	out[3 * c] = n0, out[3 * c + 1] = ncn, out[3 * c + 2] = nsm;
}
__global__ void k_sub_ptop(uint32_t C, const uint32_t *__restrict__ poff, const uint32_t *__restrict__ c_npvst, uint32_t *__restrict__ ptop)
{
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c < C) // (the vectors of find_flubbles' vertices are in place: k_sub_x_init .. k_sub_x_place)
		ptop[c] = poff[c] + (c_npvst[c] ? c_npvst[c] - 1 : 0u);
}
__global__ void k_sub_pin(uint32_t Q, const SpliceArgs A)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= Q)
		return;
	const SubT &t = A.t;
	const uint32_t c = A.comp.of_slot(q), q0 = t.doff[c], ch = q - q0;
	uint32_t pin = NIL;
	const bool own = A.cn_off[q + 1] != A.cn_off[q];
	if (ch) { // (the root has no parent)
		const uint32_t f = t.p_parent[q], qf = q0 + f;
		const uint32_t kb = A.cn_off[qf], ke = A.cn_off[qf + 1];
		if (kb != ke) {
			const uint32_t xb = A.xoff[c], base = t.base_of(c), N = t.c_ntree[c], n0 = t.c_npvst[c], cn_b = A.cn_off[q0];
			const bool is_leaf = f < N && t.nchild[base + f] == 0; // (sic) the SPANNING TREE's vertex f, concealed.cpp:1188
			if (!is_leaf) {
				const uint32_t ai = A.X.ai[xb + f], zi = A.X.zi[xb + f], n_of_f = A.mn[2 * (size_t)qf + 1];
				for (uint32_t k = kb; k < ke; k++) {
					const Slub sl = A.cn[k];
					if (sl.loc == CL_ZI_BRANCH)
						continue;
					if (cn_leaves(t, A.X, xb, sl, ai, zi, n_of_f, ch)) { // this record takes it out of f's vector
						if (sl.loc == CL_ZI_TRUNK)
							pin = n0 + (k - cn_b);
						break;
					}
				}
			}
		}
	}
	A.pin[q] = pin;
	A.act[q] = (own || pin != NIL) ? 1 : 0;
	A.touched[q] = 0;
	A.md[q] = 0;
}
static constexpr unsigned SPLICE_WAVES = 8192; // waves of a splice kernel: each takes every SPLICE_WAVES-th entry of its list
// ---- add_concealed, concealed.cpp:925-1196: one wave per flubble on the list
__global__ void __launch_bounds__(64) k_sub_splice_cn(const uint32_t *__restrict__ n_list, const uint32_t *__restrict__ list, const SpliceArgs A)
{
	const SubT &t = A.t;
	const XArrays &X = A.X;
	const uint32_t lane = threadIdx.x, n = *n_list;
	for (uint32_t it = blockIdx.x; it < n; it += gridDim.x) {
		const uint32_t q = list[it], c = A.comp.of_slot(q), q0 = t.doff[c], n0 = t.c_npvst[c], f = q - q0;
		const uint32_t base = t.base_of(c), N = t.c_ntree[c], cn_b = A.cn_off[q0];
		Splice S{X, A.xoff[c], A.ptop + c, A.poff[c + 1], t.err, lane};
		const uint32_t xb = S.xb, pin = A.pin[q], par = f ? t.p_parent[q] : 0u;
		bool got = false;
		if (pin != NIL && par < f) { // pushed by the parent's turn, which came first (a parent has the smaller idx)
			if (!S.push(f, pin))
				return;
			got = true;
		}
		const uint32_t kb = A.cn_off[q], ke = A.cn_off[q + 1];
		if (kb != ke) {
			const uint32_t ai = X.ai[xb + f], zi = X.zi[xb + f], n_of_f = A.mn[2 * (size_t)q + 1];
			const bool is_leaf = f < N && t.nchild[base + f] == 0; // (sic) the SPANNING TREE's vertex f, concealed.cpp:1188
			for (uint32_t k = kb; k < ke; k++) {
				const Slub sl = A.cn[k];
				const uint32_t v = n0 + (k - cn_b); // the k-th slubble of the component becomes its k-th concealed vertex
				if (lane == 0) {
					const uint32_t x = xb + v;
					X.fam[x] = FAM_CONCEALED;
					X.vn[x] = X.vcap[x] = 0;
					X.vbeg[x] = 0;
					X.ai[x] = X.zi[x] = NIL;
					// gen_ai_slubble, concealed.cpp:70-152 / gen_zi_slubble, :154-206
					const bool sl_r = (t.flags[sl.sl] & TF_TYPE_MASK) == 1u, black = (t.flags[sl.sl] & TF_BLACK) != 0;
					uint32_t fl_id;
					uint8_t fl_o;
					const uint32_t sl_id = t.gid[sl.sl];
					X.loc[x] = (uint8_t)sl.loc;
					X.sl[x] = sl.sl;
					if (sl.loc == CL_AI_TRUNK || sl.loc == CL_AI_BRANCH) {
						side_id_or(t, ai, true, fl_id, fl_o);
						const bool fwd_if_r = (sl.loc == CL_AI_TRUNK) == black;
						const uint8_t sl_o = (sl_r == fwd_if_r) ? 0 : 1;
						if (sl_o == 1 && fl_o == 1) {
							X.id1[x] = sl_id, X.or1[x] = 0;
							X.id2[x] = fl_id, X.or2[x] = 0;
						} else {
							X.id1[x] = fl_id, X.or1[x] = fl_o;
							X.id2[x] = sl_id, X.or2[x] = sl_o;
						}
						X.route[x] = 'R';
						if (sl.loc == CL_AI_TRUNK) {
							if (t.is_desc(ai, sl.be_src))
								X.b_up[x] = ai, X.b_lo[x] = sl.be_src;
							else
								X.b_up[x] = sl.be_src, X.b_lo[x] = ai;
						} else {
							X.b_up[x] = sl.sl, X.b_lo[x] = NIL;
						}
					} else {
						side_id_or(t, zi, false, fl_id, fl_o);
						const uint8_t sl_o = (sl_r == black) ? 0 : 1;
						X.id1[x] = sl_id, X.or1[x] = (sl_o == 1 && fl_o == 1) ? 0 : sl_o;
						X.id2[x] = fl_id, X.or2[x] = (sl_o == 1 && fl_o == 1) ? 0 : fl_o;
						X.route[x] = 'L';
						if (t.is_desc(zi, sl.sl))
							X.b_up[x] = zi, X.b_lo[x] = sl.sl;
						else
							X.b_up[x] = sl.sl, X.b_lo[x] = zi;
					}
				}
				S.sync();
				if (!S.push(f, v))
					return;
				got = true;
				if (is_leaf || sl.loc == CL_ZI_BRANCH) // zi_branch: add_conc_zi asks for ai_branch (:1134) and ends in "sl type: unknown"
					continue;
				// the nestings: which of the children (the new vertex among them: it is no flubble) leave for the slubble
				const uint32_t nch = X.vn[xb + f];
				if (!S.filter(f, nch, v, sl.loc == CL_ZI_TRUNK, [&](uint32_t ch) { return cn_leaves(t, X, xb, sl, ai, zi, n_of_f, ch); }))
					return;
			}
		}
		if (pin != NIL && par >= f) { // (never, with the PVST numbered parents first; the order of the reference all the same)
			if (!S.push(f, pin))
				return;
			got = true;
		}
		if (got && lane == 0)
			A.touched[q] = 1; // find_midi looks at it
		S.sync();
	}
}
// ---- find_midi, midi.cpp:225-268 (the branch case is undefined in the reference: nothing comes of it): one wave per
// flubble that got a concealed vertex as a child
__global__ void __launch_bounds__(64) k_sub_midi_find(const uint32_t *__restrict__ n_list, const uint32_t *__restrict__ list, const SpliceArgs A)
{
	const SubT &t = A.t;
	const XArrays &X = A.X;
	const uint32_t lane = threadIdx.x, n = *n_list;
	for (uint32_t it = blockIdx.x; it < n; it += gridDim.x) {
		const uint32_t q = list[it], c = A.comp.of_slot(q), f = q - t.doff[c], xb = A.xoff[c];
		if (X.fam[xb + f] != FAM_FLUBBLE)
			continue;
		uint32_t n_c = 0, n_trunk = 0, trunk[2] = {NIL, NIL};
		const uint32_t zi = X.zi[xb + f], nch = X.vn[xb + f], fb = X.vbeg[xb + f];
		for (uint32_t r0 = 0; r0 < nch; r0 += 64) { // the concealed children, 64 children a round, in order
			const uint32_t r = r0 + lane;
			const uint32_t ch = r < nch ? X.pool[fb + r] : 0u;
			const bool is_c = r < nch && X.fam[xb + ch] == FAM_CONCEALED;
			const bool in_trunk = is_c && X.sl[xb + ch] < zi; // in_trunk, :163-166
			n_c += (uint32_t)__popcll(__ballot(is_c));
			for (unsigned long long m = __ballot(in_trunk); m; m &= m - 1) {
				if (n_trunk < 2)
					trunk[n_trunk] = __shfl(ch, __ffsll((long long)m) - 1);
				n_trunk++;
			}
		}
		if (n_c < 2 || n_trunk != 2)
			continue;
		uint32_t g_idx = NIL, s_idx = NIL;
		for (int k = 0; k < 2; k++) {
			const uint8_t loc = X.loc[xb + trunk[k]];
			if (loc == CL_AI_BRANCH || loc == CL_AI_TRUNK)
				g_idx = trunk[k];
			else
				s_idx = trunk[k];
		}
		if (g_idx == NIL || s_idx == NIL)
			continue;
		// (all midi bubbles are found before the first is added: the pair is remembered in the flubble's spare fields)
		if (lane == 0) {
			X.b_up[xb + f] = g_idx;
			X.b_lo[xb + f] = s_idx;
			A.md[q] = 1;
		}
	}
}
// ---- add_midi, midi.cpp:19-61: the flubbles in ascending idx get the vertices behind the concealed ones
__global__ void __launch_bounds__(64) k_sub_midi_add(const uint32_t *__restrict__ n_list, const uint32_t *__restrict__ list, const SpliceArgs A)
{
	const SubT &t = A.t;
	const XArrays &X = A.X;
	const uint32_t lane = threadIdx.x, n = *n_list;
	for (uint32_t it = blockIdx.x; it < n; it += gridDim.x) {
		const uint32_t q = list[it];
		if (!A.md[q])
			continue;
		const uint32_t c = A.comp.of_slot(q), q0 = t.doff[c], n0 = t.c_npvst[c], f = q - q0, base = t.base_of(c), N = t.c_ntree[c];
		Splice S{X, A.xoff[c], A.ptop + c, A.poff[c + 1], t.err, lane};
		const uint32_t xb = S.xb;
		const uint32_t g_idx = X.b_up[xb + f], s_idx = X.b_lo[xb + f];
		const uint32_t up = min(g_idx, s_idx), lo = max(g_idx, s_idx);
		const uint32_t nch = X.vn[xb + f]; // children before the midi bubble is attached
		const uint32_t v = n0 + (A.cn_off[q0 + n0] - A.cn_off[q0]) + (A.md_ps[q] - A.md_ps[q0]);
		if (lane == 0) {
			const uint32_t x = xb + v;
			X.fam[x] = FAM_MIDI;
			X.vn[x] = X.vcap[x] = 0;
			X.vbeg[x] = 0;
			X.ai[x] = X.zi[x] = X.sl[x] = NIL;
			X.loc[x] = 0;
			// cn_b of a concealed vertex: the second boundary when it was formed with a, the first with z (Concealed::as_str)
			X.id1[x] = X.id2[xb + g_idx], X.or1[x] = X.or2[xb + g_idx];
			X.id2[x] = X.id1[xb + s_idx], X.or2[x] = X.or1[xb + s_idx];
			X.route[x] = 'L';
			X.b_up[x] = up, X.b_lo[x] = lo;
		}
		S.sync();
		// (sic) spanning-tree depths at the PVST indices of the two concealed vertices; past the tree: +infinity
		const uint32_t d_up = up < N ? t.dep(base + up) : 0xFFFFFFFFu, d_lo = lo < N ? t.dep(base + lo) : 0xFFFFFFFFu;
		// the children before the bubble (a copy in the reference) are filtered, then the bubble itself goes behind them
		const bool ok = S.filter(f, nch, v, false, [&](uint32_t ch) {
			return X.fam[xb + ch] == FAM_FLUBBLE && d_up < t.depth[X.ai[xb + ch]] && d_lo > t.depth[X.zi[xb + ch]];
		});
		if (!ok || !S.push(f, v))
			return;
	}
}
// ---- add_smothered, smothered.cpp:349-383: one wave per concealed vertex, its records in search order
__global__ void __launch_bounds__(64) k_sub_smothered(uint32_t NC, const SpliceArgs A)
{
	const SubT &t = A.t;
	const XArrays &X = A.X;
	const uint32_t lane = threadIdx.x;
	for (uint32_t k = blockIdx.x; k < NC; k += gridDim.x) {
		const uint32_t rb = A.smo_off[k], re = A.smo_off[k + 1];
		if (rb == re)
			continue;
		const uint32_t q = A.cn[k].q, c = A.comp.of_slot(q), q0 = t.doff[c], n0 = t.c_npvst[c];
		const uint32_t cn_b = A.cn_off[q0], cn_e = A.cn_off[q0 + n0];
		Splice S{X, A.xoff[c], A.ptop + c, A.poff[c + 1], t.err, lane};
		const uint32_t xb = S.xb;
		const uint32_t cv = n0 + (k - cn_b); // the k-th slubble became the k-th concealed vertex
		const uint32_t v0 = n0 + (cn_e - cn_b) + (A.md_ps[q0 + n0] - A.md_ps[q0]) - A.smo_off[cn_b]; // + r: behind the midi bubbles, in search order
		const bool is_g = A.cn[k].loc == CL_AI_TRUNK || A.cn[k].loc == CL_AI_BRANCH;
		const uint32_t cnb_id = is_g ? X.id2[xb + cv] : X.id1[xb + cv];
		const uint8_t cnb_or = is_g ? X.or2[xb + cv] : X.or1[xb + cv];
		for (uint32_t r = rb; r < re; r++) {
			const Smo m = A.smo[r];
			const uint32_t v = v0 + r;
			if (lane == 0) {
				const uint32_t x = xb + v;
				X.fam[x] = FAM_SMOTHERED;
				X.vn[x] = X.vcap[x] = 0;
				X.vbeg[x] = 0;
				X.ai[x] = X.zi[x] = X.sl[x] = NIL;
				X.loc[x] = 0;
				const uint32_t sm_id = t.gid[m.sm_st];
				const uint8_t sm_or = (t.flags[m.sm_st] & TF_TYPE_MASK) == 0u ? 1 : 0; // comp_e / comp_w: type l -> reverse
				if ((m.flags & 2u) && (m.flags & 1u)) { // Smothered::as_str, pvst.hpp:612-636
					X.id1[x] = cnb_id, X.or1[x] = cnb_or;
					X.id2[x] = sm_id, X.or2[x] = sm_or;
				} else {
					X.id1[x] = sm_id, X.or1[x] = sm_or;
					X.id2[x] = cnb_id, X.or2[x] = cnb_or;
				}
				X.route[x] = (m.flags & 2u) ? 'L' : 'R';
				X.b_up[x] = m.b_up, X.b_lo[x] = m.b_lo;
			}
			S.sync();
			if (!S.push(cv, v))
				return;
			// nest, :332-347: a range-for over children_v[cv] while del_edge erases from it, as libstdc++ runs it: the loop
			// goes to the OLD end, the slots behind the live end hold what erase left there
			const uint32_t old_end = X.vn[xb + cv];
			for (uint32_t i = 0; i < old_end; i++) {
				const uint32_t ch = X.pool[X.vbeg[xb + cv] + i];
				uint32_t c_up, c_lo;
				if (S.fl_like(ch))
					c_up = X.ai[xb + ch], c_lo = X.zi[xb + ch];
				else if (X.fam[xb + ch] == FAM_CONCEALED)
					c_up = X.b_up[xb + ch], c_lo = X.b_lo[xb + ch];
				else
					continue;
				if (c_up == NIL || c_lo == NIL)
					continue;
				if (t.is_desc(m.b_up, c_up) && t.is_desc(c_lo, m.b_lo)) {
					S.erase(cv, ch);
					if (!S.push(v, ch))
						return;
				}
			}
		}
	}
}
__global__ void k_sub_counts(uint32_t C, const SpliceArgs A, uint32_t *__restrict__ counts)
{
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= C)
		return;
	const uint32_t n0 = A.t.c_npvst[c], q0 = A.t.doff[c];
	uint32_t n_cn = 0, n_md = 0, n_sm = 0;
	if (n0) {
		const uint32_t cn_b = A.cn_off[q0], cn_e = A.cn_off[q0 + n0];
		n_cn = cn_e - cn_b;
		n_md = A.md_ps[q0 + n0] - A.md_ps[q0];
		n_sm = A.smo_off[cn_e] - A.smo_off[cn_b];
	}
	counts[3 * c] = n_cn, counts[3 * c + 1] = n_md, counts[3 * c + 2] = n_sm;
}
// sizes of the final children lists (one lane per X slot), then the lists themselves into one compact array
__global__ void k_sub_child_counts(uint32_t NX, const uint32_t *__restrict__ xoff, const uint32_t *__restrict__ counts,
				   const uint32_t *__restrict__ c_npvst, const CompAt comp_x, const uint32_t *__restrict__ vn,
				   uint32_t *__restrict__ out)
{
	const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x > NX)
		return;
	uint32_t n = 0;
	if (x < NX) {
		const uint32_t c = comp_x.of_slot(x); // (doff = xoff here)
		const uint32_t nt = c_npvst[c] ? c_npvst[c] + counts[3 * c] + counts[3 * c + 1] + counts[3 * c + 2] : 0;
		if (x - xoff[c] < nt)
			n = vn[x];
	}
	out[x] = n;
}
// one lane per child entry (a chromosome's root has a million children: one lane per LIST took 0.1 s): the list that owns
// output slot i is the last vertex whose offset is <= i
__global__ void k_sub_child_gather(uint32_t NCH, uint32_t NX, const uint32_t *__restrict__ coff, const uint32_t *__restrict__ vbeg,
				   const uint32_t *__restrict__ pool, uint32_t *__restrict__ out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= NCH)
		return;
	uint32_t lo = 0, hi = NX; // coff[lo] <= i < coff[hi]
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (coff[mid] <= i)
			lo = mid;
		else
			hi = mid;
	}
	out[i] = pool[vbeg[lo] + (i - coff[lo])];
}
// the vertices of every component without the spare slots of X-space: dense slot d = dvoff[c] + (index inside the component)
__global__ void k_sub_compact(uint32_t NV, uint32_t C, const uint32_t *__restrict__ dvoff, const uint32_t *__restrict__ xoff, const XArrays X,
			      const uint32_t *__restrict__ coff, uint8_t *__restrict__ fam, uint8_t *__restrict__ or1, uint8_t *__restrict__ or2,
			      uint8_t *__restrict__ route, uint32_t *__restrict__ id1, uint32_t *__restrict__ id2, uint32_t *__restrict__ coff_d)
{
	const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
	if (d >= NV)
		return;
	uint32_t lo = 0, hi = C; // last c with dvoff[c] <= d (components without a PVST own no slot)
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (dvoff[mid] <= d)
			lo = mid;
		else
			hi = mid;
	}
	const uint32_t x = xoff[lo] + (d - dvoff[lo]);
	fam[d] = X.fam[x], or1[d] = X.or1[x], or2[d] = X.or2[x], route[d] = X.route[x];
	id1[d] = X.id1[x], id2[d] = X.id2[x];
	coff_d[d] = coff[x];
}
} // namespace

SubForest::~SubForest()
{
	if (blk && pool)
		pool->put(blk, blk_cap, blk_seg);
}

void run_subflubbles(const CompState &cs, const SeqWs &sw, const ParWs &pw, const TreeWs &tw, const LeafState &ls, uint32_t C,
		     HostScratch &host, SubForest &out, const std::shared_ptr<::PinnedPool> &pool, hipStream_t s, Arena *arena,
		     size_t *arena_hint)
{
	const uint32_t V = sw.V, T = 2 * V + C;
	const uint32_t NB0 = pw.nb0, NB = pw.nb0 + pw.ncap + pw.nsimp;
	const uint32_t Q = (uint32_t)pw.d_total; // dense PVST slots
	const LeafIn &in = ls.in;
	// POVU_HIP_SUB_TIMES=1: wall-clock of the phases on stderr (each mark synchronises the stream)
	const bool times = getenv("POVU_HIP_SUB_TIMES") != nullptr;
	auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double t_prev = now();
	auto mark = [&](const char *what) {
		if (!times)
			return;
		HIP_CHECK(hipStreamSynchronize(s));
		const double t_now = now();
		fprintf(stderr, "[subflubbles] %-22s %9.1f ms\n", what, t_now - t_prev);
		t_prev = t_now;
	};
	std::deque<DevBuf> bufs; // (what came from hipMalloc is freed when the stage returns or throws)
	size_t need = 0;
	if (arena)
		arena->reserve(arena_hint ? *arena_hint : 0);
	auto dev32 = [&](size_t n) { return bufs.emplace_back().get<uint32_t>(n, arena, need); };
	auto dev8 = [&](size_t n) { return bufs.emplace_back().get<uint8_t>(n, arena, need); };
	const size_t tmp_bytes = scan_tmp_bytes(std::max<size_t>(std::max<size_t>(T, NB), Q) + 8);
	void *tmp = bufs.emplace_back().get<char>(tmp_bytes, arena, need);
	uint32_t *err = dev32(4);
	HIP_CHECK(hipMemsetAsync(err, 0, 16, s));

	// ---- tables
	uint32_t *depth = dev32((size_t)T + 16), *hi0 = dev32((size_t)T + 16), *eat = dev32((size_t)T + 16);
	LAUNCH(k_sub_depth_hi0, (size_t)T + 1, s, T, sw.t_size, sw.t_depth, depth, hi0, eat);
	uint32_t *ocnt = dev32((size_t)T + 2), *icnt = dev32((size_t)T + 2), *o_off = dev32((size_t)T + 2), *i_off = dev32((size_t)T + 2);
	HIP_CHECK(hipMemsetAsync(ocnt, 0, ((size_t)T + 2) * 4, s));
	HIP_CHECK(hipMemsetAsync(icnt, 0, ((size_t)T + 2) * 4, s));
	LAUNCH(k_sub_edge_counts, NB, s, NB, NB0, pw.b_src, pw.b_tgt, pw.b_ord, ocnt, icnt, hi0, eat);
	scan_exclusive_u32(ocnt, o_off, (size_t)T + 1, tmp, tmp_bytes, s);
	scan_exclusive_u32(icnt, i_off, (size_t)T + 1, tmp, tmp_bytes, s);
	uint32_t *o_adj = dev32((size_t)NB + 1), *i_adj = dev32((size_t)NB + 1);
	HIP_CHECK(hipMemsetAsync(ocnt, 0, ((size_t)T + 2) * 4, s));
	HIP_CHECK(hipMemsetAsync(icnt, 0, ((size_t)T + 2) * 4, s));
	LAUNCH(k_sub_edge_fill, NB, s, NB, pw.b_src, pw.b_tgt, o_off, i_off, ocnt, icnt, o_adj, i_adj);
	// (the rows are in whatever order the atomics came: every question asked of them is about a set -- any, count, deepest --
	// and sorting them cost 0.6 s on a segment with 2*10^5 links, one lane on its row)
	// literal hi
	SegTree segH, segD;
	segH.tree = dev32(SegTree::tree_words((size_t)T + 1) + 16);
	segD.tree = dev32(SegTree::tree_words((size_t)T + 1) + 16);
	seg_build(segH, hi0, (size_t)T + 1, s);
	seg_build(segD, depth, (size_t)T + 1, s);
	uint32_t *simp_ps = dev32((size_t)T + 4), *hi = dev32((size_t)T + 4);
	scan_exclusive_u8(in.simp, simp_ps, (size_t)T + 1, nullptr, nullptr, 0, tmp, tmp_bytes, s);
	const CompAt comp{cs.voff, pw.doff, C};
	LAUNCH(k_sub_hi, T, s, T, sw.t_size, segH, simp_ps, comp, hi);
	mark("edge tables, hi");
	// creation keys of the ordinary edges
	unsigned long long *ekey = bufs.emplace_back().get<unsigned long long>((size_t)NB0 + 2, arena, need);
	uint32_t *wbefore = dev32((size_t)T + 8), *wtail = dev32((size_t)T + 8);
	HIP_CHECK(hipMemsetAsync(wbefore, 0, ((size_t)T + 8) * 4, s));
	HIP_CHECK(hipMemsetAsync(wtail, 0, ((size_t)T + 8) * 4, s));
	debug_edge_id_weights(cs, sw, tw, wbefore, wtail, s);
	LAUNCH(k_sub_edge_keys, T, s, T, sw.t_size, in.out_ord, eat, wbefore, ekey);
	// LoA
	uint32_t *lo = dev32((size_t)T + 4), *heap = dev32((size_t)NB0 + 4), *has_self = dev32((size_t)C + 4);
	{
		const uint32_t P2 = SegTree::pow2((size_t)T + 1);
		uint32_t *mx = dev32(2 * (size_t)P2 + 4);
		HIP_CHECK(hipMemsetAsync(mx, 0, (2 * (size_t)P2 + 4) * 4, s));
		HIP_CHECK(hipMemsetAsync(has_self, 0, ((size_t)C + 4) * 4, s));
		LAUNCH(k_sub_lo_mark, NB0, s, NB0, pw.b_src, pw.b_tgt, P2, mx);
		LAUNCH(k_sub_lo_query, T, s, T, sw.t_size, P2, mx, in.nself, comp, lo, has_self);
		if (getenv("POVU_HIP_SUB_LITERAL_LOA")) // (tests: the reference's heap on every component, whatever its edges)
			fill_u32(has_self, C, 1u, s);
		if (getenv("POVU_HIP_SUB_FORBID_HEAP")) { // (tests: fail when any component would take the one-lane heap)
			std::vector<uint32_t> hs(C);
			HIP_CHECK(hipMemcpyAsync(hs.data(), has_self, (size_t)C * 4, hipMemcpyDeviceToHost, s));
			HIP_CHECK(hipStreamSynchronize(s));
			for (uint32_t c = 0; c < C; c++)
				if (hs[c])
					throw HipError("subflubble passes: component " + std::to_string(c + 1) + " takes the literal LoA heap (POVU_HIP_SUB_FORBID_HEAP is set)");
		}
		LAUNCH(k_sub_lo, C, s, C, cs.voff, sw.c_ntree, sw.t_size, depth, in.out_ord, eat, pw.b_tgt, in.O, has_self, heap, lo);
	}
	mark("creation keys, lo");
	// bracket rows: counts in closed form, the edges in source order with a segment tree over their targets
	uint32_t *br_cnt = dev32((size_t)T + 4);
	LAUNCH(k_sub_br_counts, (size_t)T + 1, s, T, sw.t_size, in.gp, in.P, in.out_ord, in.nself, br_cnt);
	uint32_t *tgt_s = dev32((size_t)NB0 + 32), *slot_s = dev32((size_t)NB0 + 32);
	LAUNCH(k_sub_edges_by_source, T, s, T, sw.t_size, in.out_ord, eat, in.O, pw.b_tgt, tgt_s, slot_s);
	SegTree segT;
	segT.tree = dev32(SegTree::tree_words((size_t)NB0 + 1) + 16);
	seg_build(segT, tgt_s, NB0, s);
	mark("bracket rows");

	SubT t{};
	t.T = T, t.C = C, t.NB0 = NB0, t.NB = NB;
	t.voff = cs.voff, t.c_ntree = sw.c_ntree, t.doff = pw.doff, t.c_npvst = sw.c_npvst;
	t.size = sw.t_size, t.gp = in.gp, t.nchild = in.nchild, t.depth = depth, t.gid = sw.t_gid, t.flags = sw.t_flags;
	t.b_src = pw.b_src, t.b_tgt = pw.b_tgt, t.b_ord = pw.b_ord;
	t.o_off = o_off, t.o_adj = o_adj, t.i_off = i_off, t.i_adj = i_adj;
	t.br_cnt = br_cnt, t.O = in.O, t.slot_s = slot_s, t.segT = segT, t.lo = lo, t.hi = hi, t.ekey = ekey;
	segD.val = depth;
	t.segD = segD;
	t.p_ai = ls.dense.ai, t.p_zi = ls.dense.zi, t.p_fam = ls.dense.fam;
	t.p_parent = pw.d_parent, t.p_a = pw.d_a, t.p_z = pw.d_z, t.p_aor = pw.d_aor, t.p_zor = pw.d_zor;
	t.err = err;

	// ---- find_concealed: count, scan, emit
	uint32_t *cn_cnt = dev32((size_t)Q + 4), *cn_off = dev32((size_t)Q + 4), *mn = dev32(2 * (size_t)Q + 4);
	LAUNCH(k_sub_cn_count, (size_t)Q + 1, s, Q, t, comp, cn_cnt, mn);
	scan_exclusive_u32(cn_cnt, cn_off, (size_t)Q + 1, tmp, tmp_bytes, s);
	const uint32_t NC = host.read_u32(cn_off + Q, s);
	Slub *cn = bufs.emplace_back().get<Slub>((size_t)NC + 1, arena, need);
	LAUNCH(k_sub_cn_emit, Q, s, Q, t, comp, cn_off, cn);
	mark("find_concealed search");
	// ---- find_smothered: count, scan, emit
	uint32_t *sm_cnt = dev32((size_t)NC + 4), *sm_off = dev32((size_t)NC + 4);
	LAUNCH(k_sub_smo_count, (size_t)NC + 1, s, NC, t, comp, cn, sm_cnt);
	scan_exclusive_u32(sm_cnt, sm_off, (size_t)NC + 1, tmp, tmp_bytes, s);
	const uint32_t NS = host.read_u32(sm_off + NC, s);
	Smo *smo = bufs.emplace_back().get<Smo>((size_t)NS + 1, arena, need);
	LAUNCH(k_sub_smo_emit, NC, s, NC, t, comp, cn, sm_off, smo);
	mark("find_smothered search");

	// ---- layout of the splice: per component its stretch of X-space and of the vector pool.  The host needs three numbers a
	// component (PVST vertices, concealed and smothered records): worked out on the device and read through page-locked
	// scratch (until round 5 the whole offset array of the records came back, 4 bytes per PVST vertex into a pageable vector:
	// 5 - 20 ms on the whole-genome workload, depending on what the process's allocator made of a fresh 100 MB)
	uint32_t *comp_counts = dev32(3 * (size_t)C + 4);
	uint32_t *h_cc = host.take<uint32_t>(3 * (size_t)C + 4);
	if (C) {
		LAUNCH(k_sub_comp_counts, C, s, C, pw.doff, sw.c_npvst, cn_off, sm_off, comp_counts);
		HIP_CHECK(copy_async(h_cc, comp_counts, 3 * (size_t)C * 4, hipMemcpyDeviceToHost, s));
	}
	HIP_CHECK(hipStreamSynchronize(s));
	std::vector<uint32_t> h_np((size_t)C + 1);
	std::vector<uint32_t> h_xoff((size_t)C + 1), h_poff((size_t)C + 1);
	uint64_t xs = 0, ps = 0;
	for (uint32_t c = 0; c < C; c++) {
		h_xoff[c] = (uint32_t)xs;
		h_poff[c] = (uint32_t)ps;
		const uint64_t n0 = h_np[c] = h_cc[3 * (size_t)c];
		if (n0) {
			const uint64_t ncn = h_cc[3 * (size_t)c + 1], nsm = h_cc[3 * (size_t)c + 2];
			xs += 2 * n0 + ncn + nsm;
			ps += 32 * n0 + 16 * (ncn + nsm) + 256;
		}
		if (xs > 0xFFFFFFF0ull || ps > 0xFFFFFFF0ull)
			throw HipError("subflubble passes: more than 2^32 PVST vertices or child slots in one pass");
	}
	h_xoff[C] = (uint32_t)xs;
	h_poff[C] = (uint32_t)ps;
	const uint32_t NX = (uint32_t)xs;
	uint32_t *xoff = dev32((size_t)C + 2), *poff = dev32((size_t)C + 2);
	HIP_CHECK(hipMemcpyAsync(xoff, h_xoff.data(), ((size_t)C + 1) * 4, hipMemcpyHostToDevice, s));
	HIP_CHECK(hipMemcpyAsync(poff, h_poff.data(), ((size_t)C + 1) * 4, hipMemcpyHostToDevice, s));
	XArrays X{};
	X.fam = dev8((size_t)NX + 4), X.or1 = dev8((size_t)NX + 4), X.or2 = dev8((size_t)NX + 4), X.route = dev8((size_t)NX + 4);
	X.loc = dev8((size_t)NX + 4);
	X.id1 = dev32((size_t)NX + 4), X.id2 = dev32((size_t)NX + 4), X.ai = dev32((size_t)NX + 4), X.zi = dev32((size_t)NX + 4);
	X.sl = dev32((size_t)NX + 4), X.b_up = dev32((size_t)NX + 4), X.b_lo = dev32((size_t)NX + 4);
	X.vbeg = dev32((size_t)NX + 4), X.vn = dev32((size_t)NX + 4), X.vcap = dev32((size_t)NX + 4);
	X.pool = dev32((size_t)ps + 4);
	HIP_CHECK(hipMemsetAsync(X.vn, 0, ((size_t)NX + 4) * 4, s));
	HIP_CHECK(hipMemsetAsync(X.vcap, 0, ((size_t)NX + 4) * 4, s));
	const CompAt comp_q{cs.voff, pw.doff, C};
	uint32_t *cap_ps = dev32((size_t)NX + 4);
	const size_t tmpx_bytes = scan_tmp_bytes((size_t)NX + 8);
	void *tmpx = bufs.emplace_back().get<char>(tmpx_bytes, arena, need);
	LAUNCH(k_sub_x_init, Q, s, Q, t, comp_q, xoff, X);
	{
		uint32_t *k0 = dev32((size_t)Q + 4), *k1 = dev32((size_t)Q + 4), *v0 = dev32((size_t)Q + 4), *v1 = dev32((size_t)Q + 4);
		const size_t sort_bytes = sort_tmp_bytes((size_t)Q + 8);
		void *sort_tmp = bufs.emplace_back().get<char>(sort_bytes, arena, need);
		LAUNCH(k_sub_x_keys, Q, s, Q, NX, t, comp_q, xoff, k0, v0);
		sort_pairs_u32(k0, k1, v0, v1, Q, bits_for(NX), sort_tmp, sort_bytes, s);
		LAUNCH(k_sub_x_vbeg, Q, s, Q, t, comp_q, xoff, poff, k1, cap_ps, X);
		LAUNCH(k_sub_x_place, Q, s, Q, NX, k1, v1, cap_ps, X);
	}
	mark("layout, PVST vectors");
	uint32_t *counts = dev32(3 * (size_t)C + 4);
	if (C && Q) {
		SpliceArgs A{};
		A.t = t, A.comp = comp_q, A.xoff = xoff, A.poff = poff, A.ptop = dev32((size_t)C + 4);
		A.cn_off = cn_off, A.cn = cn, A.mn = mn, A.smo_off = sm_off, A.smo = smo, A.X = X;
		A.pin = dev32((size_t)Q + 4);
		A.act = dev8((size_t)Q + 16), A.touched = dev8((size_t)Q + 16), A.md = dev8((size_t)Q + 16);
		uint32_t *md_ps = dev32((size_t)Q + 8), *list = dev32((size_t)Q + 8), *n_list = dev32(8);
		A.md_ps = md_ps;
		const size_t ctmp_bytes = compact_tmp_bytes((size_t)Q + 8);
		void *ctmp = bufs.emplace_back().get<char>(ctmp_bytes, arena, need);
		LAUNCH(k_sub_ptop, C, s, C, poff, sw.c_npvst, A.ptop);
		LAUNCH(k_sub_pin, Q, s, Q, A);
		compact_flagged_u8(A.act, Q, list, n_list, ctmp, ctmp_bytes, s);
		KLAUNCH(k_sub_splice_cn, dim3(SPLICE_WAVES), dim3(64), 0, s, n_list, list, A);
		compact_flagged_u8(A.touched, Q, list, n_list, ctmp, ctmp_bytes, s);
		KLAUNCH(k_sub_midi_find, dim3(SPLICE_WAVES), dim3(64), 0, s, n_list, list, A);
		HIP_CHECK(hipMemsetAsync(A.md + Q, 0, 1, s));
		scan_exclusive_u8(A.md, md_ps, (size_t)Q + 1, nullptr, nullptr, 0, tmp, tmp_bytes, s);
		KLAUNCH(k_sub_midi_add, dim3(SPLICE_WAVES), dim3(64), 0, s, n_list, list, A);
		if (NS)
			KLAUNCH(k_sub_smothered, dim3(std::min<unsigned>(NC, SPLICE_WAVES)), dim3(64), 0, s, NC, A);
		LAUNCH(k_sub_counts, C, s, C, A, counts);
	} else if (C) {
		HIP_CHECK(hipMemsetAsync(counts, 0, 3 * (size_t)C * 4, s));
	}
	mark("splice");
	const uint32_t e = host.read_u32(err, s);
	if (e & E_POOL)
		throw HipError("subflubble passes: the children vectors outgrew their pool (internal sizing bug)");
	if (e & E_LAYOUT)
		throw HipError("subflubble passes: more inserted vertices than the layout has room for (internal sizing bug)");
	// ---- the children lists, compact
	uint32_t *ccnt = dev32((size_t)NX + 4), *coff = dev32((size_t)NX + 4);
	const CompAt comp_x{cs.voff, xoff, C};
	LAUNCH(k_sub_child_counts, (size_t)NX + 1, s, NX, xoff, counts, sw.c_npvst, comp_x, X.vn, ccnt);
	scan_exclusive_u32(ccnt, coff, (size_t)NX + 1, tmpx, tmpx_bytes, s);
	const uint32_t NCH = host.read_u32(coff + NX, s);
	uint32_t *child = dev32((size_t)NCH + 4);
	LAUNCH(k_sub_child_gather, NCH, s, NCH, NX, coff, X.vbeg, X.pool, child);
	mark("children lists");

	// ---- to the host: the vertices compacted on the device (X-space has spare slots behind every component), then one copy
	// per array straight into the forest's vectors
	std::vector<uint32_t> h_counts(3 * (size_t)C), h_dvoff((size_t)C + 1);
	if (C)
		HIP_CHECK(copy_async(h_counts.data(), counts, 3 * (size_t)C * 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	out.voff.assign((size_t)C + 1, 0);
	uint64_t n_vtx = 0;
	for (uint32_t c = 0; c < C; c++) {
		out.voff[c] = n_vtx;
		h_dvoff[c] = (uint32_t)n_vtx;
		if (h_np[c])
			n_vtx += (uint64_t)h_np[c] + h_counts[3 * c] + h_counts[3 * c + 1] + h_counts[3 * c + 2];
	}
	out.voff[C] = n_vtx;
	h_dvoff[C] = (uint32_t)n_vtx;
	out.counts = h_counts;
	const uint32_t NV = (uint32_t)n_vtx;
	uint32_t *dvoff = dev32((size_t)C + 2);
	HIP_CHECK(hipMemcpyAsync(dvoff, h_dvoff.data(), ((size_t)C + 1) * 4, hipMemcpyHostToDevice, s));
	uint8_t *d_fam = dev8((size_t)NV + 4), *d_or1 = dev8((size_t)NV + 4), *d_or2 = dev8((size_t)NV + 4), *d_route = dev8((size_t)NV + 4);
	uint32_t *d_id1 = dev32((size_t)NV + 4), *d_id2 = dev32((size_t)NV + 4), *d_coff = dev32((size_t)NV + 4);
	LAUNCH(k_sub_compact, NV, s, NV, C, dvoff, xoff, X, coff, d_fam, d_or1, d_or2, d_route, d_id1, d_id2, d_coff);
	// one page-locked block: fam | or1 | or2 | route | id1 | id2 | coff (+ 1) | child
	auto pad = [](size_t b) { return (b + 63) & ~size_t(63); };
	const size_t b8 = pad((size_t)NV + 1), b32 = pad(((size_t)NV + 1) * 4), bch = pad(((size_t)NCH + 1) * 4);
	const size_t total_bytes = 4 * b8 + 3 * b32 + bch + 64;
	if (!pool)
		throw HipError("subflubble passes: no pool of page-locked memory (internal)");
	out.pool = pool;
	out.blk = pool->get(total_bytes, out.blk_cap, &out.blk_seg);
	char *hb = static_cast<char *>(out.blk);
	uint8_t *o_fam = reinterpret_cast<uint8_t *>(hb), *o_or1 = o_fam + b8, *o_or2 = o_or1 + b8, *o_route = o_or2 + b8;
	uint32_t *o_id1 = reinterpret_cast<uint32_t *>(hb + 4 * b8), *o_id2 = reinterpret_cast<uint32_t *>(hb + 4 * b8 + b32);
	uint32_t *o_coff = reinterpret_cast<uint32_t *>(hb + 4 * b8 + 2 * b32), *o_child = reinterpret_cast<uint32_t *>(hb + 4 * b8 + 3 * b32);
	auto d2h = [&](void *dst, const void *src, size_t bytes) {
		if (bytes)
			HIP_CHECK(copy_async(dst, src, bytes, hipMemcpyDeviceToHost, s));
	};
	d2h(o_fam, d_fam, NV);
	d2h(o_or1, d_or1, NV);
	d2h(o_or2, d_or2, NV);
	d2h(o_route, d_route, NV);
	d2h(o_id1, d_id1, (size_t)NV * 4);
	d2h(o_id2, d_id2, (size_t)NV * 4);
	d2h(o_coff, d_coff, (size_t)NV * 4);
	d2h(o_child, child, (size_t)NCH * 4); // (the lists are compact already, in vertex order)
	HIP_CHECK(hipStreamSynchronize(s));
	o_coff[NV] = NCH;
	out.n_vtx = n_vtx, out.n_child = NCH;
	out.fam = o_fam, out.or1 = o_or1, out.or2 = o_or2, out.route = o_route;
	out.id1 = o_id1, out.id2 = o_id2, out.coff = o_coff, out.child = o_child;
	if (arena_hint)
		*arena_hint = need; // (the next call on this context reserves that much up front)
	mark("to the host");
}

} // namespace povu_hip
