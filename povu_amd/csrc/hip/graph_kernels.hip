// graph_kernels.hip -- rows A and B of the scope table on the GPU:
//   per-side CSR of the bidirected graph           (bd::VG::add_edge, bidirected.cpp:317-335)
//   weakly connected components                    (bd::VG::componetize, bidirected.cpp:477-602)
//   component re-indexing: local vertex idx, first-encounter local edge idx,
//   local per-side adjacency                       (bidirected.cpp:552-569, Edge::get_other_vtx :66-77)
// All kernels are integer/index work bounded by HBM traffic; accesses are
// coalesced over edge / side / slot ids, the scattered part goes through L2.
#include "graph_kernels.hpp"

#include <algorithm>

namespace povu_hip
{

static constexpr int TPB = 256;
static inline unsigned nblk(size_t n) { return (unsigned)((n + TPB - 1) / TPB); }

// ---------------------------------------------------------------- row A: CSR
// (also validates the operands: the first link that names an unknown vertex or side lands in *bad)
__global__ void k_side_degree(uint32_t E, uint32_t V, const uint32_t *__restrict__ v1, const uint8_t *__restrict__ s1,
			      const uint32_t *__restrict__ v2, const uint8_t *__restrict__ s2,
			      uint32_t *__restrict__ deg, uint32_t *__restrict__ bad)
{
	uint32_t e = BIDX * blockDim.x + threadIdx.x;
	if (e >= E)
		return;
	if (v1[e] >= V || v2[e] >= V || s1[e] > 1 || s2[e] > 1) {
		atomicMin(bad, e); // (the host stops the upload before anything reads the degrees)
		return;
	}
	uint32_t a = 2 * v1[e] + s1[e], b = 2 * v2[e] + s2[e];
	atomicAdd(&deg[a], 1u);
	if (b != a) // a same-side self loop sits once in the side's std::set (bidirected.cpp:324-333)
		atomicAdd(&deg[b], 1u);
}
// The adjacency lists without a sort (every side has at most FILL_MAX_SIDE_DEGREE links): a link takes the next free
// slot of either of its sides, in whatever order the lanes arrive, and k_side_sort then puts every list in ascending link
// order -- the order a stable sort of the (side, link) pairs gives, at a third of its cost.  The links are valid here.
static constexpr uint32_t FILL_MAX_SIDE_DEGREE = 64;
__global__ void k_slot_fill(uint32_t E, const uint32_t *__restrict__ v1, const uint8_t *__restrict__ s1,
			    const uint32_t *__restrict__ v2, const uint8_t *__restrict__ s2, const uint32_t *__restrict__ off,
			    uint32_t *__restrict__ cursor, uint32_t *__restrict__ adj)
{
	uint32_t e = BIDX * blockDim.x + threadIdx.x;
	if (e >= E)
		return;
	const uint32_t a = 2 * v1[e] + s1[e], b = 2 * v2[e] + s2[e];
	adj[off[a] + atomicAdd(&cursor[a], 1u)] = e;
	if (b != a)
		adj[off[b] + atomicAdd(&cursor[b], 1u)] = e;
}
__global__ void k_side_sort(uint32_t nS, const uint32_t *__restrict__ off, uint32_t *__restrict__ adj)
{
	uint32_t S = BIDX * blockDim.x + threadIdx.x;
	if (S >= nS)
		return;
	const uint32_t lo = off[S], n = off[S + 1] - lo;
	if (n < 2)
		return;
	if (n == 2) {
		const uint32_t x = adj[lo], y = adj[lo + 1];
		if (y < x)
			adj[lo] = y, adj[lo + 1] = x;
		return;
	}
	for (uint32_t i = 1; i < n; i++) { // (n <= FILL_MAX_SIDE_DEGREE)
		const uint32_t x = adj[lo + i];
		uint32_t j = i;
		while (j > 0 && adj[lo + j - 1] > x) {
			adj[lo + j] = adj[lo + j - 1];
			j--;
		}
		adj[lo + j] = x;
	}
}
// hub graphs: the (side, link) pairs of the stable radix sort
__global__ void k_side_pairs(uint32_t E, const uint32_t *__restrict__ v1, const uint8_t *__restrict__ s1,
			     const uint32_t *__restrict__ v2, const uint8_t *__restrict__ s2, uint32_t *__restrict__ keys,
			     uint32_t *__restrict__ vals, uint32_t sentinel)
{
	uint32_t e = BIDX * blockDim.x + threadIdx.x;
	if (e >= E)
		return;
	const uint32_t a = 2 * v1[e] + s1[e], b = 2 * v2[e] + s2[e];
	keys[2 * e] = a;
	keys[2 * e + 1] = b != a ? b : sentinel; // (the sentinel sorts behind every side: the slot count leaves it out)
	vals[2 * e] = vals[2 * e + 1] = e;
}

// other end of every adjacency slot, as a global side id
__global__ void k_slot_other(uint32_t nS, const uint32_t *__restrict__ off, const uint32_t *__restrict__ adj,
			     const uint32_t *__restrict__ v1, const uint8_t *__restrict__ s1, const uint32_t *__restrict__ v2,
			     const uint8_t *__restrict__ s2, uint32_t *__restrict__ aoth)
{
	uint32_t S = BIDX * blockDim.x + threadIdx.x;
	if (S >= nS)
		return;
	for (uint32_t k = off[S]; k < off[S + 1]; k++) {
		uint32_t e = adj[k];
		uint32_t a = 2 * v1[e] + s1[e], b = 2 * v2[e] + s2[e];
		aoth[k] = a == S ? b : a; // same-side self loop: a == b == S
	}
}
// twin slot of every adjacency slot: where the same link sits in the list of its other end (lists are ascending by
// link idx, so a binary search finds it)
__global__ void k_slot_twin(uint32_t E, const uint32_t *__restrict__ off, const uint32_t *__restrict__ adj,
			    const uint32_t *__restrict__ v1, const uint8_t *__restrict__ s1, const uint32_t *__restrict__ v2,
			    const uint8_t *__restrict__ s2, uint32_t *__restrict__ atwin)
{
	uint32_t e = BIDX * blockDim.x + threadIdx.x;
	if (e >= E)
		return;
	const uint32_t a = 2 * v1[e] + s1[e], b = 2 * v2[e] + s2[e];
	auto find = [&](uint32_t side) {
		uint32_t lo = off[side], hi = off[side + 1];
		while (lo < hi) {
			uint32_t mid = (lo + hi) >> 1;
			if (adj[mid] < e)
				lo = mid + 1;
			else
				hi = mid;
		}
		return lo;
	};
	const uint32_t ka = find(a), kb = a == b ? ka : find(b);
	atwin[ka] = kb;
	atwin[kb] = ka;
}
__global__ void k_vertex_degree(uint32_t V, const uint32_t *__restrict__ off, uint32_t *__restrict__ deg)
{
	uint32_t v = BIDX * blockDim.x + threadIdx.x;
	if (v < V)
		deg[v] = off[2 * v + 2] - off[2 * v];
}

// tips as the loader infers them, src/mto/from_gfa.cpp:262-277
__global__ void k_infer_tips(uint32_t V, const uint32_t *__restrict__ off, uint8_t *__restrict__ tip)
{
	uint32_t v = BIDX * blockDim.x + threadIdx.x;
	if (v >= V)
		return;
	bool le = off[2 * v + 1] == off[2 * v], re = off[2 * v + 2] == off[2 * v + 1];
	tip[v] = le ? 1 : (re ? 2 : 0);
}
__global__ void k_check_tips(uint32_t V, const uint8_t *__restrict__ tip, uint32_t *__restrict__ bad)
{
	uint32_t v = BIDX * blockDim.x + threadIdx.x;
	if (v < V && tip[v] > 2)
		atomicMin(bad, v);
}

// ---------------------------------------------------------------- row B: WCC
// Lock-free union-find over the links (k_uf_tiles in LDS, k_uf_cross in global memory): roots are
// hooked larger-under-smaller with a CAS, finds use path halving.  The surviving root of a component is
// its smallest vertex idx, which is exactly the key componetize orders
// components by (next-unvisited linear rescan, bidirected.cpp:585-596).
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x)
{
	uint32_t p = parent[x];
	while (p != x) {
		uint32_t gp = parent[p];
		if (gp != p)
			parent[x] = gp; // halving; racy but monotone
		x = p;
		p = gp;
	}
	return x;
}

// ---- tile-local union-find straight from the CSR.  Pangenome GFAs are (mostly) sorted along the genome,
// so almost every link joins two segments of nearby idx.  A workgroup owns UF_TILE consecutive vertices, keeps
// their parent pointers in LDS and walks the adjacency slots of its sides (contiguous in aoth): a link is
// handled from its smaller endpoint; when both ends fall inside the tile it is united there (LDS CAS, no
// global atomics, no long global pointer chains), otherwise the slot goes onto the cross list -- wave ballot +
// prefix popcount, one atomic per wave -- and k_uf_cross unites it in global memory afterwards.
#ifndef POVU_UF_TILE
#define POVU_UF_TILE 8192
#endif
#ifndef POVU_UF_TPB
#define POVU_UF_TPB 1024
#endif
static constexpr uint32_t UF_TILE = POVU_UF_TILE, UF_TPB = POVU_UF_TPB, UF_HEAVY = 128, UF_HEAVY_CAP = 256;
// vertices with more links than this take the radix-sorted adjacency path of the re-index (an insertion sort per side is
// quadratic in the side's links)
static constexpr uint32_t SORT_FREE_MAX_VDEG = 48;

__device__ __forceinline__ uint32_t lds_find(uint32_t *par, uint32_t x)
{
	uint32_t p = par[x];
	while (p != x) {
		uint32_t gp = par[p];
		if (gp != p)
			par[x] = gp;
		x = p;
		p = gp;
	}
	return x;
}
// Inside a tile the roots are linked by a pseudo-random PRIORITY, not by their index: a pangenome tile is one long chain of
// segments, all its links (i, i + 1) are united at the same moment by neighbouring lanes, and linking larger-under-smaller
// then builds a parent chain as long as the tile -- every later find walked it (k_uf_tiles was bound by those walks:
// 2.5 ms for 1.8 GB of input on the whole-genome graph).  With random priorities the same unions leave trees of
// logarithmic depth.  A root is only ever hooked under a root of HIGHER priority (a bijection of the index: no ties), so
// the pointers stay acyclic whatever the races; the smallest vertex of every set -- what the labels must name -- is
// worked out at the end of the kernel.
__device__ __forceinline__ uint32_t uf_prio(uint32_t x) { return x * 0x9E3779B1u; }
__device__ __forceinline__ bool lds_union(uint32_t *par, uint32_t a, uint32_t b)
{
	uint32_t ra = lds_find(par, a), rb = lds_find(par, b);
	while (ra != rb) {
		const bool a_low = uf_prio(ra) < uf_prio(rb);
		const uint32_t child = a_low ? ra : rb, up = a_low ? rb : ra;
		const uint32_t old = atomicCAS(&par[child], child, up);
		if (old == child)
			return true;
		ra = lds_find(par, old);
		rb = lds_find(par, up);
	}
	return false;
}
// appends (v, slot) of every lane with `take` to the cross list: one atomic per wave
__device__ __forceinline__ void wave_append(bool take, uint32_t v, uint32_t k, uint32_t *__restrict__ xcount,
					    uint2 *__restrict__ xlist)
{
	const unsigned long long m = __ballot(take);
	if (!m)
		return;
	const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)m) - 1u;
	uint32_t base = 0;
	if (lane == leader)
		base = atomicAdd(xcount, (uint32_t)__popcll(m));
	base = __shfl(base, (int)leader);
	if (take)
		xlist[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = make_uint2(v, k);
}

// *any_loop is raised when some link joins a segment to itself: without self loops the local adjacency of the re-index
// has exactly the global slots (a self loop is stored as (ve, complement(ve)) and owns a slot on either side,
// bidirected.cpp:529-531), so its offsets are the CSR's own.
__global__ void __launch_bounds__(UF_TPB) k_uf_tiles(uint32_t V, const uint32_t *__restrict__ off, const uint32_t *__restrict__ aoth,
						  uint32_t *__restrict__ label,
						  uint8_t *__restrict__ hook, uint32_t *__restrict__ xcount, uint2 *__restrict__ xlist,
						  uint32_t *__restrict__ any_loop)
{
	__shared__ uint32_t par[UF_TILE];
	__shared__ uint32_t minv2[UF_TILE / 2]; // smallest member of the set a root stands for, 16 bits each (filled at the end)
	__shared__ uint32_t heavy[UF_HEAVY_CAP];
	__shared__ uint32_t n_heavy;
	const uint32_t v0 = BIDX * UF_TILE, v1 = min(V, v0 + UF_TILE);
	for (uint32_t i = threadIdx.x; i < UF_TILE; i += blockDim.x)
		par[i] = i;
	if (threadIdx.x == 0)
		n_heavy = 0;
	__syncthreads();
	bool loop_seen = false;
	// hook[k] = 1: the link of slot k merged two trees of the union-find (it is a link of the spanning forest).  The flag sits
	// at the SLOT the link was united from -- the one of its smaller end --, not at the link's id: no adj[k] is loaded here,
	// and the re-index reads the flag of a slot and of its twin (until round 5: one byte per link, found through adj[k]).
	auto handle = [&](uint32_t v, uint32_t k, uint32_t o) -> bool { // o = aoth[k]; true: the slot leaves the tile upwards
		const uint32_t vo = o >> 1;
		if (vo <= v) {
			loop_seen = loop_seen || vo == v;
			return false; // handled from the other end (or a self loop: never a forest link)
		}
		if (vo >= v1)
			return true;
		if (lds_union(par, v - v0, vo - v0))
			hook[k] = 1;
		return false;
	};
	// A lane takes FOUR CONSECUTIVE sides a round, UF_GROUPS groups of them at a time: the five offsets of a group in two load
	// instructions, and the group's slots -- one contiguous stretch of aoth, five or six words on a pangenome graph -- four
	// a load; the loads of all groups are in flight together.  (The walk a side at a time this replaces issued two 4-byte
	// loads per side and one per slot, each a dependent round trip: 80 memory instructions a lane where this form has ~20,
	// and the kernel was bound by them, not by the 1.8 GB it reads on the whole-genome graph.)
	constexpr uint32_t UF_GROUPS = 2;
	const uint32_t S0 = 2 * v0, S1 = 2 * v1; // (S0 is a multiple of 4: tiles hold an even number of vertices)
	static_assert((2 * UF_TILE) % (4 * UF_GROUPS * UF_TPB) == 0, "a tile is a whole number of rounds");
	for (uint32_t base = S0; base < S1; base += 4 * UF_GROUPS * blockDim.x) { // uniform trip count: the ballots below need whole waves
		uint32_t b[UF_GROUPS][5], skip[UF_GROUPS]; // offsets of the group's sides; sides left to the workgroup (hubs)
#pragma unroll
		for (uint32_t gq = 0; gq < UF_GROUPS; gq++) {
			const uint32_t Sg = base + (gq * blockDim.x + threadIdx.x) * 4u;
			skip[gq] = 0;
			if (Sg + 4 <= S1) {
				const uint4 o4 = *reinterpret_cast<const uint4 *>(off + Sg);
				b[gq][0] = o4.x, b[gq][1] = o4.y, b[gq][2] = o4.z, b[gq][3] = o4.w, b[gq][4] = off[Sg + 4];
			} else { // the tile's (= the graph's) last sides
#pragma unroll
				for (uint32_t j = 0; j < 5; j++)
					b[gq][j] = off[min(Sg + j, S1)];
				if (Sg >= S1)
					b[gq][0] = b[gq][1] = b[gq][2] = b[gq][3] = b[gq][4] = 0;
			}
#pragma unroll
			for (uint32_t j = 0; j < 4; j++)
				if (b[gq][j + 1] - b[gq][j] > UF_HEAVY) { // a hub side: the whole workgroup walks it below
					const uint32_t q = atomicAdd(&n_heavy, 1u);
					if (q < UF_HEAVY_CAP) {
						heavy[q] = Sg + j;
						skip[gq] |= 1u << j;
					}
				}
		}
		for (uint32_t r = 0;; r += 4) {
			uint4 o[UF_GROUPS];
			bool any = false;
#pragma unroll
			for (uint32_t gq = 0; gq < UF_GROUPS; gq++) {
				const bool live = b[gq][0] + r < b[gq][4];
				o[gq] = live ? load4_unaligned(aoth + b[gq][0] + r) : make_uint4(0u, 0u, 0u, 0u);
				any = any || live;
			}
			if (!__any(any))
				break;
#pragma unroll
			for (uint32_t gq = 0; gq < UF_GROUPS; gq++) {
				const uint32_t Sg = base + (gq * blockDim.x + threadIdx.x) * 4u;
				const uint32_t os[4] = {o[gq].x, o[gq].y, o[gq].z, o[gq].w};
#pragma unroll
				for (uint32_t q = 0; q < 4; q++) {
					const uint32_t k = b[gq][0] + r + q;
					const uint32_t j = (k >= b[gq][1] ? 1u : 0u) + (k >= b[gq][2] ? 1u : 0u) + (k >= b[gq][3] ? 1u : 0u); // the side the slot belongs to
					const bool live = k < b[gq][4] && !((skip[gq] >> j) & 1u);
					const uint32_t v = (Sg + j) >> 1;
					const bool cross = live && handle(v, k, os[q]);
					wave_append(cross, v, k, xcount, xlist);
				}
			}
		}
	}
	__syncthreads();
	const uint32_t nh = min(n_heavy, UF_HEAVY_CAP);
	for (uint32_t q = 0; q < nh; q++) {
		const uint32_t S = heavy[q], lo = off[S], hi = off[S + 1];
		for (uint32_t kb = lo; kb < hi; kb += blockDim.x) {
			const uint32_t k = kb + threadIdx.x;
			const bool cross = k < hi && handle(S >> 1, k, aoth[k]);
			wave_append(cross, S >> 1, k, xcount, xlist);
		}
	}
	if (__any(loop_seen) && (threadIdx.x & 63) == 0)
		*any_loop = 1u; // (same value from every writer)
	__syncthreads();
	// flatten by pointer doubling (a long chain of segments leaves a parent chain as long: walking it once per
	// vertex would be quadratic); a round without a change ends it
	for (int round = 0; round < 14; round++) {
		uint32_t pp[UF_TILE / UF_TPB];
		int changed = 0;
#pragma unroll
		for (uint32_t k = 0; k < UF_TILE / UF_TPB; k++) {
			const uint32_t i = threadIdx.x + k * UF_TPB, p = par[i];
			pp[k] = par[p];
			changed |= pp[k] != p;
		}
		if (!__syncthreads_or(changed))
			break;
#pragma unroll
		for (uint32_t k = 0; k < UF_TILE / UF_TPB; k++)
			par[threadIdx.x + k * UF_TPB] = pp[k];
		__syncthreads();
	}
	// the label of a vertex = the smallest vertex of its set (componetize orders the components by it, and the links that
	// leave the tile are united larger-under-smaller in global memory on top of these labels)
	static_assert(UF_TILE <= 65536, "tile-local indices are kept in 16 bits here");
	for (uint32_t i = threadIdx.x; i < UF_TILE / 2; i += blockDim.x)
		minv2[i] = (2 * i) | ((2 * i + 1) << 16);
	__syncthreads();
	auto min_of = [&](uint32_t r) { return (minv2[r >> 1] >> (16 * (r & 1u))) & 0xFFFFu; };
	for (uint32_t i = threadIdx.x; i < UF_TILE; i += blockDim.x) {
		const uint32_t r = par[i];
		if (r == i)
			continue;
		const uint32_t sh = 16 * (r & 1u);
		for (uint32_t old = minv2[r >> 1];;) { // 16-bit atomic minimum (most vertices are not the smallest of their set: read first)
			if (((old >> sh) & 0xFFFFu) <= i)
				break;
			const uint32_t seen = atomicCAS(&minv2[r >> 1], old, (old & ~(0xFFFFu << sh)) | (i << sh));
			if (seen == old)
				break;
			old = seen;
		}
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < v1 - v0; i += blockDim.x)
		label[v0 + i] = v0 + min_of(par[i]);
}

// the links that leave their tile, in global memory (grid-stride: their number only exists on the device)
__global__ void k_uf_cross(const uint32_t *__restrict__ xcount, const uint2 *__restrict__ xlist,
			   const uint32_t *__restrict__ aoth, uint32_t *parent,
			   uint8_t *__restrict__ hook)
{
	const uint32_t NX = *xcount;
	for (uint32_t i = BIDX * blockDim.x + threadIdx.x; i < NX; i += gridDim.x * blockDim.x) {
		const uint2 x = xlist[i];
		uint32_t ra = uf_find(parent, x.x), rb = uf_find(parent, aoth[x.y] >> 1);
		while (ra != rb) {
			uint32_t hi = ra > rb ? ra : rb, lo = ra > rb ? rb : ra;
			uint32_t old = atomicCAS(&parent[hi], hi, lo);
			if (old == hi) {
				hook[x.y] = 1;
				break;
			}
			ra = uf_find(parent, old);
			rb = uf_find(parent, lo);
		}
	}
}

__global__ void k_uf_flatten(uint32_t V, uint32_t *parent, uint8_t *__restrict__ is_root, uint32_t *__restrict__ unsorted)
{
	// four vertices a lane: their words in one 16-byte load and store (a kernel of a few loads per element is bound by the
	// memory instructions it issues), the four walks to the roots side by side
	const uint32_t v0 = (BIDX * blockDim.x + threadIdx.x) * 4u;
	if (v0 == 0)
		*unsorted = 0; // (k_labels_sorted, the next launch, raises it)
	if (v0 >= V)
		return;
	auto root_of = [&](uint32_t r) {
		for (;;) {
			const uint32_t p = parent[r];
			if (p == r)
				return r;
			r = p;
		}
	};
	if (v0 + 4 <= V) {
		const uint4 p4 = *reinterpret_cast<const uint4 *>(parent + v0);
		// (a vertex that is its own parent is a root; the others start their walk at the parent just loaded)
		const uint32_t r0 = p4.x == v0 ? v0 : root_of(p4.x), r1 = p4.y == v0 + 1 ? v0 + 1 : root_of(p4.y);
		const uint32_t r2 = p4.z == v0 + 2 ? v0 + 2 : root_of(p4.z), r3 = p4.w == v0 + 3 ? v0 + 3 : root_of(p4.w);
		*reinterpret_cast<uint4 *>(parent + v0) = make_uint4(r0, r1, r2, r3); // readers of parent[] in this kernel only ever walk towards roots
		*reinterpret_cast<uint32_t *>(is_root + v0) =
			(r0 == v0 ? 1u : 0u) | (r1 == v0 + 1 ? 1u << 8 : 0u) | (r2 == v0 + 2 ? 1u << 16 : 0u) | (r3 == v0 + 3 ? 1u << 24 : 0u);
		return;
	}
	for (uint32_t v = v0; v < V; v++) {
		const uint32_t r = root_of(v);
		parent[v] = r;
		is_root[v] = (r == v) ? 1 : 0;
	}
}

// Are the vertices already grouped by component, in component order?  Components are ranked by their smallest vertex
// (= their union-find root), so that is the case iff the roots never decrease along the vertex order -- true for GFAs
// written chromosome by chromosome, and then the stable sort of the vertices by component is the identity.
__global__ void k_labels_sorted(uint32_t V, const uint32_t *__restrict__ label, uint32_t *__restrict__ unsorted)
{
	uint32_t v = BIDX * blockDim.x + threadIdx.x;
	if (v > 0 && v < V && label[v] < label[v - 1])
		*unsorted = 1; // (same value from every writer)
}

__global__ void k_comp_of(uint32_t V, const uint32_t *__restrict__ label, const uint32_t *__restrict__ crank,
			  uint32_t *__restrict__ comp_of, uint32_t *__restrict__ iota, uint32_t C,
			  unsigned long long *__restrict__ start_key)
{
	const uint32_t v0 = (BIDX * blockDim.x + threadIdx.x) * 4u; // (four vertices a lane, 16-byte loads and stores)
	if (v0 >= V)
		return;
	if (v0 + 4 <= V) {
		const uint4 l = *reinterpret_cast<const uint4 *>(label + v0);
		*reinterpret_cast<uint4 *>(comp_of + v0) = make_uint4(crank[l.x], crank[l.y], crank[l.z], crank[l.w]);
		if (iota)
			*reinterpret_cast<uint4 *>(iota + v0) = make_uint4(v0, v0 + 1, v0 + 2, v0 + 3);
	} else {
		for (uint32_t v = v0; v < V; v++) {
			comp_of[v] = crank[label[v]];
			if (iota)
				iota[v] = v;
		}
	}
	for (uint32_t v = v0; v < v0 + 4 && v <= C; v++) // "no tip yet" for k_sorted_vertices' atomicMin; C <= V: v = C is
		start_key[v] = ~0ull;			      // reached by the lane that holds it, or written by the last lane
	if (v0 + 4 >= V)
		start_key[C] = ~0ull;
}

// after the stable sort by component: sorted position i holds global vertex perm[i]
// perm == nullptr: the vertices already are in (component, idx) order.  Sorted space then IS the global vertex space:
// no position, degree, id or tip array is written (the callers read the resident graph's own), only the component
// boundaries and the DFS starts.
__global__ void k_sorted_vertices(uint32_t V, uint32_t C, const uint32_t *__restrict__ ckey,
				  const uint32_t *__restrict__ perm, const uint32_t *__restrict__ off,
				  const uint32_t *__restrict__ vid, const uint8_t *__restrict__ tip,
				  uint32_t *__restrict__ pos, uint32_t *__restrict__ voff, uint32_t *__restrict__ vdeg,
				  uint32_t *__restrict__ gid_s, uint8_t *__restrict__ tip_s,
				  unsigned long long *__restrict__ start_key, uint32_t *__restrict__ stats)
{
	const uint32_t i0 = (BIDX * blockDim.x + threadIdx.x) * 4u;
	if (i0 >= V)
		return;
	// sorted space = global space: four vertices that sit inside one component and hold no tip -- almost every lane's --
	// have nothing to record (two 16-byte loads and a word of tip flags say so)
	if (!perm && i0 > 0 && i0 + 4 < V) {
		const uint4 c4 = *reinterpret_cast<const uint4 *>(ckey + i0);
		if (ckey[i0 - 1] == c4.x && c4.x == c4.w && *reinterpret_cast<const uint32_t *>(tip + i0) == 0u)
			return;
	}
	for (uint32_t i = i0; i < V && i < i0 + 4; i++) {
		if (i < 4)
			stats[i] = 0; // stats[0] = most links on one side (k_mark_first2 / k_max_u32)
		const uint32_t v = perm ? perm[i] : i, c = ckey[i];
		if (i == 0 || ckey[i - 1] != c)
			voff[c] = i;
		if (i == V - 1)
			voff[C] = V;
		const uint8_t t = tip[v];
		if (perm) {
			pos[v] = i;
			vdeg[i] = off[2 * v + 2] - off[2 * v];
			gid_s[i] = vid[v];
			tip_s[i] = t;
		}
		// start of the spanning tree = *tips().begin(): smallest (id, then l<r), types.cpp:60-68
		if (t)
			atomicMin(&start_key[c], ((unsigned long long)vid[v] << 32) | (unsigned long long)(2u * i + (t == 1 ? 0u : 1u)));
	}
}

// slot order of componetize's edge loop: vertices ascending, e_l then e_r ascending
// (bidirected.cpp:558-569).  P = sbase[i] + rank of the slot inside vertex i.
__global__ void k_first_slot(uint32_t V, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ off,
			     const uint32_t *__restrict__ adj, const uint32_t *__restrict__ sbase,
			     uint32_t *__restrict__ first)
{
	uint32_t S = BIDX * blockDim.x + threadIdx.x; // sorted side id
	if (S >= 2 * V)
		return;
	uint32_t i = S >> 1, s = S & 1, v = perm[i];
	uint32_t lo = off[2 * v + s], hi = off[2 * v + s + 1], P = sbase[i] + (lo - off[2 * v]);
	for (uint32_t k = lo; k < hi; k++, P++)
		atomicMin(&first[adj[k]], P);
}

__global__ void k_mark_first(uint32_t V, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ off,
			     const uint32_t *__restrict__ adj, const uint32_t *__restrict__ sbase,
			     const uint32_t *__restrict__ first, uint32_t *__restrict__ flag)
{
	uint32_t S = BIDX * blockDim.x + threadIdx.x;
	if (S >= 2 * V)
		return;
	uint32_t i = S >> 1, s = S & 1, v = perm[i];
	uint32_t lo = off[2 * v + s], hi = off[2 * v + s + 1], P = sbase[i] + (lo - off[2 * v]);
	for (uint32_t k = lo; k < hi; k++, P++)
		flag[P] = (first[adj[k]] == P) ? 1u : 0u;
}

// local edge le = rank of the first-encounter slot; stored from the encountering side, a self loop
// always as (ve, complement(ve)) whatever its original sides (Edge::get_other_vtx(v_idx, ve), :66-77)
__global__ void k_local_edges(uint32_t V, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ pos,
			      const uint32_t *__restrict__ off, const uint32_t *__restrict__ adj,
			      const uint32_t *__restrict__ sbase, const uint32_t *__restrict__ first,
			      const uint32_t *__restrict__ erank, const uint32_t *__restrict__ v1,
			      const uint8_t *__restrict__ s1, const uint32_t *__restrict__ v2,
			      const uint8_t *__restrict__ s2, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
			      uint32_t *__restrict__ ldeg, const uint8_t *__restrict__ hook, const uint32_t *__restrict__ atwin,
			      uint32_t *__restrict__ la, uint32_t *__restrict__ lb, uint8_t *__restrict__ tgray)
{
	uint32_t S = BIDX * blockDim.x + threadIdx.x;
	if (S >= 2 * V)
		return;
	uint32_t i = S >> 1, s = S & 1, v = perm[i];
	uint32_t lo = off[2 * v + s], hi = off[2 * v + s + 1], P = sbase[i] + (lo - off[2 * v]);
	for (uint32_t k = lo; k < hi; k++, P++) {
		uint32_t e = adj[k];
		if (first[e] != P)
			continue;
		uint32_t le = erank[P];
		uint32_t a = v1[e], b = v2[e], So;
		if (a == b)
			So = 2 * i + (1 - s);
		else if (a == v)
			So = 2 * pos[b] + s2[e];
		else
			So = 2 * pos[a] + s1[e];
		// two slots per local edge, generated in local-edge order so that the stable sort by
		// side leaves every side's list ascending by local edge idx (std::set order)
		keys[2 * le] = S;
		vals[2 * le] = 2 * le; // slot origin: even = the la end, odd = the lb end
		keys[2 * le + 1] = So;
		vals[2 * le + 1] = 2 * le + 1;
		la[le] = S;
		lb[le] = So;
		tgray[le] = hook[k] | hook[atwin[k]]; // (the flag sits at the slot of the link's smaller end, k_uf_tiles)
		atomicAdd(&ldeg[S], 1u);
		atomicAdd(&ldeg[So], 1u);
	}
}

static constexpr uint32_t NIL32 = 0xFFFFFFFFu;

// ---- sort-free variant (vertices with few links): a side's local adjacency holds the links of its global
// slots, plus the self loops of the opposite side -- a self loop is stored as (ve, complement(ve)) from the
// side that met it first, so it owns one slot on either side of its vertex (bidirected.cpp:529-531).
// Which of a link's two slots componetize meets first needs no search either: slots are visited by sorted
// vertex, l side before r side, so it is the slot of the smaller sorted vertex (for a loop: the l side, or
// its only slot).  P = sbase[i] + rank of the slot inside vertex i, as in k_first_slot.
__device__ __forceinline__ bool slot_is_first(uint32_t i, uint32_t s, uint32_t v, uint32_t o, const uint32_t *__restrict__ pos)
{
	if ((o >> 1) == v) // self loop: same side (one slot) or l-r (the l slot comes first)
		return (o & 1u) == s || s == 0;
	return i < (pos ? pos[o >> 1] : (o >> 1)); // pos == nullptr: one component, sorted order = global order
}
// the local degree of every side (no atomics) + the largest of them
__global__ void k_local_degree(uint32_t V, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ pos,
			       const uint32_t *__restrict__ off, const uint32_t *__restrict__ aoth, uint8_t *__restrict__ ldeg,
			       uint32_t *__restrict__ stats)
{
	uint32_t S = BIDX * blockDim.x + threadIdx.x;
	uint32_t cnt = 0;
	if (S < 2 * V) {
		uint32_t i = S >> 1, s = S & 1, v = perm ? perm[i] : i;
		uint32_t lo = off[2 * v + s], hi = off[2 * v + s + 1];
		for (uint32_t k = lo; k < hi; k++) {
			const uint32_t o = aoth[k];
			cnt += ((o >> 1) == v) ? (slot_is_first(i, s, v, o, pos) ? 1u : 0u) : 1u;
		}
		lo = off[2 * v + (1 - s)], hi = off[2 * v + (1 - s) + 1];
		for (uint32_t k = lo; k < hi; k++) {
			const uint32_t o = aoth[k];
			if ((o >> 1) == v && slot_is_first(i, 1 - s, v, o, pos))
				cnt++;
		}
		ldeg[S] = (uint8_t)cnt; // (this path only runs when no vertex has more than SORT_FREE_MAX_VDEG links)
		if (S == 2 * V - 1)
			ldeg[2 * V] = 0; // closes the array the scan turns into loff
	}
	__shared__ uint32_t sh[TPB / 64];
	uint32_t m = cnt;
	for (int o = 32; o; o >>= 1)
		m = max(m, __shfl_down(m, o));
	if ((threadIdx.x & 63) == 0)
		sh[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < TPB / 64; w++)
			m = max(m, sh[w]);
		if (m > __hip_atomic_load(stats, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) // few blocks ever need the atomic
			atomicMax(stats, m);
	}
}

// Every side gathers its (link id, other side) pairs and keeps them ascending by link id with an insertion sort in place
// (std::set order of the per-side edge lists).  componetize numbers a component's links in first-encounter order
// (bidirected.cpp:558-569); all the traversal needs of that number is (a) the ORDER it gives the links of one side and (b) a
// name that both ends of a link agree on.  The position of the link's first-encounter slot in the sorted slot order
// has both properties and needs no counting: it is this slot's own position for the end that meets the link first, the
// twin slot's (atwin, built at upload) for the other end.  lle[slot] = that id | LLE_TREE when the link is in the spanning
// forest of the segments (so that the tree kernels read the flag with the slot, not through the id).
__global__ void k_local_adj(uint32_t V, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ pos,
			    const uint32_t *__restrict__ off,
			    const uint32_t *__restrict__ aoth, const uint32_t *__restrict__ atwin,
			    const uint32_t *__restrict__ sbase, const uint32_t *__restrict__ loff,
			    const uint8_t *__restrict__ hook, uint32_t *ladj, uint32_t *lle, bool any_loop)
{
	uint32_t S = BIDX * blockDim.x + threadIdx.x;
	if (S >= 2 * V)
		return;
	uint32_t i = S >> 1, s = S & 1, v = perm ? perm[i] : i;
	uint32_t b0 = off[2 * v], lo = off[2 * v + s], hi = off[2 * v + s + 1];
	const uint32_t sb = sbase ? sbase[i] : b0; // (no sbase: sorted space = global space)
	uint32_t P = sb + (lo - b0);
	const uint32_t base = loff[S];
	uint32_t n = 0;
	// the first four entries of a side are kept sorted in registers (almost every side has fewer) and written once;
	// a fifth entry flushes them and the insertion sort continues in place in global memory
	uint32_t l0 = NIL32, l1 = NIL32, l2 = NIL32, l3 = NIL32, o0 = 0, o1 = 0, o2 = 0, o3 = 0;
	auto id_of = [](uint32_t w) { return w == NIL32 ? NIL32 : (w & LLE_ID); };
	auto insert = [&](uint32_t le, uint32_t other) { // le: id | flag
		if (n < 4) {
			l3 = le, o3 = other; // slot 3 is free while n < 4 (free slots hold NIL32 = +inf)
			uint32_t t;
			if (id_of(l3) < id_of(l2)) {
				t = l2, l2 = l3, l3 = t;
				t = o2, o2 = o3, o3 = t;
			}
			if (id_of(l2) < id_of(l1)) {
				t = l1, l1 = l2, l2 = t;
				t = o1, o1 = o2, o2 = t;
			}
			if (id_of(l1) < id_of(l0)) {
				t = l0, l0 = l1, l1 = t;
				t = o0, o0 = o1, o1 = t;
			}
			n++;
			return;
		}
		if (n == 4) {
			lle[base] = l0, lle[base + 1] = l1, lle[base + 2] = l2, lle[base + 3] = l3;
			ladj[base] = o0, ladj[base + 1] = o1, ladj[base + 2] = o2, ladj[base + 3] = o3;
		}
		uint32_t j = n++;
		while (j > 0 && (lle[base + j - 1] & LLE_ID) > (le & LLE_ID)) {
			lle[base + j] = lle[base + j - 1];
			ladj[base + j] = ladj[base + j - 1];
			j--;
		}
		lle[base + j] = le;
		ladj[base + j] = other;
	};
	// four slots a round: far sides and twin slots in one 16-byte load each; a link's forest flag sits at ONE of its two slots
	// (the one the union-find united it from): the four own flags in four byte loads of one line, the twins' in independent
	// gathers (the global arrays carry slack behind their last slot)
	for (uint32_t k0 = lo; k0 < hi; k0 += 4) {
		const uint4 o4 = load4_unaligned(aoth + k0), t4 = load4_unaligned(atwin + k0);
		const uint32_t rem = hi - k0;
		const uint32_t os[4] = {o4.x, o4.y, o4.z, o4.w}, ts[4] = {t4.x, t4.y, t4.z, t4.w};
		const uint32_t hs[4] = {(uint32_t)(hook[k0] | hook[t4.x]), rem > 1 ? (uint32_t)(hook[k0 + 1] | hook[t4.y]) : 0u,
					rem > 2 ? (uint32_t)(hook[k0 + 2] | hook[t4.z]) : 0u, rem > 3 ? (uint32_t)(hook[k0 + 3] | hook[t4.w]) : 0u};
#pragma unroll
		for (uint32_t q = 0; q < 4; q++) {
			if (q >= rem)
				break;
			const uint32_t o = os[q], vo = o >> 1, Pq = P + q;
			const bool loop = vo == v;
			const uint32_t tree = hs[q] ? LLE_TREE : 0u;
			if (loop) {
				if (!slot_is_first(i, s, v, o, pos))
					continue; // the l-r loop's r slot: the l side owns the edge, this side gets it below
				insert(Pq | tree, S ^ 1u);
				continue;
			}
			const uint32_t io = pos ? pos[vo] : vo, other = 2 * io + (o & 1u);
			if (i < io) // first encounter: the link is named after this slot
				insert(Pq | tree, other);
			else
				insert((sbase ? sbase[io] + (ts[q] - off[2 * vo]) : ts[q]) | tree, other);
		}
		P += 4;
	}
	if (any_loop) { // (a graph without self loops: nothing of the other side's list belongs here)
		lo = off[2 * v + (1 - s)], hi = off[2 * v + (1 - s) + 1], P = sb + (lo - b0);
		for (uint32_t k = lo; k < hi; k++, P++) {
			const uint32_t o = aoth[k];
			if ((o >> 1) == v && slot_is_first(i, 1 - s, v, o, pos))
				insert(P, S ^ 1u); // (a self loop is never a link of the forest)
		}
	}
	if (n <= 4) { // never flushed: write the registers
		if (n > 0)
			lle[base] = l0, ladj[base] = o0;
		if (n > 1)
			lle[base + 1] = l1, ladj[base + 1] = o1;
		if (n > 2)
			lle[base + 2] = l2, ladj[base + 2] = o2;
		if (n > 3)
			lle[base + 3] = l3, ladj[base + 3] = o3;
	}
}

// after the stable sort by side: other side and local edge of every adjacency slot
__global__ void k_local_slots(uint32_t n, const uint32_t *__restrict__ origin, const uint32_t *__restrict__ la,
			      const uint32_t *__restrict__ lb, const uint8_t *__restrict__ tgray, uint32_t *__restrict__ ladj,
			      uint32_t *__restrict__ lle)
{
	uint32_t k = BIDX * blockDim.x + threadIdx.x;
	if (k >= n)
		return;
	uint32_t o = origin[k], le = o >> 1;
	ladj[k] = (o & 1) ? la[le] : lb[le];
	lle[k] = le | (tgray[le] ? LLE_TREE : 0u); // (here the id is the dense first-encounter rank: same order, same agreement)
}

// maximum of v[0..n); with `zeros` also the number of entries that are 0 (sides without links)
__global__ void k_max_u32(uint32_t n, const uint32_t *__restrict__ v, uint32_t *out, uint32_t *zeros)
{
	__shared__ uint32_t sh[4], shz[4];
	uint32_t m = 0, z = 0;
	for (uint32_t i = BIDX * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t x = v[i];
		m = max(m, x);
		z += x == 0 ? 1u : 0u;
	}
	for (int off = 32; off; off >>= 1) {
		m = max(m, __shfl_down(m, off));
		z += __shfl_down(z, off);
	}
	if ((threadIdx.x & 63) == 0) {
		sh[threadIdx.x >> 6] = m;
		shz[threadIdx.x >> 6] = z;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		m = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
		if (m > *(volatile uint32_t *)out) // few blocks ever need the atomic
			atomicMax(out, m);
		z = shz[0] + shz[1] + shz[2] + shz[3];
		if (zeros && z)
			atomicAdd(zeros, z);
	}
}

// first local edge of every component; also publishes voff / eoff / stats straight into the context's
// page-locked host buffer [voff C+1 | eoff C+1 | stats 4] when one is given
// (every local link owns one slot at either end and the components are contiguous in sorted space: the links in front of
// component c are half the local slots in front of its first side)
__global__ void k_comp_edge_offsets(uint32_t C, const uint32_t *__restrict__ voff, const uint32_t *__restrict__ loff,
				    uint32_t *__restrict__ eoff, const uint32_t *__restrict__ stats,
				    uint32_t *__restrict__ host_pub)
{
	uint32_t c = BIDX * blockDim.x + threadIdx.x;
	if (c > C)
		return;
	const uint32_t vo = voff[c], eo = loff[2 * vo] / 2;
	eoff[c] = eo;
	if (host_pub) {
		host_pub[c] = vo;
		host_pub[(size_t)C + 1 + c] = eo;
		if (c < 4)
			host_pub[2 * ((size_t)C + 1) + c] = stats[c];
	}
}

__global__ void k_fill_u32(size_t n, uint32_t *p, uint32_t val)
{
	size_t i = (size_t)BIDX * blockDim.x + threadIdx.x;
	if (i < n)
		p[i] = val;
}

__global__ void k_mark_odd(size_t n, uint32_t *p)
{
	size_t i = (size_t)BIDX * blockDim.x + threadIdx.x;
	if (i < n && (i & 1))
		p[i] = 1u;
}
void mark_odd_u32(uint32_t *p, size_t n, hipStream_t s)
{
	if (n)
		KLAUNCH(k_mark_odd, dim3(nblk(n)), dim3(TPB), 0, s, n, p);
}

void fill_u32(uint32_t *p, size_t n, uint32_t val, hipStream_t s)
{
	if (n)
		KLAUNCH(k_fill_u32, dim3(nblk(n)), dim3(TPB), 0, s, n, p, val);
}

// ------------------------------------------------------------------ host side
void build_global_csr(ResidentGraph &g, Arena &tmp_arena, hipStream_t s)
{
	const uint32_t V = g.V, E = g.E;
	const size_t nS = 2 * (size_t)V;
	tmp_arena.reserve(Arena::padded(2 * (size_t)E + 2, 4) * 4 + Arena::padded(nS + 2, 4) + sort_tmp_bytes(2 * (size_t)E) +
			  scan_tmp_bytes(std::max<size_t>(nS, E) + 2) + (1 << 16));
	uint32_t *keys = tmp_arena.take<uint32_t>(2 * (size_t)E + 1), *vals = tmp_arena.take<uint32_t>(2 * (size_t)E + 1);
	uint32_t *keys2 = tmp_arena.take<uint32_t>(2 * (size_t)E + 1), *vals2 = tmp_arena.take<uint32_t>(2 * (size_t)E + 1);
	uint32_t *deg = tmp_arena.take<uint32_t>(nS + 1);
	size_t sb = sort_tmp_bytes(2 * (size_t)E), cb = scan_tmp_bytes(std::max<size_t>(nS, E) + 2);
	void *stmp = tmp_arena.take<char>(sb), *ctmp = tmp_arena.take<char>(cb);
	uint32_t *word = tmp_arena.take<uint32_t>(8); // [0] max vertex degree, [1] first bad link, [2] first bad tip, [3] max side degree, [4] sides without links
	hipEvent_t ev[3];
	for (auto &e : ev)
		HIP_CHECK(hipEventCreate(&e));
	struct EvGuard {
		hipEvent_t *e;
		~EvGuard()
		{
			for (int i = 0; i < 3; i++)
				(void)hipEventDestroy(e[i]);
		}
	} guard{ev};
	HIP_CHECK(hipEventRecord(ev[0], s));
	HIP_CHECK(hipMemsetAsync(deg, 0, (nS + 1) * 4, s));
	HIP_CHECK(hipMemsetAsync(word, 0, 4, s));
	HIP_CHECK(hipMemsetAsync(word + 1, 0xFF, 8, s));
	HIP_CHECK(hipMemsetAsync(word + 3, 0, 8, s));
	if (E) {
		KLAUNCH(k_side_degree, dim3(nblk(E)), dim3(TPB), 0, s, E, V, g.v1, g.s1, g.v2, g.s2, deg, word + 1);
	}
	if (g.tips_given) {
		KLAUNCH(k_check_tips, dim3(nblk(V)), dim3(TPB), 0, s, V, g.tip, word + 2);
	}
	scan_exclusive_u32(deg, g.off, nS + 1, ctmp, cb, s);
	if (nS) {
		KLAUNCH(k_max_u32, dim3(std::min<unsigned>(nblk(nS), 1024)), dim3(TPB), 0, s, (uint32_t)nS, deg, word + 3, word + 4);
	}
	uint32_t hw[5] = {0, 0, 0, 0, 0}; // slots, first bad link, first bad tip, most links on one side, sides without links
	HIP_CHECK(copy_async(&hw[0], g.off + nS, 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(copy_async(&hw[1], word + 1, 16, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	g.n_empty_sides = hw[4];
	if (hw[1] != POVU_NIL)
		throw HipError("link " + std::to_string(hw[1]) + " references an unknown vertex or side");
	if (hw[2] != POVU_NIL)
		throw HipError("bad tip mark");
	g.n_slots = hw[0];
	if (g.n_slots && hw[3] <= FILL_MAX_SIDE_DEGREE) {
		HIP_CHECK(hipMemsetAsync(deg, 0, (nS + 1) * 4, s)); // (the degrees live on in g.off: the array now counts filled slots)
		KLAUNCH(k_slot_fill, dim3(nblk(E)), dim3(TPB), 0, s, E, g.v1, g.s1, g.v2, g.s2, g.off, deg, g.adj);
		KLAUNCH(k_side_sort, dim3(nblk(nS)), dim3(TPB), 0, s, (uint32_t)nS, g.off, g.adj);
	} else if (g.n_slots) {
		KLAUNCH(k_side_pairs, dim3(nblk(E)), dim3(TPB), 0, s, E, g.v1, g.s1, g.v2, g.s2, keys, vals, (uint32_t)nS);
		sort_pairs_u32(keys, keys2, vals, vals2, 2 * (size_t)E, bits_for(nS), stmp, sb, s);
		HIP_CHECK(copy_async(g.adj, vals2, (size_t)g.n_slots * 4, hipMemcpyDeviceToDevice, s));
	}
	if (!g.tips_given && V) {
		KLAUNCH(k_infer_tips, dim3(nblk(V)), dim3(TPB), 0, s, V, g.off, g.tip);
	}
	g.max_vdeg = 0;
	if (V && E) {
		KLAUNCH(k_slot_other, dim3(nblk(nS)), dim3(TPB), 0, s, (uint32_t)nS, g.off, g.adj, g.v1, g.s1, g.v2, g.s2,
				   g.aoth);
		KLAUNCH(k_vertex_degree, dim3(nblk(V)), dim3(TPB), 0, s, V, g.off, deg);
		KLAUNCH(k_max_u32, dim3(std::min<unsigned>(nblk(V), 1024)), dim3(TPB), 0, s, V, deg, word, (uint32_t *)nullptr);
		HIP_CHECK(copy_async(&g.max_vdeg, word, 4, hipMemcpyDeviceToHost, s));
	}
	HIP_CHECK(hipEventRecord(ev[1], s));
	if (V && E) { // reverse-slot table: where the same link sits in the list of its other end
		KLAUNCH(k_slot_twin, dim3(nblk(E)), dim3(TPB), 0, s, E, g.off, g.adj, g.v1, g.s1, g.v2, g.s2, g.atwin);
	}
	HIP_CHECK(hipEventRecord(ev[2], s));
	HIP_CHECK(hipStreamSynchronize(s));
	HIP_CHECK(hipEventElapsedTime(&g.csr_ms, ev[0], ev[1]));
	HIP_CHECK(hipEventElapsedTime(&g.twin_ms, ev[1], ev[2]));
}

bool sort_free_adjacency(const ResidentGraph &g, bool force_sorted_adjacency) { return g.max_vdeg <= SORT_FREE_MAX_VDEG && !force_sorted_adjacency; }

uint32_t *label_components_enqueue(const ResidentGraph &g, CompState &st, StageTimer &tm, hipStream_t s)
{
	const uint32_t V = g.V, E = g.E;
	tm.begin("wcc_label");
	// one byte per adjacency slot; the length of the cross list sits behind them (zeroed by the same memset)
	const size_t xoff = (2 * (size_t)E + 7) & ~size_t(7);
	HIP_CHECK(hipMemsetAsync(st.hook, 0, xoff + 24, s));
	uint32_t *xcount = reinterpret_cast<uint32_t *>(st.hook + xoff);
	uint32_t *any_loop = xcount + 1; // cleared with the flags
	uint2 *xlist = reinterpret_cast<uint2 *>(st.keys); // [E] pairs fit the 2E+2 words; free until the re-index
	KLAUNCH(k_uf_tiles, dim3((V + UF_TILE - 1) / UF_TILE), dim3(UF_TPB), 0, s, V, g.off, g.aoth, st.label,
			   st.hook, xcount, xlist, any_loop);
	if (E) {
		KLAUNCH(k_uf_cross, dim3(std::min<unsigned>(nblk(E), 2048)), dim3(TPB), 0, s, xcount, xlist, g.aoth,
				   st.label, st.hook);
	}
	uint8_t *is_root = reinterpret_cast<uint8_t *>(st.flag);
	KLAUNCH(k_uf_flatten, dim3(nblk(((size_t)V + 3) / 4)), dim3(TPB), 0, s, V, st.label, is_root, st.stats + 9);
	KLAUNCH(k_labels_sorted, dim3(nblk(V)), dim3(TPB), 0, s, V, st.label, st.stats + 9);
	scan_exclusive_u8(is_root, st.crank, (size_t)V + 1, nullptr, nullptr, 0, st.scan_tmp, st.scan_tmp_bytes, s);
	tm.end(7);
	uint32_t *h = st.host->take<uint32_t>(3); // component count, the order flag and the self-loop flag in one round trip
	publish_words(h, WordSrc{{st.crank + V, st.stats + 9, any_loop}}, 3, s);
	return h; // (valid once the stream -- or an event recorded now -- has completed: label_components_finish)
}
uint32_t label_components_finish(CompState &st, const uint32_t *h)
{
	st.comp_sorted = h[1] == 0;
	st.has_self_loops = h[2] != 0;
	return h[0];
}
uint32_t label_components(const ResidentGraph &g, CompState &st, StageTimer &tm, hipStream_t s)
{
	const uint32_t *h = label_components_enqueue(g, st, tm, s);
	HIP_CHECK(hipStreamSynchronize(s));
	return label_components_finish(st, h);
}

// the re-index's adjacency kernel in the form it takes on a graph whose vertices are grouped by component, without hubs and
// self loops -- started by povu_hip_decompose BEFORE the labelling's answer has reached the host (see there)
void reindex_speculative_adj(const ResidentGraph &g, const CompState &st, uint32_t *ladj, uint32_t *lle, hipStream_t s)
{
	const size_t nS = 2 * (size_t)g.V;
	KLAUNCH(k_local_adj, dim3(nblk(nS)), dim3(TPB), 0, s, g.V, (const uint32_t *)nullptr, (const uint32_t *)nullptr, g.off, g.aoth, g.atwin,
		(const uint32_t *)nullptr, g.off, st.hook, ladj, lle, false);
}


void reindex_components(const ResidentGraph &g, CompState &st, uint32_t C, StageTimer &tm, hipStream_t s,
			bool force_sorted_adjacency, bool adj_done)
{
	const uint32_t V = g.V, E = g.E;
	const size_t nS = 2 * (size_t)V;
	tm.begin("component_reindex");
	uint32_t launches = 0;
	// stable sort of vertices by component rank: local vertex idx = rank inside the component,
	// ascending global idx (comp_vtxs is a std::set, bidirected.cpp:552-555)
	const bool identity = C == 1 || st.comp_sorted;
	const bool sort_free = sort_free_adjacency(g, force_sorted_adjacency);
	// One component, or components one after the other in the vertex order: the order already is (component, idx), sorted
	// space is the global vertex space.  The sort-free builder then needs no permutation, position, slot-base, id or tip
	// array at all -- it reads the resident graph's own (slot base of vertex i = off[2 i]).
	st.lean_identity = identity && sort_free;
	KLAUNCH(k_comp_of, dim3(nblk(((size_t)V + 3) / 4)), dim3(TPB), 0, s, V, st.label, st.crank, st.comp_of, st.lean_identity ? nullptr : st.tmp_a, C,
			   (unsigned long long *)st.start_key);
	if (identity) { // the key / permutation arrays simply alias what k_comp_of wrote (component ranks, identity permutation)
		st.ckey = st.comp_of;
		st.perm = st.tmp_a;
	} else {
		sort_pairs_u32(st.comp_of, st.ckey, st.tmp_a, st.perm, V, bits_for(C), st.sort_tmp, st.sort_tmp_bytes, s);
	}
	if (st.lean_identity) {
		st.gid_s = g.vid;
		st.tip_s = g.tip;
	}
	KLAUNCH(k_sorted_vertices, dim3(nblk(((size_t)V + 3) / 4)), dim3(TPB), 0, s, V, C, st.ckey, st.lean_identity ? nullptr : st.perm, g.off, g.vid, g.tip,
			   st.pos, st.voff, st.vdeg, st.gid_s, st.tip_s, (unsigned long long *)st.start_key, st.stats);
	if (!st.lean_identity)
		scan_exclusive_u32(st.vdeg, st.sbase, (size_t)V + 1, st.scan_tmp, st.scan_tmp_bytes, s);
	launches += 5;
	// first-encounter rank of every edge
	if (sort_free) {
		const uint32_t *pos_or_identity = identity ? nullptr : st.pos; // sorted order = global order: no vertex is renumbered
		const uint32_t *perm = st.lean_identity ? nullptr : st.perm, *sbase = st.lean_identity ? nullptr : st.sbase;
		uint8_t *ldeg8 = reinterpret_cast<uint8_t *>(st.ldeg); // bytes here
		// Local offsets: without self loops (the labelling kernel looked) a side's local slots are its global ones; when the
		// vertices also keep their places the local offsets ARE the CSR's (no degree pass, no scan, no array)
		if (st.lean_identity && !st.has_self_loops) {
			st.loff = g.off;
			HIP_CHECK(copy_async(st.stats, &g.max_vdeg, 4, hipMemcpyHostToDevice, s)); // (an upper bound of the most links on one side)
		} else {
			KLAUNCH(k_local_degree, dim3(nblk(nS)), dim3(TPB), 0, s, V, perm, pos_or_identity, g.off, g.aoth, ldeg8, st.stats);
			scan_exclusive_u8(ldeg8, st.loff, nS + 1, nullptr, nullptr, 0, st.scan_tmp, st.scan_tmp_bytes, s);
		}
		if (!(adj_done && st.lean_identity && !st.has_self_loops)) // (adj_done: reindex_speculative_adj wrote exactly this into ladj / lle)
			KLAUNCH(k_local_adj, dim3(nblk(nS)), dim3(TPB), 0, s, V, perm, pos_or_identity, g.off, g.aoth, g.atwin, sbase,
				   st.loff, st.hook, st.ladj, st.lle, st.has_self_loops);
		KLAUNCH(k_comp_edge_offsets, dim3(nblk((size_t)C + 1)), dim3(TPB), 0, s, C, st.voff, st.loff, st.eoff, st.stats,
				   st.host_pub);
		st.dense_edges = false;
		tm.end(launches + 7);
		return;
	}
	fill_u32(st.first, E, POVU_NIL, s);
	KLAUNCH(k_first_slot, dim3(nblk(nS)), dim3(TPB), 0, s, V, st.perm, g.off, g.adj, st.sbase, st.first);
	HIP_CHECK(hipMemsetAsync(st.flag, 0, ((size_t)g.n_slots + 1) * 4, s));
	KLAUNCH(k_mark_first, dim3(nblk(nS)), dim3(TPB), 0, s, V, st.perm, g.off, g.adj, st.sbase, st.first,
			   st.flag);
	scan_exclusive_u32(st.flag, st.erank, (size_t)g.n_slots + 1, st.scan_tmp, st.scan_tmp_bytes, s);
	HIP_CHECK(hipMemsetAsync(st.ldeg, 0, (nS + 1) * 4, s));
	KLAUNCH(k_local_edges, dim3(nblk(nS)), dim3(TPB), 0, s, V, st.perm, st.pos, g.off, g.adj, st.sbase,
			   st.first, st.erank, g.v1, g.s1, g.v2, g.s2, st.keys, st.vals, st.ldeg, st.hook, g.atwin, st.la, st.lb,
			   st.tgray);
	HIP_CHECK(hipMemsetAsync(st.stats, 0, 16, s));
	KLAUNCH(k_max_u32, dim3(std::min<unsigned>(nblk(nS), 1024)), dim3(TPB), 0, s, (uint32_t)nS, st.ldeg, st.stats, (uint32_t *)nullptr);
	scan_exclusive_u32(st.ldeg, st.loff, nS + 1, st.scan_tmp, st.scan_tmp_bytes, s);
	KLAUNCH(k_comp_edge_offsets, dim3(nblk((size_t)C + 1)), dim3(TPB), 0, s, C, st.voff, st.loff, st.eoff, st.stats, st.host_pub);
	st.dense_edges = true; // la / lb hold the links in local edge order (povu_hip_componetize)
	// local per-side adjacency (other side ids), ascending local edge idx
	sort_pairs_u32(st.keys, st.keys2, st.vals, st.vals2, 2 * (size_t)E, bits_for(nS), st.sort_tmp, st.sort_tmp_bytes, s);
	if (E)
		KLAUNCH(k_local_slots, dim3(nblk(2 * (size_t)E)), dim3(TPB), 0, s, 2 * E, st.vals2, st.la, st.lb, st.tgray,
				   st.ladj, st.lle);
	launches += 10;
	tm.end(launches);
}

} // namespace povu_hip
