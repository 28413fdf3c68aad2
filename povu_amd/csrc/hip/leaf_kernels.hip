// leaf_kernels.hip -- the two relabelling passes of `povu decompose -s` on the state of an all-parallel pass.
//
// The reference runs five passes after find_flubbles when -s is given (app/subcommand/decompose.cpp:63-70).  The first
// two only relabel LEAF flubbles of the PVST (their line letter becomes T or O): find_tiny (src/povu/algorithms/
// tiny.cpp:100-129) and find_parallel (src/povu/algorithms/parallel.cpp:263-287).  Both ask questions about the
// spanning tree around a flubble's inner boundary vertices ai / zi (compute_ai_zi, flubbles.cpp:264-290) and about
// gen_tree_meta's bracket table (src/povu/graph/tree_utils.cpp:531-690): for a tree vertex c, the back edges of type
// back_edge that start strictly below c and end strictly above it.  The reference materialises that table per vertex
// (quadratic on deep trees); here every question has a closed form over pre-order intervals [v, v + size(v)):
//
//   |brackets(c)|         = sum over subtree(c) of (ordinary edges leaving - arriving) - those leaving c itself
//   a bracket of c ends at ai, for the only c that is ever asked (one descendant, c+1)
//                         = an ordinary back edge of c+1 ends at ai                       (two flag bytes per vertex)
//   back-edge INDEX of an edge of OBE(c) equals ai  (sic: tiny.cpp:52-56 compares indices with a vertex)
//                         : from_bd creates the ordinary edges of such a c right after those of c+1, so their indices
//                           are a RANGE that starts at (back edges created before c was discovered) + |OBE(c+1)|; the
//                           simplifying edge of c has index n_be0 + (capping / simplifying edges of later vertices)
//   child counts, |OBE|, |IBE| by type: atomics over the dense edge list.
//
// Undefined behaviour of the reference that the oracle DEFINES and this file follows (oracle/povu_oracle.c, "leaf
// subflubble passes"): in_branch without a gray child of zi answers "no"; a back edge whose source is its target is in
// no bracket table.  The three inserting passes (concealed, midi, smothered) run after these two: sub_kernels.hip.
// PARITY UNPINNED: the reference holds no T or O line anywhere; tests compare with the oracle's literal restatement.
#include "leaf_kernels.hpp"

namespace povu_hip
{

#define NIL POVU_NIL
static constexpr int TPB = 256;
static inline unsigned nblk(size_t n) { return (unsigned)((n + TPB - 1) / TPB); }
#define LAUNCH(k, n, s, ...)                                                                     \
	do {                                                                                     \
		if ((n) > 0) {                                                                   \
			hipLaunchKernelGGL(k, dim3(nblk(n)), dim3(TPB), 0, s, __VA_ARGS__);      \
			HIP_CHECK(hipGetLastError());                                            \
		}                                                                                \
	} while (0)

// component of a tree vertex: component c owns [2 voff[c] + c, 2 voff[c+1] + c]
struct CompOf {
	const uint32_t *voff;
	uint32_t C;
	__device__ __forceinline__ uint32_t operator()(uint32_t t) const
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
};

// global parent and child counts
__global__ void k_leaf_tree(uint32_t T, const uint32_t *__restrict__ t_size, const uint32_t *__restrict__ t_par, const CompOf comp_of,
			    uint32_t *__restrict__ gp, uint32_t *__restrict__ nchild)
{
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= T)
		return;
	uint32_t g = NIL;
	if (t_size[t] && t_par[t] != NIL) {
		const uint32_t c = comp_of(t);
		g = 2 * comp_of.voff[c] + c + t_par[t];
		atomicAdd(&nchild[g], 1u);
	}
	gp[t] = g;
}

// counts per tree vertex over the dense edge list: [0, NB0) ordinary, [NB0, NB0 + ncap) capping, the rest simplifying
__global__ void k_leaf_edges(uint32_t NB, uint32_t NB0, uint32_t ncap, const uint32_t *__restrict__ b_src,
			     const uint32_t *__restrict__ b_tgt, const uint32_t *__restrict__ t_size, const uint8_t *__restrict__ t_flags,
			     const uint32_t *__restrict__ gp, uint32_t *__restrict__ out_ord, uint32_t *__restrict__ in_ord,
			     uint32_t *__restrict__ nself, uint32_t *__restrict__ in_ext, uint8_t *__restrict__ capf,
			     uint8_t *__restrict__ simp, uint8_t *__restrict__ hit1, uint8_t *__restrict__ hit3)
{
	uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= NB)
		return;
	const uint32_t sv = b_src[j], tv = b_tgt[j];
	if (j < NB0) {
		atomicAdd(&out_ord[sv], 1u);
		atomicAdd(&in_ord[tv], 1u);
		if (sv == tv)
			atomicAdd(&nself[sv], 1u);
		// sv = c + 1 for a gray child c with exactly one descendant: does this edge end right above c's parent zi
		// (ai = zi - 1) or three above it (ai = zi - 3)?  Those are the only two ai find_tiny ever asks about.
		if (sv > 0) {
			const uint32_t c = sv - 1;
			if (t_size[c] == 2 && !(t_flags[c] & TF_BLACK)) {
				const uint32_t zi = gp[c];
				if (zi != NIL) {
					if (tv + 1 == zi)
						hit1[c] = 1;
					if (tv + 3 == zi)
						hit3[c] = 1;
				}
			}
		}
	} else {
		if (j < NB0 + ncap)
			capf[sv] = 1;
		else
			simp[sv] = 1;
		atomicAdd(&in_ext[tv], 1u);
	}
}
__global__ void k_leaf_extra_counts(uint32_t T, const uint8_t *__restrict__ capf, const uint8_t *__restrict__ simp, uint8_t *__restrict__ nx)
{
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t <= T)
		nx[t] = t < T ? (uint8_t)(capf[t] + simp[t]) : 0;
}
// w[t] = ordinary back edges created just before t is discovered (by its parent); add the ones created after the last
// child of every vertex whose subtree ends right before t.  The tails of subtrees that end with their component are
// created after the component's last discovery and count for nobody.
__global__ void k_leaf_closed(uint32_t T, const uint32_t *__restrict__ t_size, const uint32_t *__restrict__ tail, const CompOf comp_of,
			      const uint32_t *__restrict__ c_ntree, uint32_t *__restrict__ w)
{
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= T)
		return;
	const uint32_t sz = t_size[t], tl = tail[t];
	if (!sz || !tl)
		return;
	const uint32_t c = comp_of(t), base = 2 * comp_of.voff[c] + c, end = t + sz;
	if (end < base + c_ntree[c])
		atomicAdd(&w[end], tl);
}


// branches, tiny.cpp:33-73 (trunk, :76-97, is false on every path)
__device__ static bool leaf_tiny(const LeafIn &in, uint32_t ai, uint32_t zi, uint32_t d, uint32_t base, uint32_t N)
{
	const uint32_t end = zi + in.t_size[zi], ai_l = ai - base;
	for (uint32_t c = zi + 1; c < end; c += max(in.t_size[c], 1u)) { // Y = the gray children of zi
		if (in.t_flags[c] & TF_BLACK)
			continue;
		if (in.t_size[c] != 2) // has_one_descendants: post - pre == 3
			return false;
		bool hit = d == 1 ? in.hit1[c] != 0 : in.hit3[c] != 0; // a bracket of c (= an ordinary edge of c + 1) ends at ai
		if (!hit) { // (sic) the back-edge indices of OBE(c) against the vertex idx ai
			const uint32_t b0 = in.B[c + 1] - in.B[base] + in.out_ord[c + 1];
			hit = ai_l >= b0 && ai_l - b0 < in.out_ord[c];
			if (!hit && in.simp[c]) {
				const uint32_t idx = (in.O[base + N] - in.O[base]) + (in.X[base + N] - in.X[c + 1]) + in.capf[c];
				hit = idx == ai_l;
			}
		}
		if (!hit)
			return false;
	}
	return true;
}
// in_branch, parallel.cpp:186-261
__device__ static bool leaf_in_branch(const LeafIn &in, uint32_t ai, uint32_t zi, uint32_t d)
{
	if (d != 1)
		return false;
	const uint32_t end = zi + in.t_size[zi];
	uint32_t c = NIL;
	for (uint32_t x = zi + 1; x < end; x += max(in.t_size[x], 1u)) {
		if (in.t_flags[x] & TF_BLACK)
			continue;
		if (c != NIL)
			return false;
		c = x;
	}
	if (c == NIL) // undefined in the reference (tm.off[INVALID_IDX]); defined as "no" by the oracle
		return false;
	const uint32_t br = (in.P[c + in.t_size[c]] - in.P[c]) - (in.out_ord[c] - in.nself[c]);
	const uint32_t ch_obe = in.out_ord[c] + in.capf[c] + in.simp[c];
	if (br <= 2)
		return false;
	if (in.in_ord[ai] >= br + ch_obe)
		return true;
	return in.out_ord[zi] + in.capf[zi] + in.simp[zi] >= br + ch_obe;
}
// in_trunk + inspect_trunk, parallel.cpp:15-184
__device__ static bool leaf_in_trunk(const LeafIn &in, uint32_t ai, uint32_t zi, uint32_t d)
{
	if (d <= 3 && in.in_ord[ai] + in.in_ext[ai] <= 1) // condition i
		return false;
	if (in.nchild[zi] != 1) // condition iii
		return false;
	// condition iv: no branching vertex from zi up to (not including) ai, or exactly one that has a child behind zi
	// with a single descendant
	uint32_t branching = NIL;
	for (uint32_t v = zi; v != ai; v = in.gp[v]) {
		if (in.gp[v] == NIL)
			return false;
		if (in.nchild[v] > 1) {
			if (branching != NIL)
				return false;
			branching = v;
		}
	}
	if (branching != NIL) {
		bool ok = false;
		const uint32_t end = branching + in.t_size[branching];
		for (uint32_t x = branching + 1; x < end && !ok; x += max(in.t_size[x], 1u))
			ok = x > zi && in.t_size[x] == 2;
		if (!ok)
			return false;
	}
	const uint32_t in_ai = in.in_ord[ai];
	if (2 * in_ai >= d - 3) // u32 on purpose: wraps like the reference when zi - ai < 3
		return true;
	if (in_ai != 0)
		return false;
	return 2 * in.out_ord[zi] >= d - 3;
}

// the letter of a flubble with inner boundary vertices ai / zi (global tree vertex idx) of the component at `base`
__device__ static uint8_t leaf_label(const LeafIn &in, uint32_t ai, uint32_t zi, uint32_t base, uint32_t N, bool leaf)
{
	if (!leaf)
		return FAM_FLUBBLE;
	const uint32_t d = zi - ai;
	if ((d == 1 || d == 3) && leaf_tiny(in, ai, zi, d, base, N)) // find_tiny first; find_parallel skips what it relabelled
		return FAM_TINY;
	if (leaf_in_branch(in, ai, zi, d) || leaf_in_trunk(in, ai, zi, d))
		return FAM_PARALLEL;
	return FAM_FLUBBLE;
}
__device__ __forceinline__ void sort4(uint32_t (&v)[4])
{
#pragma unroll
	for (int a = 0; a < 3; a++)
#pragma unroll
		for (int b = 0; b < 3 - a; b++)
			if (v[b] > v[b + 1]) {
				const uint32_t x = v[b];
				v[b] = v[b + 1];
				v[b + 1] = x;
			}
}

__global__ void k_leaf_flubbles(const LeafIn in, LeafOut out)
{
	uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= in.NE)
		return;
	const uint32_t i = in.e_i[j], c = in.s_comp[i], base = 2 * in.voff[c] + c, N = in.c_ntree[c];
	const uint32_t va = in.s_vtx[i], vz = in.s_vtx[in.ns[i]];
	// compute_ai_zi, flubbles.cpp:264-290: the middle two of the four end vertices of the two boundary tree edges
	uint32_t v[4] = {va, in.gp[va], vz, in.gp[vz]};
	sort4(v);
	const uint32_t ai = v[1], zi = v[2];
	const uint64_t q = (uint64_t)j + in.cproc_ps[c] + 1;
	out.ai[q] = ai - base;
	out.zi[q] = zi - base;
	// the next emitted flubble is this one's child iff it lies deeper (k_pvst_emit: parent = nearest earlier flubble
	// with a smaller level)
	const bool leaf = !(j + 1 < in.NE && in.s_comp[in.e_i[j + 1]] == c && in.lev[j + 1] > in.lev[j]);
	out.fam[q] = leaf_label(in, ai, zi, base, N, leaf);
}
__global__ void k_leaf_roots(uint32_t C, const uint32_t *__restrict__ c_ntree, const uint32_t *__restrict__ doff, LeafOut out)
{
	uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= C || c_ntree[c] == 0)
		return;
	const uint32_t q = doff[c];
	out.ai[q] = out.zi[q] = NIL;
	out.fam[q] = FAM_DUMMY;
}

// ---- components whose add_flubbles went through the sequential redo: their PVST sits in the per-component layout of the
// one-lane kernels (component c owns the slots from voff[c] + c; seq_pvst also left ai / zi there)
struct SlotComp {
	const uint32_t *voff;
	uint32_t C;
	__device__ __forceinline__ uint32_t operator()(uint32_t slot) const
	{
		uint32_t lo = 0, hi = C; // last c with voff[c] + c <= slot
		while (hi - lo > 1) {
			const uint32_t mid = (lo + hi) >> 1;
			if (voff[mid] + mid <= slot)
				lo = mid;
			else
				hi = mid;
		}
		return lo;
	}
};
__global__ void k_leaf_seq_children(uint32_t P, const SlotComp comp_of, const uint32_t *__restrict__ comp_bad,
				    const uint32_t *__restrict__ c_npvst, const uint32_t *__restrict__ p_parent,
				    uint8_t *__restrict__ has_child)
{
	uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= P)
		return;
	const uint32_t c = comp_of(slot), pb = comp_of.voff[c] + c, k = slot - pb;
	if (!comp_bad[c] || k == 0 || k >= c_npvst[c])
		return;
	has_child[pb + p_parent[slot]] = 1;
}
__global__ void k_leaf_seq_eval(uint32_t P, const LeafIn in, const SlotComp comp_of, const uint32_t *__restrict__ comp_bad,
				const uint32_t *__restrict__ c_npvst, const uint32_t *__restrict__ p_ai,
				const uint32_t *__restrict__ p_zi, const uint8_t *__restrict__ has_child, uint8_t *__restrict__ p_fam)
{
	uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= P)
		return;
	const uint32_t c = comp_of(slot), pb = comp_of.voff[c] + c, k = slot - pb;
	if (!comp_bad[c] || k >= c_npvst[c])
		return;
	if (k == 0) {
		p_fam[slot] = FAM_DUMMY;
		return;
	}
	const uint32_t base = 2 * in.voff[c] + c;
	p_fam[slot] = leaf_label(in, base + p_ai[slot], base + p_zi[slot], base, in.c_ntree[c], has_child[slot] == 0);
}

template <typename F>
static void leaf_spans(size_t V, size_t C, size_t total, F &&take)
{
	const size_t T = 2 * V + C, P = V + C;
	for (int k = 0; k < 12; k++) // gp nchild out_ord in_ord nself in_ext P O X w tail B
		take((T + 4) * 4);
	for (int k = 0; k < 5; k++) // capf simp hit1 hit3 nx
		take(T + 64);
	take(total * 4 + 64);
	take(total * 4 + 64);
	take(total + 64);
	take((P + 4) * 4); // p_ai p_zi p_fam has_child: the redo's layout
	take((P + 4) * 4);
	take(P + 64);
	take(P + 64);
	take(scan_tmp_bytes(T + 4));
}
size_t leaf_workspace_bytes(size_t V, size_t C, size_t total)
{
	size_t sum = 0;
	leaf_spans(V, C, total, [&](size_t b) { sum += ((b + 255) & ~size_t(255)) + 256; });
	return sum + (1 << 16);
}

void leaf_prepare(const CompState &cs, const SeqWs &sw, const ParWs &pw, const TreeWs &tw, uint32_t C, Arena &ar, LeafState &ls,
		  hipStream_t s)
{
	const uint32_t V = sw.V, T = 2 * V + C;
	const size_t total = pw.d_total, P = (size_t)V + C;
	ar.reserve(leaf_workspace_bytes(V, C, total));
	uint32_t *w32[12];
	for (auto &p : w32)
		p = ar.take<uint32_t>((size_t)T + 4);
	uint32_t *gp = w32[0], *nchild = w32[1], *out_ord = w32[2], *in_ord = w32[3], *nself = w32[4], *in_ext = w32[5], *Ps = w32[6],
		 *O = w32[7], *X = w32[8], *w = w32[9], *tail = w32[10], *B = w32[11];
	uint8_t *w8[5];
	for (auto &p : w8)
		p = ar.take<uint8_t>((size_t)T + 64);
	uint8_t *capf = w8[0], *simp = w8[1], *hit1 = w8[2], *hit3 = w8[3], *nx = w8[4];
	ls.dense.ai = ar.take<uint32_t>(total + 16);
	ls.dense.zi = ar.take<uint32_t>(total + 16);
	ls.dense.fam = ar.take<uint8_t>(total + 64);
	ls.p_ai = ar.take<uint32_t>(P + 4);
	ls.p_zi = ar.take<uint32_t>(P + 4);
	ls.p_fam = ar.take<uint8_t>(P + 64);
	ls.p_has_child = ar.take<uint8_t>(P + 64);
	ls.P = P;
	const size_t tmp_bytes = scan_tmp_bytes((size_t)T + 4);
	void *tmp = ar.take<char>(tmp_bytes);

	for (uint32_t *p : {nchild, out_ord, in_ord, nself, in_ext, w, tail})
		HIP_CHECK(hipMemsetAsync(p, 0, ((size_t)T + 4) * 4, s));
	for (uint8_t *p : {capf, simp, hit1, hit3})
		HIP_CHECK(hipMemsetAsync(p, 0, (size_t)T + 64, s));
	const CompOf comp_of{cs.voff, C};
	LAUNCH(k_leaf_tree, T, s, T, sw.t_size, sw.t_par, comp_of, gp, nchild);
	const uint32_t NB = pw.nb0 + pw.ncap + pw.nsimp;
	LAUNCH(k_leaf_edges, NB, s, NB, pw.nb0, pw.ncap, pw.b_src, pw.b_tgt, sw.t_size, sw.t_flags, gp, out_ord, in_ord, nself, in_ext, capf, simp,
	       hit1, hit3);
	LAUNCH(k_leaf_extra_counts, (size_t)T + 1, s, T, capf, simp, nx);
	scan_exclusive_diff_u32(out_ord, in_ord, Ps, (size_t)T + 1, tmp, tmp_bytes, s);
	scan_exclusive_u32(out_ord, O, (size_t)T + 1, tmp, tmp_bytes, s);
	scan_exclusive_u8(nx, X, (size_t)T + 1, nullptr, nullptr, 0, tmp, tmp_bytes, s);
	debug_edge_id_weights(cs, sw, tw, w, tail, s);
	LAUNCH(k_leaf_closed, T, s, T, sw.t_size, tail, comp_of, sw.c_ntree, w);
	scan_exclusive_u32(w, B, (size_t)T + 2, tmp, tmp_bytes, s);

	LeafIn &in = ls.in;
	in = LeafIn{};
	in.NE = pw.n_emitted;
	in.C = C;
	in.voff = cs.voff;
	in.c_ntree = sw.c_ntree;
	in.cproc_ps = pw.cproc_ps;
	in.e_i = pw.e_i;
	in.lev = pw.lev;
	in.s_comp = pw.s_comp;
	in.s_vtx = pw.s_vtx;
	in.ns = pw.ns;
	in.t_size = sw.t_size;
	in.gp = gp;
	in.nchild = nchild;
	in.t_flags = sw.t_flags;
	in.out_ord = out_ord;
	in.in_ord = in_ord;
	in.nself = nself;
	in.in_ext = in_ext;
	in.capf = capf;
	in.simp = simp;
	in.hit1 = hit1;
	in.hit3 = hit3;
	in.P = Ps;
	in.O = O;
	in.X = X;
	in.B = B;
}

void leaf_dense(const LeafState &ls, const SeqWs &sw, const ParWs &pw, uint32_t C, hipStream_t s)
{
	LAUNCH(k_leaf_roots, C, s, C, sw.c_ntree, pw.doff, ls.dense);
	LAUNCH(k_leaf_flubbles, ls.in.NE, s, ls.in, ls.dense);
}

void leaf_seq(const LeafState &ls, const CompState &cs, const SeqWs &sw, const uint32_t *comp_bad, uint32_t C, hipStream_t s)
{
	const uint32_t P = (uint32_t)ls.P;
	const SlotComp comp_of{cs.voff, C};
	HIP_CHECK(hipMemsetAsync(ls.p_has_child, 0, ls.P + 64, s));
	LAUNCH(k_leaf_seq_children, P, s, P, comp_of, comp_bad, sw.c_npvst, sw.p_parent, ls.p_has_child);
	LAUNCH(k_leaf_seq_eval, P, s, P, ls.in, comp_of, comp_bad, sw.c_npvst, ls.p_ai, ls.p_zi, ls.p_has_child, ls.p_fam);
}

} // namespace povu_hip
