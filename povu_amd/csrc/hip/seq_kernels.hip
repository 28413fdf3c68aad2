// seq_kernels.hip -- rows C-G of the scope table, one lane per component.
//
// This is the always-correct traversal path: every stage follows the
// reference's sequential semantics exactly (lexicographic biedged DFS, reverse
// pre-order bracket pass, candidate stack, next_seen, PVST stack machine).  It
// is parallel across components only -- many small components fill the chip, a
// single giant component is bounded by dependent-load latency.  The parallel
// stage kernels (par_kernels.hip) replace it stage by stage where an exact
// parallel formulation exists; this file remains the fallback they defer to.
//
// Reference: pst::Tree::from_bd            spanning_tree.cpp:262-463
//            handle_vertex / cycle equiv   flubbles.cpp:503-719, bracket_list.cpp:61-100
//            br_desc + eq-class stack      tree_utils.cpp:19-155, flubbles.cpp:412-501
//            compute_eq_class_metadata     flubbles.cpp:375-410
//            add_flubbles                  flubbles.cpp:295-367
#include "seq_kernels.hpp"

#include <algorithm>

namespace povu_hip
{

#define NIL POVU_NIL

struct CompView {
	uint32_t nv, ne, N;	 // vertices, links, tree vertices
	uint32_t Sb;		 // first sorted side id of the component
	const uint32_t *loff, *ladj; // global arrays (indexed by sorted side id / slot)
	const uint32_t *gid_s;
	// tree arrays, local tree idx
	uint32_t *gid, *par, *cls, *hi, *fchild, *nsib, *lchild, *size, *depth;
	uint8_t *tf;
	uint32_t *ctr, *cur; // indexed by (sorted side id - Sb)
	uint32_t *stk;
	uint8_t *selfloop; // indexed by local vertex
	uint32_t *be_src, *be_tgt, *o_next, *i_next, *b_prev, *b_next, *b_rsize, *b_rclass;
	uint8_t *be_type, *b_in, *be_cdef;
	uint32_t *o_head, *o_tail, *i_head, *i_tail, *l_head, *l_tail, *l_size, *bl;
	uint32_t *nxt, *st_head, *st_tail;
	uint32_t *s_vtx, *s_cls, *next_seen, *last;
	uint32_t *p_parent, *p_a, *p_z, *aux, *p_ai, *p_zi;
	uint8_t *p_or, *in_s;
	uint64_t *hairpins;
	uint32_t n_be, n_class, n_bry;
};

__device__ static inline void add_child(CompView &c, uint32_t p, uint32_t ch)
{
	// children are created in increasing tree idx, so appending keeps std::set order
	c.nsib[ch] = NIL;
	if (c.fchild[p] == NIL)
		c.fchild[p] = ch;
	else
		c.nsib[c.lchild[p]] = ch;
	c.lchild[p] = ch;
}

__device__ static inline uint32_t add_be(CompView &c, uint32_t src, uint32_t tgt, uint8_t type)
{
	uint32_t b = c.n_be++;
	c.be_src[b] = src;
	c.be_tgt[b] = tgt;
	c.be_type[b] = type;
	c.b_in[b] = 0;
	c.be_cdef[b] = 0;
	c.o_next[b] = NIL;
	if (c.o_head[src] == NIL)
		c.o_head[src] = b;
	else
		c.o_next[c.o_tail[src]] = b;
	c.o_tail[src] = b;
	c.i_next[b] = NIL;
	if (c.i_head[tgt] == NIL)
		c.i_head[tgt] = b;
	else
		c.i_next[c.i_tail[tgt]] = b;
	c.i_tail[tgt] = b;
	return b;
}

// ------------------------------------------------------------------ row C
__device__ static void seq_spanning_tree(CompView &c, uint64_t start_key)
{
	const bool has_tips = start_key != ~0ull;
	c.N = 2 * c.nv + (has_tips ? 1u : 0u);
	// fchild / o_head / i_head / ctr are pre-filled with NIL, cur and selfloop with 0 (host memsets)
	uint32_t counter = 0, sp = 0, p = NIL;
	c.n_be = 0;
	if (has_tips) { // dummy root, spanning_tree.cpp:397-402
		c.gid[0] = NIL;
		c.tf[0] = 2;
		c.par[0] = NIL;
		c.depth[0] = 0;
		counter = 1;
		p = 0;
	}
	const uint32_t start = has_tips ? (uint32_t)(start_key & 0xFFFFFFFFu) - c.Sb : 0u; // (l, idx 0) otherwise

	auto add_segment = [&](uint32_t ls) { // add_vertex_to_tree, :324-356 (ls = local side entered)
		uint32_t a = counter++, b = counter++;
		uint32_t id = c.gid_s[(c.Sb + ls) >> 1];
		c.gid[a] = id;
		c.tf[a] = (uint8_t)(ls & 1);
		c.gid[b] = id;
		c.tf[b] = (uint8_t)(((ls & 1) ^ 1) | TF_BLACK);
		c.ctr[ls] = a;
		c.ctr[ls ^ 1] = b;
		c.par[a] = p;
		c.depth[a] = p != NIL ? c.depth[p] + 1 : 0;
		c.depth[b] = c.depth[a] + 1;
		if (p != NIL)
			add_child(c, p, a);
		c.par[b] = a;
		add_child(c, a, b);
		c.stk[sp++] = ls;
		c.stk[sp++] = ls ^ 1;
	};
	add_segment(start);

	while (sp) { // :419-453
		const uint32_t ls = c.stk[sp - 1];
		p = c.ctr[ls];
		const uint32_t lo = c.loff[c.Sb + ls], hi = c.loff[c.Sb + ls + 1];
		if (lo == hi) {
			// a side without links points back at the root unless already joined to it
			// (:433-438); the pair (p, root) can only be joined by the tree edge root->p
			if (p == 0 || c.par[p] != 0)
				add_be(c, p, 0, 0);
		}
		bool found = false;
		uint32_t k = lo + c.cur[ls];
		for (; k < hi; k++) {
			const uint32_t lo_side = c.ladj[k] - c.Sb; // other side, local
			const uint32_t x = c.ctr[lo_side];
			if (x == NIL) { // unvisited segment: tree edge, descend (:366-376)
				add_segment(lo_side);
				found = true;
				k++;
				break;
			}
			if ((lo_side >> 1) == (ls >> 1)) {
				// self loop: the two sides are joined by the black edge, one extra back
				// edge per segment (:387-395)
				if (!c.selfloop[ls >> 1]) {
					add_be(c, p, x, 0);
					c.selfloop[ls >> 1] = 1;
				}
				continue;
			}
			// are_connected(p, x), :293-307: a finished descendant has already joined itself
			// to p; towards the root only the tree parent or an earlier parallel link is
			if (x > p || x == c.par[p])
				continue;
			bool dup = false;
			for (uint32_t j = lo; j < k; j++)
				if (c.ladj[j] - c.Sb == lo_side) {
					dup = true;
					break;
				}
			if (!dup)
				add_be(c, p, x, 0);
		}
		c.cur[ls] = k - lo;
		if (!found) {
			c.size[p] = counter - p; // tree idx = pre-order number, so the subtree is [p, counter)
			sp--;
		}
	}
	if (has_tips)
		c.size[0] = counter;
	if (counter != c.N)
		c.N = 0; // not every side was reached: the caller reports an internal error
}

// ------------------------------------------------------------------ row D
__device__ static inline void bl_push(CompView &c, uint32_t v, uint32_t b)
{
	if (c.bl[v] == NIL) {
		c.bl[v] = v;
		c.l_head[v] = c.l_tail[v] = NIL;
		c.l_size[v] = 0;
	}
	const uint32_t L = c.bl[v];
	c.b_rsize[b] = NIL;
	c.b_rclass[b] = NIL;
	c.b_prev[b] = NIL;
	c.b_next[b] = c.l_head[L];
	if (c.l_head[L] != NIL)
		c.b_prev[c.l_head[L]] = b;
	else
		c.l_tail[L] = b;
	c.l_head[L] = b;
	c.l_size[L]++;
	c.b_in[b] = 1;
}

__device__ static void seq_cycle_classes(CompView &c, bool want_hairpins)
{
	const uint32_t N = c.N; // bl / hi / cls are pre-filled with NIL
	uint32_t n_class = 0;
	bool in_hairpin = false;
	uint64_t b1 = NIL, b2 = NIL;
	c.n_bry = 0;
	for (uint32_t v = N; v-- > 0;) {
		uint32_t hi_0 = NIL; // :515-519
		for (uint32_t b = c.o_head[v]; b != NIL; b = c.o_next[b])
			hi_0 = min(hi_0, c.be_tgt[b]);
		const bool is_leaf = c.fchild[v] == NIL, is_root = c.par[v] == NIL;
		if (in_hairpin && ((is_leaf && !is_root) || is_root)) { // :531-535
			if (want_hairpins) {
				c.hairpins[2 * c.n_bry] = b1;
				c.hairpins[2 * c.n_bry + 1] = b2;
			}
			c.n_bry++;
			b1 = b2 = NIL;
			in_hairpin = false;
		}
		uint32_t hi_1 = NIL; // :540-550
		for (uint32_t ch = c.fchild[v]; ch != NIL; ch = c.nsib[ch])
			hi_1 = min(hi_1, c.hi[ch]);
		c.hi[v] = min(hi_0, hi_1);
		uint32_t hi_child = NIL; // :555-561
		for (uint32_t ch = c.fchild[v]; ch != NIL; ch = c.nsib[ch])
			if (c.hi[ch] == hi_1) {
				hi_child = ch;
				break;
			}
		uint32_t hi_2 = NIL; // :566-574
		for (uint32_t ch = c.fchild[v]; ch != NIL; ch = c.nsib[ch])
			if (ch != hi_child && c.hi[ch] < v) {
				hi_2 = c.hi[ch];
				break;
			}
		// children's bracket lists, ascending: the first is adopted, later ones go in front
		for (uint32_t ch = c.fchild[v]; ch != NIL; ch = c.nsib[ch]) {
			const uint32_t Lc = c.bl[ch];
			if (c.bl[v] == NIL) {
				c.bl[v] = Lc;
			} else if (Lc != NIL && c.l_head[Lc] != NIL) {
				const uint32_t Lp = c.bl[v];
				if (c.l_head[Lp] != NIL) {
					c.b_next[c.l_tail[Lc]] = c.l_head[Lp];
					c.b_prev[c.l_head[Lp]] = c.l_tail[Lc];
				} else {
					c.l_tail[Lp] = c.l_tail[Lc];
				}
				c.l_head[Lp] = c.l_head[Lc];
				c.l_size[Lp] += c.l_size[Lc];
			}
		}
		// brackets that end here, :594-605
		for (uint32_t b = c.i_head[v]; b != NIL; b = c.i_next[b]) {
			if (c.bl[v] != NIL && c.b_in[b]) {
				const uint32_t L = c.bl[v];
				if (c.b_prev[b] != NIL)
					c.b_next[c.b_prev[b]] = c.b_next[b];
				else
					c.l_head[L] = c.b_next[b];
				if (c.b_next[b] != NIL)
					c.b_prev[c.b_next[b]] = c.b_prev[b];
				else
					c.l_tail[L] = c.b_prev[b];
				c.l_size[L]--;
				c.b_in[b] = 0;
			}
			if (c.be_type[b] != 1 && !c.be_cdef[b]) {
				c.be_cdef[b] = 1;
				n_class++;
			}
		}
		// brackets that start here, ascending back-edge idx, :608-611 (the list of outgoing
		// edges is walked before the capping edge is appended to it)
		{
			const uint32_t last_o = c.o_tail[v];
			for (uint32_t b = c.o_head[v]; b != NIL; b = c.o_next[b]) {
				bl_push(c, v, b);
				if (b == last_o)
					break;
			}
		}
		if (hi_2 < hi_0) // capping, :613-619
			bl_push(c, v, add_be(c, v, hi_2, 1));
		if (c.bl[v] == NIL || c.l_size[c.bl[v]] == 0) { // simplifying, :621-643
			if ((c.tf[v] & TF_TYPE_MASK) != 2)
				b1 = c.gid[v];
			bl_push(c, v, add_be(c, v, 0, 2));
			c.hi[v] = 0;
			in_hairpin = true;
		} else if (in_hairpin) { // :644-656
			if (c.be_type[c.l_head[c.bl[v]]] == 2)
				b2 = c.gid[v];
		}
		if (!is_root) { // :664-686
			const uint32_t L = c.bl[v], b = c.l_head[L];
			if (c.l_size[L] != c.b_rsize[b]) {
				c.b_rsize[b] = c.l_size[L];
				c.b_rclass[b] = n_class++;
			}
			c.cls[v] = c.b_rclass[b];
			if (c.b_rsize[b] == 1)
				c.be_cdef[b] = 1;
		}
	}
	c.n_class = n_class;
}

// ------------------------------------------------------------------ row E
__device__ static uint32_t seq_candidate_stack(CompView &c)
{
	const uint32_t N = c.N; // st_head / st_tail are pre-filled with NIL
	uint32_t m_head = NIL, m_tail = NIL;
	for (uint32_t v = N; v-- > 0;) {
		const uint32_t fc = c.fchild[v];
		const bool branching = fc != NIL && c.nsib[fc] != NIL;
		if (v == 0 || branching) {
			// sorted_br = [black child, gray children by idx descending] (tree_utils.cpp:47-76),
			// each spliced to the front (flubbles.cpp:446-457) => gray ascending, black last.
			// The black child, when there is one, is always the first (lowest idx) child.
			uint32_t black = (fc != NIL && (c.tf[fc] & TF_BLACK)) ? fc : NIL;
			auto splice = [&](uint32_t ch) {
				if (c.st_head[ch] == NIL)
					return;
				c.nxt[c.st_tail[ch]] = m_head;
				if (m_head == NIL)
					m_tail = c.st_tail[ch];
				m_head = c.st_head[ch];
				c.st_head[ch] = c.st_tail[ch] = NIL;
			};
			if (black != NIL)
				splice(black);
			// gray children descending: walk the sibling list backwards via a reversal
			// in place (restored afterwards is unnecessary: the lists are not used again
			// by this stage, and later stages only need fchild/nsib of other vertices)
			uint32_t rev = NIL, ch = (black != NIL) ? c.nsib[fc] : fc;
			while (ch != NIL) {
				uint32_t nx = c.nsib[ch];
				c.lchild[ch] = rev;
				rev = ch;
				ch = nx;
			}
			for (ch = rev; ch != NIL; ch = c.lchild[ch])
				splice(ch);
		}
		if (v == 0)
			break;
		if (c.tf[v] & TF_BLACK) {
			c.nxt[v] = m_head;
			if (m_head == NIL)
				m_tail = v;
			m_head = v;
		}
		const uint32_t pfc = c.fchild[c.par[v]];
		if (c.nsib[pfc] != NIL) { // parent is branching: park
			c.st_head[v] = m_head;
			c.st_tail[v] = m_tail;
			m_head = m_tail = NIL;
		}
	}
	uint32_t n = 0;
	for (uint32_t e = m_head; e != NIL; e = c.nxt[e]) {
		c.s_vtx[n] = e;
		c.s_cls[n] = c.cls[e];
		n++;
	}
	return n;
}

// ------------------------------------------------------------------ rows F, G
__device__ static uint32_t seq_pvst(CompView &c, uint32_t n, bool given_stack)
{
	// last is pre-filled with NIL, in_s with 0
	uint32_t cls_base = 0;
	if (given_stack) { // next_seen is there already; the class ids count over all components: this one's are a contiguous range
		cls_base = NIL;
		for (uint32_t i = 0; i < n; i++)
			cls_base = min(cls_base, c.s_cls[i]);
	} else {
		for (uint32_t i = n; i-- > 0;) { // compute_eq_class_metadata, flubbles.cpp:391-399
			const uint32_t cl = c.s_cls[i];
			c.next_seen[i] = c.last[cl] != NIL ? c.last[cl] : i;
			c.last[cl] = i;
		}
	}
	uint32_t np = 1, sp = 0, prt = 0;
	c.p_parent[0] = NIL;
	c.p_a[0] = c.p_z[0] = NIL;
	c.p_or[0] = 0;
	if (c.p_ai)
		c.p_ai[0] = c.p_zi[0] = NIL;
	for (uint32_t i = 0; i < n; i++) { // add_flubbles, flubbles.cpp:316-365
		const uint32_t cl = c.s_cls[i] - cls_base;
		if (c.in_s[cl]) {
			while (sp) {
				const uint32_t k = c.aux[--sp];
				c.in_s[k] = 0;
				if (k == cl)
					break;
			}
			if (prt != 0)
				prt = c.p_parent[prt];
		}
		if (i + 1 < c.next_seen[i]) {
			const uint32_t va = c.s_vtx[i], vz = c.s_vtx[c.next_seen[i]];
			// orientation: child vertex of type r => forward (flubbles.cpp:466-473)
			const uint32_t ra = ((c.tf[va] & TF_TYPE_MASK) == 1) ? 0u : 1u;
			const uint32_t rz = ((c.tf[vz] & TF_TYPE_MASK) == 1) ? 0u : 1u;
			const uint32_t k = np++;
			if (ra && rz) { // normalize_endpoints, :233-244
				c.p_a[k] = c.gid[vz];
				c.p_z[k] = c.gid[va];
				c.p_or[k] = 0;
			} else {
				c.p_a[k] = c.gid[va];
				c.p_z[k] = c.gid[vz];
				c.p_or[k] = (uint8_t)(ra | (rz << 1));
			}
			if (c.p_ai) { // compute_ai_zi, flubbles.cpp:264-290: the middle two of the boundary edges' four end vertices
				uint32_t v[4] = {va, c.par[va], vz, c.par[vz]};
				for (int a = 0; a < 3; a++)
					for (int b = 0; b < 3 - a; b++)
						if (v[b] > v[b + 1]) {
							const uint32_t x = v[b];
							v[b] = v[b + 1];
							v[b + 1] = x;
						}
				c.p_ai[k] = v[1];
				c.p_zi[k] = v[2];
			}
			c.p_parent[k] = prt;
			prt = k;
		}
		c.aux[sp++] = cl;
		c.in_s[cl] = 1;
	}
	return np;
}

__global__ void __launch_bounds__(64) k_seq_components(SeqWs w)
{
	const uint32_t lane = threadIdx.x;
	// component order is size-descending; slot (lane, block) takes order[lane * gridDim + block]
	// so that the largest components land in different waves
	for (uint64_t slot = (uint64_t)lane * gridDim.x + blockIdx.x; slot < w.C; slot += 64ull * gridDim.x) {
		const uint32_t ci = w.order[slot];
		if (w.world > 1 && w.owner[ci] != w.rank)
			continue;
		if (w.comp_sel && !w.comp_sel[ci])
			continue;
		CompView c;
		c.nv = w.voff[ci + 1] - w.voff[ci];
		c.ne = w.eoff[ci + 1] - w.eoff[ci];
		if (c.nv < 3) // decompose.cpp:135-142
			continue;
		const uint64_t tb = 2ull * w.voff[ci] + ci;
		const uint64_t bb = (uint64_t)w.eoff[ci] + w.voff[ci] + 2 * tb;
		const uint64_t pb = (uint64_t)w.voff[ci] + ci;
		c.Sb = 2 * w.voff[ci];
		c.loff = w.loff;
		c.ladj = w.ladj;
		c.gid_s = w.gid_s;
		c.gid = w.t_gid + tb;
		c.par = w.t_par + tb;
		c.cls = w.t_cls + tb;
		c.hi = w.t_hi + tb;
		c.fchild = w.first_child + tb;
		c.nsib = w.next_sib + tb;
		c.lchild = w.last_child + tb;
		c.size = w.t_size + tb;
		c.depth = w.t_depth + tb;
		c.tf = w.t_flags + tb;
		c.ctr = w.ctr + c.Sb;
		c.cur = w.cur + c.Sb;
		c.stk = w.stk + tb;
		c.selfloop = w.selfloop + w.voff[ci];
		c.be_src = w.be_src + bb;
		c.be_tgt = w.be_tgt + bb;
		c.o_next = w.o_next + bb;
		c.i_next = w.i_next + bb;
		c.b_prev = w.b_prev + bb;
		c.b_next = w.b_next + bb;
		c.b_rsize = w.b_rsize + bb;
		c.b_rclass = w.b_rclass + bb;
		c.be_type = w.be_type + bb;
		c.b_in = w.b_in + bb;
		c.be_cdef = w.be_cdef + bb;
		c.o_head = w.o_head + tb;
		c.o_tail = w.o_tail + tb;
		c.i_head = w.i_head + tb;
		c.i_tail = w.i_tail + tb;
		c.l_head = w.l_head + tb;
		c.l_tail = w.l_tail + tb;
		c.l_size = w.l_size + tb;
		c.bl = w.bl + tb;
		c.nxt = w.nxt + tb;
		c.st_head = w.st_head + tb;
		c.st_tail = w.st_tail + tb;
		c.s_vtx = w.s_vtx + w.voff[ci];
		c.s_cls = w.s_cls + w.voff[ci];
		c.next_seen = w.next_seen + w.voff[ci];
		c.last = w.last + bb + tb;
		c.in_s = w.in_s + bb + tb;
		c.p_parent = w.p_parent + pb;
		c.p_a = w.p_a + pb;
		c.p_z = w.p_z + pb;
		c.p_or = w.p_or + pb;
		c.p_ai = w.p_ai ? w.p_ai + pb : nullptr;
		c.p_zi = w.p_zi ? w.p_zi + pb : nullptr;
		c.aux = w.aux + pb;
		c.hairpins = w.hairpins ? w.hairpins + 2 * pb : nullptr;

		if (w.stages & SEQ_STAGE_TREE) {
			seq_spanning_tree(c, w.start_key[ci]);
			w.c_ntree[ci] = c.N;
			w.c_nbe0[ci] = c.n_be;
			if (c.N == 0) {
				w.c_status[ci] = 2;
				w.c_npvst[ci] = 0;
				continue;
			}
		} else {
			c.N = w.c_ntree[ci];
			c.n_be = w.c_nbe0[ci];
		}
		if (w.stages & SEQ_STAGE_CLASSES) {
			seq_cycle_classes(c, w.hairpins != nullptr);
			w.c_nbe[ci] = c.n_be;
			w.c_nclass[ci] = c.n_class;
			w.c_nbry[ci] = c.n_bry;
		} else {
			c.n_class = w.c_nclass[ci];
		}
		uint32_t n = w.c_nstack[ci];
		if (w.stages & SEQ_STAGE_STACK) {
			n = seq_candidate_stack(c);
			w.c_nstack[ci] = n;
		}
		if (w.stages & SEQ_STAGE_PVST)
			w.c_npvst[ci] = seq_pvst(c, n, (w.stages & SEQ_STAGE_GIVEN_STACK) != 0);
		w.c_status[ci] = 1;
	}
}

// per-component counters, plus (parallel stages) the flagged-component marks and the 16 error words
__global__ void k_zero_counters(uint32_t n, uint32_t *a, uint32_t *b, uint32_t *c, uint32_t *d, uint32_t *e, uint32_t *f,
				uint32_t *g, uint32_t *h, uint32_t *bad, uint32_t *err)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		a[i] = b[i] = c[i] = d[i] = e[i] = f[i] = g[i] = h[i] = 0;
		if (bad)
			bad[i] = 0;
	}
	if (err && i < 16)
		err[i] = 0;
}

void zero_component_counters(const SeqWs &ws, uint32_t C, uint32_t *comp_bad, uint32_t *err, hipStream_t s)
{
	KLAUNCH(k_zero_counters, dim3((C + 256) / 256), dim3(256), 0, s, C + 1, ws.c_ntree, ws.c_nbe0, ws.c_nbe,
			   ws.c_nstack, ws.c_npvst, ws.c_nclass, ws.c_nbry, ws.c_status, comp_bad, err);
}

void launch_seq_components(const SeqWs &ws, hipStream_t s)
{
	if (ws.C == 0)
		return;
	unsigned grid = (unsigned)std::min<uint64_t>(ws.C, 2048);
	KLAUNCH(k_seq_components, dim3(grid), dim3(64), 0, s, ws);
}

} // namespace povu_hip
