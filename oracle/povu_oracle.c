/*
 * povu_oracle.c -- TEST INFRASTRUCTURE ONLY (see povu_oracle.h).
 *
 * CPU restatement of the reference `povu decompose` hot path.  Array based,
 * single threaded, linear time; semantics (orderings, quirks) follow the
 * reference line by line where it matters -- citations are file:line in the
 * reference checkout.
 */
#define _POSIX_C_SOURCE 200809L
#include "povu_oracle.h"

#include <errno.h>
#include <inttypes.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define NIL ORC_NIL

static void *xcalloc(size_t n, size_t sz)
{
	void *p = calloc(n ? n : 1, sz);
	if (!p) {
		fprintf(stderr, "povu_oracle: out of memory\n");
		abort();
	}
	return p;
}
static void *xmalloc(size_t n)
{
	void *p = malloc(n ? n : 1);
	if (!p) {
		fprintf(stderr, "povu_oracle: out of memory\n");
		abort();
	}
	return p;
}
static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* when non-zero, from_bd rescans a side's edge list from the start after
 * every return to it, literally as spanning_tree.cpp:442-447 does; the default
 * resumes after the last tree child, which creates the same tree and the same
 * back edges in the same order (tests/test_oracle.py checks both). */
static int g_faithful_rescan = 0;
static int g_leaf_sub = 0;
void orc_set_leaf_subflubbles(int on)
{
	g_leaf_sub = on;
}
void orc_set_faithful_rescan(int on)
{
	g_faithful_rescan = on;
}

/* ------------------------------------------------------------------ graph */

orc_graph *orc_graph_new(uint32_t nv, uint32_t ne)
{
	orc_graph *g = xcalloc(1, sizeof *g);
	g->nv = nv;
	g->ne = ne;
	g->vid = xcalloc(nv, 4);
	g->ev1 = xcalloc(ne, 4);
	g->ev2 = xcalloc(ne, 4);
	g->es1 = xcalloc(ne, 1);
	g->es2 = xcalloc(ne, 1);
	g->tip = xcalloc(nv, 1);
	return g;
}

void orc_graph_free(orc_graph *g)
{
	if (!g)
		return;
	free(g->vid);
	free(g->ev1);
	free(g->ev2);
	free(g->es1);
	free(g->es2);
	free(g->tip);
	free(g->off);
	free(g->adj);
	free(g->gidx);
	free(g);
}

/* VG::add_edge, bidirected.cpp:317-335: the edge idx goes into the std::set of
 * each endpoint side; a same-side self loop is therefore listed once. */
void orc_graph_build_csr(orc_graph *g)
{
	const uint32_t ns = 2 * g->nv;
	free(g->off);
	free(g->adj);
	g->off = xcalloc((size_t)ns + 1, 4);
	for (uint32_t e = 0; e < g->ne; e++) {
		uint32_t a = 2 * g->ev1[e] + g->es1[e], b = 2 * g->ev2[e] + g->es2[e];
		g->off[a + 1]++;
		if (b != a)
			g->off[b + 1]++;
	}
	for (uint32_t s = 0; s < ns; s++)
		g->off[s + 1] += g->off[s];
	g->adj = xcalloc(g->off[ns], 4);
	uint32_t *cur = xmalloc((size_t)(ns + 1) * 4);
	memcpy(cur, g->off, (size_t)(ns + 1) * 4);
	for (uint32_t e = 0; e < g->ne; e++) {
		uint32_t a = 2 * g->ev1[e] + g->es1[e], b = 2 * g->ev2[e] + g->es2[e];
		g->adj[cur[a]++] = e;
		if (b != a)
			g->adj[cur[b]++] = e;
	}
	free(cur);
}

/* src/mto/from_gfa.cpp:262-277 */
void orc_graph_infer_tips(orc_graph *g)
{
	for (uint32_t v = 0; v < g->nv; v++) {
		int le = g->off[2 * v + 1] == g->off[2 * v];
		int re = g->off[2 * v + 2] == g->off[2 * v + 1];
		if (le && re)
			g->tip[v] = ORC_TIP_L;
		else if (le)
			g->tip[v] = ORC_TIP_L;
		else if (re)
			g->tip[v] = ORC_TIP_R;
		else
			g->tip[v] = ORC_TIP_NONE;
	}
}

/* ------------------------------------------------------------------ GFA
 * Row A of the scope table.  liteseq is not available, so the loader contract
 * is the one DESIGN.md defines: vertices ascending by numeric segment id,
 * edges in L-line order, `+` on the source = right side, `+` on the sink = left
 * side (src/mto/from_gfa.cpp:223-243, inverse writer src/mto/to_gfa.cpp:24-33).
 * Validation follows validate_gfa_for_liteseq, from_gfa.cpp:57-98. */

typedef struct {
	uint32_t id;
	uint32_t pos;
} idpos;
static int cmp_idpos(const void *a, const void *b)
{
	const idpos *x = a, *y = b;
	if (x->id != y->id)
		return x->id < y->id ? -1 : 1;
	return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

static int parse_u32(const char *s, const char *e, uint32_t *out)
{
	if (s == e)
		return 0;
	uint64_t v = 0;
	for (const char *p = s; p < e; p++) {
		if (*p < '0' || *p > '9')
			return 0;
		v = v * 10 + (uint64_t)(*p - '0');
		if (v > 0xFFFFFFFEull)
			return 0;
	}
	*out = (uint32_t)v;
	return 1;
}

orc_graph *orc_graph_from_gfa(const char *path, char *err, size_t errlen)
{
	FILE *f = fopen(path, "rb");
	if (!f) {
		snprintf(err, errlen, "Invalid GFA '%s': could not open file", path);
		return NULL;
	}
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	char *buf = xmalloc((size_t)sz + 1);
	if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) {
		fclose(f);
		free(buf);
		snprintf(err, errlen, "Invalid GFA '%s': could not read file", path);
		return NULL;
	}
	fclose(f);
	buf[sz] = 0;

	size_t ns = 0, nl = 0, cap_s = 1024, cap_l = 1024;
	idpos *segs = xmalloc(cap_s * sizeof *segs);
	uint32_t *l1 = xmalloc(cap_l * 4), *l2 = xmalloc(cap_l * 4);
	uint8_t *s1 = xmalloc(cap_l), *s2 = xmalloc(cap_l);
	size_t line_no = 1;
	char *p = buf, *end = buf + sz;
	orc_graph *g = NULL;
	while (p < end) {
		char *nlp = memchr(p, '\n', (size_t)(end - p));
		char *le = nlp ? nlp : end;
		char *next = nlp ? nlp + 1 : end;
		if (le > p && le[-1] == '\r')
			le--;
		if (le == p) {
			line_no++;
			p = next;
			continue;
		}
		char *fld[6];
		char *fe[6];
		int nf = 0;
		char *q = p;
		while (nf < 6) {
			fld[nf] = q;
			char *t = memchr(q, '\t', (size_t)(le - q));
			fe[nf] = t ? t : le;
			nf++;
			if (!t)
				break;
			q = t + 1;
		}
		switch (*p) {
		case 'H':
		case 'P':
		case 'W':
			break;
		case 'S': {
			if (nf < 2) {
				snprintf(err, errlen,
					 "Invalid GFA '%s': S record on line %zu is missing a segment id "
					 "and sequence",
					 path, line_no);
				goto fail;
			}
			if (nf < 3) {
				snprintf(err, errlen,
					 "Invalid GFA '%s': S record on line %zu is missing a sequence",
					 path, line_no);
				goto fail;
			}
			if (fe[2] == fld[2]) {
				snprintf(err, errlen,
					 "Invalid GFA '%s': S record on line %zu has an empty sequence",
					 path, line_no);
				goto fail;
			}
			uint32_t id;
			if (!parse_u32(fld[1], fe[1], &id)) {
				snprintf(err, errlen,
					 "Invalid GFA '%s': S record on line %zu has a non-numeric "
					 "segment id",
					 path, line_no);
				goto fail;
			}
			if (ns == cap_s) {
				cap_s *= 2;
				segs = realloc(segs, cap_s * sizeof *segs);
			}
			segs[ns].id = id;
			segs[ns].pos = (uint32_t)ns;
			ns++;
			break;
		}
		case 'L': {
			uint32_t a, b;
			if (nf < 5 || !parse_u32(fld[1], fe[1], &a) || !parse_u32(fld[3], fe[3], &b) ||
			    fe[2] - fld[2] != 1 || fe[4] - fld[4] != 1 ||
			    (*fld[2] != '+' && *fld[2] != '-') || (*fld[4] != '+' && *fld[4] != '-')) {
				snprintf(err, errlen,
					 "Invalid GFA '%s': malformed L record on line %zu", path,
					 line_no);
				goto fail;
			}
			if (nl == cap_l) {
				cap_l *= 2;
				l1 = realloc(l1, cap_l * 4);
				l2 = realloc(l2, cap_l * 4);
				s1 = realloc(s1, cap_l);
				s2 = realloc(s2, cap_l);
			}
			l1[nl] = a;
			l2[nl] = b;
			s1[nl] = (*fld[2] == '+') ? ORC_R : ORC_L;
			s2[nl] = (*fld[4] == '+') ? ORC_L : ORC_R;
			nl++;
			break;
		}
		default:
			snprintf(err, errlen, "Invalid GFA '%s': unsupported record type '%c' on line %zu",
				 path, *p, line_no);
			goto fail;
		}
		line_no++;
		p = next;
	}
	if (ns == 0) {
		snprintf(err, errlen, "Invalid GFA '%s': liteseq returned no vertices", path);
		goto fail;
	}
	qsort(segs, ns, sizeof *segs, cmp_idpos);
	/* duplicate S ids: keep the first (one vertex slot per id) */
	size_t nu = 0;
	for (size_t i = 0; i < ns; i++)
		if (i == 0 || segs[i].id != segs[i - 1].id)
			segs[nu++] = segs[i];
	g = orc_graph_new((uint32_t)nu, (uint32_t)nl);
	for (size_t i = 0; i < nu; i++)
		g->vid[i] = segs[i].id;
	for (size_t e = 0; e < nl; e++) {
		uint32_t ids[2] = {l1[e], l2[e]}, idx[2];
		for (int k = 0; k < 2; k++) {
			size_t lo = 0, hi = nu;
			while (lo < hi) {
				size_t mid = (lo + hi) / 2;
				if (g->vid[mid] < ids[k])
					lo = mid + 1;
				else
					hi = mid;
			}
			if (lo == nu || g->vid[lo] != ids[k]) {
				snprintf(err, errlen,
					 "Invalid GFA '%s': L record %zu references unknown segment %u",
					 path, e, ids[k]);
				orc_graph_free(g);
				g = NULL;
				goto fail;
			}
			idx[k] = (uint32_t)lo;
		}
		g->ev1[e] = idx[0];
		g->es1[e] = s1[e];
		g->ev2[e] = idx[1];
		g->es2[e] = s2[e];
	}
	orc_graph_build_csr(g);
	orc_graph_infer_tips(g);
fail:
	free(segs);
	free(l1);
	free(l2);
	free(s1);
	free(s2);
	free(buf);
	return g;
}

/* ------------------------------------------------------------ componetize
 * bidirected.cpp:477-602.  Components are discovered from the lowest
 * unvisited vertex idx (:585-596), so they are ordered by their minimum
 * vertex idx; membership is plain connectivity (:497-510). */

orc_comp_map *orc_comp_map_of(const orc_graph *g)
{
	orc_comp_map *m = xcalloc(1, sizeof *m);
	m->comp_of = xmalloc((size_t)g->nv * 4);
	m->local_idx = xmalloc((size_t)g->nv * 4);
	memset(m->comp_of, 0xFF, (size_t)g->nv * 4);
	uint32_t *stack = xmalloc((size_t)g->nv * 4);
	uint32_t nc = 0;
	for (uint32_t s = 0; s < g->nv; s++) {
		if (m->comp_of[s] != NIL)
			continue;
		uint32_t sp = 0;
		stack[sp++] = s;
		m->comp_of[s] = nc;
		while (sp) {
			uint32_t v = stack[--sp];
			for (uint32_t k = g->off[2 * v]; k < g->off[2 * v + 2]; k++) {
				uint32_t e = g->adj[k];
				/* Edge::get_other_vtx(v_idx), bidirected.cpp:60-64 */
				uint32_t o = (g->ev1[e] == v) ? g->ev2[e] : g->ev1[e];
				if (m->comp_of[o] == NIL) {
					m->comp_of[o] = nc;
					stack[sp++] = o;
				}
			}
		}
		nc++;
	}
	free(stack);
	m->n_comp = nc;
	uint32_t *cnt = xcalloc(nc, 4);
	for (uint32_t v = 0; v < g->nv; v++)
		m->local_idx[v] = cnt[m->comp_of[v]]++;
	free(cnt);
	return m;
}

void orc_comp_map_free(orc_comp_map *m)
{
	if (!m)
		return;
	free(m->comp_of);
	free(m->local_idx);
	free(m);
}

uint32_t orc_componetize(const orc_graph *g, orc_graph ***out)
{
	orc_comp_map *m = orc_comp_map_of(g);
	uint32_t nc = m->n_comp;
	uint32_t *cnv = xcalloc(nc, 4), *cne = xcalloc(nc, 4);
	for (uint32_t v = 0; v < g->nv; v++)
		cnv[m->comp_of[v]]++;
	for (uint32_t e = 0; e < g->ne; e++)
		cne[m->comp_of[g->ev1[e]]]++;
	orc_graph **cs = xcalloc(nc, sizeof *cs);
	for (uint32_t c = 0; c < nc; c++) {
		cs[c] = orc_graph_new(cnv[c], cne[c]);
		cs[c]->gidx = xmalloc((size_t)cnv[c] * 4);
	}
	/* vertices re-added in ascending global idx (comp_vtxs is a std::set, :552-555) */
	for (uint32_t v = 0; v < g->nv; v++) {
		orc_graph *c = cs[m->comp_of[v]];
		uint32_t lv = m->local_idx[v];
		c->vid[lv] = g->vid[v];
		c->gidx[lv] = v;
		c->tip[lv] = g->tip[v]; /* tips by membership, :572-577 */
	}
	/* edges: vertices ascending, e_l then e_r ascending, first encounter wins
	 * (:558-569); the edge is stored from the encountering side; self loops
	 * become (ve, complement(ve)) via Edge::get_other_vtx(v_idx, ve), :66-77 */
	uint8_t *added = xcalloc(g->ne, 1);
	uint32_t *ecur = xcalloc(nc, 4);
	for (uint32_t v = 0; v < g->nv; v++) {
		uint32_t ci = m->comp_of[v];
		orc_graph *c = cs[ci];
		for (uint32_t ve = 0; ve < 2; ve++) {
			for (uint32_t k = g->off[2 * v + ve]; k < g->off[2 * v + ve + 1]; k++) {
				uint32_t e = g->adj[k];
				if (added[e])
					continue;
				added[e] = 1;
				uint32_t ov, os;
				if (g->ev1[e] == g->ev2[e]) {
					ov = g->ev1[e];
					os = 1 - ve;
				} else if (g->ev1[e] == v) {
					ov = g->ev2[e];
					os = g->es2[e];
				} else {
					ov = g->ev1[e];
					os = g->es1[e];
				}
				uint32_t le = ecur[ci]++;
				c->ev1[le] = m->local_idx[v];
				c->es1[le] = (uint8_t)ve;
				c->ev2[le] = m->local_idx[ov];
				c->es2[le] = (uint8_t)os;
			}
		}
	}
	for (uint32_t c = 0; c < nc; c++)
		orc_graph_build_csr(cs[c]);
	free(added);
	free(ecur);
	free(cnv);
	free(cne);
	orc_comp_map_free(m);
	*out = cs;
	return nc;
}

/* ------------------------------------------------------------------ tree */

typedef struct {
	uint64_t *slot;
	uint64_t mask;
} pairset;
#define PS_EMPTY 0xFFFFFFFFFFFFFFFFull
static void ps_init(pairset *s, uint64_t expect)
{
	uint64_t cap = 16;
	while (cap < 2 * expect + 16)
		cap <<= 1;
	s->slot = xmalloc(cap * 8);
	memset(s->slot, 0xFF, cap * 8);
	s->mask = cap - 1;
}
static uint64_t ps_key(uint32_t a, uint32_t b)
{
	uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
	return ((uint64_t)lo << 32) | hi;
}
static uint64_t ps_hash(uint64_t k)
{
	k ^= k >> 33;
	k *= 0xff51afd7ed558ccdull;
	k ^= k >> 33;
	k *= 0xc4ceb9fe1a85ec53ull;
	k ^= k >> 33;
	return k;
}
static int ps_has(const pairset *s, uint32_t a, uint32_t b)
{
	uint64_t k = ps_key(a, b), h = ps_hash(k) & s->mask;
	while (s->slot[h] != PS_EMPTY) {
		if (s->slot[h] == k)
			return 1;
		h = (h + 1) & s->mask;
	}
	return 0;
}
static void ps_add(pairset *s, uint32_t a, uint32_t b)
{
	uint64_t k = ps_key(a, b), h = ps_hash(k) & s->mask;
	while (s->slot[h] != PS_EMPTY) {
		if (s->slot[h] == k)
			return;
		h = (h + 1) & s->mask;
	}
	s->slot[h] = k;
}

static uint32_t tree_add_be(orc_tree *t, uint32_t src, uint32_t tgt, uint8_t type, uint32_t n_tree_edges)
{
	/* Tree::add_be, spanning_tree.cpp:795-805: id from the shared counter */
	if (t->n_be == t->be_cap) {
		fprintf(stderr, "povu_oracle: back edge capacity exceeded\n");
		abort();
	}
	uint32_t b = t->n_be++;
	t->be_src[b] = src;
	t->be_tgt[b] = tgt;
	t->be_type[b] = type;
	t->be_id[b] = b + n_tree_edges;
	return b;
}

void orc_tree_free(orc_tree *t)
{
	if (!t)
		return;
	free(t->gid);
	free(t->typ);
	free(t->par);
	free(t->pe_id);
	free(t->pe_black);
	free(t->cls);
	free(t->hi);
	free(t->pre);
	free(t->post);
	free(t->be_src);
	free(t->be_tgt);
	free(t->be_id);
	free(t->be_type);
	free(t->bry);
	free(t);
}

/* pst::Tree::from_bd, spanning_tree.cpp:262-463 */
orc_tree *orc_from_bd(const orc_graph *g)
{
	int has_tips = 0;
	for (uint32_t v = 0; v < g->nv; v++)
		if (g->tip[v]) {
			has_tips = 1;
			break;
		}
	const uint32_t N = has_tips ? 2 * g->nv + 1 : 2 * g->nv; /* :286-288 */
	orc_tree *t = xcalloc(1, sizeof *t);
	t->n = N;
	t->gid = xmalloc((size_t)N * 4);
	t->typ = xmalloc(N);
	t->par = xmalloc((size_t)N * 4);
	t->pe_id = xmalloc((size_t)N * 4);
	t->pe_black = xcalloc(N, 1);
	t->cls = xmalloc((size_t)N * 4);
	t->hi = xmalloc((size_t)N * 4);
	t->pre = xcalloc(N, 4);
	t->post = xcalloc(N, 4);
	memset(t->cls, 0xFF, (size_t)N * 4);
	memset(t->hi, 0xFF, (size_t)N * 4);
	/* original back edges <= ne + nv (tips); the class pass adds <= 2 per tree vertex */
	t->be_cap = g->ne + g->nv + 2 * N + 8;
	t->be_src = xmalloc((size_t)t->be_cap * 4);
	t->be_tgt = xmalloc((size_t)t->be_cap * 4);
	t->be_id = xmalloc((size_t)t->be_cap * 4);
	t->be_type = xmalloc(t->be_cap);

	uint8_t *visited = xcalloc(g->nv, 1);
	uint8_t *self_loop = xcalloc(g->nv, 1);
	uint32_t *ctr = xcalloc((size_t)2 * g->nv + 1, 4); /* be_idx_to_ctr */
	uint32_t *stack = xmalloc((size_t)(2 * g->nv + 2) * 4);
	uint32_t *cursor = xmalloc((size_t)2 * g->nv * 4);
	for (uint32_t s = 0; s < 2 * g->nv; s++)
		cursor[s] = g->off[s];
	pairset conn;
	ps_init(&conn, (uint64_t)N + g->ne + g->nv + 8);

	uint32_t sp = 0, counter = 0, order = 0, n_te = 0;
	uint32_t p_idx = NIL;
	const uint32_t root_idx = 0;

#define ADD_VERTEX_TO_TREE(e_, v_)                                                               \
	do { /* add_vertex_to_tree, :324-356 */                                                  \
		uint32_t e__ = (e_), v__ = (v_);                                                 \
		uint32_t a__ = counter++, b__ = counter++;                                       \
		t->gid[a__] = g->vid[v__];                                                       \
		t->typ[a__] = (uint8_t)e__;                                                      \
		t->pre[a__] = order++;                                                           \
		t->gid[b__] = g->vid[v__];                                                       \
		t->typ[b__] = (uint8_t)(1 - e__);                                                \
		t->pre[b__] = order++;                                                           \
		ctr[2 * v__ + e__] = a__;                                                        \
		ctr[2 * v__ + (1 - e__)] = b__;                                                  \
		t->par[a__] = p_idx;                                                             \
		t->pe_id[a__] = NIL;                                                             \
		if (p_idx != NIL) {                                                              \
			t->pe_id[a__] = n_te + t->n_be; /* add_tree_edge, :784-793 */            \
			n_te++;                                                                  \
			ps_add(&conn, p_idx, a__);                                               \
		}                                                                                \
		t->par[b__] = a__;                                                               \
		t->pe_id[b__] = n_te + t->n_be;                                                  \
		t->pe_black[b__] = 1;                                                            \
		n_te++;                                                                          \
		ps_add(&conn, a__, b__);                                                         \
	} while (0)

	if (has_tips) { /* dummy root, :397-402 */
		p_idx = counter;
		t->gid[counter] = NIL;
		t->typ[counter] = ORC_DUMMY;
		t->par[counter] = NIL;
		t->pe_id[counter] = NIL;
		t->pre[counter] = order++;
		counter++;
	}
	/* start = *tips().begin(): smallest (id, then l<r), types.cpp:60-68; else (l, idx 0) */
	uint32_t s_end = ORC_L, s_idx = 0;
	if (has_tips) {
		uint32_t best = NIL;
		for (uint32_t v = 0; v < g->nv; v++)
			if (g->tip[v] && (best == NIL || g->vid[v] < g->vid[best]))
				best = v;
		s_idx = best;
		s_end = g->tip[best] == ORC_TIP_L ? ORC_L : ORC_R;
	}
	stack[sp++] = 2 * s_idx + s_end;
	stack[sp++] = 2 * s_idx + (1 - s_end);
	visited[s_idx] = 1;
	ADD_VERTEX_TO_TREE(s_end, s_idx);

	while (sp) { /* main loop, :419-453 */
		uint32_t be_v = stack[sp - 1];
		p_idx = ctr[be_v];
		uint32_t bd_v = be_v >> 1, syd = be_v & 1;
		uint32_t lo = g->off[be_v], hi = g->off[be_v + 1];
		if (lo == hi && !ps_has(&conn, p_idx, root_idx)) { /* tip -> root, :433-438 */
			tree_add_be(t, p_idx, root_idx, ORC_BE_BACK, n_te);
			ps_add(&conn, p_idx, root_idx);
		}
		int found = 0;
		uint32_t k = g_faithful_rescan ? lo : cursor[be_v];
		for (; k < hi; k++) {
			uint32_t e = g->adj[k];
			/* process_edge, :360-398; Edge::get_other_vtx(v_idx, ve), bidirected.cpp:66-77 */
			uint32_t ov, os;
			if (g->ev1[e] == g->ev2[e]) {
				ov = g->ev1[e];
				os = 1 - syd;
			} else if (g->ev1[e] == bd_v) {
				ov = g->ev2[e];
				os = g->es2[e];
			} else {
				ov = g->ev1[e];
				os = g->es1[e];
			}
			uint32_t o_be = 2 * ov + os;
			if (!visited[ov]) {
				ADD_VERTEX_TO_TREE(os, ov);
				visited[ov] = 1;
				stack[sp++] = o_be;
				stack[sp++] = o_be ^ 1;
				found = 1;
				k++;
				break;
			} else if (!ps_has(&conn, p_idx, ctr[o_be])) {
				tree_add_be(t, p_idx, ctr[o_be], ORC_BE_BACK, n_te);
				ps_add(&conn, p_idx, ctr[o_be]);
			} else if (bd_v == ov && !self_loop[bd_v]) {
				tree_add_be(t, p_idx, ctr[o_be], ORC_BE_BACK, n_te);
				self_loop[bd_v] = 1;
			}
		}
		cursor[be_v] = k;
		if (!found) {
			t->post[p_idx] = order++;
			sp--;
		}
	}
	if (has_tips)
		t->post[0] = order++;
#undef ADD_VERTEX_TO_TREE
	t->n_be0 = t->n_be;
	if (counter != N) {
		fprintf(stderr, "povu_oracle: from_bd visited %u of %u tree vertices\n", counter, N);
		abort();
	}
	free(visited);
	free(self_loop);
	free(ctr);
	free(stack);
	free(cursor);
	free(conn.slot);
	return t;
}

/* -------------------------------------------------------- cycle equivalence
 * simple_cycle_equiv + handle_vertex, flubbles.cpp:503-719, over
 * WBracketList (bracket_list.cpp:61-100) and Tree::{concat_bracket_lists,
 * del_bracket,push,top,new_class} (spanning_tree.cpp:821-894). */

#define U64MAX 0xFFFFFFFFFFFFFFFFull

void orc_cycle_equiv(orc_tree *t)
{
	const uint32_t N = t->n;
	const uint32_t n_te = N - 1; /* every non-root vertex has a parent tree edge */
	/* children, ascending idx (std::set, spanning_tree.cpp:662-669) */
	uint32_t *c_off = xcalloc((size_t)N + 1, 4), *c_adj = xmalloc((size_t)N * 4);
	for (uint32_t v = 1; v < N; v++)
		c_off[t->par[v] + 1]++;
	for (uint32_t v = 0; v < N; v++)
		c_off[v + 1] += c_off[v];
	{
		uint32_t *cur = xmalloc((size_t)(N + 1) * 4);
		memcpy(cur, c_off, (size_t)(N + 1) * 4);
		for (uint32_t v = 1; v < N; v++)
			c_adj[cur[t->par[v]]++] = v;
		free(cur);
	}
	/* outgoing back edges of from_bd, ascending idx (std::set<idx>) */
	uint32_t *o_off = xcalloc((size_t)N + 1, 4), *o_adj = xmalloc((size_t)(t->n_be0 + 1) * 4);
	for (uint32_t b = 0; b < t->n_be0; b++)
		o_off[t->be_src[b] + 1]++;
	for (uint32_t v = 0; v < N; v++)
		o_off[v + 1] += o_off[v];
	{
		uint32_t *cur = xmalloc((size_t)(N + 1) * 4);
		memcpy(cur, o_off, (size_t)(N + 1) * 4);
		for (uint32_t b = 0; b < t->n_be0; b++)
			o_adj[cur[t->be_src[b]]++] = b;
		free(cur);
	}
	/* incoming back edges: grows during the pass (capping / simplifying), ascending idx */
	const uint32_t cap = t->be_cap;
	uint32_t *i_head = xmalloc((size_t)N * 4), *i_tail = xmalloc((size_t)N * 4);
	uint32_t *i_next = xmalloc((size_t)cap * 4);
	memset(i_head, 0xFF, (size_t)N * 4);
	memset(i_tail, 0xFF, (size_t)N * 4);
#define IBE_APPEND(v_, b_)                                                                       \
	do {                                                                                     \
		uint32_t v__ = (v_), b__ = (b_);                                                 \
		i_next[b__] = NIL;                                                               \
		if (i_head[v__] == NIL)                                                          \
			i_head[v__] = b__;                                                       \
		else                                                                             \
			i_next[i_tail[v__]] = b__;                                               \
		i_tail[v__] = b__;                                                               \
	} while (0)
	for (uint32_t b = 0; b < t->n_be0; b++)
		IBE_APPEND(t->be_tgt[b], b);

	/* brackets: one node per back edge idx; lists are identified by the vertex that created them */
	uint32_t *b_prev = xmalloc((size_t)cap * 4), *b_next = xmalloc((size_t)cap * 4);
	uint8_t *b_in = xcalloc(cap, 1);
	uint64_t *b_rsize = xmalloc((size_t)cap * 8), *b_rclass = xmalloc((size_t)cap * 8);
	uint8_t *be_cls_def = xcalloc(cap, 1);
	uint32_t *l_head = xmalloc((size_t)N * 4), *l_tail = xmalloc((size_t)N * 4);
	uint64_t *l_size = xcalloc(N, 8);
	uint32_t *bl = xmalloc((size_t)N * 4); /* vertex -> list id or NIL (nullptr) */
	memset(bl, 0xFF, (size_t)N * 4);

	uint32_t n_class = 0;
	int in_hairpin = 0;
	uint64_t bry_b1 = NIL, bry_b2 = NIL;
	uint32_t bry_cap = 16;
	t->bry = xmalloc((size_t)bry_cap * 16);
	t->n_bry = 0;

#define BL_PUSH(v_, b_)                                                                          \
	do { /* Tree::push, spanning_tree.cpp:855-869 ; WBracketList::push, bracket_list.cpp:61-65 */ \
		uint32_t v__ = (v_), b__ = (b_);                                                 \
		if (bl[v__] == NIL) {                                                            \
			bl[v__] = v__;                                                           \
			l_head[v__] = l_tail[v__] = NIL;                                         \
			l_size[v__] = 0;                                                         \
		}                                                                                \
		uint32_t L__ = bl[v__];                                                          \
		b_rsize[b__] = U64MAX;                                                           \
		b_rclass[b__] = U64MAX;                                                          \
		b_prev[b__] = NIL;                                                               \
		b_next[b__] = l_head[L__];                                                       \
		if (l_head[L__] != NIL)                                                          \
			b_prev[l_head[L__]] = b__;                                               \
		else                                                                             \
			l_tail[L__] = b__;                                                       \
		l_head[L__] = b__;                                                               \
		l_size[L__]++;                                                                   \
		b_in[b__] = 1;                                                                   \
	} while (0)

	for (uint32_t v = N; v-- > 0;) {
		/* hi_0, :515-519 */
		uint32_t hi_0 = NIL;
		for (uint32_t k = o_off[v]; k < o_off[v + 1]; k++) {
			uint32_t tg = t->be_tgt[o_adj[k]];
			if (tg < hi_0)
				hi_0 = tg;
		}
		const uint32_t cb = c_off[v], ce = c_off[v + 1];
		const int is_leaf = cb == ce, is_root = t->par[v] == NIL;
		/* :531-535 */
		if (in_hairpin && ((is_leaf && !is_root) || is_root)) {
			if (t->n_bry == bry_cap) {
				bry_cap *= 2;
				t->bry = realloc(t->bry, (size_t)bry_cap * 16);
			}
			t->bry[2 * t->n_bry] = bry_b1;
			t->bry[2 * t->n_bry + 1] = bry_b2;
			t->n_bry++;
			bry_b1 = bry_b2 = NIL;
			in_hairpin = 0;
		}
		/* hi_1 = smallest child hi, :540-550 */
		uint32_t hi_1 = NIL;
		for (uint32_t k = cb; k < ce; k++)
			if (k == cb || t->hi[c_adj[k]] < hi_1)
				hi_1 = t->hi[c_adj[k]];
		t->hi[v] = hi_0 < hi_1 ? hi_0 : hi_1; /* :552 */
		/* hi_child = first child (ascending) whose hi == hi_1, :555-561 */
		uint32_t hi_child = NIL;
		for (uint32_t k = cb; k < ce; k++)
			if (t->hi[c_adj[k]] == hi_1) {
				hi_child = c_adj[k];
				break;
			}
		/* hi_2 = hi of the first child != hi_child with hi < v, :566-574
		 * (articulated_vertices is never filled, :699) */
		uint64_t hi_2 = NIL;
		for (uint32_t k = cb; k < ce; k++) {
			uint32_t c = c_adj[k];
			if (c != hi_child && t->hi[c] < v) {
				hi_2 = t->hi[c];
				break;
			}
		}
		/* concat children, ascending: first child adopted, later ones spliced to the
		 * front (:586-588, spanning_tree.cpp:821-836, bracket_list.cpp:90-93) */
		for (uint32_t k = cb; k < ce; k++) {
			uint32_t c = c_adj[k];
			uint32_t Lc = bl[c];
			if (bl[v] == NIL) {
				bl[v] = Lc;
				bl[c] = NIL;
			} else if (Lc != NIL) {
				uint32_t Lp = bl[v];
				if (l_head[Lc] != NIL) {
					if (l_head[Lp] != NIL) {
						b_next[l_tail[Lc]] = l_head[Lp];
						b_prev[l_head[Lp]] = l_tail[Lc];
					} else {
						l_tail[Lp] = l_tail[Lc];
					}
					l_head[Lp] = l_head[Lc];
					l_size[Lp] += l_size[Lc];
					l_head[Lc] = l_tail[Lc] = NIL;
					l_size[Lc] = 0;
				}
			}
			/* (a null child list with a non-null parent list dereferences null in the
			 * reference; it cannot occur: every processed vertex ends with a bracket) */
		}
		/* delete incoming, :594-605 */
		for (uint32_t b = i_head[v]; b != NIL; b = i_next[b]) {
			if (bl[v] != NIL && b_in[b]) { /* WBracketList::del, bracket_list.cpp:72-83 */
				uint32_t L = bl[v];
				if (b_prev[b] != NIL)
					b_next[b_prev[b]] = b_next[b];
				else
					l_head[L] = b_next[b];
				if (b_next[b] != NIL)
					b_prev[b_next[b]] = b_prev[b];
				else
					l_tail[L] = b_prev[b];
				l_size[L]--;
				b_in[b] = 0;
			}
			if (t->be_type[b] != ORC_BE_CAPPING && !be_cls_def[b]) {
				be_cls_def[b] = 1;
				n_class++; /* be.set_class(t.new_class()) */
			}
		}
		/* push outgoing, ascending idx, :608-611 */
		for (uint32_t k = o_off[v]; k < o_off[v + 1]; k++)
			BL_PUSH(v, o_adj[k]);
		/* capping, :613-619 */
		if (hi_2 < (uint64_t)hi_0) {
			uint32_t b = tree_add_be(t, v, (uint32_t)hi_2, ORC_BE_CAPPING, n_te);
			IBE_APPEND((uint32_t)hi_2, b);
			BL_PUSH(v, b);
		}
		/* simplifying / hairpin, :621-656 (a null list throws in the reference,
		 * spanning_tree.cpp:871-881; treated as empty here) */
		if (bl[v] == NIL || l_size[bl[v]] == 0) {
			if (t->typ[v] != ORC_DUMMY)
				bry_b1 = t->gid[v];
			uint32_t b = tree_add_be(t, v, 0, ORC_BE_SIMPLIFYING, n_te);
			IBE_APPEND(0, b);
			BL_PUSH(v, b);
			t->hi[v] = 0;
			in_hairpin = 1;
		} else if (in_hairpin) {
			uint32_t b = l_head[bl[v]];
			if (t->be_type[b] == ORC_BE_SIMPLIFYING)
				bry_b2 = t->gid[v];
		}
		/* class of the tree edge (parent(v), v), :664-686 */
		if (!is_root) {
			uint32_t L = bl[v], b = l_head[L];
			if (l_size[L] != b_rsize[b]) {
				b_rsize[b] = l_size[L];
				b_rclass[b] = n_class++;
			}
			t->cls[v] = (uint32_t)b_rclass[b];
			if (b_rsize[b] == 1)
				be_cls_def[b] = 1;
		}
	}
	t->n_class = n_class;
#undef BL_PUSH
#undef IBE_APPEND
	free(c_off);
	free(c_adj);
	free(o_off);
	free(o_adj);
	free(i_head);
	free(i_tail);
	free(i_next);
	free(b_prev);
	free(b_next);
	free(b_in);
	free(b_rsize);
	free(b_rclass);
	free(be_cls_def);
	free(l_head);
	free(l_tail);
	free(l_size);
	free(bl);
}

/* ---------------------------------------------------------- candidate stack
 * ptu::br_desc (tree_utils.cpp:19-155, sort_branches :47-76) +
 * compute_eq_class_stack (flubbles.cpp:412-501).  Vertices are processed
 * N-1 .. 0; every black tree edge pushes one entry to the front of the running
 * list; a child of a branching vertex parks its list in the parent's cache;
 * a branching vertex (or the root) splices the parked lists of its child edges
 * to the front in the order sorted_br = [black edge, gray edges by child idx
 * descending], so the final order is gray children ascending, black child last. */
uint32_t orc_eq_class_stack(const orc_tree *t, orc_oic **out)
{
	const uint32_t N = t->n;
	uint32_t *nchild = xcalloc(N, 4);
	for (uint32_t v = 1; v < N; v++)
		nchild[t->par[v]]++;
	/* children CSR ascending */
	uint32_t *c_off = xcalloc((size_t)N + 1, 4), *c_adj = xmalloc((size_t)N * 4);
	for (uint32_t v = 0; v < N; v++)
		c_off[v + 1] = c_off[v] + nchild[v];
	{
		uint32_t *cur = xmalloc((size_t)(N + 1) * 4);
		memcpy(cur, c_off, (size_t)(N + 1) * 4);
		for (uint32_t v = 1; v < N; v++)
			c_adj[cur[t->par[v]]++] = v;
		free(cur);
	}
	/* entry nodes are keyed by child tree vertex; lists by head/tail */
	uint32_t *nxt = xmalloc((size_t)N * 4);
	uint32_t *st_head = xmalloc((size_t)N * 4), *st_tail = xmalloc((size_t)N * 4); /* parked per child */
	memset(st_head, 0xFF, (size_t)N * 4);
	memset(st_tail, 0xFF, (size_t)N * 4);
	uint32_t m_head = NIL, m_tail = NIL;
	uint32_t *sorted = xmalloc((size_t)N * 4);
	for (uint32_t v = N; v-- > 0;) {
		if (v == 0 || nchild[v] > 1) {
			/* sorted_br: black first, then gray by child idx descending */
			uint32_t ns = 0, black = NIL;
			for (uint32_t k = c_off[v]; k < c_off[v + 1]; k++)
				if (t->pe_black[c_adj[k]])
					black = c_adj[k];
			if (black != NIL)
				sorted[ns++] = black;
			for (uint32_t k = c_off[v + 1]; k-- > c_off[v];)
				if (c_adj[k] != black)
					sorted[ns++] = c_adj[k];
			for (uint32_t i = 0; i < ns; i++) { /* splice parked list to the front */
				uint32_t c = sorted[i];
				if (st_head[c] == NIL)
					continue;
				nxt[st_tail[c]] = m_head;
				if (m_head == NIL)
					m_tail = st_tail[c];
				m_head = st_head[c];
				st_head[c] = st_tail[c] = NIL;
			}
		}
		if (v == 0)
			break;
		if (t->pe_black[v]) { /* push_front, :466-473 */
			nxt[v] = m_head;
			if (m_head == NIL)
				m_tail = v;
			m_head = v;
		}
		if (nchild[t->par[v]] > 1) { /* park, :476-492 */
			st_head[v] = m_head;
			st_tail[v] = m_tail;
			m_head = m_tail = NIL;
		}
	}
	uint32_t n = 0;
	for (uint32_t e = m_head; e != NIL; e = nxt[e])
		n++;
	orc_oic *s = xmalloc((size_t)(n ? n : 1) * sizeof *s);
	uint32_t i = 0;
	for (uint32_t e = m_head; e != NIL; e = nxt[e], i++) {
		s[i].orient = t->typ[e] == ORC_R ? 0 : 1;
		s[i].id = t->gid[e];
		s[i].st_idx = e - 1; /* tree edge idx of vertex e */
		s[i].cls = t->cls[e];
	}
	free(nchild);
	free(c_off);
	free(c_adj);
	free(nxt);
	free(st_head);
	free(st_tail);
	free(sorted);
	*out = s;
	return n;
}

/* compute_eq_class_metadata, flubbles.cpp:375-410 */
void orc_next_seen(const orc_oic *s, uint32_t n, uint32_t n_class, uint32_t *next_seen)
{
	uint32_t *last = xmalloc((size_t)(n_class + 1) * 4);
	memset(last, 0xFF, (size_t)(n_class + 1) * 4);
	for (uint32_t i = n; i-- > 0;) {
		uint32_t c = s[i].cls;
		next_seen[i] = last[c] != NIL ? last[c] : i;
		last[c] = i;
	}
	free(last);
}

void orc_pvst_free(orc_pvst *p)
{
	if (!p)
		return;
	free(p->a_id);
	free(p->z_id);
	free(p->a_or);
	free(p->z_or);
	free(p->parent);
	free(p->ai);
	free(p->zi);
	free(p->fam);
	free(p);
}

static int cmp_u32(const void *a, const void *b)
{
	uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
	return x < y ? -1 : (x > y);
}

/* add_flubbles, flubbles.cpp:295-367 */
orc_pvst *orc_add_flubbles(const orc_tree *t, const orc_oic *st, const uint32_t *next_seen, uint32_t n)
{
	orc_pvst *p = xcalloc(1, sizeof *p);
	uint32_t cap = n + 1;
	p->a_id = xmalloc((size_t)cap * 4);
	p->z_id = xmalloc((size_t)cap * 4);
	p->a_or = xmalloc(cap);
	p->z_or = xmalloc(cap);
	p->parent = xmalloc((size_t)cap * 4);
	p->ai = xmalloc((size_t)cap * 4);
	p->zi = xmalloc((size_t)cap * 4);
	p->n = 1; /* dummy root, flubbles.cpp:736-741 */
	p->a_id[0] = p->z_id[0] = NIL;
	p->a_or[0] = p->z_or[0] = 0;
	p->parent[0] = NIL;
	p->ai[0] = p->zi[0] = NIL;

	uint32_t *s = xmalloc((size_t)(n + 1) * 4);
	uint32_t sp = 0;
	uint8_t *in_s = xcalloc((size_t)t->n_class + 1, 1);
	uint32_t prt = 0;
	for (uint32_t i = 0; i < n; i++) {
		uint32_t cl = st[i].cls;
		if (st[i].id == NIL)
			continue;
		if (in_s[cl]) { /* :326-343 */
			while (sp) {
				uint32_t c = s[--sp];
				in_s[c] = 0;
				if (c == cl)
					break;
			}
			if (prt != 0)
				prt = p->parent[prt];
		}
		if (i + 1 < next_seen[i]) { /* :345-361 */
			const orc_oic *z = &st[next_seen[i]];
			if (z->id == NIL)
				continue;
			/* compute_ai_zi, :264-290 */
			uint32_t vt[4] = {st[i].st_idx + 1, t->par[st[i].st_idx + 1], z->st_idx + 1,
					  t->par[z->st_idx + 1]};
			qsort(vt, 4, 4, cmp_u32);
			uint32_t k = p->n++;
			/* normalize_endpoints, :233-244 */
			if (st[i].orient == 1 && z->orient == 1) {
				p->a_id[k] = z->id;
				p->a_or[k] = 0;
				p->z_id[k] = st[i].id;
				p->z_or[k] = 0;
			} else {
				p->a_id[k] = st[i].id;
				p->a_or[k] = st[i].orient;
				p->z_id[k] = z->id;
				p->z_or[k] = z->orient;
			}
			p->ai[k] = vt[1];
			p->zi[k] = vt[2];
			p->parent[k] = prt;
			prt = k;
		}
		s[sp++] = cl;
		in_s[cl] = 1;
	}
	free(s);
	free(in_s);
	return p;
}

/* find_flubbles, flubbles.cpp:721-745 */
orc_pvst *orc_find_flubbles(orc_tree *t)
{
	orc_cycle_equiv(t);
	orc_oic *st = NULL;
	uint32_t n = orc_eq_class_stack(t, &st);
	uint32_t *ns = xmalloc((size_t)(n + 1) * 4);
	orc_next_seen(st, n, t->n_class, ns);
	orc_pvst *p = orc_add_flubbles(t, st, ns, n);
	free(st);
	free(ns);
	return p;
}

/* ------------------------------------------------- leaf subflubble passes
 * `povu decompose -s` runs five passes after find_flubbles (app/subcommand/decompose.cpp:63-70).  The first two only
 * RELABEL leaf flubbles of the PVST: find_tiny (src/povu/algorithms/tiny.cpp:100-129) and find_parallel
 * (src/povu/algorithms/parallel.cpp:263-287), both over gen_tree_meta's bracket table (src/povu/graph/
 * tree_utils.cpp:531-574 count_brackets, :169-216 collect_backedges_by_vertex, :634-672 pre_process).  They are
 * restated here LITERALLY, accidents included:
 *   - has_be_to_ai (tiny.cpp:39-57) compares the back-edge INDICES of OBE(c) with the vertex index ai;
 *   - trunk (tiny.cpp:76-97) returns false on every path;
 *   - the u32 arithmetic of in_trunk (parallel.cpp:139-181) wraps when zi - ai < 3.
 * Two behaviours of the reference are undefined and are DEFINED here (nothing in the reference pins either):
 *   - in_branch (parallel.cpp:203-225) indexes tm.off and the vertex table with INVALID_IDX when zi has no gray
 *     child: here in_branch answers "no" (in_trunk is still asked);
 *   - a back edge whose source is its target (a self loop on one side, spanning_tree.cpp:387-395) makes
 *     count_brackets subtract one at the vertex without ever adding it and sends the fill loop past the target to the
 *     root, over the ends of the blocks: here such an edge is in nobody's bracket table.
 * The three later passes (find_concealed, find_midi, find_smothered) insert vertices: povu_oracle_sub.inc, included
 * below.  PARITY UNPINNED: no test, fixture or golden file of the reference holds a T or O line. */
/* which rule decided, summed over all calls since the last orc_leaf_stats(reset) -- coverage evidence for the tests */
enum { LS_TINY_NO_Y, LS_TINY_BRACKET, LS_TINY_IDX_ORD, LS_TINY_IDX_EXTRA, LS_PAR_BRANCH_AI, LS_PAR_BRANCH_ZI, LS_PAR_TRUNK_AI,
       LS_PAR_TRUNK_ZI, LS_TRUNK_COND_B, LS_LEAVES, LS_TINY_IDX_ASKED, LS_N };
static uint64_t g_leaf_stats[LS_N];
#define LS_INC(i) ((void)__atomic_fetch_add(&g_leaf_stats[i], 1, __ATOMIC_RELAXED)) /* (the threaded entry points count too) */
void orc_leaf_stats(uint64_t *out, int reset)
{
	if (out)
		memcpy(out, g_leaf_stats, sizeof g_leaf_stats);
	if (reset)
		memset(g_leaf_stats, 0, sizeof g_leaf_stats);
}

typedef struct {
	uint32_t n;
	uint64_t *off;	    /* tm.off: 64-bit here (the table is quadratic on deep trees: 10^10 entries on a graph with large tangles) */
	uint32_t *be;	    /* tm.BE */
	uint32_t *c_off, *c_adj; /* children, ascending vertex idx (std::set order) */
	uint32_t *o_off, *o_adj; /* OBE(v): back-edge idx ascending */
	uint32_t *i_off, *i_adj; /* IBE(v) */
} tree_meta;

static void tree_meta_free(tree_meta *m)
{
	free(m->off);
	free(m->be);
	free(m->c_off);
	free(m->c_adj);
	free(m->o_off);
	free(m->o_adj);
	free(m->i_off);
	free(m->i_adj);
}

static void csr_by(uint32_t n, uint32_t m, const uint32_t *key, uint32_t skip_nil, uint32_t **off_out, uint32_t **adj_out)
{
	uint32_t *off = xcalloc((size_t)n + 2, 4), *adj = xmalloc(((size_t)m + 1) * 4);
	for (uint32_t j = 0; j < m; j++)
		if (!(skip_nil && key[j] == NIL))
			off[key[j] + 1]++;
	for (uint32_t v = 0; v < n; v++)
		off[v + 1] += off[v];
	uint32_t *cur = xmalloc(((size_t)n + 1) * 4);
	memcpy(cur, off, ((size_t)n + 1) * 4);
	for (uint32_t j = 0; j < m; j++)
		if (!(skip_nil && key[j] == NIL))
			adj[cur[key[j]]++] = j;
	free(cur);
	*off_out = off;
	*adj_out = adj;
}

static void tree_meta_build(const orc_tree *t, tree_meta *m)
{
	const uint32_t n = t->n;
	memset(m, 0, sizeof *m);
	m->n = n;
	csr_by(n, n, t->par, 1, &m->c_off, &m->c_adj); /* children of v = the vertices whose parent is v */
	csr_by(n, t->n_be, t->be_src, 0, &m->o_off, &m->o_adj);
	csr_by(n, t->n_be, t->be_tgt, 0, &m->i_off, &m->i_adj);
	/* count_brackets, tree_utils.cpp:531-574 (B = the edges of type back_edge, :640-648) */
	uint32_t *cnt = xcalloc((size_t)n + 1, 4);
	for (uint32_t j = 0; j < t->n_be; j++) {
		if (t->be_type[j] != ORC_BE_BACK || t->be_src[j] == t->be_tgt[j])
			continue;
		uint32_t u = t->be_src[j], w = t->be_tgt[j];
		if (t->par[u] != NIL)
			cnt[t->par[u]] += 1;
		cnt[w] -= 1;
	}
	for (uint32_t v = n; v-- > 1;) /* children before parents: a child's idx is larger than its parent's */
		if (t->par[v] != NIL)
			cnt[t->par[v]] += cnt[v];
	m->off = xcalloc((size_t)n + 2, 8);
	uint64_t total = 0;
	for (uint32_t v = 0; v < n; v++) {
		total += cnt[v];
		m->off[v + 1] = total;
	}
	if (total > (1ull << 36)) { /* 256 GB of table: not on this machine either */
		fprintf(stderr, "povu_oracle: bracket table of %" PRIu64 " entries (tree_utils.cpp:169-216 is quadratic on deep trees)\n", total);
		abort();
	}
	m->be = xmalloc(((size_t)total + 1) * 4);
	/* collect_backedges_by_vertex, tree_utils.cpp:169-216 */
	uint64_t *cursor = xcalloc((size_t)n + 1, 8);
	for (uint32_t j = 0; j < t->n_be; j++) {
		if (t->be_type[j] != ORC_BE_BACK || t->be_src[j] == t->be_tgt[j])
			continue;
		uint32_t u = t->be_src[j], w = t->be_tgt[j];
		if (t->par[u] == NIL)
			continue;
		uint32_t v = t->par[u];
		while (t->par[v] != NIL && v != w) {
			if (cursor[v] >= m->off[v + 1] - m->off[v]) {
				fprintf(stderr, "povu_oracle: bracket table overflow at vertex %u\n", v);
				abort();
			}
			m->be[m->off[v] + cursor[v]++] = j;
			v = t->par[v];
		}
	}
	free(cursor);
	free(cnt);
}

/* branches, tiny.cpp:33-73 (find_Y :18-31) */
static int tiny_branches(const orc_tree *t, const tree_meta *m, uint32_t ai, uint32_t zi)
{
	int any = 0;
	for (uint32_t k = m->c_off[zi]; k < m->c_off[zi + 1]; k++) {
		uint32_t c = m->c_adj[k];
		if (t->pe_black[c])
			continue;
		if (t->post[c] - t->pre[c] != 3) /* has_one_descendants */
			return 0;
		int hit = 0;
		for (uint64_t b = m->off[c]; b < m->off[c + 1] && !hit; b++)
			hit = t->be_tgt[m->be[b]] == ai;
		if (hit)
			LS_INC(LS_TINY_BRACKET);
		else if (m->o_off[c + 1] > m->o_off[c])
			LS_INC(LS_TINY_IDX_ASKED);
		for (uint32_t b = m->o_off[c]; b < m->o_off[c + 1] && !hit; b++) {
			hit = m->o_adj[b] == ai; /* (sic) a back-edge idx against a vertex idx, tiny.cpp:52-56 */
			if (hit)
				LS_INC(t->be_type[m->o_adj[b]] == ORC_BE_BACK ? LS_TINY_IDX_ORD : LS_TINY_IDX_EXTRA);
		}
		if (!hit)
			return 0;
		any = 1;
	}
	if (!any)
		LS_INC(LS_TINY_NO_Y);
	return 1;
}

static uint32_t count_type(const orc_tree *t, const uint32_t *off, const uint32_t *adj, uint32_t v, uint8_t type)
{
	uint32_t k = 0;
	for (uint32_t b = off[v]; b < off[v + 1]; b++)
		k += t->be_type[adj[b]] == type;
	return k;
}

/* inspect_trunk, parallel.cpp:15-108 */
static int par_inspect_trunk(const orc_tree *t, const tree_meta *m, uint32_t ai, uint32_t zi)
{
	uint32_t branching = NIL, n_branching = 0;
	for (uint32_t v = zi; v != ai; v = t->par[v]) {
		if (t->par[v] == NIL) /* (ai is an ancestor of zi for every flubble; the reference would walk off the root) */
			return 0;
		if (m->c_off[v + 1] - m->c_off[v] > 1) {
			if (branching == NIL)
				branching = v;
			n_branching++;
		}
	}
	if (n_branching == 0) /* cond_a */
		return 1;
	if (n_branching > 1) /* cond_b: more than one branching vertex */
		return 0;
	for (uint32_t k = m->c_off[branching]; k < m->c_off[branching + 1]; k++) {
		uint32_t c = m->c_adj[k];
		if (c > zi && t->post[c] - t->pre[c] == 3) {
			LS_INC(LS_TRUNK_COND_B);
			return 1;
		}
	}
	return 0;
}

/* in_trunk, parallel.cpp:110-184 */
static int par_in_trunk(const orc_tree *t, const tree_meta *m, uint32_t ai, uint32_t zi)
{
	if (zi - ai <= 3 && m->i_off[ai + 1] - m->i_off[ai] <= 1) /* condition i */
		return 0;
	if (m->c_off[zi + 1] - m->c_off[zi] != 1) /* condition iii */
		return 0;
	if (!par_inspect_trunk(t, m, ai, zi)) /* condition iv */
		return 0;
	uint32_t in_ai = count_type(t, m->i_off, m->i_adj, ai, ORC_BE_BACK);
	if (2 * in_ai >= (zi - ai) - 3) { /* u32, as the reference */
		LS_INC(LS_PAR_TRUNK_AI);
		return 1;
	}
	if (in_ai != 0)
		return 0;
	uint32_t out_zi = count_type(t, m->o_off, m->o_adj, zi, ORC_BE_BACK);
	if (2 * out_zi >= (zi - ai) - 3) {
		LS_INC(LS_PAR_TRUNK_ZI);
		return 1;
	}
	return 0;
}

/* in_branch, parallel.cpp:186-261 */
static int par_in_branch(const orc_tree *t, const tree_meta *m, uint32_t ai, uint32_t zi)
{
	if (zi - ai != 1)
		return 0;
	uint32_t c = NIL;
	for (uint32_t k = m->c_off[zi]; k < m->c_off[zi + 1]; k++) {
		uint32_t x = m->c_adj[k];
		if (t->pe_black[x])
			continue;
		if (c != NIL)
			return 0;
		c = x;
	}
	if (c == NIL) /* undefined in the reference (see the header of this section) */
		return 0;
	const uint64_t br = m->off[c + 1] - m->off[c], ch_obe = m->o_off[c + 1] - m->o_off[c];
	if (br <= 2)
		return 0;
	if (count_type(t, m->i_off, m->i_adj, ai, ORC_BE_BACK) >= br + ch_obe) {
		LS_INC(LS_PAR_BRANCH_AI);
		return 1;
	}
	if (m->o_off[zi + 1] - m->o_off[zi] >= br + ch_obe) {
		LS_INC(LS_PAR_BRANCH_ZI);
		return 1;
	}
	return 0;
}

/* find_tiny + find_parallel over the leaves of the PVST; fam[v] = 'D' 'F' 'T' 'O' */
void orc_leaf_subflubbles(const orc_tree *t, orc_pvst *p)
{
	tree_meta m;
	tree_meta_build(t, &m);
	free(p->fam);
	p->fam = xmalloc((size_t)p->n + 1);
	uint8_t *has_child = xcalloc((size_t)p->n + 1, 1);
	for (uint32_t v = 1; v < p->n; v++)
		has_child[p->parent[v]] = 1;
	p->fam[0] = 'D';
	for (uint32_t v = 1; v < p->n; v++)
		p->fam[v] = 'F';
	for (uint32_t v = 1; v < p->n; v++) { /* find_tiny */
		if (has_child[v])
			continue;
		LS_INC(LS_LEAVES);
		uint32_t ai = p->ai[v], zi = p->zi[v];
		if (!(zi - ai == 1 || zi - ai == 3))
			continue;
		if (tiny_branches(t, &m, ai, zi)) /* trunk() is false on every path */
			p->fam[v] = 'T';
	}
	for (uint32_t v = 1; v < p->n; v++) { /* find_parallel */
		if (has_child[v] || p->fam[v] != 'F')
			continue;
		if (par_in_branch(t, &m, p->ai[v], p->zi[v]) || par_in_trunk(t, &m, p->ai[v], p->zi[v]))
			p->fam[v] = 'O';
	}
	free(has_child);
	tree_meta_free(&m);
}

/* --------------------------------------------------------------- PVST text
 * write_pvst, src/mto/to_pvst.cpp:23-109; children joined by ", "
 * (print_with_comma, include/povu/common/utils.hpp:44-55); id_or_t::as_str
 * (include/povu/graph/types.hpp:85-95). */
typedef struct {
	char *b;
	size_t n, cap;
} sbuf;
static void sb_need(sbuf *s, size_t more)
{
	if (s->n + more + 1 > s->cap) {
		while (s->n + more + 1 > s->cap)
			s->cap = s->cap ? s->cap * 2 : 4096;
		s->b = realloc(s->b, s->cap);
		if (!s->b)
			abort();
	}
}
static void sb_str(sbuf *s, const char *x)
{
	size_t l = strlen(x);
	sb_need(s, l);
	memcpy(s->b + s->n, x, l);
	s->n += l;
}
static void sb_u32(sbuf *s, uint32_t v)
{
	char tmp[16];
	int l = snprintf(tmp, sizeof tmp, "%" PRIu32, v);
	sb_need(s, (size_t)l);
	memcpy(s->b + s->n, tmp, (size_t)l);
	s->n += (size_t)l;
}

char *orc_pvst_text(const orc_pvst *p, size_t *len)
{
	sbuf s = {0};
	uint32_t *c_off = xcalloc((size_t)p->n + 1, 4), *c_adj = xmalloc((size_t)p->n * 4);
	for (uint32_t v = 1; v < p->n; v++)
		c_off[p->parent[v] + 1]++;
	for (uint32_t v = 0; v < p->n; v++)
		c_off[v + 1] += c_off[v];
	uint32_t *cur = xmalloc((size_t)(p->n + 1) * 4);
	memcpy(cur, c_off, (size_t)(p->n + 1) * 4);
	for (uint32_t v = 1; v < p->n; v++)
		c_adj[cur[p->parent[v]]++] = v;
	free(cur);
	sb_str(&s, "H\t0.0.3\t.\t.\t.\n");
	for (uint32_t i = 0; i < p->n; i++) {
		if (p->fam) {
			char l[3] = {(char)p->fam[i], '\t', 0};
			sb_str(&s, l);
		} else {
			sb_str(&s, i == 0 ? "D\t" : "F\t");
		}
		sb_u32(&s, i);
		sb_str(&s, "\t");
		if (i == 0) {
			sb_str(&s, ".");
		} else {
			sb_str(&s, p->a_or[i] ? "<" : ">");
			sb_u32(&s, p->a_id[i]);
			sb_str(&s, p->z_or[i] ? "<" : ">");
			sb_u32(&s, p->z_id[i]);
		}
		sb_str(&s, "\t");
		if (c_off[i] == c_off[i + 1]) {
			sb_str(&s, ".");
		} else {
			for (uint32_t k = c_off[i]; k < c_off[i + 1]; k++) {
				sb_u32(&s, c_adj[k]);
				if (k + 1 < c_off[i + 1])
					sb_str(&s, ", ");
			}
		}
		sb_str(&s, i == 0 ? "\t.\n" : "\tL\n");
	}
	free(c_off);
	free(c_adj);
	sb_need(&s, 1);
	s.b[s.n] = 0;
	*len = s.n;
	return s.b;
}

#include "povu_oracle_sub.inc" /* the three inserting passes of -s and the writer of their PVST */

/* ------------------------------------------------------------ orchestration
 * do_decompose, app/subcommand/decompose.cpp:94-160: component id = position
 * + 1 (:129), components with < 3 vertices are skipped (:135-142). */

/* one component through from_bd .. add_flubbles (decompose_component, decompose.cpp:30-76); the
 * stage seconds are added to t[0..3] (tree, classes, stack + next_seen, pvst) */
static void decompose_one(const orc_graph *cg, uint32_t c, orc_forest *f, int want_text, double *t)
{
	double a = now_s();
	orc_tree *tr = orc_from_bd(cg);
	double b = now_s();
	orc_cycle_equiv(tr);
	double d = now_s();
	orc_oic *st = NULL;
	uint32_t n = orc_eq_class_stack(tr, &st);
	uint32_t *ns = xmalloc((size_t)(n + 1) * 4);
	orc_next_seen(st, n, tr->n_class, ns);
	double e = now_s();
	orc_pvst *p = orc_add_flubbles(tr, st, ns, n);
	double h = now_s();
	t[0] += b - a;
	t[1] += d - b;
	t[2] += e - d;
	t[3] += h - e;
	if (g_leaf_sub == 1)
		orc_leaf_subflubbles(tr, p);
	f->n_pvst[c] = p->n;
	if (want_text)
		f->text[c] = g_leaf_sub == 2 ? orc_subflubbles_text(tr, p, &f->text_len[c]) : orc_pvst_text(p, &f->text_len[c]);
	free(st);
	free(ns);
	orc_pvst_free(p);
	orc_tree_free(tr);
}

typedef struct {
	orc_graph **cs;
	orc_forest *f;
	const uint32_t *list; /* components of this thread, in order */
	uint32_t n_list;
	int want_text;
	double t[4];
} mt_job;

static void *mt_worker(void *arg)
{
	mt_job *j = arg;
	for (uint32_t k = 0; k < j->n_list; k++) {
		const uint32_t c = j->list[k];
		if (j->cs[c]->nv >= 3) /* decompose.cpp:135-142 */
			decompose_one(j->cs[c], c, j->f, j->want_text, j->t);
	}
	return NULL;
}

/* do_decompose, decompose.cpp:94-160.  threads <= 1: the plain loop.  threads > 1, lpt == 0: the
 * reference's own scheme -- chunk_size = n_components / threads contiguous components per thread,
 * the last thread takes the remainder (thread_count, :78-92, :116-123).  lpt != 0: components
 * bin-packed by (segments + links), heaviest first (what the HIP path's sharding does). */
static orc_forest *decompose_graph_mt(const orc_graph *g, int want_text, int threads, int lpt)
{
	orc_forest *f = xcalloc(1, sizeof *f);
	double t0 = now_s();
	orc_graph **cs = NULL;
	uint32_t nc = orc_componetize(g, &cs);
	f->t_componetize = now_s() - t0;
	f->n_comp = nc;
	f->comp_nv = xcalloc(nc, 4);
	f->comp_ne = xcalloc(nc, 4);
	f->text = xcalloc(nc, sizeof *f->text);
	f->text_len = xcalloc(nc, sizeof *f->text_len);
	f->n_pvst = xcalloc(nc, 4);
	for (uint32_t c = 0; c < nc; c++) {
		f->comp_nv[c] = cs[c]->nv;
		f->comp_ne[c] = cs[c]->ne;
	}
	if (threads < 1)
		threads = 1;
	if ((uint32_t)threads > nc)
		threads = nc ? (int)nc : 1;
	uint32_t *order = xmalloc(((size_t)nc + 1) * 4);
	uint32_t *first = xcalloc((size_t)threads + 1, 4);
	if (!lpt) {
		const uint32_t chunk = nc / (uint32_t)threads;
		for (uint32_t c = 0; c < nc; c++)
			order[c] = c;
		for (int t = 0; t < threads; t++)
			first[t] = (uint32_t)t * chunk;
		first[threads] = nc;
	} else {
		/* heaviest first (stable), each to the least loaded thread */
		uint32_t *byw = xmalloc(((size_t)nc + 1) * 4), *owner = xmalloc(((size_t)nc + 1) * 4);
		uint64_t *load = xcalloc((size_t)threads, 8);
		for (uint32_t c = 0; c < nc; c++)
			byw[c] = c;
		/* insertion into a bucket sort would be overkill: components are few next to their size */
		for (uint32_t i = 1; i < nc; i++) {
			uint32_t c = byw[i];
			uint64_t w = (uint64_t)cs[c]->nv + cs[c]->ne;
			uint32_t j = i;
			while (j > 0 && (uint64_t)cs[byw[j - 1]]->nv + cs[byw[j - 1]]->ne < w) {
				byw[j] = byw[j - 1];
				j--;
			}
			byw[j] = c;
		}
		for (uint32_t k = 0; k < nc; k++) {
			int best = 0;
			for (int t = 1; t < threads; t++)
				if (load[t] < load[best])
					best = t;
			owner[byw[k]] = (uint32_t)best;
			load[best] += (uint64_t)cs[byw[k]]->nv + cs[byw[k]]->ne + 1;
			first[best + 1]++;
		}
		for (int t = 0; t < threads; t++)
			first[t + 1] += first[t];
		uint32_t *cur = xmalloc(((size_t)threads + 1) * 4);
		memcpy(cur, first, ((size_t)threads + 1) * 4);
		for (uint32_t k = 0; k < nc; k++)
			order[cur[owner[byw[k]]]++] = byw[k];
		free(cur);
		free(byw);
		free(owner);
		free(load);
	}
	mt_job *jobs = xcalloc((size_t)threads, sizeof *jobs);
	pthread_t *th = xcalloc((size_t)threads, sizeof *th);
	double w0 = now_s();
	for (int t = 0; t < threads; t++) {
		jobs[t].cs = cs;
		jobs[t].f = f;
		jobs[t].list = order + first[t];
		jobs[t].n_list = first[t + 1] - first[t];
		jobs[t].want_text = want_text;
		if (threads == 1)
			mt_worker(&jobs[t]);
		else if (pthread_create(&th[t], NULL, mt_worker, &jobs[t]) != 0) {
			fprintf(stderr, "povu_oracle: pthread_create failed\n");
			abort();
		}
	}
	if (threads > 1)
		for (int t = 0; t < threads; t++)
			pthread_join(th[t], NULL);
	f->t_wall_components = now_s() - w0;
	f->threads = (uint32_t)threads;
	for (int t = 0; t < threads; t++) {
		f->t_tree += jobs[t].t[0];
		f->t_classes += jobs[t].t[1];
		f->t_stack += jobs[t].t[2];
		f->t_pvst += jobs[t].t[3];
	}
	for (uint32_t c = 0; c < nc; c++) {
		if (f->n_pvst[c])
			f->total_flubbles += f->n_pvst[c] - 1;
		orc_graph_free(cs[c]);
	}
	free(jobs);
	free(th);
	free(order);
	free(first);
	free(cs);
	return f;
}

static orc_forest *decompose_graph(const orc_graph *g, int want_text)
{
	return decompose_graph_mt(g, want_text, 1, 0);
}

orc_forest *orc_decompose_arrays_mt(uint32_t nv, const uint32_t *vid, uint32_t ne, const uint32_t *ev1,
				    const uint8_t *es1, const uint32_t *ev2, const uint8_t *es2,
				    const uint8_t *tips, int want_text, int threads, int lpt)
{
	orc_graph *g = orc_graph_new(nv, ne);
	memcpy(g->vid, vid, (size_t)nv * 4);
	memcpy(g->ev1, ev1, (size_t)ne * 4);
	memcpy(g->ev2, ev2, (size_t)ne * 4);
	memcpy(g->es1, es1, ne);
	memcpy(g->es2, es2, ne);
	orc_graph_build_csr(g);
	if (tips)
		memcpy(g->tip, tips, nv);
	else
		orc_graph_infer_tips(g);
	orc_forest *f = decompose_graph_mt(g, want_text, threads, lpt);
	orc_graph_free(g);
	return f;
}

orc_forest *orc_decompose_arrays(uint32_t nv, const uint32_t *vid, uint32_t ne, const uint32_t *ev1,
				 const uint8_t *es1, const uint32_t *ev2, const uint8_t *es2,
				 const uint8_t *tips, int want_text)
{
	return orc_decompose_arrays_mt(nv, vid, ne, ev1, es1, ev2, es2, tips, want_text, 1, 0);
}

void orc_forest_free(orc_forest *f)
{
	if (!f)
		return;
	for (uint32_t c = 0; c < f->n_comp; c++)
		free(f->text[c]);
	free(f->text);
	free(f->text_len);
	free(f->comp_nv);
	free(f->comp_ne);
	free(f->n_pvst);
	free(f);
}

static void *dupmem(const void *p, size_t n)
{
	void *q = xmalloc(n);
	if (n)
		memcpy(q, p, n);
	return q;
}

orc_dump *orc_dump_component(uint32_t nv, const uint32_t *vid, uint32_t ne, const uint32_t *ev1,
			     const uint8_t *es1, const uint32_t *ev2, const uint8_t *es2,
			     const uint8_t *tips, uint32_t comp)
{
	orc_graph *g = orc_graph_new(nv, ne);
	memcpy(g->vid, vid, (size_t)nv * 4);
	memcpy(g->ev1, ev1, (size_t)ne * 4);
	memcpy(g->ev2, ev2, (size_t)ne * 4);
	memcpy(g->es1, es1, ne);
	memcpy(g->es2, es2, ne);
	orc_graph_build_csr(g);
	if (tips)
		memcpy(g->tip, tips, nv);
	else
		orc_graph_infer_tips(g);
	orc_graph **cs = NULL;
	uint32_t nc = orc_componetize(g, &cs);
	orc_dump *d = NULL;
	if (comp < nc) {
		orc_graph *c = cs[comp];
		d = xcalloc(1, sizeof *d);
		d->nv = c->nv;
		d->ne = c->ne;
		d->gidx = dupmem(c->gidx, (size_t)c->nv * 4);
		d->ev1 = dupmem(c->ev1, (size_t)c->ne * 4);
		d->ev2 = dupmem(c->ev2, (size_t)c->ne * 4);
		d->es1 = dupmem(c->es1, c->ne);
		d->es2 = dupmem(c->es2, c->ne);
		if (c->nv >= 3) {
			orc_tree *t = orc_from_bd(c);
			orc_cycle_equiv(t);
			orc_oic *st = NULL;
			uint32_t n = orc_eq_class_stack(t, &st);
			uint32_t *ns = xmalloc((size_t)(n + 1) * 4);
			orc_next_seen(st, n, t->n_class, ns);
			orc_pvst *p = orc_add_flubbles(t, st, ns, n);
			d->n_tree = t->n;
			d->gid = dupmem(t->gid, (size_t)t->n * 4);
			d->par = dupmem(t->par, (size_t)t->n * 4);
			d->pe_id = dupmem(t->pe_id, (size_t)t->n * 4);
			d->cls = dupmem(t->cls, (size_t)t->n * 4);
			d->hi = dupmem(t->hi, (size_t)t->n * 4);
			d->typ = dupmem(t->typ, t->n);
			d->pe_black = dupmem(t->pe_black, t->n);
			d->n_be0 = t->n_be0;
			d->n_be = t->n_be;
			d->be_src = dupmem(t->be_src, (size_t)t->n_be * 4);
			d->be_tgt = dupmem(t->be_tgt, (size_t)t->n_be * 4);
			d->be_type = dupmem(t->be_type, t->n_be);
			d->n_stack = n;
			d->s_id = xmalloc((size_t)(n + 1) * 4);
			d->s_st_idx = xmalloc((size_t)(n + 1) * 4);
			d->s_edge_id = xmalloc((size_t)(n + 1) * 4);
			d->s_cls = xmalloc((size_t)(n + 1) * 4);
			d->s_orient = xmalloc((size_t)n + 1);
			for (uint32_t i = 0; i < n; i++) {
				d->s_id[i] = st[i].id;
				d->s_st_idx[i] = st[i].st_idx;
				d->s_edge_id[i] = t->pe_id[st[i].st_idx + 1];
				d->s_cls[i] = st[i].cls;
				d->s_orient[i] = st[i].orient;
			}
			d->next_seen = ns;
			d->n_pvst = p->n;
			d->p_parent = dupmem(p->parent, (size_t)p->n * 4);
			d->p_a_id = dupmem(p->a_id, (size_t)p->n * 4);
			d->p_z_id = dupmem(p->z_id, (size_t)p->n * 4);
			d->p_ai = dupmem(p->ai, (size_t)p->n * 4);
			d->p_zi = dupmem(p->zi, (size_t)p->n * 4);
			d->p_a_or = dupmem(p->a_or, p->n);
			d->p_z_or = dupmem(p->z_or, p->n);
			if (g_leaf_sub) {
				orc_leaf_subflubbles(t, p);
				d->p_fam = dupmem(p->fam, p->n);
			}
			d->pre = dupmem(t->pre, (size_t)t->n * 4);
			d->post = dupmem(t->post, (size_t)t->n * 4);
			d->n_bry = t->n_bry;
			d->bry = dupmem(t->bry, (size_t)t->n_bry * 16);
			free(st);
			orc_pvst_free(p);
			orc_tree_free(t);
		}
	}
	for (uint32_t c = 0; c < nc; c++)
		orc_graph_free(cs[c]);
	free(cs);
	orc_graph_free(g);
	return d;
}

void orc_dump_free(orc_dump *d)
{
	if (!d)
		return;
	free(d->gidx); free(d->ev1); free(d->ev2); free(d->es1); free(d->es2);
	free(d->gid); free(d->par); free(d->pe_id); free(d->cls); free(d->hi); free(d->typ);
	free(d->pe_black); free(d->be_src); free(d->be_tgt); free(d->be_type);
	free(d->s_id); free(d->s_st_idx); free(d->s_edge_id); free(d->s_cls); free(d->next_seen);
	free(d->s_orient); free(d->p_parent); free(d->p_a_id); free(d->p_z_id); free(d->p_ai);
	free(d->p_zi); free(d->p_a_or); free(d->p_z_or); free(d->bry); free(d->p_fam); free(d->pre); free(d->post);
	free(d);
}

int orc_decompose_gfa(const char *gfa, const char *outdir, char *err, size_t errlen)
{
	orc_graph *g = orc_graph_from_gfa(gfa, err, errlen);
	if (!g)
		return -1;
	orc_forest *f = decompose_graph(g, 1);
	orc_graph_free(g);
	int written = 0;
	for (uint32_t c = 0; c < f->n_comp; c++) {
		if (!f->text[c])
			continue;
		char path[4096];
		snprintf(path, sizeof path, "%s/%u.pvst", outdir, c + 1);
		FILE *o = fopen(path, "wb");
		if (!o) {
			snprintf(err, errlen, "Could not open file %s", path);
			orc_forest_free(f);
			return -1;
		}
		fwrite(f->text[c], 1, f->text_len[c], o);
		fclose(o);
		written++;
	}
	orc_forest_free(f);
	return written;
}

#ifdef ORC_MAIN
int main(int argc, char **argv)
{
	if (argc < 3) {
		fprintf(stderr, "usage: %s in.gfa outdir [0|1|2: passes of -s]\n", argv[0]);
		return 2;
	}
	if (argc > 3)
		orc_set_leaf_subflubbles(atoi(argv[3]));
	char err[512] = {0};
	int n = orc_decompose_gfa(argv[1], argv[2], err, sizeof err);
	if (n < 0) {
		fprintf(stderr, "%s\n", err);
		return 1;
	}
	return 0;
}
#endif
