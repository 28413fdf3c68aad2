/*
 * povu_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded, CPU restatement of the reference `povu decompose`
 * hot path (pangenome/povu v0.0.1-alpha).  It exists to CHECK the HIP path; it
 * is never linked into, imported by, or executed from the product
 * (povu_amd/, the `povu` CLI, libpovu_ffi).  Only tests/, the smoke test in
 * __graft_entry__.py and the `cpu_baseline` leg of bench.py may load it.
 *
 * Pinning: the reference's hot path cannot be built in this image (it needs
 * the un-vendored liteseq headers), so this restatement is
 * pinned by (a) the reference's own known-answer tests and fixtures
 * (tests/integration_tests/pvst_tests.cc, tests/lean4_conformance fixtures and
 * lean_reference.lean / src/main.rs oracles) and (b) the survey-time md5
 * anchors of reference outputs recorded in BASELINE.md section 4.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference checkout).
 */
#ifndef POVU_ORACLE_H
#define POVU_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NIL 0xFFFFFFFFu /* include/povu/common/constants.hpp:26-35 */

/* vertex ends / tree vertex types (include/povu/graph/types.hpp:39-66) */
enum { ORC_L = 0, ORC_R = 1, ORC_DUMMY = 2 };
/* tip marks per vertex */
enum { ORC_TIP_NONE = 0, ORC_TIP_L = 1, ORC_TIP_R = 2 };
/* back-edge kinds (include/povu/graph/spanning_tree.hpp be_type_e) */
enum { ORC_BE_BACK = 0, ORC_BE_CAPPING = 1, ORC_BE_SIMPLIFYING = 2 };

/* bd::VG restated as arrays (include/povu/graph/bidirected.hpp:95-210) */
typedef struct {
	uint32_t nv, ne;
	uint32_t *vid;	     /* [nv] segment id of vertex idx */
	uint32_t *ev1, *ev2; /* [ne] endpoint vertex idx */
	uint8_t *es1, *es2;  /* [ne] endpoint side ORC_L / ORC_R */
	uint8_t *tip;	     /* [nv] ORC_TIP_* */
	uint32_t *off;	     /* [2*nv+1] per-side CSR (side = 2*v+end) */
	uint32_t *adj;	     /* incident edge idx, ascending (std::set order) */
	uint32_t *gidx;	     /* [nv] global vertex idx (components only) or NULL */
} orc_graph;

/* pst::Tree restated (include/povu/graph/spanning_tree.hpp:30-402) */
typedef struct {
	uint32_t n;	   /* tree vertices = 2*nv (+1 with dummy root) */
	uint32_t *gid;	   /* [n] segment id or ORC_NIL for the dummy */
	uint8_t *typ;	   /* [n] ORC_L / ORC_R / ORC_DUMMY */
	uint32_t *par;	   /* [n] parent tree vertex, ORC_NIL for root */
	uint32_t *pe_id;   /* [n] id of parent tree edge (shared edge counter) */
	uint8_t *pe_black; /* [n] 1 if the parent edge is black */
	uint32_t *cls;	   /* [n] class of parent tree edge (after cycle_equiv) */
	uint32_t *hi;	   /* [n] */
	uint32_t *pre, *post;
	/* back edges (capacity n_be_cap) */
	uint32_t n_be, n_be0 /* count after from_bd */, be_cap;
	uint32_t *be_src, *be_tgt, *be_id;
	uint8_t *be_type;
	uint32_t n_class;
	/* hairpin boundaries (flubbles.cpp:621-656) */
	uint32_t n_bry;
	uint64_t *bry; /* pairs b1,b2 */
} orc_tree;

/* candidate stack entry, flubbles.hpp:37-42 (oic_t) */
typedef struct {
	uint8_t orient; /* 0 forward '>' , 1 reverse '<' */
	uint32_t id;
	uint32_t st_idx; /* tree edge idx (= child tree vertex - 1) */
	uint32_t cls;
} orc_oic;

/* pvst::Tree restated (include/povu/graph/pvst.hpp:719-932) */
typedef struct {
	uint32_t n;	  /* vertices incl. dummy root 0 */
	uint32_t *a_id;	  /* [n] (unused for vertex 0) */
	uint32_t *z_id;	  /* [n] */
	uint8_t *a_or;	  /* [n] 0 '>' 1 '<' */
	uint8_t *z_or;	  /* [n] */
	uint32_t *parent; /* [n] ORC_NIL for root */
	uint32_t *ai, *zi; /* [n] flubbles.cpp:264-290 */
	uint8_t *fam;	  /* [n] line letter 'D' 'F' 'T' 'O', or NULL before orc_leaf_subflubbles */
} orc_pvst;

/* ---- construction ---- */
orc_graph *orc_graph_new(uint32_t nv, uint32_t ne);
void orc_graph_free(orc_graph *g);
/* (re)build per-side CSR from the edge arrays (bidirected.cpp:317-340) */
void orc_graph_build_csr(orc_graph *g);
/* tips as the loader infers them (src/mto/from_gfa.cpp:262-277) */
void orc_graph_infer_tips(orc_graph *g);
/* GFA S/L tokenizer with the loader contract of DESIGN.md (row A). NULL + msg on error */
orc_graph *orc_graph_from_gfa(const char *path, char *err, size_t errlen);

/* bd::VG::componetize, bidirected.cpp:477-602. Returns number of components. */
uint32_t orc_componetize(const orc_graph *g, orc_graph ***out);

/* pst::Tree::from_bd, spanning_tree.cpp:262-463 */
orc_tree *orc_from_bd(const orc_graph *g);
void orc_tree_free(orc_tree *t);
/* simple_cycle_equiv + handle_vertex, flubbles.cpp:503-719 */
void orc_cycle_equiv(orc_tree *t);
/* br_desc + compute_eq_class_stack, tree_utils.cpp:19-155, flubbles.cpp:412-501 */
uint32_t orc_eq_class_stack(const orc_tree *t, orc_oic **out);
/* compute_eq_class_metadata, flubbles.cpp:375-410 */
void orc_next_seen(const orc_oic *s, uint32_t n, uint32_t n_class, uint32_t *next_seen);
/* add_flubbles, flubbles.cpp:295-367 */
orc_pvst *orc_add_flubbles(const orc_tree *t, const orc_oic *s, const uint32_t *next_seen,
			   uint32_t n);
void orc_pvst_free(orc_pvst *p);
/* find_flubbles, flubbles.cpp:721-745 */
orc_pvst *orc_find_flubbles(orc_tree *t);
/* the two relabelling passes of `-s`: find_tiny (tiny.cpp:100-129) + find_parallel (parallel.cpp:263-287) over
 * gen_tree_meta's bracket table (tree_utils.cpp:531-690); the three inserting passes: orc_subflubbles_text.  Fills p->fam.
 * PARITY UNPINNED (the reference holds no T / O line anywhere). */
void orc_leaf_subflubbles(const orc_tree *t, orc_pvst *p);
/* when on, the decompose entry points and orc_dump_component run orc_leaf_subflubbles on every PVST */
void orc_set_leaf_subflubbles(int on); /* 0 = plain, 1 = the two relabelling passes, 2 = all five passes of -s in the PVST text */
/* all five passes of `povu decompose -s` (app/subcommand/decompose.cpp:63-70) on one component and the text write_pvst
 * makes of the result (povu_oracle_sub.inc: find_concealed, find_midi, find_smothered; PARITY UNPINNED).  Fills p->fam. */
char *orc_subflubbles_text(const orc_tree *t, orc_pvst *p, size_t *len);
/* which rule of the three inserting passes fired, counted since the last reset: out[23] = slubbles by kind (ai trunk, ai
 * branch, zi trunk, zi branch), children moved by nest_trunk_ai / nest_branch_ai / nest_trunk_zi, zi-branch slubbles left
 * unnested, slubbles of a flubble that is a "leaf" by the tree-index accident, midi bubbles, pairs of one kind (no bubble),
 * children moved under a midi bubble, smothered vertices by rule (g trunk target / source, g branch, s trunk, s branch source /
 * target), children moved under one, reads of the stale tail in smothered::nest, depth[] of an invalid index, self-loop
 * targets pushed by compute_LoA, override_ji_trunk answers */
void orc_sub_stats(uint64_t *out, int reset);

/* which rule decided, counted since the last reset (tests: does the fuzz reach every rule?): out[11] = leaves whose Y is
 * empty, tiny by a bracket to ai, tiny by an ordinary / a capping-or-simplifying back-edge INDEX equal to ai, parallel by
 * in_branch with ai / with zi, by in_trunk with ai / with zi, inspect_trunk's cond_b, leaves looked at, times the index comparison was reached with a
 * non-empty OBE(c). */
void orc_leaf_stats(uint64_t *out, int reset);
/* write_pvst, src/mto/to_pvst.cpp:30-109: returns malloc'd text, length in *len */
char *orc_pvst_text(const orc_pvst *p, size_t *len);

/* do_decompose, app/subcommand/decompose.cpp:94-160 (single thread).
 * Writes <outdir>/<k>.pvst; returns number of files written or -1. */
int orc_decompose_gfa(const char *gfa, const char *outdir, char *err, size_t errlen);

/* ---- flat helpers for ctypes-driven tests / bench ---- */
typedef struct {
	uint32_t n_comp;       /* all components incl. skipped ones */
	uint32_t *comp_nv;     /* [n_comp] */
	uint32_t *comp_ne;     /* [n_comp] */
	char **text;	       /* [n_comp] PVST text or NULL when skipped (<3 vertices) */
	size_t *text_len;      /* [n_comp] */
	uint32_t *n_pvst;      /* [n_comp] PVST vertex count (0 when skipped) */
	uint64_t total_flubbles;
	double t_componetize, t_tree, t_classes, t_stack, t_pvst; /* seconds (summed over threads) */
	double t_wall_components; /* wall seconds of the per-component part (all threads) */
	uint32_t threads;
} orc_forest;

/* edges given by vertex IDX (not id). tips: NULL = infer as the GFA loader does,
 * else explicit ORC_TIP_* per vertex. want_text=0 skips text formatting. */
orc_forest *orc_decompose_arrays(uint32_t nv, const uint32_t *vid, uint32_t ne, const uint32_t *ev1,
				 const uint8_t *es1, const uint32_t *ev2, const uint8_t *es2,
				 const uint8_t *tips, int want_text);
/* the same with the per-component part on `threads` threads: lpt == 0 = the reference's contiguous
 * chunks of components (decompose.cpp:78-92,116-157), lpt != 0 = components bin-packed by size */
orc_forest *orc_decompose_arrays_mt(uint32_t nv, const uint32_t *vid, uint32_t ne, const uint32_t *ev1,
				    const uint8_t *es1, const uint32_t *ev2, const uint8_t *es2,
				    const uint8_t *tips, int want_text, int threads, int lpt);
void orc_forest_free(orc_forest *f);

/* intermediate dumps for stage-level parity tests (single component graphs) */
typedef struct {
	uint32_t n_comp;
	uint32_t *comp_of;   /* [nv] component rank (0-based) of each global vertex */
	uint32_t *local_idx; /* [nv] local vertex idx inside its component */
} orc_comp_map;
orc_comp_map *orc_comp_map_of(const orc_graph *g);
void orc_comp_map_free(orc_comp_map *m);

/* every intermediate array of ONE component (0-based rank), for stage-level parity tests */
typedef struct {
	uint32_t nv, ne;	/* component graph */
	uint32_t *gidx;		/* [nv] global vertex idx */
	uint32_t *ev1, *ev2;	/* [ne] local edge endpoints in local edge order */
	uint8_t *es1, *es2;
	uint32_t n_tree;	/* tree vertices */
	uint32_t *gid, *par, *pe_id, *cls, *hi;
	uint8_t *typ, *pe_black;
	uint32_t n_be0, n_be;
	uint32_t *be_src, *be_tgt;
	uint8_t *be_type;
	uint32_t n_stack;
	uint32_t *s_id, *s_st_idx, *s_edge_id, *s_cls, *next_seen;
	uint8_t *s_orient;
	uint32_t n_pvst;
	uint32_t *p_parent, *p_a_id, *p_z_id, *p_ai, *p_zi;
	uint8_t *p_a_or, *p_z_or;
	uint8_t *p_fam; /* NULL unless orc_set_leaf_subflubbles(1) */
	uint32_t *pre, *post; /* [n_tree] the shared visit counter of from_bd */
	uint32_t n_bry;
	uint64_t *bry;
} orc_dump;
orc_dump *orc_dump_component(uint32_t nv, const uint32_t *vid, uint32_t ne, const uint32_t *ev1,
			     const uint8_t *es1, const uint32_t *ev2, const uint8_t *es2,
			     const uint8_t *tips, uint32_t comp);
void orc_dump_free(orc_dump *d);
void orc_set_faithful_rescan(int on);

#ifdef __cplusplus
}
#endif
#endif
