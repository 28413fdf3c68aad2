/*
 * povu_ffi.h -- C ABI of the povu-rs / povu-ffi boundary, served by the MI355X build.
 *
 * Drop-in for the reference's `povu_ffi` static library (povu-rs/povu-ffi/povu_ffi.h:147-437,
 * bound by bindgen with allowlist povu_.* / Povu.*, povu-rs/build.rs:87-93): every symbol, struct
 * and enum the reference exports is exported here with the same layout, ownership rules and
 * failure values.  Per symbol the reference line is given as [ffi.h:N].
 *
 * Differences, all documented in INTEGRATION.md:
 *  - povu_graph_find_flubbles really decomposes the graph (on the GPU).  The reference's version
 *    can only fail (it runs find_flubbles on an empty spanning tree, povu_ffi.cpp:369-391).
 *  - additive symbols at the end (povu_graph_decompose / povu_forest_*) expose the full forest.
 *
 * Ownership: opaque handles are freed by their povu_*_free; arrays and strings handed out are
 * new[]-allocated and freed by povu_vertices_free / povu_edges_free / povu_paths_free /
 * povu_string_free; PovuError.message is freed by povu_error_free (or povu_string_free, which is
 * what povu-rs does, src/error.rs:93-101).  Nothing throws across this boundary.
 */
#ifndef POVU_FFI_H
#define POVU_FFI_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PovuGraph PovuGraph;	    /* [ffi.h:33] */
typedef struct PovuFlubbles PovuFlubbles;   /* [ffi.h:34] */
typedef struct PovuPvstTree PovuPvstTree;   /* [ffi.h:35] */
typedef struct PovuVcfOutput PovuVcfOutput; /* [ffi.h:36] */

typedef enum { POVU_ORIENTATION_FORWARD = 0, POVU_ORIENTATION_REVERSE = 1 } PovuOrientation; /* [ffi.h:53-56] */

typedef struct { uint64_t id; const char *sequence; size_t sequence_len; } PovuVertex; /* [ffi.h:63-67] */
typedef struct { uint64_t from_id; PovuOrientation from_orientation; uint64_t to_id; PovuOrientation to_orientation; } PovuEdge; /* [ffi.h:74-79] */
typedef struct { uint64_t vertex_id; PovuOrientation orientation; } PovuStep; /* [ffi.h:86-89] */
typedef struct { const char *name; size_t name_len; PovuStep *steps; size_t steps_count; } PovuPath; /* [ffi.h:97-102] */
typedef struct { /* [ffi.h:110-118] */
	uint64_t id; const char *type_name; uint64_t start_vertex_id; uint64_t end_vertex_id;
	PovuStep **walks; size_t *walk_lengths; size_t walks_count;
} PovuFlubble;
typedef struct { int code; char *message; } PovuError; /* [ffi.h:126-129] */

/* graph life cycle [ffi.h:147-166] */
PovuGraph *povu_graph_new(size_t vertex_capacity, size_t edge_capacity, size_t path_capacity);
PovuGraph *povu_graph_from_gfa(const char *gfa_path, PovuError *error);
void povu_graph_free(PovuGraph *graph);

/* in-memory construction [ffi.h:182-223]; add_* return the new idx or (size_t)-1.
 * Orientation FORWARD maps to the left end, REVERSE to the right end (povu_ffi.cpp:148-156). */
size_t povu_graph_add_vertex(PovuGraph *graph, uint64_t id, const char *sequence);
size_t povu_graph_add_edge(PovuGraph *graph, uint64_t from_id, PovuOrientation from_orientation, uint64_t to_id,
			   PovuOrientation to_orientation);
bool povu_graph_add_path(PovuGraph *graph, const char *name, const PovuStep *steps, size_t steps_count); /* always false, as povu_ffi.cpp:165-181 */
void povu_graph_finalize(PovuGraph *graph);

/* topology queries [ffi.h:232-273] */
size_t povu_graph_vertex_count(const PovuGraph *graph);
size_t povu_graph_edge_count(const PovuGraph *graph);
size_t povu_graph_path_count(const PovuGraph *graph);
PovuVertex *povu_graph_get_vertices(const PovuGraph *graph, size_t *count);
PovuEdge *povu_graph_get_edges(const PovuGraph *graph, size_t *count);
PovuPath *povu_graph_get_paths(const PovuGraph *graph, size_t *count);
void povu_vertices_free(PovuVertex *vertices, size_t count);
void povu_edges_free(PovuEdge *edges, size_t count);
void povu_paths_free(PovuPath *paths, size_t count);

/* reference selection [ffi.h:290-304] (stored only; `decompose` does not use references) */
bool povu_graph_set_references_from_file(PovuGraph *graph, const char *ref_file, PovuError *error);
bool povu_graph_set_references_from_prefixes(PovuGraph *graph, const char **prefixes, size_t count, PovuError *error);

/* flubble detection [ffi.h:319-341]: the hot path.  NULL + PovuError{code 1} on failure (no GPU,
 * bad graph).  povu_flubbles_count = PVST vertices (pvst_tree.vtx_count(), povu_ffi.cpp:397-400);
 * with several components the forest counts as one tree under a single dummy root. */
PovuFlubbles *povu_graph_find_flubbles(PovuGraph *graph, PovuError *error);
void povu_flubbles_free(PovuFlubbles *flubbles);
size_t povu_flubbles_count(const PovuFlubbles *flubbles);
PovuFlubble *povu_flubbles_get(const PovuFlubbles *flubbles, size_t index); /* always NULL, as povu_ffi.cpp:402-411 */
void povu_flubble_free(PovuFlubble *flubble);

/* PVST view [ffi.h:358-364]; non-owning view into its PovuFlubbles */
PovuPvstTree *povu_flubbles_get_pvst_tree(const PovuFlubbles *flubbles);
void povu_pvst_tree_free(PovuPvstTree *tree);
size_t povu_pvst_tree_vertex_count(const PovuPvstTree *tree);

/* VCF side [ffi.h:381-427]: outside the decompose path, same stubs as the reference */
PovuVcfOutput *povu_flubbles_call_variants(PovuFlubbles *flubbles, PovuError *error); /* always an error */
bool povu_vcf_write_to_file(const PovuVcfOutput *vcf, const char *path, PovuError *error);
char *povu_vcf_to_string(const PovuVcfOutput *vcf, size_t *length);
void povu_vcf_free(PovuVcfOutput *vcf);
void povu_string_free(char *str);
bool povu_gfa_to_vcf(const char *gfa_path, const char *vcf_path, const char *ref_file, PovuError *error); /* always an error */
void povu_error_free(PovuError *error); /* [ffi.h:437] */

/* ---- additive: the whole decompose result (what `povu decompose` writes) ---- */
typedef struct PovuForest PovuForest;
/* decomposes on HIP device `device`; hairpins != 0 also collects the --hairpins boundaries */
PovuForest *povu_graph_decompose(PovuGraph *graph, int device, int hairpins, PovuError *error);
size_t povu_forest_tree_count(const PovuForest *forest);      /* components with >= 3 vertices */
size_t povu_forest_component_count(const PovuForest *forest); /* all components */
uint32_t povu_forest_component_id(const PovuForest *forest, size_t i);	/* 1-based file number */
size_t povu_forest_pvst_vertex_count(const PovuForest *forest, size_t i);
char *povu_forest_pvst_text(const PovuForest *forest, size_t i, size_t *length); /* free with povu_string_free */
void povu_forest_free(PovuForest *forest);

#ifdef __cplusplus
}
#endif
#endif
