// graph_kernels.hpp -- device state of rows A/B (see graph_kernels.hip)
#pragma once
#include "common.hpp"

namespace povu_hip
{

// The bidirected graph resident in HBM after povu_hip_graph_upload.
// side id = 2 * vertex idx + end (0 = l, 1 = r).
struct ResidentGraph {
	uint32_t V = 0, E = 0, n_slots = 0;
	bool tips_given = false;
	uint32_t *vid = nullptr;		    // [V]   segment id
	uint32_t *v1 = nullptr, *v2 = nullptr;	    // [E]   link endpoints (vertex idx)
	uint8_t *s1 = nullptr, *s2 = nullptr;	    // [E]   link endpoint sides
	uint8_t *tip = nullptr;			    // [V]   POVU_TIP_*
	uint32_t *off = nullptr;		    // [2V+1] per-side CSR offsets
	uint32_t *adj = nullptr;		    // [n_slots] incident link idx, ascending per side
	uint32_t *aoth = nullptr;		    // [n_slots] global side id (2 * vertex idx + end) at the other end of the slot's
						    //           link; a self loop points back into its own vertex
	uint32_t *atwin = nullptr;		    // [n_slots] the slot of the same link at its other end (itself for a same-side self loop)
	uint32_t max_vdeg = 0;			    // most links on one vertex (both sides)
	uint32_t n_empty_sides = 0;		    // sides without links (each may start one back edge to the root, spanning_tree.cpp:433-438)
	// device time of the last upload, by HIP events: host-to-device copies, CSR build, reverse-slot table
	float h2d_ms = 0, csr_ms = 0, twin_ms = 0;
	void *block = nullptr;			    // one allocation backing all of the above
};

static constexpr uint32_t LLE_ID = 0x1FFFFFFFu, LLE_TREE = 0x40000000u; // lle words (below)

// Rows B outputs, all in the "sorted" vertex space: sorted position i = vertices
// ordered by (component rank, global idx); component c owns positions
// [voff[c], voff[c+1]) and local vertex idx = i - voff[c]; sorted side id = 2i + end.
struct CompState {
	uint32_t *label, *flag, *crank, *comp_of, *tmp_a, *ckey, *perm, *pos;
	uint32_t *voff, *eoff, *vdeg, *sbase, *first, *erank, *ldeg, *loff, *ladj;
	uint32_t *keys, *vals, *keys2, *vals2;
	uint8_t *hook;	   // [2E] per adjacency slot: 1 = the link was united from this slot and merged two union-find trees (spanning forest of the segments)
	uint32_t *la, *lb; // [E]  sorted side ids of local edge le (la = first-encounter side); only the sorted-adjacency builder fills them
	uint32_t *lle;	   // [2E] per local adjacency slot: the link's id (LLE_ID bits) | LLE_TREE.  Both ends of a link carry the same id
			   //      and a side's slots ascend by it in first-encounter order: the position of the link's first-encounter slot
			   //      (sort-free builder) or its dense rank = componetize's local edge idx (sorted-adjacency builder)
	uint8_t *tgray;	   // [E+1] local edge is in the spanning forest (sorted-adjacency builder only; the traversal reads LLE_TREE)
	bool has_self_loops; // some link joins a segment to itself (the labelling kernel looked)
	bool dense_edges;  // la / lb / dense local edge ids are valid (sorted-adjacency builder)
	uint32_t *stats;   // [4]  stats[0] = max links on one side
	uint32_t *gid_s; // (may alias the resident graph's vid / tip: never written through)
	uint8_t *tip_s;
	uint64_t *start_key; // [C+1] (segment id << 32 | sorted side) of the smallest tip, ~0 if none
	void *scan_tmp, *sort_tmp;
	size_t scan_tmp_bytes, sort_tmp_bytes;
	bool lean_identity; // sorted space = global vertex space and no perm / pos / sbase / vdeg was written (gid_s / tip_s alias the graph's)
	bool comp_sorted;   // the vertices already are in (component, idx) order: re-indexing keeps every vertex where it is
	HostScratch *host;  // pinned read-back scratch of the owning context
	uint32_t *host_pub; // device view of a pinned [voff C+1 | eoff C+1 | stats 4] the re-index publishes into (or null)
};

void fill_u32(uint32_t *p, size_t n, uint32_t val, hipStream_t s);
void mark_odd_u32(uint32_t *p, size_t n, hipStream_t s); // p[i] = 1 for odd i (test hook)
// Builds off / adj / aoth / atwin / tip from the link arrays already in g (device memory).  Throws when a link names
// an unknown vertex or side (validated on the device).
void build_global_csr(ResidentGraph &g, Arena &tmp_arena, hipStream_t s);
uint32_t label_components(const ResidentGraph &g, CompState &st, StageTimer &tm, hipStream_t s);
// the same in two halves: everything queued, the answer (count, order flag, self-loop flag) published into page-locked memory
// by the last kernel -- read by _finish once the stream (or an event recorded behind _enqueue) has completed
uint32_t *label_components_enqueue(const ResidentGraph &g, CompState &st, StageTimer &tm, hipStream_t s);
uint32_t label_components_finish(CompState &st, const uint32_t *h);
void reindex_speculative_adj(const ResidentGraph &g, const CompState &st, uint32_t *ladj, uint32_t *lle, hipStream_t s);
// adj_done: reindex_speculative_adj already filled st.ladj / st.lle (only honoured when this graph's re-index has that form)
void reindex_components(const ResidentGraph &g, CompState &st, uint32_t C, StageTimer &tm, hipStream_t s,
			bool force_sorted_adjacency = false, bool adj_done = false);
// which builder the re-index will take for this graph (known from the upload: most links on one vertex)
bool sort_free_adjacency(const ResidentGraph &g, bool force_sorted_adjacency);

} // namespace povu_hip
