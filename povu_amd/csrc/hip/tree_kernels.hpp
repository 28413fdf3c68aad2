// tree_kernels.hpp -- parallel spanning tree (row C), see tree_kernels.hip
#pragma once
#include "common.hpp"
#include "graph_kernels.hpp"
#include "par_kernels.hpp"
#include "seq_kernels.hpp"

namespace povu_hip
{

struct TreeWs {
	HostScratch *host; // pinned read-back scratch of the owning context
	// unrooted spanning forest of the biedged graph H, as arcs
	uint32_t *dist;					  // [2E] arcs behind an adjacency slot in its Euler tour (slots of hooked links = arcs, see tree_kernels.hip)
	ulonglong2 *xval, *xps;				  // [max(V, E) + 16] xor values (two 64-bit hashes) of the segments that have any, in tour order, and their running xor (inside the shared block, see tree_spans)
	uint4 *xrec;					  // [(4V+8)/64 + 4] per 64 tour positions {which of them carry a value (64 bits), set bits in front of the word, -}
	uint32_t *xrank;				  // [(4V+8)/64 + 4] scan buffer of the bit counts (the last entry: their total)
	uint4 *t0seg;					  // [V] rooted forest, per segment: {parent of the entered side, link to it | r bit, tour position in, out}
	uint2 *dps;					  // [2V] {DFS parent side, scan slot of the parent it was found through}
	uint8_t *dvis;					  // [2V]
	// the wave-cooperative walk of large 2-edge-connected classes
	uint32_t *wadj;					  // [2V + 2E] class-filtered scan lists (sides), deg + 1 slots per side
	uint4 *wrec;					  // [2 x 2V] 32-byte record per side: {count, list begin, first six candidates}
	uint2 *wstk;					  // [3 x 2V] pool of the walks' stacks {side, next candidate}
	uint32_t *wpar;					  // [2V] DFS parent of a side the walk reached, W_UNVIS before
	uint8_t *dvis_slots;				  // [2E] slot repeats an earlier link of its side (hub graphs)
	uint32_t *entry_ps, *entry_list;		  // [2V+1]
	uint32_t *side_tidx;				  // [2V] tree vertex (T-space) of a side
	uint32_t *be_cnt;				  // [2V+1] first arc of a side (tour); free afterwards
	uint32_t *rk_pk, *rk_heads;			  // list ranking: packed list words [4V+8], list heads [C]
	uint32_t *rk_nx, *rk_wa, *rk_wb, *rk_tA, *rk_tB, *rk_tC; // pools of the levels above the list itself
	uint2 *evt;					  // [4V+2] event ranks {enter count, depth} of the pre-order ranking
	uint32_t *cproc;				  // [C+1] 1 = component is decomposed by this shard
	const uint8_t *last_dupflag;			  // dvis_slots when the last pass filled it, else null
	bool tour_words_done = false;			  // tree_tour_words ran for this pass already (started ahead of the host's wait for the component sizes)
	hipStream_t walk_stream = nullptr; // second stream of the wave walks (the context's; null: both forms on the pass's stream)
	hipEvent_t walk_fork = nullptr, walk_join = nullptr;
	Arena *walk_arena = nullptr;			  // where the wave walk's arrays are taken from when the pass needs them and they are not inside the stage block
	bool walk_inline = true;
};

// groups: bit 0 = the arrays that outlive the tree stage, bit 1 = those that are dead when the class stage starts (tree_spans)
size_t tree_workspace_bytes(size_t V, size_t E, size_t Cmax, int groups = 3, const StageWsOpts &o = StageWsOpts{});
void tree_carve(Arena &ar, TreeWs &tw, size_t V, size_t E, size_t Cmax, int groups = 3, const StageWsOpts &o = StageWsOpts{});
// both parallel stages in one arena, the class stage's own arrays over the tree stage's dead ones
size_t stage_workspace_bytes(size_t V, size_t E, size_t Cmax, const StageWsOpts &o = StageWsOpts{});
void stage_workspace_carve(Arena &ar, ParWs &pw, TreeWs &tw, size_t V, size_t E, size_t Cmax, const StageWsOpts &o = StageWsOpts{});
// the wave walk's arrays (records of all sides, stack pool, parents): only a pass with large 2-edge-connected classes needs them
size_t walk_workspace_bytes(size_t V);

// the tree stage's first kernel (Euler-tour words of the spanning forest): it reads the re-indexed adjacency only, so
// povu_hip_decompose may start it before the host has the component sizes; run_parallel_tree then skips it
void tree_tour_words(const CompState &cs, uint32_t V, uint32_t E, TreeWs &tw, bool force_sparse_splitters, hipStream_t s);

// Builds the reference's spanning tree of every processed component (tree arrays in sw, T-space
// layout) and the dense list of from_bd back edges (pw.b_src / pw.b_tgt).  Returns their count.
int64_t run_parallel_tree(const CompState &cs, SeqWs &sw, ParWs &pw, TreeWs &tw, uint32_t C, uint32_t event_lists,
			   uint32_t max_side_links, bool force_big_class_dfs, bool force_sparse_splitters, StageTimer &tm,
			   hipStream_t s);

// conformance export: back edges from_bd creates just before each tree vertex (w, T-space) and after the last child of
// each (tail, T-space); both arrays must be zeroed by the caller.  Reads the state of the last run_parallel_tree.
void debug_edge_id_weights(const CompState &cs, const SeqWs &sw, const TreeWs &tw, uint32_t *w, uint32_t *tail, hipStream_t s);

} // namespace povu_hip
