// seq_kernels.hpp -- workspace of the per-component traversal kernels (rows C-G)
#pragma once
#include "common.hpp"

namespace povu_hip
{

// tree vertex flags
#define TF_TYPE_MASK 3u /* 0 = l, 1 = r, 2 = dummy */
#define TF_BLACK 4u	/* parent tree edge is black */

#define SEQ_STAGE_TREE 1u
#define SEQ_STAGE_CLASSES 2u
#define SEQ_STAGE_STACK 4u
#define SEQ_STAGE_PVST 8u
#define SEQ_STAGE_ALL 15u
#define SEQ_STAGE_GIVEN_STACK 16u /* with SEQ_STAGE_PVST alone: s_vtx / s_cls / next_seen come from the parallel stages (class ids \
				      are global: the component's smallest is subtracted); only add_flubbles itself runs */

// Sizes: V vertices, E links, C components.
//   T = 2V + C tree vertices;  component c owns [toff(c), toff(c)+2*nv_c+1),  toff(c) = 2*voff[c] + c
//   B = E + V + 2T back edges; component c owns [boff(c), ...), boff(c) = eoff[c] + voff[c] + 2*toff(c)
//   candidate stack entries: one per segment, component c owns [voff[c], voff[c+1])
//   PVST vertices: <= nv_c + 1, component c owns [voff[c] + c, ...)
struct SeqWs {
	uint32_t V, E, C;
	uint32_t rank, world, flags;
	uint32_t stages;	  // SEQ_STAGE_* mask
	const uint32_t *comp_sel; // optional [C]: only components with a non-zero entry are processed
	// inputs (sorted space)
	const uint32_t *voff, *eoff, *loff, *ladj, *gid_s;
	const uint8_t *tip_s;
	const uint64_t *start_key;
	const uint32_t *order; // [C] components by size descending
	const uint32_t *owner; // [C] shard that owns the component
	uint32_t *tables;      // [5(C+1)] order | owner | processed-before | processed | stack entries before: one upload per pass
	// spanning tree
	uint32_t *t_gid, *t_par, *t_cls, *t_hi, *first_child, *next_sib, *last_child; // [T]
	bool want_depth = false; // the parallel tree stage writes t_depth only for its readers: hairpin reports, the inserting passes of -s
	uint32_t *t_size, *t_depth;						       // [T] subtree sizes, depths
	uint8_t *t_flags;							       // [T]
	uint32_t *ctr, *cur;							       // [2V]
	uint32_t *stk;								       // [T]
	uint8_t *selfloop;							       // [V]
	// back edges and brackets
	uint32_t *be_src, *be_tgt, *o_next, *i_next, *b_prev, *b_next, *b_rsize, *b_rclass; // [B]
	uint8_t *be_type, *b_in, *be_cdef;						     // [B]
	uint32_t *o_head, *o_tail, *i_head, *i_tail;					     // [T]
	uint32_t *l_head, *l_tail, *l_size, *bl;					     // [T]
	// candidate stack
	uint32_t *nxt, *st_head, *st_tail; // [T]
	uint32_t *s_vtx, *s_cls, *next_seen; // [V]
	uint32_t *last;			     // [B+T]
	// PVST
	uint32_t *p_parent, *p_a, *p_z; // [V+C]
	uint8_t *p_or;			// [V+C] bit0 a reverse, bit1 z reverse
	uint32_t *p_ai, *p_zi;		// [V+C] or null: compute_ai_zi of every flubble (flubbles.cpp:264-290), for the leaf subflubble passes
	uint32_t *aux;			// [V+C]
	uint8_t *in_s;			// [B+T]
	uint64_t *hairpins;		// [2*(V+C)] or null
	// per component results
	uint32_t *c_ntree, *c_nbe0, *c_nbe, *c_nstack, *c_npvst, *c_nclass, *c_nbry, *c_status; // [C]
};

__host__ __device__ inline uint64_t seq_toff(const uint32_t *voff, uint32_t c) { return 2ull * voff[c] + c; }

void launch_seq_components(const SeqWs &ws, hipStream_t s);
// zeroes the eight per-component result arrays (one launch)
void zero_component_counters(const SeqWs &ws, uint32_t C, uint32_t *comp_bad, uint32_t *err, hipStream_t s);

} // namespace povu_hip
