// leaf_kernels.hpp -- the two relabelling passes of `povu decompose -s` (the well-defined part).
#pragma once
#include "graph_kernels.hpp"
#include "par_kernels.hpp"
#include "seq_kernels.hpp"
#include "tree_kernels.hpp"

namespace povu_hip
{

// PVST line letters (include/povu/common/constants.hpp:53-60)
static constexpr uint8_t FAM_DUMMY = 'D', FAM_FLUBBLE = 'F', FAM_TINY = 'T', FAM_PARALLEL = 'O';

// per PVST vertex of the dense output (slot = the slot of the five PVST arrays): ai / zi of compute_ai_zi
// (flubbles.cpp:264-290; tree vertex idx inside the component, NIL for a root) and the line letter
struct LeafOut {
	uint32_t *ai = nullptr, *zi = nullptr;
	uint8_t *fam = nullptr;
};

struct LeafIn {
	uint32_t NE, C;
	const uint32_t *voff, *c_ntree, *cproc_ps;
	const uint32_t *e_i, *lev, *s_comp, *s_vtx, *ns;
	const uint32_t *t_size, *gp, *nchild;
	const uint8_t *t_flags;
	const uint32_t *out_ord, *in_ord, *nself, *in_ext;
	const uint8_t *capf, *simp, *hit1, *hit3;
	const uint32_t *P; // exclusive sums of (ordinary edges leaving - arriving)
	const uint32_t *O; // exclusive sums of out_ord
	const uint32_t *X; // exclusive sums of capping + simplifying edges per vertex
	const uint32_t *B; // exclusive sums of w (after k_leaf_closed)
};

// what one pass of the leaf stage keeps between its steps (all device memory of the caller's arena)
struct LeafState {
	LeafIn in;
	LeafOut dense;		       // labels of the dense (parallel) PVST output
	uint32_t *p_ai = nullptr, *p_zi = nullptr; // [V + C] per-component layout of the one-lane kernels: written by seq_pvst
	uint8_t *p_fam = nullptr, *p_has_child = nullptr;
	size_t P = 0;
};

size_t leaf_workspace_bytes(size_t V, size_t C, size_t total);

// find_tiny (tiny.cpp:100-129) + find_parallel (parallel.cpp:263-287) over gen_tree_meta's bracket table
// (tree_utils.cpp:531-690).  Needs the state of an all-parallel pass (tree + classes from the parallel kernels, dense back
// edges).  leaf_prepare carves `ar` from offset 0 and builds the per-vertex tables; leaf_dense labels every PVST the
// parallel stages emitted (ls.dense, indexed like the dense PVST output); leaf_seq labels the PVSTs of the components the
// sequential redo of add_flubbles rebuilt (comp_bad; seq_pvst wrote ls.p_ai / ls.p_zi) into ls.p_fam.
void leaf_prepare(const CompState &cs, const SeqWs &sw, const ParWs &pw, const TreeWs &tw, uint32_t C, Arena &ar, LeafState &ls,
		  hipStream_t s);
void leaf_dense(const LeafState &ls, const SeqWs &sw, const ParWs &pw, uint32_t C, hipStream_t s);
void leaf_seq(const LeafState &ls, const CompState &cs, const SeqWs &sw, const uint32_t *comp_bad, uint32_t C, hipStream_t s);

} // namespace povu_hip
