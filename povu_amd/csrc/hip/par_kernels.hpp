// par_kernels.hpp -- data-parallel formulations of rows D-G (see par_kernels.hip)
#pragma once
#include "common.hpp"
#include "graph_kernels.hpp"
#include "seq_kernels.hpp"

#include <functional>

namespace povu_hip
{

// Min segment tree over n values, kept COARSE: its leaves are the minima of blocks of SEG_BLK consecutive values, the
// values themselves stay where they are (val must outlive the queries).  A query reads the one or two blocks at its ends
// as whole cache lines and walks the tree only over the blocks in between: 1/16 of the writes of a full tree, and the
// short ranges most queries have never touch it (segtree.hpp).
struct SegTree {
	uint32_t *tree = nullptr;     // [2P], node 1 = root, block minima at [P, 2P)
	const uint32_t *val = nullptr; // [n] the values (16-byte aligned, readable up to the next multiple of SEG_BLK)
	uint32_t P = 1;		      // blocks, padded to a power of two with +inf
	static constexpr uint32_t BLK = 16;
	static uint32_t pow2(size_t n)
	{
		uint32_t p = 1;
		while (p < n)
			p <<= 1;
		return p;
	}
	static size_t tree_words(size_t n) { return 2 * (size_t)pow2((n + BLK - 1) / BLK + 1); }
};

struct ParWs {
	HostScratch *host; // pinned read-back scratch of the owning context
	uint32_t V, E, C, T;
	bool all_vertex_classes = false; // in: number the classes of all tree edges even when the black ones would do (A/B tests)
	bool check_laminar = false;	 // in: run the laminarity check of the candidate stack even when the class stage was exact
	bool laminar_checked = false;	 // out: the last pass ran it
	bool black_only_used = false;	 // out: the last pass numbered the classes of the black tree edges only
	bool s_cls_valid = false;	 // s_cls holds class ids (a black-only pass numbers its classes only when a debug hook asks)
	bool gcls_valid = false;	 // gcls holds the classes in T-space (a black-only pass keeps them in stack order only)
	// T-space (global tree vertex idx)
	uint32_t *t_comp, *t_root, *gpar, *gsize;
	uint32_t *hi0, *cov, *psA, *psB, *flagC, *psC; // exclusive scans of the byte flags below [T+1]; flagC: run marks
	uint8_t *f8a, *f8b, *f8c, *f8d; // [T+1] one-byte flags (bridge / simplifying / capping / branching vertex, class and stack flags)
	uint32_t *cap_tgt, *mpre, *dlt, *dlt_ps, *incnt, *psin, *topi, *lsz, *gcls;
	uint32_t *sdl;		 // [V+2] row E's difference array (k_shift_delta) when the tree stage's emit kernel already filled it
	bool sdl_filled = false; // ... which it says here
	uint32_t *vals_t, *vals_t2;
	uint32_t *keys_t, *keys_t2; // [T]
	// dense back edges / brackets
	uint32_t *dbo;			 // [C+1]
	uint32_t *b_src, *b_tgt, *b_val, *b_val2, *tgtR, *b_ord; // [NBmax]; b_ord = rank among the source's ordinary edges, bottom first
	uint32_t *b_key, *b_key2;	 // [NBmax] (only behind a sequential tree stage)
	size_t nb_cap = 0;		 // entries the bracket arrays were carved for
	// candidate stack space
	uint32_t *s_vtx, *s_cls, *s_comp, *ns, *prev; // [V+1]
	uint32_t *soff;				     // [C+1] first stack entry of a component (host-built table, set by the caller)
	uint32_t *walk, *walk_ps, *wrun; // [V+2] steps of the stack machine (one per entry), their prefix sums, running minimum (complemented)
	uint32_t *erank, *lev, *e_i;	 // [V+1]
	uint32_t *comp_bad;		 // [C+1] components that must be redone sequentially
	// dense PVST output (all processed components back to back): what goes over PCIe
	uint32_t *cproc_ps, *doff;	 // [C+1] processed components before c; first dense PVST slot of c
	size_t d_total;			 // PVST vertices of all processed components (dense output)
	uint32_t n_stack;		 // candidate-stack entries of the last pass
	uint32_t nb0 = 0, ncap = 0, nsimp = 0; // out: ordinary / capping / simplifying back edges of the last pass (b_src / b_tgt hold them in that order)
	uint32_t n_emitted = 0;		 // out: flubbles the last pass emitted (e_i / lev hold one entry each)
	uint32_t *d_a, *d_z, *d_parent;	 // [d_total] device views into the forest's page-locked result block
	uint8_t *d_aor, *d_zor;		 // [d_total] 0 forward, 1 reverse
	void *stage;			 // device block with the layout of the result block (large results are copied out by the DMA engine)
	uint32_t *err;			 // [4] internal error words
	SegTree segA, segB, segP, segL;
	// --hairpins on the parallel path
	uint8_t *hpf;			 // [T] bit0 simplifying vertex, bit1 top bracket is a simplifying edge
	uint32_t *hp1, *hp2, *hp3;	 // [T+1] segment-tree inputs / push flags
	unsigned long long *hp_b12;	 // [2T] boundary pairs before they are compacted
	SegTree segH1, segH2, segH3;
	void *scan_tmp, *sort_tmp;
	size_t scan_tmp_bytes, sort_tmp_bytes;
};

// second stream of a context + the two events that fork it from / join it to the main stream
struct SideStream {
	hipStream_t stream = nullptr;
	hipEvent_t fork = nullptr, join = nullptr;
	hipEvent_t fork2 = nullptr; // main -> side a second time: the PVST parents are ready for their copy
};

// The end of a pass (run_parallel_dg).  The pass reads everything the host needs to build the forest's tree table
// back TOGETHER with the PVST count (one synchronisation), so nothing but the PVST arrays themselves is still in
// flight when run_parallel_dg returns: the emit kernels and the copies of the finished arrays to the host, ALL of the
// copies on the side stream.  `done` is recorded behind the last of them.  With want_overlap the main stream does not
// wait for the side stream: the caller may start the next pass on it while the copy engine still moves this one's
// result over PCIe (povu_hip_decompose: POVU_HIP_F_ASYNC).
struct PassTail {
	bool want_overlap = false;		 // in
	hipEvent_t done = nullptr;		 // in: recorded when all work of the pass (kernels and copies) is complete
	hipEvent_t done2 = nullptr;		 // in: a second event recorded at the same point (the context's own)
	const uint32_t *early_summary = nullptr; // out: pass_summary's words in page-locked host memory, as of the PVST count
	bool summary_final = false;		 // out: nothing after the count changes them (no laminarity check ran)
	bool overlapped = false;		 // out: the main stream was left free (want_overlap and summary_final)
};

// What a pass will NOT need is not carved (the arrays' pointers are then null): the default is everything.
struct StageWsOpts {
	size_t nb_cap = 0;	     // brackets the pass can have at most (0: the loose bound E + V + T)
	bool full_t = true;	     // per-vertex component / root / parent / size tables: a sequential tree stage or the hairpin report
	bool sorted_brackets = true; // keys of the bracket sort: only behind a sequential tree stage
	bool hairpins = true;	     // the hairpin report's arrays
	bool walk_inline = true;     // the wave walk's records / stack pool / parents inside the tree stage's block (else taken from
				     // TreeWs::walk_arena when a pass has large classes)
};
// groups: bit 0 = the arrays the tree stage already writes, bit 1 = those first written by the class stage (see for_each_span)
size_t par_workspace_bytes(size_t V, size_t E, size_t Cmax, int groups = 3, const StageWsOpts &o = StageWsOpts{});
void par_carve(Arena &ar, ParWs &pw, size_t V, size_t E, size_t Cmax, int groups = 3, const StageWsOpts &o = StageWsOpts{});

// Runs rows D-G for every processed component from the spanning trees / back edges the tree stage
// left in `sw`.  Components whose candidate stack is not laminar are flagged in pw.comp_bad (see pass_summary).
// dense_nb0 < 0: densify the back edges the sequential tree stage wrote; otherwise b_src/b_tgt hold them -- NB0_ON_DEVICE:
// their number still sits in pw.err[6] (the parallel tree stage does not wait for it: the class stage reads it with the
// capping / simplifying counts, one host synchronisation instead of two).
// alloc_result_block(total) returns the device view of a page-locked host block laid out a | z | parent | a_or |
// z_or (each padded to 64 B) for `total` PVST vertices; the emit kernels write into it.
static constexpr int64_t NB0_ON_DEVICE = int64_t(1) << 40;
void run_parallel_dg(const CompState &cs, SeqWs &sw, ParWs &pw, uint32_t C, uint32_t n_processed, uint32_t n_stack,
		     int64_t dense_nb0, const std::function<void *(size_t)> &alloc_result_block, StageTimer &tm, hipStream_t s,
		     const SideStream &side, PassTail &tail);

// Hairpin boundaries (`--hairpins`, flubbles.cpp:531-535, 621-656, 712-717) from the parallel class stage's
// per-vertex flags; writes sw.hairpins / sw.c_nbry like the sequential kernels do.
void run_parallel_hairpins(const CompState &cs, SeqWs &sw, ParWs &pw, uint32_t C, StageTimer &tm, hipStream_t s);

// Copies the candidate stack / classes / next_seen of the last parallel pass into the per-component layout the
// debug hooks (and the sequential kernels) use; not needed by the pass itself.
// debug hook after a black-only pass: the classes of the black tree vertices into pw.gcls (NIL elsewhere)
void classes_to_tree_space(ParWs &pw, hipStream_t s);
void export_parallel_stack(const CompState &cs, SeqWs &sw, ParWs &pw, hipStream_t s);

// One launch that writes the outcome of a pass into page-locked host memory (5*C + 8 words):
// [0..3] error words of the parallel stages, then bad[C], status[C], npvst[C], nbry[C], doff[C+1].
// pw == nullptr (sequential-only pass): error words, bad and doff read as 0.  The caller synchronises.
void pass_summary(const SeqWs &sw, const ParWs *pw, uint32_t C, uint32_t *host_out, hipStream_t s);

} // namespace povu_hip
