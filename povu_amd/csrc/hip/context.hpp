// context.hpp -- the objects behind the opaque handles of include/povu_hip.h (shared by the .hip files that
// implement the C ABI).
#pragma once
#include "../../../include/povu_hip.h"

#include "graph_kernels.hpp"
#include "leaf_kernels.hpp"
#include "par_kernels.hpp"
#include "seq_kernels.hpp"
#include "sub_kernels.hpp"
#include "tree_kernels.hpp"

#include <map>
#include <memory>
#include <mutex>
#include <string>

using namespace povu_hip;

// Pinned host blocks for the PVST arrays: D2H into page-locked memory runs at PCIe speed, and a
// block returns to its context's pool when the forest is freed (steady state: no allocation).
// In SHARED mode (povu_hip_share_results) a block is a POSIX shared-memory segment "/povu.<tag>.<k>", page-locked and
// mapped for the device with hipHostRegister: another process of the node maps it by name and reads the PVST arrays where
// this GPU's copy engine put them -- a multi-process gather without a second trip over PCIe (shard.hip).
struct PinnedPool {
	struct Block {
		void *p;
		size_t cap;
		int seg; // shared mode: k of the segment name, else -1
	};
	std::mutex m;
	std::vector<Block> free_blocks;
	std::vector<Block> segments; // every segment this pool created (shared mode): unmapped and unlinked with the pool
	std::string shared_tag;	     // empty: plain hipHostMalloc blocks
	int next_seg = 0;
	~PinnedPool();
	void *get(size_t bytes, size_t &cap, int *seg = nullptr);
	void put(void *p, size_t cap, int seg = -1)
	{
		std::lock_guard<std::mutex> g(m);
		if (seg >= 0 || free_blocks.size() < 24) // (a gather on 8+ ranks cycles through one block per rank)
			free_blocks.push_back(Block{p, cap, seg}); // (a segment stays mapped until the pool goes)
		else
			(void)hipHostFree(p);
	}
	static std::string segment_name(const std::string &tag, int k) { return "/povu." + tag + "." + std::to_string(k); }
};

// A host array whose memory comes out of the context's pool of page-locked blocks when it is given one (the label arrays of
// -s: 9 bytes per PVST vertex -- as pageable std::vectors their copy off the device took 10 - 30 ms on the whole-genome
// workload, depending on what the allocator made of 220 fresh megabytes), else from malloc.  Just enough of std::vector's
// surface for the code that uses it; resize() does not initialise (every caller overwrites).
template <typename T>
struct PinnedVec {
	T *p = nullptr;
	size_t n = 0, cap_bytes = 0;
	int seg = -1;
	std::shared_ptr<PinnedPool> pool; // null: malloc
	PinnedVec() = default;
	PinnedVec(const PinnedVec &) = delete;
	PinnedVec &operator=(const PinnedVec &) = delete;
	PinnedVec(PinnedVec &&o) noexcept { *this = std::move(o); }
	PinnedVec &operator=(PinnedVec &&o) noexcept
	{
		if (this != &o) {
			release();
			p = o.p, n = o.n, cap_bytes = o.cap_bytes, seg = o.seg, pool = std::move(o.pool);
			o.p = nullptr, o.n = 0, o.cap_bytes = 0, o.seg = -1;
		}
		return *this;
	}
	~PinnedVec() { release(); }
	void release()
	{
		if (p) {
			if (pool)
				pool->put(p, cap_bytes, seg);
			else
				free(p);
		}
		p = nullptr, n = 0, cap_bytes = 0, seg = -1;
		pool.reset();
	}
	void resize(size_t m, const std::shared_ptr<PinnedPool> &from = nullptr)
	{
		release();
		if (!m)
			return;
		if (from && from->shared_tag.empty()) { // (a pool of named shared-memory segments keeps them for the PVST blocks)
			pool = from;
			p = static_cast<T *>(pool->get(m * sizeof(T), cap_bytes, &seg));
		} else {
			p = static_cast<T *>(malloc(m * sizeof(T)));
			if (!p)
				throw std::bad_alloc();
		}
		n = m;
	}
	void assign(size_t m, T v)
	{
		resize(m);
		std::fill(p, p + m, v);
	}
	void assign(const T *a, const T *b)
	{
		resize((size_t)(b - a));
		if (n)
			memcpy(p, a, n * sizeof(T));
	}
	T *data() { return p; }
	const T *data() const { return p; }
	T *begin() { return p; }
	const T *begin() const { return p; }
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
	T &operator[](size_t i) { return p[i]; }
	const T &operator[](size_t i) const { return p[i]; }
};

template <typename T>
struct Span { // just enough of std::vector's surface for the code below
	T *p = nullptr;
	T *data() const { return p; }
	T *begin() const { return p; }
	T &operator[](size_t i) const { return p[i]; }
};

struct povu_hip_forest {
	uint32_t total_components = 0;
	// the pass that fills this forest: ev0 at its first kernel, ev1 behind its last copy.  `pending` while the arrays may
	// still be on their way (POVU_HIP_F_ASYNC); every accessor calls ready() first
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	bool pending = false;
	double pass_ms = -1.0;
	// a merged forest took over blocks whose arrays were still on their way: the events behind those passes (owned here)
	std::vector<hipEvent_t> more_events;
	// (may be reached from several threads at once -- the writer threads of `povu decompose` read one forest: the waits are
	// idempotent, and the pass time is taken exactly once)
	std::once_flag pass_ms_once;
	void ready()
	{
		if (pending) {
			if (ev1)
				(void)hipEventSynchronize(ev1);
			for (hipEvent_t e : more_events)
				(void)hipEventSynchronize(e);
			pending = false;
		}
		if (ev0 && ev1)
			std::call_once(pass_ms_once, [this] {
				float ms = 0;
				if (pass_ms < 0 && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess)
					pass_ms = ms;
			});
	}
	std::shared_ptr<PinnedPool> pool;
	void *block = nullptr;
	size_t block_cap = 0, block_bytes = 0, total_entries = 0;
	int block_seg = -1;	 // shared-memory segment of the block (PinnedPool shared mode), else -1
	size_t meta_reserve = 0; // trees the block leaves room for behind the arrays (povu_hip_forest_share writes their table there)
	static size_t meta_bytes(size_t n_trees) { return 64 + 32 * n_trees; }
	void alloc(size_t total)
	{
		total_entries = total;
		const size_t bytes = ((total * 4 + 63) & ~size_t(63)) * 3 + ((total + 63) & ~size_t(63)) * 2 + 64 + meta_bytes(meta_reserve);
		block = pool->get(bytes, block_cap, &block_seg);
		char *q = static_cast<char *>(block);
		auto carve = [&](size_t b) {
			char *r = q;
			q += (b + 63) & ~size_t(63);
			return r;
		};
		a_id.p = (uint32_t *)carve(total * 4);
		z_id.p = (uint32_t *)carve(total * 4);
		parent.p = (uint32_t *)carve(total * 4);
		a_or.p = (uint8_t *)carve(total);
		z_or.p = (uint8_t *)carve(total);
		block_bytes = (size_t)(q - static_cast<char *>(block));
	}
	void release_block()
	{
		if (block && pool)
			pool->put(block, block_cap, block_seg);
		block = nullptr;
		block_cap = block_bytes = total_entries = 0;
		block_seg = -1;
	}
	// more page-locked blocks with the same five arrays: taken over from other forests or received from other ranks
	// (povu_hip_forest_merge / povu_hip_comm_gather), so that merging never copies a PVST array
	struct ExtraBlock {
		void *p = nullptr;
		size_t cap = 0, total = 0;
		int seg = -1;
		std::shared_ptr<PinnedPool> pool; // null: the memory is not this forest's (a segment of another rank, mapped by the context)
		uint32_t *a = nullptr, *z = nullptr, *parent = nullptr;
		uint8_t *aor = nullptr, *zor = nullptr;
		PinnedVec<uint32_t> sub_ai, sub_zi; // with POVU_HIP_F_LEAF_SUBFLUBBLES (see the forest's own sub_ai)
		PinnedVec<uint8_t> sub_fam;
		std::shared_ptr<SubForest> subx;
		void carve(size_t total_entries)
		{
			total = total_entries;
			char *q = static_cast<char *>(p);
			auto take = [&](size_t b) {
				char *r = q;
				q += (b + 63) & ~size_t(63);
				return r;
			};
			a = (uint32_t *)take(total * 4);
			z = (uint32_t *)take(total * 4);
			parent = (uint32_t *)take(total * 4);
			aor = (uint8_t *)take(total);
			zor = (uint8_t *)take(total);
		}
		static size_t bytes_for(size_t total) { return ((total * 4 + 63) & ~size_t(63)) * 3 + ((total + 63) & ~size_t(63)) * 2 + 64; }
	};
	std::vector<ExtraBlock> extra;
	~povu_hip_forest()
	{
		if (pending) { // the copy engine may still be writing the blocks
			if (ev1)
				(void)hipEventSynchronize(ev1);
			for (hipEvent_t e : more_events)
				(void)hipEventSynchronize(e);
		}
		if (ev0)
			(void)hipEventDestroy(ev0);
		if (ev1)
			(void)hipEventDestroy(ev1);
		for (hipEvent_t e : more_events)
			(void)hipEventDestroy(e);
		release_block();
		if (xblk && pool)
			pool->put(xblk, xblk_cap, xblk_seg);
		for (auto &b : extra)
			if (b.p && b.pool)
				b.pool->put(b.p, b.cap, b.seg);
	}
	struct Tree {
		uint32_t component_id, n_vtx, n_links, n_pvst;
		size_t off;	// into the flat arrays below
		size_t hp_off;	// into hairpins (pairs)
		uint32_t n_hairpins;
		int blk = -1;	// -1: the arrays of this forest's own block, else extra[blk]
		uint32_t sub_c = 0; // with POVU_HIP_F_SUBFLUBBLES: its component in `subx` (of the forest, or of extra[blk])
	};
	std::vector<Tree> trees;
	Span<uint32_t> a_id, z_id, parent;
	Span<uint8_t> a_or, z_or;
	std::vector<uint64_t> hairpins;
	// with POVU_HIP_F_LEAF_SUBFLUBBLES: ai / zi (flubbles.cpp:264-290) and the line letter of every PVST vertex, indexed
	// like the arrays of this forest's own block
	PinnedVec<uint32_t> sub_ai, sub_zi;
	PinnedVec<uint8_t> sub_fam;
	std::shared_ptr<SubForest> subx; // with POVU_HIP_F_SUBFLUBBLES: the trees after all five passes of -s
	// povu_hip_forest_share: a second shared-memory segment with what the five arrays do not hold (labels, hairpin
	// boundaries, the extended trees of -s), kept alive as long as the forest
	void *xblk = nullptr;
	size_t xblk_cap = 0;
	int xblk_seg = -1;
};

struct povu_hip_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	SideStream side; // PCIe-bound result writes run beside the main stream's kernels
	ResidentGraph g;
	Arena ws_sub;		// tables of the inserting passes of -s (sub_kernels.hip), reserved for what the call before needed
	size_t ws_sub_hint = 0;
	Arena ws, ws_b, ws2, ws_seq, ws_leaf, ws_walk, upload_tmp; // (ws_b: what the re-index needs beyond the labelling's arrays; // (ws_walk: the wave walk's arrays, taken by the first pass that meets large classes)
	HostScratch host;
	std::shared_ptr<PinnedPool> pool = std::make_shared<PinnedPool>();
	StageTimer timer;
	std::vector<povu_hip_stage_time> last_times;
	uint64_t last_links = 0;
	// device state of the last decompose (debug / parity hooks)
	bool have_state = false;
	uint32_t C = 0;
	CompState cs{};
	SeqWs sw{};
	ParWs pw{};
	TreeWs tw{};
	uint32_t last_seq_redo = 0;
	bool redo_pvst_only = false; // the last redo only re-ran add_flubbles: tree, classes and stack are the parallel stages'
	bool last_mixed = false; // the last pass redid SOME components sequentially: the stage state is half parallel layout, half sequential
	bool stack_export_pending = false; // the parallel stages' candidate stack is still in its dense layout
	bool tree_in_par = false;	   // the tree of the last pass came from the parallel kernels (their per-side state is still there)
	bool classes_in_par = false;	   // the classes of the last pass sit in the parallel stage's own array (pw.gcls)
	// when the resident graph is a shard (povu_hip_graph_upload_shard): ids of its components in the whole graph
	// (1-based, ascending = the shard's own component order) and the component count of the whole graph
	std::vector<uint32_t> shard_comp_ids;
	uint32_t shard_total_components = 0;
	Arena shard_buf;   // a shard received from another rank
	Arena graph_arena; // backs the resident graph
	// the tail of the last POVU_HIP_F_ASYNC pass (side-stream kernels and copies that read the stage workspace): recorded
	// behind it; the next pass waits for it before it touches that workspace, every other entry point before anything
	hipEvent_t tail_done = nullptr;
	SideStream walk_side; // the wave walks' second stream (tree stage: the two forms of the walk run side by side)
	hipEvent_t host_wait = nullptr; // recorded behind a kernel whose words the host reads while LATER kernels already run (povu_hip_decompose)
	bool tail_pending = false;
	void wait_tail()
	{
		if (tail_pending) {
			(void)hipEventSynchronize(tail_done);
			tail_pending = false;
		}
	}
	void quiesce() // debug hooks: nothing of the last pass may still be running
	{
		if (tail_pending) {
			(void)hipStreamSynchronize(stream);
			wait_tail();
		}
	}
	Arena part_arena;  // the packed shards of the last povu_hip_shard_partition (kept warm: a step of a sharded job re-partitions)
	// bytes this context moved over PCIe / to peers since it was created (povu_hip_transfer_bytes)
	uint64_t xfer_h2d = 0, xfer_d2h = 0, xfer_peer_out = 0, xfer_peer_in = 0;
	// result segments of other ranks, mapped read-only by name (povu_hip_forest_attach); unmapped with the context
	struct Mapped {
		void *p;
		size_t bytes;
	};
	std::map<std::string, Mapped> attached;
};

// adds what the calling thread moved over PCIe during one C ABI call to the context's totals
struct XferScope {
	povu_hip_ctx *ctx;
	XferTally at;
	explicit XferScope(povu_hip_ctx *c) : ctx(c), at(xfer_tally()) {}
	~XferScope()
	{
		if (!ctx)
			return;
		const XferTally now = xfer_tally();
		ctx->xfer_h2d += now.h2d - at.h2d;
		ctx->xfer_d2h += now.d2h - at.d2h;
	}
};


// device block of a resident graph (link arrays + CSR); build_global_csr fills the CSR part
void alloc_resident_graph(Arena &arena, ResidentGraph &g, uint32_t n_vtx, uint32_t n_links, bool tips_given);
void free_resident_graph(ResidentGraph &g);
void check_graph_size(uint32_t n_vtx, uint32_t n_links);
void set_err(char *err, size_t errlen, const std::string &msg);

struct Sizes {
	size_t V, E, Cmax, T, B, nS, slots;
};
// Rows A/B in two steps (povu_hip_decompose): before the components are labelled only what the labelling writes is carved
// (rowb_carve_label); what the re-index needs follows when the component count, the order of the vertices and the
// builder are known (rowb_carve_reindex, from a second arena) -- a graph whose vertices already come grouped by component,
// without hub vertices and self loops (a pangenome GFA), needs a quarter of what the general case does.
struct RowBNeeds {
	bool identity;	// one component, or the vertices already in (component, idx) order
	bool sort_free; // no hub vertex: the builder that needs no sort
	bool self_loops;
};
size_t rowb_carve_label(Arena *ar, const Sizes &z, CompState &cs);
size_t rowb_carve_reindex(Arena *ar, const Sizes &z, size_t C, const RowBNeeds &need, CompState &cs);
// Workspace carving (or just measuring when `ar` is null); part 0 = rows A/B state (CompState), see povu_hip.hip
size_t carve_workspace(Arena *ar, int part, const Sizes &z, CompState &cs, SeqWs &sw, bool hairpins);
