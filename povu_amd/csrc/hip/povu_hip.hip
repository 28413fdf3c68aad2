// povu_hip.hip -- C ABI (include/povu_hip.h) and host orchestration of the gfx950
// decompose path.  Mirrors povu::subcommands::decompose::do_decompose
// (app/subcommand/decompose.cpp:94-160) from "graph built" to "PVST ready to write".
#include "context.hpp"

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <numeric>
#include <type_traits>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

using namespace povu_hip;

// ---- page-locked result blocks (context.hpp)
PinnedPool::~PinnedPool()
{
	for (auto &b : free_blocks)
		if (b.seg < 0)
			(void)hipHostFree(b.p);
	for (auto &sg : segments) {
		(void)hipHostUnregister(sg.p);
		(void)munmap(sg.p, sg.cap);
		(void)shm_unlink(segment_name(shared_tag, sg.seg).c_str());
	}
}

void *PinnedPool::get(size_t bytes, size_t &cap, int *seg)
{
	const bool shared = !shared_tag.empty();
	{
		// best fit: a pass with -s takes two blocks of different sizes (the PVST arrays, the extended trees); first fit let the
		// smaller request walk off with the larger block, and the larger one page-locked a fresh 0.7 GB every call (60 ms)
		std::lock_guard<std::mutex> g(m);
		size_t best = free_blocks.size();
		for (size_t i = 0; i < free_blocks.size(); i++)
			if (free_blocks[i].cap >= bytes && (free_blocks[i].seg >= 0) == shared &&
			    (best == free_blocks.size() || free_blocks[i].cap < free_blocks[best].cap))
				best = i;
		if (best != free_blocks.size()) {
			const Block b = free_blocks[best];
			free_blocks.erase(free_blocks.begin() + best);
			cap = b.cap;
			if (seg)
				*seg = b.seg;
			return b.p;
		}
	}
	void *p = nullptr;
	cap = bytes + bytes / 4 + 4096;
	if (!shared) {
		if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess)
			throw HipError("hipHostMalloc failed for the PVST result block");
		if (seg)
			*seg = -1;
		return p;
	}
	// a named segment, page-locked and mapped for the device where it is
	cap = (cap + 4095) & ~size_t(4095);
	int k;
	{
		std::lock_guard<std::mutex> g(m);
		k = next_seg++;
	}
	const std::string name = segment_name(shared_tag, k);
	(void)shm_unlink(name.c_str()); // (a stale segment of a crashed job with the same tag)
	const int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
	if (fd < 0)
		throw HipError("shm_open failed for the shared PVST result block " + name);
	if (ftruncate(fd, (off_t)cap) != 0) {
		(void)close(fd);
		(void)shm_unlink(name.c_str());
		throw HipError("not enough shared memory for the PVST result block " + name + " (" + std::to_string(cap >> 20) + " MiB)");
	}
	p = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
	(void)close(fd);
	if (p == MAP_FAILED) {
		(void)shm_unlink(name.c_str());
		throw HipError("mmap failed for the shared PVST result block " + name);
	}
	if (hipHostRegister(p, cap, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
		(void)hipGetLastError();
		(void)munmap(p, cap);
		(void)shm_unlink(name.c_str());
		throw HipError("hipHostRegister failed for the shared PVST result block " + name);
	}
	{
		std::lock_guard<std::mutex> g(m);
		segments.push_back(Block{p, cap, k});
	}
	if (seg)
		*seg = k;
	return p;
}

extern "C" int povu_hip_share_results(povu_hip_ctx *ctx, const char *tag, char *err, size_t errlen)
{
	if (!ctx || !tag || !*tag || strlen(tag) > 96 || strchr(tag, '/')) {
		set_err(err, errlen, "share_results: bad tag");
		return 1;
	}
	// a fresh pool: blocks of the old one go back to it as their forests are freed
	auto pool = std::make_shared<PinnedPool>();
	pool->shared_tag = tag;
	ctx->pool = pool;
	return 0;
}

extern "C" int povu_hip_transfer_bytes(const povu_hip_ctx *ctx, uint64_t out[4])
{
	if (!ctx || !out)
		return 1;
	out[0] = ctx->xfer_h2d, out[1] = ctx->xfer_d2h, out[2] = ctx->xfer_peer_out, out[3] = ctx->xfer_peer_in;
	return 0;
}

void set_err(char *err, size_t errlen, const std::string &msg)
{
	if (err && errlen) {
		snprintf(err, errlen, "%s", msg.c_str());
	}
}

extern "C" const char *povu_hip_version(void) { return "povu-hip 0.1.0 (gfx950)"; }

extern "C" int povu_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

extern "C" povu_hip_ctx *povu_hip_create(int device, char *err, size_t errlen)
{
	try {
		int n = 0;
		if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
			throw HipError("no HIP device available: the decompose path has no CPU fallback");
		if (device < 0 || device >= n)
			throw HipError("HIP device index out of range");
		HIP_CHECK(hipSetDevice(device));
		auto ctx = std::make_unique<povu_hip_ctx>();
		ctx->device = device;
		HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
		HIP_CHECK(hipStreamCreateWithFlags(&ctx->side.stream, hipStreamNonBlocking));
		HIP_CHECK(hipEventCreateWithFlags(&ctx->side.fork, hipEventDisableTiming));
		HIP_CHECK(hipEventCreateWithFlags(&ctx->side.join, hipEventDisableTiming));
		HIP_CHECK(hipEventCreateWithFlags(&ctx->side.fork2, hipEventDisableTiming));
		HIP_CHECK(hipStreamCreateWithFlags(&ctx->walk_side.stream, hipStreamNonBlocking));
		HIP_CHECK(hipEventCreateWithFlags(&ctx->walk_side.fork, hipEventDisableTiming));
		HIP_CHECK(hipEventCreateWithFlags(&ctx->walk_side.join, hipEventDisableTiming));
		HIP_CHECK(hipEventCreateWithFlags(&ctx->tail_done, hipEventDisableTiming));
		HIP_CHECK(hipEventCreateWithFlags(&ctx->host_wait, hipEventDisableTiming));
		ctx->timer.stream = ctx->stream;
		return ctx.release();
	} catch (const std::exception &e) {
		set_err(err, errlen, e.what());
		return nullptr;
	}
}

void free_resident_graph(ResidentGraph &g)
{
	g = ResidentGraph{}; // the block belongs to the context's graph arena, which keeps it for the next graph
}

extern "C" void povu_hip_destroy(povu_hip_ctx *ctx)
{
	if (!ctx)
		return;
	(void)hipSetDevice(ctx->device);
	ctx->wait_tail();
	free_resident_graph(ctx->g);
	ctx->ws.release();
	ctx->ws2.release();
	ctx->ws_seq.release();
	ctx->ws_walk.release();
	ctx->ws_b.release();
	ctx->upload_tmp.release();
	ctx->shard_buf.release();
	ctx->graph_arena.release();
	ctx->part_arena.release();
	for (auto &kv : ctx->attached)
		(void)munmap(kv.second.p, kv.second.bytes);
	if (ctx->stream)
		(void)hipStreamDestroy(ctx->stream);
	if (ctx->side.stream)
		(void)hipStreamDestroy(ctx->side.stream);
	if (ctx->side.fork)
		(void)hipEventDestroy(ctx->side.fork);
	if (ctx->side.join)
		(void)hipEventDestroy(ctx->side.join);
	if (ctx->side.fork2)
		(void)hipEventDestroy(ctx->side.fork2);
	if (ctx->walk_side.stream)
		(void)hipStreamDestroy(ctx->walk_side.stream);
	if (ctx->walk_side.fork)
		(void)hipEventDestroy(ctx->walk_side.fork);
	if (ctx->walk_side.join)
		(void)hipEventDestroy(ctx->walk_side.join);
	if (ctx->tail_done)
		(void)hipEventDestroy(ctx->tail_done);
	if (ctx->host_wait)
		(void)hipEventDestroy(ctx->host_wait);
	delete ctx;
}

// device block of a resident graph: the link arrays + CSR (off / adj / aoth / atwin), carved from the context's graph
// arena (which only reallocates when a graph is larger than every one before it; the previous graph is gone afterwards)
void alloc_resident_graph(Arena &arena, ResidentGraph &g, uint32_t n_vtx, uint32_t n_links, bool tips_given)
{
	g.V = n_vtx;
	g.E = n_links;
	g.tips_given = tips_given;
	const size_t V = n_vtx, E = n_links;
	const size_t bytes = Arena::padded(V, 4) + 2 * Arena::padded(E + 1, 4) + 2 * Arena::padded(E + 1, 1) + Arena::padded(V, 1) +
			     Arena::padded(2 * V + 2, 4) + 3 * Arena::padded(2 * E + 8, 4) + 16 * 256;
	arena.reserve(bytes);
	g.vid = arena.take<uint32_t>(V);
	g.block = g.vid;
	g.v1 = arena.take<uint32_t>(E + 1);
	g.v2 = arena.take<uint32_t>(E + 1);
	g.s1 = arena.take<uint8_t>(E + 1);
	g.s2 = arena.take<uint8_t>(E + 1);
	g.tip = arena.take<uint8_t>(V);
	g.off = arena.take<uint32_t>(2 * V + 2);
	g.adj = arena.take<uint32_t>(2 * E + 8); // (+8: kernels read a side's slots four at a time)
	g.aoth = arena.take<uint32_t>(2 * E + 8);
	g.atwin = arena.take<uint32_t>(2 * E + 8);
}

void check_graph_size(uint32_t n_vtx, uint32_t n_links)
{
	if (n_vtx == 0)
		throw HipError("graph has no vertices");
	// 32-bit index spaces: the packed list-ranking words hold 30-bit successors -- 3 events per segment and 2 adjacency
	// slots per link must stay below 2^30 (round 3: 2^29).  The reference's own limit is 2V+1 < 2^32 (core.hpp:20-21); at
	// ~54 GB of HBM per 10^8 segments the card's 288 GB run out near 5 * 10^8, just above this limit.
	if (3ull * n_vtx >= (1u << 30) || 2ull * n_links >= (1u << 30))
		throw HipError("graph too large for this build: at most 357 913 941 segments and 536 870 911 links");
}

extern "C" int povu_hip_graph_upload(povu_hip_ctx *ctx, uint32_t n_vtx, const uint32_t *vid, uint32_t n_links,
				     const uint32_t *v1, const uint8_t *s1, const uint32_t *v2, const uint8_t *s2,
				     const uint8_t *tips, char *err, size_t errlen)
{
	ResidentGraph g;
	XferScope xfer(ctx);
	try {
		if (!ctx)
			throw HipError("null context");
		ctx->wait_tail(); // (an overlapped pass may still be reading its workspace)
		// the old graph (or shard) goes first, whatever happens next: a failed upload leaves the context without a graph
		free_resident_graph(ctx->g);
		ctx->have_state = false;
		ctx->shard_comp_ids.clear();
		ctx->shard_total_components = 0;
		check_graph_size(n_vtx, n_links);
		HIP_CHECK(hipSetDevice(ctx->device));
		alloc_resident_graph(ctx->graph_arena, g, n_vtx, n_links, tips != nullptr);
		const size_t V = n_vtx, E = n_links;
		hipStream_t s = ctx->stream;
		hipEvent_t e0, e1;
		HIP_CHECK(hipEventCreate(&e0));
		HIP_CHECK(hipEventCreate(&e1));
		HIP_CHECK(hipEventRecord(e0, s));
		HIP_CHECK(copy_async(g.vid, vid, V * 4, hipMemcpyHostToDevice, s));
		if (E) {
			HIP_CHECK(copy_async(g.v1, v1, E * 4, hipMemcpyHostToDevice, s));
			HIP_CHECK(copy_async(g.v2, v2, E * 4, hipMemcpyHostToDevice, s));
			HIP_CHECK(copy_async(g.s1, s1, E, hipMemcpyHostToDevice, s));
			HIP_CHECK(copy_async(g.s2, s2, E, hipMemcpyHostToDevice, s));
		}
		if (tips)
			HIP_CHECK(copy_async(g.tip, tips, V, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipEventRecord(e1, s));
		HIP_CHECK(hipStreamSynchronize(s));
		(void)hipEventElapsedTime(&g.h2d_ms, e0, e1);
		(void)hipEventDestroy(e0);
		(void)hipEventDestroy(e1);
		build_global_csr(g, ctx->upload_tmp, s); // validates the operands on the device
		ctx->g = g;
		return 0;
	} catch (const std::exception &e) {
		if (ctx)
			(void)hipStreamSynchronize(ctx->stream);
		set_err(err, errlen, e.what());
		return 1;
	}
}

extern "C" int povu_hip_last_upload_times(const povu_hip_ctx *ctx, double out_ms[3])
{
	if (!ctx || !out_ms || !ctx->g.block)
		return 1;
	out_ms[0] = ctx->g.h2d_ms;
	out_ms[1] = ctx->g.csr_ms;
	out_ms[2] = ctx->g.twin_ms;
	return 0;
}


// Workspace carving (or just measuring when `ar` is null), in three parts with different life times:
//   part 0  rows A/B state (CompState), sized before the component count is known (C <= V)
//   part 1  tree / result arrays every execution mode shares, sized with the real component count
//   part 2  the one-lane kernels' own lists (back edges, brackets, stacks): only reserved when the
//           sequential kernels actually run (POVU_HIP_F_SEQUENTIAL / _SEQ_TREE, or a redo)
size_t carve_workspace(Arena *ar, int part, const Sizes &z, CompState &cs, SeqWs &sw, bool hairpins)
{
	size_t total = 0;
	auto take = [&](auto **dst, size_t n, size_t elem) {
		total += Arena::padded(n, elem) + 256;
		if (ar) {
			using P = std::remove_reference_t<decltype(*dst)>;
			*dst = reinterpret_cast<P>(ar->take<char>(n * elem));
		}
	};
	const size_t V = z.V, E = z.E, C = z.Cmax, T = z.T, B = z.B, nS = z.nS;
	if (part == 0) {
		take(&cs.label, V + 1, 4);
		take(&cs.flag, std::max(V, z.slots) + 2, 4);
		take(&cs.crank, V + 2, 4);
		take(&cs.comp_of, V + 1, 4);
		take(&cs.tmp_a, V + 1, 4);
		take(&cs.ckey, V + 1, 4);
		take(&cs.perm, V + 1, 4);
		take(&cs.pos, V + 1, 4);
		take(&cs.voff, C + 2, 4);
		take(&cs.eoff, C + 2, 4);
		take(&cs.vdeg, V + 2, 4);
		take(&cs.sbase, V + 2, 4);
		take(&cs.first, E + 1, 4);
		take(&cs.erank, z.slots + 2, 4);
		take(&cs.ldeg, nS + 2, 4);
		take(&cs.loff, nS + 2, 4);
		take(&cs.ladj, 2 * E + 8, 4); // (+8: the class walk reads a side's list words four at a time)
		take(&cs.keys, 2 * E + 2, 4);
		take(&cs.vals, 2 * E + 2, 4);
		take(&cs.keys2, 2 * E + 2, 4);
		take(&cs.vals2, 2 * E + 2, 4);
		take(&cs.hook, 2 * E + 32, 1);
		take(&cs.la, E + 2, 4);
		take(&cs.lb, E + 2, 4);
		take(&cs.lle, 2 * E + 8, 4); // (+8: the tour kernel reads a segment's slot words four at a time)
		take(&cs.tgray, E + 32, 1);
		take(&cs.stats, 16, 4);
		take(&cs.gid_s, V + 1, 4);
		take(&cs.tip_s, V + 1, 1);
		take(&cs.start_key, C + 2, 8);
		cs.scan_tmp_bytes = scan_tmp_bytes(std::max<size_t>(nS, z.slots) + 2);
		cs.sort_tmp_bytes = sort_tmp_bytes(std::max<size_t>(2 * E, V) + 2);
		take((char **)&cs.scan_tmp, cs.scan_tmp_bytes, 1);
		take((char **)&cs.sort_tmp, cs.sort_tmp_bytes, 1);
	} else if (part == 1) {
		// host-built per-component tables, one upload: order | owner | processed-before | processed | stack entries before
		uint32_t *tables = nullptr;
		take(&tables, 5 * (C + 1), 4);
		sw.order = tables;
		sw.owner = tables + (C + 1);
		sw.tables = tables;
		take(&sw.t_gid, T, 4);
		take(&sw.t_par, T, 4);
		take(&sw.t_cls, T, 4);
		take(&sw.t_size, T + 8, 4); // (+8: the class stage reads sizes eight words at a time)
		take(&sw.t_depth, T, 4);
		take(&sw.t_flags, T, 1);
		take(&sw.cur, nS + 1, 4);
		take(&sw.s_vtx, V + 1, 4);
		take(&sw.s_cls, V + 1, 4);
		take(&sw.next_seen, V + 1, 4);
		if (hairpins)
			take(&sw.hairpins, 2 * (V + C + 1), 8);
		else
			sw.hairpins = nullptr;
		take(&sw.c_ntree, C + 1, 4);
		take(&sw.c_nbe0, C + 1, 4);
		take(&sw.c_nbe, C + 1, 4);
		take(&sw.c_nstack, C + 1, 4);
		take(&sw.c_npvst, C + 1, 4);
		take(&sw.c_nclass, C + 1, 4);
		take(&sw.c_nbry, C + 1, 4);
		take(&sw.c_status, C + 1, 4);
	} else {
		take(&sw.t_hi, T, 4);
		take(&sw.first_child, T, 4);
		take(&sw.next_sib, T, 4);
		take(&sw.last_child, T, 4);
		take(&sw.ctr, nS + 1, 4);
		take(&sw.stk, T, 4);
		take(&sw.selfloop, V + 1, 1);
		take(&sw.be_src, B, 4);
		take(&sw.be_tgt, B, 4);
		take(&sw.o_next, B, 4);
		take(&sw.i_next, B, 4);
		take(&sw.b_prev, B, 4);
		take(&sw.b_next, B, 4);
		take(&sw.b_rsize, B, 4);
		take(&sw.b_rclass, B, 4);
		take(&sw.be_type, B, 1);
		take(&sw.b_in, B, 1);
		take(&sw.be_cdef, B, 1);
		take(&sw.o_head, T, 4);
		take(&sw.o_tail, T, 4);
		take(&sw.i_head, T, 4);
		take(&sw.i_tail, T, 4);
		take(&sw.l_head, T, 4);
		take(&sw.l_tail, T, 4);
		take(&sw.l_size, T, 4);
		take(&sw.bl, T, 4);
		take(&sw.nxt, T, 4);
		take(&sw.st_head, T, 4);
		take(&sw.st_tail, T, 4);
		take(&sw.last, B + T, 4);
		take(&sw.p_parent, V + C + 1, 4);
		take(&sw.p_a, V + C + 1, 4);
		take(&sw.p_z, V + C + 1, 4);
		take(&sw.p_or, V + C + 1, 1);
		take(&sw.aux, V + C + 1, 4);
		take(&sw.in_s, B + T, 1);
	}
	return total + (1 << 20);
}

size_t rowb_carve_label(Arena *ar, const Sizes &z, CompState &cs)
{
	size_t total = 0;
	auto take = [&](auto **dst, size_t n, size_t elem) {
		total += Arena::padded(n, elem) + 256;
		if (ar) {
			using P = std::remove_reference_t<decltype(*dst)>;
			*dst = reinterpret_cast<P>(ar->take<char>(n * elem));
		}
	};
	const size_t V = z.V, E = z.E;
	take(&cs.label, V + 1, 4);
	take(&cs.flag, V / 4 + 16, 4); // (one byte per vertex: is-root flags)
	take(&cs.crank, V + 2, 4);
	take(&cs.comp_of, V + 1, 4);
	take(&cs.keys, 2 * E + 2, 4); // the cross list of the union-find tiles: [E] pairs
	take(&cs.hook, 2 * E + 32, 1);
	take(&cs.stats, 16, 4);
	cs.scan_tmp_bytes = scan_tmp_bytes(std::max<size_t>(z.nS, z.slots) + 2);
	take((char **)&cs.scan_tmp, cs.scan_tmp_bytes, 1);
	if (ar) { // (what rowb_carve_reindex adds; null until then so that a use before it is a clean failure, not a stale pointer)
		cs.tmp_a = cs.ckey = cs.perm = cs.pos = cs.voff = cs.eoff = cs.vdeg = cs.sbase = cs.first = cs.erank = cs.ldeg = cs.loff =
			cs.ladj = cs.vals = cs.keys2 = cs.vals2 = cs.la = cs.lb = cs.lle = cs.gid_s = nullptr;
		cs.tgray = cs.tip_s = nullptr;
		cs.start_key = nullptr;
		cs.sort_tmp = nullptr;
		cs.sort_tmp_bytes = 0;
	}
	return total + (1 << 16);
}

// what the re-index arena must hold already for the adjacency kernel to be started ahead of the component count
size_t rowb_speculative_adj_bytes(const Sizes &z) { return 2 * (Arena::padded(2 * z.E + 8, 4) + 256) + 4096; }

size_t rowb_carve_reindex(Arena *ar, const Sizes &z, size_t C, const RowBNeeds &need, CompState &cs)
{
	size_t total = 0;
	auto take = [&](auto **dst, size_t n, size_t elem) {
		total += Arena::padded(n, elem) + 256;
		if (ar) {
			using P = std::remove_reference_t<decltype(*dst)>;
			*dst = reinterpret_cast<P>(ar->take<char>(n * elem));
		}
	};
	const size_t V = z.V, E = z.E, nS = z.nS;
	const bool lean = need.identity && need.sort_free; // sorted space = global vertex space, nothing is renumbered
	// (ladj and lle FIRST: their place in the arena does not depend on the component count -- povu_hip_decompose starts the
	// kernel that fills them before the count has reached the host, rowb_speculative_adj_bytes)
	take(&cs.ladj, 2 * E + 8, 4); // (+8: the class walk reads a side's list words four at a time)
	take(&cs.lle, 2 * E + 8, 4);  // (+8: the tour kernel reads a segment's slot words four at a time)
	take(&cs.voff, C + 2, 4);
	take(&cs.eoff, C + 2, 4);
	take(&cs.start_key, C + 2, 8);
	if (!lean) { // the vertices are renumbered (or the sorting builder wants the tables anyway)
		take(&cs.tmp_a, V + 1, 4);
		take(&cs.ckey, V + 1, 4);
		take(&cs.perm, V + 1, 4);
		take(&cs.pos, V + 1, 4);
		take(&cs.vdeg, V + 2, 4);
		take(&cs.sbase, V + 2, 4);
		take(&cs.gid_s, V + 1, 4);
		take(&cs.tip_s, V + 1, 1);
	}
	if (!(lean && !need.self_loops)) { // local degrees and offsets of the sides (else the CSR's own)
		take(&cs.ldeg, nS + 2, 4);
		take(&cs.loff, nS + 2, 4);
	}
	if (!need.sort_free) { // the builder that numbers the local edges densely (hub vertices; povu_hip_componetize takes the full set)
		take(&cs.flag, std::max(V, z.slots) + 2, 4);
		take(&cs.first, E + 1, 4);
		take(&cs.erank, z.slots + 2, 4);
		take(&cs.vals, 2 * E + 2, 4);
		take(&cs.keys2, 2 * E + 2, 4);
		take(&cs.vals2, 2 * E + 2, 4);
		take(&cs.la, E + 2, 4);
		take(&cs.lb, E + 2, 4);
		take(&cs.tgray, E + 32, 1);
	}
	if (!need.identity || !need.sort_free) {
		cs.sort_tmp_bytes = sort_tmp_bytes(std::max<size_t>(2 * E, V) + 2);
		take((char **)&cs.sort_tmp, cs.sort_tmp_bytes, 1);
	}
	return total + (1 << 16);
}

namespace
{
struct ComponentsOwner {
	povu_hip_components view{};
	std::vector<uint32_t> voff, eoff, vid, v1, v2;
	std::vector<uint8_t> tip, s1, s2;
};
} // namespace

extern "C" povu_hip_components *povu_hip_componetize(povu_hip_ctx *ctx, char *err, size_t errlen)
{
	try {
		if (!ctx || !ctx->g.block)
			throw HipError("no graph resident: call povu_hip_graph_upload first");
		HIP_CHECK(hipSetDevice(ctx->device));
		ctx->wait_tail();
		const ResidentGraph &g = ctx->g;
		hipStream_t s = ctx->stream;
		Sizes z;
		z.V = g.V;
		z.E = g.E;
		z.Cmax = g.V;
		z.nS = 2 * z.V;
		z.slots = g.n_slots;
		z.T = z.B = 0;
		CompState &cs = ctx->cs;
		SeqWs &sw = ctx->sw;
		ctx->have_state = false;
		ctx->host.reset();
		cs.host = &ctx->host;
		cs.host_pub = nullptr;
		ctx->ws.reserve(carve_workspace(nullptr, 0, z, cs, sw, false));
		carve_workspace(&ctx->ws, 0, z, cs, sw, false);
		StageTimer &tm = ctx->timer;
		tm.reset();
		tm.enabled = true;
		const uint32_t C = label_components(g, cs, tm, s);
		reindex_components(g, cs, C, tm, s, true); // (the builder that numbers the local edges: la / lb in local edge order)
		if (!cs.dense_edges)
			throw HipError("componetize: local edges were not numbered (internal)");
		auto o = std::make_unique<ComponentsOwner>();
		const size_t V = g.V, E = g.E;
		o->voff.resize(C + 1);
		o->eoff.resize(C + 1);
		o->vid.resize(V);
		o->tip.resize(V);
		std::vector<uint32_t> la(E), lb(E);
		HIP_CHECK(copy_async(o->voff.data(), cs.voff, (size_t)(C + 1) * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(copy_async(o->eoff.data(), cs.eoff, (size_t)(C + 1) * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(copy_async(o->vid.data(), cs.gid_s, V * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(copy_async(o->tip.data(), cs.tip_s, V, hipMemcpyDeviceToHost, s));
		if (E) {
			HIP_CHECK(copy_async(la.data(), cs.la, E * 4, hipMemcpyDeviceToHost, s));
			HIP_CHECK(copy_async(lb.data(), cs.lb, E * 4, hipMemcpyDeviceToHost, s));
		}
		HIP_CHECK(hipStreamSynchronize(s));
		o->v1.resize(E);
		o->v2.resize(E);
		o->s1.resize(E);
		o->s2.resize(E);
		for (uint32_t c = 0; c < C; c++)
			for (uint32_t e = o->eoff[c]; e < o->eoff[c + 1]; e++) { // sorted side id = 2 * position + end
				o->v1[e] = (la[e] >> 1) - o->voff[c];
				o->s1[e] = (uint8_t)(la[e] & 1);
				o->v2[e] = (lb[e] >> 1) - o->voff[c];
				o->s2[e] = (uint8_t)(lb[e] & 1);
			}
		o->view = povu_hip_components{C, g.V, g.E, o->voff.data(), o->eoff.data(), o->vid.data(), o->tip.data(),
					      o->v1.data(), o->v2.data(), o->s1.data(), o->s2.data()};
		ComponentsOwner *raw = o.release();
		return &raw->view; // view is the first member: the owner is recovered from it in _free
	} catch (const std::exception &e) {
		set_err(err, errlen, e.what());
		return nullptr;
	}
}

extern "C" void povu_hip_components_free(povu_hip_components *c)
{
	delete reinterpret_cast<ComponentsOwner *>(c);
}

// What the stage workspace of a pass has to hold (par_kernels.hpp: StageWsOpts).  The plain all-parallel pass needs neither
// the per-vertex tables of a sequential tree stage, nor the keys of its bracket sort, nor the hairpin report's arrays, and
// takes the wave walk's arrays only when it meets large classes.  Brackets: every link outside the spanning forest of
// the segments starts at most one back edge (E - (V - C) of them), every side without links at most one to the root
// (spanning_tree.cpp:433-438), and the class stage adds at most one capping or simplifying edge per tree vertex.
static StageWsOpts stage_opts(size_t V, size_t E, size_t C, size_t empty_sides, bool seq_tree, bool hairpins)
{
	StageWsOpts o;
	const size_t T = 2 * V + C;
	o.nb_cap = std::min(E + V + T, (E + C > V ? E + C - V : 0) + empty_sides + T + 64);
	o.full_t = seq_tree || hairpins;
	o.sorted_brackets = seq_tree;
	o.hairpins = hairpins;
	o.walk_inline = false;
	return o;
}

extern "C" uint64_t povu_hip_workspace_estimate(uint32_t n_vtx, uint32_t n_links, uint32_t n_components)
{
	try {
		Sizes z;
		z.V = n_vtx;
		z.E = n_links;
		z.nS = 2 * z.V;
		z.slots = 2 * z.E;
		CompState cs{};
		SeqWs sw{};
		z.Cmax = n_vtx;
		z.T = z.B = 0;
		// (rows A/B of a graph whose vertices come grouped by component, without hub vertices or self loops: the pangenome
		// case; povu_hip_workspace_breakdown's out[6] has the general case)
		const RowBNeeds lean{true, true, false};
		uint64_t total = rowb_carve_label(nullptr, z, cs) + rowb_carve_reindex(nullptr, z, n_components ? n_components : n_vtx, lean, cs);
		z.Cmax = n_components ? n_components : n_vtx;
		z.T = 2 * z.V + z.Cmax;
		z.B = z.E + z.V + 2 * z.T;
		// (sides without links: two per component and a few more -- an estimate; a graph full of tips reserves more)
		total += carve_workspace(nullptr, 1, z, cs, sw, false) +
			 stage_workspace_bytes(z.V, z.E, z.Cmax, stage_opts(z.V, z.E, z.Cmax, 2 * z.Cmax + z.V / 64, false, false));
		return total;
	} catch (...) {
		return 0;
	}
}

extern "C" int povu_hip_workspace_breakdown(uint32_t n_vtx, uint32_t n_links, uint32_t n_components, uint64_t out[7])
{
	try {
		Sizes z;
		z.V = n_vtx, z.E = n_links, z.nS = 2 * z.V, z.slots = 2 * z.E;
		CompState cs{};
		SeqWs sw{};
		z.Cmax = n_vtx;
		z.T = z.B = 0;
		const RowBNeeds lean{true, true, false}, general{false, false, true};
		const size_t Cn = n_components ? n_components : n_vtx;
		out[0] = rowb_carve_label(nullptr, z, cs) + rowb_carve_reindex(nullptr, z, Cn, lean, cs);
		out[6] = rowb_carve_label(nullptr, z, cs) + rowb_carve_reindex(nullptr, z, Cn, general, cs);
		z.Cmax = Cn;
		z.T = 2 * z.V + z.Cmax;
		z.B = z.E + z.V + 2 * z.T;
		out[1] = carve_workspace(nullptr, 1, z, cs, sw, false);
		const StageWsOpts so = stage_opts(z.V, z.E, z.Cmax, 2 * z.Cmax + z.V / 64, false, false);
		out[2] = par_workspace_bytes(z.V, z.E, z.Cmax, 1, so);
		out[3] = par_workspace_bytes(z.V, z.E, z.Cmax, 2, so);
		out[4] = tree_workspace_bytes(z.V, z.E, z.Cmax, 1, so);
		out[5] = tree_workspace_bytes(z.V, z.E, z.Cmax, 2, so);
		return 0;
	} catch (...) {
		return 1;
	}
}

// Reserve, ahead of time, the device arenas a graph of this size will need (the resident graph, the CSR build's scratch and
// the decompose workspace for a graph of few components): the first upload + decompose on a fresh context otherwise
// pay for ~1 KB of device memory per segment being mapped (0.3 s and more for a whole-genome graph).  The CLI calls this on
// the thread that brought HIP up, as soon as the tokenizer knows the counts and while the rest of the parse runs.
// Best effort: when that does not fit, the arenas are left alone and the real calls allocate exactly.
extern "C" int povu_hip_prewarm(povu_hip_ctx *ctx, uint32_t n_vtx, uint32_t n_links, char *err, size_t errlen)
{
	try {
		if (!ctx)
			throw HipError("null context");
		ctx->wait_tail();
		if (ctx->g.block || ctx->have_state)
			return 0; // (only for a context that holds nothing yet: a reserve invalidates what an arena holds)
		check_graph_size(n_vtx, n_links);
		HIP_CHECK(hipSetDevice(ctx->device));
		const size_t V = n_vtx, E = n_links, nS = 2 * V;
		size_t free_b = 0, total_b = 0;
		HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
		const size_t graph_b = Arena::padded(V, 4) + 2 * Arena::padded(E + 1, 4) + 2 * Arena::padded(E + 1, 1) + Arena::padded(V, 1) +
				       Arena::padded(2 * V + 2, 4) + 3 * Arena::padded(2 * E + 8, 4) + 16 * 256;
		const size_t tmp_b = Arena::padded(2 * E + 2, 4) * 4 + Arena::padded(nS + 2, 4) + sort_tmp_bytes(2 * E) +
				     scan_tmp_bytes(std::max<size_t>(nS, E) + 2) + (1 << 16);
		Sizes z;
		// (components: a guess -- a pangenome graph has few, and a larger count only means that the decompose call grows
		// its arena after all; reserving for V components took three times as long as the exact size)
		z.V = V, z.E = E, z.nS = nS, z.slots = 2 * E, z.Cmax = V, z.T = z.B = 0; // (rows A/B are sized before the count is known)
		CompState cs{};
		SeqWs sw{};
		z.Cmax = std::min<size_t>(V, std::max<size_t>(1024, V / 64));
		const size_t ws_b = rowb_carve_label(nullptr, z, cs), ws_b2 = rowb_carve_reindex(nullptr, z, z.Cmax, RowBNeeds{true, true, false}, cs);
		z.T = 2 * V + z.Cmax;
		z.B = E + V + 2 * z.T;
		const size_t ws2_b = carve_workspace(nullptr, 1, z, cs, sw, false) +
				     stage_workspace_bytes(V, E, z.Cmax, stage_opts(V, E, z.Cmax, 2 * z.Cmax + V / 64, false, false));
		const size_t need = graph_b + tmp_b + ws_b + ws_b2 + ws2_b;
		if (need + need / 8 + (size_t(64) << 20) > free_b)
			return 0; // the worst case does not fit beside what is there: let the real calls size things
		// (exact sizes: a context that is warmed for one graph is usually there for that graph only)
		ctx->graph_arena.reserve(graph_b, false);
		ctx->upload_tmp.reserve(tmp_b, false);
		ctx->ws.reserve(ws_b, false);
		ctx->ws_b.reserve(ws_b2, false);
		ctx->ws2.reserve(ws2_b, false);
		return 0;
	} catch (const std::exception &e) {
		set_err(err, errlen, e.what());
		return 1;
	}
}

extern "C" uint64_t povu_hip_leaf_workspace_estimate(uint32_t n_vtx, uint32_t n_components)
{
	// the PVST of a component has at most one vertex per segment and a root
	return leaf_workspace_bytes(n_vtx, n_components ? n_components : n_vtx, (size_t)n_vtx + (n_components ? n_components : n_vtx));
}

extern "C" povu_hip_forest *povu_hip_decompose(povu_hip_ctx *ctx, const povu_hip_opts *opts, char *err, size_t errlen)
{
	// declared outside the try block: on a failure the stream is drained BEFORE the forest returns its pinned
	// result block to the pool (kernels that write into it may still be queued)
	std::unique_ptr<povu_hip_forest> f;
	XferScope xfer(ctx);
	try {
		if (ctx && !ctx->g.block && ctx->shard_total_components) { // a shard without components: nothing to do
			f = std::make_unique<povu_hip_forest>();
			f->pool = ctx->pool;
			f->total_components = ctx->shard_total_components;
			ctx->last_times.clear();
			ctx->last_links = 0;
			return f.release();
		}
		if (!ctx || !ctx->g.block)
			throw HipError("no graph resident: call povu_hip_graph_upload first");
		HIP_CHECK(hipSetDevice(ctx->device));
		const ResidentGraph &g = ctx->g;
		hipStream_t s = ctx->stream;
		povu_hip_opts o{0, 1, 0};
		if (opts)
			o = *opts;
		if (o.world == 0)
			o.world = 1;
		if (o.rank >= o.world)
			throw HipError("shard rank >= world");
		const bool hairpins = (o.flags & POVU_HIP_F_HAIRPINS) != 0;
		// POVU_HIP_F_ASYNC: the pass may leave its last kernels and the copies of the PVST arrays in flight when it returns
		// (and the next pass may then start under them).  Only the plain all-parallel pass has that form.
		const bool want_async = (o.flags & POVU_HIP_F_ASYNC) && (o.flags & POVU_HIP_F_NO_STAGE_TIMES) && !hairpins &&
					!(o.flags & (POVU_HIP_F_SEQUENTIAL | POVU_HIP_F_SEQ_TREE | POVU_HIP_F_FORCE_REDO | POVU_HIP_F_REDO_ODD |
						     POVU_HIP_F_LEAF_SUBFLUBBLES | POVU_HIP_F_SUBFLUBBLES | POVU_HIP_F_CHECK_LAMINAR));
		// The tail of the pass before (POVU_HIP_F_ASYNC) reads the stage workspace (ws2) from the side stream.  A pass that may
		// itself overlap waits for it ON THE STREAM, right before its own first write there (below); every other pass -- and
		// any pass whose arenas must grow, which frees them -- waits here.
		auto reserve = [&](Arena &ar, size_t bytes) {
			if (bytes > ar.capacity())
				ctx->wait_tail();
			ar.reserve(bytes);
		};
		if (!want_async)
			ctx->wait_tail();

		Sizes z;
		z.V = g.V;
		z.E = g.E;
		z.Cmax = g.V; // rows A/B run before the component count is known
		z.nS = 2 * z.V;
		z.slots = g.n_slots;
		z.T = z.B = 0;
		CompState &cs = ctx->cs;
		SeqWs &sw = ctx->sw;
		ctx->have_state = false;
		ctx->last_mixed = false;
		ctx->redo_pvst_only = false;
		ctx->stack_export_pending = false;
		ctx->classes_in_par = false;
		ctx->tree_in_par = false;
		ctx->host.reset();
		cs.host = ctx->pw.host = ctx->tw.host = &ctx->host;
		const bool all_seq = (o.flags & POVU_HIP_F_SEQUENTIAL) != 0;
		const bool all_sub = (o.flags & POVU_HIP_F_SUBFLUBBLES) != 0; // all five passes of -s
		const bool leaf_sub = all_sub || (o.flags & POVU_HIP_F_LEAF_SUBFLUBBLES) != 0;
		LeafState leaf_state;
		if (leaf_sub && (o.flags & (POVU_HIP_F_SEQUENTIAL | POVU_HIP_F_SEQ_TREE)))
			throw HipError("the leaf subflubble passes read the state of the parallel stages: not with the sequential tree / all-sequential test modes");
		reserve(ctx->ws, rowb_carve_label(nullptr, z, cs));
		rowb_carve_label(&ctx->ws, z, cs);

		StageTimer &tm = ctx->timer;
		tm.reset();
		tm.enabled = (o.flags & POVU_HIP_F_NO_STAGE_TIMES) == 0;
		// the forest owns the two events that time its pass: ev0 at the first kernel, ev1 behind the last byte that reaches the host
		f = std::make_unique<povu_hip_forest>();
		HIP_CHECK(hipEventCreate(&f->ev0));
		HIP_CHECK(hipEventCreate(&f->ev1));
		HIP_CHECK(hipEventRecord(f->ev0, s));

		// ---- row B
		// The host has to know the component count (it sizes the workspaces) and then the components' sizes (the tables of the
		// stages): two reads.  Neither is waited for with the stream idle: the words are published by a kernel, an EVENT is
		// recorded behind it, and the stream is given its next kernel before the host waits for the event --
		//  (1) behind the labelling: the re-index's adjacency kernel, on the assumption that holds for nearly every GFA (vertices
		//      grouped by component, no hub, no self loop: it then needs nothing the labels say but the hooks; when the
		//      assumption fails its output is simply overwritten by the real re-index);
		//  (2) behind the re-index: the tree stage's first kernel (it reads the re-indexed adjacency only).
		// Not with stage timers (the kernels would be booked on the wrong stage).
		const bool force_sorted = (o.flags & POVU_HIP_F_SORTED_ADJ) != 0;
		const bool par_tree = !all_seq && !(o.flags & POVU_HIP_F_SEQ_TREE);
		uint32_t *lab = label_components_enqueue(g, cs, tm, s);
		HIP_CHECK(hipEventRecord(ctx->host_wait, s));
		bool spec_adj = false;
		if (!tm.enabled && g.E && sort_free_adjacency(g, force_sorted) && ctx->ws_b.capacity() >= rowb_speculative_adj_bytes(z)) {
			ctx->ws_b.reserve(0); // (rewinds the arena: the two arrays get the places rowb_carve_reindex will give them)
			uint32_t *ladj = ctx->ws_b.take<uint32_t>(2 * z.E + 8), *lle = ctx->ws_b.take<uint32_t>(2 * z.E + 8);
			reindex_speculative_adj(g, cs, ladj, lle, s);
			spec_adj = true;
		}
		HIP_CHECK(hipEventSynchronize(ctx->host_wait));
		const uint32_t C = label_components_finish(cs, lab);
		{ // what the re-index of THIS graph needs, now that the count, the order and the self loops are known
			const RowBNeeds need{C == 1 || cs.comp_sorted, sort_free_adjacency(g, force_sorted), cs.has_self_loops};
			const size_t need_b = rowb_carve_reindex(nullptr, z, C, need, cs);
			if (spec_adj && need_b > ctx->ws_b.capacity()) { // the arena has to grow under a kernel that writes into it: let it finish, forget it
				HIP_CHECK(hipStreamSynchronize(s));
				spec_adj = false;
			}
			reserve(ctx->ws_b, need_b);
			rowb_carve_reindex(&ctx->ws_b, z, C, need, cs);
			spec_adj = spec_adj && need.identity && need.sort_free && !need.self_loops; // (else: a kernel that wrote nonsense into arrays about to be rewritten)
		}
		// component sizes on the host (shard assignment = LPT over link counts, launch order): the last
		// re-index kernel writes them into pinned memory itself
		uint32_t *pub = ctx->host.take<uint32_t>(2 * ((size_t)C + 1) + 4);
		HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&cs.host_pub), pub, 0));
		count_kernel_d2h((2 * ((size_t)C + 1) + 4) * 4);
		reindex_components(g, cs, C, tm, s, force_sorted, spec_adj);
		HIP_CHECK(hipEventRecord(ctx->host_wait, s));
		const uint32_t *voff = pub, *eoff = pub + (size_t)C + 1, *gstats = pub + 2 * ((size_t)C + 1);
		// stage workspaces, sized with the real component count
		z.Cmax = C;
		z.T = 2 * z.V + C;
		z.B = z.E + z.V + 2 * z.T;
		const StageWsOpts so = stage_opts(z.V, z.E, C, g.n_empty_sides, (o.flags & POVU_HIP_F_SEQ_TREE) != 0, hairpins);
		reserve(ctx->ws2, carve_workspace(nullptr, 1, z, cs, sw, hairpins) + (all_seq ? 0 : stage_workspace_bytes(z.V, z.E, C, so)));
		carve_workspace(&ctx->ws2, 1, z, cs, sw, hairpins);
		bool tail_waited = false;
		if (!all_seq) {
			stage_workspace_carve(ctx->ws2, ctx->pw, ctx->tw, z.V, z.E, C, so);
			ctx->tw.walk_arena = &ctx->ws_walk;
			ctx->tw.walk_stream = ctx->walk_side.stream;
			ctx->tw.walk_fork = ctx->walk_side.fork;
			ctx->tw.walk_join = ctx->walk_side.join;
			ctx->tw.tour_words_done = false;
			if (par_tree && !tm.enabled) {
				if (ctx->tail_pending) { // first write into the stage workspace: behind the tail of the pass before
					HIP_CHECK(hipStreamWaitEvent(s, ctx->tail_done, 0));
					tail_waited = true;
				}
				cs.host = ctx->pw.host = ctx->tw.host = &ctx->host;
				tree_tour_words(cs, (uint32_t)z.V, (uint32_t)z.E, ctx->tw, (o.flags & POVU_HIP_F_SPARSE_SPLITTERS) != 0, s);
			}
		}
		HIP_CHECK(hipEventSynchronize(ctx->host_wait)); // the component sizes are in `pub`
		bool seq_ws_ready = false;
		auto need_seq_workspace = [&]() { // the one-lane kernels' lists live in their own arena
			if (seq_ws_ready)
				return;
			ctx->wait_tail();
			ctx->ws_seq.reserve(carve_workspace(nullptr, 2, z, cs, sw, hairpins));
			carve_workspace(&ctx->ws_seq, 2, z, cs, sw, hairpins);
			seq_ws_ready = true;
		};
		// host-built tables live in pinned scratch: their uploads need no synchronisation
		uint32_t *tab_h = ctx->host.take<uint32_t>(5 * ((size_t)C + 1));
		uint32_t *order = tab_h, *owner = tab_h + ((size_t)C + 1), *pc = tab_h + 2 * ((size_t)C + 1),
			 *cproc = tab_h + 3 * ((size_t)C + 1), *stack_off = tab_h + 4 * ((size_t)C + 1);
		std::iota(order, order + C, 0u);
		std::fill(owner, owner + C, 0u);
		auto weight = [&](uint32_t c) { return (uint64_t)(eoff[c + 1] - eoff[c]) + (voff[c + 1] - voff[c]); };
		// heaviest first: the shard assignment below and the launch order of the one-lane kernels (the parallel
		// stages do not care, so a single-shard parallel pass skips the sort)
		if (o.world > 1 || (o.flags & (POVU_HIP_F_SEQUENTIAL | POVU_HIP_F_SEQ_TREE | POVU_HIP_F_FORCE_REDO)))
			std::stable_sort(order, order + C, [&](uint32_t a, uint32_t b) { return weight(a) > weight(b); });
		if (o.world > 1) { // greedy longest-processing-time assignment, deterministic on every rank
			std::vector<uint64_t> load(o.world, 0);
			for (uint32_t k = 0; k < C; k++) {
				const uint32_t c = order[k];
				uint32_t best = 0;
				for (uint32_t r = 1; r < o.world; r++)
					if (load[r] < load[best])
						best = r;
				owner[c] = best;
				load[best] += weight(c) + 1;
			}
		}
		uint64_t links = 0;
		for (uint32_t c = 0; c < C; c++)
			if (owner[c] == o.rank || o.world == 1)
				links += eoff[c + 1] - eoff[c];
		ctx->last_links = links;
		// processed components (>= 3 vertices, owned by this shard), their running count = where a component
		// starts in the dense PVST output, and the number of event lists of the pre-order ranking
		uint32_t event_lists = 0, n_stack = 0;
		order[C] = owner[C] = cproc[C] = 0;
		pc[0] = 0;
		for (uint32_t c = 0; c < C; c++) {
			const uint32_t nv = voff[c + 1] - voff[c];
			cproc[c] = (nv >= 3 && (o.world == 1 || owner[c] == o.rank)) ? 1u : 0u;
			pc[c + 1] = pc[c] + cproc[c];
			event_lists += cproc[c] ? 1u : 2 * nv;
			stack_off[c] = n_stack;
			n_stack += cproc[c] ? nv : 0u; // one candidate-stack entry per segment (its black tree edge)
		}
		stack_off[C] = n_stack;
		const uint32_t n_processed = pc[C];
		if (ctx->tail_pending && !tail_waited) // first write into the stage workspace: behind the tail of the pass before (see above)
			HIP_CHECK(hipStreamWaitEvent(s, ctx->tail_done, 0));
		HIP_CHECK(copy_async(sw.tables, tab_h, 5 * ((size_t)C + 1) * 4, hipMemcpyHostToDevice, s));

		// ---- rows C-G
		sw.V = g.V;
		sw.E = g.E;
		sw.C = C;
		sw.rank = o.rank;
		sw.world = o.world;
		sw.flags = o.flags;
		sw.want_depth = all_sub;
		sw.voff = cs.voff;
		sw.eoff = cs.eoff;
		sw.loff = cs.loff;
		sw.ladj = cs.ladj;
		sw.gid_s = cs.gid_s;
		sw.tip_s = cs.tip_s;
		sw.start_key = cs.start_key;
		sw.p_ai = sw.p_zi = nullptr;
		const size_t T = 2 * (size_t)g.V + C, B = (size_t)g.E + g.V + 2 * T;
		// the one-lane kernels expect their lists empty; only paid for when they actually run
		auto init_seq_workspace = [&]() {
			need_seq_workspace();
			for (uint32_t *p : {sw.first_child, sw.o_head, sw.i_head, sw.bl, sw.t_hi, sw.t_cls, sw.st_head, sw.st_tail})
				HIP_CHECK(hipMemsetAsync(p, 0xFF, T * 4, s));
			HIP_CHECK(hipMemsetAsync(sw.ctr, 0xFF, z.nS * 4, s));
			HIP_CHECK(hipMemsetAsync(sw.cur, 0, z.nS * 4, s));
			HIP_CHECK(hipMemsetAsync(sw.selfloop, 0, g.V, s));
			HIP_CHECK(hipMemsetAsync(sw.last, 0xFF, (B + T) * 4, s));
			HIP_CHECK(hipMemsetAsync(sw.in_s, 0, B + T, s));
		};
		// the result block is allocated as soon as the number of PVST vertices is known (before the emit kernel):
		// the parallel stages write it straight into pinned host memory, there is no device-to-host copy
		f->pool = ctx->pool;
		f->meta_reserve = (size_t)C + 1; // room behind the arrays for the tree table of povu_hip_forest_share
		auto alloc_result_block = [&](size_t total) -> void * {
			f->release_block();
			f->alloc(total);
			void *dev = nullptr;
			HIP_CHECK(hipHostGetDevicePointer(&dev, f->block, 0));
			return dev;
		};
		const uint32_t *sum = nullptr; // outcome of the pass in pinned memory (pass_summary)
		PassTail tail;
		bool fast_tail = false; // the summary came with the PVST count: nothing was synchronised after the tail was enqueued
		auto read_summary = [&](bool with_par) -> const uint32_t * {
			uint32_t *h = ctx->host.take<uint32_t>(5 * (size_t)C + 8);
			count_kernel_d2h((5 * (size_t)C + 8) * 4);
			pass_summary(sw, with_par ? &ctx->pw : nullptr, C, h, s);
			HIP_CHECK(hipEventRecord(f->ev1, s)); // (recorded again if more work follows)
			HIP_CHECK(hipStreamSynchronize(s));
			return h;
		};
		tm.begin("traversal_init");
		zero_component_counters(sw, C, all_seq ? nullptr : ctx->pw.comp_bad, all_seq ? nullptr : ctx->pw.err, s);
		if (all_seq)
			init_seq_workspace();
		tm.end(all_seq ? 14 : 1);
		sw.comp_sel = nullptr;
		ctx->last_seq_redo = 0;
		bool mixed = false; // parallel result for most components, sequential redo for the flagged ones
		if (all_seq) {
			tm.begin("traversal_seq");
			sw.stages = SEQ_STAGE_ALL;
			launch_seq_components(sw, s);
			tm.end(1);
		} else {
			int64_t dense_nb0 = -1;
			ctx->pw.cproc_ps = sw.tables + 2 * ((size_t)C + 1);
			ctx->tw.cproc = sw.tables + 3 * ((size_t)C + 1);
			ctx->pw.soff = sw.tables + 4 * ((size_t)C + 1);
			if (o.flags & POVU_HIP_F_SEQ_TREE) {
				init_seq_workspace();
				tm.begin("tree_seq");
				sw.stages = SEQ_STAGE_TREE;
				launch_seq_components(sw, s);
				tm.end(1);
			} else {
				dense_nb0 = run_parallel_tree(cs, sw, ctx->pw, ctx->tw, C, event_lists, gstats[0],
							      (o.flags & POVU_HIP_F_BIG_CLASS_DFS) != 0,
							      (o.flags & POVU_HIP_F_SPARSE_SPLITTERS) != 0, tm, s);
				ctx->tree_in_par = true;
			}
			ctx->pw.all_vertex_classes = (o.flags & POVU_HIP_F_ALL_VERTEX_CLASSES) != 0;
			ctx->pw.check_laminar = (o.flags & POVU_HIP_F_CHECK_LAMINAR) != 0;
			tail.want_overlap = want_async;
			tail.done = f->ev1;
			tail.done2 = ctx->tail_done;
			run_parallel_dg(cs, sw, ctx->pw, C, n_processed, n_stack, dense_nb0, alloc_result_block, tm, s, ctx->side, tail);
			ctx->stack_export_pending = true;
			ctx->classes_in_par = true;
			if (o.flags & POVU_HIP_F_REDO_ODD) // (tests: flag every other component as if its stack were not laminar)
				mark_odd_u32(ctx->pw.comp_bad, C, s);
			// Everything the host needs came back with the PVST count, unless something after it can still flag a component
			// (the laminarity check, the test modes) or add to the result (labels, boundaries): then the summary is read again
			// when all of that is done.
			fast_tail = tail.summary_final && !leaf_sub && !hairpins && !(o.flags & (POVU_HIP_F_FORCE_REDO | POVU_HIP_F_REDO_ODD));
			sum = fast_tail ? tail.early_summary : read_summary(true);
			if (sum[0])
				throw HipError("parallel class stage: a tree vertex has no live bracket (internal invariant broken)");
			if (sum[1] & 1u)
				throw HipError("list ranking: splitter capacity exceeded (internal sizing bug)");
			if (sum[1] & 2u)
				throw HipError("class walk: stack pool exhausted (internal sizing bug)");
			if (sum[2])
				throw HipError("spanning forest of the links has the wrong size (internal)");
			if (sum[3])
				throw HipError("candidate stack has the wrong size (internal)");
			uint32_t nbad = 0;
			for (uint32_t c = 0; c < C; c++)
				nbad += sum[4 + c] ? 1 : 0;
			if (leaf_sub) {
				// find_tiny + find_parallel relabel leaf flubbles (leaf_kernels.hip): the PVSTs the parallel stages just
				// emitted here, those of components that go through the redo of add_flubbles after it (below)
				if (nbad && hairpins)
					throw HipError("leaf subflubble passes: with --hairpins a component that needs the sequential redo is rebuilt "
						       "from scratch by the one-lane kernels, whose tree state the passes do not read");
				tm.begin("leaf_subflubbles");
				leaf_prepare(cs, sw, ctx->pw, ctx->tw, C, ctx->ws_leaf, leaf_state, s);
				leaf_dense(leaf_state, sw, ctx->pw, C, s);
				const size_t n = ctx->pw.d_total;
				f->sub_ai.resize(n, ctx->pool); // (page-locked, out of the context's pool)
				f->sub_zi.resize(n, ctx->pool);
				f->sub_fam.resize(n, ctx->pool);
				if (n) {
					HIP_CHECK(copy_async(f->sub_ai.data(), leaf_state.dense.ai, n * 4, hipMemcpyDeviceToHost, s));
					HIP_CHECK(copy_async(f->sub_zi.data(), leaf_state.dense.zi, n * 4, hipMemcpyDeviceToHost, s));
					HIP_CHECK(copy_async(f->sub_fam.data(), leaf_state.dense.fam, n, hipMemcpyDeviceToHost, s));
				}
				tm.end(16);
				HIP_CHECK(hipStreamSynchronize(s));
				if (all_sub) {
					// find_concealed, find_midi, find_smothered insert vertices (sub_kernels.hip); they read the dense PVST the
					// parallel stages wrote, so a component that needs the sequential redo has no place here
					if (nbad || (o.flags & (POVU_HIP_F_FORCE_REDO | POVU_HIP_F_REDO_ODD)))
						throw HipError("subflubble passes: a component went (or was sent) through the sequential redo of add_flubbles, "
							       "whose PVST layout the inserting passes do not read");
					tm.begin("subflubbles_insert");
					f->subx = std::make_shared<SubForest>();
					run_subflubbles(cs, sw, ctx->pw, ctx->tw, leaf_state, C, ctx->host, *f->subx, ctx->pool, s, &ctx->ws_sub, &ctx->ws_sub_hint);
					tm.end(40);
				}
				sum = nullptr; // (read again below: the pass total then includes this stage)
			}
			if (hairpins && !nbad) {
				run_parallel_hairpins(cs, sw, ctx->pw, C, tm, s);
				sum = nullptr;
			}
			if ((o.flags & POVU_HIP_F_FORCE_REDO) || (nbad && hairpins)) {
				// (tests; and the boundary report, which the parallel path only has for a whole pass)
				fill_u32(ctx->pw.comp_bad, C, 1u, s);
				nbad = C;
			} else if (nbad) {
				// only the flagged components go through the sequential kernels; the others keep the dense
				// result the parallel stages wrote
				mixed = true;
			}
			ctx->last_seq_redo = nbad;
			ctx->last_mixed = mixed;
			if (nbad) { // components whose candidate stack is not laminar: exact sequential redo
				tm.begin("redo_seq");
				if (dense_nb0 >= 0 && !hairpins) {
					// Only add_flubbles' stack machine (flubbles.cpp:316-365) cannot be evaluated in closed form on
					// such a stack; tree, classes, candidate stack and next_seen of the parallel stages stand.  They
					// are copied into the per-component layout, one lane per flagged component runs the machine.
					need_seq_workspace();
					export_parallel_stack(cs, sw, ctx->pw, s);
					ctx->stack_export_pending = false;
					HIP_CHECK(hipMemsetAsync(sw.in_s, 0, B + T, s));
					sw.stages = SEQ_STAGE_PVST | SEQ_STAGE_GIVEN_STACK;
					ctx->redo_pvst_only = true;
				} else {
					if (leaf_sub)
						throw HipError("leaf subflubble passes: this pass rebuilds the flagged components from scratch with the "
							       "one-lane kernels, whose tree state the passes do not read");
					if (dense_nb0 >= 0) // parallel tree: the one-lane kernels start from scratch
						init_seq_workspace();
					sw.stages = dense_nb0 >= 0 ? SEQ_STAGE_ALL : (SEQ_STAGE_CLASSES | SEQ_STAGE_STACK | SEQ_STAGE_PVST);
				}
				sw.comp_sel = ctx->pw.comp_bad;
				if (leaf_sub) { // (only reached with the PVST-only redo, see above)
					sw.p_ai = leaf_state.p_ai;
					sw.p_zi = leaf_state.p_zi;
				}
				launch_seq_components(sw, s);
				sw.comp_sel = nullptr;
				if (leaf_sub)
					leaf_seq(leaf_state, cs, sw, ctx->pw.comp_bad, C, s);
				tm.end(1);
				sum = nullptr;
			}
		}

		// ---- PVST arrays back to the host
		tm.begin("pvst_d2h");
		if (!sum)
			sum = read_summary(!all_seq);
		const uint32_t *cstat = sum + 4 + (size_t)C, *npvst = sum + 4 + 2 * (size_t)C, *nbry = sum + 4 + 3 * (size_t)C,
			       *doff = sum + 4 + 4 * (size_t)C;
		for (uint32_t c = 0; c < C; c++)
			if (cstat[c] == 2)
				throw HipError("internal error: the spanning tree of component " + std::to_string(c + 1) +
					       " did not reach every side");
		f->total_components = C;
		size_t total = 0, total_hp = 0;
		for (uint32_t c = 0; c < C; c++) {
			if (npvst[c] == 0)
				continue;
			povu_hip_forest::Tree t;
			t.component_id = c + 1; // decompose.cpp:129
			t.n_vtx = voff[c + 1] - voff[c];
			t.n_links = eoff[c + 1] - eoff[c];
			t.n_pvst = npvst[c];
			t.off = total;
			t.hp_off = total_hp;
			t.n_hairpins = hairpins ? nbry[c] : 0;
			t.sub_c = c;
			total += npvst[c];
			total_hp += t.n_hairpins;
			f->trees.push_back(t);
		}
		const bool dense_out = !all_seq && (ctx->last_seq_redo == 0 || mixed);
		const uint32_t *bad = sum + 4;
		// results of the sequential kernels (per-component layout sw.p_*) -> `dst` arrays at every tree's `off`
		auto fetch_seq = [&](std::vector<povu_hip_forest::Tree *> &ts, uint32_t *da, uint32_t *dz, uint32_t *dp, uint8_t *dao,
				     uint8_t *dzo, size_t n_total) {
			std::vector<uint8_t> ors(n_total);
			if (ts.size() <= 32) { // few trees: copy exactly their spans
				for (const auto *t : ts) {
					const size_t pb = (size_t)voff[t->component_id - 1] + (t->component_id - 1);
					HIP_CHECK(copy_async(da + t->off, sw.p_a + pb, (size_t)t->n_pvst * 4, hipMemcpyDeviceToHost, s));
					HIP_CHECK(copy_async(dz + t->off, sw.p_z + pb, (size_t)t->n_pvst * 4, hipMemcpyDeviceToHost, s));
					HIP_CHECK(copy_async(dp + t->off, sw.p_parent + pb, (size_t)t->n_pvst * 4, hipMemcpyDeviceToHost, s));
					HIP_CHECK(copy_async(ors.data() + t->off, sw.p_or + pb, t->n_pvst, hipMemcpyDeviceToHost, s));
					if (t->n_hairpins)
						HIP_CHECK(copy_async(f->hairpins.data() + 2 * t->hp_off, sw.hairpins + 2 * pb,
									 (size_t)t->n_hairpins * 16, hipMemcpyDeviceToHost, s));
				}
				HIP_CHECK(hipEventRecord(f->ev1, s));
				HIP_CHECK(hipStreamSynchronize(s));
			} else { // many trees: one bulk copy per array, sliced on the host
				const size_t P = (size_t)g.V + C;
				std::vector<uint32_t> ha(P), hz(P), hp(P);
				std::vector<uint8_t> ho(P);
				std::vector<uint64_t> hh(hairpins ? 2 * P : 0);
				HIP_CHECK(copy_async(ha.data(), sw.p_a, P * 4, hipMemcpyDeviceToHost, s));
				HIP_CHECK(copy_async(hz.data(), sw.p_z, P * 4, hipMemcpyDeviceToHost, s));
				HIP_CHECK(copy_async(hp.data(), sw.p_parent, P * 4, hipMemcpyDeviceToHost, s));
				HIP_CHECK(copy_async(ho.data(), sw.p_or, P, hipMemcpyDeviceToHost, s));
				if (hairpins)
					HIP_CHECK(copy_async(hh.data(), sw.hairpins, 2 * P * 8, hipMemcpyDeviceToHost, s));
				HIP_CHECK(hipEventRecord(f->ev1, s));
				HIP_CHECK(hipStreamSynchronize(s));
				for (const auto *t : ts) {
					const size_t pb = (size_t)voff[t->component_id - 1] + (t->component_id - 1);
					std::copy_n(ha.begin() + pb, t->n_pvst, da + t->off);
					std::copy_n(hz.begin() + pb, t->n_pvst, dz + t->off);
					std::copy_n(hp.begin() + pb, t->n_pvst, dp + t->off);
					std::copy_n(ho.begin() + pb, t->n_pvst, ors.begin() + t->off);
					if (t->n_hairpins)
						std::copy_n(hh.begin() + 2 * pb, 2 * (size_t)t->n_hairpins, f->hairpins.begin() + 2 * t->hp_off);
				}
			}
			for (size_t i = 0; i < n_total; i++) {
				dao[i] = ors[i] & 1;
				dzo[i] = (ors[i] >> 1) & 1;
			}
		};
		// ... and their subflubble labels (leaf_seq wrote them in the same per-component layout)
		auto fetch_seq_sub = [&](std::vector<povu_hip_forest::Tree *> &ts, PinnedVec<uint32_t> &dai, PinnedVec<uint32_t> &dzi,
					 PinnedVec<uint8_t> &dfam, size_t n_total) {
			dai.assign(n_total, POVU_NIL);
			dzi.assign(n_total, POVU_NIL);
			dfam.assign(n_total, 0);
			const size_t P = (size_t)g.V + C;
			std::vector<uint32_t> ha(P), hz(P);
			std::vector<uint8_t> hf(P);
			HIP_CHECK(copy_async(ha.data(), leaf_state.p_ai, P * 4, hipMemcpyDeviceToHost, s));
			HIP_CHECK(copy_async(hz.data(), leaf_state.p_zi, P * 4, hipMemcpyDeviceToHost, s));
			HIP_CHECK(copy_async(hf.data(), leaf_state.p_fam, P, hipMemcpyDeviceToHost, s));
			HIP_CHECK(hipStreamSynchronize(s));
			for (const auto *t : ts) {
				const size_t pb = (size_t)voff[t->component_id - 1] + (t->component_id - 1);
				std::copy_n(ha.begin() + pb, t->n_pvst, dai.begin() + t->off);
				std::copy_n(hz.begin() + pb, t->n_pvst, dzi.begin() + t->off);
				std::copy_n(hf.begin() + pb, t->n_pvst, dfam.begin() + t->off);
			}
		};
		f->hairpins.resize(2 * total_hp);
		if (dense_out) { // the parallel stages wrote every PVST back to back into f->block
			if (!mixed && (doff[C] != total || total != ctx->pw.d_total))
				throw HipError("internal error: dense PVST size mismatch");
			if (!f->block)
				f->alloc(ctx->pw.d_total);
			std::vector<povu_hip_forest::Tree *> redo;
			size_t redo_total = 0;
			for (auto &t : f->trees) {
				if (mixed && bad[t.component_id - 1]) {
					t.off = redo_total;
					redo_total += t.n_pvst;
					redo.push_back(&t);
				} else {
					t.off = doff[t.component_id - 1];
				}
			}
			bool more = false;
			for (const auto &t : f->trees)
				if (t.n_hairpins) {
					const size_t pb = (size_t)voff[t.component_id - 1] + (t.component_id - 1);
					HIP_CHECK(copy_async(f->hairpins.data() + 2 * t.hp_off, sw.hairpins + 2 * pb,
								 (size_t)t.n_hairpins * 16, hipMemcpyDeviceToHost, s));
					more = true;
				}
			tm.end(0);
			if (!redo.empty()) { // the redone components get a block of their own
				povu_hip_forest::ExtraBlock blk;
				blk.pool = ctx->pool;
				blk.p = ctx->pool->get(povu_hip_forest::ExtraBlock::bytes_for(redo_total), blk.cap);
				blk.carve(redo_total);
				f->extra.push_back(std::move(blk)); // (the raw pointers of blk stay valid: only the owning members move)
				for (auto *t : redo)
					t->blk = 0;
				fetch_seq(redo, blk.a, blk.z, blk.parent, blk.aor, blk.zor, redo_total);
				if (leaf_sub)
					fetch_seq_sub(redo, f->extra[0].sub_ai, f->extra[0].sub_zi, f->extra[0].sub_fam, redo_total);
			} else if (fast_tail && tail.overlapped && !more) {
				// the arrays are still on their way: the caller (or the next accessor of the forest) waits for ev1
				f->pending = true;
				ctx->tail_pending = true;
			} else if (fast_tail && !more) {
				HIP_CHECK(hipEventSynchronize(f->ev1)); // (recorded behind the last copy by run_parallel_dg)
				if (tm.enabled)
					HIP_CHECK(hipStreamSynchronize(s)); // (the stage events themselves have to complete before they are read)
			} else if (more || tm.enabled || fast_tail) {
				HIP_CHECK(hipEventRecord(f->ev1, s));
				HIP_CHECK(hipStreamSynchronize(s));
			}
		} else {
			if (f->block && f->total_entries != total)
				f->release_block();
			if (!f->block)
				f->alloc(total);
			std::vector<povu_hip_forest::Tree *> all;
			for (auto &t : f->trees)
				all.push_back(&t);
			tm.end(0);
			fetch_seq(all, f->a_id.data(), f->z_id.data(), f->parent.data(), f->a_or.data(), f->z_or.data(), total);
			if (leaf_sub)
				fetch_seq_sub(all, f->sub_ai, f->sub_zi, f->sub_fam, total);
		}
		if (!leaf_sub)
			ctx->ws_leaf.release(); // (kept from one -s pass to the next, like the inserting passes' arena below: a hipMalloc of
						// several GB right after the hipFree of the pass before stalled for over a second once; a
						// pass without the leaf passes gives it back)
		if (!all_sub)
			ctx->ws_sub.release(); // (the inserting passes keep their tables' arena from one -s pass to the next, no longer)
		// stage times
		ctx->last_times.clear();
		for (auto &r : tm.recs) {
			povu_hip_stage_time st{};
			snprintf(st.name, sizeof st.name, "%s", r.name.c_str());
			float ms = 0;
			HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
			st.ms = ms;
			st.launches = r.launches;
			ctx->last_times.push_back(st);
		}
		if (!f->pending) {
			povu_hip_stage_time st{};
			snprintf(st.name, sizeof st.name, "total");
			f->ready();
			st.ms = f->pass_ms;
			ctx->last_times.push_back(st);
		}
		ctx->C = C;
		ctx->have_state = true;
		return f.release();
	} catch (const std::exception &e) {
		if (ctx && ctx->stream)
			(void)hipStreamSynchronize(ctx->stream);
		if (ctx && ctx->side.stream)
			(void)hipStreamSynchronize(ctx->side.stream);
		f.reset();
		set_err(err, errlen, e.what());
		return nullptr;
	}
}

extern "C" uint32_t povu_hip_forest_total_components(const povu_hip_forest *f) { return f ? f->total_components : 0; }
extern "C" uint32_t povu_hip_forest_tree_count(const povu_hip_forest *f) { return f ? (uint32_t)f->trees.size() : 0; }

extern "C" int povu_hip_forest_wait(povu_hip_forest *f)
{
	if (!f)
		return 1;
	f->ready();
	return 0;
}
extern "C" double povu_hip_forest_pass_ms(povu_hip_forest *f)
{
	if (!f)
		return -1.0;
	f->ready();
	return f->pass_ms;
}

extern "C" double povu_hip_forest_span_ms(povu_hip_forest *first, povu_hip_forest *last)
{
	if (!first || !last || !first->ev0 || !last->ev1)
		return -1.0;
	first->ready();
	last->ready();
	float ms = 0;
	return hipEventElapsedTime(&ms, first->ev0, last->ev1) == hipSuccess ? (double)ms : -1.0;
}

extern "C" int povu_hip_forest_get(const povu_hip_forest *f, uint32_t i, povu_hip_tree *out)
{
	if (!f || !out || i >= f->trees.size())
		return 1;
	const_cast<povu_hip_forest *>(f)->ready(); // (the arrays of a POVU_HIP_F_ASYNC forest may still be on their way)
	const auto &t = f->trees[i];
	out->component_id = t.component_id;
	out->n_vtx = t.n_vtx;
	out->n_links = t.n_links;
	out->n_pvst = t.n_pvst;
	if (t.blk < 0) {
		out->a_id = f->a_id.data() + t.off;
		out->z_id = f->z_id.data() + t.off;
		out->a_or = f->a_or.data() + t.off;
		out->z_or = f->z_or.data() + t.off;
		out->parent = f->parent.data() + t.off;
	} else {
		const auto &b = f->extra[(size_t)t.blk];
		out->a_id = b.a + t.off;
		out->z_id = b.z + t.off;
		out->a_or = b.aor + t.off;
		out->z_or = b.zor + t.off;
		out->parent = b.parent + t.off;
	}
	out->n_hairpins = t.n_hairpins;
	out->hairpins = t.n_hairpins ? f->hairpins.data() + 2 * t.hp_off : nullptr;
	return 0;
}

extern "C" int povu_hip_forest_raw(const povu_hip_forest *f, const void **block, size_t *bytes, uint64_t *total,
				   uint64_t offsets[5])
{
	if (!f || !block || !bytes || !total || !offsets || !f->extra.empty())
		return 1; // (a merged forest has one block per rank: no single raw view)
	const_cast<povu_hip_forest *>(f)->ready();
	*block = f->block;
	*bytes = f->block_bytes;
	*total = f->total_entries;
	const char *b = static_cast<const char *>(f->block);
	offsets[0] = (uint64_t)((const char *)f->a_id.p - b);
	offsets[1] = (uint64_t)((const char *)f->z_id.p - b);
	offsets[2] = (uint64_t)((const char *)f->parent.p - b);
	offsets[3] = (uint64_t)((const char *)f->a_or.p - b);
	offsets[4] = (uint64_t)((const char *)f->z_or.p - b);
	return 0;
}

extern "C" uint64_t povu_hip_forest_first(const povu_hip_forest *f, uint32_t i)
{
	return (f && i < f->trees.size()) ? (uint64_t)f->trees[i].off : 0;
}

extern "C" void povu_hip_forest_free(povu_hip_forest *f) { delete f; }
extern "C" void povu_hip_buffer_free(void *p) { free(p); }

// mto::to_pvst::write_pvst, src/mto/to_pvst.cpp:23-109
extern "C" int povu_hip_forest_get_sub(const povu_hip_forest *f, uint32_t i, const uint32_t **ai, const uint32_t **zi,
				       const uint8_t **fam)
{
	if (!f || i >= f->trees.size())
		return 1;
	const auto &t = f->trees[i];
	const PinnedVec<uint32_t> &va = t.blk < 0 ? f->sub_ai : f->extra[(size_t)t.blk].sub_ai;
	const PinnedVec<uint32_t> &vz = t.blk < 0 ? f->sub_zi : f->extra[(size_t)t.blk].sub_zi;
	const PinnedVec<uint8_t> &vf = t.blk < 0 ? f->sub_fam : f->extra[(size_t)t.blk].sub_fam;
	if (f->sub_fam.empty() || t.off + t.n_pvst > vf.size())
		return 3; // the forest was not decomposed with POVU_HIP_F_LEAF_SUBFLUBBLES
	if (ai)
		*ai = va.data() + t.off;
	if (zi)
		*zi = vz.data() + t.off;
	if (fam)
		*fam = vf.data() + t.off;
	return 0;
}

extern "C" int povu_hip_forest_get_subtree(const povu_hip_forest *f, uint32_t i, povu_hip_subtree *out)
{
	if (!f || !out || i >= f->trees.size())
		return 1;
	const_cast<povu_hip_forest *>(f)->ready();
	const auto &t = f->trees[i];
	const SubForest *x = t.blk < 0 ? f->subx.get() : f->extra[(size_t)t.blk].subx.get();
	if (!x || t.sub_c + 1 >= x->voff.size())
		return 3; // the forest was not decomposed with POVU_HIP_F_SUBFLUBBLES
	const uint64_t b = x->voff[t.sub_c], e = x->voff[t.sub_c + 1];
	out->n_total = (uint32_t)(e - b);
	out->n_flubble_like = t.n_pvst;
	out->n_concealed = x->counts[3 * (size_t)t.sub_c];
	out->n_midi = x->counts[3 * (size_t)t.sub_c + 1];
	out->n_smothered = x->counts[3 * (size_t)t.sub_c + 2];
	out->fam = x->fam + b;
	out->or1 = x->or1 + b;
	out->or2 = x->or2 + b;
	out->route = x->route + b;
	out->id1 = x->id1 + b;
	out->id2 = x->id2 + b;
	out->child_off = x->coff + b;
	out->child = x->child;
	return 0;
}

// write_pvst, src/mto/to_pvst.cpp:31-109, of a tree with its -s vertices
extern "C" char *povu_hip_pvst_format_subtree(const povu_hip_subtree *t, size_t *len)
{
	if (!t)
		return nullptr;
	std::string o;
	o.reserve(32 * (size_t)t->n_total + 64);
	o += "H\t0.0.3\t.\t.\t.\n";
	for (uint32_t v = 0; v < t->n_total; v++) {
		o += (char)t->fam[v];
		o += '\t';
		o += std::to_string(v);
		o += '\t';
		if (t->fam[v] == 'D') {
			o += '.';
		} else { // id_or_t::as_str, include/povu/graph/types.hpp:85-95
			o += t->or1[v] ? '<' : '>';
			o += std::to_string(t->id1[v]);
			o += t->or2[v] ? '<' : '>';
			o += std::to_string(t->id2[v]);
		}
		o += '\t';
		const uint32_t c0 = t->child_off[v], c1 = t->child_off[v + 1];
		if (c0 == c1) {
			o += '.';
		} else { // print_with_comma, include/povu/common/utils.hpp:44-55
			for (uint32_t k = c0; k < c1; k++) {
				o += std::to_string(t->child[k]);
				if (k + 1 < c1)
					o += ", ";
			}
		}
		o += '\t';
		if (t->route[v])
			o += (char)t->route[v];
		else
			o += '.';
		o += '\n';
	}
	char *buf = static_cast<char *>(malloc(o.size() + 1));
	if (!buf)
		return nullptr;
	memcpy(buf, o.data(), o.size() + 1);
	if (len)
		*len = o.size();
	return buf;
}

extern "C" char *povu_hip_forest_pvst_text(const povu_hip_forest *f, uint32_t i, size_t *len)
{
	povu_hip_subtree st;
	if (povu_hip_forest_get_subtree(f, i, &st) == 0)
		return povu_hip_pvst_format_subtree(&st, len);
	povu_hip_tree t;
	if (povu_hip_forest_get(f, i, &t) != 0)
		return nullptr;
	const uint8_t *fam = nullptr;
	if (povu_hip_forest_get_sub(f, i, nullptr, nullptr, &fam) != 0)
		fam = nullptr;
	return povu_hip_pvst_format_fam(t.n_pvst, t.a_id, t.z_id, t.a_or, t.z_or, t.parent, fam, len);
}

extern "C" char *povu_hip_pvst_format(uint32_t n_pvst, const uint32_t *a_id, const uint32_t *z_id, const uint8_t *a_or,
				      const uint8_t *z_or, const uint32_t *parent, size_t *len)
{
	return povu_hip_pvst_format_fam(n_pvst, a_id, z_id, a_or, z_or, parent, nullptr, len);
}

extern "C" char *povu_hip_pvst_format_fam(uint32_t n_pvst, const uint32_t *a_id, const uint32_t *z_id, const uint8_t *a_or,
					  const uint8_t *z_or, const uint32_t *parent, const uint8_t *fam, size_t *len)
{
	if (n_pvst == 0 || !a_id || !z_id || !a_or || !z_or || !parent)
		return nullptr;
	if (fam)
		for (uint32_t v = 0; v < n_pvst; v++)
			if (fam[v] != (v == 0 ? 'D' : 'F') && (v == 0 || (fam[v] != 'T' && fam[v] != 'O')))
				return nullptr; // line letters this writer knows: D for the root, F / T / O below it
	struct {
		uint32_t n_pvst;
		const uint32_t *a_id, *z_id, *parent;
		const uint8_t *a_or, *z_or;
	} t{n_pvst, a_id, z_id, parent, a_or, z_or};
	for (uint32_t v = 1; v < n_pvst; v++)
		if (parent[v] >= v)
			return nullptr; // a PVST parent always precedes its children (emission order)
	const uint32_t n = t.n_pvst;
	// children of every PVST vertex in emission order (children_v push_back, pvst.hpp:882-892)
	std::unique_ptr<uint32_t[]> coff(new uint32_t[(size_t)n + 2]), cadj(new uint32_t[n]); // (uninitialised)
	std::fill(coff.get(), coff.get() + n + 2, 0u);
	for (uint32_t v = 1; v < n; v++)
		coff[t.parent[v] + 2]++; // (shifted by one: after the sums coff[p + 1] is where p's children start, and the fill
	for (uint32_t v = 0; v < n; v++) //  below advances it to where they end = where those of p + 1 start)
		coff[v + 2] += coff[v + 1];
	for (uint32_t v = 1; v < n; v++)
		cadj[coff[t.parent[v] + 1]++] = v;
	// One buffer of the largest size the text can have, written front to back (a 10^6-vertex PVST is 5 * 10^6 numbers: a
	// library call per number and a growing string were most of the CLI's write time).  Per line at most: letter + tab 2,
	// vertex 10, tab 1, the two oriented endpoints 22, tab 1, '.' 1, closing field 3; per child entry 10 + ", ".
	const size_t cap = 16 + (size_t)n * 40 + (size_t)n * 12 + 1;
	char *buf = (char *)malloc(cap);
	if (!buf)
		return nullptr;
	static const char D2[] = "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
				 "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
	char *o = buf;
	auto put = [&](uint32_t v) {
		const int len = v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6 :
				v < 10000000u ? 7 : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
		char *e = o + len;
		o = e;
		while (v >= 100u) {
			const uint32_t r = v % 100u;
			v /= 100u;
			e -= 2;
			e[0] = D2[2 * r], e[1] = D2[2 * r + 1];
		}
		if (v >= 10u)
			e[-2] = D2[2 * v], e[-1] = D2[2 * v + 1];
		else
			e[-1] = (char)('0' + v);
	};
	memcpy(o, "H\t0.0.3\t.\t.\t.\n", 14); // to_pvst.cpp:23-28
	o += 14;
	for (uint32_t v = 0; v < n; v++) {
		*o++ = fam ? (char)fam[v] : (v == 0 ? 'D' : 'F'); // to_pvst.cpp:52-79: the line identifier follows the vertex family
		*o++ = '\t';
		put(v);
		*o++ = '\t';
		if (v == 0) {
			*o++ = '.';
		} else { // id_or_t::as_str, include/povu/graph/types.hpp:85-95
			*o++ = t.a_or[v] ? '<' : '>';
			put(t.a_id[v]);
			*o++ = t.z_or[v] ? '<' : '>';
			put(t.z_id[v]);
		}
		*o++ = '\t';
		const uint32_t c0 = coff[v], c1 = coff[v + 1];
		if (c0 == c1) {
			*o++ = '.';
		} else { // print_with_comma, include/povu/common/utils.hpp:44-55
			for (uint32_t k = c0; k < c1; k++) {
				put(cadj[k]);
				if (k + 1 < c1)
					*o++ = ',', *o++ = ' ';
			}
		}
		*o++ = '\t';
		*o++ = v == 0 ? '.' : 'L';
		*o++ = '\n';
	}
	*o = 0;
	if (len)
		*len = (size_t)(o - buf);
	return buf;
}

extern "C" int povu_hip_last_stage_times(const povu_hip_ctx *ctx, povu_hip_stage_time *out, int max)
{
	if (!ctx)
		return 0;
	int n = (int)ctx->last_times.size();
	if (out)
		for (int i = 0; i < n && i < max; i++)
			out[i] = ctx->last_times[i];
	return n;
}

extern "C" uint32_t povu_hip_last_seq_redo(const povu_hip_ctx *ctx) { return ctx ? ctx->last_seq_redo : 0; }

extern "C" uint64_t povu_hip_last_links_processed(const povu_hip_ctx *ctx) { return ctx ? ctx->last_links : 0; }

// ---- unit-test hook for the scans
extern "C" int povu_hip_debug_scan(povu_hip_ctx *ctx, int op, const uint32_t *in, uint32_t *out, size_t n, const uint32_t *in2,
				   uint32_t *out2, size_t n2)
{
	if (!ctx || !in || !out || (in2 && !out2))
		return 1;
	try {
		HIP_CHECK(hipSetDevice(ctx->device));
		hipStream_t s = ctx->stream;
		Arena ar;
		const size_t tb = scan_tmp_bytes(std::max(n, n2));
		ar.reserve(2 * Arena::padded(n + 16, 4) + 2 * Arena::padded(n2 + 16, 4) + tb + 4096);
		uint32_t *di = ar.take<uint32_t>(n + 16), *dout = ar.take<uint32_t>(n + 16);
		uint32_t *di2 = ar.take<uint32_t>(n2 + 16), *dout2 = ar.take<uint32_t>(n2 + 16);
		void *tmp = ar.take<char>(tb);
		HIP_CHECK(copy_async(di, in, n * 4, hipMemcpyHostToDevice, s));
		if (in2)
			HIP_CHECK(copy_async(di2, in2, n2 * 4, hipMemcpyHostToDevice, s));
		if (in2 && op == 0)
			scan_exclusive_u32_pair(di, dout, n, di2, dout2, n2, tmp, tb, s);
		else if (op == 0)
			scan_exclusive_u32(di, dout, n, tmp, tb, s);
		else
			scan_exclusive_max_u32(di, dout, n, tmp, tb, s);
		HIP_CHECK(copy_async(out, dout, n * 4, hipMemcpyDeviceToHost, s));
		if (in2 && op == 0)
			HIP_CHECK(copy_async(out2, dout2, n2 * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		return 0;
	} catch (const std::exception &) {
		return 2;
	}
}

// ---- timing hook for the scans: `reps` exclusive sum scans of n words (device resident, all ones), ms per scan by HIP events
extern "C" double povu_hip_debug_scan_time(povu_hip_ctx *ctx, size_t n, int reps, int op)
{
	if (!ctx || !n || reps <= 0)
		return -1.0;
	try {
		HIP_CHECK(hipSetDevice(ctx->device));
		hipStream_t s = ctx->stream;
		Arena ar;
		const size_t tb = scan_tmp_bytes(n);
		ar.reserve(2 * Arena::padded(n + 16, 4) + tb + 4096);
		uint32_t *di = ar.take<uint32_t>(n + 16), *dout = ar.take<uint32_t>(n + 16);
		void *tmp = ar.take<char>(tb);
		HIP_CHECK(hipMemsetAsync(di, 1, n * 4, s));
		hipEvent_t e0, e1;
		HIP_CHECK(hipEventCreate(&e0));
		HIP_CHECK(hipEventCreate(&e1));
		for (int w = 0; w < 2; w++)
			op ? scan_exclusive_max_u32(di, dout, n, tmp, tb, s) : scan_exclusive_u32(di, dout, n, tmp, tb, s);
		HIP_CHECK(hipEventRecord(e0, s));
		for (int r = 0; r < reps; r++)
			op ? scan_exclusive_max_u32(di, dout, n, tmp, tb, s) : scan_exclusive_u32(di, dout, n, tmp, tb, s);
		HIP_CHECK(hipEventRecord(e1, s));
		HIP_CHECK(hipEventSynchronize(e1));
		float ms = 0;
		HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
		(void)hipEventDestroy(e0);
		(void)hipEventDestroy(e1);
		return (double)ms / reps;
	} catch (const std::exception &) {
		return -2.0;
	}
}

// ---- stage-level parity hooks
extern "C" int povu_hip_debug_components(povu_hip_ctx *ctx, uint32_t *comp_of, uint32_t *local_idx)
{
	if (!ctx || !ctx->have_state)
		return 1;
	ctx->quiesce();
	try {
		HIP_CHECK(hipSetDevice(ctx->device));
		const uint32_t V = ctx->g.V, C = ctx->C;
		std::vector<uint32_t> pos(V), voff(C + 1);
		HIP_CHECK(hipMemcpy(comp_of, ctx->cs.comp_of, (size_t)V * 4, hipMemcpyDeviceToHost));
		if (ctx->cs.lean_identity)
			std::iota(pos.begin(), pos.end(), 0u);
		else
			HIP_CHECK(hipMemcpy(pos.data(), ctx->cs.pos, (size_t)V * 4, hipMemcpyDeviceToHost));
		HIP_CHECK(hipMemcpy(voff.data(), ctx->cs.voff, (size_t)(C + 1) * 4, hipMemcpyDeviceToHost));
		for (uint32_t v = 0; v < V; v++)
			local_idx[v] = pos[v] - voff[comp_of[v]];
		return 0;
	} catch (const std::exception &) {
		return 2;
	}
}

extern "C" int povu_hip_debug_tree(povu_hip_ctx *ctx, uint32_t comp, uint32_t *n_tree, uint32_t *gid, uint8_t *typ,
				   uint32_t *par, uint32_t *cls)
{
	if (!ctx || !ctx->have_state || comp >= ctx->C || !n_tree)
		return 1;
	if (cls && ctx->last_mixed)
		return 4; // classes of a mixed pass sit in two layouts (parallel stage / one-lane kernels): not exported
	ctx->quiesce();
	try {
		HIP_CHECK(hipSetDevice(ctx->device));
		uint32_t voff = 0, N = 0;
		HIP_CHECK(hipMemcpy(&voff, ctx->cs.voff + comp, 4, hipMemcpyDeviceToHost));
		HIP_CHECK(hipMemcpy(&N, ctx->sw.c_ntree + comp, 4, hipMemcpyDeviceToHost));
		*n_tree = N;
		const size_t tb = 2 * (size_t)voff + comp;
		if (gid)
			HIP_CHECK(hipMemcpy(gid, ctx->sw.t_gid + tb, (size_t)N * 4, hipMemcpyDeviceToHost));
		if (par)
			HIP_CHECK(hipMemcpy(par, ctx->sw.t_par + tb, (size_t)N * 4, hipMemcpyDeviceToHost));
		const bool par_cls = ctx->classes_in_par && (ctx->last_seq_redo == 0 || ctx->redo_pvst_only);
		if (cls && par_cls) {
			classes_to_tree_space(ctx->pw, ctx->stream);
			HIP_CHECK(hipStreamSynchronize(ctx->stream));
		}
		if (cls) // the parallel class stage keeps the classes in its own T-space array
			HIP_CHECK(hipMemcpy(cls, (par_cls ? ctx->pw.gcls : ctx->sw.t_cls) + tb, (size_t)N * 4, hipMemcpyDeviceToHost));
		if (typ)
			HIP_CHECK(hipMemcpy(typ, ctx->sw.t_flags + tb, N, hipMemcpyDeviceToHost));
		return 0;
	} catch (const std::exception &) {
		return 2;
	}
}

extern "C" int povu_hip_last_black_only_classes(const povu_hip_ctx *ctx)
{
	return ctx && ctx->have_state && ctx->classes_in_par && ctx->pw.black_only_used ? 1 : 0;
}

extern "C" int povu_hip_last_crossings(povu_hip_ctx *ctx, uint32_t out[2])
{
	if (!ctx || !out)
		return 1;
	out[0] = out[1] = 0;
	if (!ctx->have_state || !ctx->classes_in_par || !ctx->pw.laminar_checked)
		return 0;
	ctx->quiesce();
	if (hipSetDevice(ctx->device) != hipSuccess)
		return 2;
	return hipMemcpy(out, ctx->pw.err + 11, 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}

extern "C" int povu_hip_last_laminar_check_ran(const povu_hip_ctx *ctx)
{
	return ctx && ctx->have_state && ctx->classes_in_par && ctx->pw.laminar_checked ? 1 : 0;
}

extern "C" int povu_hip_debug_edge_ids(povu_hip_ctx *ctx, uint32_t comp, uint32_t *n_tree, uint32_t *tree_edge_id)
{
	if (!ctx || !ctx->have_state || comp >= ctx->C || !n_tree)
		return 1;
	if (!ctx->tree_in_par)
		return 3; // the one-lane tree kernels keep no per-side scan state
	if (ctx->last_mixed)
		return 4; // (see povu_hip_debug_tree)
	uint32_t *dw = nullptr;
	ctx->quiesce();
	try {
		HIP_CHECK(hipSetDevice(ctx->device));
		uint32_t voff = 0, N = 0;
		HIP_CHECK(hipMemcpy(&voff, ctx->cs.voff + comp, 4, hipMemcpyDeviceToHost));
		HIP_CHECK(hipMemcpy(&N, ctx->sw.c_ntree + comp, 4, hipMemcpyDeviceToHost));
		*n_tree = N;
		if (!tree_edge_id || N == 0)
			return 0;
		const size_t T = 2 * (size_t)ctx->sw.V + ctx->C, tb = 2 * (size_t)voff + comp;
		HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dw), 2 * T * 4));
		HIP_CHECK(hipMemsetAsync(dw, 0, 2 * T * 4, ctx->stream));
		debug_edge_id_weights(ctx->cs, ctx->sw, ctx->tw, dw, dw + T, ctx->stream);
		std::vector<uint32_t> w(N), tail(N), size(N);
		HIP_CHECK(copy_async(w.data(), dw + tb, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
		HIP_CHECK(copy_async(tail.data(), dw + T + tb, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
		HIP_CHECK(copy_async(size.data(), ctx->sw.t_size + tb, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
		HIP_CHECK(hipStreamSynchronize(ctx->stream));
		HIP_CHECK(hipFree(dw));
		dw = nullptr;
		// back edges created before vertex t is discovered: those in front of every vertex up to t, and the tails of
		// the vertices whose subtree closed before t
		std::vector<uint32_t> closed((size_t)N + 1, 0);
		for (uint32_t t = 0; t < N; t++)
			closed[std::min<size_t>((size_t)t + size[t], N)] += tail[t];
		uint32_t before = 0;
		tree_edge_id[0] = POVU_NIL;
		for (uint32_t t = 0; t < N; t++) {
			before += w[t] + closed[t];
			if (t > 0)
				tree_edge_id[t] = t - 1 + before;
		}
		return 0;
	} catch (const std::exception &) {
		if (dw)
			(void)hipFree(dw);
		return 2;
	}
}

extern "C" int povu_hip_debug_stack(povu_hip_ctx *ctx, uint32_t comp, uint32_t *n, uint32_t *tree_vtx, uint32_t *cls,
				    uint32_t *next_seen)
{
	if (!ctx || !ctx->have_state || comp >= ctx->C || !n)
		return 1;
	if (ctx->last_mixed)
		return 4; // the candidate stacks of a mixed pass sit in two layouts: not exported (see povu_hip_debug_tree)
	ctx->quiesce();
	try {
		HIP_CHECK(hipSetDevice(ctx->device));
		if (ctx->stack_export_pending && ctx->last_seq_redo == 0) {
			export_parallel_stack(ctx->cs, ctx->sw, ctx->pw, ctx->stream);
			HIP_CHECK(hipStreamSynchronize(ctx->stream));
			ctx->stack_export_pending = false;
		}
		uint32_t voff = 0, ns = 0;
		HIP_CHECK(hipMemcpy(&voff, ctx->cs.voff + comp, 4, hipMemcpyDeviceToHost));
		HIP_CHECK(hipMemcpy(&ns, ctx->sw.c_nstack + comp, 4, hipMemcpyDeviceToHost));
		*n = ns;
		if (tree_vtx)
			HIP_CHECK(hipMemcpy(tree_vtx, ctx->sw.s_vtx + voff, (size_t)ns * 4, hipMemcpyDeviceToHost));
		if (cls)
			HIP_CHECK(hipMemcpy(cls, ctx->sw.s_cls + voff, (size_t)ns * 4, hipMemcpyDeviceToHost));
		if (next_seen)
			HIP_CHECK(hipMemcpy(next_seen, ctx->sw.next_seen + voff, (size_t)ns * 4, hipMemcpyDeviceToHost));
		return 0;
	} catch (const std::exception &) {
		return 2;
	}
}
