// shard.hip -- multi-GPU component sharding (replaces the per-thread chunks of
// do_decompose, app/subcommand/decompose.cpp:78-92,116-157): device-side partition of the resident graph into
// per-rank sub-graphs, the packed shard format and its loader, forest packing / merging, and the RCCL
// communicator (ncclSend / ncclRecv over xGMI) behind povu_hip_comm_*.
#include "context.hpp"

#include "rccl_api.hpp"

#include <algorithm>
#include <chrono>
#include <dlfcn.h>
#include <fcntl.h>
#include <numeric>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace
{
constexpr int TPB = 256;
inline unsigned nblk(size_t n) { return (unsigned)((n + TPB - 1) / TPB); }
constexpr uint64_t SHARD_MAGIC = 0x31647268735F7670ull; // "pv_shrd1"

inline size_t pad256(size_t b) { return (b + 255) & ~size_t(255); }

// packed shard: [header 8 x u64 | vid | v1 | v2 | s1 | s2 | tip | comp ids], every section padded to 256 B
struct ShardLayout {
	size_t vid, v1, v2, s1, s2, tip, ids, bytes;
	ShardLayout(size_t nv, size_t ne, size_t nc)
	{
		size_t o = 256;
		vid = o, o += pad256(nv * 4);
		v1 = o, o += pad256(ne * 4);
		v2 = o, o += pad256(ne * 4);
		s1 = o, o += pad256(ne);
		s2 = o, o += pad256(ne);
		tip = o, o += pad256(nv);
		ids = o, o += pad256(nc * 4);
		bytes = o;
	}
};

double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------- kernels
__global__ void k_shard_comp_of(uint32_t V, const uint32_t *__restrict__ label, const uint32_t *__restrict__ crank,
				uint32_t *__restrict__ comp_of)
{
	uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v < V)
		comp_of[v] = crank[label[v]];
}
// items per component (vertices: idx == nullptr; links: by their first endpoint).  The items of a component mostly
// sit next to one another, so the count is taken from the RUN BOUNDARIES: a run [b, e) of component c adds e - b, i.e.
// "+e" at its end and "-b" at its start (mod 2^32).  Two atomics per boundary -- a handful for a sorted graph, and
// spread over all components when it is not -- instead of one hot counter per chromosome-sized component.
__global__ void k_comp_count(uint32_t n, const uint32_t *__restrict__ comp_of, const uint32_t *__restrict__ idx,
			     uint32_t *__restrict__ cnt)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const uint32_t c = comp_of[idx ? idx[i] : i];
	if (i > 0) {
		const uint32_t p = comp_of[idx ? idx[i - 1] : i - 1];
		if (p != c) {
			atomicAdd(&cnt[p], i);
			atomicSub(&cnt[c], i);
		}
	}
	if (i == n - 1)
		atomicAdd(&cnt[c], n);
}
__global__ void k_owner_keys(uint32_t n, const uint32_t *__restrict__ comp_of, const uint32_t *__restrict__ idx,
			     const uint32_t *__restrict__ owner, uint32_t *__restrict__ key, uint32_t *__restrict__ val)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		key[i] = owner[comp_of[idx ? idx[i] : i]];
		val[i] = i;
	}
}
struct PartTable { // per rank: first sorted position and the device address of every output section
	uint32_t vbase, ebase;
	uint32_t *vid, *v1, *v2;
	uint8_t *s1, *s2, *tip;
};
__global__ void k_part_vertices(uint32_t V, const uint32_t *__restrict__ key, const uint32_t *__restrict__ perm,
				const PartTable *__restrict__ tab, const uint32_t *__restrict__ vid, const uint8_t *__restrict__ tip,
				uint32_t *__restrict__ newidx)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= V)
		return;
	const PartTable t = tab[key[i]];
	const uint32_t v = perm[i], li = i - t.vbase;
	newidx[v] = li;
	t.vid[li] = vid[v];
	t.tip[li] = tip[v];
}
__global__ void k_part_links(uint32_t E, const uint32_t *__restrict__ key, const uint32_t *__restrict__ eperm,
			     const PartTable *__restrict__ tab, const uint32_t *__restrict__ v1, const uint32_t *__restrict__ v2,
			     const uint8_t *__restrict__ s1, const uint8_t *__restrict__ s2, const uint32_t *__restrict__ newidx)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= E)
		return;
	const PartTable t = tab[key[i]];
	const uint32_t e = eperm[i], li = i - t.ebase;
	t.v1[li] = newidx[v1[e]];
	t.v2[li] = newidx[v2[e]];
	t.s1[li] = s1[e];
	t.s2[li] = s2[e];
}
// the bytes between the end of a section and the next 256-byte boundary travel with a shard: they are cleared (a few hundred
// bytes per section) instead of clearing the whole block before it is written (gigabytes)
struct PadSpan {
	char *p;
	uint32_t n;
};
__global__ void k_zero_pads(uint32_t n_spans, const PadSpan *__restrict__ spans)
{
	if (blockIdx.x >= n_spans)
		return;
	const PadSpan sp = spans[blockIdx.x];
	if (threadIdx.x < sp.n)
		sp.p[threadIdx.x] = 0;
}
} // namespace

// ---------------------------------------------------------------- LPT (host)
extern "C" int povu_hip_lpt_assign(const uint64_t *weights, uint32_t n, uint32_t world, uint32_t *owner_out)
{
	if ((!weights && n) || !owner_out || world == 0)
		return 1;
	std::vector<uint32_t> order(n);
	std::iota(order.begin(), order.end(), 0u);
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return weights[a] > weights[b]; });
	std::vector<uint64_t> load(world, 0);
	for (uint32_t k = 0; k < n; k++) {
		const uint32_t c = order[k];
		uint32_t best = 0;
		for (uint32_t r = 1; r < world; r++)
			if (load[r] < load[best])
				best = r;
		owner_out[c] = best;
		load[best] += weights[c] + 1;
	}
	return 0;
}

// ---------------------------------------------------------------- partition
struct povu_hip_shards {
	uint32_t world = 0, C = 0;
	int device = 0;
	void *block = nullptr; // in the context's partition arena: valid until the next partition on that context
	struct Part {
		uint32_t nv = 0, ne = 0, nc = 0;
		uint64_t weight = 0;
		size_t off = 0, bytes = 0;
	};
	std::vector<Part> parts;
	double ms[3] = {0, 0, 0};
};

extern "C" povu_hip_shards *povu_hip_shard_partition(povu_hip_ctx *ctx, uint32_t world, char *err, size_t errlen)
{
	std::unique_ptr<povu_hip_shards> sh;
	XferScope xfer(ctx);
	try {
		if (!ctx || !ctx->g.block)
			throw HipError("no graph resident: call povu_hip_graph_upload first");
		if (world == 0 || world > 4096)
			throw HipError("bad world size");
		HIP_CHECK(hipSetDevice(ctx->device));
		ctx->wait_tail();
		const ResidentGraph &g = ctx->g;
		hipStream_t s = ctx->stream;
		const uint32_t V = g.V, E = g.E;
		Sizes z;
		z.V = V;
		z.E = E;
		z.Cmax = V;
		z.nS = 2 * z.V;
		z.slots = g.n_slots;
		z.T = z.B = 0;
		CompState &cs = ctx->cs;
		ctx->have_state = false;
		ctx->host.reset();
		cs.host = &ctx->host;
		cs.host_pub = nullptr;
		ctx->ws.reserve(carve_workspace(nullptr, 0, z, cs, ctx->sw, false));
		carve_workspace(&ctx->ws, 0, z, cs, ctx->sw, false);
		StageTimer &tm = ctx->timer;
		tm.reset();
		tm.enabled = false;
		hipEvent_t ev[4];
		for (auto &e : ev)
			e = tm.get();
		HIP_CHECK(hipEventRecord(ev[0], s));
		// ---- components (row B's union-find kernels) and their sizes
		const uint32_t C = label_components(g, cs, tm, s);
		HIP_CHECK(hipEventRecord(ev[1], s));
		uint32_t *comp_of = cs.comp_of, *cntv = cs.voff, *cnte = cs.eoff; // [V+1], [C+2] each
		KLAUNCH(k_shard_comp_of, dim3(nblk(V)), dim3(TPB), 0, s, V, cs.label, cs.crank, comp_of);
		HIP_CHECK(hipMemsetAsync(cntv, 0, (size_t)C * 4, s));
		HIP_CHECK(hipMemsetAsync(cnte, 0, (size_t)C * 4, s));
		KLAUNCH(k_comp_count, dim3(nblk(V)), dim3(TPB), 0, s, V, comp_of, (const uint32_t *)nullptr, cntv);
		if (E)
			KLAUNCH(k_comp_count, dim3(nblk(E)), dim3(TPB), 0, s, E, comp_of, g.v1, cnte);
		uint32_t *hcnt = ctx->host.take<uint32_t>(2 * (size_t)C);
		HIP_CHECK(copy_async(hcnt, cntv, (size_t)C * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(copy_async(hcnt + C, cnte, (size_t)C * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		// ---- bin packing on the host (components are few next to their size)
		std::vector<uint64_t> w(C);
		for (uint32_t c = 0; c < C; c++)
			w[c] = (uint64_t)hcnt[c] + hcnt[C + c];
		uint32_t *howner = ctx->host.take<uint32_t>((size_t)C + 1);
		povu_hip_lpt_assign(w.data(), C, world, howner);
		sh = std::make_unique<povu_hip_shards>();
		sh->world = world;
		sh->C = C;
		sh->device = ctx->device;
		sh->parts.resize(world);
		std::vector<std::vector<uint32_t>> ids(world);
		for (uint32_t c = 0; c < C; c++) {
			auto &p = sh->parts[howner[c]];
			p.nv += hcnt[c];
			p.ne += hcnt[C + c];
			p.nc++;
			p.weight += w[c] + 1;
			ids[howner[c]].push_back(c + 1); // component ids are 1-based (decompose.cpp:129)
		}
		size_t total = pad256((size_t)world * sizeof(PartTable)); // the partition table sits in front of the shards
		for (auto &p : sh->parts) {
			p.off = total;
			p.bytes = ShardLayout(p.nv, p.ne, p.nc).bytes;
			total += p.bytes;
		}
		const size_t pads_off = total; // the table of padding spans sits behind the shards
		total += pad256((size_t)world * 8 * sizeof(PadSpan));
		// the block lives in an arena of the context (a step of a sharded job partitions again: no allocation, nothing to clear
		// but the section padding, which travels too and stays defined)
		ctx->part_arena.reserve(total + 512);
		sh->block = ctx->part_arena.take<char>(total);
		char *blk = static_cast<char *>(sh->block);
		PartTable *htab = ctx->host.take<PartTable>(world);
		uint64_t *hhdr = ctx->host.take<uint64_t>(32 * (size_t)world); // (a whole 256-byte header per rank)
		PadSpan *hpads = ctx->host.take<PadSpan>(8 * (size_t)world);
		uint32_t n_pads = 0;
		uint32_t vb = 0, eb = 0;
		for (uint32_t r = 0; r < world; r++) {
			const auto &p = sh->parts[r];
			const ShardLayout L(p.nv, p.ne, p.nc);
			char *b = blk + p.off;
			htab[r] = PartTable{vb,
					    eb,
					    (uint32_t *)(b + L.vid),
					    (uint32_t *)(b + L.v1),
					    (uint32_t *)(b + L.v2),
					    (uint8_t *)(b + L.s1),
					    (uint8_t *)(b + L.s2),
					    (uint8_t *)(b + L.tip)};
			vb += p.nv;
			eb += p.ne;
			uint64_t *h = hhdr + 32 * (size_t)r;
			std::fill(h, h + 32, 0ull);
			h[0] = SHARD_MAGIC, h[1] = p.nv, h[2] = p.ne, h[3] = p.nc, h[4] = C;
			HIP_CHECK(copy_async(b, h, 256, hipMemcpyHostToDevice, s));
			if (p.nc)
				HIP_CHECK(copy_async(b + L.ids, ids[r].data(), (size_t)p.nc * 4, hipMemcpyHostToDevice, s));
			const size_t ends[7][2] = {{L.vid, (size_t)p.nv * 4}, {L.v1, (size_t)p.ne * 4}, {L.v2, (size_t)p.ne * 4}, {L.s1, p.ne},
						   {L.s2, p.ne},	      {L.tip, p.nv},	      {L.ids, (size_t)p.nc * 4}};
			for (const auto &e : ends) {
				const size_t end = e[0] + e[1], n = pad256(end) - end;
				if (n)
					hpads[n_pads++] = PadSpan{b + end, (uint32_t)n};
			}
		}
		if (n_pads) {
			PadSpan *dpads = reinterpret_cast<PadSpan *>(blk + pads_off);
			HIP_CHECK(copy_async(dpads, hpads, (size_t)n_pads * sizeof(PadSpan), hipMemcpyHostToDevice, s));
			KLAUNCH(k_zero_pads, dim3(n_pads), dim3(256), 0, s, n_pads, dpads);
		}
		HIP_CHECK(hipEventRecord(ev[2], s));
		// ---- partition on the device: a stable 1-pass radix sort by owner keeps vertex and link order
		uint32_t *owner = cs.tmp_a; // [V+1] >= C
		HIP_CHECK(copy_async(owner, howner, (size_t)C * 4, hipMemcpyHostToDevice, s));
		PartTable *tab = reinterpret_cast<PartTable *>(sh->block);
		HIP_CHECK(copy_async(tab, htab, (size_t)world * sizeof(PartTable), hipMemcpyHostToDevice, s));
		const unsigned kbits = bits_for(world - 1);
		uint32_t *kv = cs.ckey, *iv = cs.perm, *kv2 = cs.vdeg, *perm = cs.sbase, *newidx = cs.pos; // [V+1] each
		KLAUNCH(k_owner_keys, dim3(nblk(V)), dim3(TPB), 0, s, V, comp_of, (const uint32_t *)nullptr, owner, kv, iv);
		sort_pairs_u32(kv, kv2, iv, perm, V, kbits, cs.sort_tmp, cs.sort_tmp_bytes, s);
		KLAUNCH(k_part_vertices, dim3(nblk(V)), dim3(TPB), 0, s, V, kv2, perm, tab, g.vid, g.tip, newidx);
		if (E) {
			KLAUNCH(k_owner_keys, dim3(nblk(E)), dim3(TPB), 0, s, E, comp_of, g.v1, owner, cs.keys, cs.vals);
			sort_pairs_u32(cs.keys, cs.keys2, cs.vals, cs.vals2, E, kbits, cs.sort_tmp, cs.sort_tmp_bytes, s);
			KLAUNCH(k_part_links, dim3(nblk(E)), dim3(TPB), 0, s, E, cs.keys2, cs.vals2, tab, g.v1, g.v2, g.s1, g.s2, newidx);
		}
		HIP_CHECK(hipEventRecord(ev[3], s));
		HIP_CHECK(hipStreamSynchronize(s));
		for (int i = 0; i < 3; i++) {
			float ms = 0;
			HIP_CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
			sh->ms[i] = ms;
		}
		return sh.release();
	} catch (const std::exception &e) {
		if (ctx && ctx->stream)
			(void)hipStreamSynchronize(ctx->stream);
		set_err(err, errlen, e.what());
		return nullptr;
	}
}

extern "C" uint32_t povu_hip_shards_world(const povu_hip_shards *s) { return s ? s->world : 0; }
extern "C" uint32_t povu_hip_shards_total_components(const povu_hip_shards *s) { return s ? s->C : 0; }
extern "C" int povu_hip_shards_get(const povu_hip_shards *s, uint32_t rank, povu_hip_shard_info *out)
{
	if (!s || !out || rank >= s->world)
		return 1;
	const auto &p = s->parts[rank];
	out->n_vtx = p.nv;
	out->n_links = p.ne;
	out->n_components = p.nc;
	out->weight = p.weight;
	out->bytes = p.bytes;
	out->device_ptr = static_cast<const char *>(s->block) + p.off;
	return 0;
}
extern "C" int povu_hip_shards_times(const povu_hip_shards *s, double out_ms[3])
{
	if (!s || !out_ms)
		return 1;
	out_ms[0] = s->ms[0], out_ms[1] = s->ms[1], out_ms[2] = s->ms[2];
	return 0;
}
extern "C" int povu_hip_shards_export(const povu_hip_shards *s, povu_hip_ctx *ctx, uint32_t rank, void *dst)
{
	if (!s || !ctx || !dst || rank >= s->world)
		return 1;
	if (hipSetDevice(s->device) != hipSuccess)
		return 2;
	const auto &p = s->parts[rank];
	return hipMemcpy(dst, static_cast<const char *>(s->block) + p.off, p.bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
extern "C" void povu_hip_shards_free(povu_hip_shards *s) { delete s; }

// ---------------------------------------------------------------- shard loader
extern "C" int povu_hip_graph_upload_shard(povu_hip_ctx *ctx, const void *packed, size_t bytes, int on_device, char *err, size_t errlen)
{
	ResidentGraph g;
	XferScope xfer(ctx);
	try {
		if (!ctx || !packed || bytes < 256)
			throw HipError("bad shard");
		HIP_CHECK(hipSetDevice(ctx->device));
		ctx->wait_tail();
		hipStream_t s = ctx->stream;
		const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
		uint64_t h[8];
		if (on_device)
			HIP_CHECK(hipMemcpy(h, packed, 64, hipMemcpyDeviceToHost));
		else
			memcpy(h, packed, 64);
		if (h[0] != SHARD_MAGIC)
			throw HipError("not a packed shard");
		const uint32_t nv = (uint32_t)h[1], ne = (uint32_t)h[2], nc = (uint32_t)h[3];
		const ShardLayout L(nv, ne, nc);
		if (L.bytes != bytes)
			throw HipError("packed shard has the wrong size");
		if (nv == 0) { // a rank that owns no component: nothing resident, decompose returns an empty forest
			free_resident_graph(ctx->g);
			ctx->have_state = false;
			ctx->shard_comp_ids.clear();
			ctx->shard_total_components = (uint32_t)h[4];
			return 0;
		}
		check_graph_size(nv, ne);
		free_resident_graph(ctx->g);
		ctx->have_state = false;
		alloc_resident_graph(ctx->graph_arena, g, nv, ne, true);
		const char *b = static_cast<const char *>(packed);
		hipEvent_t e0, e1;
		HIP_CHECK(hipEventCreate(&e0));
		HIP_CHECK(hipEventCreate(&e1));
		HIP_CHECK(hipEventRecord(e0, s));
		HIP_CHECK(copy_async(g.vid, b + L.vid, (size_t)nv * 4, kind, s));
		if (ne) {
			HIP_CHECK(copy_async(g.v1, b + L.v1, (size_t)ne * 4, kind, s));
			HIP_CHECK(copy_async(g.v2, b + L.v2, (size_t)ne * 4, kind, s));
			HIP_CHECK(copy_async(g.s1, b + L.s1, ne, kind, s));
			HIP_CHECK(copy_async(g.s2, b + L.s2, ne, kind, s));
		}
		HIP_CHECK(copy_async(g.tip, b + L.tip, nv, kind, s));
		std::vector<uint32_t> ids(nc);
		if (nc)
			HIP_CHECK(copy_async(ids.data(), b + L.ids, (size_t)nc * 4, on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost, s));
		HIP_CHECK(hipEventRecord(e1, s));
		HIP_CHECK(hipStreamSynchronize(s));
		(void)hipEventElapsedTime(&g.h2d_ms, e0, e1);
		(void)hipEventDestroy(e0);
		(void)hipEventDestroy(e1);
		build_global_csr(g, ctx->upload_tmp, s);
		ctx->g = g;
		ctx->shard_comp_ids = std::move(ids);
		ctx->shard_total_components = (uint32_t)h[4];
		return 0;
	} catch (const std::exception &e) {
		if (ctx)
			(void)hipStreamSynchronize(ctx->stream);
		set_err(err, errlen, e.what());
		return 1;
	}
}

extern "C" uint32_t povu_hip_shard_total_components(const povu_hip_ctx *ctx) { return ctx ? ctx->shard_total_components : 0; }

extern "C" int povu_hip_forest_globalize(povu_hip_forest *f, const povu_hip_ctx *ctx)
{
	if (!f || !ctx || ctx->shard_total_components == 0)
		return 1;
	for (auto &t : f->trees) {
		if (t.component_id == 0 || t.component_id > ctx->shard_comp_ids.size())
			return 2;
		t.component_id = ctx->shard_comp_ids[t.component_id - 1];
	}
	f->total_components = ctx->shard_total_components;
	return 0;
}

// ---------------------------------------------------------------- forest wire format
namespace
{
inline size_t pad64(size_t b) { return (b + 63) & ~size_t(63); }
struct ForestLayout {
	size_t meta, a, z, parent, aor, zor, bytes;
	ForestLayout(size_t n_trees, size_t total)
	{
		size_t o = 64;
		meta = o, o += pad64(n_trees * 16);
		a = o, o += pad64(total * 4);
		z = o, o += pad64(total * 4);
		parent = o, o += pad64(total * 4);
		aor = o, o += pad64(total);
		zor = o, o += pad64(total);
		bytes = o;
	}
};
static constexpr uint64_t FOREST_MAGIC = 0x31747372665F7670ull; // "pv_frst1"
// copies `n` bytes with a few threads when the block is large (page-locked host memory on both sides)
void big_copy(void *dst, const void *src, size_t n)
{
	if (n < (size_t(8) << 20)) {
		memcpy(dst, src, n);
		return;
	}
	const unsigned nt = 4;
	std::vector<std::thread> th;
	const size_t chunk = (n / nt + 4095) & ~size_t(4095);
	for (unsigned t = 0; t < nt; t++) {
		const size_t b = (size_t)t * chunk, e = std::min(n, b + chunk);
		if (b < e)
			th.emplace_back([=] { memcpy((char *)dst + b, (const char *)src + b, e - b); });
	}
	for (auto &x : th)
		x.join();
}
} // namespace

extern "C" size_t povu_hip_forest_pack_size(const povu_hip_forest *f)
{
	if (!f)
		return 0;
	size_t total = 0;
	for (const auto &t : f->trees)
		total += t.n_pvst;
	return ForestLayout(f->trees.size(), total).bytes;
}

extern "C" int povu_hip_forest_pack(const povu_hip_forest *f, void *dst, size_t cap)
{
	if (!f || !dst)
		return 1;
	if (!f->hairpins.empty() || !f->sub_fam.empty())
		return 4; // the wire format carries no hairpin boundaries and no subflubble labels: refuse rather than drop them
	size_t total = 0;
	for (const auto &t : f->trees)
		total += t.n_pvst;
	const ForestLayout L(f->trees.size(), total);
	if (cap < L.bytes)
		return 2;
	char *b = static_cast<char *>(dst);
	uint64_t *h = reinterpret_cast<uint64_t *>(b);
	memset(h, 0, 64);
	h[0] = f->trees.size(), h[1] = total, h[2] = f->total_components, h[3] = FOREST_MAGIC;
	uint32_t *meta = reinterpret_cast<uint32_t *>(b + L.meta);
	size_t at = 0;
	for (size_t i = 0; i < f->trees.size(); i++) {
		povu_hip_tree t;
		if (povu_hip_forest_get(f, (uint32_t)i, &t) != 0)
			return 3;
		meta[4 * i] = t.component_id, meta[4 * i + 1] = t.n_vtx, meta[4 * i + 2] = t.n_links, meta[4 * i + 3] = t.n_pvst;
		memcpy(b + L.a + at * 4, t.a_id, (size_t)t.n_pvst * 4);
		memcpy(b + L.z + at * 4, t.z_id, (size_t)t.n_pvst * 4);
		memcpy(b + L.parent + at * 4, t.parent, (size_t)t.n_pvst * 4);
		memcpy(b + L.aor + at, t.a_or, t.n_pvst);
		memcpy(b + L.zor + at, t.z_or, t.n_pvst);
		at += t.n_pvst;
	}
	return 0;
}

// one packed forest -> an extra block of `out` + its trees
static void adopt_packed(povu_hip_forest &out, std::shared_ptr<PinnedPool> pool, const char *b, size_t bytes)
{
	if (bytes < 64)
		throw HipError("packed forest too short");
	const uint64_t *h = reinterpret_cast<const uint64_t *>(b);
	if (h[3] != FOREST_MAGIC)
		throw HipError("packed forest: bad magic word");
	// (counts from the wire: bound them by the buffer before any arithmetic that could wrap)
	if (h[0] > bytes / 16 || h[1] > bytes / 4)
		throw HipError("packed forest has the wrong size");
	const size_t n_trees = h[0], total = h[1];
	const ForestLayout L(n_trees, total);
	if (L.bytes > bytes)
		throw HipError("packed forest has the wrong size");
	out.total_components = std::max<uint32_t>(out.total_components, (uint32_t)h[2]);
	if (n_trees == 0)
		return;
	povu_hip_forest::ExtraBlock blk;
	blk.pool = pool;
	blk.p = pool->get(povu_hip_forest::ExtraBlock::bytes_for(total), blk.cap);
	blk.carve(total);
	big_copy(blk.a, b + L.a, total * 4);
	big_copy(blk.z, b + L.z, total * 4);
	big_copy(blk.parent, b + L.parent, total * 4);
	big_copy(blk.aor, b + L.aor, total);
	big_copy(blk.zor, b + L.zor, total);
	const int bi = (int)out.extra.size();
	out.extra.push_back(std::move(blk));
	const uint32_t *meta = reinterpret_cast<const uint32_t *>(b + L.meta);
	size_t at = 0;
	for (size_t i = 0; i < n_trees; i++) {
		povu_hip_forest::Tree t{};
		t.component_id = meta[4 * i], t.n_vtx = meta[4 * i + 1], t.n_links = meta[4 * i + 2], t.n_pvst = meta[4 * i + 3];
		t.off = at;
		t.hp_off = 0;
		t.n_hairpins = 0;
		t.blk = bi;
		if (at + t.n_pvst > total)
			throw HipError("packed forest: tree sizes do not add up");
		at += t.n_pvst;
		out.trees.push_back(t);
	}
	if (at != total)
		throw HipError("packed forest: tree sizes do not add up");
}

static void sort_trees(povu_hip_forest &f)
{
	std::stable_sort(f.trees.begin(), f.trees.end(),
			 [](const povu_hip_forest::Tree &a, const povu_hip_forest::Tree &b) { return a.component_id < b.component_id; });
}

extern "C" povu_hip_forest *povu_hip_forest_merge(povu_hip_ctx *ctx, const void *const *packed, const size_t *bytes, uint32_t n,
						  char *err, size_t errlen)
{
	try {
		if (!ctx || (n && (!packed || !bytes)))
			throw HipError("bad arguments");
		auto f = std::make_unique<povu_hip_forest>();
		f->pool = ctx->pool;
		for (uint32_t i = 0; i < n; i++)
			adopt_packed(*f, ctx->pool, static_cast<const char *>(packed[i]), bytes[i]);
		sort_trees(*f);
		return f.release();
	} catch (const std::exception &e) {
		set_err(err, errlen, e.what());
		return nullptr;
	}
}

// ---------------------------------------------------------------- merging without copies
// `m`'s blocks, trees, hairpin boundaries and subflubble labels move into `out` (m is left empty): what the root does with
// its own forest in every gather, and what the one-process engine (multi.hip) does with every worker's forest.
void adopt_forest(povu_hip_forest &out, povu_hip_forest &m)
{
	// a POVU_HIP_F_ASYNC forest whose arrays are still on their way: the blocks change hands now, the event behind their
	// copies goes with them (the merged forest waits for it before anybody reads)
	if (m.pending) {
		if (m.ev1) {
			out.more_events.push_back(m.ev1);
			m.ev1 = nullptr;
		}
		for (hipEvent_t e : m.more_events)
			out.more_events.push_back(e);
		m.more_events.clear();
		m.pending = false;
		out.pending = true;
	}
	const int base = (int)out.extra.size();
	int own = -1;
	if (m.block) {
		povu_hip_forest::ExtraBlock b;
		b.pool = m.pool;
		b.p = m.block;
		b.cap = m.block_cap;
		b.seg = m.block_seg;
		b.carve(m.total_entries);
		b.sub_ai = std::move(m.sub_ai);
		b.sub_zi = std::move(m.sub_zi);
		b.sub_fam = std::move(m.sub_fam);
		b.subx = std::move(m.subx);
		own = (int)out.extra.size() + (int)m.extra.size();
		m.block = nullptr;
		m.block_cap = m.block_bytes = m.total_entries = 0;
		m.block_seg = -1;
		for (auto &e : m.extra)
			out.extra.push_back(std::move(e));
		out.extra.push_back(std::move(b));
	} else {
		for (auto &e : m.extra)
			out.extra.push_back(std::move(e));
	}
	m.extra.clear();
	const size_t hp_base = out.hairpins.size() / 2;
	out.hairpins.insert(out.hairpins.end(), m.hairpins.begin(), m.hairpins.end());
	m.hairpins.clear();
	for (auto t : m.trees) {
		t.blk = t.blk < 0 ? own : base + t.blk;
		t.hp_off += hp_base;
		out.trees.push_back(t);
	}
	m.trees.clear();
	out.total_components = std::max(out.total_components, m.total_components);
	for (const auto &e : out.extra)
		if (!e.sub_fam.empty() && out.sub_fam.empty())
			out.sub_fam.assign(1, 0); // (povu_hip_forest_get_sub: "this forest carries labels"; the arrays are the blocks' own)
}

extern "C" int povu_hip_shard_component_ids(const povu_hip_ctx *ctx, const uint32_t **ids, uint32_t *n)
{
	if (!ctx || !ids || !n || ctx->shard_comp_ids.empty())
		return 1;
	*ids = ctx->shard_comp_ids.data();
	*n = (uint32_t)ctx->shard_comp_ids.size();
	return 0;
}

// ---------------------------------------------------------------- gather through shared memory (several processes, one node)
// Every rank's PVST block already sits in page-locked HOST memory when its decompose returns -- copied there by its own
// GPU over its own PCIe link.  When that memory is a named shared-memory segment (povu_hip_share_results) the root only
// has to MAP it: no block goes back to a device, over xGMI and down the root's link again.  What travels between the
// processes is one 64-byte descriptor per rank (any transport: an RCCL all-gather, torch.distributed, a pipe).
static constexpr uint64_t SHARE_MAGIC = 0x3165726168735F76ull; // "v_share1"
static constexpr uint64_t SHARE_EMPTY = ~0ull;

// a forest whose trees sit in several blocks -> one block of its own pool, trees back to back (in place)
static void compact_in_place(povu_hip_forest &f);

// What the five arrays do not hold -- the leaf passes' labels (ai, zi, line letter per PVST vertex), the hairpin boundaries,
// the extended trees of `-s` -- travels in a SECOND shared-memory segment of the rank's pool, announced in the header of the
// tree table (words 3..5: segment, its mapped size, bytes used; ~0 = none).  Sections, each padded to 64 bytes, behind a
// 128-byte header {magic, PVST vertices, hairpin pairs, flags, components + 1 of the extended trees, their vertices, their
// child entries, bytes}: [ai][zi] u32 x total, [letter] u8 x total | [pairs] 2 x u64 | [voff] u64, [counts] u32 x 3,
// [letter][or1][or2][route] u8 x vertices, [id1][id2] u32 x vertices, [coff] u32 x (vertices + 1), [child] u32.
static constexpr uint64_t SHAREX_MAGIC = 0x7865726168735F76ull; // "v_sharex"
namespace
{
struct XLayout {
	size_t total = 0, pairs = 0, c1 = 0, nv = 0, nc = 0;
	bool labels = false, hp = false, sub = false;
	size_t o_ai = 0, o_zi = 0, o_fam = 0, o_hp = 0, o_voff = 0, o_cnt = 0, o_xfam = 0, o_or1 = 0, o_or2 = 0, o_route = 0, o_id1 = 0, o_id2 = 0,
	       o_coff = 0, o_child = 0, bytes = 0;
	void plan()
	{
		size_t q = 128;
		auto sec = [&](size_t b) {
			const size_t r = q;
			q += (b + 63) & ~size_t(63);
			return r;
		};
		if (labels)
			o_ai = sec(total * 4), o_zi = sec(total * 4), o_fam = sec(total);
		if (hp)
			o_hp = sec(pairs * 16);
		if (sub) {
			o_voff = sec(c1 * 8), o_cnt = sec((c1 ? c1 - 1 : 0) * 12);
			o_xfam = sec(nv), o_or1 = sec(nv), o_or2 = sec(nv), o_route = sec(nv);
			o_id1 = sec(nv * 4), o_id2 = sec(nv * 4), o_coff = sec((nv + 1) * 4), o_child = sec(nc * 4);
		}
		bytes = q;
	}
};
} // namespace

extern "C" int povu_hip_forest_share(povu_hip_forest *f, uint64_t desc[8])
{
	if (!f || !desc)
		return 1;
	try {
		const bool has_x = !f->hairpins.empty() || !f->sub_fam.empty() || f->subx;
		if (has_x && !f->extra.empty())
			return 4; // (labels / boundaries of a MERGED forest: share the parts; a rank's own forest has no extra blocks)
		if (has_x)
			f->ready(); // (the labels were copied by the pass itself; nothing of them is still in flight after this)
		// (no wait for a POVU_HIP_F_ASYNC forest here: the descriptor and the tree table do not depend on the arrays still in
		// flight.  The READER must not look at them before this rank has waited for the forest -- povu_hip_forest_wait --
		// and told it so: in a loop of collectives, by taking part in the next one.)
		std::fill(desc, desc + 8, 0ull);
		desc[0] = SHARE_MAGIC;
		desc[1] = SHARE_EMPTY;
		desc[5] = f->total_components;
		if (f->trees.empty())
			return 0;
		if (!f->extra.empty())
			compact_in_place(*f);
		if (f->block_seg < 0)
			return 2; // the block is no shared segment: povu_hip_share_results was not called on the context
		const size_t nt = f->trees.size(), meta_off = f->block_bytes;
		if (meta_off + povu_hip_forest::meta_bytes(nt) > f->block_cap)
			return 3;
		char *b = static_cast<char *>(f->block);
		uint32_t *meta = reinterpret_cast<uint32_t *>(b + meta_off + 64);
		for (size_t i = 0; i < nt; i++) {
			const auto &t = f->trees[i];
			if (t.blk >= 0 || t.off + t.n_pvst > f->total_entries)
				return 3;
			uint32_t *q = meta + 8 * i;
			q[0] = t.component_id, q[1] = t.n_vtx, q[2] = t.n_links, q[3] = t.n_pvst;
			q[4] = (uint32_t)(t.off & 0xFFFFFFFFu), q[5] = (uint32_t)((uint64_t)t.off >> 32), q[6] = q[7] = 0;
		}
		uint64_t *mh = reinterpret_cast<uint64_t *>(b + meta_off);
		mh[0] = SHARE_MAGIC, mh[1] = nt, mh[2] = f->total_entries;
		mh[3] = ~0ull, mh[4] = mh[5] = 0;
		if (has_x) {
			for (size_t i = 0; i < nt; i++) { // hairpin pairs and the component in the extended trees: the table's two spare words
				meta[8 * i + 6] = f->trees[i].n_hairpins;
				meta[8 * i + 7] = f->trees[i].sub_c;
			}
			XLayout L;
			L.total = f->total_entries;
			L.labels = !f->sub_fam.empty();
			L.hp = !f->hairpins.empty();
			L.pairs = f->hairpins.size() / 2;
			const SubForest *x = f->subx.get();
			L.sub = x != nullptr;
			if (x)
				L.c1 = x->voff.size(), L.nv = x->n_vtx, L.nc = x->n_child;
			if (L.labels && (f->sub_ai.size() < L.total || f->sub_zi.size() < L.total || f->sub_fam.size() < L.total))
				return 3;
			L.plan();
			if (f->xblk)
				f->pool->put(f->xblk, f->xblk_cap, f->xblk_seg);
			f->xblk = f->pool->get(L.bytes, f->xblk_cap, &f->xblk_seg);
			if (f->xblk_seg < 0)
				return 2;
			char *xb = static_cast<char *>(f->xblk);
			uint64_t *h = reinterpret_cast<uint64_t *>(xb);
			std::fill(h, h + 16, 0ull);
			h[0] = SHAREX_MAGIC, h[1] = L.total, h[2] = L.pairs, h[3] = (L.labels ? 1u : 0u) | (L.hp ? 2u : 0u) | (L.sub ? 4u : 0u);
			h[4] = L.c1, h[5] = L.nv, h[6] = L.nc, h[7] = L.bytes;
			if (L.labels) {
				memcpy(xb + L.o_ai, f->sub_ai.data(), L.total * 4);
				memcpy(xb + L.o_zi, f->sub_zi.data(), L.total * 4);
				memcpy(xb + L.o_fam, f->sub_fam.data(), L.total);
			}
			if (L.hp)
				memcpy(xb + L.o_hp, f->hairpins.data(), L.pairs * 16);
			if (x) {
				memcpy(xb + L.o_voff, x->voff.data(), L.c1 * 8);
				if (L.c1 > 1)
					memcpy(xb + L.o_cnt, x->counts.data(), (L.c1 - 1) * 12);
				if (L.nv) {
					memcpy(xb + L.o_xfam, x->fam, L.nv);
					memcpy(xb + L.o_or1, x->or1, L.nv);
					memcpy(xb + L.o_or2, x->or2, L.nv);
					memcpy(xb + L.o_route, x->route, L.nv);
					memcpy(xb + L.o_id1, x->id1, L.nv * 4);
					memcpy(xb + L.o_id2, x->id2, L.nv * 4);
				}
				memcpy(xb + L.o_coff, x->coff, (L.nv + 1) * 4);
				if (L.nc)
					memcpy(xb + L.o_child, x->child, L.nc * 4);
			}
			mh[3] = (uint64_t)f->xblk_seg, mh[4] = f->xblk_cap, mh[5] = L.bytes;
		}
		__atomic_thread_fence(__ATOMIC_RELEASE); // (the descriptor leaves through a system call anyway)
		desc[1] = (uint64_t)f->block_seg;
		desc[2] = f->block_cap;
		desc[3] = nt;
		desc[4] = f->total_entries;
		desc[6] = meta_off;
		return 0;
	} catch (const std::exception &) {
		return 5;
	}
}

extern "C" povu_hip_forest *povu_hip_forest_attach(povu_hip_ctx *ctx, povu_hip_forest *own, uint32_t own_rank, const char *job_tag,
						   const uint64_t *descs, uint32_t n, char *err, size_t errlen)
{
	try {
		if (!ctx || !job_tag || (n && !descs))
			throw HipError("attach: bad arguments");
		// (the tag becomes part of a shared-memory name: the rule of povu_hip_share_results, which made the segments; the ranks'
		// own tags are "<job>.<rank>", so the job's leaves room for the rank)
		if (!*job_tag || strlen(job_tag) > 84 || strchr(job_tag, '/'))
			throw HipError("attach: bad job tag");
		auto out = std::make_unique<povu_hip_forest>();
		out->pool = ctx->pool;
		for (uint32_t i = 0; i < n; i++) {
			const uint64_t *d = descs + 8 * (size_t)i;
			if (d[0] != SHARE_MAGIC)
				throw HipError("attach: bad descriptor from rank " + std::to_string(d[7]));
			if (d[7] >= n) // (the rank word comes from another process and names a segment: it must be one of the n ranks)
				throw HipError("attach: descriptor " + std::to_string(i) + " names rank " + std::to_string(d[7]) + " of " + std::to_string(n));
			out->total_components = std::max<uint32_t>(out->total_components, (uint32_t)d[5]);
			if (own && d[7] == own_rank)
				continue; // the root's own trees stay where they are (below)
			if (d[1] == SHARE_EMPTY)
				continue;
			const std::string name = PinnedPool::segment_name(std::string(job_tag) + "." + std::to_string(d[7]), (int)d[1]);
			const size_t seg_bytes = d[2], nt = d[3], total = d[4], meta_off = d[6];
			// (numbers from another process: bound them before any arithmetic on them)
			if (nt == 0 || nt > seg_bytes / 32 || total > seg_bytes / 4 || meta_off > seg_bytes ||
			    meta_off + povu_hip_forest::meta_bytes(nt) > seg_bytes || povu_hip_forest::ExtraBlock::bytes_for(total) - 64 > meta_off)
				throw HipError("attach: descriptor of rank " + std::to_string(d[7]) + " does not fit its segment");
			// maps a segment of another rank read-only (mappings are kept by the context: the ranks reuse their segments)
			auto map_segment = [&](const std::string &nm, size_t bytes) -> const char * {
				auto it = ctx->attached.find(nm);
				if (it == ctx->attached.end()) {
					const int fd = shm_open(nm.c_str(), O_RDONLY, 0);
					if (fd < 0)
						throw HipError("attach: cannot open the result segment " + nm);
					struct stat st;
					if (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) {
						(void)close(fd);
						throw HipError("attach: the result segment " + nm + " is smaller than its descriptor says");
					}
					void *p = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
					(void)close(fd);
					if (p == MAP_FAILED)
						throw HipError("attach: cannot map the result segment " + nm);
					it = ctx->attached.emplace(nm, povu_hip_ctx::Mapped{p, bytes}).first;
				} else if (it->second.bytes < bytes) {
					throw HipError("attach: the result segment " + nm + " changed its size");
				}
				return static_cast<const char *>(it->second.p);
			};
			const char *b = map_segment(name, seg_bytes);
			const uint64_t *mh = reinterpret_cast<const uint64_t *>(b + meta_off);
			if (mh[0] != SHARE_MAGIC || mh[1] != nt || mh[2] != total)
				throw HipError("attach: the tree table of rank " + std::to_string(d[7]) + " does not match its descriptor");
			povu_hip_forest::ExtraBlock blk; // (no pool: the memory is the other rank's)
			blk.p = const_cast<char *>(b);
			blk.cap = seg_bytes;
			blk.carve(total);
			// what the five arrays do not hold (povu_hip_forest_share): labels, hairpin boundaries, the extended trees of -s
			XLayout L;
			const char *xb = nullptr;
			if (mh[3] != ~0ull) {
				const size_t xcap = mh[4], xbytes = mh[5];
				if (xbytes < 128 || xbytes > xcap)
					throw HipError("attach: bad extras segment of rank " + std::to_string(d[7]));
				xb = map_segment(PinnedPool::segment_name(std::string(job_tag) + "." + std::to_string(d[7]), (int)mh[3]), xcap);
				const uint64_t *h = reinterpret_cast<const uint64_t *>(xb);
				if (h[0] != SHAREX_MAGIC || h[1] != total || h[7] != xbytes)
					throw HipError("attach: the extras segment of rank " + std::to_string(d[7]) + " does not match its forest");
				L.total = total, L.pairs = h[2], L.labels = h[3] & 1u, L.hp = h[3] & 2u, L.sub = h[3] & 4u;
				L.c1 = h[4], L.nv = h[5], L.nc = h[6];
				// (numbers from another process: bound them before they size anything)
				if (L.pairs > xbytes / 16 || L.c1 > xbytes / 8 || L.nv > xbytes || L.nc > xbytes / 4)
					throw HipError("attach: the extras segment of rank " + std::to_string(d[7]) + " names sizes beyond itself");
				L.plan();
				if (L.bytes != xbytes)
					throw HipError("attach: the extras segment of rank " + std::to_string(d[7]) + " has another layout than its header says");
				if (L.labels) {
					blk.sub_ai.assign(reinterpret_cast<const uint32_t *>(xb + L.o_ai), reinterpret_cast<const uint32_t *>(xb + L.o_ai) + total);
					blk.sub_zi.assign(reinterpret_cast<const uint32_t *>(xb + L.o_zi), reinterpret_cast<const uint32_t *>(xb + L.o_zi) + total);
					blk.sub_fam.assign(reinterpret_cast<const uint8_t *>(xb + L.o_fam), reinterpret_cast<const uint8_t *>(xb + L.o_fam) + total);
					if (out->sub_fam.empty())
						out->sub_fam.assign(1, 0); // (povu_hip_forest_get_sub: "this forest carries labels"; the arrays are the blocks' own)
				}
				if (L.sub) { // a view into the mapped segment (no pool, no block of its own: nothing to give back)
					auto x = std::make_shared<SubForest>();
					const uint64_t *vo = reinterpret_cast<const uint64_t *>(xb + L.o_voff);
					x->voff.assign(vo, vo + L.c1);
					const uint32_t *cn = reinterpret_cast<const uint32_t *>(xb + L.o_cnt);
					x->counts.assign(cn, cn + (L.c1 ? 3 * (L.c1 - 1) : 0));
					x->n_vtx = L.nv, x->n_child = L.nc;
					if (!x->voff.empty() && x->voff.back() != L.nv)
						throw HipError("attach: the extended trees of rank " + std::to_string(d[7]) + " do not add up");
					x->fam = reinterpret_cast<const uint8_t *>(xb + L.o_xfam), x->or1 = reinterpret_cast<const uint8_t *>(xb + L.o_or1);
					x->or2 = reinterpret_cast<const uint8_t *>(xb + L.o_or2), x->route = reinterpret_cast<const uint8_t *>(xb + L.o_route);
					x->id1 = reinterpret_cast<const uint32_t *>(xb + L.o_id1), x->id2 = reinterpret_cast<const uint32_t *>(xb + L.o_id2);
					x->coff = reinterpret_cast<const uint32_t *>(xb + L.o_coff), x->child = reinterpret_cast<const uint32_t *>(xb + L.o_child);
					blk.subx = x;
				}
			}
			const int bi = (int)out->extra.size();
			out->extra.push_back(std::move(blk));
			const uint32_t *meta = reinterpret_cast<const uint32_t *>(b + meta_off + 64);
			size_t hp_seen = 0;
			for (size_t k = 0; k < nt; k++) {
				const uint32_t *q = meta + 8 * k;
				povu_hip_forest::Tree t{};
				t.component_id = q[0], t.n_vtx = q[1], t.n_links = q[2], t.n_pvst = q[3];
				t.off = (size_t)q[4] | ((size_t)q[5] << 32);
				t.blk = bi;
				if (t.off > total || t.n_pvst > total - t.off)
					throw HipError("attach: a tree of rank " + std::to_string(d[7]) + " lies outside its block");
				if (xb) {
					t.n_hairpins = L.hp ? q[6] : 0;
					t.sub_c = q[7];
					if (L.sub && (size_t)t.sub_c + 1 >= L.c1)
						throw HipError("attach: a tree of rank " + std::to_string(d[7]) + " names a component its extended trees do not have");
					if (t.n_hairpins) { // the rank's pairs sit in tree order: they go behind the merged forest's, tree by tree
						if (hp_seen + t.n_hairpins > L.pairs)
							throw HipError("attach: the hairpin boundaries of rank " + std::to_string(d[7]) + " do not add up");
						const uint64_t *hp = reinterpret_cast<const uint64_t *>(xb + L.o_hp) + 2 * hp_seen;
						t.hp_off = out->hairpins.size() / 2;
						out->hairpins.insert(out->hairpins.end(), hp, hp + 2 * (size_t)t.n_hairpins);
						hp_seen += t.n_hairpins;
					}
				}
				out->trees.push_back(t);
			}
		}
		if (own)
			adopt_forest(*out, *own);
		sort_trees(*out);
		return out.release();
	} catch (const std::exception &e) {
		set_err(err, errlen, e.what());
		return nullptr;
	}
}

// ---------------------------------------------------------------- RCCL (loaded on first use)
namespace
{
static_assert(sizeof(ncclUniqueId) == POVU_HIP_COMM_ID_BYTES, "ncclUniqueId size");
} // namespace

struct povu_hip_comm {
	ncclComm_t comm = nullptr;
	povu_hip_ctx *ctx = nullptr;
	uint32_t rank = 0, world = 1;
	hipStream_t side = nullptr; // the root's sends / receives run beside its own kernels
	uint64_t *dsmall = nullptr; // device scratch for the size tables
	uint64_t *hsmall = nullptr; // pinned
	Arena stage;		    // device staging of forest blocks
	double ms[2] = {0, 0};
};

extern "C" int povu_hip_comm_unique_id(char id[POVU_HIP_COMM_ID_BYTES], char *err, size_t errlen)
{
	try {
		ncclUniqueId u;
		NCCL_CHECK(rccl().GetUniqueId(&u));
		memcpy(id, &u, sizeof u);
		return 0;
	} catch (const std::exception &e) {
		set_err(err, errlen, e.what());
		return 1;
	}
}

extern "C" povu_hip_comm *povu_hip_comm_create(povu_hip_ctx *ctx, const char id[POVU_HIP_COMM_ID_BYTES], uint32_t rank, uint32_t world,
					       char *err, size_t errlen)
{
	std::unique_ptr<povu_hip_comm> c;
	try {
		if (!ctx || !id || world == 0 || rank >= world || world > 512)
			throw HipError("bad communicator arguments");
		HIP_CHECK(hipSetDevice(ctx->device));
		c = std::make_unique<povu_hip_comm>();
		c->ctx = ctx;
		c->rank = rank;
		c->world = world;
		ncclUniqueId u;
		memcpy(&u, id, sizeof u);
		NCCL_CHECK(rccl().CommInitRank(&c->comm, (int)world, u, (int)rank));
		HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
		HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&c->dsmall), 8 * 8 * (size_t)world + 128));
		HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&c->hsmall), 8 * 8 * (size_t)world + 128, hipHostMallocDefault));
		return c.release();
	} catch (const std::exception &e) {
		if (c)
			povu_hip_comm_destroy(c.release());
		set_err(err, errlen, e.what());
		return nullptr;
	}
}

extern "C" void povu_hip_comm_destroy(povu_hip_comm *c)
{
	if (!c)
		return;
	if (c->ctx)
		(void)hipSetDevice(c->ctx->device);
	if (c->comm) {
		try {
			(void)rccl().CommDestroy(c->comm);
		} catch (...) {
		}
	}
	if (c->side)
		(void)hipStreamDestroy(c->side);
	if (c->dsmall)
		(void)hipFree(c->dsmall);
	if (c->hsmall)
		(void)hipHostFree(c->hsmall);
	c->stage.release();
	delete c;
}

// Every RCCL call of a communicator is issued on ITS stream (c->side): one communicator, one stream.  A rank that finds
// something wrong BEFORE a transfer says so in the status word all ranks exchange first, so that nobody is left waiting in
// a send or a receive for a peer that has already given up; a group that was opened is closed on every path.
extern "C" int povu_hip_comm_scatter(povu_hip_comm *c, const povu_hip_shards *shards, povu_hip_ctx *dst, char *err, size_t errlen)
{
	bool in_group = false;
	try {
		if (!c || !dst || dst != c->ctx)
			throw HipError("scatter: the destination context must be the communicator's");
		const bool root = c->rank == 0;
		HIP_CHECK(hipSetDevice(dst->device));
		Rccl &R = rccl();
		hipStream_t s = c->side;
		const double t0 = now_ms();
		// sizes first (one broadcast; an all-ones size = the root cannot serve this call), then every rank's shard point to
		// point, all transfers in one group
		const bool root_ok = !root || (shards && shards->world == c->world);
		uint64_t *h = c->hsmall;
		if (root)
			for (uint32_t r = 0; r < c->world; r++)
				h[r] = root_ok ? shards->parts[r].bytes : ~0ull;
		HIP_CHECK(copy_async(c->dsmall, h, 8 * (size_t)c->world, hipMemcpyHostToDevice, s));
		NCCL_CHECK(R.Broadcast(c->dsmall, c->dsmall, c->world, ncclUint64, 0, c->comm, s));
		HIP_CHECK(copy_async(h, c->dsmall, 8 * (size_t)c->world, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		if (h[c->rank] == ~0ull)
			throw HipError("scatter: the root needs a partition for exactly `world` ranks");
		const size_t my_bytes = h[c->rank];
		// second word of the handshake: can every receiver take its shard?  (a failed reservation must not leave the root
		// blocked in its sends)
		char *buf = nullptr;
		uint64_t ok = 1;
		if (!root) {
			try {
				dst->shard_buf.reserve(my_bytes + 256);
				buf = dst->shard_buf.take<char>(my_bytes);
			} catch (const std::exception &) {
				ok = 0;
			}
		}
		h[0] = ok;
		uint64_t *dmy = c->dsmall, *dall = c->dsmall + 4;
		HIP_CHECK(copy_async(dmy, h, 8, hipMemcpyHostToDevice, s));
		NCCL_CHECK(R.AllGather(dmy, dall, 1, ncclUint64, c->comm, s));
		HIP_CHECK(copy_async(h + 4, dall, 8 * (size_t)c->world, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		for (uint32_t r = 0; r < c->world; r++)
			if (!h[4 + r])
				throw HipError("scatter: rank " + std::to_string(r) + " has no room for its shard");
		const void *mine = nullptr;
		if (root) {
			const char *blk = static_cast<const char *>(shards->block);
			NCCL_CHECK(R.GroupStart());
			in_group = true;
			for (uint32_t r = 1; r < c->world; r++)
				NCCL_CHECK(R.Send(blk + shards->parts[r].off, shards->parts[r].bytes, ncclChar, (int)r, c->comm, s));
			in_group = false;
			NCCL_CHECK(R.GroupEnd());
			mine = blk + shards->parts[0].off; // same device: loaded straight from the partition, beside the sends
		} else {
			NCCL_CHECK(R.Recv(buf, my_bytes, ncclChar, 0, c->comm, s));
			HIP_CHECK(hipStreamSynchronize(s));
			mine = buf;
		}
		char e2[512] = {0};
		if (povu_hip_graph_upload_shard(dst, mine, my_bytes, 1, e2, sizeof e2) != 0)
			throw HipError(std::string("scatter: ") + e2);
		if (root)
			HIP_CHECK(hipStreamSynchronize(s)); // the partition may be freed once this returns
		c->ms[0] = now_ms() - t0;
		return 0;
	} catch (const std::exception &e) {
		if (in_group) {
			try {
				(void)rccl().GroupEnd();
			} catch (...) {
			}
		}
		if (c && c->ctx) {
			(void)hipStreamSynchronize(c->ctx->stream);
			if (c->side)
				(void)hipStreamSynchronize(c->side);
		}
		set_err(err, errlen, e.what());
		return 1;
	}
}

// a forest whose trees sit in several blocks (sequential redo of some components) -> one block, trees back to back
static std::unique_ptr<povu_hip_forest> compact_forest(const povu_hip_forest *f)
{
	auto out = std::make_unique<povu_hip_forest>();
	out->pool = f->pool;
	out->total_components = f->total_components;
	size_t total = 0;
	for (const auto &t : f->trees)
		total += t.n_pvst;
	out->meta_reserve = f->trees.size() + 1;
	out->alloc(total);
	size_t at = 0;
	for (size_t i = 0; i < f->trees.size(); i++) {
		povu_hip_tree t;
		if (povu_hip_forest_get(f, (uint32_t)i, &t) != 0)
			throw HipError("gather: bad forest");
		memcpy(out->a_id.p + at, t.a_id, (size_t)t.n_pvst * 4);
		memcpy(out->z_id.p + at, t.z_id, (size_t)t.n_pvst * 4);
		memcpy(out->parent.p + at, t.parent, (size_t)t.n_pvst * 4);
		memcpy(out->a_or.p + at, t.a_or, t.n_pvst);
		memcpy(out->z_or.p + at, t.z_or, t.n_pvst);
		povu_hip_forest::Tree nt = f->trees[i];
		nt.blk = -1;
		nt.off = at;
		out->trees.push_back(nt);
		at += t.n_pvst;
	}
	return out;
}

static void compact_in_place(povu_hip_forest &f)
{
	f.ready();
	std::unique_ptr<povu_hip_forest> c = compact_forest(&f);
	f.release_block();
	for (auto &b : f.extra)
		if (b.p && b.pool)
			b.pool->put(b.p, b.cap, b.seg);
	f.extra.clear();
	f.block = c->block, f.block_cap = c->block_cap, f.block_bytes = c->block_bytes, f.total_entries = c->total_entries;
	f.block_seg = c->block_seg;
	f.a_id = c->a_id, f.z_id = c->z_id, f.parent = c->parent, f.a_or = c->a_or, f.z_or = c->z_or;
	f.trees = std::move(c->trees);
	c->block = nullptr;
	c->block_cap = c->block_bytes = c->total_entries = 0;
}

extern "C" povu_hip_forest *povu_hip_comm_gather(povu_hip_comm *c, const povu_hip_forest *mine, char *err, size_t errlen)
{
	std::unique_ptr<povu_hip_forest> compacted;
	bool in_group = false;
	try {
		if (!c || !c->ctx)
			throw HipError("gather: bad arguments");
		povu_hip_ctx *ctx = c->ctx;
		HIP_CHECK(hipSetDevice(ctx->device));
		Rccl &R = rccl();
		hipStream_t s = c->side; // (see povu_hip_comm_scatter: one communicator, one stream)
		const double t0 = now_ms();
		const bool root = c->rank == 0;
		// ---- everything that can go wrong on this rank alone is found out BEFORE the first collective, and travels in the
		// status word of the size exchange: all ranks then fail together instead of waiting for each other
		std::string my_error;
		size_t nt = 0, total = 0;
		uint32_t *dmeta = nullptr, *hmeta = nullptr;
		char *dblk = nullptr;
		size_t mb = 0, bb = 0;
		try {
			if (!mine)
				throw HipError("gather: bad arguments");
			const_cast<povu_hip_forest *>(mine)->ready();
			if (!mine->hairpins.empty())
				throw HipError("gather: hairpin boundaries do not travel");
			if (!mine->sub_fam.empty())
				throw HipError("gather: subflubble labels do not travel");
			if (!mine->extra.empty()) { // (rare: some components went through the sequential redo)
				compacted = compact_forest(mine);
				mine = compacted.get();
			}
			nt = mine->trees.size(), total = mine->total_entries;
			if (!root && nt) { // tree table + block go through device staging, one message each
				mb = nt * 32, bb = povu_hip_forest::ExtraBlock::bytes_for(total); // 8 words per tree
				c->stage.reserve(mb + bb + 512);
				dmeta = c->stage.take<uint32_t>(mb / 4);
				dblk = c->stage.take<char>(bb);
				hmeta = ctx->host.take<uint32_t>(8 * nt);
				for (size_t i = 0; i < nt; i++) {
					const auto &t = mine->trees[i];
					if (t.blk >= 0)
						throw HipError("gather: unexpected merged forest");
					if (t.off + t.n_pvst > total)
						throw HipError("gather: tree outside its block");
					uint32_t *q = hmeta + 8 * i;
					q[0] = t.component_id, q[1] = t.n_vtx, q[2] = t.n_links, q[3] = t.n_pvst, q[4] = (uint32_t)t.off, q[5] = q[6] = q[7] = 0;
				}
			}
		} catch (const std::exception &e) {
			my_error = e.what();
			nt = total = 0;
		}
		// what this rank contributes: [n_trees, total entries, total components, status] + tree table + its result block
		uint64_t *h = c->hsmall;
		h[0] = nt, h[1] = nt ? total : 0, h[2] = mine ? mine->total_components : 0, h[3] = my_error.empty() ? 0 : 1;
		uint64_t *dmy = c->dsmall, *dall = c->dsmall + 4;
		HIP_CHECK(copy_async(dmy, h, 32, hipMemcpyHostToDevice, s));
		NCCL_CHECK(R.AllGather(dmy, dall, 4, ncclUint64, c->comm, s));
		HIP_CHECK(copy_async(h + 4, dall, 32 * (size_t)c->world, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		const uint64_t *all = h + 4;
		if (!my_error.empty())
			throw HipError(my_error);
		for (uint32_t r = 0; r < c->world; r++)
			if (all[4 * r + 3])
				throw HipError("gather: rank " + std::to_string(r) + " could not contribute its forest");
		// the root sizes its staging with everybody's numbers and says whether it can take them (second, one-word handshake)
		struct In {
			uint32_t rank;
			size_t nt, total, mb, bb;
			uint32_t *dmeta;
			char *dblk;
		};
		std::vector<In> in;
		uint64_t go = 1;
		if (root) {
			try {
				size_t need = 1024;
				for (uint32_t r = 1; r < c->world; r++)
					if (all[4 * r])
						need += pad256(all[4 * r] * 32) + pad256(povu_hip_forest::ExtraBlock::bytes_for(all[4 * r + 1])) + 512;
				c->stage.reserve(need);
				for (uint32_t r = 1; r < c->world; r++) {
					if (!all[4 * r])
						continue;
					In x{r, (size_t)all[4 * r], (size_t)all[4 * r + 1], 0, 0, nullptr, nullptr};
					x.mb = x.nt * 32;
					x.bb = povu_hip_forest::ExtraBlock::bytes_for(x.total);
					x.dmeta = c->stage.take<uint32_t>(x.mb / 4);
					x.dblk = c->stage.take<char>(x.bb);
					in.push_back(x);
				}
			} catch (const std::exception &) {
				go = 0;
			}
		}
		uint64_t *hgo = c->hsmall + 4 + 4 * (size_t)c->world, *dgo = c->dsmall + 4 + 4 * (size_t)c->world;
		*hgo = go;
		HIP_CHECK(copy_async(dgo, hgo, 8, hipMemcpyHostToDevice, s));
		NCCL_CHECK(R.Broadcast(dgo, dgo, 1, ncclUint64, 0, c->comm, s));
		HIP_CHECK(copy_async(hgo, dgo, 8, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		if (!*hgo)
			throw HipError("gather: the root has no room for the ranks' forests");
		auto out = std::make_unique<povu_hip_forest>();
		out->pool = ctx->pool;
		if (!root) {
			if (nt) {
				HIP_CHECK(copy_async(dmeta, hmeta, nt * 32, hipMemcpyHostToDevice, s));
				HIP_CHECK(copy_async(dblk, mine->block, bb, hipMemcpyHostToDevice, s));
				NCCL_CHECK(R.GroupStart());
				in_group = true;
				NCCL_CHECK(R.Send(dmeta, mb, ncclChar, 0, c->comm, s));
				NCCL_CHECK(R.Send(dblk, bb, ncclChar, 0, c->comm, s));
				in_group = false;
				NCCL_CHECK(R.GroupEnd());
				HIP_CHECK(hipStreamSynchronize(s));
			}
			c->ms[1] = now_ms() - t0;
			return out.release(); // empty
		}
		// root: receive every rank's table + block into device staging, land the blocks in page-locked memory
		NCCL_CHECK(R.GroupStart());
		in_group = true;
		for (auto &x : in) {
			NCCL_CHECK(R.Recv(x.dmeta, x.mb, ncclChar, (int)x.rank, c->comm, s));
			NCCL_CHECK(R.Recv(x.dblk, x.bb, ncclChar, (int)x.rank, c->comm, s));
		}
		in_group = false;
		NCCL_CHECK(R.GroupEnd());
		uint32_t tc = mine->total_components;
		std::vector<uint32_t *> hmetas;
		for (auto &x : in) {
			povu_hip_forest::ExtraBlock blk;
			blk.pool = ctx->pool;
			blk.p = ctx->pool->get(x.bb, blk.cap);
			blk.carve(x.total);
			out->extra.push_back(std::move(blk));
			uint32_t *hm = ctx->host.take<uint32_t>(8 * x.nt);
			hmetas.push_back(hm);
			HIP_CHECK(copy_async(hm, x.dmeta, x.nt * 32, hipMemcpyDeviceToHost, s));
			HIP_CHECK(copy_async(blk.p, x.dblk, x.bb, hipMemcpyDeviceToHost, s)); // same layout as the sender's block
			tc = std::max<uint32_t>(tc, (uint32_t)all[4 * x.rank + 2]);
		}
		// the root's own trees stay where they are: the merged forest takes over the block of `mine`
		povu_hip_forest *m = const_cast<povu_hip_forest *>(mine);
		if (m->block) {
			povu_hip_forest::ExtraBlock own;
			own.pool = m->pool;
			own.p = m->block;
			own.cap = m->block_cap;
			own.seg = m->block_seg;
			own.carve(m->total_entries);
			const int bi = (int)out->extra.size();
			out->extra.push_back(std::move(own));
			for (auto t : m->trees) {
				t.blk = bi;
				out->trees.push_back(t);
			}
			m->block = nullptr;
			m->block_cap = m->block_bytes = m->total_entries = 0;
			m->block_seg = -1;
			m->trees.clear();
		}
		HIP_CHECK(hipStreamSynchronize(s));
		for (size_t k = 0; k < in.size(); k++) {
			for (size_t i = 0; i < in[k].nt; i++) {
				povu_hip_forest::Tree t{};
				const uint32_t *q = hmetas[k] + 8 * i;
				t.component_id = q[0], t.n_vtx = q[1], t.n_links = q[2], t.n_pvst = q[3];
				t.off = q[4];
				t.blk = (int)k;
				if (t.off + t.n_pvst > in[k].total)
					throw HipError("gather: a rank's tree lies outside its block");
				out->trees.push_back(t);
			}
		}
		out->total_components = tc;
		sort_trees(*out);
		c->ms[1] = now_ms() - t0;
		return out.release();
	} catch (const std::exception &e) {
		if (in_group) {
			try {
				(void)rccl().GroupEnd();
			} catch (...) {
			}
		}
		if (c && c->ctx) {
			(void)hipStreamSynchronize(c->ctx->stream);
			if (c->side)
				(void)hipStreamSynchronize(c->side);
		}
		set_err(err, errlen, e.what());
		return nullptr;
	}
}

extern "C" int povu_hip_comm_times(const povu_hip_comm *c, double out_ms[2])
{
	if (!c || !out_ms)
		return 1;
	out_ms[0] = c->ms[0], out_ms[1] = c->ms[1];
	return 0;
}
