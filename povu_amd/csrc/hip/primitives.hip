// primitives.hip -- device-wide primitives between the decompose kernels: exclusive scans and the stable radix sort
// of (key, value) pairs, all hand-written.
#include "common.hpp"

namespace povu_hip
{

// ---- exclusive scans (sum / max) of u32, two launches: per-chunk partials, then every block adds the
// partials in front of it and scans its own chunk tile by tile (8 elements per lane, wave shuffles,
// one LDS exchange per tile).  The arrays here are 1-30 M elements; the second read of the input comes
// from L2 / Infinity Cache.
static constexpr int SC_TPB = 256, SC_ITEMS = 8, SC_TILE = SC_TPB * SC_ITEMS, SC_MAX_BLOCKS = 1024;

// MAX: 0 = sum (mod 2^32), 1 = maximum, 2 = xor; the identity of all three is 0
template <int MAX>
__device__ __forceinline__ uint32_t sc_op(uint32_t a, uint32_t b)
{
	return MAX == 1 ? (a > b ? a : b) : (MAX == 2 ? (a ^ b) : a + b);
}
template <int MAX>
__device__ __forceinline__ uint32_t sc_block_reduce(uint32_t v, uint32_t *sh)
{
	for (int off = 32; off; off >>= 1)
		v = sc_op<MAX>(v, __shfl_down(v, off));
	if ((threadIdx.x & 63) == 0)
		sh[threadIdx.x >> 6] = v;
	__syncthreads();
	uint32_t r = sc_op<MAX>(sc_op<MAX>(sh[0], sh[1]), sc_op<MAX>(sh[2], sh[3]));
	__syncthreads();
	return r;
}
// One scan job; a launch carries one or two of them (blockIdx.y picks) so that independent scans that
// are due at the same point of the pass share their two launches.
struct ScanJob {
	const uint32_t *in;
	const uint32_t *sub; // when set (sum scans of words): the element is in[i] - sub[i]
	const uint8_t *in8; // when set: the input is one BYTE per element (flags, small counts) -- a quarter of the reads
	uint32_t *out;
	size_t n, chunk;
	uint32_t blocks;
	uint32_t *partial;
	// single-launch form (k_scan_lookback): tiles of LB_TILE elements, one status word a tile, handed out by a ticket
	uint32_t tiles;
	unsigned long long *status;
	uint32_t *ticket;
};
struct ScanJobs {
	ScanJob j[2];
};
template <int MAX>
__global__ void __launch_bounds__(SC_TPB) k_scan_partials(const ScanJobs jobs)
{
	__shared__ uint32_t sh[4];
	const ScanJob &J = jobs.j[blockIdx.y];
	if (blockIdx.x >= J.blocks)
		return;
	const uint32_t *__restrict__ in = J.in;
	uint32_t *__restrict__ partial = J.partial;
	const size_t n = J.n, chunk = J.chunk;
	const size_t b0 = (size_t)blockIdx.x * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
	uint32_t acc = 0;
	// chunks start on multiples of the tile and every operand is 16-byte aligned: sixteen bytes per load (four words, or
	// sixteen one-byte elements), the tail of the last chunk element by element
	auto fold4 = [&](const uint4 &a) { acc = sc_op<MAX>(acc, sc_op<MAX>(sc_op<MAX>(a.x, a.y), sc_op<MAX>(a.z, a.w))); };
	if (J.in8) {
		const uint8_t *__restrict__ in8 = J.in8;
		const size_t w1 = b0 + ((b1 - b0) & ~size_t(15));
		auto bytes4 = [](uint32_t w) {
			return sc_op<MAX>(sc_op<MAX>(w & 0xFFu, (w >> 8) & 0xFFu), sc_op<MAX>((w >> 16) & 0xFFu, w >> 24));
		};
		for (size_t i = b0 + 16 * (size_t)threadIdx.x; i < w1; i += 16 * SC_TPB) {
			const uint4 a = *reinterpret_cast<const uint4 *>(in8 + i);
			fold4(make_uint4(bytes4(a.x), bytes4(a.y), bytes4(a.z), bytes4(a.w)));
		}
		for (size_t i = w1 + threadIdx.x; i < b1; i += SC_TPB)
			acc = sc_op<MAX>(acc, in8[i]);
	} else {
		const uint32_t *__restrict__ sub = J.sub;
		const size_t w1 = b0 + ((b1 - b0) & ~size_t(3));
		if (sub) {
			for (size_t i = b0 + 4 * (size_t)threadIdx.x; i < w1; i += 4 * SC_TPB) {
				const uint4 a = *reinterpret_cast<const uint4 *>(in + i), c = *reinterpret_cast<const uint4 *>(sub + i);
				fold4(make_uint4(a.x - c.x, a.y - c.y, a.z - c.z, a.w - c.w));
			}
		} else {
			for (size_t i = b0 + 4 * (size_t)threadIdx.x; i < w1; i += 4 * SC_TPB)
				fold4(*reinterpret_cast<const uint4 *>(in + i));
		}
		for (size_t i = w1 + threadIdx.x; i < b1; i += SC_TPB)
			acc = sc_op<MAX>(acc, in[i] - (sub ? sub[i] : 0u));
	}
	acc = sc_block_reduce<MAX>(acc, sh);
	if (threadIdx.x == 0)
		partial[blockIdx.x] = acc;
}
// eight consecutive elements of a job starting at e0 (zeros behind b1)
__device__ __forceinline__ void sc_load_tile(const ScanJob &J, size_t e0, size_t b1, uint32_t (&v)[SC_ITEMS])
{
	if (J.in8) {
		const uint8_t *__restrict__ in8 = J.in8;
		if (e0 + SC_ITEMS <= b1) {
			const uint2 a = *reinterpret_cast<const uint2 *>(in8 + e0);
			v[0] = a.x & 0xFFu, v[1] = (a.x >> 8) & 0xFFu, v[2] = (a.x >> 16) & 0xFFu, v[3] = a.x >> 24;
			v[4] = a.y & 0xFFu, v[5] = (a.y >> 8) & 0xFFu, v[6] = (a.y >> 16) & 0xFFu, v[7] = a.y >> 24;
		} else {
			for (int k = 0; k < SC_ITEMS; k++)
				v[k] = e0 + k < b1 ? in8[e0 + k] : 0u;
		}
	} else if (e0 + SC_ITEMS <= b1) {
		const uint32_t *__restrict__ in = J.in;
		const uint4 a = *reinterpret_cast<const uint4 *>(in + e0), b = *reinterpret_cast<const uint4 *>(in + e0 + 4);
		v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
		if (J.sub) {
			const uint4 c = *reinterpret_cast<const uint4 *>(J.sub + e0), d = *reinterpret_cast<const uint4 *>(J.sub + e0 + 4);
			v[0] -= c.x, v[1] -= c.y, v[2] -= c.z, v[3] -= c.w, v[4] -= d.x, v[5] -= d.y, v[6] -= d.z, v[7] -= d.w;
		}
	} else {
		for (int k = 0; k < SC_ITEMS; k++)
			v[k] = e0 + k < b1 ? J.in[e0 + k] - (J.sub ? J.sub[e0 + k] : 0u) : 0u;
	}
}
template <int MAX>
__global__ void __launch_bounds__(SC_TPB) k_scan_chunks(const ScanJobs jobs)
{
	__shared__ uint32_t sh[4];
	__shared__ uint32_t wave_tot[4];
	const ScanJob &J = jobs.j[blockIdx.y];
	if (blockIdx.x >= J.blocks)
		return;
	uint32_t *__restrict__ out = J.out;
	const uint32_t *__restrict__ partial = J.partial;
	const size_t n = J.n, chunk = J.chunk;
	uint32_t base = 0;
	for (uint32_t k = threadIdx.x; k < blockIdx.x; k += SC_TPB)
		base = sc_op<MAX>(base, partial[k]);
	uint32_t carry = sc_block_reduce<MAX>(base, sh);
	const size_t b0 = (size_t)blockIdx.x * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	// the next tile's loads are issued before this tile's exchange through LDS: two tiles of a block are in flight
	uint32_t v[SC_ITEMS], nx[SC_ITEMS];
	if (b0 < b1)
		sc_load_tile(J, b0 + (size_t)threadIdx.x * SC_ITEMS, b1, v);
	for (size_t t0 = b0; t0 < b1; t0 += SC_TILE) {
		const size_t e0 = t0 + (size_t)threadIdx.x * SC_ITEMS;
		const bool more = t0 + SC_TILE < b1;
		if (more)
			sc_load_tile(J, e0 + SC_TILE, b1, nx);
		uint32_t tot = 0; // lane-local exclusive scan
		for (int k = 0; k < SC_ITEMS; k++) {
			const uint32_t x = v[k];
			v[k] = tot;
			tot = sc_op<MAX>(tot, x);
		}
		uint32_t inc = tot; // inclusive scan of lane totals across the wave
		for (int off = 1; off < 64; off <<= 1) {
			const uint32_t y = __shfl_up(inc, off);
			if (lane >= off)
				inc = sc_op<MAX>(inc, y);
		}
		if (lane == 63)
			wave_tot[wave] = inc;
		__syncthreads();
		uint32_t pre = carry; // everything in front of this lane: carry, earlier waves, earlier lanes
		for (int w = 0; w < wave; w++)
			pre = sc_op<MAX>(pre, wave_tot[w]);
		const uint32_t lane_excl = __shfl_up(inc, 1);
		if (lane > 0)
			pre = sc_op<MAX>(pre, lane_excl);
		const uint32_t tile_tot = sc_op<MAX>(sc_op<MAX>(wave_tot[0], wave_tot[1]), sc_op<MAX>(wave_tot[2], wave_tot[3]));
		if (e0 + SC_ITEMS <= b1) {
			uint4 a, b;
			a.x = sc_op<MAX>(pre, v[0]), a.y = sc_op<MAX>(pre, v[1]), a.z = sc_op<MAX>(pre, v[2]), a.w = sc_op<MAX>(pre, v[3]);
			b.x = sc_op<MAX>(pre, v[4]), b.y = sc_op<MAX>(pre, v[5]), b.z = sc_op<MAX>(pre, v[6]), b.w = sc_op<MAX>(pre, v[7]);
			*reinterpret_cast<uint4 *>(out + e0) = a;
			*reinterpret_cast<uint4 *>(out + e0 + 4) = b;
		} else {
			for (int k = 0; k < SC_ITEMS; k++)
				if (e0 + k < b1)
					out[e0 + k] = sc_op<MAX>(pre, v[k]);
		}
		carry = sc_op<MAX>(carry, tile_tot);
		__syncthreads();
		if (more)
			for (int k = 0; k < SC_ITEMS; k++)
				v[k] = nx[k];
	}
}

// ---- the same scans in ONE launch and one read of the input (decoupled look-back).  A workgroup takes a ticket = its
// tile (LB_TILE elements, all loaded up front: 32 a lane), publishes the tile's aggregate, then its first wave looks back
// over the status words of the tiles in front of it, 64 at a time -- aggregates are added up until a tile is met that
// already knows its inclusive prefix --, publishes its own inclusive prefix and the workgroup writes the tile out.  A tile
// only ever waits for tiles with smaller tickets, i.e. for workgroups that are already running: no assumption about the
// order workgroups are dispatched in.  Value and state of a tile share one 64-bit word (read and written with
// agent-scope atomics: the L2 of another XCD never serves a stale one), so no fence orders anything.  The two-launch form
// above reads its input twice (4.3 GB of the 170 GB a whole-genome pass moved in round 5, and 1.05 ms for the partials).
static constexpr unsigned long long LB_AGG = 1ull << 32, LB_PREFIX = 2ull << 32;
template <int MAX, int LB_SUB, bool TICKET>
__global__ void __launch_bounds__(SC_TPB) k_scan_lookback(const ScanJobs jobs)
{
	constexpr int LB_TILE = SC_TILE * LB_SUB;
	__shared__ uint32_t wave_tot[LB_SUB][4];
	__shared__ uint32_t s_tile, s_carry;
	const ScanJob &J = jobs.j[blockIdx.y];
	if (blockIdx.x >= J.tiles)
		return;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t t = blockIdx.x;
	if (TICKET && J.tiles > 1) { // (uniform: a one-tile job needs neither ticket nor status)
		if (threadIdx.x == 0)
			s_tile = atomicAdd(J.ticket, 1u);
		__syncthreads();
		t = s_tile;
	}
	const size_t b0 = (size_t)t * LB_TILE, b1 = b0 + LB_TILE < J.n ? b0 + LB_TILE : J.n;
	uint32_t v[LB_SUB][SC_ITEMS], inc[LB_SUB];
	if (b0 + LB_TILE <= J.n && !J.in8) { // a whole tile of words (uniform): every load of the lane is issued before the first is needed
		const uint32_t *__restrict__ in = J.in + b0 + (size_t)threadIdx.x * SC_ITEMS;
		uint4 a[LB_SUB][2];
#pragma unroll
		for (int q = 0; q < LB_SUB; q++)
			a[q][0] = *reinterpret_cast<const uint4 *>(in + q * SC_TILE), a[q][1] = *reinterpret_cast<const uint4 *>(in + q * SC_TILE + 4);
		if (J.sub) {
			const uint32_t *__restrict__ sub = J.sub + b0 + (size_t)threadIdx.x * SC_ITEMS;
#pragma unroll
			for (int q = 0; q < LB_SUB; q++) {
				const uint4 c = *reinterpret_cast<const uint4 *>(sub + q * SC_TILE), d = *reinterpret_cast<const uint4 *>(sub + q * SC_TILE + 4);
				a[q][0].x -= c.x, a[q][0].y -= c.y, a[q][0].z -= c.z, a[q][0].w -= c.w;
				a[q][1].x -= d.x, a[q][1].y -= d.y, a[q][1].z -= d.z, a[q][1].w -= d.w;
			}
		}
#pragma unroll
		for (int q = 0; q < LB_SUB; q++) {
			v[q][0] = a[q][0].x, v[q][1] = a[q][0].y, v[q][2] = a[q][0].z, v[q][3] = a[q][0].w;
			v[q][4] = a[q][1].x, v[q][5] = a[q][1].y, v[q][6] = a[q][1].z, v[q][7] = a[q][1].w;
		}
	} else if (b0 + LB_TILE <= J.n) { // a whole tile of bytes
		const uint8_t *__restrict__ in8 = J.in8 + b0 + (size_t)threadIdx.x * SC_ITEMS;
		uint2 a[LB_SUB];
#pragma unroll
		for (int q = 0; q < LB_SUB; q++)
			a[q] = *reinterpret_cast<const uint2 *>(in8 + q * SC_TILE);
#pragma unroll
		for (int q = 0; q < LB_SUB; q++) {
			v[q][0] = a[q].x & 0xFFu, v[q][1] = (a[q].x >> 8) & 0xFFu, v[q][2] = (a[q].x >> 16) & 0xFFu, v[q][3] = a[q].x >> 24;
			v[q][4] = a[q].y & 0xFFu, v[q][5] = (a[q].y >> 8) & 0xFFu, v[q][6] = (a[q].y >> 16) & 0xFFu, v[q][7] = a[q].y >> 24;
		}
	} else {
#pragma unroll
		for (int q = 0; q < LB_SUB; q++)
			sc_load_tile(J, b0 + (size_t)q * SC_TILE + (size_t)threadIdx.x * SC_ITEMS, b1, v[q]);
	}
#pragma unroll
	for (int q = 0; q < LB_SUB; q++) {
		uint32_t tot = 0; // lane-local exclusive scan
#pragma unroll
		for (int k = 0; k < SC_ITEMS; k++) {
			const uint32_t x = v[q][k];
			v[q][k] = tot;
			tot = sc_op<MAX>(tot, x);
		}
		uint32_t i = tot; // inclusive scan of the lane totals across the wave
		for (int off = 1; off < 64; off <<= 1) {
			const uint32_t y = __shfl_up(i, off);
			if (lane >= off)
				i = sc_op<MAX>(i, y);
		}
		inc[q] = i;
		if (lane == 63)
			wave_tot[q][wave] = i;
	}
	__syncthreads();
	if (wave == 0) {
		uint32_t agg = 0;
#pragma unroll
		for (int q = 0; q < LB_SUB; q++)
#pragma unroll
			for (int w = 0; w < 4; w++)
				agg = sc_op<MAX>(agg, wave_tot[q][w]);
		uint32_t carry = 0;
		if (t > 0) {
			if (lane == 0)
				__hip_atomic_store(J.status + t, LB_AGG | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			for (int64_t base = (int64_t)t - 1;;) {
				const int64_t idx = base - lane;
				// (in front of tile 0: an inclusive prefix of nothing)
				const unsigned long long w = idx >= 0 ? __hip_atomic_load(J.status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : LB_PREFIX;
				const unsigned long long has_p = __ballot((w >> 32) == 2ull), unset = __ballot((w >> 32) == 0ull);
				const int stop = has_p ? __ffsll((long long)has_p) - 1 : 64; // nearest tile that knows its prefix
				const unsigned long long need = stop >= 63 ? ~0ull : ((2ull << stop) - 1ull);
				if (unset & need)
					continue; // some tile up to there has not published yet: its workgroup is running, read again
				uint32_t x = lane <= stop ? (uint32_t)w : 0u;
				for (int off = 32; off; off >>= 1)
					x = sc_op<MAX>(x, __shfl_down(x, off));
				carry = sc_op<MAX>(carry, __shfl(x, 0));
				if (has_p)
					break;
				base -= 64;
			}
		}
		if (lane == 0) {
			if (J.tiles > 1)
				__hip_atomic_store(J.status + t, LB_PREFIX | sc_op<MAX>(carry, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			s_carry = carry;
		}
	}
	__syncthreads();
	uint32_t pre_sub = s_carry; // everything in front of the sub-tile
#pragma unroll
	for (int q = 0; q < LB_SUB; q++) {
		const size_t e0 = b0 + (size_t)q * SC_TILE + (size_t)threadIdx.x * SC_ITEMS;
		uint32_t pre = pre_sub; // ... in front of this lane: earlier waves, earlier lanes
		for (int w = 0; w < wave; w++)
			pre = sc_op<MAX>(pre, wave_tot[q][w]);
		const uint32_t lane_excl = __shfl_up(inc[q], 1);
		if (lane > 0)
			pre = sc_op<MAX>(pre, lane_excl);
		if (e0 + SC_ITEMS <= b1) { // (always, in a whole tile)
			uint4 a, b;
			a.x = sc_op<MAX>(pre, v[q][0]), a.y = sc_op<MAX>(pre, v[q][1]), a.z = sc_op<MAX>(pre, v[q][2]), a.w = sc_op<MAX>(pre, v[q][3]);
			b.x = sc_op<MAX>(pre, v[q][4]), b.y = sc_op<MAX>(pre, v[q][5]), b.z = sc_op<MAX>(pre, v[q][6]), b.w = sc_op<MAX>(pre, v[q][7]);
			*reinterpret_cast<uint4 *>(J.out + e0) = a;
			*reinterpret_cast<uint4 *>(J.out + e0 + 4) = b;
		} else {
			for (int k = 0; k < SC_ITEMS; k++)
				if (e0 + k < b1)
					J.out[e0 + k] = sc_op<MAX>(pre, v[q][k]);
		}
		pre_sub = sc_op<MAX>(pre_sub, sc_op<MAX>(sc_op<MAX>(wave_tot[q][0], wave_tot[q][1]), sc_op<MAX>(wave_tot[q][2], wave_tot[q][3])));
	}
}

static constexpr size_t SC_LEGACY_BYTES = SC_MAX_BLOCKS * 16 + 256; // (two u32 jobs of the two-launch form, or one job of 128-bit words)
// variants (POVU_HIP_LB_VARIANT, measured with tools/scan_time.py on 10^8 words: two launches 0.268 ms; 0: 0.271; 1: 0.249;
// 2: 0.242; 3: 0.231): bit 1 = 64 elements a lane instead of 32, bit 0 = the tile is the workgroup's index instead of a
// ticket (relies on workgroups being dispatched in index order: not the default).  Tried and dropped in round 5: a tile per
// WAVE (2048 elements, no barrier behind the ticket: 0.248), and as many workgroups as are resident at once, each striding
// over the tiles without a ticket (0.251) -- a streaming read + write of 8 bytes an element stays at ~3.4 TB/s either way.
static int lb_variant()
{
	static const int v = getenv("POVU_HIP_LB_VARIANT") ? atoi(getenv("POVU_HIP_LB_VARIANT")) : 2;
	return v;
}
// below this many elements the two-launch form is the faster one: its second read comes out of the Infinity Cache
// (10^7 words: 0.023 ms against 0.035) -- except when every job fits one tile (one launch instead of two, no fill)
static constexpr size_t LB_MIN = 48u << 20;
static size_t lb_tile() { return (size_t)SC_TILE * ((lb_variant() & 2) ? 8 : 4); }
static size_t lb_tiles(size_t n) { return (n + lb_tile() - 1) / lb_tile(); }
static size_t lb_job_bytes(size_t n) { return 16 + 8 * lb_tiles(n); } // ticket (+ padding), status words
size_t scan_tmp_bytes(size_t n)
{
	return SC_LEGACY_BYTES + 2 * lb_job_bytes(n) + 64; // the partials of the two-launch form, then ticket + status words of two jobs
}
// one or two jobs in one launch; false: the temporary storage was sized for fewer elements (the caller takes two launches)
template <int MAX>
static bool scan_lookback(ScanJobs &jobs, int nj, void *tmp, size_t tmp_bytes, hipStream_t s)
{
	static const bool off = getenv("POVU_HIP_SCAN_TWO_LAUNCH") != nullptr; // (A/B hook)
	if (off)
		return false;
	size_t need = SC_LEGACY_BYTES, clear = 0;
	unsigned gx = 0;
	char *at = static_cast<char *>(tmp) + SC_LEGACY_BYTES;
	for (int j = 0; j < nj; j++) {
		ScanJob &J = jobs.j[j];
		const size_t tiles = lb_tiles(J.n);
		if (tiles > 0xFFFFFFF0ull)
			return false;
		J.tiles = (uint32_t)tiles;
		J.ticket = reinterpret_cast<uint32_t *>(at);
		J.status = reinterpret_cast<unsigned long long *>(at + 16);
		at += lb_job_bytes(J.n);
		need += lb_job_bytes(J.n);
		if (tiles > 1)
			clear = need - SC_LEGACY_BYTES; // (the jobs lie back to back: one fill covers every job that needs one)
		gx = std::max<unsigned>(gx, J.tiles);
	}
	if (need > tmp_bytes)
		return false;
	size_t n_max = 0;
	for (int j = 0; j < nj; j++)
		n_max = std::max(n_max, jobs.j[j].n);
	if (clear && n_max < LB_MIN)
		return false;
	if (clear)
		HIP_CHECK(hipMemsetAsync(static_cast<char *>(tmp) + SC_LEGACY_BYTES, 0, clear, s));
	switch (lb_variant()) {
	case 0: KLAUNCH((k_scan_lookback<MAX, 4, true>), dim3(gx, nj), dim3(SC_TPB), 0, s, jobs); break;
	case 1: KLAUNCH((k_scan_lookback<MAX, 4, false>), dim3(gx, nj), dim3(SC_TPB), 0, s, jobs); break;
	case 2: KLAUNCH((k_scan_lookback<MAX, 8, true>), dim3(gx, nj), dim3(SC_TPB), 0, s, jobs); break;
	default: KLAUNCH((k_scan_lookback<MAX, 8, false>), dim3(gx, nj), dim3(SC_TPB), 0, s, jobs); break;
	}
	return true;
}

static ScanJob make_scan_job(const uint32_t *in, uint32_t *out, size_t n, uint32_t *partial, const uint8_t *in8 = nullptr,
			     const uint32_t *sub = nullptr)
{
	if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(in8) | reinterpret_cast<uintptr_t>(out) |
	     reinterpret_cast<uintptr_t>(sub)) & 15)
		throw HipError("scan: operands must be 16-byte aligned");
	size_t blocks = (n + SC_TILE - 1) / SC_TILE;
	if (blocks > SC_MAX_BLOCKS)
		blocks = SC_MAX_BLOCKS;
	size_t chunk = (n + blocks - 1) / blocks;
	chunk = (chunk + SC_TILE - 1) / SC_TILE * SC_TILE; // whole tiles: every tile base stays 16-byte aligned
	blocks = (n + chunk - 1) / chunk;
	return ScanJob{in, sub, in8, out, n, chunk, (uint32_t)blocks, partial, 0u, nullptr, nullptr};
}

template <int MAX>
static void scan_exclusive(const uint32_t *in, uint32_t *out, size_t n, void *tmp, size_t tmp_bytes, hipStream_t s)
{
	if (n == 0)
		return;
	if (tmp_bytes < SC_MAX_BLOCKS * sizeof(uint32_t))
		throw HipError("scan: temporary storage too small");
	ScanJobs jobs{};
	jobs.j[0] = make_scan_job(in, out, n, static_cast<uint32_t *>(tmp));
	if (scan_lookback<MAX>(jobs, 1, tmp, tmp_bytes, s))
		return;
	KLAUNCH(k_scan_partials<MAX>, dim3(jobs.j[0].blocks, 1), dim3(SC_TPB), 0, s, jobs);
	KLAUNCH(k_scan_chunks<MAX>, dim3(jobs.j[0].blocks, 1), dim3(SC_TPB), 0, s, jobs);
}

// byte inputs: one or two independent jobs in the same two launches (in1 may be null)
void scan_exclusive_u8(const uint8_t *in0, uint32_t *out0, size_t n0, const uint8_t *in1, uint32_t *out1, size_t n1, void *tmp,
		       size_t tmp_bytes, hipStream_t s)
{
	if (tmp_bytes < 2 * SC_MAX_BLOCKS * sizeof(uint32_t))
		throw HipError("scan: temporary storage too small");
	ScanJobs jobs{};
	unsigned gx = 0, gy = 0;
	if (n0) {
		jobs.j[gy] = make_scan_job(nullptr, out0, n0, static_cast<uint32_t *>(tmp) + gy * SC_MAX_BLOCKS, in0);
		gx = std::max(gx, jobs.j[gy].blocks);
		gy++;
	}
	if (in1 && n1) {
		jobs.j[gy] = make_scan_job(nullptr, out1, n1, static_cast<uint32_t *>(tmp) + gy * SC_MAX_BLOCKS, in1);
		gx = std::max(gx, jobs.j[gy].blocks);
		gy++;
	}
	if (!gy)
		return;
	if (scan_lookback<0>(jobs, (int)gy, tmp, tmp_bytes, s))
		return;
	KLAUNCH(k_scan_partials<0>, dim3(gx, gy), dim3(SC_TPB), 0, s, jobs);
	KLAUNCH(k_scan_chunks<0>, dim3(gx, gy), dim3(SC_TPB), 0, s, jobs);
}

void scan_exclusive_u32(const uint32_t *in, uint32_t *out, size_t n, void *tmp, size_t tmp_bytes, hipStream_t s)
{
	scan_exclusive<0>(in, out, n, tmp, tmp_bytes, s);
}
void scan_exclusive_diff_u32(const uint32_t *in, const uint32_t *sub, uint32_t *out, size_t n, void *tmp, size_t tmp_bytes,
			     hipStream_t s)
{
	if (n == 0)
		return;
	if (tmp_bytes < SC_MAX_BLOCKS * sizeof(uint32_t))
		throw HipError("scan: temporary storage too small");
	ScanJobs jobs{};
	jobs.j[0] = make_scan_job(in, out, n, static_cast<uint32_t *>(tmp), nullptr, sub);
	if (scan_lookback<0>(jobs, 1, tmp, tmp_bytes, s))
		return;
	KLAUNCH(k_scan_partials<0>, dim3(jobs.j[0].blocks, 1), dim3(SC_TPB), 0, s, jobs);
	KLAUNCH(k_scan_chunks<0>, dim3(jobs.j[0].blocks, 1), dim3(SC_TPB), 0, s, jobs);
}

template <int OP>
static void scan_exclusive_pair(const uint32_t *in0, uint32_t *out0, size_t n0, const uint32_t *in1, uint32_t *out1, size_t n1,
				void *tmp, size_t tmp_bytes, hipStream_t s)
{
	if (n0 == 0 || n1 == 0) {
		scan_exclusive<OP>(in0, out0, n0, tmp, tmp_bytes, s);
		scan_exclusive<OP>(in1, out1, n1, tmp, tmp_bytes, s);
		return;
	}
	if (tmp_bytes < 2 * SC_MAX_BLOCKS * sizeof(uint32_t))
		throw HipError("scan: temporary storage too small");
	ScanJobs jobs{};
	jobs.j[0] = make_scan_job(in0, out0, n0, static_cast<uint32_t *>(tmp));
	jobs.j[1] = make_scan_job(in1, out1, n1, static_cast<uint32_t *>(tmp) + SC_MAX_BLOCKS);
	if (scan_lookback<OP>(jobs, 2, tmp, tmp_bytes, s))
		return;
	const unsigned gx = std::max(jobs.j[0].blocks, jobs.j[1].blocks);
	KLAUNCH(k_scan_partials<OP>, dim3(gx, 2), dim3(SC_TPB), 0, s, jobs);
	KLAUNCH(k_scan_chunks<OP>, dim3(gx, 2), dim3(SC_TPB), 0, s, jobs);
}

void scan_exclusive_u32_pair(const uint32_t *in0, uint32_t *out0, size_t n0, const uint32_t *in1, uint32_t *out1, size_t n1,
			     void *tmp, size_t tmp_bytes, hipStream_t s)
{
	scan_exclusive_pair<0>(in0, out0, n0, in1, out1, n1, tmp, tmp_bytes, s);
}

void scan_exclusive_xor_u32_pair(const uint32_t *in0, uint32_t *out0, const uint32_t *in1, uint32_t *out1, size_t n, void *tmp,
				 size_t tmp_bytes, hipStream_t s)
{
	scan_exclusive_pair<2>(in0, out0, n, in1, out1, n, tmp, tmp_bytes, s);
}

void scan_exclusive_max_u32(const uint32_t *in, uint32_t *out, size_t n, void *tmp, size_t tmp_bytes, hipStream_t s)
{
	scan_exclusive<1>(in, out, n, tmp, tmp_bytes, s);
}

// ---- exclusive running xor of 128-bit words (the bridge test's two hashes): same two-launch scheme, 2 words per lane
static constexpr int X128_ITEMS = 2, X128_TILE = SC_TPB * X128_ITEMS;
struct Xor128Job {
	const ulonglong2 *in;
	ulonglong2 *out;
	size_t n, chunk;
	uint32_t blocks;
	ulonglong2 *partial;
	const uint32_t *n_dev; // optional: only the first *n_dev + 1 words exist (n is then the most there can be)
};
__device__ __forceinline__ size_t x128_len(const Xor128Job &J)
{
	if (!J.n_dev)
		return J.n;
	const size_t m = (size_t)*J.n_dev + 1;
	return m < J.n ? m : J.n;
}
__device__ __forceinline__ ulonglong2 x128(const ulonglong2 a, const ulonglong2 b) { return make_ulonglong2(a.x ^ b.x, a.y ^ b.y); }
__device__ __forceinline__ ulonglong2 x128_shfl_down(const ulonglong2 v, int off) { return make_ulonglong2(__shfl_down(v.x, off), __shfl_down(v.y, off)); }
__device__ __forceinline__ ulonglong2 x128_shfl_up(const ulonglong2 v, int off) { return make_ulonglong2(__shfl_up(v.x, off), __shfl_up(v.y, off)); }
__device__ __forceinline__ ulonglong2 x128_block_reduce(ulonglong2 v, ulonglong2 *sh)
{
	for (int off = 32; off; off >>= 1)
		v = x128(v, x128_shfl_down(v, off));
	if ((threadIdx.x & 63) == 0)
		sh[threadIdx.x >> 6] = v;
	__syncthreads();
	const ulonglong2 r = x128(x128(sh[0], sh[1]), x128(sh[2], sh[3]));
	__syncthreads();
	return r;
}
__global__ void __launch_bounds__(SC_TPB) k_xor128_partials(const Xor128Job J)
{
	__shared__ ulonglong2 sh[4];
	const size_t n = x128_len(J);
	const size_t b0 = (size_t)blockIdx.x * J.chunk, b1 = b0 + J.chunk < n ? b0 + J.chunk : n;
	ulonglong2 acc = make_ulonglong2(0ull, 0ull);
	for (size_t i = b0 + threadIdx.x; i < b1; i += SC_TPB)
		acc = x128(acc, J.in[i]);
	acc = x128_block_reduce(acc, sh);
	if (threadIdx.x == 0)
		J.partial[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(SC_TPB) k_xor128_chunks(const Xor128Job J)
{
	__shared__ ulonglong2 sh[4], wave_tot[4];
	ulonglong2 base = make_ulonglong2(0ull, 0ull);
	for (uint32_t k = threadIdx.x; k < blockIdx.x; k += SC_TPB)
		base = x128(base, J.partial[k]);
	ulonglong2 carry = x128_block_reduce(base, sh);
	const size_t n = x128_len(J);
	const size_t b0 = (size_t)blockIdx.x * J.chunk, b1 = b0 + J.chunk < n ? b0 + J.chunk : n;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (size_t t0 = b0; t0 < b1; t0 += X128_TILE) {
		const size_t e0 = t0 + (size_t)threadIdx.x * X128_ITEMS;
		ulonglong2 v[X128_ITEMS];
		for (int k = 0; k < X128_ITEMS; k++)
			v[k] = e0 + k < b1 ? J.in[e0 + k] : make_ulonglong2(0ull, 0ull);
		ulonglong2 tot = make_ulonglong2(0ull, 0ull);
		for (int k = 0; k < X128_ITEMS; k++) {
			const ulonglong2 x = v[k];
			v[k] = tot;
			tot = x128(tot, x);
		}
		ulonglong2 inc = tot;
		for (int off = 1; off < 64; off <<= 1) {
			const ulonglong2 y = x128_shfl_up(inc, off);
			if (lane >= off)
				inc = x128(inc, y);
		}
		if (lane == 63)
			wave_tot[wave] = inc;
		__syncthreads();
		ulonglong2 pre = carry;
		for (int w = 0; w < wave; w++)
			pre = x128(pre, wave_tot[w]);
		const ulonglong2 lane_excl = x128_shfl_up(inc, 1);
		if (lane > 0)
			pre = x128(pre, lane_excl);
		const ulonglong2 tile_tot = x128(x128(wave_tot[0], wave_tot[1]), x128(wave_tot[2], wave_tot[3]));
		for (int k = 0; k < X128_ITEMS; k++)
			if (e0 + k < b1)
				J.out[e0 + k] = x128(pre, v[k]);
		carry = x128(carry, tile_tot);
		__syncthreads();
	}
}
void scan_exclusive_xor_u128(const ulonglong2 *in, ulonglong2 *out, size_t n, void *tmp, size_t tmp_bytes, hipStream_t s,
			     const uint32_t *n_dev)
{
	if (n == 0)
		return;
	if (tmp_bytes < SC_MAX_BLOCKS * sizeof(ulonglong2))
		throw HipError("scan: temporary storage too small");
	size_t blocks = (n + X128_TILE - 1) / X128_TILE;
	if (blocks > SC_MAX_BLOCKS)
		blocks = SC_MAX_BLOCKS;
	size_t chunk = (n + blocks - 1) / blocks;
	chunk = (chunk + X128_TILE - 1) / X128_TILE * X128_TILE;
	blocks = (n + chunk - 1) / chunk;
	const Xor128Job J{in, out, n, chunk, (uint32_t)blocks, static_cast<ulonglong2 *>(tmp), n_dev};
	KLAUNCH(k_xor128_partials, dim3((unsigned)blocks), dim3(SC_TPB), 0, s, J);
	KLAUNCH(k_xor128_chunks, dim3((unsigned)blocks), dim3(SC_TPB), 0, s, J);
}

// ---- indices of the set bytes of a flag array, in order (stream compaction without a prefix array): per-tile counts,
// one workgroup scans the counts, the tiles then rank their own flags again and write the indices.  The flags are read
// twice (a byte each); the 4-byte prefix array a scan would hand to a separate compaction kernel is never written.
static constexpr int CP_TPB = 256, CP_ITEMS = 16, CP_TILE = CP_TPB * CP_ITEMS;
__device__ __forceinline__ uint32_t cp_load16(const uint8_t *__restrict__ flag, size_t e0, size_t n, uint32_t &bits)
{
	bits = 0;
	if (e0 + CP_ITEMS <= n) {
		const uint4 a = *reinterpret_cast<const uint4 *>(flag + e0);
		const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
		for (int k = 0; k < 4; k++)
#pragma unroll
			for (int b = 0; b < 4; b++)
				if ((w[k] >> (8 * b)) & 0xFFu)
					bits |= 1u << (4 * k + b);
	} else {
		for (int k = 0; k < CP_ITEMS; k++)
			if (e0 + k < n && flag[e0 + k])
				bits |= 1u << k;
	}
	return (uint32_t)__popc(bits);
}
__device__ __forceinline__ uint32_t cp_block_exclusive(uint32_t v, uint32_t *sh, uint32_t &total)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t inc = v;
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t y = __shfl_up(inc, off);
		if (lane >= off)
			inc += y;
	}
	if (lane == 63)
		sh[wave] = inc;
	__syncthreads();
	uint32_t before = inc - v;
	total = 0;
	for (int w = 0; w < CP_TPB / 64; w++) {
		if (w < wave)
			before += sh[w];
		total += sh[w];
	}
	return before;
}
__global__ void __launch_bounds__(CP_TPB) k_cp_count(const uint8_t *__restrict__ flag, size_t n, uint32_t *__restrict__ tile_cnt)
{
	__shared__ uint32_t sh[CP_TPB / 64];
	uint32_t bits, total;
	const uint32_t bx = BIDX;
	const uint32_t c = cp_load16(flag, (size_t)bx * CP_TILE + (size_t)threadIdx.x * CP_ITEMS, n, bits);
	(void)cp_block_exclusive(c, sh, total);
	if (threadIdx.x == 0)
		tile_cnt[bx] = total;
}
// exclusive scan of the tile counts in place, one workgroup (a few ten thousand tiles); *count_out = the grand total
__global__ void __launch_bounds__(1024) k_cp_scan_tiles(uint32_t *tile_cnt, uint32_t ntiles, uint32_t *__restrict__ count_out)
{
	__shared__ uint32_t sh[16];
	const uint32_t per = (ntiles + 1023) / 1024, lo = min(ntiles, threadIdx.x * per), hi = min(ntiles, lo + per);
	uint32_t sum = 0;
	for (uint32_t k = lo; k < hi; k++)
		sum += tile_cnt[k];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t inc = sum;
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t y = __shfl_up(inc, off);
		if (lane >= off)
			inc += y;
	}
	if (lane == 63)
		sh[wave] = inc;
	__syncthreads();
	uint32_t before = inc - sum, total = 0;
	for (int w = 0; w < 16; w++) {
		if (w < wave)
			before += sh[w];
		total += sh[w];
	}
	for (uint32_t k = lo; k < hi; k++) {
		const uint32_t c = tile_cnt[k];
		tile_cnt[k] = before;
		before += c;
	}
	if (threadIdx.x == 0)
		*count_out = total;
}
__global__ void __launch_bounds__(CP_TPB) k_cp_write(const uint8_t *__restrict__ flag, size_t n, const uint32_t *__restrict__ tile_base,
						      uint32_t *__restrict__ out)
{
	__shared__ uint32_t sh[CP_TPB / 64];
	const uint32_t bx = BIDX;
	const size_t e0 = (size_t)bx * CP_TILE + (size_t)threadIdx.x * CP_ITEMS;
	uint32_t bits, total;
	const uint32_t c = cp_load16(flag, e0, n, bits);
	uint32_t at = tile_base[bx] + cp_block_exclusive(c, sh, total);
	while (bits) {
		const int k = __ffs((int)bits) - 1;
		bits &= bits - 1;
		out[at++] = (uint32_t)(e0 + k);
	}
}
__global__ void k_bitrank_counts(uint32_t W, uint4 *__restrict__ rec, uint32_t *__restrict__ cnt)
{
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w > W)
		return;
	if (w == W) { // the closing record: rank(n) reads it
		rec[W] = make_uint4(0u, 0u, 0u, 0u);
		cnt[W] = 0;
		return;
	}
	const uint4 r = rec[w];
	cnt[w] = (uint32_t)(__popc(r.x) + __popc(r.y));
}
__global__ void k_bitrank_ranks(uint32_t W, const uint32_t *__restrict__ ps, uint4 *__restrict__ rec)
{
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w <= W)
		rec[w].z = ps[w];
}
void bitrank_build(uint4 *rec, size_t n_rec, uint32_t *tmp_counts, void *scan_tmp, size_t scan_tmp_bytes, hipStream_t s)
{
	const uint32_t W = (uint32_t)n_rec;
	const unsigned blocks = (unsigned)((n_rec + 1 + 255) / 256);
	KLAUNCH(k_bitrank_counts, dim3(blocks), dim3(256), 0, s, W, rec, tmp_counts);
	scan_exclusive_u32(tmp_counts, tmp_counts, n_rec + 1, scan_tmp, scan_tmp_bytes, s);
	KLAUNCH(k_bitrank_ranks, dim3(blocks), dim3(256), 0, s, W, tmp_counts, rec);
}
__global__ void k_publish_words(WordSrc src, int n, uint32_t *__restrict__ dst)
{
	if (threadIdx.x < (unsigned)n)
		dst[threadIdx.x] = *src.p[threadIdx.x];
}
void publish_words(uint32_t *host_dst, const WordSrc &src, int n, hipStream_t s)
{
	if (n <= 0 || n > 8)
		throw HipError("publish_words: 1..8 words");
	uint32_t *dev = nullptr;
	HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), host_dst, 0));
	count_kernel_d2h((size_t)n * 4);
	KLAUNCH(k_publish_words, dim3(1), dim3(64), 0, s, src, n, dev);
}
size_t compact_tmp_bytes(size_t n) { return ((n + CP_TILE - 1) / CP_TILE + 1) * 4 + 256; }
void compact_flagged_u8(const uint8_t *flag, size_t n, uint32_t *out, uint32_t *count_dev, void *tmp, size_t tmp_bytes, hipStream_t s)
{
	if (n == 0) {
		HIP_CHECK(hipMemsetAsync(count_dev, 0, 4, s));
		return;
	}
	if (reinterpret_cast<uintptr_t>(flag) & 15)
		throw HipError("compaction: the flags must be 16-byte aligned");
	if (tmp_bytes < compact_tmp_bytes(n))
		throw HipError("compaction: temporary storage too small");
	const uint32_t ntiles = (uint32_t)((n + CP_TILE - 1) / CP_TILE);
	uint32_t *tile_cnt = static_cast<uint32_t *>(tmp);
	KLAUNCH(k_cp_count, dim3(ntiles), dim3(CP_TPB), 0, s, flag, n, tile_cnt);
	KLAUNCH(k_cp_scan_tiles, dim3(1), dim3(1024), 0, s, tile_cnt, ntiles, count_dev);
	KLAUNCH(k_cp_write, dim3(ntiles), dim3(CP_TPB), 0, s, flag, n, tile_cnt, out);
}

// ---- stable LSD radix sort of (key, value) pairs, hand-written.  Per place of RB bits: a histogram per tile of RS_TILE
// consecutive pairs, an exclusive scan over the [digit][tile] table (the scans above), and a scatter kernel in which
// every wave ranks its stretch of the tile with ballots (the lanes that hold the same digit find each other by RB
// ballots: no atomics, the order of equal keys is the order of the input), the tile is sorted into LDS, and consecutive
// lanes then write consecutive pairs of one digit to consecutive addresses.  Pangenome keys cluster (brackets of one
// bubble, slots of one side): runs of one digit inside a tile are long, and the table says where each run goes.
static constexpr int RS_TPB = 256, RS_WAVES = RS_TPB / 64, RS_ITEMS = 16, RS_TILE = RS_TPB * RS_ITEMS, RS_MAX_BINS = 1024;

__global__ void __launch_bounds__(RS_TPB) k_rs_hist(const uint32_t *__restrict__ keys, uint32_t n, unsigned shift, unsigned rb,
						     uint32_t *__restrict__ hist, uint32_t ntiles)
{
	__shared__ uint32_t h[RS_MAX_BINS];
	const uint32_t bins = 1u << rb, mask = bins - 1u;
	for (uint32_t d = threadIdx.x; d < bins; d += RS_TPB)
		h[d] = 0;
	__syncthreads();
	const uint32_t bx = BIDX, base = bx * RS_TILE;
	// four consecutive keys a lane and load (a histogram does not care which lane counts which key): a quarter of the
	// memory instructions
#pragma unroll
	for (int r = 0; r < RS_ITEMS / 4; r++) {
		const uint32_t i0 = base + (r * RS_TPB + threadIdx.x) * 4u;
		if (__all(i0 + 4 <= n)) {
			const uint4 k = *reinterpret_cast<const uint4 *>(keys + i0);
			const uint32_t d[4] = {(k.x >> shift) & mask, (k.y >> shift) & mask, (k.z >> shift) & mask, (k.w >> shift) & mask};
			// a whole wave on one digit (clustered keys) adds once
			const uint32_t d0 = __builtin_amdgcn_readfirstlane(d[0]);
			if (__all(d[0] == d0 && d[1] == d0 && d[2] == d0 && d[3] == d0)) {
				if ((threadIdx.x & 63) == 0)
					atomicAdd(&h[d0], 256u);
			} else {
#pragma unroll
				for (int q = 0; q < 4; q++)
					atomicAdd(&h[d[q]], 1u);
			}
		} else {
			for (uint32_t i = i0; i < i0 + 4 && i < n; i++)
				atomicAdd(&h[(keys[i] >> shift) & mask], 1u);
		}
	}
	__syncthreads();
	for (uint32_t d = threadIdx.x; d < bins; d += RS_TPB)
		hist[(size_t)d * ntiles + bx] = h[d];
}

__global__ void __launch_bounds__(RS_TPB) k_rs_scatter(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin,
							uint32_t *__restrict__ kout, uint32_t *__restrict__ vout, uint32_t n, unsigned shift,
							unsigned rb, const uint32_t *__restrict__ gofs, uint32_t ntiles)
{
	// the counters are dead once every pair knows its tile-local position: the staging area lies over them (36 KB of
	// LDS in all, four workgroups per CU)
	struct Counters {
		uint32_t wcnt[RS_WAVES][RS_MAX_BINS]; // per wave: pairs of a digit seen so far; later: pairs in the waves before
		uint32_t bbase[RS_MAX_BINS];	      // first tile-local position of a digit
	};
	union Overlay {
		Counters c;
		uint2 stage[RS_TILE];
	};
	__shared__ Overlay sh;
	__shared__ uint32_t gof[RS_MAX_BINS]; // where the digit's run of this tile starts in the output, less bbase
	__shared__ uint32_t wsum[RS_WAVES];
	const uint32_t bins = 1u << rb, mask = bins - 1u;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const unsigned long long lt = (1ull << lane) - 1ull;
	for (uint32_t k = 0; k < RS_WAVES; k++)
		for (uint32_t d = threadIdx.x; d < bins; d += RS_TPB)
			sh.c.wcnt[k][d] = 0;
	const uint32_t bx = BIDX, base = bx * RS_TILE, wbase = base + wave * (RS_ITEMS * 64);
	uint32_t key[RS_ITEMS], val[RS_ITEMS], pos[RS_ITEMS];
#pragma unroll
	for (int r = 0; r < RS_ITEMS; r++) {
		const uint32_t i = wbase + r * 64 + lane;
		key[r] = i < n ? kin[i] : 0xFFFFFFFFu;
		val[r] = i < n ? vin[i] : 0u;
	}
	__syncthreads();
	// ---- rank inside the wave's stretch
#pragma unroll
	for (int r = 0; r < RS_ITEMS; r++) {
		const bool live = wbase + r * 64 + lane < n;
		const uint32_t d = (key[r] >> shift) & mask;
		unsigned long long peers = __ballot(live);
		for (unsigned b = 0; b < rb; b++) {
			const unsigned long long m = __ballot((d >> b) & 1u);
			peers &= ((d >> b) & 1u) ? m : ~m;
		}
		uint32_t old = 0;
		if (live) {
			const int leader = __ffsll((long long)peers) - 1;
			if ((int)lane == leader) {
				old = sh.c.wcnt[wave][d];
				sh.c.wcnt[wave][d] = old + (uint32_t)__popcll(peers);
			}
			old = __shfl(old, leader);
		}
		pos[r] = old + (uint32_t)__popcll(peers & lt);
	}
	__syncthreads();
	// ---- per digit: the pairs of the waves before (exclusive over waves), the tile's count; then the tile-local bases
	uint32_t mine = 0; // pairs of the digits this thread owns: digits [t * per, (t + 1) * per)
	const uint32_t per = (bins + RS_TPB - 1) / RS_TPB;
	for (uint32_t k = 0; k < per; k++) {
		const uint32_t d = threadIdx.x * per + k;
		if (d < bins) {
			uint32_t run = 0;
			for (int w = 0; w < RS_WAVES; w++) {
				const uint32_t c = sh.c.wcnt[w][d];
				sh.c.wcnt[w][d] = run;
				run += c;
			}
			sh.c.bbase[d] = run; // (count for now)
			mine += run;
		}
	}
	uint32_t inc = mine;
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t y = __shfl_up(inc, off);
		if ((int)lane >= off)
			inc += y;
	}
	if (lane == 63)
		wsum[wave] = inc;
	__syncthreads();
	uint32_t before = inc - mine;
	for (uint32_t w = 0; w < wave; w++)
		before += wsum[w];
	for (uint32_t k = 0; k < per; k++) {
		const uint32_t d = threadIdx.x * per + k;
		if (d < bins) {
			const uint32_t c = sh.c.bbase[d];
			sh.c.bbase[d] = before;
			gof[d] = gofs[(size_t)d * ntiles + bx] - before;
			before += c;
		}
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < RS_ITEMS; r++) {
		const uint32_t d = (key[r] >> shift) & mask;
		pos[r] += sh.c.bbase[d] + sh.c.wcnt[wave][d];
	}
	__syncthreads(); // the counters are dead: the tile, sorted by digit, goes over them
#pragma unroll
	for (int r = 0; r < RS_ITEMS; r++)
		if (wbase + r * 64 + lane < n)
			sh.stage[pos[r]] = make_uint2(key[r], val[r]);
	__syncthreads();
	const uint32_t cnt = min((uint32_t)RS_TILE, n - base);
	for (uint32_t j = threadIdx.x; j < cnt; j += RS_TPB) {
		const uint2 e = sh.stage[j];
		const uint32_t gp = gof[(e.x >> shift) & mask] + j;
		kout[gp] = e.x;
		vout[gp] = e.y;
	}
}

static unsigned rs_places(size_t n, unsigned bits, unsigned &rb)
{
	// 9-bit digits (three places for the 25..27 key bits of a whole-genome graph); small inputs, whose table stays small,
	// take 10 and save a place where that helps
	const unsigned widest = n <= (size_t(1) << 24) ? 10u : 9u;
	const unsigned places = (bits + widest - 1) / widest;
	rb = (bits + places - 1) / places;
	return places;
}
static size_t rs_table_words(size_t n) { return (size_t)RS_MAX_BINS / (n <= (size_t(1) << 24) ? 1 : 2) * ((n + RS_TILE - 1) / RS_TILE) + 64; }

size_t sort_tmp_bytes(size_t n)
{
	// ping-pong buffer for keys and values + the [digit][tile] table (scanned in place) + the scan's own scratch
	return 2 * ((n * 4 + 255) & ~size_t(255)) + ((rs_table_words(n) * 4 + 255) & ~size_t(255)) + scan_tmp_bytes(rs_table_words(n)) + 1024;
}

void sort_pairs_u32(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned bits,
		    void *tmp, size_t tmp_bytes, hipStream_t s)
{
	if (n == 0)
		return;
	if (n >= (size_t(1) << 32) - RS_TILE)
		throw HipError("sort: too many pairs");
	if (tmp_bytes < sort_tmp_bytes(n))
		throw HipError("sort: temporary storage too small");
	if (bits == 0)
		bits = 1;
	char *q = static_cast<char *>(tmp);
	uint32_t *kt = reinterpret_cast<uint32_t *>(q);
	q += (n * 4 + 255) & ~size_t(255);
	uint32_t *vt = reinterpret_cast<uint32_t *>(q);
	q += (n * 4 + 255) & ~size_t(255);
	uint32_t *table = reinterpret_cast<uint32_t *>(q);
	q += (rs_table_words(n) * 4 + 255) & ~size_t(255);
	void *stmp = q;
	const size_t stmp_bytes = scan_tmp_bytes(rs_table_words(n));
	unsigned rb = 0;
	const unsigned places = rs_places(n, bits, rb);
	const uint32_t ntiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
	// the last place writes kout / vout: in -> [tmp ->] out for an even / odd number of places
	const uint32_t *ki = kin, *vi = vin;
	for (unsigned p = 0; p < places; p++) {
		const bool to_out = ((places - 1 - p) & 1u) == 0;
		uint32_t *ko = to_out ? kout : kt, *vo = to_out ? vout : vt;
		const unsigned shift = p * rb, w = std::min(rb, bits - std::min(bits, shift));
		const unsigned prb = w ? w : 1u;
		const size_t words = ((size_t)1 << prb) * ntiles;
		KLAUNCH(k_rs_hist, dim3(ntiles), dim3(RS_TPB), 0, s, ki, (uint32_t)n, shift, prb, table, ntiles);
		scan_exclusive_u32(table, table, words, stmp, stmp_bytes, s);
		KLAUNCH(k_rs_scatter, dim3(ntiles), dim3(RS_TPB), 0, s, ki, vi, ko, vo, (uint32_t)n, shift, prb, table, ntiles);
		ki = ko, vi = vo;
	}
}

} // namespace povu_hip
