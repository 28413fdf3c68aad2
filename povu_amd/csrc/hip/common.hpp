// common.hpp -- shared declarations of the gfx950 decompose library.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#define POVU_NIL 0xFFFFFFFFu

// four consecutive words from a 4-byte aligned address in ONE load instruction (global_load_dwordx4 only needs dword
// alignment); the caller makes sure the array carries three words of slack behind the last one it may name
__device__ __forceinline__ uint4 load4_unaligned(const unsigned *__restrict__ p)
{
	typedef unsigned v4a __attribute__((ext_vector_type(4), aligned(4)));
	const v4a v = *reinterpret_cast<const v4a *>(p);
	return make_uint4(v.x, v.y, v.z, v.w);
}

// Workgroup index with the 8 XCDs in mind: the dispatcher deals workgroups round-robin over the XCDs (blocks b and b + 8
// share one, MI355X_MICROARCH.md "Workgroup dispatch"), so neighbouring blocks -- which touch neighbouring lines in
// almost every kernel here -- land in eight different L2s.  BIDX renumbers the blocks so that every XCD works on one
// contiguous stretch of the grid (bijective for any grid size); a speed choice only, nothing depends on placement.
#ifdef POVU_XCD_SWIZZLE
__device__ __forceinline__ unsigned povu_xcd_bid()
{
	const unsigned nwg = gridDim.x, q = nwg >> 3, r = nwg & 7u, x = blockIdx.x & 7u, i = blockIdx.x >> 3;
	return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
#define BIDX povu_xcd_bid()
#else
#define BIDX blockIdx.x
#endif

namespace povu_hip
{

struct HipError : std::runtime_error {
	using std::runtime_error::runtime_error;
};

#define HIP_CHECK(expr)                                                                          \
	do {                                                                                     \
		hipError_t e__ = (expr);                                                         \
		if (e__ != hipSuccess)                                                           \
			throw povu_hip::HipError(std::string(#expr) + ": " + hipGetErrorString(e__) + " (" + \
						 __FILE__ + ":" + std::to_string(__LINE__) + ")");   \
	} while (0)

// kernel launch + launch-configuration check (a bad grid / LDS size / missing code object is not reported by the
// stream synchronisation that follows)
#define KLAUNCH(kernel, grid, block, shmem, stream, ...)                                          \
	do {                                                                                      \
		hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);              \
		HIP_CHECK(hipGetLastError());                                                     \
	} while (0)

// Bytes that cross PCIe, tallied per thread (a context is used by one thread at a time; the C ABI entry points add the
// difference over a call to their context's totals, povu_hip_transfer_bytes).  Every host<->device copy of the library goes
// through copy_async; results that kernels write straight into page-locked host memory are counted where they are sized.
struct XferTally {
	uint64_t h2d = 0, d2h = 0;
};
inline XferTally &xfer_tally()
{
	static thread_local XferTally t;
	return t;
}
inline hipError_t copy_async(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s)
{
	if (kind == hipMemcpyHostToDevice)
		xfer_tally().h2d += n;
	else if (kind == hipMemcpyDeviceToHost)
		xfer_tally().d2h += n;
	return hipMemcpyAsync(dst, src, n, kind, s);
}
inline void count_kernel_d2h(size_t n) { xfer_tally().d2h += n; }

// One growable device arena per context: stages carve typed spans with a bump
// pointer, nothing is hipMalloc'd inside the timed path once the arena is warm.
class Arena
{
public:
	~Arena() { release(); }
	void release()
	{
		if (base_)
			(void)hipFree(base_);
		base_ = nullptr;
		cap_ = 0;
		top_ = 0;
	}
	// make sure `bytes` are available from offset 0; invalidates earlier spans
	// (head_room: an eighth more than asked for, so that the next, slightly larger graph of a long-lived context does not
	// cost a new allocation; a caller that knows there will be no next graph asks for none)
	void reserve(size_t bytes, bool head_room = true)
	{
		if (bytes > cap_) {
			release();
			size_t want = head_room ? bytes + bytes / 8 + (1u << 20) : bytes + 4096;
			if (hipMalloc(&base_, want) != hipSuccess) {
				(void)hipGetLastError();
				base_ = nullptr;
				want = bytes + 4096; // no head room: retry with exactly what this graph needs
				if (hipMalloc(&base_, want) != hipSuccess) {
					(void)hipGetLastError();
					base_ = nullptr;
					size_t free_b = 0, total_b = 0;
					(void)hipMemGetInfo(&free_b, &total_b);
					throw HipError("not enough device memory for the decompose workspace: need " +
						       std::to_string(want >> 20) + " MiB in one arena, " + std::to_string(free_b >> 20) +
						       " MiB free of " + std::to_string(total_b >> 20) +
						       " MiB (povu_hip_workspace_estimate gives the total for a graph)");
				}
			}
			cap_ = want;
		}
		top_ = 0;
	}
	template <typename T>
	T *take(size_t n)
	{
		size_t off = (top_ + 255) & ~size_t(255);
		size_t bytes = n * sizeof(T);
		if (off + bytes > cap_)
			throw HipError("povu_hip arena overflow (internal sizing bug)");
		top_ = off + bytes;
		return reinterpret_cast<T *>(static_cast<char *>(base_) + off);
	}
	bool fits(size_t bytes) const { return ((top_ + 255) & ~size_t(255)) + bytes <= cap_; }
	static size_t padded(size_t n, size_t elem) { return ((n * elem) + 255) & ~size_t(255); }
	size_t capacity() const { return cap_; }
	size_t used() const { return top_; }
	// two groups of arrays that are never live together share a stretch: carve one, rewind to where it began, carve the
	// other, continue behind the longer of the two (rewind(m); ...; advance_to(end of the first))
	void rewind(size_t mark) { top_ = mark; }
	void advance_to(size_t mark)
	{
		if (mark > top_)
			top_ = mark;
	}

private:
	void *base_ = nullptr;
	size_t cap_ = 0, top_ = 0;
};

// Page-locked host scratch of one context.  Every small read-back (a count that sizes the next
// launch) and every host-built table goes through it, so no copy is staged through pageable
// memory; spans stay valid until the next reset() (= the next decompose on that context).
class HostScratch
{
public:
	~HostScratch()
	{
		for (auto &c : chunks_)
			(void)hipHostFree(c.p);
		if (ev_)
			(void)hipEventDestroy(ev_);
	}
	HostScratch() = default;
	HostScratch(const HostScratch &) = delete;
	HostScratch &operator=(const HostScratch &) = delete;
	// Words a kernel published into a span of this scratch (publish_words, or a kernel that writes page-locked memory itself)
	// are read WITHOUT leaving the stream idle: mark() right behind the publishing kernel, then the stream is given the
	// kernels that do not depend on the words, then wait() -- the host wakes up while they run.
	void mark(hipStream_t s)
	{
		if (!ev_)
			HIP_CHECK(hipEventCreateWithFlags(&ev_, hipEventDisableTiming));
		HIP_CHECK(hipEventRecord(ev_, s));
	}
	void wait() { HIP_CHECK(hipEventSynchronize(ev_)); }
	void reset()
	{
		for (auto &c : chunks_)
			c.top = 0;
	}
	template <typename T>
	T *take(size_t n)
	{
		const size_t bytes = (n * sizeof(T) + 63) & ~size_t(63);
		for (auto &c : chunks_)
			if (c.cap - c.top >= bytes) {
				char *r = c.p + c.top;
				c.top += bytes;
				return reinterpret_cast<T *>(r);
			}
		Chunk c{nullptr, std::max<size_t>(2 * bytes, size_t(1) << 16), bytes};
		HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&c.p), c.cap, hipHostMallocDefault));
		chunks_.push_back(c);
		return reinterpret_cast<T *>(c.p);
	}
	// one 32-bit word of device memory, through the stream
	uint32_t read_u32(const uint32_t *dptr, hipStream_t s)
	{
		uint32_t *h = take<uint32_t>(1);
		HIP_CHECK(copy_async(h, dptr, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		return *h;
	}

private:
	struct Chunk {
		char *p;
		size_t cap, top;
	};
	std::vector<Chunk> chunks_;
	hipEvent_t ev_ = nullptr;
};

// HIP-event stage timer on the context's stream.
struct StageTimer {
	struct Rec {
		std::string name;
		hipEvent_t a, b;
		uint32_t launches;
	};
	std::vector<Rec> recs;
	std::vector<hipEvent_t> pool;
	size_t pool_used = 0;
	hipStream_t stream = nullptr;
	bool enabled = true; // per-stage event pairs cost a few microseconds each; the pass total is always timed
	hipEvent_t get()
	{
		if (pool_used == pool.size()) {
			hipEvent_t e;
			HIP_CHECK(hipEventCreate(&e));
			pool.push_back(e);
		}
		return pool[pool_used++];
	}
	void reset()
	{
		recs.clear();
		pool_used = 0;
	}
	void begin(const char *name)
	{
		if (!enabled)
			return;
		Rec r{name, get(), get(), 0};
		HIP_CHECK(hipEventRecord(r.a, stream));
		recs.push_back(r);
	}
	void end(uint32_t launches)
	{
		if (!enabled)
			return;
		recs.back().launches = launches;
		HIP_CHECK(hipEventRecord(recs.back().b, stream));
	}
	~StageTimer()
	{
		for (auto e : pool)
			(void)hipEventDestroy(e);
	}
};

// device-wide primitives (primitives.hip)
void scan_exclusive_u32(const uint32_t *in, uint32_t *out, size_t n, void *tmp, size_t tmp_bytes, hipStream_t s);
// two independent scans that are due at the same point of the pass, in the same two launches
void scan_exclusive_u32_pair(const uint32_t *in0, uint32_t *out0, size_t n0, const uint32_t *in1, uint32_t *out1, size_t n1,
			     void *tmp, size_t tmp_bytes, hipStream_t s);
// exclusive running xor of two arrays of the same length (the two halves of 64-bit words), in the same two launches
void scan_exclusive_xor_u32_pair(const uint32_t *in0, uint32_t *out0, const uint32_t *in1, uint32_t *out1, size_t n, void *tmp,
				 size_t tmp_bytes, hipStream_t s);
// exclusive sums of BYTE inputs (flags, counts below 256): one job, or two independent ones in the same two launches
void scan_exclusive_u8(const uint8_t *in0, uint32_t *out0, size_t n0, const uint8_t *in1, uint32_t *out1, size_t n1, void *tmp,
		       size_t tmp_bytes, hipStream_t s);
// exclusive prefix sums of in[i] - sub[i] (mod 2^32)
void scan_exclusive_diff_u32(const uint32_t *in, const uint32_t *sub, uint32_t *out, size_t n, void *tmp, size_t tmp_bytes,
			     hipStream_t s);
// exclusive running xor of 128-bit words (two independent 64-bit hashes side by side)
// (n_dev, device memory, optional: only the first *n_dev + 1 words exist -- n then is the most there can be and sizes the launch)
void scan_exclusive_xor_u128(const ulonglong2 *in, ulonglong2 *out, size_t n, void *tmp, size_t tmp_bytes, hipStream_t s,
			     const uint32_t *n_dev = nullptr);
size_t scan_tmp_bytes(size_t n);
// indices of the non-zero bytes of flag[0..n), ascending, into out; their number into *count_dev (device memory).  No
// prefix array is written: tiles are counted, the counts scanned, the tiles ranked again.
// ---- bit-rank directory: one 16-byte record per 64 flags = {bits 0..31, bits 32..63, set bits in front of the record, 0}.
// rank(x) = flags set at positions < x is ONE 16-byte load and a popcount, and a table over 2 * 10^8 flags is 50 MB (it
// stays in the Infinity Cache) where the byte flags + their 4-byte prefix sums were 1 GB.  The producer writes the bit
// words (x, y); bitrank_build fills in z (three small launches over n / 64 records).  The array carries one record of
// slack: rank(n) is asked for.
__device__ __forceinline__ uint32_t bitrank(const uint4 *__restrict__ rec, uint32_t x)
{
	const uint4 r = rec[x >> 6];
	const unsigned long long bits = (unsigned long long)r.x | ((unsigned long long)r.y << 32);
	return r.z + (uint32_t)__popcll(bits & ((1ull << (x & 63u)) - 1ull));
}
__device__ __forceinline__ bool bitrank_test(const uint4 *__restrict__ rec, uint32_t x)
{
	const uint4 r = rec[x >> 6];
	return (((x & 32u) ? r.y : r.x) >> (x & 31u)) & 1u;
}
// spreads the low 16 bits of x to every fourth bit position (bit k -> bit 4 k)
__device__ __forceinline__ unsigned long long spread4(unsigned long long x)
{
	x &= 0xFFFFull;
	x = (x | (x << 24)) & 0x000000FF000000FFull;
	x = (x | (x << 12)) & 0x000F000F000F000Full;
	x = (x | (x << 6)) & 0x0303030303030303ull;
	x = (x | (x << 3)) & 0x1111111111111111ull;
	return x;
}
// A wave whose lane l holds four flags f (bits 0..3) of the positions 256 w + 4 l + {0..3} writes the four records of its
// 256 positions (rec = the wave's first record; every lane of the wave calls this)
__device__ __forceinline__ void bitrank_store_wave(uint4 *__restrict__ rec, uint32_t f, bool in_range)
{
	const uint32_t lane = threadIdx.x & 63u;
	const unsigned long long m0 = __ballot(f & 1u), m1 = __ballot(f & 2u), m2 = __ballot(f & 4u), m3 = __ballot(f & 8u);
	if (lane < 4 && in_range) {
		const unsigned sh = 16u * lane;
		const unsigned long long w = spread4(m0 >> sh) | (spread4(m1 >> sh) << 1) | (spread4(m2 >> sh) << 2) | (spread4(m3 >> sh) << 3);
		rec[lane] = make_uint4((uint32_t)w, (uint32_t)(w >> 32), 0u, 0u);
	}
}
// fills the z words of `rec` (n_rec records + one closing record whose bits are cleared here); tmp: (n_rec + 2) words + the scan's scratch
void bitrank_build(uint4 *rec, size_t n_rec, uint32_t *tmp_counts, void *scan_tmp, size_t scan_tmp_bytes, hipStream_t s);

// ---- a workgroup of 256 lanes appends the flagged positions of LIST_ITER rounds x 1024 positions (four a lane and round, bit
// 4 it + j of `fw`: position B0 + 1024 it + 4 tid + j) to a list, IN POSITION ORDER, with ONE atomic add on the list's
// length.  Atomic adds on one word retire at ~90 M/s on this chip (one per wave of 256 positions made such a kernel 8.9 ms
// on 2 * 10^8 positions), so they are kept to one per 16 384 positions.  The order matters for speed, not for results: the
// kernels that work through these lists want neighbouring lanes on neighbouring positions (a list that interleaved a
// workgroup's rounds made the class walks 35 % slower).
static constexpr uint32_t LIST_ITER = 16, LIST_TPB = 256, LIST_SPAN = LIST_TPB * 4 * LIST_ITER;
__device__ __forceinline__ void append_in_order(unsigned long long fw, uint32_t B0, uint32_t *__restrict__ list, uint32_t *__restrict__ n_list)
{
	__shared__ uint32_t tot[LIST_ITER * (LIST_TPB / 64)], base_sh;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
	for (uint32_t it = 0; it < LIST_ITER; it++) {
		const uint32_t f = (uint32_t)(fw >> (4 * it)) & 15u;
		const uint32_t t = (uint32_t)(__popcll(__ballot(f & 1u)) + __popcll(__ballot(f & 2u)) + __popcll(__ballot(f & 4u)) + __popcll(__ballot(f & 8u)));
		if (lane == 0)
			tot[it * (LIST_TPB / 64) + wave] = t;
	}
	__syncthreads();
	if (threadIdx.x < 64) { // rounds first, waves inside a round: exclusive prefix of the 64 counts, the total to the list
		const uint32_t v = tot[threadIdx.x];
		uint32_t inc = v;
		for (int off = 1; off < 64; off <<= 1) {
			const uint32_t y = __shfl_up(inc, off);
			if ((int)lane >= off)
				inc += y;
		}
		tot[threadIdx.x] = inc - v;
		if (lane == 63)
			base_sh = inc ? atomicAdd(n_list, inc) : 0u;
	}
	__syncthreads();
	const uint32_t base = base_sh;
#pragma unroll
	for (uint32_t it = 0; it < LIST_ITER; it++) {
		uint32_t f = (uint32_t)(fw >> (4 * it)) & 15u;
		// (every lane of the wave takes part in the ballots; lanes without a flag of this round only skip the stores)
		const uint32_t before = (uint32_t)(__popcll(__ballot(f & 1u) & lt) + __popcll(__ballot(f & 2u) & lt) + __popcll(__ballot(f & 4u) & lt) +
						   __popcll(__ballot(f & 8u) & lt));
		uint32_t at = base + tot[it * (LIST_TPB / 64) + wave] + before;
		while (f) {
			const int k = __ffs((int)f) - 1;
			f &= f - 1;
			list[at++] = B0 + it * (LIST_TPB * 4u) + threadIdx.x * 4u + (uint32_t)k;
		}
	}
}

// Up to 8 words of device memory into page-locked host memory by ONE single-lane kernel: a hipMemcpyAsync per word is a
// copy submission each (3-4 us apiece on the stream; config 2's whole pass is 1.1 ms).  `host_dst` is the HOST pointer of
// page-locked memory (HostScratch); read it after the stream is synchronised.
struct WordSrc {
	const uint32_t *p[8];
};
void publish_words(uint32_t *host_dst, const WordSrc &src, int n, hipStream_t s);
void compact_flagged_u8(const uint8_t *flag, size_t n, uint32_t *out, uint32_t *count_dev, void *tmp, size_t tmp_bytes, hipStream_t s);
size_t compact_tmp_bytes(size_t n);
// exclusive running maximum (identity 0)
void scan_exclusive_max_u32(const uint32_t *in, uint32_t *out, size_t n, void *tmp, size_t tmp_bytes, hipStream_t s);
// stable LSD radix sort of (key,value) pairs on the low `bits` bits of the key
void sort_pairs_u32(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned bits,
		    void *tmp, size_t tmp_bytes, hipStream_t s);
size_t sort_tmp_bytes(size_t n);

inline unsigned bits_for(uint64_t max_value)
{
	unsigned b = 1;
	while (b < 32 && (uint64_t(1) << b) <= max_value)
		b++;
	return b;
}

} // namespace povu_hip
