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
	}
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
