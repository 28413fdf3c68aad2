// primitives.hip -- device-wide scan / stable radix sort (rocPRIM) used as plumbing
// between the hand-written decompose kernels.
#include "common.hpp"

#include <rocprim/rocprim.hpp>

namespace povu_hip
{

size_t scan_tmp_bytes(size_t n)
{
	size_t bytes = 0;
	(void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, n,
				      rocprim::plus<uint32_t>());
	return bytes + 256;
}

void scan_exclusive_u32(const uint32_t *in, uint32_t *out, size_t n, void *tmp, size_t tmp_bytes, hipStream_t s)
{
	if (n == 0)
		return;
	HIP_CHECK(rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s));
}

void scan_exclusive_max_u32(const uint32_t *in, uint32_t *out, size_t n, void *tmp, size_t tmp_bytes, hipStream_t s)
{
	if (n == 0)
		return;
	HIP_CHECK(rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0u, n, rocprim::maximum<uint32_t>(), s));
}

size_t sort_tmp_bytes(size_t n)
{
	size_t bytes = 0;
	(void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
					(const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 32);
	return bytes + 256;
}

void sort_pairs_u32(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n, unsigned bits,
		    void *tmp, size_t tmp_bytes, hipStream_t s)
{
	if (n == 0)
		return;
	HIP_CHECK(rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, bits, s));
}

} // namespace povu_hip
