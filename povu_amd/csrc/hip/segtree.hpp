// segtree.hpp -- array-based min segment tree: build kernels + device-side queries
// (range min, first / last index below a threshold).  Included by the .hip files that query it.
#pragma once
#include "par_kernels.hpp"

#include <algorithm>

namespace povu_hip
{
#ifndef NIL
#define NIL POVU_NIL
#endif
#ifndef SEG_TPB
#define SEG_TPB 256
#endif
// ------------------------------------------------------------- segment tree
static __global__ void k_seg_leaves(uint32_t P, uint32_t n, const uint32_t *__restrict__ val, uint32_t *__restrict__ tree)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < P)
		tree[P + i] = i < n ? val[i] : NIL;
}
static __global__ void k_seg_level(uint32_t first, uint32_t count, uint32_t *tree)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < count) {
		uint32_t k = first + i;
		tree[k] = min(tree[2 * k], tree[2 * k + 1]);
	}
}
static void seg_build(SegTree &st, const uint32_t *val, size_t n, hipStream_t s)
{
	st.P = SegTree::pow2(std::max<size_t>(n, 1));
	hipLaunchKernelGGL(k_seg_leaves, dim3((st.P + SEG_TPB - 1) / SEG_TPB), dim3(SEG_TPB), 0, s, st.P, (uint32_t)n, val, st.tree);
	for (uint32_t first = st.P / 2; first >= 1; first /= 2) {
		hipLaunchKernelGGL(k_seg_level, dim3((first + SEG_TPB - 1) / SEG_TPB), dim3(SEG_TPB), 0, s, first, first, st.tree);
		if (first == 1)
			break;
	}
}

__device__ __forceinline__ uint32_t seg_min(const uint32_t *__restrict__ tree, uint32_t P, uint32_t l, uint32_t r)
{
	uint32_t m = NIL;
	for (l += P, r += P; l < r; l >>= 1, r >>= 1) {
		if (l & 1)
			m = min(m, tree[l++]);
		if (r & 1)
			m = min(m, tree[--r]);
	}
	return m;
}
__device__ __forceinline__ uint32_t seg_descend_first(const uint32_t *__restrict__ tree, uint32_t P, uint32_t node,
						      uint32_t x)
{
	while (node < P)
		node = tree[2 * node] < x ? 2 * node : 2 * node + 1;
	return node - P;
}
__device__ __forceinline__ uint32_t seg_descend_last(const uint32_t *__restrict__ tree, uint32_t P, uint32_t node,
						     uint32_t x)
{
	while (node < P)
		node = tree[2 * node + 1] < x ? 2 * node + 1 : 2 * node;
	return node - P;
}
// first idx in [l, r) whose value is < x, NIL if none
static __device__ uint32_t seg_first_less(const uint32_t *__restrict__ tree, uint32_t P, uint32_t l, uint32_t r, uint32_t x)
{
	if (l >= r)
		return NIL;
	uint32_t right[32];
	int nr = 0;
	for (l += P, r += P; l < r; l >>= 1, r >>= 1) {
		if (l & 1) {
			if (tree[l] < x)
				return seg_descend_first(tree, P, l, x);
			l++;
		}
		if (r & 1)
			right[nr++] = --r;
	}
	for (int k = nr - 1; k >= 0; k--)
		if (tree[right[k]] < x)
			return seg_descend_first(tree, P, right[k], x);
	return NIL;
}
// last idx in [l, r) whose value is < x, NIL if none
static __device__ uint32_t seg_last_less(const uint32_t *__restrict__ tree, uint32_t P, uint32_t l, uint32_t r, uint32_t x)
{
	if (l >= r)
		return NIL;
	uint32_t left[32];
	int nl = 0;
	for (l += P, r += P; l < r; l >>= 1, r >>= 1) {
		if (r & 1) {
			--r;
			if (tree[r] < x)
				return seg_descend_last(tree, P, r, x);
		}
		if (l & 1)
			left[nl++] = l++;
	}
	for (int k = nl - 1; k >= 0; k--)
		if (tree[left[k]] < x)
			return seg_descend_last(tree, P, left[k], x);
	return NIL;
}


} // namespace povu_hip
