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
// bottom kernel: a block stages 2*SEG_TPB leaves in LDS and writes that subtree's levels;
// top kernel: one block finishes the remaining (few thousand) nodes.
static __global__ void __launch_bounds__(SEG_TPB) k_seg_bottom(uint32_t P, uint32_t n, const uint32_t *__restrict__ val,
							       uint32_t *__restrict__ tree)
{
	__shared__ uint32_t sh[2 * SEG_TPB];
	const uint32_t W = 2 * SEG_TPB; // leaves per block
	const uint32_t base = blockIdx.x * W;
	for (uint32_t k = threadIdx.x; k < W; k += SEG_TPB) {
		uint32_t i = base + k;
		uint32_t v = (i < n) ? val[i] : NIL;
		sh[k] = v;
		if (i < P)
			tree[P + i] = v;
	}
	__syncthreads();
	// level with `w` nodes inside this block, global node index = (P/(W/w)) + blockIdx*w + k
	uint32_t lvlP = P;
	for (uint32_t w = W / 2; w >= 1; w >>= 1) {
		lvlP >>= 1;
		if (lvlP == 0)
			break;
		uint32_t a = 0;
		if (threadIdx.x < w)
			a = min(sh[2 * threadIdx.x], sh[2 * threadIdx.x + 1]);
		__syncthreads();
		if (threadIdx.x < w) {
			sh[threadIdx.x] = a;
			uint32_t node = lvlP + blockIdx.x * w + threadIdx.x;
			if (blockIdx.x * w + threadIdx.x < lvlP)
				tree[node] = a;
		}
		__syncthreads();
	}
}
static __global__ void __launch_bounds__(1024) k_seg_top(uint32_t top_nodes, uint32_t *tree)
{
	// nodes [1, top_nodes) ; level by level from the bottom (top_nodes is a power of two)
	for (uint32_t first = top_nodes / 2; first >= 1; first >>= 1) {
		for (uint32_t k = first + threadIdx.x; k < 2 * first; k += blockDim.x)
			tree[k] = min(tree[2 * k], tree[2 * k + 1]);
		__syncthreads();
		if (first == 1)
			break;
	}
}
static void seg_build(SegTree &st, const uint32_t *val, size_t n, hipStream_t s)
{
	st.P = SegTree::pow2(std::max<size_t>(n, 1));
	const uint32_t W = 2 * SEG_TPB;
	const uint32_t blocks = (st.P + W - 1) / W;
	KLAUNCH(k_seg_bottom, dim3(blocks), dim3(SEG_TPB), 0, s, st.P, (uint32_t)n, val, st.tree);
	if (st.P > W) // levels above the per-block subtrees: nodes [1, P/W)
		KLAUNCH(k_seg_top, dim3(1), dim3(1024), 0, s, st.P / W, st.tree);
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
// first idx in [l, r) whose value is < x, NIL if none.  Walks the disjoint subtrees to the right of l (leaf, then
// right siblings going up), descends into the first one whose minimum is < x; no per-thread node stack.
__device__ __forceinline__ uint32_t seg_first_less(const uint32_t *__restrict__ tree, uint32_t P, uint32_t l, uint32_t r, uint32_t x)
{
	if (l >= r)
		return NIL;
	uint32_t node = l + P;
	for (;;) {
		if (tree[node] < x) {
			uint32_t i = seg_descend_first(tree, P, node, x);
			return i < r ? i : NIL;
		}
		while (node & 1) // a right child: everything under its parent is either rejected or left of l
			node >>= 1;
		if (node == 0)
			return NIL; // ran off the right edge
		node += 1;
		// leftmost leaf under `node` is already >= r: nothing left to find (keeps the walk inside the query)
		const uint32_t first = node << (__clz(node) - __clz(P)); // P is a power of two
		if (first - P >= r)
			return NIL;
	}
}
// last idx in [l, r) whose value is < x, NIL if none (mirror image of seg_first_less)
__device__ __forceinline__ uint32_t seg_last_less(const uint32_t *__restrict__ tree, uint32_t P, uint32_t l, uint32_t r, uint32_t x)
{
	if (l >= r)
		return NIL;
	uint32_t node = r - 1 + P;
	for (;;) {
		if (tree[node] < x) {
			uint32_t i = seg_descend_last(tree, P, node, x);
			return i >= l ? i : NIL;
		}
		while (!(node & 1)) // a left child
			node >>= 1;
		if (node == 1)
			return NIL; // ran off the left edge
		node -= 1;
		const uint32_t last = ((node + 1) << (__clz(node) - __clz(P))) - 1;
		if (last - P < l)
			return NIL;
	}
}

} // namespace povu_hip
