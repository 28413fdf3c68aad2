// segtree.hpp -- coarse min segment tree (par_kernels.hpp: SegTree): build kernels + device-side queries
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
static constexpr uint32_t SEG_BLK = SegTree::BLK;
// ------------------------------------------------------------- build
// bottom kernel: every lane reduces one block of SEG_BLK values (four 16-byte loads), the workgroup stages its SEG_TPB
// block minima in LDS and writes that subtree's levels; top kernel: one workgroup finishes the remaining nodes.
static __global__ void __launch_bounds__(SEG_TPB) k_seg_bottom(uint32_t P, uint32_t n, const uint32_t *__restrict__ val,
							       uint32_t *__restrict__ tree)
{
	__shared__ uint32_t sh[SEG_TPB];
	const uint32_t b = blockIdx.x * SEG_TPB + threadIdx.x; // block of values
	uint32_t m = NIL;
	const uint32_t e0 = b * SEG_BLK;
	if (e0 + SEG_BLK <= n) {
		const uint4 *q = reinterpret_cast<const uint4 *>(val + e0);
#pragma unroll
		for (uint32_t k = 0; k < SEG_BLK / 4; k++) {
			const uint4 a = q[k];
			m = min(min(m, a.x), min(min(a.y, a.z), a.w));
		}
	} else {
		for (uint32_t i = e0; i < n; i++)
			m = min(m, val[i]);
	}
	sh[threadIdx.x] = m;
	if (b < P)
		tree[P + b] = m;
	__syncthreads();
	// level with `w` nodes inside this workgroup, global node index = (P / (SEG_TPB / w)) + blockIdx * w + k
	uint32_t lvlP = P;
	for (uint32_t w = SEG_TPB / 2; w >= 1; w >>= 1) {
		lvlP >>= 1;
		if (lvlP == 0)
			break;
		uint32_t a = 0;
		if (threadIdx.x < w)
			a = min(sh[2 * threadIdx.x], sh[2 * threadIdx.x + 1]);
		__syncthreads();
		if (threadIdx.x < w) {
			sh[threadIdx.x] = a;
			if (blockIdx.x * w + threadIdx.x < lvlP)
				tree[lvlP + blockIdx.x * w + threadIdx.x] = a;
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
	if (reinterpret_cast<uintptr_t>(val) & 15)
		throw HipError("segment tree: the values must be 16-byte aligned");
	st.val = val;
	st.P = SegTree::pow2(std::max<size_t>((n + SEG_BLK - 1) / SEG_BLK, 1));
	const uint32_t blocks = (st.P + SEG_TPB - 1) / SEG_TPB;
	KLAUNCH(k_seg_bottom, dim3(blocks), dim3(SEG_TPB), 0, s, st.P, (uint32_t)n, val, st.tree);
	if (st.P > SEG_TPB) // levels above the per-workgroup subtrees: nodes [1, P / SEG_TPB)
		KLAUNCH(k_seg_top, dim3(1), dim3(1024), 0, s, st.P / SEG_TPB, st.tree);
}

// ------------------------------------------------------------- queries on the tree of block minima
__device__ __forceinline__ uint32_t cseg_min(const uint32_t *__restrict__ tree, uint32_t P, uint32_t l, uint32_t r)
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
__device__ __forceinline__ uint32_t cseg_descend_first(const uint32_t *__restrict__ tree, uint32_t P, uint32_t node, uint32_t x)
{
	while (node < P)
		node = tree[2 * node] < x ? 2 * node : 2 * node + 1;
	return node - P;
}
__device__ __forceinline__ uint32_t cseg_descend_last(const uint32_t *__restrict__ tree, uint32_t P, uint32_t node, uint32_t x)
{
	while (node < P)
		node = tree[2 * node + 1] < x ? 2 * node + 1 : 2 * node;
	return node - P;
}
// first block in [l, r) whose minimum is < x, NIL if none.  Walks the disjoint subtrees to the right of l (leaf, then
// right siblings going up), descends into the first one whose minimum is < x; no per-thread node stack.
__device__ __forceinline__ uint32_t cseg_first_less(const uint32_t *__restrict__ tree, uint32_t P, uint32_t l, uint32_t r, uint32_t x)
{
	if (l >= r)
		return NIL;
	uint32_t node = l + P;
	for (;;) {
		if (tree[node] < x) {
			uint32_t i = cseg_descend_first(tree, P, node, x);
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
// last block in [l, r) whose minimum is < x, NIL if none (mirror image of cseg_first_less)
__device__ __forceinline__ uint32_t cseg_last_less(const uint32_t *__restrict__ tree, uint32_t P, uint32_t l, uint32_t r, uint32_t x)
{
	if (l >= r)
		return NIL;
	uint32_t node = r - 1 + P;
	for (;;) {
		if (tree[node] < x) {
			uint32_t i = cseg_descend_last(tree, P, node, x);
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

// ------------------------------------------------------------- one block of values: four 16-byte loads, no loop over memory
// bit k of the result: value k of block b is < x (k restricted to the indices in [lo, hi))
__device__ __forceinline__ uint32_t blk_less_mask(const uint32_t *__restrict__ val, uint32_t b, uint32_t lo, uint32_t hi, uint32_t x)
{
	const uint4 *q = reinterpret_cast<const uint4 *>(val + b * SEG_BLK);
	uint32_t m = 0;
#pragma unroll
	for (uint32_t k = 0; k < SEG_BLK / 4; k++) {
		const uint4 a = q[k];
		m |= (a.x < x ? 1u : 0u) << (4 * k) | (a.y < x ? 2u : 0u) << (4 * k) | (a.z < x ? 4u : 0u) << (4 * k) |
		     (a.w < x ? 8u : 0u) << (4 * k);
	}
	const uint32_t e0 = b * SEG_BLK, k0 = lo > e0 ? lo - e0 : 0u, k1 = hi < e0 + SEG_BLK ? hi - e0 : SEG_BLK; // (hi > e0)
	const uint32_t keep = (k1 >= 32 ? 0xFFFFFFFFu : (1u << k1) - 1u) & ~((1u << k0) - 1u);
	return m & keep;
}
__device__ __forceinline__ uint32_t blk_min(const uint32_t *__restrict__ val, uint32_t b, uint32_t lo, uint32_t hi)
{
	const uint4 *q = reinterpret_cast<const uint4 *>(val + b * SEG_BLK);
	const uint32_t e0 = b * SEG_BLK;
	uint32_t m = NIL;
#pragma unroll
	for (uint32_t k = 0; k < SEG_BLK / 4; k++) {
		const uint4 a = q[k];
		const uint32_t i = e0 + 4 * k;
		m = min(m, (i >= lo && i < hi) ? a.x : NIL);
		m = min(m, (i + 1 >= lo && i + 1 < hi) ? a.y : NIL);
		m = min(m, (i + 2 >= lo && i + 2 < hi) ? a.z : NIL);
		m = min(m, (i + 3 >= lo && i + 3 < hi) ? a.w : NIL);
	}
	return m;
}

// ------------------------------------------------------------- queries over the values
__device__ __forceinline__ uint32_t seg_min(const SegTree &st, uint32_t l, uint32_t r)
{
	if (l >= r)
		return NIL;
	const uint32_t bl = l / SEG_BLK, br = (r - 1) / SEG_BLK;
	uint32_t m = blk_min(st.val, bl, l, r);
	if (br > bl) {
		m = min(m, blk_min(st.val, br, l, r));
		if (br > bl + 1)
			m = min(m, cseg_min(st.tree, st.P, bl + 1, br));
	}
	return m;
}
// first idx in [l, r) whose value is < x, NIL if none
__device__ __forceinline__ uint32_t seg_first_less(const SegTree &st, uint32_t l, uint32_t r, uint32_t x)
{
	if (l >= r)
		return NIL;
	const uint32_t bl = l / SEG_BLK, br = (r - 1) / SEG_BLK;
	uint32_t m = blk_less_mask(st.val, bl, l, r, x);
	if (m)
		return bl * SEG_BLK + (uint32_t)__ffs((int)m) - 1u;
	if (br == bl)
		return NIL;
	if (br > bl + 1) {
		const uint32_t b = cseg_first_less(st.tree, st.P, bl + 1, br, x);
		if (b != NIL) {
			m = blk_less_mask(st.val, b, l, r, x);
			return b * SEG_BLK + (uint32_t)__ffs((int)m) - 1u;
		}
	}
	m = blk_less_mask(st.val, br, l, r, x);
	return m ? br * SEG_BLK + (uint32_t)__ffs((int)m) - 1u : NIL;
}
// last idx in [l, r) whose value is < x, NIL if none
__device__ __forceinline__ uint32_t seg_last_less(const SegTree &st, uint32_t l, uint32_t r, uint32_t x)
{
	if (l >= r)
		return NIL;
	const uint32_t bl = l / SEG_BLK, br = (r - 1) / SEG_BLK;
	uint32_t m = blk_less_mask(st.val, br, l, r, x);
	if (m)
		return br * SEG_BLK + 31u - (uint32_t)__clz((int)m);
	if (br == bl)
		return NIL;
	if (br > bl + 1) {
		const uint32_t b = cseg_last_less(st.tree, st.P, bl + 1, br, x);
		if (b != NIL) {
			m = blk_less_mask(st.val, b, l, r, x);
			return b * SEG_BLK + 31u - (uint32_t)__clz((int)m);
		}
	}
	m = blk_less_mask(st.val, bl, l, r, x);
	return m ? bl * SEG_BLK + 31u - (uint32_t)__clz((int)m) : NIL;
}

} // namespace povu_hip
