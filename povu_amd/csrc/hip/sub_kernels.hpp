// sub_kernels.hpp -- the three inserting passes of `povu decompose -s`: find_concealed, find_midi,
// find_smothered, see sub_kernels.hip
#pragma once
#include "leaf_kernels.hpp"

#include <memory>
#include <vector>

struct PinnedPool; // (context.hpp: the pool of page-locked result blocks of a context)

namespace povu_hip
{

// PVST line letters of the inserted vertices (include/povu/common/constants.hpp:53-60)
static constexpr uint8_t FAM_CONCEALED = 'C', FAM_MIDI = 'M', FAM_SMOTHERED = 'S';

// The PVSTs of one pass after all five passes of -s, on the host: component c owns the vertices [voff[c], voff[c + 1])
// (its own numbering starts at 0 = the dummy root; the flubble-like vertices keep their indices, the inserted ones
// follow) and every vertex its children in the reference's order.  The arrays lie in ONE page-locked block of the
// context's pool (the copies from the device run at link speed; the block goes back to the pool with the forest).
struct SubForest {
	std::vector<uint64_t> voff;   // [C + 1]
	std::vector<uint32_t> counts; // [3 C] concealed, midi, smothered vertices per component
	uint64_t n_vtx = 0, n_child = 0;
	const uint8_t *fam = nullptr, *or1 = nullptr, *or2 = nullptr, *route = nullptr; // per vertex: line letter, orientations ('>' = 0), 'L' / 'R' / 0
	const uint32_t *id1 = nullptr, *id2 = nullptr; // the two boundaries in print order
	const uint32_t *coff = nullptr;		       // [vertices + 1] children of a vertex: child[coff[x] .. coff[x + 1])
	const uint32_t *child = nullptr;	       // PVST indices inside the component
	std::shared_ptr<::PinnedPool> pool;
	void *blk = nullptr;
	size_t blk_cap = 0;
	int blk_seg = -1;
	SubForest() = default;
	SubForest(const SubForest &) = delete;
	SubForest &operator=(const SubForest &) = delete;
	~SubForest();
};

// Runs find_concealed, find_midi and find_smothered on the state leaf_prepare / leaf_dense left (an all-parallel pass whose
// tree stage also wrote the depths).  Throws HipError when a table outgrows its bound.  `arena` (optional): where the
// stage's tables go while it has room; *arena_hint = bytes to reserve there up front, updated to what this call needed.
void run_subflubbles(const CompState &cs, const SeqWs &sw, const ParWs &pw, const TreeWs &tw, const LeafState &ls, uint32_t C,
		     HostScratch &host, SubForest &out, const std::shared_ptr<::PinnedPool> &pool, hipStream_t s, Arena *arena = nullptr,
		     size_t *arena_hint = nullptr);

} // namespace povu_hip
