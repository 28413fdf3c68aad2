// gfa.hpp -- GFA v1 tokenizer for the decompose path (row A, host part).
//
// Stands in for mto::from_gfa::to_bd + liteseq::gfa_new (src/mto/from_gfa.cpp:141-280; liteseq is an
// un-vendored third-party dependency).  Loader contract: vertices ascending by numeric
// segment id, links in L-line order, `+` on the source = right side, `+` on the sink = left side
// (from_gfa.cpp:223-243, inverse writer src/mto/to_gfa.cpp:24-33).  Validation and messages follow
// validate_gfa_for_liteseq (from_gfa.cpp:28-98).
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace povu_host
{

struct GfaPath {
	std::string name;
	std::vector<uint64_t> step_ids;
	std::vector<uint8_t> step_rev; // 0 forward, 1 reverse
};

struct GfaGraph {
	std::vector<uint32_t> vid;	 // ascending segment ids
	std::vector<uint32_t> v1, v2;	 // link endpoints (vertex idx), L-line order
	std::vector<uint8_t> s1, s2;	 // link endpoint sides (0 = l, 1 = r)
	std::vector<std::string> seq;	 // only with want_labels
	std::vector<GfaPath> paths;	 // only with want_paths (P and W records)
};

// throws std::runtime_error("Invalid GFA '<path>': ...") for the first malformed record in file order;
// `threads` tokenizer threads share the file (slices of whole lines, >= 4 MiB each)
// on_counts(segments, links), when given, is called once as soon as the tokenizer knows both numbers (the stitching and
// the link mapping still lie ahead): the CLI uses it to have the device memory reserved meanwhile
GfaGraph load_gfa(const std::string &path, bool want_labels = false, bool want_paths = false, int threads = 1,
		  const std::function<void(size_t, size_t)> &on_counts = nullptr);

} // namespace povu_host
