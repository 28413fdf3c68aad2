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
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace povu_host
{

struct GfaPath {
	std::string name;
	std::vector<uint64_t> step_ids;
	std::vector<uint8_t> step_rev; // 0 forward, 1 reverse
};

// std::vector whose resize() leaves trivially constructible elements uninitialised: the big arrays of a whole-genome graph
// are filled by the tokenizer's threads, which then also are the first to touch their pages -- value-initialising a few
// gigabytes on ONE thread beforehand was a third of the parse
template <class T>
struct NoInitAlloc {
	using value_type = T;
	NoInitAlloc() = default;
	template <class U>
	NoInitAlloc(const NoInitAlloc<U> &)
	{}
	T *allocate(size_t n)
	{
		if (n > size_t(-1) / sizeof(T))
			throw std::bad_array_new_length();
		return static_cast<T *>(::operator new(n * sizeof(T)));
	}
	void deallocate(T *p, size_t) { ::operator delete(p); }
	template <class U, class... A>
	void construct(U *p, A &&...a)
	{
		if constexpr (sizeof...(A) == 0)
			::new ((void *)p) U; // default-initialisation: nothing for an integer
		else
			::new ((void *)p) U(std::forward<A>(a)...);
	}
	template <class U>
	bool operator==(const NoInitAlloc<U> &) const
	{
		return true;
	}
	template <class U>
	bool operator!=(const NoInitAlloc<U> &) const
	{
		return false;
	}
};
using U32Vec = std::vector<uint32_t, NoInitAlloc<uint32_t>>;
using U8Vec = std::vector<uint8_t, NoInitAlloc<uint8_t>>;

struct GfaGraph {
	U32Vec vid;			 // ascending segment ids
	U32Vec v1, v2;			 // link endpoints (vertex idx), L-line order
	U8Vec s1, s2;			 // link endpoint sides (0 = l, 1 = r)
	std::vector<std::string> seq;	 // only with want_labels
	std::vector<GfaPath> paths;	 // only with want_paths (P and W records)
};

// What the loader needs only while it runs -- the file mapping, the link ends as segment ids, the id table: gigabytes on a
// whole-genome graph, and giving them back to the system (munmap) takes a few hundred milliseconds.  A caller that passes a
// GfaScratch receives them instead of having them released before load_gfa returns, and drops them when and where it
// suits (the CLI: on a thread of its own, while the graph is uploaded).
struct GfaScratch {
	std::shared_ptr<void> held;
};

// throws std::runtime_error("Invalid GFA '<path>': ...") for the first malformed record in file order;
// `threads` threads share the file (slices of whole lines, >= 4 MiB each): a counting pass, then every thread tokenizes
// its slice straight into the graph's arrays.
// on_counts(segments, links), when given, is called once after the counting pass (the tokenizing and the link mapping
// still lie ahead; a malformed record may yet make the load fail): the CLI has the device memory reserved meanwhile
GfaGraph load_gfa(const std::string &path, bool want_labels = false, bool want_paths = false, int threads = 1,
		  const std::function<void(size_t, size_t)> &on_counts = nullptr, GfaScratch *keep = nullptr);

} // namespace povu_host
