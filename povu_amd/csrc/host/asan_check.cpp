// asan_check.cpp -- CPU-only sanitizer harness for the host-side code of the decompose path (built by
// `make -C povu_amd/csrc asan` with -fsanitize=address,undefined; GPU sanitizers are not available on the test pool).
// Usage: host_asan_check <file.gfa|file.pvst>...   Every .gfa goes through the tokenizer (1 and 4 threads, with
// labels and paths), every .pvst through the reader; malformed inputs must fail with an error, not with a fault.
#include "../../../include/povu_hip.h"
#include "gfa.hpp"

#include <cstdio>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <string>

static bool ends_with(const std::string &s, const char *suf)
{
	const std::string t(suf);
	return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0;
}

int main(int argc, char **argv)
{
	unsigned long ok = 0, rejected = 0;
	for (int i = 1; i < argc; i++) {
		const std::string path = argv[i];
		if (ends_with(path, ".gfa")) {
			for (int threads : {1, 4}) {
				try {
					povu_host::GfaGraph g = povu_host::load_gfa(path, threads == 4, threads == 4, threads);
					if (g.v1.size() != g.v2.size() || g.v1.size() != g.s1.size() || g.v1.size() != g.s2.size())
						throw std::logic_error("link arrays of different length");
					for (size_t e = 0; e < g.v1.size(); e++)
						if (g.v1[e] >= g.vid.size() || g.v2[e] >= g.vid.size() || g.s1[e] > 1 || g.s2[e] > 1)
							throw std::logic_error("link out of range");
					ok++;
				} catch (const std::runtime_error &) {
					rejected++;
				}
			}
		} else if (ends_with(path, ".pvst")) {
			std::ifstream in(path, std::ios::binary);
			std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
			// the text as it is, then truncated at every length: the reader must never read past the buffer
			for (size_t len = text.size();; len = len > 64 ? len - 17 : len - 1) {
				char err[256];
				povu_pvst_doc *d = povu_pvst_parse(text.data(), len, err, sizeof err);
				if (d) {
					for (uint32_t k = 0; k < d->n; k++)
						if (d->parent[k] != POVU_HIP_NIL && d->parent[k] >= d->n)
							return 3;
					povu_pvst_doc_free(d);
					ok++;
				} else {
					rejected++;
				}
				if (len == 0)
					break;
			}
		}
	}
	printf("host_asan_check: %lu parsed, %lu rejected\n", ok, rejected);
	return 0;
}
