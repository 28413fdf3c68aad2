#include "gfa.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <memory>
#include <stdexcept>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace povu_host
{
namespace
{
std::string invalid(const std::string &fp, const std::string &detail) { return "Invalid GFA '" + fp + "': " + detail; }

bool parse_id(const char *b, const char *e, uint32_t &out)
{
	if (b == e)
		return false;
	uint64_t v = 0;
	for (const char *p = b; p < e; p++) {
		if (*p < '0' || *p > '9')
			return false;
		v = v * 10 + (uint64_t)(*p - '0');
		if (v > 0xFFFFFFFEull)
			return false;
	}
	out = (uint32_t)v;
	return true;
}

struct Mapped {
	const char *p = nullptr;
	size_t n = 0;
	int fd = -1;
	explicit Mapped(const std::string &fp)
	{
		fd = open(fp.c_str(), O_RDONLY);
		if (fd < 0)
			throw std::runtime_error(invalid(fp, "could not open file"));
		struct stat st;
		if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
			close(fd);
			throw std::runtime_error(invalid(fp, "could not read file"));
		}
		n = (size_t)st.st_size;
		if (n) {
			void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
			if (m == MAP_FAILED) {
				close(fd);
				throw std::runtime_error(invalid(fp, "could not read file"));
			}
			p = (const char *)m;
			madvise(m, n, MADV_SEQUENTIAL);
		}
	}
	~Mapped()
	{
		if (p)
			munmap((void *)p, n);
		if (fd >= 0)
			close(fd);
	}
};
} // namespace

namespace
{
struct Seg {
	uint32_t id;
	const char *sb, *se;
};
// what the loader needs only while it runs (see GfaScratch in gfa.hpp)
struct Scratch {
	std::unique_ptr<Mapped> file;
	U32Vec table;  // id -> idx
	std::vector<Seg> segs;
};
// what one tokenizer thread does with its slice of the file (whole lines, in file order): the counting pass fills
// n_seg / n_link, the tokenizing pass writes the records at their final places in the graph's arrays
struct Slice {
	size_t n_seg = 0, n_link = 0; // S and L lines of the slice
	uint32_t *ids = nullptr;      // [n_seg] segment ids in file order
	Seg *segs = nullptr;	      // ... with their sequences (only with want_labels)
	uint32_t *la = nullptr, *lb = nullptr;
	uint8_t *sa = nullptr, *sb = nullptr;
	bool ascending = true; // ids strictly ascending inside the slice
	std::vector<GfaPath> paths;
	size_t lines = 0;	  // lines seen (all of the slice unless an error stopped it)
	size_t err_line = 0;	  // 1-based line inside the slice of the first malformed record, 0 = none
	int err_kind = 0;	  // see slice_error
	char err_char = 0;
};

// S and L lines of a slice (the record type is the first byte of a line): what sizes the graph's arrays before any
// record is parsed, so that every tokenizer writes its records where they stay
#if defined(__GNUC__) && !defined(__clang__)
__attribute__((optimize("O3"))) // (GCC only: the counting pass is built at -O3 whatever the file's level)
#endif
void count_slice(const char *p, const char *end, Slice &out)
{
	// a record starts the slice or follows a line feed: every byte pair is looked at without a branch (the loop
	// vectorises; a search for the next line feed per 20-byte line was four times slower)
	size_t ns = 0, nl = 0;
	if (p < end) {
		ns += *p == 'S';
		nl += *p == 'L';
	}
	const size_t n = (size_t)(end - p);
	for (size_t i0 = 0; i0 + 1 < n;) {
		const size_t i1 = std::min(n - 1, i0 + 4096);
		uint32_t cs = 0, cl = 0;
		for (size_t i = i0; i < i1; i++) {
			const bool lf = p[i] == '\n';
			cs += lf & (p[i + 1] == 'S');
			cl += lf & (p[i + 1] == 'L');
		}
		ns += cs;
		nl += cl;
		i0 = i1;
	}
	out.n_seg = ns;
	out.n_link = nl;
}

std::string slice_error(const std::string &fp, int kind, size_t line, char c)
{
	const std::string ln = std::to_string(line);
	switch (kind) {
	case 1:
		return invalid(fp, "S record on line " + ln + " is missing a segment id and sequence");
	case 2:
		return invalid(fp, "S record on line " + ln + " is missing a sequence");
	case 3:
		return invalid(fp, "S record on line " + ln + " has an empty sequence");
	case 4:
		return invalid(fp, "S record on line " + ln + " has a non-numeric segment id");
	case 5:
		return invalid(fp, "malformed L record on line " + ln);
	default:
		return invalid(fp, "unsupported record type '" + std::string(1, c) + "' on line " + ln);
	}
}

void tokenize_slice(const char *p, const char *end, bool want_paths, bool want_labels, Slice &out)
{
	size_t line_no = 1;
	uint32_t *ids = out.ids, *la = out.la, *lb = out.lb; // (the counting pass saw the same lines: the cursors end where
	uint8_t *sa = out.sa, *sb = out.sb;		       //  the next slice's records begin)
	Seg *segs = out.segs;
	uint32_t last_id = 0;
	auto fail = [&](int kind, char c = 0) {
		out.err_kind = kind;
		out.err_line = line_no;
		out.err_char = c;
	};
	// digits at q (at least one, value below 2^32 - 1) followed by a tab: the value, q behind the tab; false = not that shape
	auto id_tab = [&](const char *&q, uint32_t &v) {
		uint64_t x = 0;
		const char *q0 = q;
		while (q < end && (unsigned)(*q - '0') < 10u && q - q0 < 10) {
			x = x * 10 + (uint64_t)(*q - '0');
			q++;
		}
		if (q == q0 || q >= end || *q != '\t' || x > 0xFFFFFFFEull)
			return false;
		v = (uint32_t)x;
		q++;
		return true;
	};
	while (p < end) {
		// The two records a pangenome GFA consists of, in the shape every writer gives them, are read front to back in one
		// go (the id is evaluated while its end is being looked for; what follows the last field the path needs is skipped
		// with a library search for the line feed).  Anything else about a line -- other records, missing fields, ids that
		// are not numbers -- is left to the general code below, which also words the errors.
		if (p + 1 < end && p[1] == '\t') {
			const char *q = p + 2;
			uint32_t a, b;
			if (*p == 'L') {
				if (id_tab(q, a) && q + 1 < end && (*q == '+' || *q == '-') && q[1] == '\t') {
					const char oa = *q;
					q += 2;
					if (id_tab(q, b) && q < end && (*q == '+' || *q == '-') &&
					    (q + 1 == end || q[1] == '\t' || q[1] == '\n' ||
					     (q[1] == '\r' && (q + 2 == end || q[2] == '\n')))) {
						const char ob = *q;
						const char *nl = (const char *)memchr(q, '\n', (size_t)(end - q));
						*la++ = a;
						*lb++ = b;
						*sa++ = oa == '+' ? 1 : 0;
						*sb++ = ob == '+' ? 0 : 1;
						line_no++;
						p = nl ? nl + 1 : end;
						continue;
					}
				}
			} else if (*p == 'S' && !want_labels) {
				if (id_tab(q, a) && q < end && *q != '\t' && *q != '\n' && *q != '\r') { // (a sequence of at least one byte)
					const char *nl = (const char *)memchr(q, '\n', (size_t)(end - q));
					if (ids != out.ids && a <= last_id)
						out.ascending = false;
					last_id = a;
					*ids++ = a;
					line_no++;
					p = nl ? nl + 1 : end;
					continue;
				}
			}
		}
		// one pass over the line: its end and the first eight tab-separated fields (S and L lines of a pangenome GFA are
		// 13 - 30 bytes long: a library call per field cost more than the bytes)
		const char *fb[8], *fe[8];
		int nf = 0;
		bool open = true; // field nf is still being read
		fb[0] = p;
		const char *q = p;
		for (; q < end && *q != '\n'; q++)
			if (*q == '\t' && open) {
				fe[nf++] = q;
				if (nf < 8)
					fb[nf] = q + 1;
				else
					open = false;
			}
		const char *le = q, *next = q < end ? q + 1 : end;
		if (le > p && le[-1] == '\r')
			le--;
		if (le == p) {
			line_no++;
			p = next;
			continue;
		}
		if (open) {
			fe[nf] = le < fb[nf] ? fb[nf] : le;
			nf++;
		}
		switch (*p) {
		case 'H':
			break;
		case 'S': {
			if (nf < 2)
				return fail(1);
			if (nf < 3)
				return fail(2);
			if (fe[2] == fb[2])
				return fail(3);
			uint32_t id;
			if (!parse_id(fb[1], fe[1], id))
				return fail(4);
			if (ids != out.ids && id <= last_id)
				out.ascending = false;
			last_id = id;
			*ids++ = id;
			if (want_labels)
				*segs++ = Seg{id, fb[2], fe[2]};
			break;
		}
		case 'L': {
			uint32_t a, b;
			if (nf < 5 || !parse_id(fb[1], fe[1], a) || !parse_id(fb[3], fe[3], b) || fe[2] - fb[2] != 1 ||
			    fe[4] - fb[4] != 1 || (*fb[2] != '+' && *fb[2] != '-') || (*fb[4] != '+' && *fb[4] != '-'))
				return fail(5);
			*la++ = a;
			*lb++ = b;
			*sa++ = *fb[2] == '+' ? 1 : 0;
			*sb++ = *fb[4] == '+' ? 0 : 1;
			break;
		}
		case 'P':
			if (want_paths && nf >= 3) {
				GfaPath pa;
				pa.name.assign(fb[1], fe[1]);
				for (const char *q = fb[2]; q < fe[2];) {
					const char *c = (const char *)memchr(q, ',', (size_t)(fe[2] - q));
					const char *te = c ? c : fe[2];
					if (te - q >= 2) {
						uint32_t id;
						if (parse_id(q, te - 1, id)) {
							pa.step_ids.push_back(id);
							pa.step_rev.push_back(te[-1] == '-' ? 1 : 0);
						}
					}
					q = c ? c + 1 : fe[2];
				}
				out.paths.push_back(std::move(pa));
			}
			break;
		case 'W':
			if (want_paths && nf >= 7) {
				GfaPath pa;
				pa.name = std::string(fb[1], fe[1]) + "#" + std::string(fb[2], fe[2]) + "#" + std::string(fb[3], fe[3]);
				const char *q = fb[6];
				while (q < fe[6]) {
					char o = *q++;
					const char *st = q;
					while (q < fe[6] && *q != '>' && *q != '<')
						q++;
					uint32_t id;
					if ((o == '>' || o == '<') && parse_id(st, q, id)) {
						pa.step_ids.push_back(id);
						pa.step_rev.push_back(o == '<' ? 1 : 0);
					}
				}
				out.paths.push_back(std::move(pa));
			}
			break;
		default:
			return fail(6, *p);
		}
		line_no++;
		p = next;
	}
	out.lines = line_no - 1;
}
} // namespace

namespace
{
// fn(t, lo, hi) over [0, n) cut into T contiguous ranges, one thread each
template <typename F>
void parallel_ranges(size_t T, size_t n, F &&fn)
{
	T = std::max<size_t>(1, std::min(T, n / 65536 + 1));
	if (T == 1) {
		fn(0, 0, n);
		return;
	}
	std::vector<std::thread> pool;
	for (size_t t = 0; t < T; t++)
		pool.emplace_back([&, t]() { fn(t, n / T * t, t + 1 == T ? n : n / T * (t + 1)); });
	for (auto &th : pool)
		th.join();
}
} // namespace

GfaGraph load_gfa(const std::string &fp, bool want_labels, bool want_paths, int threads,
		  const std::function<void(size_t, size_t)> &on_counts, GfaScratch *keep)
{
	// POVU_GFA_TIMING=1: the loader's phases on stderr (tools/first_call_cost.py and DESIGN.md section 6 quote them)
	const bool timing = std::getenv("POVU_GFA_TIMING") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto phase = [&](const char *what) {
		if (!timing)
			return;
		const auto now = std::chrono::steady_clock::now();
		std::fprintf(stderr, "gfa %-12s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
		t_last = now;
	};
	auto scratch = std::make_shared<Scratch>();
	scratch->file = std::make_unique<Mapped>(fp);
	if (keep)
		keep->held = scratch; // (on every path out of here, errors included, the caller decides when it goes)
	const Mapped &f = *scratch->file;
	GfaGraph g;
	// slices of whole lines, one thread each: a counting pass sizes the arrays, the tokenizing pass fills them in place
	const size_t TH = (size_t)std::max(1, threads);
	size_t T = std::min<size_t>(TH, std::max<size_t>(1, f.n >> 22)); // at least 4 MiB per thread
	std::vector<const char *> cut(T + 1);
	cut[0] = f.p;
	cut[T] = f.p + f.n;
	for (size_t t = 1; t < T; t++) {
		const char *q = f.p + f.n / T * t;
		if (q < cut[t - 1])
			q = cut[t - 1];
		const char *nl = (const char *)memchr(q, '\n', (size_t)(cut[T] - q));
		cut[t] = nl ? nl + 1 : cut[T];
	}
	phase("map");
	auto for_slices = [&](auto &&fn) {
		if (T == 1) {
			fn(0);
			return;
		}
		std::vector<std::thread> pool;
		for (size_t t = 0; t < T; t++)
			pool.emplace_back([&, t]() { fn(t); });
		for (auto &th : pool)
			th.join();
	};
	std::vector<Slice> slices(T);
	for_slices([&](size_t t) { count_slice(cut[t], cut[t + 1], slices[t]); });
	std::vector<size_t> seg_at(T + 1, 0), link_at(T + 1, 0);
	for (size_t t = 0; t < T; t++) {
		seg_at[t + 1] = seg_at[t] + slices[t].n_seg;
		link_at[t + 1] = link_at[t] + slices[t].n_link;
	}
	const size_t n_seg = seg_at[T], E = link_at[T];
	phase("count");
	if (on_counts && n_seg)
		on_counts(n_seg, E);
	// the graph's arrays at their final size (uninitialised: every element is written, and its page first touched, by the
	// thread that owns the slice); v1 / v2 hold the link ends as segment ids until the ids can be mapped, in place
	Scratch &sc = *scratch;
	g.v1.resize(E);
	g.v2.resize(E);
	g.vid.resize(n_seg);
	g.s1.resize(E);
	g.s2.resize(E);
	sc.segs.resize(want_labels ? n_seg : 0);
	U32Vec &la = g.v1, &lb = g.v2;
	std::vector<Seg> &segs = sc.segs;
	for_slices([&](size_t t) {
		Slice &sl = slices[t];
		sl.ids = g.vid.data() + seg_at[t];
		sl.segs = want_labels ? segs.data() + seg_at[t] : nullptr;
		sl.la = la.data() + link_at[t];
		sl.lb = lb.data() + link_at[t];
		sl.sa = g.s1.data() + link_at[t];
		sl.sb = g.s2.data() + link_at[t];
		tokenize_slice(cut[t], cut[t + 1], want_paths, want_labels, sl);
	});
	phase("tokenize");
	size_t lines_before = 0;
	bool ascending = true; // segment ids strictly ascending over the whole file (what GFA writers usually produce)
	for (size_t t = 0; t < T; t++) { // the first malformed record in file order
		const Slice &sl = slices[t];
		if (sl.err_kind)
			throw std::runtime_error(slice_error(fp, sl.err_kind, lines_before + sl.err_line, sl.err_char));
		lines_before += sl.lines;
		ascending = ascending && sl.ascending;
	}
	if (n_seg == 0)
		throw std::runtime_error(invalid(fp, "liteseq returned no vertices"));
	for (size_t t = 0, last_t = T; t < T; t++) // (slices without segments do not break the order)
		if (slices[t].n_seg) {
			if (last_t != T && g.vid[seg_at[t]] <= g.vid[seg_at[last_t + 1] - 1])
				ascending = false;
			last_t = t;
		}
	for (Slice &sl : slices)
		for (auto &pa : sl.paths)
			g.paths.push_back(std::move(pa));
	if (!ascending) { // vertices ascending by segment id, the first record of an id wins
		if (want_labels) {
			std::stable_sort(segs.begin(), segs.end(), [](const Seg &x, const Seg &y) { return x.id < y.id; });
			segs.erase(std::unique(segs.begin(), segs.end(), [](const Seg &x, const Seg &y) { return x.id == y.id; }), segs.end());
			g.vid.resize(segs.size());
			for (size_t i = 0; i < segs.size(); i++)
				g.vid[i] = segs[i].id;
		} else {
			std::sort(g.vid.begin(), g.vid.end());
			g.vid.erase(std::unique(g.vid.begin(), g.vid.end()), g.vid.end());
		}
	}
	const size_t V = g.vid.size();
	if (want_labels) {
		g.seq.resize(V);
		for (size_t i = 0; i < V; i++)
			g.seq[i].assign(segs[i].sb, segs[i].se);
	}
	// id -> idx: plain arithmetic when the ids are consecutive (what graph builders write: 1 .. V), a direct table when they
	// are dense, binary search otherwise
	const uint32_t min_id = g.vid.front(), max_id = g.vid.back();
	const bool consecutive = (uint64_t)max_id - min_id + 1 == (uint64_t)V; // (the ids are distinct and ascending here)
	U32Vec &table = sc.table;
	if (!consecutive && (uint64_t)max_id < 4 * (uint64_t)V + 1024) {
		table.resize((size_t)max_id + 1);
		parallel_ranges(TH, table.size(), [&](size_t, size_t lo, size_t hi) { std::fill(table.begin() + lo, table.begin() + hi, 0xFFFFFFFFu); });
		parallel_ranges(TH, V, [&](size_t, size_t lo, size_t hi) {
			for (size_t i = lo; i < hi; i++)
				table[g.vid[i]] = (uint32_t)i;
		});
	}
	phase("id table");
	struct Bad {
		size_t link = (size_t)-1; // first link of a range that names an unknown segment
		uint32_t id = 0;	  // ... and the name
	};
	std::vector<Bad> bad(TH);
	parallel_ranges(TH, E, [&](size_t t, size_t lo, size_t hi) {
		Bad first;
		if (consecutive) {
			const uint32_t span = max_id - min_id; // (id - min_id wraps for ids below min_id: beyond span as well)
			for (size_t e = hi; e-- > lo;) { // (descending, so that the first bad link of the range is the one that stays)
				const uint32_t ia = la[e], ib = lb[e], a = ia - min_id, b = ib - min_id;
				la[e] = a <= span ? a : 0xFFFFFFFFu;
				lb[e] = b <= span ? b : 0xFFFFFFFFu;
				if (a > span || b > span)
					first = Bad{e, a > span ? ia : ib};
			}
			bad[t] = first;
			return;
		}
		auto idx_of = [&](uint32_t id) -> uint32_t {
			if (!table.empty())
				return id <= max_id ? table[id] : 0xFFFFFFFFu;
			auto it = std::lower_bound(g.vid.begin(), g.vid.end(), id);
			return (it != g.vid.end() && *it == id) ? (uint32_t)(it - g.vid.begin()) : 0xFFFFFFFFu;
		};
		for (size_t e = lo; e < hi; e++) {
			const uint32_t ia = la[e], ib = lb[e], a = idx_of(ia), b = idx_of(ib);
			if ((a == 0xFFFFFFFFu || b == 0xFFFFFFFFu) && first.link == (size_t)-1)
				first = Bad{e, a == 0xFFFFFFFFu ? ia : ib};
			la[e] = a;
			lb[e] = b;
		}
		bad[t] = first;
	});
	phase("link ends");
	Bad first_bad;
	for (const Bad &b : bad)
		if (b.link < first_bad.link)
			first_bad = b;
	if (first_bad.link != (size_t)-1)
		throw std::runtime_error(invalid(fp, "L record " + std::to_string(first_bad.link) + " references unknown segment " +
							     std::to_string(first_bad.id)));
	return g;
}

} // namespace povu_host
