#include "gfa.hpp"

#include <algorithm>
#include <cstring>
#include <fcntl.h>
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
// what one tokenizer thread found in its slice of the file (whole lines, in file order)
struct Slice {
	std::vector<uint32_t> ids;  // segment ids in file order
	std::vector<Seg> segs;	    // ... with their sequences (only with want_labels)
	bool ascending = true;	    // ids strictly ascending inside the slice
	std::vector<uint32_t> la, lb;
	std::vector<uint8_t> sa, sb;
	std::vector<GfaPath> paths;
	size_t lines = 0;	  // lines seen (all of the slice unless an error stopped it)
	size_t err_line = 0;	  // 1-based line inside the slice of the first malformed record, 0 = none
	int err_kind = 0;	  // see slice_error
	char err_char = 0;
};

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
	auto fail = [&](int kind, char c = 0) {
		out.err_kind = kind;
		out.err_line = line_no;
		out.err_char = c;
	};
	while (p < end) {
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
			if (!out.ids.empty() && id <= out.ids.back())
				out.ascending = false;
			out.ids.push_back(id);
			if (want_labels)
				out.segs.push_back({id, fb[2], fe[2]});
			break;
		}
		case 'L': {
			uint32_t a, b;
			if (nf < 5 || !parse_id(fb[1], fe[1], a) || !parse_id(fb[3], fe[3], b) || fe[2] - fb[2] != 1 ||
			    fe[4] - fb[4] != 1 || (*fb[2] != '+' && *fb[2] != '-') || (*fb[4] != '+' && *fb[4] != '-'))
				return fail(5);
			out.la.push_back(a);
			out.lb.push_back(b);
			out.sa.push_back(*fb[2] == '+' ? 1 : 0);
			out.sb.push_back(*fb[4] == '+' ? 0 : 1);
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
		  const std::function<void(size_t, size_t)> &on_counts)
{
	Mapped f(fp);
	GfaGraph g;
	// slices of whole lines, one tokenizer thread each; results are stitched together in file order
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
	std::vector<Slice> slices(T);
	if (T == 1) {
		tokenize_slice(cut[0], cut[1], want_paths, want_labels, slices[0]);
	} else {
		std::vector<std::thread> pool;
		for (size_t t = 0; t < T; t++)
			pool.emplace_back([&, t]() { tokenize_slice(cut[t], cut[t + 1], want_paths, want_labels, slices[t]); });
		for (auto &th : pool)
			th.join();
	}
	size_t lines_before = 0;
	std::vector<size_t> seg_at(T + 1, 0), link_at(T + 1, 0);
	bool ascending = true; // segment ids strictly ascending over the whole file (what GFA writers usually produce)
	for (size_t t = 0; t < T; t++) { // the first malformed record in file order
		const Slice &sl = slices[t];
		if (sl.err_kind)
			throw std::runtime_error(slice_error(fp, sl.err_kind, lines_before + sl.err_line, sl.err_char));
		lines_before += sl.lines;
		seg_at[t + 1] = seg_at[t] + sl.ids.size();
		link_at[t + 1] = link_at[t] + sl.la.size();
		ascending = ascending && sl.ascending;
	}
	{ // (slices without segments do not break the order)
		uint32_t last = 0;
		bool any = false;
		for (const Slice &sl : slices)
			if (!sl.ids.empty()) {
				if (any && sl.ids.front() <= last)
					ascending = false;
				last = sl.ids.back();
				any = true;
			}
	}
	const size_t n_seg = seg_at[T], E = link_at[T];
	if (n_seg == 0)
		throw std::runtime_error(invalid(fp, "liteseq returned no vertices"));
	if (on_counts)
		on_counts(n_seg, E);
	// every thread moves its slice to its place in the final arrays
	U32Vec la(E), lb(E); // (uninitialised: every element is written by the thread that owns its slice)
	g.vid.resize(n_seg);
	g.s1.resize(E);
	g.s2.resize(E);
	std::vector<Seg> segs(want_labels ? n_seg : 0);
	{
		std::vector<std::thread> pool;
		for (size_t t = 0; t < T; t++)
			pool.emplace_back([&, t]() {
				Slice &sl = slices[t];
				std::copy(sl.ids.begin(), sl.ids.end(), g.vid.begin() + seg_at[t]);
				std::copy(sl.la.begin(), sl.la.end(), la.begin() + link_at[t]);
				std::copy(sl.lb.begin(), sl.lb.end(), lb.begin() + link_at[t]);
				std::copy(sl.sa.begin(), sl.sa.end(), g.s1.begin() + link_at[t]);
				std::copy(sl.sb.begin(), sl.sb.end(), g.s2.begin() + link_at[t]);
				if (want_labels)
					std::copy(sl.segs.begin(), sl.segs.end(), segs.begin() + seg_at[t]);
				sl.ids = {}, sl.la = {}, sl.lb = {}, sl.sa = {}, sl.sb = {}, sl.segs = {};
			});
		for (auto &th : pool)
			th.join();
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
	// id -> idx: direct table when ids are dense, binary search otherwise
	const uint32_t max_id = g.vid.back();
	U32Vec table;
	if ((uint64_t)max_id < 4 * (uint64_t)V + 1024) {
		table.resize((size_t)max_id + 1);
		parallel_ranges(TH, table.size(), [&](size_t, size_t lo, size_t hi) { std::fill(table.begin() + lo, table.begin() + hi, 0xFFFFFFFFu); });
		parallel_ranges(TH, V, [&](size_t, size_t lo, size_t hi) {
			for (size_t i = lo; i < hi; i++)
				table[g.vid[i]] = (uint32_t)i;
		});
	}
	g.v1.resize(E);
	g.v2.resize(E);
	std::vector<size_t> bad(TH, (size_t)-1); // first link of a range that names an unknown segment
	parallel_ranges(TH, E, [&](size_t t, size_t lo, size_t hi) {
		auto idx_of = [&](uint32_t id) -> uint32_t {
			if (!table.empty())
				return id <= max_id ? table[id] : 0xFFFFFFFFu;
			auto it = std::lower_bound(g.vid.begin(), g.vid.end(), id);
			return (it != g.vid.end() && *it == id) ? (uint32_t)(it - g.vid.begin()) : 0xFFFFFFFFu;
		};
		for (size_t e = lo; e < hi; e++) {
			const uint32_t a = idx_of(la[e]), b = idx_of(lb[e]);
			if ((a == 0xFFFFFFFFu || b == 0xFFFFFFFFu) && bad[t] == (size_t)-1)
				bad[t] = e;
			g.v1[e] = a;
			g.v2[e] = b;
		}
	});
	size_t first_bad = (size_t)-1;
	for (size_t b : bad)
		first_bad = std::min(first_bad, b);
	if (first_bad != (size_t)-1) {
		const uint32_t id = g.v1[first_bad] == 0xFFFFFFFFu ? la[first_bad] : lb[first_bad];
		throw std::runtime_error(invalid(fp, "L record " + std::to_string(first_bad) + " references unknown segment " +
							     std::to_string(id)));
	}
	return g;
}

} // namespace povu_host
