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
	std::vector<Seg> segs;
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

void tokenize_slice(const char *p, const char *end, bool want_paths, Slice &out)
{
	size_t line_no = 1;
	auto fail = [&](int kind, char c = 0) {
		out.err_kind = kind;
		out.err_line = line_no;
		out.err_char = c;
	};
	while (p < end) {
		const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
		const char *le = nl ? nl : end, *next = nl ? nl + 1 : end;
		if (le > p && le[-1] == '\r')
			le--;
		if (le == p) {
			line_no++;
			p = next;
			continue;
		}
		const char *fb[8], *fe[8];
		int nf = 0;
		for (const char *q = p; nf < 8;) {
			fb[nf] = q;
			const char *t = (const char *)memchr(q, '\t', (size_t)(le - q));
			fe[nf++] = t ? t : le;
			if (!t)
				break;
			q = t + 1;
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

GfaGraph load_gfa(const std::string &fp, bool want_labels, bool want_paths, int threads)
{
	Mapped f(fp);
	GfaGraph g;
	// slices of whole lines, one tokenizer thread each; results are stitched together in file order
	size_t T = (size_t)std::max(1, threads);
	T = std::min<size_t>(T, std::max<size_t>(1, f.n >> 22)); // at least 4 MiB per thread
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
		tokenize_slice(cut[0], cut[1], want_paths, slices[0]);
	} else {
		std::vector<std::thread> pool;
		for (size_t t = 0; t < T; t++)
			pool.emplace_back([&, t]() { tokenize_slice(cut[t], cut[t + 1], want_paths, slices[t]); });
		for (auto &th : pool)
			th.join();
	}
	size_t lines_before = 0, n_seg = 0, n_link = 0;
	for (const Slice &sl : slices) { // the first malformed record in file order
		if (sl.err_kind)
			throw std::runtime_error(slice_error(fp, sl.err_kind, lines_before + sl.err_line, sl.err_char));
		lines_before += sl.lines;
		n_seg += sl.segs.size();
		n_link += sl.la.size();
	}
	std::vector<Seg> segs;
	std::vector<uint32_t> la, lb;
	std::vector<uint8_t> sa, sb;
	segs.reserve(n_seg);
	la.reserve(n_link);
	lb.reserve(n_link);
	sa.reserve(n_link);
	sb.reserve(n_link);
	for (Slice &sl : slices) {
		segs.insert(segs.end(), sl.segs.begin(), sl.segs.end());
		la.insert(la.end(), sl.la.begin(), sl.la.end());
		lb.insert(lb.end(), sl.lb.begin(), sl.lb.end());
		sa.insert(sa.end(), sl.sa.begin(), sl.sa.end());
		sb.insert(sb.end(), sl.sb.begin(), sl.sb.end());
		for (auto &pa : sl.paths)
			g.paths.push_back(std::move(pa));
		sl = Slice{};
	}
	if (segs.empty())
		throw std::runtime_error(invalid(fp, "liteseq returned no vertices"));
	std::stable_sort(segs.begin(), segs.end(), [](const Seg &x, const Seg &y) { return x.id < y.id; });
	segs.erase(std::unique(segs.begin(), segs.end(), [](const Seg &x, const Seg &y) { return x.id == y.id; }), segs.end());
	const size_t V = segs.size(), E = la.size();
	g.vid.resize(V);
	for (size_t i = 0; i < V; i++)
		g.vid[i] = segs[i].id;
	if (want_labels) {
		g.seq.resize(V);
		for (size_t i = 0; i < V; i++)
			g.seq[i].assign(segs[i].sb, segs[i].se);
	}
	// id -> idx: direct table when ids are dense, binary search otherwise
	const uint32_t max_id = g.vid.back();
	std::vector<uint32_t> table;
	if ((uint64_t)max_id < 4 * (uint64_t)V + 1024) {
		table.assign((size_t)max_id + 1, 0xFFFFFFFFu);
		for (size_t i = 0; i < V; i++)
			table[g.vid[i]] = (uint32_t)i;
	}
	auto idx_of = [&](uint32_t id, size_t e) -> uint32_t {
		uint32_t r = 0xFFFFFFFFu;
		if (!table.empty()) {
			if (id <= max_id)
				r = table[id];
		} else {
			auto it = std::lower_bound(g.vid.begin(), g.vid.end(), id);
			if (it != g.vid.end() && *it == id)
				r = (uint32_t)(it - g.vid.begin());
		}
		if (r == 0xFFFFFFFFu)
			throw std::runtime_error(invalid(fp, "L record " + std::to_string(e) + " references unknown segment " +
								     std::to_string(id)));
		return r;
	};
	g.v1.resize(E);
	g.v2.resize(E);
	g.s1 = std::move(sa);
	g.s2 = std::move(sb);
	for (size_t e = 0; e < E; e++) {
		g.v1[e] = idx_of(la[e], e);
		g.v2[e] = idx_of(lb[e], e);
	}
	return g;
}

} // namespace povu_host
