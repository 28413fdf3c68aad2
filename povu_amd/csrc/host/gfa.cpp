#include "gfa.hpp"

#include <algorithm>
#include <cstring>
#include <fcntl.h>
#include <stdexcept>
#include <sys/mman.h>
#include <sys/stat.h>
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

GfaGraph load_gfa(const std::string &fp, bool want_labels, bool want_paths)
{
	Mapped f(fp);
	struct Seg {
		uint32_t id;
		uint32_t order;
		const char *sb, *se;
	};
	std::vector<Seg> segs;
	std::vector<uint32_t> la, lb;
	std::vector<uint8_t> sa, sb;
	GfaGraph g;
	const char *p = f.p, *end = f.p + f.n;
	size_t line_no = 1;
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
			const std::string ln = std::to_string(line_no);
			if (nf < 2)
				throw std::runtime_error(invalid(fp, "S record on line " + ln + " is missing a segment id and sequence"));
			if (nf < 3)
				throw std::runtime_error(invalid(fp, "S record on line " + ln + " is missing a sequence"));
			if (fe[2] == fb[2])
				throw std::runtime_error(invalid(fp, "S record on line " + ln + " has an empty sequence"));
			uint32_t id;
			if (!parse_id(fb[1], fe[1], id))
				throw std::runtime_error(invalid(fp, "S record on line " + ln + " has a non-numeric segment id"));
			segs.push_back({id, (uint32_t)segs.size(), fb[2], fe[2]});
			break;
		}
		case 'L': {
			uint32_t a, b;
			if (nf < 5 || !parse_id(fb[1], fe[1], a) || !parse_id(fb[3], fe[3], b) || fe[2] - fb[2] != 1 ||
			    fe[4] - fb[4] != 1 || (*fb[2] != '+' && *fb[2] != '-') || (*fb[4] != '+' && *fb[4] != '-'))
				throw std::runtime_error(invalid(fp, "malformed L record on line " + std::to_string(line_no)));
			la.push_back(a);
			lb.push_back(b);
			sa.push_back(*fb[2] == '+' ? 1 : 0);
			sb.push_back(*fb[4] == '+' ? 0 : 1);
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
				g.paths.push_back(std::move(pa));
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
				g.paths.push_back(std::move(pa));
			}
			break;
		default:
			throw std::runtime_error(invalid(fp, "unsupported record type '" + std::string(1, *p) + "' on line " +
								     std::to_string(line_no)));
		}
		line_no++;
		p = next;
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
