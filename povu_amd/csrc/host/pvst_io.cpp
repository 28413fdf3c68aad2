// pvst_io.cpp -- reader for the PVST wire format (the consumer side of row H).
// Follows mto::from_pvst::read_pvst (src/mto/from_pvst.cpp:162-302) and pvst::Tree::comp_heights
// (include/povu/graph/pvst.hpp:807-836): five tab-separated columns per line, header `H <version>`
// (only 0.0.3 is supported, :37-76), vertex lines D/F/T/O/M/C/S with label `<or><id><or><id>`
// (:136-158), children as comma-separated FILE vertex ids (:83-130, :284-297), route L/R (:155-157).
#include "../../../include/povu_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace
{
void set_err(char *err, size_t n, const std::string &m)
{
	if (err && n)
		snprintf(err, n, "%s", m.c_str());
}
} // namespace

// GFA v1 text of a whole graph, the inverse of the loader contract (host/gfa.cpp; mto::to_gfa::write_gfa,
// src/mto/to_gfa.cpp:13-56, writes the same three record shapes): header, one `S <id> A` per segment in vertex order, one
// `L <a> <+|-> <b> <+|-> 0M` per link in link order, `+` = leaves a through its r side / enters b through its l side.
// Hand-rolled number formatting into a large buffer: a whole-genome graph is several gigabytes of text.
extern "C" int povu_hip_gfa_write(const char *path, uint32_t n_vtx, const uint32_t *vid, uint32_t n_links, const uint32_t *v1,
				  const uint8_t *s1, const uint32_t *v2, const uint8_t *s2, char *err, size_t errlen)
{
	if (!path || (n_vtx && !vid) || (n_links && (!v1 || !s1 || !v2 || !s2))) {
		set_err(err, errlen, "gfa_write: bad arguments");
		return 1;
	}
	FILE *o = fopen(path, "wb");
	if (!o) {
		set_err(err, errlen, std::string("gfa_write: could not open ") + path);
		return 1;
	}
	const size_t cap = size_t(8) << 20;
	std::vector<char> buf(cap + 64);
	size_t at = 0;
	bool ok = true;
	auto flush = [&]() {
		ok = ok && fwrite(buf.data(), 1, at, o) == at;
		at = 0;
	};
	auto num = [&](uint32_t v) {
		char t[10];
		int n = 0;
		do {
			t[n++] = (char)('0' + v % 10);
			v /= 10;
		} while (v);
		while (n)
			buf[at++] = t[--n];
	};
	memcpy(buf.data(), "H\tVN:Z:1.0\n", 11);
	at = 11;
	for (uint32_t v = 0; v < n_vtx; v++) {
		buf[at++] = 'S', buf[at++] = '\t';
		num(vid[v]);
		buf[at++] = '\t', buf[at++] = 'A', buf[at++] = '\n';
		if (at >= cap)
			flush();
	}
	for (uint32_t e = 0; e < n_links; e++) {
		if (v1[e] >= n_vtx || v2[e] >= n_vtx) {
			fclose(o);
			set_err(err, errlen, "gfa_write: link " + std::to_string(e) + " names an unknown vertex");
			return 1;
		}
		buf[at++] = 'L', buf[at++] = '\t';
		num(vid[v1[e]]);
		buf[at++] = '\t', buf[at++] = s1[e] == POVU_SIDE_R ? '+' : '-', buf[at++] = '\t';
		num(vid[v2[e]]);
		buf[at++] = '\t', buf[at++] = s2[e] == POVU_SIDE_L ? '+' : '-', buf[at++] = '\t', buf[at++] = '0', buf[at++] = 'M', buf[at++] = '\n';
		if (at >= cap)
			flush();
	}
	flush();
	ok = (fclose(o) == 0) && ok;
	if (!ok) {
		set_err(err, errlen, std::string("gfa_write: short write to ") + path);
		return 1;
	}
	return 0;
}

extern "C" povu_pvst_doc *povu_pvst_parse(const char *text, size_t len, char *err, size_t errlen)
{
	if (!text) {
		set_err(err, errlen, "null PVST text");
		return nullptr;
	}
	struct Line {
		char typ;
		uint32_t file_id;
		std::string label, children, route;
	};
	std::vector<Line> rows;
	size_t line_idx = 0;
	const char *p = text, *end = text + len;
	bool have_header = false;
	while (p < end) {
		const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
		const char *le = nl ? nl : end;
		std::vector<std::string> tok;
		for (const char *q = p;;) {
			const char *t = (const char *)memchr(q, '\t', (size_t)(le - q));
			tok.emplace_back(q, t ? t : le);
			if (!t)
				break;
			q = t + 1;
		}
		if (tok.size() != 5) { // EXPECTED_PVST_COL_NUMS, constants.hpp:66
			set_err(err, errlen, "invalid number of columns. line " + std::to_string(line_idx) + ". Expected 5, got " +
						     std::to_string(tok.size()));
			return nullptr;
		}
		const char typ = tok[0].empty() ? '?' : tok[0][0];
		if (typ == 'H') {
			if (tok[1] != "0.0.3") {
				set_err(err, errlen, "Unsupported PVST version, got " + tok[1] + ". Supported versions are: 0.0.3.");
				return nullptr;
			}
			have_header = true;
		} else if (strchr("DFTOMCS", typ)) {
			char *e = nullptr;
			unsigned long id = strtoul(tok[1].c_str(), &e, 10);
			if (e == tok[1].c_str()) {
				set_err(err, errlen, "non-numeric vertex id on line " + std::to_string(line_idx + 1));
				return nullptr;
			}
			rows.push_back({typ, (uint32_t)id, tok[2], tok[3], tok[4]});
		} else {
			set_err(err, errlen, "Unknown vertex type in PVST: L:" + std::to_string(line_idx + 1));
			return nullptr;
		}
		line_idx++;
		p = nl ? nl + 1 : end;
	}
	(void)have_header;
	const uint32_t n = (uint32_t)rows.size();
	auto *d = (povu_pvst_doc *)calloc(1, sizeof(povu_pvst_doc));
	d->n = n;
	d->type = (char *)calloc(n + 1, 1);
	d->file_id = (uint32_t *)calloc(n + 1, 4);
	d->a_id = (uint32_t *)calloc(n + 1, 4);
	d->z_id = (uint32_t *)calloc(n + 1, 4);
	d->parent = (uint32_t *)calloc(n + 1, 4);
	d->height = (uint32_t *)calloc(n + 1, 4);
	d->a_or = (uint8_t *)calloc(n + 1, 1);
	d->z_or = (uint8_t *)calloc(n + 1, 1);
	d->route = (uint8_t *)calloc(n + 1, 1);
	std::map<uint32_t, uint32_t> by_file_id; // file_v_idx_to_pvst_idx, :180
	uint32_t root = POVU_HIP_NIL;
	for (uint32_t i = 0; i < n; i++) {
		const Line &r = rows[i];
		d->type[i] = r.typ;
		d->file_id[i] = r.file_id;
		d->parent[i] = POVU_HIP_NIL;
		by_file_id[r.file_id] = i;
		if (r.typ == 'D') {
			d->a_id[i] = d->z_id[i] = POVU_HIP_NIL;
			root = i;
			continue;
		}
		// str_to_id_or_t, :136-151
		const size_t first = r.label.find_first_of("><"), last = r.label.find_last_of("><");
		if (first == std::string::npos || last == first) {
			povu_pvst_doc_free(d);
			set_err(err, errlen, "malformed vertex label '" + r.label + "'");
			return nullptr;
		}
		d->a_id[i] = (uint32_t)strtoull(r.label.substr(first + 1, last - first - 1).c_str(), nullptr, 10);
		d->a_or[i] = r.label[first] == '>' ? 0 : 1;
		d->z_id[i] = (uint32_t)strtoull(r.label.substr(last + 1).c_str(), nullptr, 10);
		d->z_or[i] = r.label[last] == '>' ? 0 : 1;
		d->route[i] = (!r.route.empty() && r.route[0] == 'L') ? 0 : 1;
	}
	// children column -> add_edge(parent, child) in listed order (:284-297, split_numbers :83-130): the parent pointer of a
	// vertex is its LAST lister, the children vectors hold every listing (`decompose -s` writes trees in which a concealed
	// vertex is listed under two parents and a flubble under none: concealed.cpp:1077)
	std::vector<uint32_t> coff(n + 1, 0), cadj;
	for (int pass = 0; pass < 2; pass++) {
		std::vector<uint32_t> cur(coff.begin(), coff.end() - 1);
		for (uint32_t i = 0; i < n; i++) {
			const std::string &ch = rows[i].children;
			if (ch == ".")
				continue;
			size_t pos = 0;
			while (pos < ch.size()) {
				size_t comma = ch.find(',', pos);
				std::string t = ch.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
				char *e = nullptr;
				long v = strtol(t.c_str(), &e, 10);
				if (e != t.c_str()) {
					auto it = by_file_id.find((uint32_t)v);
					if (it != by_file_id.end()) {
						if (pass == 0) {
							d->parent[it->second] = i;
							coff[i + 1]++;
						} else {
							cadj[cur[i]++] = it->second;
						}
					}
				}
				if (comma == std::string::npos)
					break;
				pos = comma + 1;
			}
		}
		if (pass == 0) {
			for (uint32_t i = 0; i < n; i++)
				coff[i + 1] += coff[i];
			cadj.resize(coff[n] + 1);
		}
	}
	// comp_heights (pvst.hpp:807-836), literally: a stack, every child of the popped vertex gets its height and is pushed --
	// a vertex listed under two parents is walked twice and keeps the height of the last visit, vertices the root does not
	// reach keep 0.  The reference does not end on a file whose listings form a cycle (and takes 2^k steps on a chain of k
	// doubly-listed diamonds); here the walk gives up after 64 n + 1024 pushes and the parse FAILS with a message -- heights
	// that are half filled in are not handed out as if they were the reference's.
	if (root != POVU_HIP_NIL) {
		std::vector<uint32_t> stack;
		uint64_t pushes = 0;
		const uint64_t limit = 64ull * n + 1024;
		stack.push_back(root);
		while (!stack.empty() && pushes < limit) {
			const uint32_t v = stack.back();
			stack.pop_back();
			for (uint32_t k = coff[v]; k < coff[v + 1]; k++) {
				const uint32_t c = cadj[k];
				d->height[c] = d->height[v] + 1;
				stack.push_back(c);
				pushes++;
			}
		}
		if (!stack.empty()) {
			povu_pvst_doc_free(d);
			set_err(err, errlen, "the children listings form a cycle or list vertices under several parents too often: comp_heights "
					     "(pvst.hpp:807-836) does not end on this file (gave up after " + std::to_string(limit) + " steps)");
			return nullptr;
		}
	}
	return d;
}

extern "C" void povu_pvst_doc_free(povu_pvst_doc *d)
{
	if (!d)
		return;
	free(d->type);
	free(d->file_id);
	free(d->a_id);
	free(d->z_id);
	free(d->parent);
	free(d->height);
	free(d->a_or);
	free(d->z_or);
	free(d->route);
	free(d);
}
