// ffi.cpp -- implementation of include/povu_ffi.h over the HIP decompose path.
// Behaviour follows povu-rs/povu-ffi/povu_ffi.cpp function by function (line refs inline).
#include "../../../include/povu_ffi.h"
#include "../../../include/povu_hip.h"
#include "gfa.hpp"

#include <cstring>
#include <filesystem>
#include <thread>
#include <fstream>
#include <string>
#include <unordered_map>
#include <vector>

struct PovuGraph {
	std::vector<uint64_t> ids;
	std::vector<std::string> seqs;
	std::unordered_map<uint64_t, uint32_t> idx_of;
	std::vector<uint32_t> v1, v2;
	std::vector<uint8_t> s1, s2;
	std::vector<povu_host::GfaPath> paths;
	bool from_gfa = false; // tips are inferred by the GFA loader only (from_gfa.cpp:262-277)
	std::string refs_file;
	std::vector<std::string> ref_prefixes;
};
struct PovuForest {
	povu_hip_forest *f = nullptr;
	~PovuForest() { povu_hip_forest_free(f); }
};
struct PovuFlubbles {
	PovuForest forest;
	size_t pvst_vertices = 0;
	PovuGraph *graph = nullptr;
};
struct PovuPvstTree {
	const PovuFlubbles *owner;
};
struct PovuVcfOutput {
	std::string vcf_content;
};

static void set_error(PovuError *error, int code, const char *message) // povu_ffi.cpp:56-64
{
	if (!error)
		return;
	const char *m = message ? message : "Unknown error";
	const size_t n = std::strlen(m);
	error->code = code;
	error->message = new char[n + 1];
	std::memcpy(error->message, m, n + 1);
}

extern "C" {

PovuGraph *povu_graph_new(size_t vcap, size_t ecap, size_t)
{
	try {
		auto *g = new PovuGraph();
		g->ids.reserve(vcap);
		g->seqs.reserve(vcap);
		g->v1.reserve(ecap);
		g->v2.reserve(ecap);
		return g;
	} catch (...) {
		return nullptr;
	}
}

PovuGraph *povu_graph_from_gfa(const char *gfa_path, PovuError *error) // povu_ffi.cpp:80-118
{
	try {
		if (!gfa_path || gfa_path[0] == '\0') {
			set_error(error, 1, "GFA path must not be empty");
			return nullptr;
		}
		std::error_code ec;
		if (!std::filesystem::is_regular_file(gfa_path, ec)) {
			std::string m = std::string("GFA file does not exist: ") + gfa_path;
			if (ec)
				m += " (" + ec.message() + ")";
			set_error(error, 1, m.c_str());
			return nullptr;
		}
		const int threads = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
		povu_host::GfaGraph gg = povu_host::load_gfa(gfa_path, true, true, threads);
		auto *g = new PovuGraph();
		g->from_gfa = true;
		g->ids.assign(gg.vid.begin(), gg.vid.end());
		g->seqs = std::move(gg.seq);
		for (size_t i = 0; i < g->ids.size(); i++)
			g->idx_of.emplace(g->ids[i], (uint32_t)i);
		g->v1.assign(gg.v1.begin(), gg.v1.end());
		g->v2.assign(gg.v2.begin(), gg.v2.end());
		g->s1.assign(gg.s1.begin(), gg.s1.end());
		g->s2.assign(gg.s2.begin(), gg.s2.end());
		g->paths = std::move(gg.paths);
		return g;
	} catch (const std::exception &e) {
		set_error(error, 1, e.what());
		return nullptr;
	}
}

void povu_graph_free(PovuGraph *graph) { delete graph; }

size_t povu_graph_add_vertex(PovuGraph *graph, uint64_t id, const char *sequence) // :121-138
{
	if (!graph || !sequence)
		return (size_t)-1;
	try {
		const uint32_t idx = (uint32_t)graph->ids.size();
		graph->ids.push_back(id);
		graph->seqs.emplace_back(sequence);
		graph->idx_of[id] = idx;
		return idx;
	} catch (...) {
		return (size_t)-1;
	}
}

size_t povu_graph_add_edge(PovuGraph *graph, uint64_t from_id, PovuOrientation fo, uint64_t to_id, PovuOrientation to)
{ // :140-163, FORWARD -> l, REVERSE -> r
	if (!graph)
		return (size_t)-1;
	try {
		auto a = graph->idx_of.find(from_id), b = graph->idx_of.find(to_id);
		if (a == graph->idx_of.end() || b == graph->idx_of.end())
			return (size_t)-1; // TwoWayMap::get_value throws in the reference -> caught -> -1
		graph->v1.push_back(a->second);
		graph->s1.push_back(fo == POVU_ORIENTATION_FORWARD ? 0 : 1);
		graph->v2.push_back(b->second);
		graph->s2.push_back(to == POVU_ORIENTATION_FORWARD ? 0 : 1);
		return graph->v1.size() - 1;
	} catch (...) {
		return (size_t)-1;
	}
}

bool povu_graph_add_path(PovuGraph *, const char *, const PovuStep *, size_t) { return false; } // :165-181
void povu_graph_finalize(PovuGraph *graph)
{
	if (!graph)
		return;
	graph->ids.shrink_to_fit();
	graph->v1.shrink_to_fit();
	graph->v2.shrink_to_fit();
}

size_t povu_graph_vertex_count(const PovuGraph *g) { return g ? g->ids.size() : 0; }
size_t povu_graph_edge_count(const PovuGraph *g) { return g ? g->v1.size() : 0; }
size_t povu_graph_path_count(const PovuGraph *g) { return g ? g->paths.size() : 0; }

PovuVertex *povu_graph_get_vertices(const PovuGraph *g, size_t *count) // :210-236
{
	if (!g || !count) {
		if (count)
			*count = 0;
		return nullptr;
	}
	const size_t n = g->ids.size();
	*count = n;
	PovuVertex *out = new PovuVertex[n];
	for (size_t i = 0; i < n; i++) {
		const std::string &s = g->seqs[i];
		char *c = new char[s.size() + 1];
		std::memcpy(c, s.c_str(), s.size() + 1);
		out[i] = PovuVertex{g->ids[i], c, s.size()};
	}
	return out;
}

PovuEdge *povu_graph_get_edges(const PovuGraph *g, size_t *count) // :238-258, ends cast straight back
{
	if (!g || !count) {
		if (count)
			*count = 0;
		return nullptr;
	}
	const size_t n = g->v1.size();
	*count = n;
	PovuEdge *out = new PovuEdge[n];
	for (size_t i = 0; i < n; i++)
		out[i] = PovuEdge{g->ids[g->v1[i]], (PovuOrientation)g->s1[i], g->ids[g->v2[i]], (PovuOrientation)g->s2[i]};
	return out;
}

PovuPath *povu_graph_get_paths(const PovuGraph *g, size_t *count) // :260-307
{
	if (!g || !count) {
		if (count)
			*count = 0;
		return nullptr;
	}
	const size_t n = g->paths.size();
	*count = n;
	if (n == 0)
		return nullptr;
	PovuPath *out = new PovuPath[n];
	for (size_t i = 0; i < n; i++) {
		const auto &p = g->paths[i];
		char *nm = new char[p.name.size() + 1];
		std::memcpy(nm, p.name.c_str(), p.name.size() + 1);
		out[i].name = nm;
		out[i].name_len = p.name.size();
		out[i].steps_count = p.step_ids.size();
		out[i].steps = p.step_ids.empty() ? nullptr : new PovuStep[p.step_ids.size()];
		for (size_t k = 0; k < p.step_ids.size(); k++)
			out[i].steps[k] = PovuStep{p.step_ids[k], (PovuOrientation)p.step_rev[k]};
	}
	return out;
}

void povu_vertices_free(PovuVertex *v, size_t count)
{
	if (!v)
		return;
	for (size_t i = 0; i < count; i++)
		delete[] v[i].sequence;
	delete[] v;
}
void povu_edges_free(PovuEdge *e, size_t) { delete[] e; }
void povu_paths_free(PovuPath *p, size_t count)
{
	if (!p)
		return;
	for (size_t i = 0; i < count; i++) {
		delete[] p[i].name;
		delete[] p[i].steps;
	}
	delete[] p;
}

bool povu_graph_set_references_from_file(PovuGraph *g, const char *ref_file, PovuError *error) // :339-353
{
	if (!g || !ref_file) {
		set_error(error, 1, "Invalid arguments");
		return false;
	}
	g->refs_file = ref_file;
	return true;
}
bool povu_graph_set_references_from_prefixes(PovuGraph *g, const char **prefixes, size_t count, PovuError *error)
{ // :355-374
	if (!g || !prefixes) {
		set_error(error, 1, "Invalid arguments");
		return false;
	}
	g->ref_prefixes.clear();
	for (size_t i = 0; i < count; i++)
		g->ref_prefixes.emplace_back(prefixes[i]);
	return true;
}

static povu_hip_forest *run_decompose(PovuGraph *g, int device, int hairpins, PovuError *error)
{
	if (g->ids.empty()) {
		set_error(error, 1, "graph has no vertices");
		return nullptr;
	}
	for (uint64_t id : g->ids)
		if (id > 0xFFFFFFFEull) {
			set_error(error, 1, "segment id does not fit 32 bits (pt::id_t is u32, core.hpp:20-21)");
			return nullptr;
		}
	if (g->ids.size() >= 0xFFFFFFFFull || g->v1.size() >= 0xFFFFFFFFull) { // the C ABI below takes 32-bit counts
		set_error(error, 1, "graph too large: vertex and edge counts must fit 32 bits");
		return nullptr;
	}
	char err[512] = {0};
	povu_hip_ctx *ctx = povu_hip_create(device, err, sizeof err);
	if (!ctx) {
		set_error(error, 1, err);
		return nullptr;
	}
	std::vector<uint32_t> vid(g->ids.begin(), g->ids.end());
	std::vector<uint8_t> no_tips;
	const uint8_t *tips = nullptr;
	if (!g->from_gfa) { // builder graphs carry no tips (nobody calls add_tip, povu_ffi.cpp:121-195)
		no_tips.assign(g->ids.size(), 0);
		tips = no_tips.data();
	}
	povu_hip_forest *f = nullptr;
	if (povu_hip_graph_upload(ctx, (uint32_t)vid.size(), vid.data(), (uint32_t)g->v1.size(), g->v1.data(), g->s1.data(),
				  g->v2.data(), g->s2.data(), tips, err, sizeof err) == 0) {
		povu_hip_opts o{0, 1, (hairpins ? POVU_HIP_F_HAIRPINS : 0u) | POVU_HIP_F_NO_STAGE_TIMES}; // (no per-stage timers: the pass starts kernels ahead of its host reads)
		f = povu_hip_decompose(ctx, &o, err, sizeof err);
	}
	povu_hip_destroy(ctx);
	if (!f)
		set_error(error, 1, err);
	return f;
}

PovuFlubbles *povu_graph_find_flubbles(PovuGraph *graph, PovuError *error)
{
	if (!graph) {
		set_error(error, 1, "Invalid graph"); // :370-373
		return nullptr;
	}
	try {
		povu_hip_forest *f = run_decompose(graph, 0, 0, error);
		if (!f)
			return nullptr;
		auto *fl = new PovuFlubbles();
		fl->forest.f = f;
		fl->graph = graph;
		size_t flub = 0;
		for (uint32_t i = 0; i < povu_hip_forest_tree_count(f); i++) {
			povu_hip_tree t;
			povu_hip_forest_get(f, i, &t);
			flub += t.n_pvst - 1;
		}
		fl->pvst_vertices = flub + 1;
		return fl;
	} catch (const std::exception &e) {
		set_error(error, 1, e.what());
		return nullptr;
	}
}
void povu_flubbles_free(PovuFlubbles *f) { delete f; }
size_t povu_flubbles_count(const PovuFlubbles *f) { return f ? f->pvst_vertices : 0; }
PovuFlubble *povu_flubbles_get(const PovuFlubbles *, size_t) { return nullptr; }
void povu_flubble_free(PovuFlubble *fl) // :413-424
{
	if (!fl)
		return;
	if (fl->walks) {
		for (size_t i = 0; i < fl->walks_count; i++)
			delete[] fl->walks[i];
		delete[] fl->walks;
	}
	delete[] fl->walk_lengths;
	delete fl;
}

PovuPvstTree *povu_flubbles_get_pvst_tree(const PovuFlubbles *f) { return f ? new PovuPvstTree{f} : nullptr; }
void povu_pvst_tree_free(PovuPvstTree *t) { delete t; }
size_t povu_pvst_tree_vertex_count(const PovuPvstTree *t) { return t ? t->owner->pvst_vertices : 0; }

PovuVcfOutput *povu_flubbles_call_variants(PovuFlubbles *f, PovuError *error) // :440-462
{
	if (!f || !f->graph) {
		set_error(error, 1, "Invalid flubbles or graph");
		return nullptr;
	}
	set_error(error, 1, "VCF generation not yet implemented in FFI layer");
	return nullptr;
}
bool povu_vcf_write_to_file(const PovuVcfOutput *vcf, const char *path, PovuError *error) // :464-482
{
	if (!vcf || !path) {
		set_error(error, 1, "Invalid arguments");
		return false;
	}
	std::ofstream out(path);
	if (!out) {
		set_error(error, 1, "Failed to open output file");
		return false;
	}
	out << vcf->vcf_content;
	return true;
}
char *povu_vcf_to_string(const PovuVcfOutput *vcf, size_t *length) // :484-494
{
	if (!vcf || !length) {
		if (length)
			*length = 0;
		return nullptr;
	}
	*length = vcf->vcf_content.size();
	char *r = new char[*length + 1];
	std::memcpy(r, vcf->vcf_content.c_str(), *length + 1);
	return r;
}
void povu_vcf_free(PovuVcfOutput *v) { delete v; }
void povu_string_free(char *s) { delete[] s; }
bool povu_gfa_to_vcf(const char *, const char *, const char *, PovuError *error) // :506-530
{
	set_error(error, 1, "gfa_to_vcf not yet implemented in FFI layer");
	return false;
}
void povu_error_free(PovuError *error)
{
	if (error && error->message) {
		delete[] error->message;
		error->message = nullptr;
	}
}

// ---- additive
PovuForest *povu_graph_decompose(PovuGraph *graph, int device, int hairpins, PovuError *error)
{
	if (!graph) {
		set_error(error, 1, "Invalid graph");
		return nullptr;
	}
	try {
		povu_hip_forest *f = run_decompose(graph, device, hairpins, error);
		if (!f)
			return nullptr;
		auto *out = new PovuForest();
		out->f = f;
		return out;
	} catch (const std::exception &e) {
		set_error(error, 1, e.what());
		return nullptr;
	}
}
size_t povu_forest_tree_count(const PovuForest *f) { return f ? povu_hip_forest_tree_count(f->f) : 0; }
size_t povu_forest_component_count(const PovuForest *f) { return f ? povu_hip_forest_total_components(f->f) : 0; }
uint32_t povu_forest_component_id(const PovuForest *f, size_t i)
{
	povu_hip_tree t;
	if (!f || povu_hip_forest_get(f->f, (uint32_t)i, &t) != 0)
		return 0;
	return t.component_id;
}
size_t povu_forest_pvst_vertex_count(const PovuForest *f, size_t i)
{
	povu_hip_tree t;
	if (!f || povu_hip_forest_get(f->f, (uint32_t)i, &t) != 0)
		return 0;
	return t.n_pvst;
}
char *povu_forest_pvst_text(const PovuForest *f, size_t i, size_t *length)
{
	if (!f)
		return nullptr;
	size_t n = 0;
	char *t = povu_hip_forest_pvst_text(f->f, (uint32_t)i, &n);
	if (!t)
		return nullptr;
	char *r = new char[n + 1]; // handed out under the povu_string_free (delete[]) rule
	std::memcpy(r, t, n + 1);
	povu_hip_buffer_free(t);
	if (length)
		*length = n;
	return r;
}
void povu_forest_free(PovuForest *f) { delete f; }

} // extern "C"
