// structure_export.cpp -- the flubble debug sidecar of `--structure-export` and the gfa2vcf glue.
//
// Sidecar: one JSON line ("frame") per decomposed component in <path>.flubble-debug.jsonl, field for field what
// append_debug_sidecar_frame writes (src/povu/algorithms/flubbles.cpp:108-216): the candidate stack with the tree
// edge, class and next_seen of every entry.  The reference's conformance harness reads it
// (tests/lean4_conformance/src/main.rs:1288-1440) and compares the class partition by tree_edge_id with its own
// oracle, so the ids must be the ones Tree::add_tree_edge hands out (spanning_tree.cpp:784-805).
//
// gfa2vcf: do_gfa2vcf (app/subcommand/gfa2vcf.cpp:18-87) = decompose into a temporary forest directory, then `call`
// on that directory.  Variant calling is outside this build; the second step runs the povu binary named by
// POVU_CALL_EXE as a child process over the forest this build wrote (the PVST files are the wire format between
// the two, src/mto/from_pvst.cpp:162-302).
#include "decompose.hpp"

#include "../../../include/povu_hip.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <spawn.h>
#include <sstream>
#include <stdexcept>
#include <sys/wait.h>
#include <unordered_map>
#include <vector>

extern char **environ;

namespace povu_host
{
namespace fs = std::filesystem;

std::string debug_sidecar_path(const std::string &structure_export_path) // flubbles.cpp:217-220
{
	return structure_export_path + ".flubble-debug.jsonl";
}

void reset_debug_sidecar(const Config &cfg) // flubbles.cpp:222-231
{
	if (cfg.structure_export.empty())
		return;
	std::error_code ignored;
	fs::remove(debug_sidecar_path(cfg.structure_export), ignored);
}

namespace
{
void key(std::ostream &o, const char *k) { o << '"' << k << "\":"; }
void key_str(std::ostream &o, const char *k, const char *v) { o << '"' << k << "\":\"" << v << '"'; }
const char *MISMATCH_ENTRY = "mismatch: next_seen does not match the next later stack entry with the same class";
const char *MISMATCH_ROW = "mismatch: next_seen table entry is not the next same-class stack index";
} // namespace

void append_debug_sidecar_frame(const Config &cfg, povu_hip_ctx *ctx, uint32_t comp_rank)
{
	if (cfg.structure_export.empty())
		return;
	uint32_t n_tree = 0, n_stack = 0;
	if (povu_hip_debug_tree(ctx, comp_rank, &n_tree, nullptr, nullptr, nullptr, nullptr) != 0 ||
	    povu_hip_debug_stack(ctx, comp_rank, &n_stack, nullptr, nullptr, nullptr) != 0)
		throw std::runtime_error("flubble debug sidecar: no decompose state for component " + std::to_string(comp_rank + 1));
	std::vector<uint32_t> gid(n_tree), par(n_tree), eid(n_tree), s_vtx(n_stack), s_cls(n_stack), next_seen(n_stack);
	std::vector<uint8_t> typ(n_tree);
	uint32_t n2 = 0;
	if (povu_hip_debug_tree(ctx, comp_rank, &n2, gid.data(), typ.data(), par.data(), nullptr) != 0 || n2 != n_tree ||
	    povu_hip_debug_edge_ids(ctx, comp_rank, &n2, eid.data()) != 0 || n2 != n_tree ||
	    povu_hip_debug_stack(ctx, comp_rank, &n2, s_vtx.data(), s_cls.data(), next_seen.data()) != 0 || n2 != n_stack)
		throw std::runtime_error("flubble debug sidecar: could not read the state of component " + std::to_string(comp_rank + 1) +
					 " (after a pass that redid some components sequentially the stage state is not exported)");
	for (uint32_t i = 0; i < n_stack; i++) // every entry names a black tree edge: a child below the root, inside the tree
		if (s_vtx[i] == 0 || s_vtx[i] >= n_tree || par[s_vtx[i]] >= n_tree)
			throw std::runtime_error("flubble debug sidecar: candidate stack entry " + std::to_string(i) + " of component " +
						 std::to_string(comp_rank + 1) + " lies outside its tree");
	// expected_next_seen (flubbles.cpp:94-106): the next later entry of the same class, else the entry itself
	std::vector<uint32_t> expected(n_stack);
	{
		std::unordered_map<uint32_t, uint32_t> later;
		for (uint32_t i = n_stack; i-- > 0;) {
			auto it = later.find(s_cls[i]);
			expected[i] = it == later.end() ? i : it->second;
			later[s_cls[i]] = i;
		}
	}
	const std::string sidecar = debug_sidecar_path(cfg.structure_export);
	std::ofstream out(sidecar, std::ios::app);
	if (!out.is_open())
		throw std::runtime_error("could not open flubble debug sidecar: " + sidecar);
	out << '{';
	key_str(out, "schema", "povu.flubble-debug.frame.v1");
	out << ',';
	key(out, "tree_vertex_count");
	out << n_tree << ',';
	key(out, "tree_edge_count");
	out << (n_tree ? n_tree - 1 : 0) << ',';
	key(out, "stack_entries");
	out << '[';
	for (uint32_t i = 0; i < n_stack; i++) {
		if (i)
			out << ',';
		const uint32_t child = s_vtx[i], parent = par[child];
		const uint8_t ct = typ[child] & 3, pt = typ[parent] & 3; // 0 = l, 1 = r, 2 = dummy
		const bool in_range = next_seen[i] < n_stack;
		const bool same_class = in_range && s_cls[next_seen[i]] == s_cls[i];
		out << '{';
		key(out, "order");
		out << i << ',';
		key(out, "tree_edge_index");
		out << child - 1 << ',';
		key(out, "tree_edge_id");
		out << eid[child] << ',';
		key(out, "boundary_vertex_id");
		out << gid[child] << ',';
		key_str(out, "orientation", ct == 1 ? ">" : "<"); // compute_eq_class_stack, flubbles.cpp:466-473
		out << ',';
		key_str(out, "provenance", (ct == 2 || pt == 2) ? "dummy" : "real"); // stack entries are black edges
		out << ',';
		key_str(out, "color", "black");
		out << ',';
		key(out, "class_id");
		out << s_cls[i] << ',';
		key(out, "parent_tree_vertex");
		out << parent << ',';
		key(out, "child_tree_vertex");
		out << child << ',';
		key(out, "next_seen");
		out << next_seen[i] << ',';
		key(out, "expected_next_seen");
		out << expected[i] << ',';
		key(out, "next_seen_in_range");
		out << (in_range ? "true" : "false") << ',';
		key(out, "next_seen_same_class");
		out << (same_class ? "true" : "false") << ',';
		key_str(out, "diagnostic", next_seen[i] == expected[i] ? "ok" : MISMATCH_ENTRY);
		out << '}';
	}
	out << "],\"next_seen_table\":[";
	for (uint32_t i = 0; i < n_stack; i++) {
		if (i)
			out << ',';
		out << '{';
		key(out, "stack_order");
		out << i << ',';
		key(out, "class_id");
		out << s_cls[i] << ',';
		key(out, "next_seen");
		out << next_seen[i] << ',';
		key(out, "expected_next_seen");
		out << expected[i] << ',';
		key_str(out, "diagnostic", next_seen[i] == expected[i] ? "ok" : MISMATCH_ROW);
		out << '}';
	}
	out << "]}\n";
	out.flush();
	if (!out)
		throw std::runtime_error("could not write flubble debug sidecar: " + sidecar);
}

// ------------------------------------------------------------------ gfa2vcf
void do_gfa2vcf(const Config &cfg, const std::vector<std::string> &call_args)
{
	const std::string fn_name = "[povu::subcommands::do_gfa2vcf]";
	const int ll = cfg.verbosity;
	const char *exe = std::getenv("POVU_CALL_EXE");
	if (!exe || !*exe) {
		std::cerr << fn_name
			  << " Error: variant calling is not part of the MI355X build; set POVU_CALL_EXE to a povu binary that "
			     "provides `call` (it is run on the forest this build writes)"
			  << std::endl;
		std::exit(EXIT_FAILURE);
	}
	char temp_template[] = "/tmp/povu_gfa2vcf_XXXXXX";
	char *temp_dir = mkdtemp(temp_template);
	if (temp_dir == nullptr) {
		std::cerr << fn_name << " Error: Could not create temporary directory" << std::endl;
		std::exit(EXIT_FAILURE);
	}
	const std::string temp_dir_str(temp_dir);
	if (ll > 0)
		std::cerr << fn_name << " Using temporary directory: " << temp_dir_str << std::endl;
	if (ll > 0)
		std::cerr << fn_name << " Step 1: Decomposing graph..." << std::endl;
	reset_debug_sidecar(cfg);
	int status = EXIT_FAILURE;
	try {
		Config dc = cfg;
		dc.output_dir = temp_dir_str;
		do_decompose(dc);
		if (ll > 0)
			std::cerr << fn_name << " Step 2: Calling variants..." << std::endl;
		std::vector<std::string> av{exe};
		if (cfg.verbosity) {
			av.push_back("-v");
			av.push_back(std::to_string(cfg.verbosity));
		}
		av.push_back("-t");
		av.push_back(std::to_string(cfg.threads));
		av.push_back("call");
		av.push_back("-i");
		av.push_back(cfg.input_gfa);
		av.push_back("-f");
		av.push_back(temp_dir_str);
		if (!cfg.structure_export.empty()) {
			av.push_back("--structure-export");
			av.push_back(cfg.structure_export);
		}
		for (const auto &a : call_args)
			av.push_back(a);
		std::vector<char *> argv;
		for (auto &a : av)
			argv.push_back(a.data());
		argv.push_back(nullptr);
		pid_t pid = 0;
		// a child process, never an exec of this one: the GPU is initialised here
		const int rc = posix_spawnp(&pid, exe, nullptr, nullptr, argv.data(), environ);
		if (rc != 0)
			throw std::runtime_error(std::string("could not start ") + exe + ": " + strerror(rc));
		int ws = 0;
		while (waitpid(pid, &ws, 0) < 0)
			if (errno != EINTR)
				throw std::runtime_error("waitpid failed");
		status = WIFEXITED(ws) ? WEXITSTATUS(ws) : EXIT_FAILURE;
	} catch (const std::exception &e) {
		fs::remove_all(temp_dir_str);
		std::cerr << fn_name << " Error: " << e.what() << std::endl;
		std::exit(EXIT_FAILURE);
	} catch (...) {
		fs::remove_all(temp_dir_str);
		std::cerr << fn_name << " Error: unknown failure" << std::endl;
		std::exit(EXIT_FAILURE);
	}
	fs::remove_all(temp_dir_str);
	if (status != 0)
		std::exit(status);
}

} // namespace povu_host
