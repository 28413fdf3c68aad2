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
	const std::string sidecar = debug_sidecar_path(cfg.structure_export);
	std::ofstream out(sidecar, std::ios::app);
	if (!out.is_open())
		throw std::runtime_error("could not open flubble debug sidecar: " + sidecar);
	write_debug_sidecar_frame(out, ctx, comp_rank);
	out.flush();
	if (!out)
		throw std::runtime_error("could not write flubble debug sidecar: " + sidecar);
}

void write_debug_sidecar_frame(std::ostream &out, povu_hip_ctx *ctx, uint32_t comp_rank)
{
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
}

// ------------------------------------------------------------------ gfa2vcf
// `povu gfa2vcf` = decompose into a scratch forest, then `povu call` on it (app/subcommand/gfa2vcf.cpp:18-87).  This build has
// the first half; the second is an external povu binary (POVU_CALL_EXE) run as a CHILD process on the forest written here.
// What a user can observe is kept: the scratch forest lives in /tmp/povu_gfa2vcf_XXXXXX and is gone afterwards whatever
// happened, with -v the same three progress lines go to stderr, a failure prints "<fn> Error: <what>" and exits with
// EXIT_FAILURE; the child's exit status becomes this process's.
namespace
{
const char *const GFA2VCF_FN = "[povu::subcommands::do_gfa2vcf]";

// the scratch forest directory: made on construction, removed on destruction (also on the way out through std::exit: see fail())
class ScratchForest
{
public:
	ScratchForest()
	{
		char tmpl[] = "/tmp/povu_gfa2vcf_XXXXXX";
		if (const char *d = mkdtemp(tmpl))
			path_ = d;
	}
	~ScratchForest() { drop(); }
	ScratchForest(const ScratchForest &) = delete;
	ScratchForest &operator=(const ScratchForest &) = delete;
	bool ok() const { return !path_.empty(); }
	const std::string &path() const { return path_; }
	void drop()
	{
		if (!path_.empty()) {
			std::error_code ec;
			fs::remove_all(path_, ec);
			path_.clear();
		}
	}

private:
	std::string path_;
};

[[noreturn]] void gfa2vcf_fail(ScratchForest *forest, const std::string &what)
{
	if (forest)
		forest->drop(); // (std::exit runs no destructors of automatic objects)
	std::cerr << GFA2VCF_FN << " Error: " << what << std::endl;
	std::exit(EXIT_FAILURE);
}

// runs argv[0] with the given arguments as a child process and waits for it; the child's exit status (EXIT_FAILURE when it was
// killed).  A child process, never an exec of this one: the GPU is initialised here.
int run_child(std::vector<std::string> args)
{
	std::vector<char *> argv;
	for (auto &a : args)
		argv.push_back(a.data());
	argv.push_back(nullptr);
	pid_t pid = 0;
	if (const int rc = posix_spawnp(&pid, argv[0], nullptr, nullptr, argv.data(), environ))
		throw std::runtime_error("could not start " + args[0] + ": " + strerror(rc));
	int ws = 0;
	while (waitpid(pid, &ws, 0) < 0)
		if (errno != EINTR)
			throw std::runtime_error("waitpid failed");
	return WIFEXITED(ws) ? WEXITSTATUS(ws) : EXIT_FAILURE;
}
} // namespace

void do_gfa2vcf(const Config &cfg, const std::vector<std::string> &call_args)
{
	const char *exe = std::getenv("POVU_CALL_EXE");
	if (!exe || !*exe)
		gfa2vcf_fail(nullptr, "variant calling is not part of the MI355X build; set POVU_CALL_EXE to a povu binary that "
				      "provides `call` (it is run on the forest this build writes)");
	ScratchForest forest;
	if (!forest.ok())
		gfa2vcf_fail(nullptr, "Could not create temporary directory");
	auto say = [&](const std::string &line) {
		if (cfg.verbosity > 0)
			std::cerr << GFA2VCF_FN << ' ' << line << std::endl;
	};
	say("Using temporary directory: " + forest.path());
	int status = EXIT_FAILURE;
	try {
		say("Step 1: Decomposing graph...");
		reset_debug_sidecar(cfg);
		Config into_forest = cfg;
		into_forest.output_dir = forest.path();
		do_decompose(into_forest);

		say("Step 2: Calling variants...");
		std::vector<std::string> call{exe};
		if (cfg.verbosity)
			call.insert(call.end(), {"-v", std::to_string(cfg.verbosity)});
		call.insert(call.end(), {"-t", std::to_string(cfg.threads), "call", "-i", cfg.input_gfa, "-f", forest.path()});
		if (!cfg.structure_export.empty())
			call.insert(call.end(), {"--structure-export", cfg.structure_export});
		call.insert(call.end(), call_args.begin(), call_args.end());
		status = run_child(std::move(call));
	} catch (const std::exception &e) {
		gfa2vcf_fail(&forest, e.what());
	} catch (...) {
		gfa2vcf_fail(&forest, "unknown failure");
	}
	forest.drop();
	if (status != 0)
		std::exit(status);
}

} // namespace povu_host
