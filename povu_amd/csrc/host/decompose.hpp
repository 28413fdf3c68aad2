// decompose.hpp -- host orchestration of `povu decompose` on the HIP path.
// Mirrors povu::subcommands::decompose::do_decompose (app/subcommand/decompose.cpp:94-160) and the
// decompose-relevant part of core::config (include/povu/common/app.hpp:129-182).
#pragma once
#include <iosfwd>
#include <cstdint>
#include <string>
#include <vector>

struct povu_hip_ctx;

namespace povu_host
{

struct Config {
	std::string input_gfa;
	std::string output_dir = "."; // app.hpp default
	int verbosity = 0;
	int threads = 1;
	bool hairpins = false;
	bool subflubbles = false;
	bool leaf_subflubbles = false; // --leaf-subflubbles: find_tiny + find_parallel only (the two passes of -s that relabel leaf flubbles)
	int device = 0;
	int gpus = 1; // --gpus N (additive): components sharded over N GPUs of this node, one worker thread per GPU
	// --structure-export <path>: also write <path>.flubble-debug.jsonl (one frame per decomposed component)
	std::string structure_export;
	// the decompose command itself: once the files are written the process ends without releasing device and host memory
	// piece by piece (do_decompose does not return then); callers that go on afterwards (gfa2vcf) leave it off
	bool exit_when_done = false;
};

// Loads the GFA, decomposes it on the GPU and writes <output_dir>/<component id>.pvst.
// Invalid GFA -> std::runtime_error (uncaught in the reference too, from_gfa.cpp:28-32);
// unwritable output -> message + exit(EXIT_FAILURE) (to_pvst.cpp:39-42).
void do_decompose(const Config &cfg);

// `povu info` (app/subcommand/info.cpp:19-45 + VG::summary, bidirected.cpp:368-380) and `povu prune`
// (app/subcommand/prune.cpp:17-41 + mto::to_gfa::write_gfa, src/mto/to_gfa.cpp:13-56): both only need
// the components of row B.
void do_info(const Config &cfg, bool print_tips);

// The flubble debug sidecar of --structure-export (src/povu/algorithms/flubbles.cpp:108-231): path, reset, and one
// frame for the component of 0-based rank `comp_rank` from the device state of the last povu_hip_decompose.
std::string debug_sidecar_path(const std::string &structure_export_path);
void reset_debug_sidecar(const Config &cfg);
void append_debug_sidecar_frame(const Config &cfg, povu_hip_ctx *ctx, uint32_t comp_rank);
// the frame itself (one line of JSON), of component `comp_rank` of the graph or shard the context holds
void write_debug_sidecar_frame(std::ostream &out, povu_hip_ctx *ctx, uint32_t comp_rank);

// `povu gfa2vcf` (app/subcommand/gfa2vcf.cpp:18-87): decompose into a temporary forest, then run `call` of the
// povu binary named by POVU_CALL_EXE on it (a child process); `call_args` are handed to it unchanged.
void do_gfa2vcf(const Config &cfg, const std::vector<std::string> &call_args);
void do_prune(const Config &cfg);

// device of every rank of `--gpus N` (POVU_HIP_DEVICES or 0 .. N-1), checked against the number of visible devices
std::vector<int> multi_devices(int gpus, const char *env, int visible);

} // namespace povu_host
