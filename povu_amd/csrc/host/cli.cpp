// cli.cpp -- the `povu` command line for the decompose path.
// Same surface as the reference for this path (app/cli/cli.cpp:14-26,236-262,324-382, app/main.cpp):
//   povu [--version] [-v <int>] [-t <int>] decompose -i <gfa> [-o <dir>] [-h|--hairpins] [-s|--subflubbles]
//   povu ... decompose ... --structure-export <json>   (additive: writes the flubble debug sidecar gfa2vcf writes)
//   povu ... gfa2vcf -i <gfa> [-h] [-s] [--structure-export <json>] <options of `call`>   (app/cli/cli.cpp:154-193)
#include "decompose.hpp"

#include <cstdlib>
#include <cxxabi.h>
#include <typeinfo>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

static const char *VERSION = "0.0.1-alpha"; // app/cli/cli.hpp:10

static void usage(std::ostream &os)
{
	os << "  povu {OPTIONS} [COMMAND]\n\n"
	      "    Explore variation in a variation graph\n\n"
	      "  OPTIONS:\n\n"
	      "      commands\n"
	      "        decompose                         Find regions of variation\n"
	      "        info                              Print graph information [uses 1 thread]\n"
	      "        prune                             Reduce GFA to graph structure\n"
	      "        gfa2vcf                           Convert GFA to VCF (decompose here + `call` of $POVU_CALL_EXE)\n"
	      "        call, vcf                         (not part of the MI355X decompose build)\n"
	      "      arguments\n"
	      "        --version                         The current version of povu\n"
	      "        -v[verbosity],\n"
	      "        --verbosity=[verbosity]           Level of output [default: 0]\n"
	      "        -t[threads], --threads=[threads]  Number of threads to use [default: 1]\n"
	      "        -h, --help                        help\n\n"
	      "  decompose OPTIONS:\n"
	      "        -i[gfa], --input-gfa=[gfa]        path to input gfa [required]\n"
	      "        -o[output_dir],\n"
	      "        --output-dir=[output_dir]         Output directory [default: .]\n"
	      "        -h, --hairpins                    Find hairpins in the variation graph [default: false]\n"
	      "        -s, --subflubbles                 Find subflubbles in the variation graph [default: false]\n"
	      "                                          (tiny, parallel, concealed, midi and smothered: T O C M S lines)\n"
	      "        --leaf-subflubbles                Relabel leaf flubbles as tiny (T) / parallel (O): the find_tiny and\n"
	      "                                          find_parallel passes of -s, without its three inserting passes\n"
	      "        --gpus=[n]                        Shard the components over n GPUs of this node, one worker per GPU\n"
	      "                                          [default: 1; devices 0..n-1 or $POVU_HIP_DEVICES]\n"
	      "        --structure-export=[structure_json]\n"
	      "                                          Write the flubble debug sidecar <structure_json>.flubble-debug.jsonl\n"
	      "                                          [conformance]\n";
}

int main(int argc, char **argv)
{
	povu_host::Config cfg;
	std::string command;
	bool version = false, help = false, have_input = false, print_tips = false;
	std::vector<std::string> call_args; // gfa2vcf: everything that belongs to `call`
	auto value = [&](int &i, const char *a, const char *shortf, const char *longf, std::string &out) -> bool {
		const size_t ls = strlen(shortf), ll = strlen(longf);
		if (!strncmp(a, longf, ll) && a[ll] == '=') {
			out = a + ll + 1;
			return true;
		}
		if (!strcmp(a, longf) || !strcmp(a, shortf)) {
			if (i + 1 >= argc) {
				std::cerr << "Flag '" << a << "' requires an argument but received none" << std::endl;
				usage(std::cerr);
				std::exit(1);
			}
			out = argv[++i];
			return true;
		}
		if (!strncmp(a, shortf, ls) && a[ls] && a[1] != '-') {
			out = a + ls;
			return true;
		}
		return false;
	};
	for (int i = 1; i < argc; i++) {
		const char *a = argv[i];
		std::string v;
		if (!strcmp(a, "--version")) {
			version = true;
		} else if (!strcmp(a, "--help") || (!strcmp(a, "-h") && command.empty())) {
			help = true;
		} else if (value(i, a, "-v", "--verbosity", v)) {
			cfg.verbosity = atoi(v.c_str());
		} else if (command == "info" && (!strcmp(a, "-t") || !strcmp(a, "--print_tips"))) {
			print_tips = true; // inside `info`, -t means "print the tips" (cli.cpp:220)
		} else if (value(i, a, "-t", "--threads", v)) {
			cfg.threads = atoi(v.c_str());
		} else if ((command == "decompose" || command == "info" || command == "prune" || command == "gfa2vcf") &&
			   value(i, a, "-i", "--input-gfa", v)) {
			cfg.input_gfa = v;
			have_input = true;
		} else if ((command == "decompose" || command == "prune") && value(i, a, "-o", "--output-dir", v)) {
			cfg.output_dir = v;
		} else if ((command == "decompose" || command == "gfa2vcf") && (!strcmp(a, "-h") || !strcmp(a, "--hairpins"))) {
			cfg.hairpins = true;
		} else if ((command == "decompose" || command == "gfa2vcf") && (!strcmp(a, "-s") || !strcmp(a, "--subflubbles"))) {
			cfg.subflubbles = true;
		} else if ((command == "decompose" || command == "gfa2vcf") && !strcmp(a, "--leaf-subflubbles")) {
			cfg.leaf_subflubbles = true;
		} else if (command == "decompose" && value(i, a, "--gpus", "--gpus", v)) {
			cfg.gpus = atoi(v.c_str());
			if (cfg.gpus < 1 || cfg.gpus > 64) {
				std::cerr << "Flag '--gpus' expects a number of GPUs between 1 and 64" << std::endl;
				return 1;
			}
		} else if ((command == "decompose" || command == "gfa2vcf") && !strncmp(a, "--structure-export", 18) &&
			   (a[18] == 0 || a[18] == '=')) {
			if (a[18] == '=') {
				cfg.structure_export = a + 19;
			} else if (i + 1 < argc) {
				cfg.structure_export = argv[++i];
			} else {
				std::cerr << "Flag '" << a << "' requires an argument but received none" << std::endl;
				usage(std::cerr);
				return 1;
			}
		} else if (command == "gfa2vcf") {
			call_args.push_back(a); // streaming / output / reference options of `call` (cli.cpp:28-88)
		} else if (command.empty() && a[0] != '-') {
			command = a;
		} else {
			if (!version) {
				std::cerr << "Flag could not be matched: " << a << std::endl;
				usage(std::cerr);
				return 1;
			}
		}
	}
	if (version) {
		std::cout << VERSION << std::endl;
		return EXIT_SUCCESS;
	}
	if (help || command.empty()) {
		usage(std::cout);
		return 0;
	}
	if (command != "decompose" && command != "info" && command != "prune" && command != "gfa2vcf") {
		std::cerr << "povu (MI355X build): only `decompose`, `info`, `prune` and `gfa2vcf` are provided; `" << command
			  << "` belongs to the reference CPU tool" << std::endl;
		return 1;
	}
	if (!have_input) {
		std::cerr << "Flag 'gfa' is required" << std::endl;
		usage(std::cerr);
		return 1;
	}
	if (const char *d = std::getenv("POVU_HIP_DEVICE"))
		cfg.device = atoi(d);
	// The reference lets exceptions escape main (uncaught -> std::terminate -> abort).  Same message, same non-zero exit,
	// but an orderly one: this process holds a GPU context, and an abort would take it down mid-flight.
	try {
		if (command == "info")
			povu_host::do_info(cfg, print_tips);
		else if (command == "prune")
			povu_host::do_prune(cfg);
		else if (command == "gfa2vcf")
			povu_host::do_gfa2vcf(cfg, call_args);
		else {
			povu_host::reset_debug_sidecar(cfg);
			cfg.exit_when_done = true;
			povu_host::do_decompose(cfg);
		}
	} catch (const std::exception &e) {
		// (the line libstdc++'s terminate handler prints, with the exception's real dynamic type; the exit code is 1 where the
		// reference's abort gives 134 -- documented in DESIGN.md section 1)
		int status = 0;
		char *name = abi::__cxa_demangle(typeid(e).name(), nullptr, nullptr, &status);
		std::cerr << "terminate called after throwing an instance of '" << (status == 0 && name ? name : typeid(e).name())
			  << "'\n  what():  " << e.what() << std::endl;
		free(name);
		return EXIT_FAILURE;
	}
	return 0;
}
