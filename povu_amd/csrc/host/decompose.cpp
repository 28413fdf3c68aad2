#include "decompose.hpp"

#include "../../../include/povu_hip.h"
#include "gfa.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <unordered_map>
#include <vector>

namespace povu_host
{
std::vector<int> multi_devices(int gpus, const char *env, int visible);
namespace
{
// INFO lines of include/povu/common/log.hpp:18-39 (no colours)
void info(const std::string &m) { std::cerr << "INFO " << m << std::endl; }

double now_ms()
{
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// one <output_dir>/<component id>.pvst per tree of `f` (mto::to_pvst::write_pvst, src/mto/to_pvst.cpp:30-109), formatting and
// writing spread over `nt` threads
void write_forest(const povu_hip_forest *f, const Config &cfg, unsigned nt, std::atomic<bool> &failed)
{
	const int ll = cfg.verbosity;
	const uint32_t n = povu_hip_forest_tree_count(f);
	nt = std::min<unsigned>(nt, std::max(1u, std::thread::hardware_concurrency()));
	nt = std::min<unsigned>(nt, std::max(1u, n));
	std::atomic<uint32_t> next{0};
	auto worker = [&]() {
		for (;;) {
			uint32_t i = next.fetch_add(1);
			if (i >= n)
				break;
			povu_hip_tree t;
			povu_hip_forest_get(f, i, &t);
			if (ll)
				info("Handling component: " + std::to_string(t.component_id));
			size_t len = 0;
			char *txt = povu_hip_forest_pvst_text(f, i, &len);
			if (!txt) {
				std::cerr << "ERR Could not serialise the PVST of component " << t.component_id << std::endl;
				failed = true;
				break;
			}
			const std::string fn = cfg.output_dir + "/" + std::to_string(t.component_id) + ".pvst";
			FILE *o = fopen(fn.c_str(), "wb");
			if (!o) {
				std::cerr << "ERR Could not open file " << fn << std::endl;
				failed = true;
				povu_hip_buffer_free(txt);
				break;
			}
			const bool short_write = fwrite(txt, 1, len, o) != len;
			if ((fclose(o) != 0) | short_write) {
				std::cerr << "ERR Could not write file " << fn << std::endl;
				failed = true;
			}
			povu_hip_buffer_free(txt);
		}
	};
	std::vector<std::thread> th;
	for (unsigned k = 1; k < nt; k++)
		th.emplace_back(worker);
	worker();
	for (auto &x : th)
		x.join();
}

// `povu decompose --gpus N` (additive): the reference's workers each own their components from graph to file
// (app/subcommand/decompose.cpp:116-157); here a worker is a GPU with a host thread (povu_hip_multi_*).  The root GPU labels
// and partitions, the shards travel over xGMI, and every worker formats and writes the <id>.pvst files of ITS components
// as soon as its own decompose is done.
struct MultiSink {
	const Config *cfg;
	unsigned threads_per_rank;
	std::atomic<bool> failed{false};
	// --structure-export: every worker renders the frames of ITS components (the state they come from lives in its context);
	// they are written in component order once all workers are done (flubbles.cpp:733: one frame per find_flubbles call)
	povu_hip_multi *engine = nullptr;
	std::mutex mu;
	std::map<uint32_t, std::string> frames; // component id -> frame
};
int multi_sink(uint32_t rank, const povu_hip_forest *f, void *user)
{
	MultiSink *s = static_cast<MultiSink *>(user);
	if (!s->cfg->structure_export.empty()) {
		try {
			povu_hip_ctx *ctx = povu_hip_multi_context(s->engine, rank);
			const uint32_t *ids = nullptr;
			uint32_t n_ids = 0;
			const uint32_t n = povu_hip_forest_tree_count(f);
			if (n && (!ctx || povu_hip_shard_component_ids(ctx, &ids, &n_ids) != 0))
				throw std::runtime_error("flubble debug sidecar: worker " + std::to_string(rank) + " holds no shard");
			std::unordered_map<uint32_t, uint32_t> local; // component id of the whole graph -> rank inside the shard
			for (uint32_t k = 0; k < n_ids; k++)
				local[ids[k]] = k;
			for (uint32_t i = 0; i < n; i++) {
				povu_hip_tree t;
				povu_hip_forest_get(f, i, &t);
				const auto it = local.find(t.component_id);
				if (it == local.end())
					throw std::runtime_error("flubble debug sidecar: component " + std::to_string(t.component_id) + " is not of worker " + std::to_string(rank));
				std::ostringstream frame;
				write_debug_sidecar_frame(frame, ctx, it->second);
				std::lock_guard<std::mutex> l(s->mu);
				s->frames[t.component_id] = frame.str();
			}
		} catch (const std::exception &e) {
			std::cerr << "[povu::decompose] " << e.what() << std::endl;
			s->failed = true;
			return 1;
		}
	}
	write_forest(f, *s->cfg, s->threads_per_rank, s->failed);
	return s->failed ? 1 : 0;
}

void do_decompose_multi(const Config &cfg)
{
	const int ll = cfg.verbosity;
	const double t0 = now_ms();
	const std::vector<int> devs = multi_devices(cfg.gpus, std::getenv("POVU_HIP_DEVICES"), povu_hip_device_count());
	char err[512] = {0};
	std::future<povu_hip_multi *> mf = std::async(std::launch::async, [&]() { // the runtime comes up while the GFA is parsed
		return povu_hip_multi_create(devs.data(), (uint32_t)devs.size(), err, sizeof err);
	});
	GfaGraph g;
	try {
		g = load_gfa(cfg.input_gfa, false, false, cfg.threads);
	} catch (...) {
		if (povu_hip_multi *m = mf.get())
			povu_hip_multi_destroy(m);
		throw;
	}
	const double t1 = now_ms();
	povu_hip_multi *m = mf.get();
	if (!m)
		throw std::runtime_error(std::string("povu_hip: ") + err);
	auto fail = [&](const char *what) {
		const std::string msg = std::string("povu_hip: ") + what;
		povu_hip_multi_destroy(m);
		throw std::runtime_error(msg);
	};
	if (povu_hip_multi_upload(m, (uint32_t)g.vid.size(), g.vid.data(), (uint32_t)g.v1.size(), g.v1.data(), g.s1.data(), g.v2.data(),
				  g.s2.data(), nullptr, err, sizeof err) != 0)
		fail(err);
	if (ll > 1)
		info("Finding components");
	if (povu_hip_multi_scatter(m, /*keep_graph=*/0, err, sizeof err) != 0)
		fail(err);
	const double t2 = now_ms();
	MultiSink sink{&cfg, (unsigned)std::max<size_t>(1, (size_t)std::max(1, cfg.threads) / devs.size())};
	sink.engine = m;
	const uint32_t flags = (cfg.hairpins ? POVU_HIP_F_HAIRPINS : 0u) | POVU_HIP_F_NO_STAGE_TIMES |
			       (cfg.leaf_subflubbles ? POVU_HIP_F_LEAF_SUBFLUBBLES : 0u) | (cfg.subflubbles ? POVU_HIP_F_SUBFLUBBLES : 0u);
	povu_hip_forest *f = povu_hip_multi_decompose(m, flags, multi_sink, &sink, err, sizeof err);
	const double t3 = now_ms();
	if (!f) {
		if (sink.failed) { // unwritable output: message already printed, exit like the reference (to_pvst.cpp:39-42)
			povu_hip_multi_destroy(m);
			std::exit(EXIT_FAILURE);
		}
		fail(err);
	}
	if (ll > 1)
		info("Found " + std::to_string(povu_hip_forest_total_components(f)) + " components");
	const uint32_t n = povu_hip_forest_tree_count(f);
	if (cfg.hairpins) // flubbles.cpp:712-717 (in component order, after the workers are done)
		for (uint32_t i = 0; i < n; i++) {
			povu_hip_tree t;
			povu_hip_forest_get(f, i, &t);
			for (uint32_t k = 0; k < t.n_hairpins; k++)
				std::cerr << "Boundary: " << t.hairpins[2 * k] << " " << t.hairpins[2 * k + 1] << std::endl;
		}
	if (!cfg.structure_export.empty()) { // the frames the workers rendered, in component order (std::map)
		const std::string sidecar = debug_sidecar_path(cfg.structure_export);
		std::ofstream out(sidecar, std::ios::app);
		if (!out.is_open())
			fail(("could not open flubble debug sidecar: " + sidecar).c_str());
		for (const auto &kv : sink.frames)
			out << kv.second;
		out.flush();
		if (!out)
			fail(("could not write flubble debug sidecar: " + sidecar).c_str());
	}
	if (std::getenv("POVU_STAGE_COST_TRACE")) {
		double ms[6] = {0};
		povu_hip_multi_times(m, ms);
		fprintf(stderr, "povu-stage-cost contract=host:gfa_parse calls=1 input_items=%zu output_items=%zu elapsed_ns=%.0f\n", g.v1.size(),
			g.vid.size(), (t1 - t0) * 1e6);
		fprintf(stderr, "povu-stage-cost contract=host:upload_partition_scatter calls=1 input_items=%zu output_items=%zu elapsed_ns=%.0f\n",
			g.v1.size(), devs.size(), (t2 - t1) * 1e6);
		fprintf(stderr, "povu-stage-cost contract=host:decompose_and_write calls=%zu input_items=%zu output_items=%u elapsed_ns=%.0f\n",
			devs.size(), g.v1.size(), n, (t3 - t2) * 1e6);
		fprintf(stderr, "povu-stage-cost contract=hip:multi transport=%s label_ns=%.0f lpt_ns=%.0f partition_ns=%.0f\n",
			povu_hip_multi_transport(m), ms[0] * 1e6, ms[1] * 1e6, ms[2] * 1e6);
		for (uint32_t r = 0; r < devs.size(); r++) {
			povu_hip_multi_rank_info ri;
			povu_hip_multi_rank(m, r, &ri);
			fprintf(stderr,
				"povu-stage-cost contract=hip:rank%u device=%d input_items=%u csr_ns=%.0f decompose_ns=%.0f write_ns=%.0f "
				"d2h_bytes=%llu xgmi_in_bytes=%llu\n",
				r, ri.device, ri.n_links, ri.csr_ms * 1e6, ri.decompose_ms * 1e6, ri.sink_ms * 1e6, (unsigned long long)ri.d2h,
				(unsigned long long)ri.peer_in);
		}
	}
	if (cfg.exit_when_done && !std::getenv("POVU_CLI_ORDERLY_EXIT")) { // every output file is closed by now (see do_decompose)
		std::cout.flush();
		std::cerr.flush();
		fflush(nullptr);
		std::_Exit(0);
	}
	povu_hip_forest_free(f);
	povu_hip_multi_destroy(m);
}
} // namespace

// which HIP device every rank of `--gpus N` runs on: POVU_HIP_DEVICES="3,1,2" names them (N entries), else 0 .. N-1; with
// fewer visible devices than ranks the call fails -- unless POVU_HIP_DEVICES repeats a device on purpose (rehearsal)
std::vector<int> multi_devices(int gpus, const char *env, int visible)
{
	if (gpus < 1 || gpus > 64)
		throw std::runtime_error("--gpus expects 1 .. 64");
	std::vector<int> d;
	if (env && *env) {
		const char *p = env;
		while (*p) {
			char *e = nullptr;
			const long v = strtol(p, &e, 10);
			if (e == p || v < 0 || v > 4096)
				throw std::runtime_error("POVU_HIP_DEVICES: expected a comma-separated list of device indices");
			d.push_back((int)v);
			p = *e == ',' ? e + 1 : e;
			if (*e && *e != ',')
				throw std::runtime_error("POVU_HIP_DEVICES: expected a comma-separated list of device indices");
		}
		if ((int)d.size() != gpus)
			throw std::runtime_error("POVU_HIP_DEVICES names " + std::to_string(d.size()) + " devices but --gpus is " + std::to_string(gpus));
	} else {
		for (int i = 0; i < gpus; i++)
			d.push_back(i);
	}
	for (int v : d)
		if (v >= visible)
			throw std::runtime_error("--gpus " + std::to_string(gpus) + ": device " + std::to_string(v) + " is not visible (" +
						 std::to_string(visible) + " HIP devices)");
	return d;
}

void do_decompose(const Config &cfg)
{
	const int ll = cfg.verbosity;
	if (cfg.gpus > 1) {
		do_decompose_multi(cfg);
		return;
	}
	const double t0 = now_ms();
	// the HIP runtime comes up (~0.1 s) while the GFA is being parsed
	char err[512] = {0}, cerr_buf[512] = {0};
	// With POVU_CLI_PREWARM=1 the same thread also reserves the device memory the graph will need (povu_hip_prewarm) as soon as
	// the loader's counting pass knows the segment and link counts.  Off by default since the parse of a whole-genome GFA
	// (0.2 s) stopped being longer than the bring-up itself: there is nothing left to hide the reservation behind, and made
	// next to 32 tokenizer threads it took longer (0.6 s) than the same memory taken by upload and decompose when they need it.
	std::promise<std::pair<size_t, size_t>> counts_p;
	std::future<std::pair<size_t, size_t>> counts_f = counts_p.get_future();
	std::future<povu_hip_ctx *> ctx_f = std::async(std::launch::async, [&]() {
		povu_hip_ctx *c = povu_hip_create(cfg.device, cerr_buf, sizeof cerr_buf);
		char perr[256];
		auto fits = [](size_t n) { return n && n < 0xFFFFFFFFull; };
		const std::pair<size_t, size_t> n = counts_f.get(); // ({0, 0}: the parse failed before it knew)
		if (c && fits(n.first) && fits(n.second + 1) && std::getenv("POVU_CLI_PREWARM"))
			(void)povu_hip_prewarm(c, (uint32_t)n.first, (uint32_t)n.second, perr, sizeof perr); // (best effort)
		return c;
	});
	// what the loader only needed while it ran (the file mapping, gigabytes of temporaries) is given back to the system on
	// a thread of its own, off the path to the first kernel
	struct Releaser {
		GfaScratch scratch;
		std::thread th;
		void start()
		{
			th = std::thread([this]() { scratch.held.reset(); });
		}
		~Releaser()
		{
			if (th.joinable())
				th.join();
		}
	} releaser;
	GfaGraph g;
	bool counted = false;
	try {
		g = load_gfa(cfg.input_gfa, false, false, cfg.threads, [&](size_t v, size_t e) {
			counted = true;
			counts_p.set_value({v, e});
		}, &releaser.scratch);
		releaser.start();
	} catch (...) {
		if (!counted)
			counts_p.set_value({0, 0});
		if (povu_hip_ctx *c = ctx_f.get())
			povu_hip_destroy(c);
		throw;
	}
	if (!counted)
		counts_p.set_value({0, 0});
	const double t1 = now_ms();
	povu_hip_ctx *ctx = ctx_f.get();
	const double t1b = now_ms(); // (the runtime and the reserved device memory: what of them the parse did not cover)
	if (!ctx)
		throw std::runtime_error(std::string("povu_hip: ") + cerr_buf);
	if (povu_hip_graph_upload(ctx, (uint32_t)g.vid.size(), g.vid.data(), (uint32_t)g.v1.size(), g.v1.data(), g.s1.data(),
				  g.v2.data(), g.s2.data(), nullptr, err, sizeof err) != 0) {
		povu_hip_destroy(ctx);
		throw std::runtime_error(std::string("povu_hip: ") + err);
	}
	if (ll > 1)
		info("Finding components");
	// per-stage HIP events only when the stage-cost lines will be printed
	povu_hip_opts opts{0, 1, (cfg.hairpins ? POVU_HIP_F_HAIRPINS : 0u) | (ll ? 0u : POVU_HIP_F_NO_STAGE_TIMES) |
				    (cfg.leaf_subflubbles ? POVU_HIP_F_LEAF_SUBFLUBBLES : 0u) | (cfg.subflubbles ? POVU_HIP_F_SUBFLUBBLES : 0u)};
	const double t2 = now_ms();
	povu_hip_forest *f = povu_hip_decompose(ctx, &opts, err, sizeof err);
	const double t3 = now_ms();
	if (!f) {
		povu_hip_destroy(ctx);
		throw std::runtime_error(std::string("povu_hip: ") + err);
	}
	if (ll > 1)
		info("Found " + std::to_string(povu_hip_forest_total_components(f)) + " components");

	const uint32_t n = povu_hip_forest_tree_count(f);
	if (cfg.hairpins) // flubbles.cpp:712-717
		for (uint32_t i = 0; i < n; i++) {
			povu_hip_tree t;
			povu_hip_forest_get(f, i, &t);
			for (uint32_t k = 0; k < t.n_hairpins; k++)
				std::cerr << "Boundary: " << t.hairpins[2 * k] << " " << t.hairpins[2 * k + 1] << std::endl;
		}
	if (!cfg.structure_export.empty()) // one frame per find_flubbles call (flubbles.cpp:733), in component order
		for (uint32_t i = 0; i < n; i++) {
			povu_hip_tree t;
			povu_hip_forest_get(f, i, &t);
			append_debug_sidecar_frame(cfg, ctx, t.component_id - 1);
		}
	// one <id>.pvst per component; formatting + writing spread over -t threads
	std::atomic<bool> failed{false};
	write_forest(f, cfg, (unsigned)std::max(1, cfg.threads), failed);
	const double t4 = now_ms();

	// per-stage cost lines, same shape as povu::stage_cost::write_report (stage_cost.cpp:59-78)
	if (std::getenv("POVU_STAGE_COST_TRACE")) {
		povu_hip_stage_time st[64];
		int k = povu_hip_last_stage_times(ctx, st, 64);
		for (int i = 0; i < k && i < 64; i++)
			fprintf(stderr, "povu-stage-cost contract=hip:%s calls=%u input_items=%zu output_items=%u elapsed_ns=%.0f\n",
				st[i].name, st[i].launches, g.vid.size() + g.v1.size(), n, st[i].ms * 1e6);
		fprintf(stderr, "povu-stage-cost contract=host:gfa_parse calls=1 input_items=%zu output_items=%zu elapsed_ns=%.0f\n",
			g.v1.size(), g.vid.size(), (t1 - t0) * 1e6);
		fprintf(stderr, "povu-stage-cost contract=host:device_wait calls=1 input_items=0 output_items=0 elapsed_ns=%.0f\n", (t1b - t1) * 1e6);
		fprintf(stderr, "povu-stage-cost contract=host:upload_csr calls=1 input_items=%zu output_items=0 elapsed_ns=%.0f\n",
			g.v1.size(), (t2 - t1b) * 1e6);
		fprintf(stderr, "povu-stage-cost contract=host:decompose_call calls=1 input_items=%zu output_items=%u elapsed_ns=%.0f\n",
			g.v1.size(), n, (t3 - t2) * 1e6);
		fprintf(stderr, "povu-stage-cost contract=host:write_pvst calls=%u input_items=%u output_items=%u elapsed_ns=%.0f\n", n,
			n, n, (t4 - t3) * 1e6);
	}
	// The files are written and closed.  Handing ~120 GB of device memory and a few gigabytes of host arrays back piece by
	// piece took 0.06 - 0.9 s on the whole-genome workload (tools/cli_phases.py); the process ends here anyway, so that is
	// left to its exit -- unless POVU_CLI_ORDERLY_EXIT asks for the orderly release (leak checkers, the stage-cost line).
	if (cfg.exit_when_done && !failed && !std::getenv("POVU_CLI_ORDERLY_EXIT")) {
		// INVARIANT: every output of the command is written AND closed above (write_forest closes each file, the sidecar frames
		// are appended and closed one by one): _Exit runs no destructor and flushes no stream of its own.  A buffered writer
		// added later must be closed before this line (the tests keep the orderly path alive with POVU_CLI_ORDERLY_EXIT).
		std::cout.flush();
		std::cerr.flush();
		fflush(nullptr);
		std::_Exit(0);
	}
	povu_hip_forest_free(f);
	povu_hip_destroy(ctx);
	if (std::getenv("POVU_STAGE_COST_TRACE"))
		fprintf(stderr, "povu-stage-cost contract=host:release_device calls=1 input_items=0 output_items=0 elapsed_ns=%.0f\n", (now_ms() - t4) * 1e6);
	if (failed)
		std::exit(EXIT_FAILURE);
}

namespace
{
povu_hip_components *components_of(const Config &cfg, povu_hip_ctx **ctx_out)
{
	GfaGraph g = load_gfa(cfg.input_gfa, false, false, cfg.threads);
	char err[512] = {0};
	povu_hip_ctx *ctx = povu_hip_create(cfg.device, err, sizeof err);
	if (!ctx)
		throw std::runtime_error(std::string("povu_hip: ") + err);
	if (povu_hip_graph_upload(ctx, (uint32_t)g.vid.size(), g.vid.data(), (uint32_t)g.v1.size(), g.v1.data(), g.s1.data(),
				  g.v2.data(), g.s2.data(), nullptr, err, sizeof err) != 0) {
		povu_hip_destroy(ctx);
		throw std::runtime_error(std::string("povu_hip: ") + err);
	}
	povu_hip_components *c = povu_hip_componetize(ctx, err, sizeof err);
	if (!c) {
		povu_hip_destroy(ctx);
		throw std::runtime_error(std::string("povu_hip: ") + err);
	}
	*ctx_out = ctx;
	return c;
}
} // namespace

void do_info(const Config &cfg, bool print_tips)
{
	povu_hip_ctx *ctx = nullptr;
	povu_hip_components *c = components_of(cfg, &ctx);
	std::cerr << "[povu::main::do_info] Component count " << c->n_components << "\n";
	for (uint32_t k = 0; k < c->n_components; k++) { // VG::summary
		const uint32_t v0 = c->vtx_off[k], v1 = c->vtx_off[k + 1];
		// tips_ is a std::set ordered by (id, then l < r), types.cpp:60-68
		std::vector<std::pair<uint32_t, uint8_t>> tips;
		for (uint32_t v = v0; v < v1; v++)
			if (c->vtx_tip[v])
				tips.emplace_back(c->vtx_id[v], c->vtx_tip[v]);
		std::sort(tips.begin(), tips.end());
		std::cout << "Bidirected Graph: " << std::endl;
		std::cout << "\t" << "vertex count: " << (v1 - v0) << std::endl;
		std::cout << "\t" << "edge count: " << (c->link_off[k + 1] - c->link_off[k]) << std::endl;
		std::cout << "\t" << "Tip count " << tips.size() << std::endl;
		if (print_tips) {
			std::cerr << "\t" << "Tips: ";
			std::cout << "\t";
			for (size_t i = 0; i < tips.size(); i++) { // operator<<(side_n_id_t): id then +/- (types.cpp:26-33,70-74)
				std::cout << tips[i].first << (tips[i].second == POVU_TIP_L ? "+" : "-");
				if (i + 1 < tips.size())
					std::cout << ", ";
			}
			std::cout << std::endl;
		}
	}
	povu_hip_components_free(c);
	povu_hip_destroy(ctx);
}

void do_prune(const Config &cfg)
{
	const int ll = cfg.verbosity;
	povu_hip_ctx *ctx = nullptr;
	if (ll > 1)
		info("Finding components");
	povu_hip_components *c = components_of(cfg, &ctx);
	if (ll > 1)
		info("Found " + std::to_string(c->n_components) + " components");
	for (uint32_t k = 0; k < c->n_components; k++) { // write_gfa, to_gfa.cpp:13-56
		const std::string fp = cfg.output_dir + "/component_" + std::to_string(k + 1) + ".gfa";
		FILE *o = fopen(fp.c_str(), "wb");
		if (!o) {
			std::cerr << "ERR Could not open file " << fp << " for writing" << std::endl;
			continue;
		}
		fputs("H\tVN:Z:1.0\n", o);
		const uint32_t v0 = c->vtx_off[k];
		for (uint32_t v = v0; v < c->vtx_off[k + 1]; v++)
			fprintf(o, "S\t%u\tA\n", c->vtx_id[v]);
		for (uint32_t e = c->link_off[k]; e < c->link_off[k + 1]; e++) {
			const uint32_t a = c->l_v1[e], b = c->l_v2[e];
			const char *ea = "+", *eb = "+"; // a self loop is always written `+ +` (to_gfa.cpp:24-25)
			if (a != b) {
				ea = c->l_s1[e] == POVU_SIDE_R ? "+" : "-";
				eb = c->l_s2[e] == POVU_SIDE_L ? "+" : "-";
			}
			fprintf(o, "L\t%u\t%s\t%u\t%s\t0M\n", c->vtx_id[v0 + a], ea, c->vtx_id[v0 + b], eb);
		}
		fclose(o);
	}
	povu_hip_components_free(c);
	povu_hip_destroy(ctx);
}

} // namespace povu_host
