#include "decompose.hpp"

#include "../../../include/povu_hip.h"
#include "gfa.hpp"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <thread>
#include <vector>

namespace povu_host
{
namespace
{
// INFO lines of include/povu/common/log.hpp:18-39 (no colours)
void info(const std::string &m) { std::cerr << "INFO " << m << std::endl; }

double now_ms()
{
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
} // namespace

void do_decompose(const Config &cfg)
{
	const int ll = cfg.verbosity;
	if (cfg.subflubbles)
		throw std::runtime_error("-s/--subflubbles is not part of the MI355X decompose path yet (flubbles only)");
	const double t0 = now_ms();
	GfaGraph g = load_gfa(cfg.input_gfa);
	const double t1 = now_ms();

	char err[512] = {0};
	povu_hip_ctx *ctx = povu_hip_create(cfg.device, err, sizeof err);
	if (!ctx)
		throw std::runtime_error(std::string("povu_hip: ") + err);
	if (povu_hip_graph_upload(ctx, (uint32_t)g.vid.size(), g.vid.data(), (uint32_t)g.v1.size(), g.v1.data(), g.s1.data(),
				  g.v2.data(), g.s2.data(), nullptr, err, sizeof err) != 0) {
		povu_hip_destroy(ctx);
		throw std::runtime_error(std::string("povu_hip: ") + err);
	}
	if (ll > 1)
		info("Finding components");
	povu_hip_opts opts{0, 1, cfg.hairpins ? POVU_HIP_F_HAIRPINS : 0u};
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
	// one <id>.pvst per component; formatting + writing spread over -t threads
	unsigned nt = (unsigned)std::max(1, cfg.threads);
	nt = std::min<unsigned>(nt, std::max(1u, std::thread::hardware_concurrency()));
	nt = std::min<unsigned>(nt, std::max(1u, n));
	std::atomic<uint32_t> next{0};
	std::atomic<bool> failed{false};
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
			const std::string fn = cfg.output_dir + "/" + std::to_string(t.component_id) + ".pvst";
			FILE *o = fopen(fn.c_str(), "wb");
			if (!o) {
				std::cerr << "ERR Could not open file " << fn << std::endl;
				failed = true;
				povu_hip_buffer_free(txt);
				break;
			}
			fwrite(txt, 1, len, o);
			fclose(o);
			povu_hip_buffer_free(txt);
		}
	};
	std::vector<std::thread> th;
	for (unsigned k = 1; k < nt; k++)
		th.emplace_back(worker);
	worker();
	for (auto &x : th)
		x.join();
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
		fprintf(stderr, "povu-stage-cost contract=host:upload_csr calls=1 input_items=%zu output_items=0 elapsed_ns=%.0f\n",
			g.v1.size(), (t2 - t1) * 1e6);
		fprintf(stderr, "povu-stage-cost contract=host:decompose_call calls=1 input_items=%zu output_items=%u elapsed_ns=%.0f\n",
			g.v1.size(), n, (t3 - t2) * 1e6);
		fprintf(stderr, "povu-stage-cost contract=host:write_pvst calls=%u input_items=%u output_items=%u elapsed_ns=%.0f\n", n,
			n, n, (t4 - t3) * 1e6);
	}
	povu_hip_forest_free(f);
	povu_hip_destroy(ctx);
	if (failed)
		std::exit(EXIT_FAILURE);
}

} // namespace povu_host
