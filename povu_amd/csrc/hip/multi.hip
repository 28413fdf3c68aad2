// multi.hip -- one process, N GPUs: the engine behind `povu decompose --gpus N` and `bench.py --gpus N` without a
// launcher (povu_hip_multi_*, include/povu_hip.h).
//
// The reference spreads the components of a graph over worker threads that each own theirs from graph to file
// (app/subcommand/decompose.cpp:116-157).  Here a worker is a GPU with a host thread: the root device labels the components
// (row B's union-find kernels), bin-packs them over the devices (LPT) and partitions vertices and links on the device
// (povu_hip_shard_partition); the packed shards travel over xGMI -- RCCL ncclSend / ncclRecv, one communicator per device,
// the root's sends in one group, every worker posting its own receive --; every device builds the CSR of its shard and
// decomposes it.  There is NO gather step: a rank's decompose already copies its PVST block into page-locked host memory over
// its own GPU's PCIe link, and in one address space the merged forest simply takes those blocks over (adopt_forest).  RCCL
// carries the scatter only; no collective runs inside the traversal.
#include "context.hpp"
#include "rccl_api.hpp"

#include <chrono>
#include <condition_variable>
#include <functional>
#include <set>
#include <thread>

void adopt_forest(povu_hip_forest &out, povu_hip_forest &m); // shard.hip

namespace
{
double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

enum Transport { T_NONE, T_RCCL, T_PEER, T_SAME };

// a host thread bound to one device; runs one job at a time
struct Worker {
	uint32_t rank = 0;
	int device = 0;
	povu_hip_ctx *ctx = nullptr;
	std::thread th;
	std::mutex mu;
	std::condition_variable cv;
	std::function<void()> job;
	bool has_job = false, stop = false, done = true;
	std::string error;
	// RCCL
	ncclComm_t comm = nullptr;
	hipStream_t xstream = nullptr; // this device's transfer stream (one communicator, one stream)
	hipEvent_t arrived = nullptr;  // peer-copy transport: the root records it behind this rank's copy
	// last scatter / decompose
	const void *shard_ptr = nullptr; // where this rank's packed shard is (its own device)
	size_t shard_bytes = 0;
	uint32_t nv = 0, ne = 0, nc = 0;
	double recv_ms = 0, csr_ms = 0, dec_ms = 0, sink_ms = 0;
	povu_hip_forest *forest = nullptr;

	void loop()
	{
		(void)hipSetDevice(device);
		for (;;) {
			std::function<void()> j;
			{
				std::unique_lock<std::mutex> l(mu);
				cv.wait(l, [&] { return has_job || stop; });
				if (stop)
					return;
				j = std::move(job);
				has_job = false;
			}
			try {
				j();
			} catch (const std::exception &e) {
				error = e.what();
			} catch (...) {
				error = "unknown error";
			}
			{
				std::lock_guard<std::mutex> l(mu);
				done = true;
			}
			cv.notify_all();
		}
	}
	void start(std::function<void()> j)
	{
		{
			std::lock_guard<std::mutex> l(mu);
			error.clear();
			job = std::move(j);
			has_job = true;
			done = false;
		}
		cv.notify_all();
	}
	void wait()
	{
		std::unique_lock<std::mutex> l(mu);
		cv.wait(l, [&] { return done; });
	}
};
} // namespace

struct povu_hip_multi {
	uint32_t world = 0;
	std::vector<std::unique_ptr<Worker>> w;
	povu_hip_ctx *full = nullptr; // the whole graph, on the root's device
	povu_hip_shards *shards = nullptr;
	Transport transport = T_NONE;
	std::string transport_name = "none";
	double ms[6] = {0, 0, 0, 0, 0, 0};
	bool have_shards = false;

	// ncclCommAbort on every communicator (from any thread, once): posted sends / receives end with an error instead of
	// waiting for a partner that will never come
	std::mutex abort_mu;
	void abort_comms()
	{
		std::lock_guard<std::mutex> l(abort_mu);
		try {
			povu_hip::Rccl &R = povu_hip::rccl();
			if (!R.CommAbort)
				return;
			for (auto &x : w)
				if (x && x->comm) {
					(void)R.CommAbort(x->comm);
					x->comm = nullptr;
				}
		} catch (...) {
		}
	}
	// runs fn(rank) on every worker's thread, returns the first error
	std::string on_all(const std::function<void(uint32_t)> &fn)
	{
		for (auto &x : w) {
			const uint32_t r = x->rank;
			x->start([&fn, r] { fn(r); });
		}
		std::string err;
		for (auto &x : w) {
			x->wait();
			if (err.empty() && !x->error.empty())
				err = "rank " + std::to_string(x->rank) + ": " + x->error;
		}
		return err;
	}
};

extern "C" void povu_hip_multi_destroy(povu_hip_multi *m)
{
	if (!m)
		return;
	for (auto &x : m->w) {
		if (!x)
			continue;
		if (x->th.joinable()) {
			{
				std::lock_guard<std::mutex> l(x->mu);
				x->stop = true;
			}
			x->cv.notify_all();
			x->th.join();
		}
		if (x->forest)
			povu_hip_forest_free(x->forest);
		(void)hipSetDevice(x->device);
		if (x->comm) {
			try {
				(void)povu_hip::rccl().CommDestroy(x->comm);
			} catch (...) {
			}
		}
		if (x->xstream)
			(void)hipStreamDestroy(x->xstream);
		if (x->arrived)
			(void)hipEventDestroy(x->arrived);
		if (x->ctx)
			povu_hip_destroy(x->ctx);
	}
	if (m->shards)
		povu_hip_shards_free(m->shards);
	if (m->full)
		povu_hip_destroy(m->full);
	delete m;
}

extern "C" povu_hip_multi *povu_hip_multi_create(const int *devices, uint32_t n, char *err, size_t errlen)
{
	std::unique_ptr<povu_hip_multi> m;
	try {
		if (!devices || n == 0 || n > 64)
			throw HipError("multi: bad device list");
		m = std::make_unique<povu_hip_multi>();
		m->world = n;
		char e2[512] = {0};
		m->full = povu_hip_create(devices[0], e2, sizeof e2);
		if (!m->full)
			throw HipError(e2);
		std::set<int> distinct(devices, devices + n);
		for (uint32_t r = 0; r < n; r++) {
			auto x = std::make_unique<Worker>();
			x->rank = r;
			x->device = devices[r];
			x->ctx = povu_hip_create(devices[r], e2, sizeof e2);
			if (!x->ctx)
				throw HipError(e2);
			HIP_CHECK(hipSetDevice(devices[r]));
			HIP_CHECK(hipStreamCreateWithFlags(&x->xstream, hipStreamNonBlocking));
			// `arrived` is only ever recorded on the ROOT's transfer stream (peer-copy transport): an event and the stream it is
			// recorded on must belong to one device (hipErrorInvalidHandle otherwise); any thread may wait for it
			HIP_CHECK(hipSetDevice(devices[0]));
			HIP_CHECK(hipEventCreateWithFlags(&x->arrived, hipEventDisableTiming));
			HIP_CHECK(hipSetDevice(devices[r]));
			m->w.push_back(std::move(x));
		}
		// transport of the scatter
		const char *want = getenv("POVU_HIP_MULTI_TRANSPORT"); // "rccl" | "peer" (tests, A/B)
		if (n == 1) {
			m->transport = T_NONE, m->transport_name = "none";
		} else if (distinct.size() != n) {
			m->transport = T_SAME, m->transport_name = "same-device";
		} else {
			m->transport = T_PEER, m->transport_name = "peer-copy";
			if (!want || strcmp(want, "peer") != 0) {
				// one communicator per device, all created in one call (a single process owns every rank)
				try {
					std::vector<ncclComm_t> comms(n, nullptr);
					NCCL_CHECK(povu_hip::rccl().CommInitAll(comms.data(), (int)n, devices));
					for (uint32_t r = 0; r < n; r++)
						m->w[r]->comm = comms[r];
					m->transport = T_RCCL, m->transport_name = "rccl";
				} catch (const std::exception &e) {
					// the shards can still travel: device-to-device copies over the same links.  Said aloud, never silent.
					m->transport_name = std::string("peer-copy (RCCL unavailable: ") + e.what() + ")";
					if (want && !strcmp(want, "rccl"))
						throw;
				}
			}
			if (m->transport == T_PEER)
				for (uint32_t r = 1; r < n; r++) { // (best effort: without peer access the copies are staged by the runtime)
					HIP_CHECK(hipSetDevice(devices[0]));
					(void)hipDeviceEnablePeerAccess(devices[r], 0);
					HIP_CHECK(hipSetDevice(devices[r]));
					(void)hipDeviceEnablePeerAccess(devices[0], 0);
					(void)hipGetLastError();
				}
		}
		for (auto &x : m->w) {
			Worker *wp = x.get();
			x->th = std::thread([wp] { wp->loop(); });
		}
		return m.release();
	} catch (const std::exception &e) {
		set_err(err, errlen, e.what());
		if (m)
			povu_hip_multi_destroy(m.release());
		return nullptr;
	}
}

extern "C" uint32_t povu_hip_multi_world(const povu_hip_multi *m) { return m ? m->world : 0; }
extern "C" povu_hip_ctx *povu_hip_multi_context(povu_hip_multi *m, uint32_t rank) { return (m && rank < m->world) ? m->w[rank]->ctx : nullptr; }
extern "C" const char *povu_hip_multi_transport(const povu_hip_multi *m) { return m ? m->transport_name.c_str() : ""; }

extern "C" int povu_hip_multi_upload(povu_hip_multi *m, uint32_t n_vtx, const uint32_t *vid, uint32_t n_links, const uint32_t *v1,
				     const uint8_t *s1, const uint32_t *v2, const uint8_t *s2, const uint8_t *tips, char *err, size_t errlen)
{
	if (!m) {
		set_err(err, errlen, "multi: null handle");
		return 1;
	}
	m->have_shards = false;
	return povu_hip_graph_upload(m->full, n_vtx, vid, n_links, v1, s1, v2, s2, tips, err, errlen);
}

extern "C" int povu_hip_multi_scatter(povu_hip_multi *m, int keep_graph, char *err, size_t errlen)
{
	try {
		if (!m)
			throw HipError("multi: null handle");
		const double t0 = now_ms();
		char e2[512] = {0};
		m->have_shards = false;
		if (m->shards) {
			povu_hip_shards_free(m->shards);
			m->shards = nullptr;
		}
		m->shards = povu_hip_shard_partition(m->full, m->world, e2, sizeof e2);
		if (!m->shards)
			throw HipError(e2);
		povu_hip_shards_times(m->shards, m->ms);
		if (!keep_graph) {
			free_resident_graph(m->full->g);
			m->full->graph_arena.release();
			m->full->ws.release();
			m->full->upload_tmp.release();
		}
		std::vector<povu_hip_shard_info> info(m->world);
		for (uint32_t r = 0; r < m->world; r++) {
			povu_hip_shards_get(m->shards, r, &info[r]);
			Worker &x = *m->w[r];
			x.nv = info[r].n_vtx, x.ne = info[r].n_links, x.nc = info[r].n_components;
			x.shard_bytes = info[r].bytes;
			x.shard_ptr = nullptr;
		}
		Worker &root = *m->w[0];
		const Transport tr = m->transport;
		if (tr == T_PEER || tr == T_RCCL) {
			// Receivers take their buffers BEFORE any transfer is queued -- the step that can fail (device memory).  With RCCL a
			// rank that threw before posting its receive would leave the root's grouped sends waiting for ever: nothing is
			// sent unless every rank is ready (the status exchange of the multi-process path, shard.hip, in one process).
			const std::string e = m->on_all([&](uint32_t r) {
				Worker &x = *m->w[r];
				if (r == 0)
					return;
				HIP_CHECK(hipSetDevice(x.device));
				x.ctx->shard_buf.reserve(x.shard_bytes + 256);
				x.shard_ptr = x.ctx->shard_buf.take<char>(x.shard_bytes);
			});
			if (!e.empty())
				throw HipError(e);
		}
		if (tr == T_PEER) {
			HIP_CHECK(hipSetDevice(root.device));
			for (uint32_t r = 1; r < m->world; r++) {
				Worker &x = *m->w[r];
				HIP_CHECK(hipMemcpyPeerAsync(const_cast<void *>(x.shard_ptr), x.device, info[r].device_ptr, root.device, x.shard_bytes,
							     root.xstream));
				HIP_CHECK(hipEventRecord(x.arrived, root.xstream));
				root.ctx->xfer_peer_out += x.shard_bytes;
				x.ctx->xfer_peer_in += x.shard_bytes;
			}
		}
		// every rank: get the shard (the root's sends and its own CSR build run side by side), build the CSR
		const std::string e = m->on_all([&](uint32_t r) {
			Worker &x = *m->w[r];
			HIP_CHECK(hipSetDevice(x.device));
			const double t1 = now_ms();
			const void *src = nullptr;
			if (r == 0 || tr == T_SAME || tr == T_NONE) {
				src = info[r].device_ptr; // the partition block is on this device: loaded where it is
				if (r == 0 && tr == T_RCCL) {
					povu_hip::Rccl &R = povu_hip::rccl();
					NCCL_CHECK(R.GroupStart());
					bool ok = true;
					std::string why;
					for (uint32_t q = 1; q < m->world && ok; q++) {
						const ncclResult_t rc = R.Send(info[q].device_ptr, info[q].bytes, ncclChar, (int)q, x.comm, x.xstream);
						if (rc != ncclSuccess)
							ok = false, why = R.GetErrorString(rc);
						else
							x.ctx->xfer_peer_out += info[q].bytes;
					}
					const ncclResult_t rc = R.GroupEnd(); // (closed on every path)
					if (!ok || rc != ncclSuccess)
						throw HipError("scatter: ncclSend failed: " + (ok ? std::string(R.GetErrorString(rc)) : why));
				}
			} else if (tr == T_RCCL) {
				char *buf = const_cast<char *>(static_cast<const char *>(x.shard_ptr)); // (taken above, before anything was sent)
				try {
					NCCL_CHECK(povu_hip::rccl().Recv(buf, x.shard_bytes, ncclChar, 0, x.comm, x.xstream));
					HIP_CHECK(hipStreamSynchronize(x.xstream));
				} catch (...) {
					m->abort_comms(); // the root's send to this rank would wait for ever: let every stream drain, then report
					throw;
				}
				x.ctx->xfer_peer_in += x.shard_bytes;
				src = buf;
			} else { // T_PEER: wait for the root's copy
				HIP_CHECK(hipEventSynchronize(x.arrived));
				src = x.shard_ptr;
			}
			x.recv_ms = now_ms() - t1;
			char e3[512] = {0};
			const double t2 = now_ms();
			if (povu_hip_graph_upload_shard(x.ctx, src, x.shard_bytes, 1, e3, sizeof e3) != 0)
				throw HipError(e3);
			x.csr_ms = now_ms() - t2;
			if (r == 0 && tr != T_NONE && tr != T_SAME)
				HIP_CHECK(hipStreamSynchronize(x.xstream)); // the partition may be overwritten once this returns
		});
		if (!e.empty()) {
			// a rank failed with transfers posted: its partner may wait for ever -- abort every communicator so that all
			// streams drain, and say that the engine has to be created again
			if (tr == T_RCCL) {
				m->abort_comms();
				m->transport = T_PEER, m->transport_name = "peer-copy (RCCL communicators aborted after: " + e + ")";
			}
			throw HipError(e);
		}
		m->have_shards = true;
		m->ms[3] = now_ms() - t0;
		return 0;
	} catch (const std::exception &e) {
		set_err(err, errlen, e.what());
		return 1;
	}
}

extern "C" povu_hip_forest *povu_hip_multi_decompose(povu_hip_multi *m, uint32_t flags, povu_hip_multi_sink sink, void *user,
						     char *err, size_t errlen)
{
	try {
		if (!m)
			throw HipError("multi: null handle");
		if (!m->have_shards)
			throw HipError("multi: no shards resident: call povu_hip_multi_upload and povu_hip_multi_scatter first");
		const double t0 = now_ms();
		const std::string e = m->on_all([&](uint32_t r) {
			Worker &x = *m->w[r];
			if (x.forest) {
				povu_hip_forest_free(x.forest);
				x.forest = nullptr;
			}
			char e2[512] = {0};
			const povu_hip_opts o{0, 1, flags};
			const double t1 = now_ms();
			povu_hip_forest *f = povu_hip_decompose(x.ctx, &o, e2, sizeof e2);
			if (!f)
				throw HipError(e2);
			x.forest = f;
			if (povu_hip_forest_tree_count(f) && povu_hip_forest_globalize(f, x.ctx) != 0)
				throw HipError("component ids of the shard do not match its forest (internal)");
			x.dec_ms = now_ms() - t1;
			x.sink_ms = 0;
			if (sink) {
				const double t2 = now_ms();
				povu_hip_forest_wait(f); // (the sink reads the arrays)
				if (sink(r, f, user) != 0)
					throw HipError("the result sink failed");
				x.sink_ms = now_ms() - t2;
			}
		});
		if (!e.empty())
			throw HipError(e);
		const double t1 = now_ms();
		auto out = std::make_unique<povu_hip_forest>();
		out->pool = m->w[0]->ctx->pool;
		for (auto &x : m->w) {
			if (!x->forest)
				continue;
			adopt_forest(*out, *x->forest);
			povu_hip_forest_free(x->forest);
			x->forest = nullptr;
		}
		std::stable_sort(out->trees.begin(), out->trees.end(),
				 [](const povu_hip_forest::Tree &a, const povu_hip_forest::Tree &b) { return a.component_id < b.component_id; });
		m->ms[5] = now_ms() - t1;
		m->ms[4] = now_ms() - t0;
		return out.release();
	} catch (const std::exception &e) {
		set_err(err, errlen, e.what());
		return nullptr;
	}
}

extern "C" int povu_hip_multi_rank(const povu_hip_multi *m, uint32_t rank, povu_hip_multi_rank_info *out)
{
	if (!m || !out || rank >= m->world)
		return 1;
	const Worker &x = *m->w[rank];
	out->device = x.device;
	out->n_vtx = x.nv, out->n_links = x.ne, out->n_components = x.nc;
	out->shard_bytes = x.shard_bytes;
	out->recv_ms = x.recv_ms, out->csr_ms = x.csr_ms, out->decompose_ms = x.dec_ms, out->sink_ms = x.sink_ms;
	uint64_t b[4] = {0, 0, 0, 0};
	povu_hip_transfer_bytes(x.ctx, b);
	out->h2d = b[0], out->d2h = b[1], out->peer_out = b[2], out->peer_in = b[3];
	return 0;
}

extern "C" int povu_hip_multi_times(const povu_hip_multi *m, double out_ms[6])
{
	if (!m || !out_ms)
		return 1;
	for (int i = 0; i < 6; i++)
		out_ms[i] = m->ms[i];
	return 0;
}
