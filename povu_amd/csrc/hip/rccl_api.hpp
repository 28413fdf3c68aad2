// rccl_api.hpp -- RCCL entry points, resolved on first use (inside a process that already loaded RCCL -- PyTorch -- the
// soname resolves to that copy).  Shared by the multi-process communicator (shard.hip) and the one-process engine (multi.hip).
#pragma once
#include "common.hpp"

#include <rccl/rccl.h>

#include <dlfcn.h>

namespace povu_hip
{
struct Rccl {
	void *h = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*CommAbort)(ncclComm_t) = nullptr; // (optional: only used to unblock streams after a failure)
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
inline Rccl &rccl()
{
	static Rccl r = [] {
		Rccl x;
		// inside a process that already loaded RCCL (PyTorch) the soname resolves to that copy
		for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			x.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (x.h)
				break;
		}
		if (!x.h)
			return x;
		auto sym = [&](const char *n) { return dlsym(x.h, n); };
		x.GetUniqueId = (decltype(x.GetUniqueId))sym("ncclGetUniqueId");
		x.CommInitRank = (decltype(x.CommInitRank))sym("ncclCommInitRank");
		x.CommInitAll = (decltype(x.CommInitAll))sym("ncclCommInitAll");
		x.CommDestroy = (decltype(x.CommDestroy))sym("ncclCommDestroy");
		x.CommAbort = (decltype(x.CommAbort))sym("ncclCommAbort");
		x.Send = (decltype(x.Send))sym("ncclSend");
		x.Recv = (decltype(x.Recv))sym("ncclRecv");
		x.GroupStart = (decltype(x.GroupStart))sym("ncclGroupStart");
		x.GroupEnd = (decltype(x.GroupEnd))sym("ncclGroupEnd");
		x.Broadcast = (decltype(x.Broadcast))sym("ncclBroadcast");
		x.AllGather = (decltype(x.AllGather))sym("ncclAllGather");
		x.GetErrorString = (decltype(x.GetErrorString))sym("ncclGetErrorString");
		return x;
	}();
	if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd ||
	    !r.Broadcast || !r.AllGather || !r.GetErrorString)
		throw HipError("RCCL (librccl.so.1) could not be loaded");
	return r;
}
#define NCCL_CHECK(expr)                                                                                   \
	do {                                                                                               \
		ncclResult_t r__ = (expr);                                                                 \
		if (r__ != ncclSuccess)                                                                    \
			throw HipError(std::string(#expr) + ": " + rccl().GetErrorString(r__));            \
	} while (0)
} // namespace povu_hip
