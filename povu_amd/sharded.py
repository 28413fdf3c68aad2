"""Component sharding across GPUs: one process per GPU, components are the unit.

The work is done by the C library (`povu_amd/csrc/hip/shard.hip`, declared in `include/povu_hip.h`):

* `povu_hip_shard_partition`  -- the root labels the components of its resident graph on the GPU (row B's
  union-find kernels), bin-packs them over the ranks (LPT, `povu_hip_lpt_assign`) and partitions vertices and
  links on the device into one packed shard per rank;
* `povu_hip_comm_scatter` / `povu_hip_comm_gather` -- RCCL `ncclSend` / `ncclRecv` over xGMI from C++ on the
  context's own stream: shards out, PVST blocks back to the root (replaces the reference's static contiguous
  chunks per thread, app/subcommand/decompose.cpp:78-92,116-157).  No collective runs inside the traversal.

This module is the launcher-side plumbing: the same scatter / gather over `torch.distributed` (`*_over_dist`: with
backend "nccl" the packed shards and forests travel device to device through RCCL, with gloo through host memory --
the CPU-side tests and single-GPU rehearsals), the hand-over of the RCCL unique id for the library's own calls, and a
numpy model of the partition for the tests.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import List, Optional

import numpy as np

from . import hip as _hip
from .workloads import Links


# ---------------------------------------------------------------- planning (host)
def assign_owners(weights, world: int) -> np.ndarray:
    """LPT: components by weight descending (stable), each to the least loaded rank (lowest rank on ties) --
    the library's own rule (`povu_hip_lpt_assign`; `povu_hip_opts.rank/world` uses the same)."""
    return _hip.lpt_assign(weights, world).astype(np.int32)


def component_labels(links: Links) -> np.ndarray:
    """Component rank (by minimum vertex idx) of every vertex -- numpy/scipy model used by the tests only."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    n = links.n_vtx
    a = coo_matrix((np.ones(links.n_links, dtype=np.int8), (links.v1.astype(np.int64), links.v2.astype(np.int64))), shape=(n, n))
    _, lab = connected_components(a, directed=False)
    first = np.full(lab.max() + 1, n, dtype=np.int64)
    np.minimum.at(first, lab, np.arange(n))
    rank_of = np.empty_like(first)
    rank_of[np.argsort(first, kind="stable")] = np.arange(len(first))
    return rank_of[lab].astype(np.int64)


def partition_links(links: Links, comp: np.ndarray, owner: np.ndarray, rank: int):
    """Model of the device partition: sub-graph of the components owned by `rank`, vertices in ascending global
    idx, links in L-line order, + the global ids (1-based) of its components."""
    keep_v = owner[comp] == rank
    new_idx = np.cumsum(keep_v) - 1
    keep_e = keep_v[links.v1]
    sub = Links(links.vid[keep_v], new_idx[links.v1[keep_e]].astype(np.uint32), links.s1[keep_e],
                new_idx[links.v2[keep_e]].astype(np.uint32), links.s2[keep_e])
    comp_ids = np.unique(comp[keep_v]) + 1
    return sub, comp_ids


SHARD_MAGIC = 0x31647268735F7670  # "pv_shrd1"


def pack_shard(sub: Links, tips: np.ndarray, comp_ids, total_components: int) -> np.ndarray:
    """Model of the packed shard format of shard.hip (the bytes `Shards.export` returns): header of 8 u64
    [magic, n_vtx, n_links, n_components, components of the whole graph, 0, 0, 0], then vid | v1 | v2 | s1 | s2 |
    tip | component ids, every section padded to 256 bytes."""
    nv, ne, nc = sub.n_vtx, sub.n_links, len(comp_ids)
    pad = lambda b: (b + 255) & ~255  # noqa: E731
    secs = [np.ascontiguousarray(sub.vid, dtype=np.uint32).view(np.uint8), np.ascontiguousarray(sub.v1, dtype=np.uint32).view(np.uint8),
            np.ascontiguousarray(sub.v2, dtype=np.uint32).view(np.uint8), np.ascontiguousarray(sub.s1, dtype=np.uint8),
            np.ascontiguousarray(sub.s2, dtype=np.uint8), np.ascontiguousarray(tips, dtype=np.uint8),
            np.ascontiguousarray(comp_ids, dtype=np.uint32).view(np.uint8)]
    out = np.zeros(256 + sum(pad(x.size) for x in secs), dtype=np.uint8)
    out[:64].view(np.uint64)[:5] = (SHARD_MAGIC, nv, ne, nc, total_components)
    o = 256
    for x in secs:
        out[o:o + x.size] = x
        o += pad(x.size)
    return out


def infer_tips(links: Links) -> np.ndarray:
    """Tips as the loader infers them (src/mto/from_gfa.cpp:262-277): 1 = no link on the l side, else 2 = none on r."""
    has = np.zeros(2 * links.n_vtx, dtype=bool)
    has[2 * links.v1.astype(np.int64) + links.s1] = True
    has[2 * links.v2.astype(np.int64) + links.s2] = True
    l, r = has[0::2], has[1::2]
    return np.where(~l, 1, np.where(~r, 2, 0)).astype(np.uint8)


def unpack_shard(buf: np.ndarray):
    """Packed shard bytes -> (Links, tips, component ids, total components): the layout of shard.hip."""
    h = buf[:64].view(np.uint64)
    nv, ne, nc, total = int(h[1]), int(h[2]), int(h[3]), int(h[4])
    pad = lambda b: (b + 255) & ~255  # noqa: E731
    o = 256
    sec = {}
    for name, nb in (("vid", nv * 4), ("v1", ne * 4), ("v2", ne * 4), ("s1", ne), ("s2", ne), ("tip", nv), ("ids", nc * 4)):
        sec[name] = buf[o:o + nb]
        o += pad(nb)
    u32 = lambda x: x.view(np.uint32).copy()  # noqa: E731
    return (Links(u32(sec["vid"]), u32(sec["v1"]), sec["s1"].copy(), u32(sec["v2"]), sec["s2"].copy()), sec["tip"].copy(),
            u32(sec["ids"]), total)


# ---------------------------------------------------------------- RCCL path (C++)
class ShardComm:
    """RCCL communicator owned by the C library, bound to one HipDecomposer (its stream carries the transfers).
    The unique id is created on rank 0 and broadcast with torch.distributed (any backend)."""

    def __init__(self, hip: "_hip.HipDecomposer", rank: int, world: int):
        import torch.distributed as dist

        self._lib = _hip.load_lib()
        self.hip, self.rank, self.world = hip, rank, world
        err = C.create_string_buffer(512)
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            if self._lib.povu_hip_comm_unique_id(buf, err, 512) != 0:
                raise RuntimeError(err.value.decode())
            ident[0] = buf.raw
        if world > 1:
            dist.broadcast_object_list(ident, src=0)
        self._h = self._lib.povu_hip_comm_create(hip._ctx, ident[0], rank, world, err, 512)
        if not self._h:
            raise RuntimeError(err.value.decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.povu_hip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def scatter(self, shards: Optional["_hip.Shards"]):
        """Root: `shards` = the partition of its full graph; others: None.  Afterwards every rank's decomposer
        holds its shard as resident graph."""
        err = C.create_string_buffer(512)
        if self._lib.povu_hip_comm_scatter(self._h, shards._h if shards is not None else None, self.hip._ctx, err, 512) != 0:
            raise RuntimeError(err.value.decode())

    def gather(self, forest: "_hip.Forest") -> "_hip.Forest":
        """Every rank passes the forest of its shard (global component ids); the root gets the merged forest (it
        takes over the arrays of the root's own forest), the others an empty one."""
        err = C.create_string_buffer(512)
        h = self._lib.povu_hip_comm_gather(self._h, forest._h, err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return _hip.Forest(self._lib, h)

    def times(self) -> dict:
        t = (C.c_double * 2)()
        self._lib.povu_hip_comm_times(self._h, t)
        return dict(scatter_ms=t[0], gather_ms=t[1])


# ---------------------------------------------------------------- the same over torch.distributed
class _DeviceBytes:
    """Zero-copy view of library-owned device memory for torch.as_tensor (CUDA array interface, works on ROCm)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3,
                                         "strides": None}


def scatter_over_dist(full: Optional["_hip.HipDecomposer"], work: "_hip.HipDecomposer", rank: int, world: int, device):
    """Scatter with a torch.distributed backend.  On GPUs (backend "nccl" = RCCL) the packed shards travel device to
    device over xGMI: the root sends straight out of its partition block in HBM (no copy), a receiver builds its CSR from
    the bytes where they land.  With a CPU backend (gloo: CPU-side tests, single-GPU rehearsals) they go through host
    memory.  The root's sends are asynchronous: its own shard is loaded and its CSR built beside them."""
    import torch
    import torch.distributed as dist

    on_gpu = device.type == "cuda"
    if rank == 0:
        shards = full.partition(world)
        sizes = torch.tensor([shards.info(r)["bytes"] for r in range(world)], dtype=torch.int64, device=device)
    else:
        shards = None
        sizes = torch.zeros(world, dtype=torch.int64, device=device)
    dist.broadcast(sizes, src=0)
    if rank == 0:
        pending = []
        for r in range(1, world):
            i = shards.info(r)
            t = None
            if on_gpu:
                try:  # zero-copy view of the shard in the partition block
                    t = torch.as_tensor(_DeviceBytes(i["device_ptr"], i["bytes"]), device=device)
                except Exception:  # (a torch build without the CUDA array interface: stage through host memory)
                    t = None
            if t is None:
                t = torch.from_numpy(shards.export(r))
                if on_gpu:
                    t = t.to(device)
            pending.append((dist.isend(t, r), t))
        i0 = shards.info(0)
        work.upload_shard(i0["device_ptr"], i0["bytes"], on_device=True)
        for w, _t in pending:
            w.wait()
        return shards
    n = int(sizes[rank].item())
    buf = torch.empty(n, dtype=torch.uint8, device=device)
    dist.recv(buf, 0)
    if on_gpu:
        torch.cuda.current_stream().synchronize()  # (the library works on its own stream)
        work.upload_shard(buf.data_ptr(), n, on_device=True)
    else:
        work.upload_shard(buf.numpy())
    return None


def gather_over_dist(work: "_hip.HipDecomposer", forest: "_hip.Forest", rank: int, world: int, device):
    """PVST gather to rank 0 over torch.distributed: packed forests (device to device on GPUs), merged by the library."""
    import torch
    import torch.distributed as dist

    mine = forest.pack()
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    sizes[rank] = mine.size
    dist.all_reduce(sizes)
    if rank != 0:
        dist.send(torch.from_numpy(mine).to(device), 0)
        return None
    parts = [mine]
    recv = []
    for r in range(1, world):
        b = torch.empty(int(sizes[r].item()), dtype=torch.uint8, device=device)
        recv.append((dist.irecv(b, r), b))
    for w, b in recv:
        w.wait()
        parts.append(b.cpu().numpy())
    return work.merge_forests(parts)


def gather_shared(work: "_hip.HipDecomposer", forest: "_hip.Forest", rank: int, world: int, device, job_tag: str):
    """PVST gather to rank 0 WITHOUT moving a block: every rank's decompose has already copied its PVST arrays into
    page-locked host memory over its own GPU's PCIe link, and with `HipDecomposer.share_results` that memory is a named
    shared-memory segment.  The ranks exchange one 64-byte descriptor each (an all-gather on whatever backend the job runs:
    RCCL on GPUs, gloo in the CPU-side rehearsals) and the root maps the segments (`povu_hip_forest_attach`).  Nothing goes
    back to a device, over xGMI, or through the root's PCIe link.

    The caller keeps `forest` alive until its NEXT gather (the root reads the arrays in place); the root gets the merged
    forest, which takes over the arrays of its own `forest`, the others get None."""
    import torch
    import torch.distributed as dist

    d = forest.share(rank)
    mine = torch.from_numpy(d.view(np.int64).copy()).to(device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    if rank != 0:
        return None
    descs = np.stack([p.cpu().numpy().view(np.uint64) for p in parts])
    return work.attach_forests(forest, 0, job_tag, descs)


# ---------------------------------------------------------------- strong-scaling bench driver
class ShardedBench:
    """bench.py --gpus N: the SAME graph every step, sharded over N ranks.  Rank 0 keeps the whole graph resident in
    a context of its own (`full`); a step = partition on rank 0's GPU, scatter, per-shard CSR build + decompose on
    every rank, gather of the PVST arrays to rank 0 -- all inside the timed region."""

    def __init__(self, work: "_hip.HipDecomposer", rank: int, world: int, comm_device, build_workload, device_index: int = 0):
        import torch.distributed as dist

        self.work, self.rank, self.world, self.dev = work, rank, world, comm_device
        # Transfers: torch.distributed (backend "nccl" = RCCL: ncclSend / ncclRecv of device buffers over xGMI) by default;
        # POVU_BENCH_NATIVE_RCCL=1 takes the library's own RCCL calls (shard.hip: povu_hip_comm_scatter / _gather), which
        # no multi-GPU box has run yet
        import os
        self.native = comm_device.type == "cuda" and os.environ.get("POVU_BENCH_NATIVE_RCCL") == "1"
        # Gather: through shared memory by default (`gather_shared`: no PVST block crosses PCIe twice); POVU_BENCH_GATHER=wire
        # takes the packed forests over the backend instead (what a job across nodes would need)
        self.shared = os.environ.get("POVU_BENCH_GATHER", "shared") == "shared" and not self.native
        self.job = None
        if self.shared:
            tag = [f"{os.getpid()}x{int(time.time() * 1e3) & 0xFFFFFFF:x}" if rank == 0 else None]
            dist.broadcast_object_list(tag, src=0)
            self.job = tag[0]
            work.share_results(f"{self.job}.{rank}")
        self._mine = None  # this rank's last forest: the root reads its arrays in place until the next gather
        self.full = None
        self.meta = None
        if rank == 0:
            g, wl = build_workload()
            self.full = _hip.HipDecomposer(device_index)
            self.full.upload(g)
            self.meta = dict(workload=wl, links=g.n_links, segments=g.n_vtx)
            del g
        self.comm = ShardComm(work, rank, world) if self.native else None
        self.phase = dict(partition=0.0, scatter=0.0, decompose=0.0, gather=0.0)
        self.steps = 0
        self.last = None
        self.last_shards = None
        dist.barrier()

    def step(self):
        import torch

        t0 = time.perf_counter()
        shards = None
        if self.native:
            if self.rank == 0:
                shards = self.full.partition(self.world)
            t1 = time.perf_counter()
            self.comm.scatter(shards)
        else:
            t1 = t0
            shards = scatter_over_dist(self.full, self.work, self.rank, self.world, self.dev)
        t2 = time.perf_counter()
        f = self.work.decompose_shard(flags=_hip.F_NO_STAGE_TIMES)
        t3 = time.perf_counter()
        merged = self._gather(f)
        t4 = time.perf_counter()
        self.phase["partition"] += t1 - t0
        self.phase["scatter"] += t2 - t1
        self.phase["decompose"] += t3 - t2
        self.phase["gather"] += t4 - t3
        self.steps += 1
        if self.rank == 0:
            self.last, self.last_shards = merged, shards
        return merged

    def _gather(self, f):
        if self.native:
            return self.comm.gather(f)
        if self.shared:
            merged = gather_shared(self.work, f, self.rank, self.world, self.dev, self.job)
            self._mine = f  # (replaces the forest of the step before: every rank has passed this step's all-gather, so the
            return merged   #  root is done with that one)
        return gather_over_dist(self.work, f, self.rank, self.world, self.dev)

    def step_resident(self):
        """The part of a step that starts from "every rank's shard CSR resident" (what the last scatter left) and ends with
        "forest merged on rank 0": per-shard decompose + gather.  Same start and end as the N = 1 measurement of bench.py."""
        # (with the shared-memory gather the passes overlap: a rank's decompose returns while the copy engine still moves its PVST
        # arrays to the host -- POVU_HIP_F_ASYNC --, the descriptors travel at once, and this rank's next pass runs under the
        # copies.  Nobody reads the arrays before `sync`, which waits for them on every rank.)
        fl = _hip.F_NO_STAGE_TIMES | (_hip.F_ASYNC if self.shared else 0)
        f = self.work.decompose_shard(flags=fl)
        merged = self._gather(f)
        if self.rank == 0:
            self.last = merged
        return merged

    def sync(self):
        import torch
        import torch.distributed as dist

        if self._mine is not None:
            self._mine.wait()  # this rank's last PVST arrays are in (shared) host memory
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def reset_phases(self):
        self.phase = {k: 0.0 for k in self.phase}
        self.steps = 0

    def summary(self) -> Optional[dict]:
        """Collective: per-rank phase times; rank 0 returns the description of the job."""
        import torch
        import torch.distributed as dist

        n = max(1, self.steps)
        mine = torch.tensor([self.phase[k] / n * 1e3 for k in ("partition", "scatter", "decompose", "gather")], dtype=torch.float64,
                            device=self.dev)
        allp = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allp, mine)
        if self.rank != 0:
            return None
        sh = self.last_shards
        shards = [sh.info(r) for r in range(self.world)]
        for s in shards:
            s.pop("device_ptr", None)
        mean_w = sum(s["weight"] for s in shards) / self.world
        flub = 0
        f = self.last
        ids = f.component_ids()
        lib = f._lib
        t = _hip._Tree()
        for i in range(len(f)):
            lib.povu_hip_forest_get(f._h, i, C.byref(t))
            flub += t.n_pvst - 1
        out = dict(self.meta)
        out.update(components=len(ids), flubbles=flub, shards=shards,
                   lpt_max_over_mean=max(s["weight"] for s in shards) / mean_w,
                   sharding=("strong scaling: rank 0 labels the components on its GPU, LPT bin-packing, device partition, "
                             + ("RCCL ncclSend/ncclRecv scatter of the packed shards and gather of the PVST blocks from C++ (shard.hip)"
                                if self.native else
                                (("RCCL send/recv (torch.distributed, backend nccl) of the packed shards out of the partition block in HBM"
                                  if self.dev.type == "cuda" else "scatter over torch.distributed through host memory (rehearsal)")
                                 + ("; gather: every rank's PVST block lands in shared page-locked host memory over its own PCIe link, "
                                    "the ranks all-gather one 64-byte descriptor each and the root maps the blocks"
                                    if self.shared else "; gather: packed forests back over the same backend")))
                             + "; all inside the timed region"),
                   phase_ms={"per_rank": [dict(zip(("partition", "scatter", "decompose", "gather"), [float(x) for x in p.tolist()]))
                                          for p in allp],
                             "partition_device": sh.times()})
        return out

    def close(self):
        self.last = None   # (the merged forest reads the other ranks' segments: gone before they are)
        self._mine = None
        if self.comm:
            self.comm.close()
        if self.full:
            self.full.close()
