"""Component sharding across GPUs: one process per GPU, components are the unit.

* :func:`assign_owners` -- greedy longest-processing-time bin packing of components over ranks
  (replaces the reference's static contiguous chunks, app/subcommand/decompose.cpp:78-92,116-157).
* :func:`partition_links` -- the link slice a rank needs for its components (the "scatter").
* :func:`scatter_links` / :func:`gather_forest` -- torch.distributed exchange (RCCL on GPUs, gloo
  in the CPU tests): sizes are all-gathered first, payloads move point to point to/from rank 0.
No collective runs inside the traversal itself.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

from .workloads import Links


def assign_owners(weights: Sequence[int], world: int) -> np.ndarray:
    """LPT: components by weight descending (stable), each to the least loaded rank (lowest rank
    on ties).  Same rule as the C ABI uses for `povu_hip_opts.rank/world`."""
    w = np.asarray(weights, dtype=np.int64)
    order = np.argsort(-w, kind="stable")
    load = np.zeros(world, dtype=np.int64)
    owner = np.zeros(len(w), dtype=np.int32)
    for c in order.tolist():
        r = int(np.argmin(load))
        owner[c] = r
        load[r] += int(w[c]) + 1
    return owner


def component_labels(links: Links) -> np.ndarray:
    """Host-side union-find labelling used only to plan the scatter (component rank by min vertex)."""
    n = links.n_vtx
    parent = np.arange(n, dtype=np.int64)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b in zip(links.v1.tolist(), links.v2.tolist()):
        ra, rb = find(a), find(b)
        if ra != rb:
            if ra < rb:
                parent[rb] = ra
            else:
                parent[ra] = rb
    root = np.array([find(v) for v in range(n)], dtype=np.int64)
    _, comp = np.unique(root, return_inverse=True)  # roots are minima => rank by min vertex idx
    return comp.astype(np.int64)


def partition_links(links: Links, comp: np.ndarray, owner: np.ndarray, rank: int):
    """Sub-graph of the components owned by `rank`, vertices kept in ascending global idx (so the
    shard's own component numbering preserves the global order) + the global ids of its components."""
    keep_v = owner[comp] == rank
    new_idx = np.cumsum(keep_v) - 1
    keep_e = keep_v[links.v1]
    sub = Links(links.vid[keep_v], new_idx[links.v1[keep_e]].astype(np.uint32), links.s1[keep_e],
                new_idx[links.v2[keep_e]].astype(np.uint32), links.s2[keep_e])
    comp_ids = np.unique(comp[keep_v]) + 1  # 1-based global component ids, ascending
    return sub, comp_ids


def _flat(forest, id_map=None):
    """Forest -> (header int32 [n_trees, 2] = (component id, n_pvst), payload int32 bit-view of
    [a_id | z_id | parent | orientation bits] per tree)."""
    hdr, pay = [], []
    for i in range(len(forest)):
        t = forest.tree(i)
        cid = t.component_id if id_map is None else int(id_map[t.component_id - 1])
        n = t.a_id.shape[0]
        hdr.append((cid, n))
        pay += [t.a_id.astype(np.uint32, copy=False), t.z_id.astype(np.uint32, copy=False),
                t.parent.astype(np.uint32, copy=False),
                t.a_or.astype(np.uint32) | (t.z_or.astype(np.uint32) << 1)]
    h = np.array(hdr, dtype=np.int32).reshape(-1, 2)
    p = np.concatenate(pay) if pay else np.zeros(0, dtype=np.uint32)
    return h, np.ascontiguousarray(p).view(np.int32)


def _unpack(out, hh, pp):
    pp = pp.view(np.uint32)
    off = 0
    for cid, n in hh.reshape(-1, 2).tolist():
        blk = pp[off:off + 4 * n]
        out[int(cid)] = dict(a_id=blk[:n], z_id=blk[n:2 * n], parent=blk[2 * n:3 * n],
                             a_or=(blk[3 * n:] & 1).astype(np.uint8), z_or=((blk[3 * n:] >> 1) & 1).astype(np.uint8))
        off += 4 * n


def _views(out, hdr, block, total, offs, id_map=None):
    """Slice one rank's raw block (numpy uint8) into per-component array views (no copies)."""
    a = block[offs[0]:offs[0] + 4 * total].view(np.uint32)
    z = block[offs[1]:offs[1] + 4 * total].view(np.uint32)
    p = block[offs[2]:offs[2] + 4 * total].view(np.uint32)
    ao = block[offs[3]:offs[3] + total]
    zo = block[offs[4]:offs[4] + total]
    for cid, n, first in hdr.reshape(-1, 3).tolist():
        if id_map is not None:
            cid = int(id_map[cid - 1])
        sl = slice(first, first + n)
        out[int(cid)] = dict(a_id=a[sl], z_id=z[sl], parent=p[sl], a_or=ao[sl], z_or=zo[sl])


_pinned_cache = {}


def _pinned(nbytes: int, key):
    import torch
    t = _pinned_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
        _pinned_cache[key] = t
    return t[:nbytes]


def gather_forest(forest, rank: int, world: int, device, id_map=None) -> Dict[int, dict] | None:
    """PVST gather to rank 0.  One all-gather of the sizes, then point-to-point payloads (RCCL
    send/recv over xGMI on GPUs, gloo on CPU); no collective touches the traversal itself.
    A HIP forest ships its page-locked result block as is (one H2D, one send per rank; rank 0 lands
    the blocks in pinned memory and slices views); other forest objects go through `_flat`."""
    import torch
    import torch.distributed as dist

    if not hasattr(forest, "raw"):
        return _gather_flat(forest, rank, world, device, id_map)
    block, total, offs, hdr = forest.raw()
    if id_map is not None and len(hdr):
        hdr = hdr.copy()
        hdr[:, 0] = np.asarray(id_map, dtype=np.int64)[hdr[:, 0] - 1]
    meta = np.array([hdr.shape[0], block.shape[0], total] + offs, dtype=np.int64)
    all_meta = torch.zeros(8 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_meta, torch.from_numpy(meta).to(device))
    if rank != 0:
        if hdr.shape[0]:
            dist.send(torch.from_numpy(hdr.reshape(-1)).to(device), 0)
            dist.send(torch.from_numpy(block).to(device, non_blocking=True), 0)
        return None
    out: Dict[int, dict] = {}
    _views(out, hdr, block, total, offs)
    am = all_meta.cpu().numpy().reshape(world, 8)
    pending = []
    for r in range(1, world):
        nh, nb = int(am[r, 0]), int(am[r, 1])
        if nh == 0:
            continue
        hb = torch.empty(3 * nh, dtype=torch.int64, device=device)
        bb = torch.empty(nb, dtype=torch.uint8, device=device)
        dist.recv(hb, r)
        dist.recv(bb, r)
        pending.append((r, hb, bb))
    landed = []
    for r, hb, bb in pending:  # device -> pinned host, all copies in flight before the single sync
        host = _pinned(bb.numel(), ("blk", r))
        host.copy_(bb, non_blocking=True)
        landed.append((r, hb.cpu().numpy(), host))
    if device.type == "cuda":
        torch.cuda.current_stream().synchronize()
    for r, h, host in landed:
        _views(out, h, host.numpy(), int(am[r, 2]), [int(x) for x in am[r, 3:8]])
    return out


class PipelinedGather:
    """PVST gather to rank 0 that overlaps with the next decompose (one process per GPU).

    submit() posts this step's transfers and returns at once: senders `isend` their forest's pinned
    block (kept alive until the send has completed), rank 0 posts `irecv`s into device buffers and
    queues the device-to-pinned-host copies behind them on the current stream.  Buffers are double
    buffered by step parity, so step k's traffic runs under step k+1's kernels.  finish() drains
    everything and returns the component views of the LAST submitted step (rank 0) / None.
    """

    def __init__(self, rank: int, world: int, device):
        self.rank, self.world, self.device = rank, world, device
        self.step = 0
        self.inflight = []          # per step: list of (work, keep-alive objects)
        self.landed = None          # rank 0: what the last step put into host memory
        self._dev = {}              # (parity, peer, kind) -> device buffer

    def _buf(self, key, n, dtype):
        import torch
        t = self._dev.get(key)
        if t is None or t.numel() < n:
            t = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._dev[key] = t
        return t[:n]

    def _drain(self, keep_last: int):
        while len(self.inflight) > keep_last:
            for work, _keep in self.inflight.pop(0):
                work.wait()

    def submit(self, forest, id_map=None):
        import torch
        import torch.distributed as dist

        par = self.step & 1
        self.step += 1
        self._drain(1)  # the buffers of this parity were used two steps ago: make sure they are free
        block, total, offs, hdr = forest.raw()
        if id_map is not None and len(hdr):
            hdr = hdr.copy()
            hdr[:, 0] = np.asarray(id_map, dtype=np.int64)[hdr[:, 0] - 1]
        meta = np.array([hdr.shape[0], block.shape[0], total] + offs, dtype=np.int64)
        all_meta = torch.zeros(8 * self.world, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(all_meta, torch.from_numpy(meta).to(self.device))
        works = []
        if self.rank != 0:
            if hdr.shape[0]:
                ht = torch.from_numpy(hdr.reshape(-1)).to(self.device)
                bt = torch.from_numpy(block).to(self.device, non_blocking=True)
                works.append((dist.isend(ht, 0), (ht, forest)))
                works.append((dist.isend(bt, 0), (bt, forest)))
            self.inflight.append(works)
            return
        am = all_meta.cpu().numpy().reshape(self.world, 8)
        landed = [(0, hdr, block, total, offs, forest)]
        for r in range(1, self.world):
            nh, nb = int(am[r, 0]), int(am[r, 1])
            if nh == 0:
                continue
            hb = self._buf((par, r, "h"), 3 * nh, torch.int64)
            bb = self._buf((par, r, "b"), nb, torch.uint8)
            wh, wb = dist.irecv(hb, r), dist.irecv(bb, r)
            wh.wait()   # NCCL: orders the current stream behind the transfer, the host does not block
            wb.wait()
            hh = _pinned(3 * nh * 8, ("ph", par, r)).view(torch.int64)
            host = _pinned(nb, ("pb", par, r))
            hh.copy_(hb, non_blocking=True)
            host.copy_(bb, non_blocking=True)
            landed.append((r, hh, host, int(am[r, 2]), [int(x) for x in am[r, 3:8]], None))
        self.inflight.append(works)
        self.landed = landed

    def finish(self) -> Dict[int, dict] | None:
        import torch

        self._drain(0)
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        if self.rank != 0 or self.landed is None:
            return None
        out: Dict[int, dict] = {}
        for r, h, blk, total, offs, _keep in self.landed:
            hn = h if isinstance(h, np.ndarray) else h.numpy()
            bn = blk if isinstance(blk, np.ndarray) else blk.numpy()
            _views(out, hn, bn, total, offs)
        return out


def _gather_flat(forest, rank: int, world: int, device, id_map=None) -> Dict[int, dict] | None:
    import torch
    import torch.distributed as dist

    h, p = _flat(forest, id_map)
    sizes = torch.tensor([h.shape[0], p.shape[0]], dtype=torch.int64, device=device)
    all_sizes = torch.zeros(2 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_sizes, sizes)
    if rank != 0:
        if h.shape[0]:
            buf = torch.from_numpy(np.concatenate([h.reshape(-1), p])).to(device)
            dist.send(buf, 0)
        return None
    out: Dict[int, dict] = {}
    _unpack(out, h, p)
    szs = all_sizes.cpu().numpy().reshape(world, 2)
    bufs = []
    for r in range(1, world):
        nh, npay = int(szs[r, 0]), int(szs[r, 1])
        if nh == 0:
            continue
        b = torch.empty(2 * nh + npay, dtype=torch.int32, device=device)
        dist.recv(b, r)
        bufs.append((nh, b))
    for nh, b in bufs:
        a = b.cpu().numpy()
        _unpack(out, a[:2 * nh], a[2 * nh:])
    return out


def scatter_links(links: Links | None, rank: int, world: int, device):
    """Initial component scatter: rank 0 labels components, bin-packs them and sends every rank the
    link slice of its components.  Returns (sub-graph, global component ids of the shard)."""
    import torch
    import torch.distributed as dist

    if rank == 0:
        comp = component_labels(links)
        nc = int(comp.max()) + 1
        wv = np.bincount(comp, minlength=nc)
        we = np.bincount(comp[links.v1], minlength=nc)
        owner = assign_owners(wv + we, world)
        parts = [partition_links(links, comp, owner, r) for r in range(world)]
        for r in range(1, world):
            sub, ids = parts[r]
            meta = torch.tensor([sub.n_vtx, sub.n_links, len(ids)], dtype=torch.int64, device=device)
            dist.send(meta, r)
            for arr in (sub.vid, sub.v1, sub.s1, sub.v2, sub.s2, ids):
                if len(arr):
                    dist.send(torch.from_numpy(np.asarray(arr).astype(np.int64)).to(device), r)
        return parts[0]
    meta = torch.zeros(3, dtype=torch.int64, device=device)
    dist.recv(meta, 0)
    nv, ne, nid = (int(x) for x in meta.tolist())
    got = []
    for n in (nv, ne, ne, ne, ne, nid):
        t = torch.zeros(n, dtype=torch.int64, device=device)
        if n:
            dist.recv(t, 0)
        got.append(t.cpu().numpy())
    sub = Links(got[0].astype(np.uint32), got[1].astype(np.uint32), got[2].astype(np.uint8),
                got[3].astype(np.uint32), got[4].astype(np.uint8))
    return sub, got[5].astype(np.int64)
