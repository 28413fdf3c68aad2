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


def gather_forest(forest, rank: int, world: int, device, id_map=None) -> Dict[int, dict] | None:
    """PVST gather to rank 0: one all-gather of (n_trees, payload words), then point-to-point
    payload transfers (RCCL send/recv over xGMI on GPUs, gloo on CPU).  No collective touches the
    traversal itself."""
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
