"""Multi-process path on CPU (gloo): the component scatter / PVST gather protocol of povu_amd/sharded.py without
a GPU.  What runs here: the library's LPT rule (povu_hip_lpt_assign, host only), the numpy model of the device
partition, the packed shard wire format, world_size 2 and 4 exchanges over gloo with the oracle standing in for
the per-shard decompose.  On the GPU box tests/test_gpu_sharded.py checks the device partition against the same
model byte for byte and drives the HIP path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from povu_amd import sharded, workloads as W


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph(kind):
    if kind == "skewed":  # a few large components of very different sizes + many tiny ones
        return W.hprc_shaped([3000, 300, 2500, 2800, 700, 1800, 120, 2900, 1500, 2200, 900], seed=5, tiny=60)
    return W.hprc_shaped([1200, 400, 800, 300], seed=11, tiny=25)


def plan(g, world):
    comp = sharded.component_labels(g)
    nc = int(comp.max()) + 1
    w = np.bincount(comp, minlength=nc) + np.bincount(comp[g.v1], minlength=nc)
    return comp, nc, sharded.assign_owners(w, world), w


def _worker(rank, world, port, kind, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # scatter: rank 0 plans and ships packed shards; everybody unpacks its own
    if rank == 0:
        g = _graph(kind)
        comp, nc, owner, _ = plan(g, world)
        tips = sharded.infer_tips(g)
        packed = []
        for r in range(world):
            sub, ids = sharded.partition_links(g, comp, owner, r)
            packed.append(sharded.pack_shard(sub, tips[owner[comp] == r], ids, nc))
        sizes = torch.tensor([p.size for p in packed], dtype=torch.int64)
    else:
        sizes = torch.zeros(world, dtype=torch.int64)
    dist.broadcast(sizes, src=0)
    if rank == 0:
        for r in range(1, world):
            dist.send(torch.from_numpy(packed[r]), r)
        mine = packed[0]
    else:
        buf = torch.zeros(int(sizes[rank]), dtype=torch.uint8)
        dist.recv(buf, 0)
        mine = buf.numpy()
    sub, tips, ids, total = sharded.unpack_shard(mine)
    # per-shard decompose (the oracle stands in for the HIP path), ids rewritten to those of the whole graph
    local = O.decompose(sub, tips=tips) if sub.n_vtx else {}
    texts = {int(ids[k - 1]): v for k, v in local.items()}
    gathered = [None] * world
    dist.gather_object((total, texts), gathered if rank == 0 else None, dst=0)
    if rank == 0:
        merged = {}
        for tot, t in gathered:
            assert tot == nc
            assert not (set(t) & set(merged))
            merged.update(t)
        torch.save(merged, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "plain"), (4, "skewed")])
def test_scatter_decompose_gather_over_gloo(tmp_path, world, kind):
    out = str(tmp_path / "merged.pt")
    mp.spawn(_worker, args=(world, _free_port(), kind, out), nprocs=world, join=True)
    got = torch.load(out)
    want = O.decompose(_graph(kind))
    assert got == want


def test_lpt_is_deterministic_and_balanced():
    rng = np.random.default_rng(0)
    w = rng.integers(1, 1000, size=200)
    o = sharded.assign_owners(w, 3)
    assert o.tolist() == sharded.assign_owners(w, 3).tolist()
    loads = np.bincount(o, weights=w, minlength=3)
    assert loads.max() - loads.min() <= w.max()
    assert sharded.assign_owners(w, 1).tolist() == [0] * len(w)
    # ties go to the lowest rank, heaviest first
    assert sharded.assign_owners([5, 3, 9, 1, 1, 7], 3).tolist() == [2, 2, 0, 1, 0, 1]


def test_lpt_load_on_skewed_components_world4():
    """Skewed component sizes over 4 ranks: the busiest rank stays within 1.25x of the mean load."""
    g = _graph("skewed")
    comp, nc, owner, w = plan(g, 4)
    loads = np.bincount(owner, weights=w + 1, minlength=4)
    assert loads.max() <= 1.25 * loads.mean(), loads
    assert loads.max() <= loads.mean() + (w + 1).max()  # the list-scheduling bound, whatever the sizes
    # whole-genome shape (chromosome-sized components), 8 ranks
    sizes = np.array(W.CHR_MBP, dtype=np.int64) * 400000
    o8 = sharded.assign_owners(sizes, 8)
    l8 = np.bincount(o8, weights=sizes + 1, minlength=8)
    assert l8.max() <= 1.25 * l8.mean(), l8


def test_partition_model_preserves_order_and_ids():
    g = _graph("plain")
    comp, nc, owner, _ = plan(g, 2)
    whole = O.decompose(g)
    seen = {}
    for r in range(2):
        sub, ids = sharded.partition_links(g, comp, owner, r)
        assert np.all(np.diff(sub.vid.astype(np.int64)) > 0)  # ascending global order kept
        tips = sharded.infer_tips(g)[owner[comp] == r]
        assert np.array_equal(tips, sharded.infer_tips(sub))  # a shard's tips are its own links' tips
        back, tips2, ids2, total = sharded.unpack_shard(sharded.pack_shard(sub, tips, ids, nc))
        assert total == nc and np.array_equal(ids2, ids) and np.array_equal(tips2, tips)
        for f in ("vid", "v1", "v2", "s1", "s2"):
            assert np.array_equal(getattr(back, f), getattr(sub, f))
        for k, v in O.decompose(sub).items():
            seen[int(ids[k - 1])] = v
    assert seen == whole
