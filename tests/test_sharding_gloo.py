"""Multi-process path on CPU (gloo, world_size 2): component scatter from rank 0, per-rank decompose,
PVST gather to rank 0.  The per-rank decompose is the oracle here (no GPU in this container); on the
GPU box the same glue drives HipDecomposer (bench.py, tests/test_gpu_sharded.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from povu_amd import sharded, workloads as W
from test_oracle import dump_component


class _Tree:
    def __init__(self, cid, d):
        self.component_id = cid
        self.a_id, self.z_id, self.parent = d["p_a_id"], d["p_z_id"], d["p_parent"]
        n = len(self.parent)
        self.a_or, self.z_or = d["p_a_or"], d["p_z_or"]
        assert len(self.a_or) == n


class OracleForest:
    """Same surface as povu_amd.hip.Forest (len / tree(i)), backed by the oracle."""

    def __init__(self, links):
        self.trees = []
        c = 0
        while True:
            d = dump_component(links, c)
            if d is None:
                break
            if len(d["p_parent"]):
                self.trees.append(_Tree(c + 1, d))
            c += 1

    def __len__(self):
        return len(self.trees)

    def tree(self, i):
        return self.trees[i]


class OracleForestRaw(OracleForest):
    """... plus the raw-block surface of povu_amd.hip.Forest (one contiguous result block)."""

    def raw(self):
        total = sum(len(t.parent) for t in self.trees)
        pad = lambda n: (n + 63) // 64 * 64  # noqa: E731
        offs = [0, pad(4 * total), 2 * pad(4 * total), 3 * pad(4 * total), 3 * pad(4 * total) + pad(total)]
        block = np.zeros(offs[4] + pad(total) + 64, dtype=np.uint8)
        hdr = np.zeros((len(self.trees), 3), dtype=np.int64)
        first = 0
        for k, t in enumerate(self.trees):
            n = len(t.parent)
            hdr[k] = (t.component_id, n, first)
            block[offs[0] + 4 * first:offs[0] + 4 * (first + n)] = t.a_id.astype(np.uint32).view(np.uint8)
            block[offs[1] + 4 * first:offs[1] + 4 * (first + n)] = t.z_id.astype(np.uint32).view(np.uint8)
            block[offs[2] + 4 * first:offs[2] + 4 * (first + n)] = t.parent.astype(np.uint32).view(np.uint8)
            block[offs[3] + first:offs[3] + first + n] = t.a_or
            block[offs[4] + first:offs[4] + first + n] = t.z_or
            first += n
        return block, total, offs, hdr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_pipelined(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    g = W.hprc_shaped([300, 120, 500, 80], seed=9, tiny=12) if rank == 0 else None
    sub, comp_ids = sharded.scatter_links(g, rank, world, dev)
    pg = sharded.PipelinedGather(rank, world, dev)
    for _ in range(4):  # several steps in flight, as bench.py does
        pg.submit(OracleForestRaw(sub), id_map=comp_ids)
    got = pg.finish()
    if rank == 0:
        torch.save({k: {kk: torch.from_numpy(np.array(vv).astype(np.int64)) for kk, vv in v.items()} for k, v in got.items()},
                   out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gather_world2(tmp_path):
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker_pipelined, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    whole = OracleForest(W.hprc_shaped([300, 120, 500, 80], seed=9, tiny=12))
    assert sorted(got) == [t.component_id for t in whole.trees]
    for t in whole.trees:
        r = got[t.component_id]
        assert np.array_equal(r["a_id"].numpy(), t.a_id) and np.array_equal(r["z_id"].numpy(), t.z_id)
        assert np.array_equal(r["parent"].numpy(), t.parent)
        assert np.array_equal(r["a_or"].numpy(), t.a_or) and np.array_equal(r["z_or"].numpy(), t.z_or)


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    g = W.hprc_shaped([300, 120, 500, 80], seed=9, tiny=12) if rank == 0 else None
    sub, comp_ids = sharded.scatter_links(g, rank, world, dev)
    forest = OracleForest(sub)
    # shard-local component numbering -> global ids
    got = sharded.gather_forest(forest, rank, world, dev, id_map=comp_ids)
    if rank == 0:
        torch.save({k: {kk: torch.from_numpy(vv.astype(np.int64)) for kk, vv in v.items()} for k, v in got.items()},
                   out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_decompose_gather_world2(tmp_path):
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    g = W.hprc_shaped([300, 120, 500, 80], seed=9, tiny=12)
    whole = OracleForest(g)
    assert sorted(got) == [t.component_id for t in whole.trees]
    for t in whole.trees:
        r = got[t.component_id]
        assert np.array_equal(r["a_id"].numpy(), t.a_id) and np.array_equal(r["z_id"].numpy(), t.z_id)
        assert np.array_equal(r["parent"].numpy(), t.parent)
        assert np.array_equal(r["a_or"].numpy(), t.a_or) and np.array_equal(r["z_or"].numpy(), t.z_or)


def test_lpt_assignment_balances_and_is_deterministic():
    w = [800, 10, 10, 700, 650, 5, 300, 300, 40, 40, 40]
    o = sharded.assign_owners(w, 3)
    assert o.tolist() == sharded.assign_owners(w, 3).tolist()
    load = np.bincount(o, weights=np.asarray(w, dtype=float), minlength=3)
    assert load.max() <= 1.25 * sum(w) / 3
    assert sharded.assign_owners(w, 1).tolist() == [0] * len(w)


def test_partition_keeps_global_component_order():
    g = W.hprc_shaped([60, 40, 50], seed=2, tiny=6)
    comp = sharded.component_labels(g)
    nc = int(comp.max()) + 1
    owner = sharded.assign_owners(np.bincount(comp, minlength=nc), 2)
    seen = []
    for r in range(2):
        sub, ids = sharded.partition_links(g, comp, owner, r)
        assert np.all(np.diff(ids) > 0)
        seen += ids.tolist()
        # the shard decomposed alone gives the same PVSTs as those components of the whole graph
        whole = O.decompose(g)
        part = O.decompose(sub)
        for k, text in part.items():
            assert whole[int(ids[k - 1])] == text
    assert sorted(seen) == list(range(1, nc + 1))
