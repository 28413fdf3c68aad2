"""Two ranks on the one GPU of the test box (gloo for the exchange, HIP for the decompose): component
scatter from rank 0, per-rank HIP decompose, PVST gather of the raw pinned blocks to rank 0."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from povu_amd import sharded, workloads as W

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph():
    return W.hprc_shaped([4000, 1500, 2500, 900], seed=17, tiny=30)


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    from povu_amd import HipDecomposer
    g = _graph() if rank == 0 else None
    sub, comp_ids = sharded.scatter_links(g, rank, world, dev)
    hip = HipDecomposer(0)
    hip.upload(sub)
    pg = sharded.PipelinedGather(rank, world, dev)
    for _ in range(3):  # the gather of one step runs under the next decompose
        pg.submit(hip.decompose(), id_map=comp_ids)
    got = pg.finish()
    again = sharded.gather_forest(hip.decompose(), rank, world, dev, id_map=comp_ids)
    if rank == 0:
        assert sorted(got) == sorted(again)
        for k in got:
            for f in ("a_id", "z_id", "parent", "a_or", "z_or"):
                assert np.array_equal(np.array(got[k][f]), np.array(again[k][f]))
    if rank == 0:
        torch.save({k: {kk: torch.from_numpy(np.array(vv).astype(np.int64)) for kk, vv in v.items()} for k, v in got.items()},
                   out_path)
    dist.barrier()
    hip.close()
    dist.destroy_process_group()


def test_two_ranks_scatter_hip_decompose_gather(tmp_path):
    import ctypes as C
    from povu_amd import hip as H
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    want = O.decompose(_graph())
    assert sorted(got) == sorted(want)
    hl = H.load_lib()
    hl.povu_hip_pvst_format.restype = C.c_void_p
    hl.povu_hip_pvst_format.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_size_t)]
    for cid, arrs in got.items():
        a = arrs["a_id"].numpy().astype(np.uint32)
        z = arrs["z_id"].numpy().astype(np.uint32)
        p = arrs["parent"].numpy().astype(np.uint32)
        ao = arrs["a_or"].numpy().astype(np.uint8)
        zo = arrs["z_or"].numpy().astype(np.uint8)
        ln = C.c_size_t(0)
        ptr = hl.povu_hip_pvst_format(len(a), a.ctypes.data, z.ctypes.data, ao.ctypes.data, zo.ctypes.data, p.ctypes.data,
                                      C.byref(ln))
        text = C.string_at(ptr, ln.value).decode()
        hl.povu_hip_buffer_free(ptr)
        assert text == want[cid], cid
