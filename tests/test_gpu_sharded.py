"""Component sharding on the GPU (povu_amd/csrc/hip/shard.hip):
* the device partition against the numpy model of tests/test_sharding_gloo.py, byte for byte;
* every shard loaded and decomposed in turn, forests packed and merged = the whole-graph oracle (also with more
  ranks than components);
* two processes on the one GPU of the test box, shards and forests moved over gloo through host memory;
* the RCCL communicator of the library with a world of one rank (dlopen, ncclCommInitRank, broadcast /
  all-gather of the size tables on the context's stream, scatter and gather degenerate to the root's own shard)."""
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


def _model_shards(g, world):
    comp = sharded.component_labels(g)
    nc = int(comp.max()) + 1
    w = np.bincount(comp, minlength=nc) + np.bincount(comp[g.v1], minlength=nc)
    owner = sharded.assign_owners(w, world)
    tips = sharded.infer_tips(g)
    out = []
    for r in range(world):
        sub, ids = sharded.partition_links(g, comp, owner, r)
        out.append(sharded.pack_shard(sub, tips[owner[comp] == r], ids, nc))
    return out, nc


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("kind", ["hprc", "random", "isolated"])
def test_device_partition_matches_model(world, kind):
    from povu_amd import HipDecomposer
    if kind == "hprc":
        g = _graph()
    elif kind == "random":
        g = W.random_bidirected(3000, 3300, seed=4)
    else:  # mostly isolated vertices and tiny components
        g = W.random_bidirected(2000, 300, seed=9)
    hip = HipDecomposer(0)
    hip.upload(g)
    sh = hip.partition(world)
    want, nc = _model_shards(g, world)
    assert sh.world == world and sh.total_components == nc
    for r in range(world):
        got = sh.export(r)
        assert got.size == want[r].size, (r, got.size, want[r].size)
        assert np.array_equal(got, want[r]), r
    hip.close()


@pytest.mark.parametrize("world", [2, 5, 64])
def test_shards_decomposed_in_turn_merge_to_the_whole(world):
    from povu_amd import HipDecomposer
    g = _graph() if world != 64 else W.hprc_shaped([700, 300], seed=2, tiny=12)  # 64 ranks > components: empty shards
    full, work = HipDecomposer(0), HipDecomposer(0)
    full.upload(g)
    sh = full.partition(world)
    packed = []
    for r in range(world):
        i = sh.info(r)
        work.upload_shard(i["device_ptr"], i["bytes"], on_device=True)
        assert work.shard_total_components() == sh.total_components
        f = work.decompose_shard()
        packed.append(f.pack())
        del f
    merged = work.merge_forests(packed)
    # the merged forest owns its blocks: later work on the context must not disturb it
    work.upload(W.chain_of_bubbles(500))
    for _ in range(3):
        work.decompose()
    assert merged.total_components == sh.total_components
    ids = merged.component_ids()
    assert ids == sorted(ids)
    assert merged.texts() == O.decompose(g)
    # the host-memory route (what another transport ships) loads the same shard
    work.upload_shard(sh.export(0))
    assert work.decompose_shard().texts() == {k: v for k, v in O.decompose(g).items() if k in set(
        sharded.unpack_shard(sh.export(0))[2].tolist())}
    full.close()
    work.close()


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    from povu_amd import HipDecomposer
    work = HipDecomposer(0)
    full = None
    if rank == 0:
        full = HipDecomposer(0)
        full.upload(_graph())
    merged = None
    for _ in range(2):  # a second round reuses every buffer
        sharded.scatter_over_dist(full, work, rank, world, dev)
        f = work.decompose_shard()
        merged = sharded.gather_over_dist(work, f, rank, world, dev)
    if rank == 0:
        work.decompose_shard()  # more work on the context after the gather: the merged forest keeps its own blocks
        torch.save(merged.texts(), out_path)
    dist.barrier()
    work.close()
    if full:
        full.close()
    dist.destroy_process_group()


def test_two_ranks_scatter_hip_decompose_gather(tmp_path):
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert torch.load(out) == O.decompose(_graph())


def test_rccl_communicator_world_of_one():
    from povu_amd import HipDecomposer
    g = _graph()
    full, work = HipDecomposer(0), HipDecomposer(0)
    full.upload(g)
    comm = sharded.ShardComm(work, 0, 1)
    for _ in range(2):
        sh = full.partition(1)
        comm.scatter(sh)
        del sh  # the partition may go once the scatter has returned
        f = work.decompose_shard()
        merged = comm.gather(f)
        assert len(f) == 0  # the root's arrays moved into the merged forest
        assert merged.texts() == O.decompose(g)
    t = comm.times()
    assert t["scatter_ms"] > 0 and t["gather_ms"] > 0
    comm.close()
    full.close()
    work.close()


def test_bench_rehearsal_two_ranks_one_gpu(tmp_path):
    """bench.py --gpus 2 on one GPU (gloo through host memory): the strong-scaling driver end to end."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, POVU_BENCH_BACKEND="gloo", POVU_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--scale", "0.002"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["components"] == 2024 and len(line["shards"]) == 2
    assert sum(s["n_links"] for s in line["shards"]) == line["config"]["links"]
    # both definitions of the rate: from resident shards (`value`: what the N = 1 line's `value` measures) and the whole job
    assert line["value_from_resident_shards"] == line["value"] > line["value_whole_job"] > 0
    assert "shared" in line["config"]["sharding"]  # the gather maps the ranks' blocks, it does not move them


def test_bench_plain_launch_drives_the_gpus_itself(tmp_path):
    """`python bench.py --gpus 2` WITHOUT a launcher must not die: one process drives the ranks through povu_hip_multi_*
    (here both ranks on the one GPU of the test box)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["POVU_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.002"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["transport"] == "same-device"
    assert line["config"]["components"] == 2024 and len(line["shards"]) == 2
    assert sum(s["n_links"] for s in line["shards"]) == line["config"]["links"]
    assert line["value_from_resident_shards"] == line["value"] > 0 and line["value_whole_job"] > 0


def _pvst_block_bytes(entries):
    return 3 * ((entries * 4 + 63) & ~63) + 2 * ((entries + 63) & ~63)


def _shared_worker(rank, world, port, out_path):
    import json
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    from povu_amd import HipDecomposer
    work = HipDecomposer(0)
    tag = [f"t{os.getpid()}" if rank == 0 else None]
    dist.broadcast_object_list(tag, src=0)
    work.share_results(f"{tag[0]}.{rank}")
    full = None
    if rank == 0:
        full = HipDecomposer(0)
        full.upload(_graph())
    keep, merged, rec = None, None, []
    for _ in range(3):  # later rounds reuse the segments
        sharded.scatter_over_dist(full, work, rank, world, dev)
        b0 = work.transfer_bytes()
        f = work.decompose_shard()
        own, trees = sum(f.pvst_sizes()), len(f)
        merged = sharded.gather_shared(work, f, rank, world, dev, tag[0])
        b1 = work.transfer_bytes()
        keep = f  # the root reads this rank's arrays in place until the next gather
        rec.append(dict(rank=rank, own_entries=own, trees=trees, d2h=b1["d2h"] - b0["d2h"], h2d=b1["h2d"] - b0["h2d"],
                        peer=b1["peer_out"] + b1["peer_in"]))
    allrec = [None] * world
    dist.gather_object(rec, allrec if rank == 0 else None, dst=0)
    if rank == 0:
        assert len(keep) == 0  # the root's own trees moved into the merged forest
        texts = merged.texts()  # (reads rank 1's arrays from ITS shared segment)
        json.dump(dict(texts={str(k): v for k, v in texts.items()}, rec=allrec), open(out_path, "w"))
    dist.barrier()  # rank 1 keeps its forest until the root is done reading
    del merged, keep
    work.close()
    if full:
        full.close()
    dist.destroy_process_group()


def test_shared_memory_gather_crosses_pcie_once(tmp_path):
    """Two ranks on the one GPU of the box (gloo for the descriptors): every PVST block crosses PCIe exactly ONCE, over
    the link of the GPU that computed it -- counted per rank by the library (povu_hip_transfer_bytes) around decompose +
    gather: device-to-host = the rank's own block (+ the small counts a pass reads back), host-to-device = the small tables
    of a pass; nothing is sent back to a device, nothing goes to a peer, and the root's traffic does not grow with the
    other ranks' results."""
    import json
    out = str(tmp_path / "shared.json")
    mp.spawn(_shared_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = json.load(open(out))
    assert {int(k): v for k, v in got["texts"].items()} == O.decompose(_graph())
    for per_rank in got["rec"]:
        for r in per_rank:
            blk = _pvst_block_bytes(r["own_entries"])
            slack = 8192 + 64 * r["trees"]  # counts and per-component tables of a pass
            assert 14 * r["own_entries"] <= r["d2h"] <= blk + slack, r
            assert r["h2d"] <= slack, r
            assert r["peer"] == 0, r
    assert all(r["own_entries"] > 0 for per_rank in got["rec"] for r in per_rank)


def _shared_extras_worker(rank, world, port, out_path):
    import json
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    from povu_amd import HipDecomposer
    from povu_amd.hip import F_HAIRPINS, F_LEAF_SUBFLUBBLES, F_SUBFLUBBLES
    work = HipDecomposer(0)
    tag = [f"x{os.getpid()}" if rank == 0 else None]
    dist.broadcast_object_list(tag, src=0)
    work.share_results(f"{tag[0]}.{rank}")
    full = None
    if rank == 0:
        full = HipDecomposer(0)
        full.upload(_graph())
    res, keep = {}, []
    for name, flags in (("leaf", F_LEAF_SUBFLUBBLES), ("all", F_SUBFLUBBLES), ("hairpins", F_HAIRPINS), ("plain", 0), ("all2", F_SUBFLUBBLES)):
        sharded.scatter_over_dist(full, work, rank, world, dev)
        f = work.decompose_shard(flags=flags)
        merged = sharded.gather_shared(work, f, rank, world, dev, tag[0])
        keep.append(f)  # (the root reads this rank's segments in place)
        if rank == 0:
            res[name] = {str(k): v for k, v in merged.texts().items()}
            if name == "hairpins":
                res["hp"] = {str(merged.tree(i).component_id): merged.tree(i).hairpins.tolist() for i in range(len(merged))}
            if name.startswith("all"):
                res[name + "_counts"] = [[merged.subtree(i)[k] for k in ("n_concealed", "n_midi", "n_smothered")] for i in range(len(merged))]
        dist.barrier()  # the other rank keeps its forest (and its extras segment) until the root has read
        del merged
    if rank == 0:
        json.dump(res, open(out_path, "w"))
    del keep
    work.close()
    if full:
        full.close()
    dist.destroy_process_group()


def test_shared_memory_gather_carries_labels_boundaries_and_extended_trees(tmp_path):
    """Two processes (two ranks on the one GPU of the box, gloo for the descriptors): the leaf passes' labels, the hairpin
    boundaries and the extended trees of all five `-s` passes reach the root through a second shared-memory segment per rank
    (povu_hip_forest_share / _attach) -- until round 5 they only moved inside one process.  The merged forests write the same
    PVST text as the oracle (T / O / C / M / S lines included), the boundaries equal a single-GPU pass, and a plain pass between
    two `-s` passes still works (the segments are reused)."""
    import json
    from povu_amd import HipDecomposer
    from povu_amd.hip import F_HAIRPINS
    out = str(tmp_path / "extras.json")
    mp.spawn(_shared_extras_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = json.load(open(out))
    g = _graph()
    as_int = lambda d: {int(k): v for k, v in d.items()}  # noqa: E731
    assert as_int(got["leaf"]) == O.decompose(g, leaf=True)
    want_all = O.decompose(g, leaf=2)
    assert as_int(got["all"]) == want_all and as_int(got["all2"]) == want_all
    assert sum(c[0] for c in got["all_counts"]) > 0
    assert as_int(got["plain"]) == O.decompose(g) == as_int(got["hairpins"])
    one = HipDecomposer(0)
    one.upload(g)
    fh = one.decompose(flags=F_HAIRPINS)
    assert {int(k): v for k, v in got["hp"].items()} == {fh.tree(i).component_id: fh.tree(i).hairpins.tolist() for i in range(len(fh))}
    assert any(v for v in got["hp"].values())
    one.close()


def test_one_process_engine_three_ranks_on_one_device():
    """povu_hip_multi_*: one process, a context + host thread per rank (all three on the one GPU here, shards loaded straight
    from the partition block); the merged forest takes the workers' blocks over."""
    from povu_amd import HipDecomposer
    from povu_amd.hip import F_HAIRPINS, F_LEAF_SUBFLUBBLES, MultiDecomposer
    g = _graph()
    want = O.decompose(g)
    md = MultiDecomposer([0, 0, 0])
    assert md.transport == "same-device"
    md.upload(g)
    for _ in range(2):
        md.scatter()
        f = md.decompose()
        assert f.texts() == want
        ids = f.component_ids()
        assert ids == sorted(ids)
    info = [md.rank_info(r) for r in range(3)]
    assert sum(i["n_links"] for i in info) == g.n_links and sum(i["n_vtx"] for i in info) == g.n_vtx
    assert all(i["peer_in"] == 0 and i["peer_out"] == 0 for i in info)
    # resident shards: decompose again without a scatter; the forest of the step before stays valid
    f2 = md.decompose()
    assert f2.texts() == want and f.texts() == want
    # what does not travel between processes does move inside one: subflubble labels and hairpin boundaries
    assert md.decompose(F_LEAF_SUBFLUBBLES).texts() == O.decompose(g, leaf=True)
    from povu_amd.hip import F_SUBFLUBBLES
    assert md.decompose(F_SUBFLUBBLES).texts() == O.decompose(g, leaf=2)  # all five passes of -s: the extended trees move too
    one = HipDecomposer(0)
    one.upload(g)
    fh1, fhm = one.decompose(flags=F_HAIRPINS), md.decompose(F_HAIRPINS)
    assert fhm.texts() == want
    hp = lambda fo: {fo.tree(i).component_id: fo.tree(i).hairpins.tolist() for i in range(len(fo))}  # noqa: E731
    assert hp(fh1) == hp(fhm)
    one.close()
    del f, f2, fh1, fhm
    md.close()
    # more ranks than components, and a world of one
    small = W.hprc_shaped([700, 300], seed=2, tiny=3)
    for devs in ([0] * 9, [0]):
        md = MultiDecomposer(devs)
        md.upload(small)
        md.scatter(keep_graph=False)
        assert md.decompose().texts() == O.decompose(small)
        md.close()


@pytest.mark.parametrize("transport", ["peer", "rccl"])
def test_one_process_engine_on_two_real_devices(transport, monkeypatch):
    """Both transports of the one-process scatter between two DIFFERENT devices (skipped on a one-GPU box, which only ever
    reaches the same-device path): peer copies (hipMemcpyPeerAsync + an event of the root's device per rank) and RCCL
    send / receive with the receive buffers taken before anything is posted."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from povu_amd.hip import MultiDecomposer
    monkeypatch.setenv("POVU_HIP_MULTI_TRANSPORT", transport)
    g = _graph()
    md = MultiDecomposer([0, 1])
    assert md.transport.startswith("peer-copy" if transport == "peer" else "rccl"), md.transport
    md.upload(g)
    for _ in range(2):
        md.scatter()
        assert md.decompose().texts() == O.decompose(g)
    info = [md.rank_info(r) for r in range(2)]
    assert info[1]["peer_in"] == info[1]["shard_bytes"] * 2 and info[0]["peer_out"] == info[1]["peer_in"]
    md.close()


def test_cli_gpus_flag_workers_write_their_own_files(tmp_path):
    """`povu decompose --gpus 2` (both workers on the one GPU via POVU_HIP_DEVICES): every worker writes the files of its
    components; together they are what the single-GPU run writes."""
    import json
    import subprocess
    from povu_amd import hip as H
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    povu = os.path.join(root, "povu_amd", "bin", "povu")
    g = _graph()
    gfa = str(tmp_path / "g.gfa")
    H.write_gfa(g, gfa)
    outs = {}
    for name, extra, env in (("one", [], {}), ("two", ["--gpus", "2"], {"POVU_HIP_DEVICES": "0,0"}),
                             ("two_leaf", ["--gpus=2", "--leaf-subflubbles"], {"POVU_HIP_DEVICES": "0,0"}),
                             ("two_sub", ["--gpus=2", "-s"], {"POVU_HIP_DEVICES": "0,0"})):
        d = tmp_path / name
        d.mkdir()
        r = subprocess.run([povu, "-t", "4", "decompose", "-i", gfa, "-o", str(d)] + extra, env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1000:]
        outs[name] = {int(p.name[:-5]): p.read_text() for p in d.glob("*.pvst")}
    assert outs["one"] == O.decompose(g) == outs["two"]
    assert outs["two_leaf"] == O.decompose(g, leaf=True)
    assert outs["two_sub"] == O.decompose(g, leaf=2)  # all five passes of -s, every worker on its own components
    # --structure-export with --gpus: every worker renders the sidecar frames of its own components (the state they come
    # from lives in its context), written in component order once all are done -- the file of the single-GPU run
    side = {}
    for name, extra, env in (("one", [], {}), ("two", ["--gpus", "2"], {"POVU_HIP_DEVICES": "0,0"})):
        d = tmp_path / ("sx_" + name)
        d.mkdir()
        sx = d / "export.json"
        r = subprocess.run([povu, "-t", "4", "decompose", "-i", gfa, "-o", str(d), "--structure-export", str(sx)] + extra,
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1000:]
        frames = [json.loads(l) for l in open(str(sx) + ".flubble-debug.jsonl")]
        for fr in frames:  # class ids are names: only which entries share one is the same in every run (a shard numbers its own)
            names = {}
            for e in fr["stack_entries"] + fr["next_seen_table"]:
                e["class_id"] = names.setdefault(e["class_id"], len(names))
        side[name] = frames
        assert {int(p.name[:-5]): p.read_text() for p in d.glob("*.pvst")} == outs["one"]
    assert len(side["one"]) == len(side["two"]) == len(outs["one"])
    for fi, (fa, fb) in enumerate(zip(side["one"], side["two"])):
        for key in fa:
            if isinstance(fa[key], list):
                assert len(fa[key]) == len(fb[key]), (fi, key)
                for ei, (ea, eb) in enumerate(zip(fa[key], fb[key])):
                    assert ea == eb, (fi, key, ei, {k: (ea[k], eb[k]) for k in ea if ea[k] != eb[k]})
            else:
                assert fa[key] == fb[key], (fi, key, fa[key], fb[key])
    r = subprocess.run([povu, "decompose", "-i", gfa, "-o", str(tmp_path), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "device 1 is not visible" in r.stderr  # a one-GPU box has no second device


def test_shards_leave_the_partition_block_without_a_copy():
    """What the torch.distributed scatter sends on GPUs: a zero-copy torch view of the packed shard in the root's HBM
    (CUDA array interface) holds exactly the bytes `export` returns, and a shard loads from a torch CUDA tensor's
    device pointer -- the receiving side of an RCCL recv."""
    from povu_amd import HipDecomposer
    g = _graph()
    want = O.decompose(g)
    full, work = HipDecomposer(0), HipDecomposer(0)
    full.upload(g)
    sh = full.partition(3)
    dev = torch.device("cuda", 0)
    got = {}
    for r in range(3):
        i = sh.info(r)
        t = torch.as_tensor(sharded._DeviceBytes(i["device_ptr"], i["bytes"]), device=dev)
        assert t.dtype == torch.uint8 and t.numel() == i["bytes"] and t.data_ptr() == i["device_ptr"]
        assert np.array_equal(t.cpu().numpy(), sh.export(r))
        landed = t.clone()  # (an RCCL recv would have written these bytes)
        torch.cuda.synchronize()
        work.upload_shard(landed.data_ptr(), landed.numel(), on_device=True)
        got.update(work.decompose_shard().texts())
    assert got == want
    full.close()
    work.close()
