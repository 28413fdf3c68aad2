"""Single-GPU probe of the strong-scaling path at full size: partition the whole-genome graph for N ranks on the
device, then load and decompose every shard in turn on the same GPU.  Reports the device time of the partition and,
per shard, CSR build + decompose -- what each rank of an N-GPU job would spend (without the xGMI transfers)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from povu_amd import HipDecomposer, workloads as W
from povu_amd.hip import F_NO_STAGE_TIMES

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
worlds = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["2", "4", "8"])]
g = W.hprc_whole_genome(1e8 * scale)
out = {"links": g.n_links, "segments": g.n_vtx, "runs": []}
# 1-GPU reference on the same box (its own context: the workspace of a whole-genome pass is ~80 GB)
ref = HipDecomposer(0)
ref.upload(g)
f = None
for _ in range(3):
    t = time.perf_counter(); f = ref.decompose(flags=F_NO_STAGE_TIMES); t1 = time.perf_counter() - t
out["single_gpu_ms"] = t1 * 1e3
out["pvst_bytes"] = 14 * sum(f.pvst_sizes())
del f
ref.close()
full, work = HipDecomposer(0), HipDecomposer(0)
full.upload(g)
for world in worlds:
    for rep in range(2):
        t = time.perf_counter(); sh = full.partition(world); tp = (time.perf_counter() - t) * 1e3
    rec = {"world": world, "partition_wall_ms": tp, "partition_device_ms": sh.times(), "shards": []}
    for r in range(world):
        i = sh.info(r)
        best = None
        for rep in range(2):
            t = time.perf_counter(); work.upload_shard(i["device_ptr"], i["bytes"], on_device=True); tu = (time.perf_counter() - t) * 1e3
            t = time.perf_counter(); f = work.decompose_shard(flags=F_NO_STAGE_TIMES); td = (time.perf_counter() - t) * 1e3
            n = len(f); del f
            best = (tu, td)
        rec["shards"].append({"rank": r, "links": i["n_links"], "segments": i["n_vtx"], "components": i["n_components"],
                              "bytes": i["bytes"], "csr_build_ms": best[0], "decompose_ms": best[1], "trees": n})
    worst = max(s["csr_build_ms"] + s["decompose_ms"] for s in rec["shards"])
    rec["critical_path_ms_without_transfers"] = tp + worst
    rec["efficiency_bound_without_transfers"] = out["single_gpu_ms"] / (world * (tp + worst))
    # the other definition (bench.py: value_from_resident_shards): from "every rank's shard CSR resident" to "forest merged
    # on rank 0".  A shard's decompose_ms already contains the copy of ITS PVST arrays to its host over its own PCIe link;
    # with the RCCL gather the root additionally lands the other ranks' blocks through ITS link (modelled at the
    # 53 GB/s this box's device-to-host copies reach), the xGMI hop itself hidden under that
    slow = max(s["decompose_ms"] for s in rec["shards"])
    root_extra_ms = out["pvst_bytes"] * (world - 1) / world / 53e9 * 1e3
    rec["resident_shards"] = {"slowest_decompose_ms": slow, "root_lands_other_blocks_ms_model": root_extra_ms,
                              "efficiency_bound_decompose_only": out["single_gpu_ms"] / (world * slow),
                              "efficiency_model_with_rccl_gather": out["single_gpu_ms"] / (world * (slow + root_extra_ms))}
    out["runs"].append(rec)
    del sh
    print(json.dumps(rec), flush=True)
print(json.dumps(out))
