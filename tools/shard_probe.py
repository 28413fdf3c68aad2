"""Single-GPU probe of the strong-scaling path at full size: partition the whole-genome graph for N ranks on the
device, then load and decompose every shard in turn on the same GPU.  Reports the device time of the partition and,
per shard, CSR build + decompose -- what each rank of an N-GPU job would spend (without the xGMI transfers) -- and the
model of the N-GPU line that follows: from resident shards a step is the slowest rank's pass (passes issued back to back,
as bench.py times them at every N); the gather moves nothing (every rank's PVST block lands in host memory over its own
PCIe link: one address space, or shared memory + one 64-byte descriptor per rank)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from povu_amd import HipDecomposer, workloads as W
from povu_amd.hip import F_NO_STAGE_TIMES, F_ASYNC

def passes_ms(d, fn, steps=6):
    """ms per pass, passes back to back (the copy of a pass's PVST arrays under the kernels of the next), and one at a time"""
    keep = [fn(F_NO_STAGE_TIMES) for _ in range(3)]; del keep
    t = time.perf_counter(); prev = None
    for _ in range(steps):
        f = fn(F_NO_STAGE_TIMES | F_ASYNC)
        if prev is not None: prev.wait()
        prev = f
    prev.wait(); over = (time.perf_counter() - t) / steps * 1e3
    t = time.perf_counter()
    for _ in range(3): f = fn(F_NO_STAGE_TIMES)
    one = (time.perf_counter() - t) / 3 * 1e3
    n = len(f); del f, prev
    return over, one, n

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
worlds = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["2", "4", "8"])]
g = W.hprc_whole_genome(1e8 * scale)
out = {"links": g.n_links, "segments": g.n_vtx, "runs": []}
ref = HipDecomposer(0)   # 1-GPU reference on the same box (its own context)
ref.upload(g)
o1, s1, _ = passes_ms(ref, lambda fl: ref.decompose(flags=fl))
out["single_gpu_ms"] = {"back_to_back": o1, "one_at_a_time": s1}
ref.close()
full, work = HipDecomposer(0), HipDecomposer(0)
full.upload(g)
for world in worlds:
    for rep in range(2):
        t = time.perf_counter(); sh = full.partition(world); tp = (time.perf_counter() - t) * 1e3
    rec = {"world": world, "partition_wall_ms": tp, "partition_device_ms": sh.times(), "shards": []}
    for r in range(world):
        i = sh.info(r)
        for rep in range(2):
            t = time.perf_counter(); work.upload_shard(i["device_ptr"], i["bytes"], on_device=True); tu = (time.perf_counter() - t) * 1e3
        over, one, n = passes_ms(work, lambda fl: work.decompose_shard(flags=fl))
        rec["shards"].append({"rank": r, "links": i["n_links"], "segments": i["n_vtx"], "components": i["n_components"],
                              "bytes": i["bytes"], "csr_build_ms": tu, "decompose_ms_back_to_back": over, "decompose_ms_one_at_a_time": one, "trees": n})
    worst = max(s["csr_build_ms"] + s["decompose_ms_one_at_a_time"] for s in rec["shards"])
    rec["whole_job"] = {"critical_path_ms_without_transfers": tp + worst,
                        "efficiency_bound_without_transfers": s1 / (world * (tp + worst))}
    slow = max(s["decompose_ms_back_to_back"] for s in rec["shards"])
    rec["resident_shards"] = {"slowest_rank_ms": slow, "descriptor_exchange_ms_assumed": 0.1,
                              "efficiency_model": o1 / (world * (slow + 0.1)),
                              "efficiency_model_one_pass_at_a_time": s1 / (world * (max(s["decompose_ms_one_at_a_time"] for s in rec["shards"]) + 0.1))}
    out["runs"].append(rec)
    del sh
    print(json.dumps(rec), flush=True)
print(json.dumps(out))
