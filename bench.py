#!/usr/bin/env python3
"""bench.py -- edges/sec decomposed (flubble + PVST) on the MI355X decompose path.

One "step" = one full pass of the hot path (rows B-G: component labelling, re-indexing, spanning
tree, cycle classes, candidate stack, PVST) over the workload, from "graph resident in HBM as CSR"
to "PVST arrays on the host" (SURVEY.md 8d).

N = 1   the workload is the configuration BASELINE.json's metric is quoted on: the HPRC-shaped
        whole-genome graph (configs[3]) at full size -- 24 chromosome-sized components + 2 000 tiny
        ones, 99.9 M segments / 122.4 M links; it fits one MI355X (~190 GB of the 288 GB).  Configs
        2 / 3 / 5 are timed briefly afterwards and reported as secondary keys of the same line.
N > 1   STRONG scaling of the same graph (`povu_amd/sharded.py`): rank 0 holds the graph; every step
        it labels the components on its GPU, bin-packs them over the ranks (LPT), partitions the
        links on the device and scatters the shards (RCCL send/recv over xGMI); every rank builds the
        CSR of its shard and decomposes it; the PVST arrays are gathered to rank 0.  Scatter and
        gather are inside the timed region.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def algorithmic_bytes(E, V, F):
    """whole path 48E + 108V + 16F bytes."""
    return 48 * E + 108 * V + 16 * F


def kernel_source_digest():
    """sha256 over the HIP sources: profiles/pmc_traffic.json records the digest it was measured with."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "povu_amd", "csrc", "hip")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".inc")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def build_workload(name, scale):
    from povu_amd import workloads as W
    if name == "hprc-wg":
        g = W.hprc_whole_genome(1e8 * scale)
        return g, (f"BASELINE config 4 at full size: HPRC-shaped whole genome, 24 chromosome-sized components + 2000 tiny, "
                   f"{g.n_vtx} segments / {g.n_links} links" + ("" if scale == 1.0 else f" (scale {scale})"))
    if name == "chain":
        k = max(1, int(333333 * scale))
        g = W.chain_of_bubbles(k)
        return g, f"BASELINE config 2: chain-of-bubbles K={k}, {g.n_vtx} segments / {g.n_links} links, 1 component"
    if name == "hprc-chr":
        n = max(8, int(4e6 * scale))
        g = W.hprc_shaped([n], seed=20260612)
        return g, f"BASELINE config 3 shape: HPRC-shaped single component, backbone {n}, {g.n_vtx} segments / {g.n_links} links"
    if name == "tangled":
        n = max(1000, int(3e6 * scale))
        g = W.hprc_tangled(n, tangle_every=100000, max_tangle=300000)
        return g, (f"HPRC-shaped with heavy-tailed tangles (2-edge-connected blocks of 10^2..3*10^5 segments every ~10^5 backbone "
                   f"segments, links of a tangle in random order): {g.n_vtx} segments / {g.n_links} links")
    if name == "nest":
        t = max(1, int(3333 * scale))
        g = W.nested_towers(1000, t)
        return g, f"BASELINE config 5: nested towers depth 1000 x {t}, {g.n_vtx} segments / {g.n_links} links"
    if name == "circular":
        n = max(8, int(6e5 * scale))
        g = W.hprc_circular(n)
        return g, (f"circular (tip-less) HPRC-shaped component: backbone {n} closed into a ring, {g.n_vtx} segments / {g.n_links} "
                   f"links, no tip: the tree is rooted at (l, vertex 0) and its root gets the 0 -> 0 back edge")
    if name == "hub":
        k = max(10, int(1e5 * scale))
        g = W.hub_on_chain(k, 2 * k)
        return g, f"hub segment: chain-of-bubbles K={k} plus {2 * k} links out of one side of segment 0, {g.n_vtx} segments / {g.n_links} links"
    raise SystemExit(f"unknown workload {name}")


def count_flubbles(forest):
    """PVST vertices that are flubbles (every tree has one root vertex besides them)."""
    return sum(n - 1 for n in forest.pvst_sizes())


def time_single(hip, g, steps, warmup, flags, overlap=True, pool_warm=True):
    """warmup + `steps` timed passes on one context; returns (seconds, HIP-event ms per pass, last forest, mean latency of a
    pass in ms).  With `overlap` the passes are issued back to back (POVU_HIP_F_ASYNC): a pass returns when its forest is laid
    out, and the copy engine moves its PVST arrays over PCIe while the kernels of the next pass run; the timed region ends
    when the LAST pass's arrays are in host memory, so every step's work is inside it.  ms per pass = HIP-event time from
    the first kernel of the first timed pass to the last byte of the last one, divided by the steps."""
    import torch
    from povu_amd.hip import F_ASYNC
    f = None
    # (the warm-up forests are alive together: the timed loop holds up to three result blocks at a time -- first, previous,
    # current -- and the context's pool of page-locked blocks must have them before the clock starts)
    keep = [hip.decompose(flags=flags) for _ in range(max(warmup, 3 if (steps > 1 and pool_warm) else 1))]
    del keep
    torch.cuda.synchronize()
    fl = flags | (F_ASYNC if overlap else 0)
    t0 = time.perf_counter()
    first = prev = None
    lat = 0.0
    for _ in range(steps):
        f = hip.decompose(flags=fl)
        if first is None:
            first = f
        if prev is not None:
            lat += prev.pass_ms()  # (waits for the pass before: long complete by now)
        prev = f
    lat += prev.pass_ms()  # waits for the last pass: its arrays are in host memory
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, first.span_ms(prev) / max(1, steps), f, lat / max(1, steps)


def cpu_baseline(g, wl):
    """The CPU port (oracle) on the host cores of this box.  `value` = the SAME workload the GPU line is quoted on, all
    cores, the reference's own threading scheme; the LPT and one-thread figures are side keys (one thread on a 1/10
    sample: the full graph would take minutes)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib  # CPU oracle: reported baseline only, never the product path
    from povu_amd import workloads as W
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t0 = time.perf_counter()
    _, ref = oracle_lib.decompose(g, want_text=False, timings=True, threads=cores, lpt=False)
    t_ref = ref["t_componetize"] + ref["t_wall_components"]
    _, lpt = oracle_lib.decompose(g, want_text=False, timings=True, threads=cores, lpt=True)
    t_lpt = lpt["t_componetize"] + lpt["t_wall_components"]
    n_comp, used = int(ref["n_comp"]), int(ref["threads"])
    chunk = max(1, n_comp // max(1, used))  # decompose.cpp:78-92: contiguous chunks of n_components / threads
    out = {"value": g.n_links / t_ref, "unit": "edges/s", "cores": used, "kind": "port",
           "scheme": "reference: contiguous chunks of components per thread (decompose.cpp:78-92,116-157)",
           "sample": (f"the benched workload itself ({wl}), componetize..add_flubbles: reference threading scheme on {used} threads "
                      f"({n_comp} components in chunks of {chunk}: thread 0 gets the first {chunk}) {t_ref:.2f} s; components "
                      f"bin-packed by size (LPT) over the same threads {t_lpt:.2f} s"),
           "value_lpt_threads": g.n_links / t_lpt, "seconds": {"reference_scheme": t_ref, "lpt": t_lpt}}
    if g.n_links > 3e7:  # one thread: a 1/10 sample of the same shape, said so
        sample = W.hprc_whole_genome(1e7)
        _, one = oracle_lib.decompose(sample, want_text=False, timings=True, threads=1)
        t_one = one["t_componetize"] + one["t_wall_components"]
        out["value_one_thread"] = sample.n_links / t_one
        out["one_thread_sample"] = f"HPRC-shaped whole genome at 1/10 size ({sample.n_vtx} segments / {sample.n_links} links) {t_one:.2f} s"
    else:
        _, one = oracle_lib.decompose(g, want_text=False, timings=True, threads=1)
        t_one = one["t_componetize"] + one["t_wall_components"]
        out["value_one_thread"] = g.n_links / t_one
        out["one_thread_sample"] = f"the benched workload, {t_one:.2f} s"
    out["wall_s"] = time.perf_counter() - t0
    return out


def end_to_end_cli(g, wl):
    """SURVEY 8d's second number on the HEADLINE workload: `povu decompose` as a user runs it -- a child process that
    starts, brings HIP up, parses the GFA text, uploads + builds the CSR, decomposes, formats and writes the PVST files.
    The GFA is written beforehand by the host library's writer (not timed)."""
    import re
    import shutil
    import subprocess
    import tempfile
    from povu_amd import hip as H
    povu = os.path.join(ROOT, "povu_amd", "bin", "povu")
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 16e9 else None
    d = tempfile.mkdtemp(prefix="povu_e2e_", dir=base)
    try:
        gfa = os.path.join(d, "graph.gfa")
        t0 = time.perf_counter()
        H.write_gfa(g, gfa)
        t_write_gfa = time.perf_counter() - t0
        size = os.path.getsize(gfa)
        threads = str(min(int(os.environ.get("POVU_BENCH_CLI_THREADS", "32")), len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8))
        env = dict(os.environ, POVU_STAGE_COST_TRACE="1")
        # process start + HIP bring-up alone: the CLI on a one-segment graph
        tiny = os.path.join(d, "tiny.gfa")
        with open(tiny, "w") as fh:
            fh.write("H\tVN:Z:1.0\nS\t1\tA\n")
        t_start = None
        for _ in range(2):
            o = os.path.join(d, "tiny_out")
            shutil.rmtree(o, ignore_errors=True)
            os.makedirs(o)
            t0 = time.perf_counter()
            subprocess.run([povu, "decompose", "-i", tiny, "-o", o], capture_output=True, text=True, env=env)
            dt = time.perf_counter() - t0
            t_start = dt if t_start is None else min(t_start, dt)
        best, parts, walls = None, {}, []
        for _ in range(3):
            o = os.path.join(d, "out")
            shutil.rmtree(o, ignore_errors=True)
            os.makedirs(o)
            time.sleep(2.0)  # (untimed: the process before gave tens of gigabytes of device memory back a moment ago, see the note)
            t0 = time.perf_counter()
            r = subprocess.run([povu, "-t", threads, "decompose", "-i", gfa, "-o", o], capture_output=True, text=True, env=env)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": r.stderr[-300:]}
            walls.append(dt)
            if best is None or dt < best:
                best = dt
                parts = {m.group(1): float(m.group(2)) / 1e6 for m in
                         re.finditer(r"contract=host:(\w+) .*?elapsed_ns=(\d+)", r.stderr)}
        outs = [f for f in os.listdir(os.path.join(d, "out")) if f.endswith(".pvst")]
        out_bytes = sum(os.path.getsize(os.path.join(d, "out", f)) for f in outs)
        med = sorted(walls)[len(walls) // 2]
        return {"workload": f"{wl} as GFA text ({size} bytes) -> {len(outs)} .pvst files ({out_bytes} bytes)",
                "wall_s": med, "wall_s_best": best, "wall_s_runs": walls, "value": g.n_links / med, "unit": "edges/s", "threads": int(threads),
                "process_start_and_hip_bringup_s": t_start,
                "value_without_process_start": g.n_links / max(1e-9, med - t_start),
                "host_ms": {k: round(v, 2) for k, v in parts.items()},
                "gfa_written_in_s": t_write_gfa, "files_on": base or tempfile.gettempdir(),
                "note": "MEDIAN of 3 runs of the CLI as a child process, two seconds apart (all in wall_s_runs, the best in wall_s_best; host_ms are the best run's; a CLI process that starts right after "
                        "another one released tens of gigabytes of device memory can spend seconds more in hipMalloc: DESIGN.md "
                        "section 6 -- it still happens to one run in three or so); process_start_and_hip_bringup_s = the same CLI on a "
                        "one-segment graph (process start, library load, HIP context, nothing else); host_ms = the CLI's own "
                        "stage-cost lines (gfa_parse, upload_csr, decompose_call, write_pvst); the GFA text is written "
                        "before the timed runs"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main_one_process(args):
    """--gpus N without a launcher: one process, N GPUs through the library's own engine (povu_hip_multi_*)."""
    from povu_amd.hip import F_NO_STAGE_TIMES, MultiDecomposer, load_lib
    n_dev = load_lib().povu_hip_device_count()
    one = bool(os.environ.get("POVU_BENCH_ONE_DEVICE"))  # rehearsal on a one-GPU box: every rank on device 0
    if not one and n_dev < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {n_dev} HIP devices are visible "
                         f"(POVU_BENCH_ONE_DEVICE=1 rehearses the path on one)")
    md = MultiDecomposer([0] * args.gpus if one else list(range(args.gpus)))
    g, wl = build_workload(args.workload, args.scale)
    E, V = g.n_links, g.n_vtx
    md.upload(g)
    del g
    f = None
    # ---- the whole job: label + LPT + partition on the root, scatter, CSR build and decompose on every GPU
    for _ in range(args.warmup):
        md.scatter()
        f = md.decompose(F_NO_STAGE_TIMES)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        md.scatter()
        f = md.decompose(F_NO_STAGE_TIMES)
    dt = time.perf_counter() - t0
    t_part = md.times()
    ranks_whole = [md.rank_info(r) for r in range(args.gpus)]
    # ---- the N = 1 line's definition: the shards of the last scatter stay resident as CSR; a step = every GPU decomposes its
    # shard and lands its PVST block in host memory (md.decompose returns when every worker has; merging copies nothing)
    # (passes issued back to back, like the N = 1 line's: a step returns while the copy engines still move its PVST arrays, the
    # next step's kernels run under the copies; the clock stops when the LAST step's arrays are in host memory)
    from povu_amd.hip import F_ASYNC
    keep = [md.decompose(F_NO_STAGE_TIMES) for _ in range(3)]  # (the timed loop holds up to three result blocks per rank)
    del keep
    fl = F_NO_STAGE_TIMES | (0 if args.no_overlap else F_ASYNC)
    t0 = time.perf_counter()
    prev = None
    for _ in range(args.steps):
        f = md.decompose(fl)
        if prev is not None:
            prev.wait()
        prev = f
    prev.wait()
    dt_res = time.perf_counter() - t0
    F, n_trees = count_flubbles(f), len(f)
    ranks = [md.rank_info(r) for r in range(args.gpus)]
    alg = algorithmic_bytes(E, V, F)
    step_s = dt_res / args.steps
    w = [r["n_links"] + r["n_vtx"] for r in ranks]
    out = {
        "metric": "edges/sec decomposed (flubble+PVST)",
        "value": E * args.steps / dt_res,
        "unit": "edges/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_s * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": wl, "links": E, "segments": V, "components": n_trees, "flubbles": F,
                   "sharding": "strong scaling, one process: the root GPU labels the components, LPT bin-packing, device partition; "
                               f"shards over xGMI ({md.transport}); every GPU lands its PVST block in host memory over its own PCIe "
                               "link, the merged forest takes the blocks over (no gather transfer)"},
        "roofline": {"bound": "hbm", "kernel": "per-shard decompose pass on every GPU (from resident shards)",
                     "achieved": alg / step_s / 1e9, "peak": HBM_PEAK_GBS * args.gpus, "unit": "GB/s",
                     "frac": alg / step_s / 1e9 / (HBM_PEAK_GBS * args.gpus), "traffic": None,
                     "algorithmic_bytes_per_launch": alg, "ms_per_launch": step_s * 1e3,
                     "formula": "48*E + 108*V + 16*F bytes (SURVEY 8d) over the aggregate peak of all GPUs"},
        "value_from_resident_shards": E * args.steps / dt_res,
        "value_whole_job": E * args.steps / dt,
        "ms_per_step_whole_job": dt / args.steps * 1e3,
        "launch": "one process, one thread + context per GPU (povu_hip_multi_*)" + (" -- REHEARSAL: all ranks on device 0" if one else ""),
        "transport": md.transport,
        "lpt_max_over_mean": max(w) / (sum(w) / len(w)),
        "phase_ms": {"root": {k: t_part[k] for k in ("label_ms", "lpt_ms", "partition_ms", "scatter_wall_ms")},
                     "merge_ms": md.times()["merge_ms"],
                     "per_rank_whole_job": [{k: r[k] for k in ("recv_ms", "csr_ms", "decompose_ms")} for r in ranks_whole],
                     "per_rank_resident": [{"decompose_ms": r["decompose_ms"]} for r in ranks]},
        "shards": [{k: r[k] for k in ("device", "n_vtx", "n_links", "n_components", "shard_bytes")} for r in ranks],
        # bytes per rank since the job began: host->device, device->host (its own PVST blocks), xGMI out / in
        "transfer_bytes": [{k: r[k] for k in ("h2d", "d2h", "peer_out", "peer_in")} for r in ranks],
    }
    del f
    md.close()
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="hprc-wg", choices=["hprc-wg", "chain", "hprc-chr", "nest", "tangled", "circular", "hub"])
    ap.add_argument("--scale", type=float, default=1.0, help="size factor of the workload (1.0 = BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="time the passes one at a time instead of back to back")
    ap.add_argument("--no-latency-leg", action="store_true", help="skip the three extra one-at-a-time passes (profiling runs)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of configs 2 / 3 / 5")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # `python bench.py --gpus N` without a launcher: ONE process drives the N GPUs (povu_hip_multi_*: one context and
        # one host thread per device, the engine behind `povu decompose --gpus N`); nothing here has touched a GPU yet
        return main_one_process(args)
    if args.gpus != world:
        # never report a 1-GPU number as an N-GPU one
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    # POVU_BENCH_BACKEND=gloo + POVU_BENCH_ONE_DEVICE=1 rehearse the N>1 path on a single-GPU box
    backend = os.environ.get("POVU_BENCH_BACKEND", "nccl")
    if os.environ.get("POVU_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    comm_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    from povu_amd import HipDecomposer
    from povu_amd.hip import F_NO_STAGE_TIMES, F_SUBFLUBBLES

    hip = HipDecomposer(local_rank)
    out = None
    if world == 1:
        g, wl = build_workload(args.workload, args.scale)
        t_up = time.perf_counter()
        hip.upload(g)  # inputs resident in HBM before the timed region
        upload_s = time.perf_counter() - t_up
        up = hip.upload_times()
        dt, pass_ms, f, lat_ms = time_single(hip, g, args.steps, args.warmup, F_NO_STAGE_TIMES, overlap=not args.no_overlap,
                                             pool_warm=not args.no_latency_leg)  # (profiling runs: exactly warmup + steps + 1 passes)
        # the same passes one at a time (each complete before the next starts): what a single decompose call costs
        if args.no_latency_leg:
            dt1, pass1_ms = (dt, pass_ms) if args.no_overlap else (float("nan"), float("nan"))
        else:
            dt1, pass1_ms, _f1, _ = time_single(hip, g, 3, 0, F_NO_STAGE_TIMES, overlap=False)
            dt1 = dt1 / 3 * args.steps
            del _f1
        E, V, F = g.n_links, g.n_vtx, count_flubbles(f)
        n_trees = len(f)
        del f
        hip.decompose()  # untimed: per-stage HIP events for the breakdown
        stages = {st["name"]: st["ms"] for st in hip.stage_times()}
        stages["total"] = pass_ms
        dom = max((k for k in stages if k != "total"), key=lambda k: stages[k])
        alg = algorithmic_bytes(E, V, F)
        achieved = alg / (pass_ms * 1e-3) / 1e9
        traffic, traffic_note = None, "no PMC run of this workload under profiles/"
        try:  # HBM bytes per pass from the committed rocprofv3 --pmc runs of this workload (profiles/)
            for pm in json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))):
                if pm.get("links") == E and pm.get("segments") == V:
                    if pm.get("kernel_source_digest") == kernel_source_digest():
                        traffic, traffic_note = pm["hbm_bytes_per_pass"], "profiles/pmc_traffic.json (same kernels, same workload)"
                    else:
                        traffic_note = "profiles/pmc_traffic.json is stale: the kernels changed since it was collected"
        except Exception:
            pass
        step_s = dt / args.steps
        out = {
            "metric": "edges/sec decomposed (flubble+PVST)",
            "value": E * args.steps / dt,
            "unit": "edges/s",
            "n_gpus": 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_s * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": wl, "links": E, "segments": V, "components": n_trees, "flubbles": F, "sharding": "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "decompose pass (all kernels of rows B-G, one HIP stream)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_note": traffic_note, "algorithmic_bytes_per_launch": alg,
                         "ms_per_launch": pass_ms, "formula": "48*E + 108*V + 16*F bytes (SURVEY 8d)",
                         "launch": ("passes issued back to back: the copy of a pass's PVST arrays to the host (PCIe) overlaps the "
                                    "kernels of the next pass; ms_per_launch = HIP events from the first kernel of the first timed "
                                    "pass to the last byte of the last, / steps" if not args.no_overlap else
                                    "one pass at a time"),
                         "pass_latency_ms": lat_ms, "ms_per_launch_one_pass_at_a_time": pass1_ms,
                         # what ONE call of povu_hip_decompose (the CLI, the FFI) reaches: the same bytes over the latency of a pass
                         "frac_single_call": alg / (lat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "dominant_stage": {"name": dom, "ms": stages[dom]}},
            "stage_ms": stages,
            # what happens before the timed region (povu_hip_graph_upload), split by HIP events:
            "upload": {"wall_ms": upload_s * 1e3, "h2d_ms": up["h2d_ms"], "csr_build_ms": up["csr_ms"],
                       "twin_index_ms": up["twin_ms"],
                       "note": "csr_build = side degrees, offsets, adjacency sorted per side, other-end table, tips; "
                               "twin_index = reverse-slot table (atwin); both device time"},
            "value_incl_index_build": E / (step_s + (up["csr_ms"] + up["twin_ms"]) * 1e-3),
            # the two definitions the --gpus N line reports, at N = 1: from the resident CSR (= `value`), and the whole job from
            # the link arrays in HBM (CSR build + decompose; at N > 1 also label + partition + scatter)
            "value_from_resident_shards": E * args.steps / dt,
            "value_whole_job": E / (step_s + (up["csr_ms"] + up["twin_ms"]) * 1e-3),
            "pcie_inclusive_value": E / (step_s + upload_s),
            "ms_per_step_one_pass_at_a_time": dt1 / args.steps * 1e3,
        }
        if not args.no_secondary and args.workload == "hprc-wg":
            # `povu decompose -s` on the same graph: all five subflubble passes on top of the pass (not part of `value`)
            for _ in range(3):  # (untimed: the first call sizes the stage's arena, the second reserves it, the third finds its page-locked blocks in the pool)
                hip.decompose(flags=F_SUBFLUBBLES)
            walls = []
            for _ in range(3):  # as the CLI calls it (no stage timers); the MEDIAN of three
                t_s = time.perf_counter()
                f_s = hip.decompose(flags=F_SUBFLUBBLES | F_NO_STAGE_TIMES)
                walls.append(time.perf_counter() - t_s)
                del f_s
            dt_s = sorted(walls)[1]
            f_s = hip.decompose(flags=F_SUBFLUBBLES)  # once more with stage timers, for the split (one pass at a time, every stage synchronised)
            st_s = {st["name"]: round(st["ms"], 3) for st in hip.stage_times()}
            kinds = [0, 0, 0]
            for i in range(len(f_s)):
                sub_t = f_s.subtree(i)
                kinds = [kinds[0] + sub_t["n_concealed"], kinds[1] + sub_t["n_midi"], kinds[2] + sub_t["n_smothered"]]
            out["subflubbles"] = {"wall_ms": dt_s * 1e3, "wall_ms_runs": [round(w * 1e3, 2) for w in walls], "leaf_passes_ms": st_s.get("leaf_subflubbles"),
                                  "inserting_passes_ms": st_s.get("subflubbles_insert"), "pass_total_ms": st_s.get("total"),
                                  "concealed": int(kinds[0]), "midi": int(kinds[1]), "smothered": int(kinds[2]),
                                  "note": "POVU_HIP_F_SUBFLUBBLES: find_tiny, find_parallel, find_concealed, find_midi, find_smothered on "
                                          "every PVST, extended trees copied to the host; parity unpinned (DESIGN.md section 4)"}
            del f_s
            sec = {}
            for key, name, k_steps in (("config2_chain_1M", "chain", 5), ("config3_hprc_chr", "hprc-chr", 5), ("config5_nest_10M", "nest", 5),
                                      ("tangled_hprc_shape", "tangled", 2), ("circular_1M", "circular", 5), ("hub_300k", "hub", 2)):
                g2, wl2 = build_workload(name, 1.0)
                hip.upload(g2)
                dt2, ms2, f2, lat2 = time_single(hip, g2, k_steps, 1 if k_steps < 5 else 2, F_NO_STAGE_TIMES, overlap=not args.no_overlap)
                a2 = algorithmic_bytes(g2.n_links, g2.n_vtx, count_flubbles(f2))
                hip.decompose()  # untimed: per-stage HIP events
                sec[key] = {"workload": wl2, "value": g2.n_links * k_steps / dt2, "ms_per_step": dt2 / k_steps * 1e3, "ms_per_launch": ms2,
                            "roofline_frac": a2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "pass_latency_ms": lat2,
                            # which path the pass took: components redone by the one-lane stack machine, whether the class pass
                            # ran over the black tree edges only, whether the laminarity check ran ("Row G")
                            "seq_redo": hip.seq_redo_count(), "black_only_classes": hip.last_black_only_classes(),
                            "laminar_check_ran": hip.last_laminar_check_ran(),
                            "stage_ms": {st["name"]: round(st["ms"], 4) for st in hip.stage_times()}}
                if name in ("tangled", "circular", "hub"):
                    # all five passes of -s where the reference's bracket table would hold 1e10 entries
                    hip.decompose(flags=F_SUBFLUBBLES)
                    t_s = time.perf_counter()
                    f_s = hip.decompose(flags=F_SUBFLUBBLES)
                    sec[key]["subflubbles_wall_ms"] = (time.perf_counter() - t_s) * 1e3
                    sec[key]["subflubbles_concealed"] = int(sum(f_s.subtree(i)["n_concealed"] for i in range(len(f_s))))
                    del f_s
                del f2, g2
            out["secondary"] = sec
        if not args.no_secondary:
            hip.close()  # the CLI child takes the GPU memory this context holds
            out["end_to_end"] = end_to_end_cli(g, wl)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(g, wl)
    else:
        from povu_amd.sharded import ShardedBench
        sb = ShardedBench(hip, rank, world, comm_dev, lambda: build_workload(args.workload, args.scale), device_index=local_rank)
        for _ in range(args.warmup):
            sb.step()
        sb.sync()
        sb.reset_phases()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sb.step()
        sb.sync()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # second measurement, same start / end as the N = 1 line: the shards the last step scattered stay resident (CSR built),
        # a step = per-shard decompose + gather to rank 0
        sb.step_resident()
        sb.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sb.step_resident()
        sb.sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_res = float(t.item())
        info = sb.summary()  # collective: per-rank loads and phase times
        if rank == 0:
            E, V, F = info["links"], info["segments"], info["flubbles"]
            alg = algorithmic_bytes(E, V, F)
            step_s = dt / args.steps
            achieved = alg / step_s / 1e9
            out = {
                "metric": "edges/sec decomposed (flubble+PVST)",
                # the N = 1 line's definition: from "CSR resident in HBM" (here: every rank's shard) to "PVST arrays on the
                # host", K timed steps; the whole job (label + partition + scatter + CSR build on top) is value_whole_job,
                # like-for-like with the N = 1 line's key of that name
                "value": E * args.steps / dt_res,
                "unit": "edges/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": dt_res / args.steps * 1e3,
                "higher_is_better": True,
                "scaling": "strong",
                "vs_baseline": None,
                "dtype": "u32",
                "data": "synthetic",
                "config": {"workload": info["workload"], "links": E, "segments": V, "components": info["components"], "flubbles": F,
                           "sharding": info["sharding"]},
                "roofline": {"bound": "hbm", "kernel": "per-shard decompose pass on every GPU + gather (from resident shards)",
                             "achieved": alg / (dt_res / args.steps) / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                             "frac": alg / (dt_res / args.steps) / 1e9 / (HBM_PEAK_GBS * world),
                             "traffic": None, "algorithmic_bytes_per_launch": alg, "ms_per_launch": dt_res / args.steps * 1e3,
                             "formula": "48*E + 108*V + 16*F bytes (SURVEY 8d) over the aggregate peak of all GPUs"},
                # like-for-like with the N = 1 line's `value` (graph / shards resident as CSR -> forest on rank 0's host); `value`
                # above is the whole job (label + partition + scatter + CSR build + decompose + gather), like-for-like with
                # the N = 1 line's `value_whole_job`
                "value_from_resident_shards": E * args.steps / dt_res,
                "ms_per_step_from_resident_shards": dt_res / args.steps * 1e3,
                "value_whole_job": E * args.steps / dt,
                "ms_per_step_whole_job": step_s * 1e3,
                "launch": "one process per GPU (torch.distributed launcher)",
                "shards": info["shards"],
                "lpt_max_over_mean": info["lpt_max_over_mean"],
                "phase_ms": info["phase_ms"],
            }
        sb.close()
    if rank == 0:
        print(json.dumps(out))
    hip.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
