#!/usr/bin/env python3
"""bench.py -- edges/sec decomposed (flubble + PVST) on the MI355X decompose path.

One "step" = one full pass of the hot path (rows B-G: component labelling, re-indexing, spanning
tree, cycle classes, candidate stack, PVST) over the workload, from "graph resident in HBM as CSR"
to "PVST arrays on the host" (SURVEY.md 8d).  At N=1 the workload is BASELINE.json configs[1]:
the synthetic chain-of-bubbles GFA, 1 000 000 segments / 1 999 998 links, one component.  At N>1
every rank owns one such component (components are the sharding unit; weak scaling; no data-path
collective: like the reference's threads, every rank ends with the PVST arrays of its own components
in host memory and would write their files).  `--gather` adds the pipelined PVST gather to rank 0 over
RCCL to the timed region.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def algorithmic_bytes(E, V, F):
    """whole path 48E + 108V + 16F; traversal kernels (rows C-G) 24E + 80V + 16F."""
    return 48 * E + 108 * V + 16 * F, 24 * E + 80 * V + 16 * F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--units", type=int, default=333333, help="bubble units per component (K)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="chain", choices=["chain", "nest", "hprc"])
    ap.add_argument("--gather", action="store_true", help="N > 1: also gather every rank's PVST arrays to rank 0 (pipelined)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    from povu_amd import HipDecomposer, workloads
    from povu_amd.sharded import PipelinedGather

    if args.workload == "chain":
        g = workloads.chain_of_bubbles(args.units)
        wl = f"chain-of-bubbles K={args.units}: {g.n_vtx} segments / {g.n_links} links, 1 component per GPU"
    elif args.workload == "nest":
        g = workloads.nested_towers(1000, max(1, args.units // 100))
        wl = f"nested towers depth 1000 x {max(1, args.units // 100)}: {g.n_vtx} segments / {g.n_links} links"
    else:
        g = workloads.hprc_shaped([args.units * 3], seed=20260612 + rank)
        wl = f"HPRC-shaped backbone {args.units * 3}: {g.n_vtx} segments / {g.n_links} links"

    hip = HipDecomposer(local_rank)
    t_up = time.perf_counter()
    hip.upload(g)  # inputs resident in HBM before the timed region
    upload_s = time.perf_counter() - t_up
    # every rank holds ONE component of the job: its global component id is rank + 1
    id_map = np.array([rank + 1], dtype=np.int64)

    # N > 1: the PVST gather to rank 0 of step k overlaps with the kernels of step k+1 (separate streams);
    # sync() drains it, so the timed region contains every transfer of its K steps
    gather = PipelinedGather(rank, world, comm_dev) if (world > 1 and args.gather) else None

    from povu_amd.hip import F_NO_STAGE_TIMES

    def step():
        # timed passes record only the pass-total HIP events; the per-stage breakdown comes from one extra pass
        f = hip.decompose(flags=F_NO_STAGE_TIMES)
        if gather:
            gather.submit(f, id_map=id_map)
        return f

    def sync():
        if gather:
            gather.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    f = None
    for _ in range(args.warmup):
        f = step()
    stage_acc = {}
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f = step()
        for st in hip.stage_times():
            a = stage_acc.setdefault(st["name"], [0.0, 0])
            a[0] += st["ms"]
            a[1] += 1
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    pass_ms_timed = stage_acc["total"][0] / max(1, stage_acc["total"][1]) if "total" in stage_acc else None
    hip.decompose()  # untimed: per-stage HIP events (a few microseconds each) for the breakdown below
    stage_acc = {st["name"]: [st["ms"], 1] for st in hip.stage_times()}
    if pass_ms_timed is not None:
        stage_acc["total"] = [pass_ms_timed, 1]
    n_flub = sum(f.tree(i).a_id.shape[0] - 1 for i in range(len(f)))
    E, V = g.n_links, g.n_vtx
    total_links = E * world * args.steps
    value = total_links / dt

    if rank == 0:
        stages = {k: v[0] / max(1, v[1]) for k, v in stage_acc.items()}
        dom = max((k for k in stages if k != "total"), key=lambda k: stages[k])
        whole_b, trav_b = algorithmic_bytes(E, V, n_flub)
        # the "kernel" of this path is one decompose pass = ~330 short launches; its duration is the HIP-event
        # time around the whole pass on the library's stream (sum of kernel durations in profiles/ agrees)
        pass_ms = stages.get("total", dt / args.steps * 1e3)
        achieved = whole_b / (pass_ms * 1e-3) / 1e9
        traffic = None
        try:  # HBM bytes per pass from the committed rocprofv3 --pmc runs of this workload (profiles/)
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pm.get("links") == E and world == 1:
                traffic = pm["hbm_bytes_per_pass"]
        except Exception:
            traffic = None
        out = {
            "metric": "edges/sec decomposed (flubble+PVST)",
            "value": value,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": wl, "links_per_gpu": E, "segments_per_gpu": V, "flubbles_per_gpu": n_flub,
                       "sharding": ("one weakly-connected component per GPU; "
                                    + ("pipelined PVST gather to rank 0 over RCCL inside the timed region" if gather else
                                       "every rank keeps (and would write) the PVST of its own components, no data-path collective"))
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "decompose pass (all kernels of rows B-G, one HIP stream)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": whole_b, "ms_per_launch": pass_ms,
                         "formula": "48*E + 108*V + 16*F bytes (SURVEY 8d)",
                         "dominant_stage": {"name": dom, "ms": stages[dom]}},
            "stage_ms": stages,
            "upload_ms": upload_s * 1e3,
            "pcie_inclusive_value": E * world / (dt / args.steps + upload_s),
        }
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib  # CPU oracle: reported baseline only, never the product path
            k = min(args.units, 333333)
            sample = workloads.chain_of_bubbles(k) if args.workload == "chain" else g
            # repeat the one-core port until ~6 s of CPU work have been spent; report the mean pass
            t1 = time.perf_counter()
            reps, cpu_dt = 0, 0.0
            while reps < 3 or (cpu_dt < 6.0 and reps < 50):
                _, info = oracle_lib.decompose(sample, want_text=False, timings=True)
                cpu_dt += info["t_componetize"] + info["t_tree"] + info["t_classes"] + info["t_stack"] + info["t_pvst"]
                reps += 1
            out["cpu_baseline"] = {"value": sample.n_links * reps / cpu_dt, "unit": "edges/s", "cores": 1, "kind": "port",
                                   "sample": f"{sample.n_links} links of the same workload (one component = one thread in the "
                                             f"reference's scheme), {reps} passes, componetize..add_flubbles "
                                             f"{cpu_dt / reps:.2f} s per pass (wall incl. graph build {time.perf_counter() - t1:.1f} s)"}
        print(json.dumps(out))
    hip.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
