"""profiles/<tag>_other.md from the committed bench line: the secondary workloads, end_to_end, upload, cpu_baseline."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
b = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_hprc-wg_bench.json")))
L = [f"# Round {tag[1:]}: the other workloads of the bench line (`{tag}_hprc-wg_bench.json`, one MI355X)\n",
     "Secondary keys of the default `python bench.py` run (5 timed passes each, 2 for the tangled and the hub graph), same context, same code as the headline number; "
     "`-s ms` = one decompose with all five subflubble passes (wall clock) where the line carries it.\n",
     "| key | workload | ms/pass | links/s | roofline frac | -s ms | heaviest stages |", "|---|---|---|---|---|---|---|"]
for k, v in b["secondary"].items():
    st = sorted(((n, m) for n, m in v["stage_ms"].items() if n != "total"), key=lambda x: -x[1])[:3]
    sub = f"{v['subflubbles_wall_ms']:.0f}" if "subflubbles_wall_ms" in v else "-"
    L.append(f"| `{k}` | {v['workload']} | {v['ms_per_step']:.2f} | {v['value']:.3e} | {v['roofline_frac']:.4f} | {sub} | " + ", ".join(f"{n} {m:.2f}" for n, m in st) + " |")
e = b["end_to_end"]
L.append(f"\nEnd to end (`end_to_end`): {e['workload']}: {e['wall_s']:.2f} s wall (MEDIAN of the runs: {', '.join(f'{x:.2f}' for x in e.get('wall_s_runs', [e['wall_s']]))} s) = {e['value']:.3e} links/s with {e['threads']} host threads; "
         f"process start + HIP bring-up alone {e['process_start_and_hip_bringup_s']:.2f} s; the CLI's own stage-cost lines: " +
         ", ".join(f"{k} {v:.0f} ms" for k, v in e["host_ms"].items()) + f" (the GFA text was written beforehand in {e['gfa_written_in_s']:.1f} s, files on {e['files_on']}).\n")
u = b["upload"]
L.append(f"Upload of the headline graph (outside the timed region): wall {u['wall_ms']:.1f} ms = H2D {u['h2d_ms']:.1f} + CSR build {u['csr_build_ms']:.1f} + "
         f"reverse-slot table {u['twin_index_ms']:.1f} (device times by HIP events); `value_whole_job` {b['value_whole_job']:.3e}, "
         f"`pcie_inclusive_value` {b['pcie_inclusive_value']:.3e} links/s.\n")
c = b["cpu_baseline"]
L.append(f"CPU baseline (`cpu_baseline`, kind {c['kind']}, {c['cores']} threads): {c['sample']}; one thread: {c['one_thread_sample']} = {c['value_one_thread']:.3e} links/s.\n")
L.append("The tangled, circular and hub workloads are each dominated by ONE 2-edge-connected class walked by ONE WAVE (DESIGN.md section 4, 'Large classes'): "
         "`tree_class_dfs` is all of their pass (the hub also pays the dense re-index and one-lane loops over the hub side).  Which path a pass took: " +
         "; ".join(f"{k}: seq_redo {v.get('seq_redo')}, black_only_classes {v.get('black_only_classes')}, laminar_check_ran {v.get('laminar_check_ran')}" for k, v in b["secondary"].items()) + ".\n")
L.append(f"One pass at a time: {b.get('ms_per_step_one_pass_at_a_time', float('nan')):.2f} ms per pass (the headline's {b['ms_per_step']:.2f} ms are passes issued back to back, "
         f"POVU_HIP_F_ASYNC); latency of a pass in the overlapped run {b['roofline'].get('pass_latency_ms', float('nan')):.2f} ms = roofline fraction "
         f"{b['roofline'].get('frac_single_call', float('nan')):.4f} for a single call (`roofline.frac_single_call`).  `-s` on the headline workload: "
         f"{b.get('subflubbles', {}).get('wall_ms', float('nan')):.0f} ms per decompose.\n")
L.append(f"Fuzz: `{tag}_fuzz_*.log` (differential runs of `tools/fuzz_gpu.py` against the oracle, all execution modes in rotation, "
         "the laminarity check forced in one of them; a run stops if a pass with exact classes ever meets crossing intervals, or if anything is redone).  "
         f"Multi-GPU probe: `{tag}_shard_probe.json` (`tools/shard_probe.py`).  Crossing intervals at size: `{tag}_redo_cost.log` (`tools/redo_cost.py`).  "
         f"The wave walk: `{tag}_nest_kernel_stats.csv`, `{tag}_tangled_kernel_stats.csv`, `{tag}_circular_kernel_stats.csv` (kernel traces of the shipped walk), `{tag}_walk_stats.log` (its own counters).\n")
open(os.path.join(ROOT, "profiles", f"{tag}_other.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L))
