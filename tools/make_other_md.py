"""profiles/<tag>_other.md from the committed bench line: the secondary workloads, end_to_end, upload, cpu_baseline."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
b = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_hprc-wg_bench.json")))
L = [f"# Round {tag[1:]}: the other workloads of the bench line (`{tag}_hprc-wg_bench.json`, one MI355X)\n",
     "Secondary keys of the default `python bench.py` run (5 timed passes each, 2 for the tangled graph), same context, same code as the headline number.\n",
     "| key | workload | ms/pass | links/s | roofline frac | heaviest stages |", "|---|---|---|---|---|---|"]
for k, v in b["secondary"].items():
    st = sorted(((n, m) for n, m in v["stage_ms"].items() if n != "total"), key=lambda x: -x[1])[:3]
    L.append(f"| `{k}` | {v['workload']} | {v['ms_per_step']:.2f} | {v['value']:.3e} | {v['roofline_frac']:.4f} | " + ", ".join(f"{n} {m:.2f}" for n, m in st) + " |")
e = b["end_to_end"]
L.append(f"\nEnd to end (`end_to_end`): {e['workload']}: {e['wall_s']:.2f} s wall = {e['value']:.3e} links/s with {e['threads']} host threads; "
         f"process start + HIP bring-up alone {e['process_start_and_hip_bringup_s']:.2f} s; the CLI's own stage-cost lines: " +
         ", ".join(f"{k} {v:.0f} ms" for k, v in e["host_ms"].items()) + f" (the GFA text was written beforehand in {e['gfa_written_in_s']:.1f} s, files on {e['files_on']}).\n")
u = b["upload"]
L.append(f"Upload of the headline graph (outside the timed region): wall {u['wall_ms']:.1f} ms = H2D {u['h2d_ms']:.1f} + CSR build {u['csr_build_ms']:.1f} + "
         f"reverse-slot table {u['twin_index_ms']:.1f} (device times by HIP events); `value_whole_job` {b['value_whole_job']:.3e}, "
         f"`pcie_inclusive_value` {b['pcie_inclusive_value']:.3e} links/s.\n")
c = b["cpu_baseline"]
L.append(f"CPU baseline (`cpu_baseline`, kind {c['kind']}, {c['cores']} threads): {c['sample']}; one thread: {c['one_thread_sample']} = {c['value_one_thread']:.3e} links/s.\n")
L.append("The tangled workload: one 2-edge-connected class of ~6e5 sides, walked by ONE WAVE (DESIGN.md section 4, 'Large classes'): "
         "`tree_class_dfs` is all of its pass, at one Infinity-Cache round trip per side.\n")
L.append(f"Fuzz: `{tag}_fuzz.log` (differential run of `tools/fuzz_gpu.py` against the oracle on the final code, all execution modes in rotation, "
         "the laminarity check forced in one of them; the run stops if a pass with exact classes is ever sent to the sequential redo).  "
         f"Multi-GPU probe: `{tag}_shard_probe.json` (`tools/shard_probe.py`).  Redo cost: `tools/redo_cost.py` (DESIGN.md section 4, Row G).\n")
open(os.path.join(ROOT, "profiles", f"{tag}_other.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L))
