"""Turn gpurun_out/final/ (tools/collect_profiles.sh) into the committed files under profiles/."""
import csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
bench = json.loads(open(os.path.join(F, "bench.json")).read().strip().splitlines()[-1])
under = json.loads(open(os.path.join(F, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
fetch = json.load(open(os.path.join(F, "fetch_summary.json")))
write = json.load(open(os.path.join(F, "write_summary.json")))
shutil.copy(os.path.join(F, "bench.json"), os.path.join(P, f"{tag}_bench_final.json"))
shutil.copy(os.path.join(F, "bench_under_rocprof.json"), os.path.join(P, f"{tag}_bench_under_rocprof.json"))
ks = sorted(glob.glob(os.path.join(F, "kt", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
shutil.copy(ks, os.path.join(P, f"{tag}_kernel_stats_bench_chain1M.csv"))
E, V = bench["config"]["links_per_gpu"], bench["config"]["segments_per_gpu"]
hbm = (fetch["per_pass"] + write["per_pass"]) * 1024.0
json.dump({"workload": "chain-of-bubbles K=333333", "links": E, "segments": V, "hbm_bytes_per_pass": int(hbm),
           "fetch_size_kb_per_pass": fetch["per_pass"], "write_size_kb_per_pass": write["per_pass"],
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1` "
                   "(4 decompose passes + one upload, divided by 4; tools/collect_profiles.sh). Counter values as reported (KB); "
                   "FETCH_SIZE is not doubled because the accesses are mostly 4-byte gathers, not 16-byte streams "
                   "(MI355X_MICROARCH.md: uncalibrated for other widths); Infinity-Cache hits are counted."},
          open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
rows = list(csv.DictReader(open(ks)))
passes = under["steps"] + under["warmup"] + 1  # + the stage-breakdown pass
tot_ns = sum(int(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
def short(n):
    n = re.sub(r"\(.*", "", n).replace("povu_hip::", "").replace("void ", "")
    if "rocprim" in n:
        m = re.search(r"(radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|scan_impl|init_lookback\w*)", n)
        n = "rocprim " + (m.group(1) if m else "?")
    return n[:60]
tl = open(os.path.join(F, "timeline_summary.txt")).read().strip()
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
L = []
L.append(f"# Round 1 profiles (MI355X, ROCm 7.2) -- regenerate with tools/collect_profiles.sh + tools/make_profile_summary.py\n")
L.append(f"Workload: BASELINE config 2, chain-of-bubbles K=333333 ({V:,} segments / {E:,} links / {bench['config']['flubbles_per_gpu']:,} flubbles).\n")
L.append(f"Bench line (`{tag}_bench_final.json`, default `python bench.py`): **{bench['value']:.3e} links/s, {bench['ms_per_step']:.2f} ms per pass**, "
         f"HIP-event time of the pass {bench['roofline']['ms_per_launch']:.2f} ms, roofline frac {bench['roofline']['frac']:.4f} "
         f"(algorithmic 48E+108V+16F = {alg/1e6:.0f} MB per pass); CPU port on the same box: {bench['cpu_baseline']['value']:.3e} links/s on 1 core.\n")
L.append("## Kernel trace\n")
L.append(f"`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2` ({passes} decompose passes + one upload)\n")
L.append(f"Sum of kernel durations {tot_ns/1e6:.2f} ms over {calls} launches; the last pass of the trace: `{tl}` "
         f"(HIP-event time of a pass in that run: {under['roofline']['ms_per_launch']:.2f} ms, `{tag}_bench_under_rocprof.json`).\n")
L.append("| kernel | calls/pass | us/pass | avg us | % |\n|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:32]:
    L.append(f"| `{short(r['Name'])}` | {int(r['Calls'])/passes:.1f} | {int(r['TotalDurationNs'])/passes/1e3:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
L.append(f"\nFull table: `{tag}_kernel_stats_bench_chain1M.csv`.\n")
L.append("## HBM traffic (PMC)\n")
L.append(f"`rocprofv3 --pmc FETCH_SIZE ...` and `--pmc WRITE_SIZE ...` (separate passes): FETCH_SIZE {fetch['per_pass']*1024/1e9:.2f} GB + WRITE_SIZE "
         f"{write['per_pass']*1024/1e9:.2f} GB = **{hbm/1e9:.2f} GB per pass** against {alg/1e9:.3f} GB algorithmic: ~{hbm/alg:.0f}x "
         f"(= {hbm/1e9/bench['roofline']['ms_per_launch']:.2f} TB/s while the pass runs).  Where the bytes go:\n")
L.append("| kernel | fetch MB/pass | | kernel | write MB/pass |\n|---|---|---|---|---|")
for (a, b), (c, d) in zip(fetch["top"][:10], write["top"][:10]):
    L.append(f"| `{a[:40]}` | {b*1024/1e6:.0f} | | `{c[:40]}` | {d*1024/1e6:.0f} |")
# per kernel: counted bytes / time it runs = how close each kernel is to the HBM roofline (6.3 TB/s achievable)
fmap = {k: v for k, v in fetch["top"]}; wmap = {k: v for k, v in write["top"]}
dur = {}
for r in rows:
    k = short(r["Name"]).replace("rocprim radix_sort_onesweep_iteration", "rocprim:radix_sort").replace("rocprim radix_sort_onesweep_global_offsets", "rocprim:radix_sort")
    dur[k] = dur.get(k, 0.0) + int(r["TotalDurationNs"]) / passes / 1e3
L.append("\nBytes moved per kernel against the time it runs (both per pass; FETCH_SIZE + WRITE_SIZE as counted):\n")
L.append("| kernel | us/pass | MB/pass | TB/s | |\n|---|---|---|---|---|")
tbl = []
for k, us in dur.items():
    mb = (fmap.get(k, 0.0) + wmap.get(k, 0.0)) * 1024 / 1e6
    if us > 15 and mb > 0:
        tbl.append((us, k, mb))
for us, k, mb in sorted(tbl, reverse=True)[:18]:
    tbs = mb / us  # MB per microsecond = TB/s
    note = "bandwidth (of its amplified traffic)" if tbs > 3.0 else ("latency / dependent loads" if tbs < 1.0 else "")
    L.append(f"| `{k[:40]}` | {us:.0f} | {mb:.0f} | {tbs:.2f} | {note} |")
extra = os.path.join(P, f"{tag}_other_workloads.md")
if os.path.exists(extra):
    L.append("\n" + open(extra).read())
open(os.path.join(P, f"{tag}_summary.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L)[:3000])
