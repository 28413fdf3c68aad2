"""Turn gpurun_out/final/<workload>/ (tools/collect_profiles.sh) into the committed files under profiles/:
<tag>_<workload>_bench.json, _bench_under_rocprof.json, _kernel_stats.csv, pmc_traffic.json and <tag>_summary.md."""
import csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # kernel_source_digest
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
PASSES = 4  # bench.py --no-overlap --no-latency-leg --steps 2 --warmup 1: 1 warm-up + 2 timed + 1 stage-breakdown pass, one at a time


def short(n):
    n = re.sub(r"\(.*", "", n).replace("povu_hip::", "").replace("void ", "")
    if "rocprim" in n:
        m = re.search(r"(radix_sort_onesweep|radix_sort_\w+|scan_impl|init_lookback\w*)", n)
        n = "rocprim:" + (m.group(1) if m else "?")
    return n[:60]


traffic = []
L = [f"# Round {tag[1:]} profiles (MI355X, ROCm 7.2) -- regenerate with tools/collect_profiles.sh + tools/make_profile_summary.py\n"]
for wl, title in (("hprc-wg", "BASELINE config 4 at full size (the headline workload of bench.py)"), ("chain", "BASELINE config 2")):
    D = os.path.join(F, wl)
    if not os.path.isdir(D):
        continue
    bench_line = json.loads(open(os.path.join(D, "bench.json")).read().strip().splitlines()[-1])
    under = json.loads(open(os.path.join(D, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
    fetch = json.load(open(os.path.join(D, "fetch_summary.json")))
    write = json.load(open(os.path.join(D, "write_summary.json")))
    for src, dst in (("bench.json", "bench.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"), ("kernel_stats.csv", "kernel_stats.csv")):
        shutil.copy(os.path.join(D, src), os.path.join(P, f"{tag}_{wl}_{dst}"))
    E, V = bench_line["config"]["links"], bench_line["config"]["segments"]
    rows = list(csv.DictReader(open(os.path.join(D, "kernel_stats.csv"))))
    # calibration of FETCH_SIZE on kernels of this very run whose bytes are known: the 128-bit xor scan of the bridge
    # test reads its input exactly once in each of its two kernels (k_xor128_partials with 16-byte-per-lane loads,
    # k_xor128_chunks with 32-byte-per-lane loads) and k_xor128_chunks writes as many bytes as it reads (WRITE_SIZE is
    # exact for streaming stores of whole lines, MI355X_MICROARCH.md)
    fmap = {k: v for k, v in fetch["top"]}
    wmap = {k: v for k, v in write["top"]}
    cal = {}
    if fmap.get("k_xor128_chunks") and wmap.get("k_xor128_chunks"):
        known = wmap["k_xor128_chunks"]
        cal = {"read_32B_per_lane": fmap["k_xor128_chunks"] / known, "read_16B_per_lane": fmap.get("k_xor128_partials", 0) / known}
    raw = (fetch["per_pass"] + write["per_pass"]) * 1024.0
    # corrected: the guide's gfx950 rule (coalesced reads are counted at one half) holds for wide AND for narrow per
    # lane loads here (both calibrate at ~0.5), so FETCH_SIZE is doubled as a whole; scattered 4-byte gathers are
    # uncalibrated and may be over- or under-stated by this
    corrected = (2 * fetch["per_pass"] + write["per_pass"]) * 1024.0
    traffic.append({"workload": bench_line["config"]["workload"], "links": E, "segments": V, "hbm_bytes_per_pass": int(corrected),
                    "raw_counter_bytes_per_pass": int(raw), "fetch_size_kb_per_pass": fetch["per_pass"],
                    "write_size_kb_per_pass": write["per_pass"], "fetch_calibration": cal,
                    "kernel_source_digest": bench.kernel_source_digest(),
                    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `bench.py --workload " + wl +
                            " --steps 2 --warmup 1` (4 decompose passes + one upload, divided by 4). Counter values are KB. "
                            "Correction per MI355X_MICROARCH.md (gfx950 counts coalesced streaming reads at half): FETCH_SIZE is "
                            "doubled -- fetch_calibration = counted / known bytes on the xor-scan kernels of this run shows ~0.5 "
                            "for wide and for narrow per lane loads alike; WRITE_SIZE as counted; scattered gathers are "
                            "uncalibrated; Infinity-Cache hits are counted."})
    passes = PASSES
    tot_ns = sum(int(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    alg = bench_line["roofline"]["algorithmic_bytes_per_launch"]
    L.append(f"## {title}\n")
    L.append(f"Workload: {bench_line['config']['workload']}.\n")
    cb = bench_line.get("cpu_baseline")
    L.append(f"Bench line (`{tag}_{wl}_bench.json`): **{bench_line['value']:.3e} links/s, {bench_line['ms_per_step']:.2f} ms per pass**, HIP-event time of "
             f"the pass {bench_line['roofline']['ms_per_launch']:.2f} ms, roofline frac {bench_line['roofline']['frac']:.4f} (algorithmic 48E+108V+16F = "
             f"{alg/1e9:.3f} GB per pass)" + (f"; CPU port on the box: {cb['value']:.3e} links/s on {cb['cores']} threads (reference scheme), "
                                              f"{cb['value_lpt_threads']:.3e} bin-packed, {cb['value_one_thread']:.3e} on one thread ({cb.get('one_thread_sample', '')})" if cb else "") + ".\n")
    L.append(f"### Kernel trace\n\n`rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {wl} --no-cpu-baseline --no-secondary --no-overlap --no-latency-leg --steps 2 --warmup 1` "
             f"({passes} passes + one upload): sum of kernel durations {tot_ns/1e6:.2f} ms over {calls} launches; the last pass: "
             f"`{open(os.path.join(D, 'timeline_summary.txt')).read().strip()}` (HIP-event time of a pass in that run: {under['roofline']['ms_per_launch']:.2f} ms).\n")
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0])
        a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"])
    T = sum(v[1] for v in agg.values())
    L.append("| kernel | calls/pass | ms/pass | avg us | % |\n|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:30]:
        L.append(f"| `{k}` | {v[0]/passes:.1f} | {v[1]/passes/1e6:.3f} | {v[1]/max(1,v[0])/1e3:.1f} | {v[1]/T*100:.1f} |")
    L.append(f"\nFull table: `{tag}_{wl}_kernel_stats.csv`.\n")
    L.append(f"### HBM traffic (PMC)\n\nFETCH_SIZE {fetch['per_pass']*1024/1e9:.2f} GB + WRITE_SIZE {write['per_pass']*1024/1e9:.2f} GB = {raw/1e9:.2f} GB per pass as counted; "
             f"with FETCH_SIZE doubled (gfx950 counts coalesced reads at half, see `pmc_traffic.json`) **{corrected/1e9:.2f} GB per pass** against {alg/1e9:.3f} GB algorithmic: "
             f"{corrected/alg:.1f}x (= {corrected/1e9/bench_line['roofline']['ms_per_launch']:.2f} TB/s while the pass runs).  Calibration on the scan kernels: {json.dumps(cal)}.\n")
    L.append("| kernel | fetch MB/pass | | kernel | write MB/pass |\n|---|---|---|---|---|")
    for (a, b), (c, d) in zip(fetch["top"][:12], write["top"][:12]):
        L.append(f"| `{a[:40]}` | {b*1024/1e6:.0f} | | `{c[:40]}` | {d*1024/1e6:.0f} |")
    L.append("")
# ---- kernel -> bytes, the round before against this one (the headline workload): the top fetchers / writers of either round
prev = os.path.join(P, f"r{int(tag[1:]) - 1:02d}_summary.md")
D = os.path.join(F, "hprc-wg")
if os.path.exists(prev) and os.path.isdir(D):
    old_f, old_w = {}, {}
    sect = 0
    for line in open(prev):
        if line.startswith("## "):
            sect += 1
        m = re.match(r"\| `([^`]+)` \| (\d+) \| \| `([^`]+)` \| (\d+) \|", line)
        if m and sect == 1:  # (the first section is the headline workload)
            old_f[m.group(1)] = int(m.group(2))
            old_w[m.group(3)] = int(m.group(4))
    fetch = json.load(open(os.path.join(D, "fetch_summary.json")))
    write = json.load(open(os.path.join(D, "write_summary.json")))
    new_f = {k[:40]: v * 1024 / 1e6 for k, v in fetch["top"]}
    new_w = {k[:40]: v * 1024 / 1e6 for k, v in write["top"]}
    L.append(f"### Kernel -> bytes, round {int(tag[1:]) - 1} against round {tag[1:]} (headline workload; MB per pass as counted; `-`: not among that round's top twelve / gone)\n")
    L.append("| kernel | fetch before | fetch now | write before | write now |\n|---|---|---|---|---|")
    names = list(dict.fromkeys(list(old_f) + list(old_w) + [k for k, _ in sorted(new_f.items(), key=lambda x: -x[1])[:12]] + [k for k, _ in sorted(new_w.items(), key=lambda x: -x[1])[:12]]))
    fmt = lambda d, k: (f"{d[k]:.0f}" if k in d else "-")  # noqa: E731
    for k in names:
        L.append(f"| `{k}` | {fmt(old_f, k)} | {fmt(new_f, k)} | {fmt(old_w, k)} | {fmt(new_w, k)} |")
    L.append("")
# ---- the workloads the class walks dominate: kernel trace of the walk that ships
for wl, title in (("nest", "BASELINE config 5 (nested towers)"), ("tangled", "tangled HPRC shape"), ("circular", "circular (tip-less) component")):
    D = os.path.join(F, wl)
    if not os.path.isdir(D) or not os.path.exists(os.path.join(D, "kernel_stats.csv")):
        continue
    shutil.copy(os.path.join(D, "kernel_stats.csv"), os.path.join(P, f"{tag}_{wl}_kernel_stats.csv"))
    under = json.loads(open(os.path.join(D, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
    rows = list(csv.DictReader(open(os.path.join(D, "kernel_stats.csv"))))
    agg = {}
    for r in rows:
        a = agg.setdefault(short(r["Name"]), [0, 0])
        a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"])
    L.append(f"## {title}: kernel trace of the shipped walk\n")
    L.append(f"Workload: {under['config']['workload']}.  `rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {wl} --no-cpu-baseline --no-secondary "
             f"--no-overlap --no-latency-leg --steps 2 --warmup 1` ({PASSES} passes): {under['ms_per_step']:.2f} ms per pass under the tracer; full table `{tag}_{wl}_kernel_stats.csv`.\n")
    L.append("| kernel | calls/pass | ms/pass |\n|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:8]:
        L.append(f"| `{k}` | {v[0]/PASSES:.1f} | {v[1]/PASSES/1e6:.3f} |")
    L.append("")
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
extra = os.path.join(P, f"{tag}_other.md")
if os.path.exists(extra):
    L.append(open(extra).read())
open(os.path.join(P, f"{tag}_summary.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L)[:4000])
