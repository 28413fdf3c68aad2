"""profiles/<round>_sq_counters.md from gpurun_out/sq/*.json (tools/pmc_sq.sh) and the round's kernel stats."""
import csv, glob, json, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
d = {}
for f in glob.glob("gpurun_out/sq/*.json"):
    j = json.load(open(f))
    d[j["counter"]] = dict(j["top"])
ms = {}
for r in csv.DictReader(open(f"profiles/{rnd}_hprc-wg_kernel_stats.csv")):
    n = r["Name"].split("(")[0].replace("povu_hip::", "").replace("void ", "")
    ms[n] = float(r["TotalDurationNs"]) / 1e6 / 4
out = [f"# Round {rnd[1:]}: SQ counters per kernel, headline workload", "",
       "`bash tools/pmc_sq.sh` (three `rocprofv3 --pmc` runs of `bench.py --workload hprc-wg --steps 2 --warmup 1`, counters only; per pass =",
       "total / 4).  Read with `DESIGN.md` section 4: the adjacency-walking kernels issue few instructions per wave, are far from",
       "the HBM rate of their own traffic, and spend 25-35 cycles of their CU per vector-memory instruction -- the address unit takes",
       "one cycle per distinct line a wave instruction touches.  `cyc/VMEM` = kernel time x 2.1 GHz x 256 CUs / vector-memory instructions.", "",
       "| kernel | ms | waves | VALU / wave | SALU / wave | VMEM rd / wave | VMEM wr / wave | LDS / wave | cyc / VMEM |", "|---|---|---|---|---|---|---|---|---|"]
ks = sorted(ms, key=lambda k: -ms[k])
for k in ks[:32]:
    g = lambda c: d.get(c, {}).get(k, 0)
    w = g("SQ_WAVES")
    if not w:
        continue
    vm = g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")
    cyc = ms[k] * 1e-3 * 2.1e9 * 256 / vm if vm else 0
    out.append(f"| `{k}` | {ms[k]:.2f} | {w:.0f} | {g('SQ_INSTS_VALU') / w:.0f} | {g('SQ_INSTS_SALU') / w:.0f} | {g('SQ_INSTS_VMEM_RD') / w:.1f} | "
               f"{g('SQ_INSTS_VMEM_WR') / w:.1f} | {g('SQ_INSTS_LDS') / w:.0f} | {cyc:.0f} |")
open(f"profiles/{rnd}_sq_counters.md", "w").write("\n".join(out) + "\n")
print("\n".join(out[:14]))
