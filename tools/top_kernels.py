"""The heaviest kernels of a rocprofv3 kernel_stats.csv: python tools/top_kernels.py <csv> [n]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("povu_hip::", "").replace("void ", "")
    print(f"{n[:50]:50s} calls {r['Calls']:>4s} total ms {int(r['TotalDurationNs']) / 1e6:9.2f}")
