"""Per-pass kernel table from a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> <passes> [top]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
passes = float(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
def short(n):
    n = re.sub(r"\(.*", "", n).replace("povu_hip::", "").replace("void ", "")
    if "rocprim" in n:
        m = re.search(r"(radix_sort_onesweep|radix_sort_\w+|scan_impl|init_lookback\w*)", n); n = "rocprim " + (m.group(1) if m else "?")
    return n[:50]
agg = {}
for r in rows:
    k = short(r['Name']); a = agg.setdefault(k, [0, 0]); a[0] += int(r['Calls']); a[1] += int(r['TotalDurationNs'])
T = sum(v[1] for v in agg.values()) / passes / 1e6
print('total kernel ms/pass', round(T, 3), 'launches/pass', sum(v[0] for v in agg.values()) / passes)
for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
    print(f"{k:50s} calls/pass {v[0]/passes:6.1f}  ms/pass {v[1]/passes/1e6:8.4f}  {v[1]/passes/1e6/T*100:5.1f}%")
