"""Sum a rocprofv3 --pmc counter over kernels: python tools/pmc_summary.py <counter_collection.csv> <COUNTER> [passes]
Prints total, per-pass value (divided by `passes`) and the top kernels."""
import csv, sys, collections, re, json
rows = list(csv.DictReader(open(sys.argv[1])))
ctr = sys.argv[2]
passes = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
per = collections.Counter(); tot = 0.0
for r in rows:
    if r.get('Counter_Name') != ctr:
        continue
    v = float(r['Counter_Value']); tot += v
    nm = re.sub(r'\(.*', '', r['Kernel_Name']).replace('povu_hip::', '').replace('void ', '')
    if 'rocprim' in nm:
        nm = "rocprim:radix_sort"
    per[nm] += v
out = {"counter": ctr, "total": tot, "per_pass": tot / passes,
       "top": [[k, v / passes] for k, v in per.most_common(300)]}
print(json.dumps(out))
