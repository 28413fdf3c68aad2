#!/usr/bin/env python3
"""Per-kernel ms per pass of two rocprofv3 kernel_stats.csv files side by side (the pass count of each is taken from the
calls of a kernel that runs once a pass).  python tools/kdiff.py <a.csv> <b.csv> [n]"""
import csv, re, sys
def load(p):
    d, calls = {}, {}
    for r in csv.DictReader(open(p)):
        nm = re.sub(r'\(.*', '', r['Name']).replace('povu_hip::', '').replace('void ', '')
        d[nm] = d.get(nm, 0) + float(r['TotalDurationNs']) / 1e6
        calls[nm] = calls.get(nm, 0) + int(r['Calls'])
    passes = calls.get('k_tour_ends') or calls.get('k_uf_tiles') or 1
    return {k: v / passes for k, v in d.items()}, passes
a, pa = load(sys.argv[1]); b, pb = load(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
skip = ('k_side_degree', 'k_slot_fill', 'k_slot_other', 'k_slot_twin', 'k_side_sort', 'k_max_u32', 'k_vertex_degree', 'k_infer_tips')  # upload
sa = sum(v for k, v in a.items() if k not in skip); sb = sum(v for k, v in b.items() if k not in skip)
print(f"passes {pa} / {pb}; kernel sum per pass (without the upload's) {sa:.3f} -> {sb:.3f} ms")
for k in sorted((set(a) | set(b)) - set(skip), key=lambda k: -abs(a.get(k, 0) - b.get(k, 0)))[:n]:
    print(f"{k[:52]:52s} {a.get(k, 0):8.3f} {b.get(k, 0):8.3f} {b.get(k, 0) - a.get(k, 0):+8.3f}")
