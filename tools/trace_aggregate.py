"""Per-kernel totals of the LAST decompose pass in a rocprofv3 --kernel-trace CSV."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
starts = [i for i, n in enumerate(names) if 'k_uf_tiles' in n]
seg = rows[starts[-1]:]
agg = collections.OrderedDict(); tot = 0
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    nm = re.sub(r'\(.*', '', r['Kernel_Name']).replace('povu_hip::', '').replace('void ', '')
    if 'rocprim' in nm:
        m = re.search(r'(radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|scan_impl|init_lookback\w*|histogram\w*)', r['Kernel_Name']); nm = 'rocprim:' + (m.group(1) if m else '?')
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = agg.setdefault(nm, [0, 0, int(r['Start_Timestamp'])]); a[0] += d; a[1] += 1; tot += d
print('launches', len(seg), 'total kernel ms', tot / 1e6, 'span ms', (int(seg[-1]['End_Timestamp']) - t0) / 1e6)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{v[0]/1e6:8.3f} ms  x{v[1]:3d}  first@{(v[2]-t0)/1e6:7.2f}  {k[:70]}")
