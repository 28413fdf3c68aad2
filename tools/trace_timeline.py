"""Per-pass kernel timeline (start, gap, duration) from a rocprofv3 --kernel-trace CSV."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
starts = [i for i, n in enumerate(names) if 'k_uf_tiles' in n]
a, b = starts[-2], starts[-1]
seg = rows[a:b]
t0 = int(seg[0]['Start_Timestamp'])
tot = 0; prev_end = t0; gaps = 0
for r in seg:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    nm = re.sub(r'\(.*', '', r['Kernel_Name']).replace('povu_hip::', '').replace('void ', '')
    if 'rocprim' in nm:
        m = re.search(r'(radix_sort_\w+|scan_impl|init_lookback\w*|histogram\w*)', r['Kernel_Name']); nm = 'rocprim:' + (m.group(1) if m else '?')
    if len(sys.argv) < 3:
        print(f"{(s-t0)/1e3:9.1f} gap {(s-prev_end)/1e3:6.1f} dur {(e-s)/1e3:7.1f} {nm[:60]}")
    tot += e - s; gaps += max(0, s - prev_end); prev_end = e
print('launches', len(seg), 'kernel sum us', tot / 1e3, 'gaps us', gaps / 1e3, 'span us', (prev_end - t0) / 1e3)
