"""Per-kernel durations and inter-kernel gaps of one frame from a rocprofv3 kernel trace (csv)."""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# pick a frame in the middle: from a k_raygen to the next k_resolve
idx = [i for i, r in enumerate(rows) if 'k_raygen' in r['Kernel_Name']]
start = idx[len(idx) // 2]
end = next(i for i in range(start, len(rows)) if 'k_resolve' in rows[i]['Kernel_Name'])
prev_end = None
tot_k = tot_g = 0
for r in rows[start:end + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.search(r'(k_[a-z_]+)', r['Kernel_Name']).group(1)
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{name:22s} dur {(e - s) / 1e3:8.1f} us   gap before {gap:7.1f} us   grid {r.get('Grid_Size', r.get('Grid_Size_X', ''))}")
    tot_k += (e - s) / 1e3; tot_g += gap
    prev_end = e
print(f"sum kernels {tot_k:.1f} us, sum gaps {tot_g:.1f} us")
