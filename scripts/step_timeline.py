"""Developer aid: per-kernel timeline (start, duration, gap) of the second-to-last training step in a
rocprofv3 --kernel-trace CSV.   usage: python scripts/step_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_kernel')]
a, b = idx[-3], idx[-2]
seg = rows[a + 1:b + 1]
t0 = int(seg[0]['Start_Timestamp'])
prev_end = int(rows[a]['End_Timestamp'])
tot_gap = 0
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3
    tot_gap += max(gap, 0)
    print(f"{(s - t0) / 1e3:9.1f} us  dur={(e - s) / 1e3:7.1f} gap={gap:6.1f}  {r['Kernel_Name'][:50]}")
    prev_end = max(prev_end, e)
print("total gap us", round(tot_gap, 1), "step us", (int(seg[-1]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3)
