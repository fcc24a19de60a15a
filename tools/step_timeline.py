#!/usr/bin/env python3
"""One training step out of a rocprofv3 --kernel-trace CSV: start, duration, gap to the previous kernel, stream, name.
    python tools/step_timeline.py <kernel_trace.csv> [step index from the end, default 3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
idx = [i for i, r in enumerate(rows) if 'noise_kernel' in r['Kernel_Name']]
s, e = idx[-back], idx[-back + 1]
t0 = int(rows[s]['Start_Timestamp']); prev = None; ksum = 0
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = 0 if prev is None else st - prev
    ksum += en - st
    print(f"{(st - t0) / 1e3:8.1f} {(en - st) / 1e3:7.1f} gap {gap / 1e3:6.1f} q{r['Queue_Id']:>2} s{r['Stream_Id']:>2} {r['Kernel_Name'][:100]}")
    prev = max(prev or 0, en)
print('step span', (int(rows[e]['Start_Timestamp']) - t0) / 1e3, 'kernel sum', ksum / 1e3, 'launches', e - s)
