#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV (steps delimited by the AdamW launch)."""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
a, b = idx[which] + 1, idx[which + 1] + 1
step = rows[a:b + 1]
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print(f"kernels {len(step)}  wall {(t1-t0)/1e3:.1f} us  busy(union) {busy/1e3:.1f} us  sum {sum(e-s for s,e in iv)/1e3:.1f} us")
verbose = len(sys.argv) > 2
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'_ZN2mm\d+', '', r['Kernel_Name'])
    name = re.sub(r'IDF16b|NS_|EEvT0_.*|EEEE.*', ' ', name)[:64]
    if verbose or (e - s) > 30000:
        print(f"{(s-t0)/1e3:8.1f} {(e-s)/1e3:7.1f} q{r['Queue_Id']} {name}")
