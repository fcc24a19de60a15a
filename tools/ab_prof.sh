#!/bin/bash
# kernel-time A/B under rocprofv3 on ONE box:  tools/ab_prof.sh "ENV_A" "ENV_B" 'regex of kernel names'   ("-" = no variable)
# prints name, calls, average ns of the matching rows of each run's kernel_stats.csv
A="$1"; B="$2"; RE="${3:-.}"
i=0
for v in "$A" "$B"; do
  i=$((i+1)); d=gpurun_out/abprof_$i; rm -rf $d
  if [ "$v" = "-" ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 30 --warmup 10 --cpu-steps 0 --no-probe > $d.json 2> $d.err
  else
    export "$v"
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 30 --warmup 10 --cpu-steps 0 --no-probe > $d.json 2> $d.err
    unset "${v%%=*}"
  fi
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "[$v] $(cat $d.json | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  if [ -n "$f" ]; then python3 - "$f" "$RE" <<'P'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]): print("  %-70s %5s %10.0f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])))
P
  fi
done
