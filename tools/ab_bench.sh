#!/bin/bash
# A/B of two environments on ONE box, interleaved rounds:  tools/ab_bench.sh "ENV_A=.." "ENV_B=.." [rounds]   ("-" = no variable)
# prints ms/step of every run (bench.py --no-probe --cpu-steps 0 --steps 30 --warmup 10)
A="$1"; B="$2"; R=${3:-3}
for i in $(seq $R); do
  for v in "$A" "$B"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    ms=$(env $e python3 bench.py --steps 30 --warmup 10 --cpu-steps 0 --no-probe 2>/dev/null | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
    echo "round $i [$v] $ms"
  done
done
