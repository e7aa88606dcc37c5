#!/bin/bash
# A/B of one environment variable on the same box: scratch/ab_env.sh VAR v1 v2 ...   (two rounds, interleaved)
VAR=$1; shift
for round in 1 2; do
for v in "$@"; do
  r=$(env $VAR=$v python bench.py --steps 60 --warmup 10 --no-sub-records --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$VAR=$v round $round: $r"
done
done
