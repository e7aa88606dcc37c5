#!/bin/bash
# A/B of one env var under extra bench flags: scratch/ab_cfg.sh "<flags>" VAR v1 v2
FLAGS=$1; VAR=$2; shift; shift
for round in 1 2; do
for v in "$@"; do
  r=$(env $VAR=$v python bench.py $FLAGS --no-sub-records --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "[$FLAGS] $VAR=$v round $round: $r"
done
done
