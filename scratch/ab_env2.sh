#!/bin/bash
# A/B of several env settings on one box: scratch/ab_env2.sh "A=1 B=2" "A=3" ...   ("-" = no setting)
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e="SP_NOOP=1"; else e="$v"; fi
  r=$(env $e python bench.py --steps 60 --warmup 10 --no-sub-records --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "[$v] round $round: $r"
done
done
