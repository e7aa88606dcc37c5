#!/bin/bash
# same-box A/B of this tree against the baseline worktree scratch/basetree (built beforehand): two interleaved rounds
run() { (cd $1 && python bench.py --steps 60 --warmup 10 --no-sub-records --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); }
for round in 1 2 3; do
  echo "base  round $round: $(run scratch/basetree)"
  echo "tree  round $round: $(run .)"
done
