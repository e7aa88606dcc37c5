#!/bin/bash
# same-box A/B of two builds of the library under the same Python tree: scratch/ab_libs.sh <base.so>   (other = the in-tree one)
run() { env $1 python bench.py --steps 60 --warmup 10 --no-sub-records --no-cpu-baseline --no-kernel-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for round in 1 2 3; do
  echo "base    round $round: $(run SEMPYR_LIB=$(realpath $1))"
  echo "in-tree round $round: $(run SP_NOOP=1)"
done
