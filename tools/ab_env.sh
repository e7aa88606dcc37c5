#!/bin/bash
# A/B of one environment switch on one box: tools/ab_env.sh VAR "v0 v1 ..." [reps] [extra bench.py args]
# prints images/sec and ms/step of bench.py's headline window per value, interleaved reps (same box, same thermal history)
VAR=$1; VALS=$2; REPS=${3:-3}; shift 3
for r in $(seq 1 $REPS); do
  for v in $VALS; do
    line=$(env $VAR=$v BENCH_NO_SMI=1 python3 bench.py --steps 80 --warmup 15 --no-sub-records --no-cpu-baseline --no-kernel-probe "$@" 2>/dev/null | tail -1)
    echo "$VAR=$v $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
