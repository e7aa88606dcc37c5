#!/bin/bash
# plain step time + kernel-trace breakdown of TWO trees on one box: scratch/floor2.sh <treeA> <treeB>
cd /root/repo
for T0 in "$@"; do
  T=$(realpath /root/repo/$T0)
  TAG=$(basename $T)
  OUT=/root/repo/gpurun_out/floor_$TAG
  rm -rf $OUT; mkdir -p $OUT
  cd $T
  python bench.py --steps 60 --warmup 10 --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/plain.json 2> $OUT/plain.err
  PLAIN=$(tail -1 $OUT/plain.json | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  BENCH=$T/bench.py
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $BENCH --steps 8 --warmup 3 --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/bench.json 2> $OUT/err.txt
  cd /root/repo
  TR=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
  python3 profiles/step_timeline.py $TR --floor $PLAIN > gpurun_out/floor_$TAG.json
  python3 profiles/step_timeline.py $TR --list > gpurun_out/timeline_$TAG.txt
  rm -rf $OUT/trace
  echo "== $TAG"; cat gpurun_out/floor_$TAG.json
done
