#!/bin/bash
# The probe's non-convolution floor, measured on ONE box: plain step time, then the kernel trace of the same command
OUT=/root/repo/gpurun_out/floor
rm -rf $OUT; mkdir -p $OUT
cd /root/repo
python bench.py --steps 60 --warmup 10 --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/plain.json 2> $OUT/plain.err
PLAIN=$(tail -1 $OUT/plain.json | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/bench.py --steps 8 --warmup 3 --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/bench.json 2> $OUT/err.txt
cd /root/repo
TR=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python3 profiles/step_timeline.py $TR --floor $PLAIN > gpurun_out/floor.json
python3 profiles/step_timeline.py $TR --list > gpurun_out/timeline.txt
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) gpurun_out/floor_kernel_stats.csv
rm -rf $OUT/trace
cat gpurun_out/floor.json
