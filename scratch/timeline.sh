#!/bin/bash
# Ordered timeline of one replayed step -> gpurun_out/timeline.txt (+ the torch-native call sites -> gpurun_out/native_kernels.txt)
OUT=/root/repo/gpurun_out/tl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 /root/repo/bench.py --steps 6 --warmup 3 --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/bench.json 2> $OUT/err.txt
cd /root/repo
python3 profiles/step_timeline.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) --list > gpurun_out/timeline.txt 2>&1
rm -rf $OUT/trace
python3 scratch/native_kernels.py > gpurun_out/native_kernels.txt 2> gpurun_out/native_kernels.err
tail -5 gpurun_out/native_kernels.err
head -3 gpurun_out/timeline.txt
