#!/bin/bash
# How much of a replayed step is the space BETWEEN kernels: kernel trace of the graph-replayed bench, then per step the sum of
# (start[i+1] - end[i]) over consecutive kernels.  Output: gpurun_out/gaps/summary.txt
OUT=/root/repo/gpurun_out/gaps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 /root/repo/bench.py --steps 6 --warmup 3 --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/bench.json 2> $OUT/err.txt
cd /root/repo
python3 scratch/gaps.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) > $OUT/summary.txt 2>&1
rm -rf $OUT/trace
cat $OUT/summary.txt
