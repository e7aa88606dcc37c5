#!/bin/bash
# kernel-trace stats of a short graphed bench run -> gpurun_out/kstats.csv
OUT=/root/repo/gpurun_out/kst
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/bench.json 2> $OUT/err.txt
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) /root/repo/gpurun_out/kstats.csv
rm -rf $OUT/trace
tail -1 $OUT/bench.json | cut -c1-200
