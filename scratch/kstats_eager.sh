#!/bin/bash
# kernel-trace stats of an EAGER bench run (12 steps: 10 timed + 2 warm-up) -> gpurun_out/<tag>_kernel_stats.csv
TAG=${1:-r4}
OUT=/root/repo/gpurun_out/kst_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-graphs --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/bench.json 2> $OUT/err.txt
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) /root/repo/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $OUT/trace
tail -1 $OUT/bench.json | cut -c1-160
