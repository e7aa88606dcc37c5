#!/bin/bash
# average duration of the kernels matching $1 under the library given by SEMPYR_LIB ($2, empty = in-tree)
PAT=$1; LIBP=$2
OUT=/root/repo/gpurun_out/ksg; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "$LIBP" ]; then export SEMPYR_LIB=$LIBP; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/bench.py --steps 6 --warmup 2 --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/bench.json 2> $OUT/err.txt
grep -E "$PAT" $(ls $OUT/trace/*/*kernel_stats.csv | head -1) | sed "s/.*)\",//" | cut -c1-80
rm -rf $OUT/trace
