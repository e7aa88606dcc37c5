#!/bin/bash
# Round-4 profile collection on the GPU box: kernel-trace stats of bench.py, FETCH_SIZE / WRITE_SIZE passes (separate runs, as
# MI355X_MICROARCH.md prescribes), SQ counters of the dominant kernel on its typical launch.  Output: gpurun_out/prof_r4/
set -x
OUT=/root/repo/gpurun_out/prof_r4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/bench.py --steps 8 --warmup 2 --no-graphs --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-graphs --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/fetch_bench.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-graphs --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/write_bench.json 2> $OUT/write.err
cd /root/repo
python3 profiles/extract_traffic.py $(ls $OUT/fetch/*/*counter_collection.csv | head -1) $(ls $OUT/write/*/*counter_collection.csv | head -1) $OUT/traffic.json
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace/*/*kernel_trace.csv $OUT/fetch $OUT/write      # keep the pull small
ls -la $OUT
# graph-replay run (the headline's launch mode): kernel-trace stats only
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_graph -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/trace_graph_bench.json 2> $OUT/trace_graph.err
cd /root/repo
cp $(ls $OUT/trace_graph/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_graph.csv
rm -rf $OUT/trace_graph
# (the probe's non-convolution floor: scratch/floor.sh - plain step + kernel trace of the replayed steps on one box)
ls -la $OUT
