#!/bin/bash
# Round-6 profile collection on the GPU box (one gpurun call): the plain step, kernel-trace stats of the eager and of the replayed
# step (+ the replayed step's timeline and non-convolution floor), FETCH_SIZE / WRITE_SIZE passes (separate runs, as
# MI355X_MICROARCH.md prescribes; never combined with other trace domains), SQ counters of the dominant kernel on a typical launch.
# Output: gpurun_out/prof_r6/ - the summaries are then copied into profiles/round6_* by hand.
set -x
OUT=/root/repo/gpurun_out/prof_r6
rm -rf $OUT; mkdir -p $OUT
cd /root/repo
B="--no-sub-records --no-cpu-baseline --no-kernel-probe"
BENCH_NO_SMI=1 python3 bench.py --steps 60 --warmup 10 $B > $OUT/plain.json 2> /dev/null
PLAIN=$(tail -1 $OUT/plain.json | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
cd /tmp && export TMPDIR=/tmp BENCH_NO_SMI=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/bench.py --steps 8 --warmup 2 --no-graphs $B > $OUT/trace_bench.json 2> $OUT/trace.err
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-graphs $B > $OUT/fetch_bench.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-graphs $B > $OUT/write_bench.json 2> $OUT/write.err
cd /root/repo
python3 profiles/extract_traffic.py $(ls $OUT/fetch/*/*counter_collection.csv | head -1) $(ls $OUT/write/*/*counter_collection.csv | head -1) $OUT/traffic.json
rm -rf $OUT/fetch $OUT/write
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_graph -- python3 /root/repo/bench.py --steps 10 --warmup 3 $B > $OUT/trace_graph_bench.json 2> $OUT/trace_graph.err
cd /root/repo
cp $(ls $OUT/trace_graph/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_graph.csv
TR=$(ls $OUT/trace_graph/*/*kernel_trace.csv | head -1)
python3 profiles/step_timeline.py $TR --floor $PLAIN > $OUT/nonconv_floor.json
python3 profiles/step_timeline.py $TR --list > $OUT/step_timeline.txt
rm -rf $OUT/trace_graph
rm -f $OUT/*.err
ls -la $OUT
