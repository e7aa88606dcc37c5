#!/bin/bash
# Round-5 profile collection on the GPU box (one call): plain step, kernel-trace stats of the eager and of the replayed step (+ the
# replayed step's timeline and non-convolution floor), FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md
# prescribes), SQ counters of the dominant kernel on its typical launch.  Output: gpurun_out/prof_r5/
set -x
OUT=/root/repo/gpurun_out/prof_r5
rm -rf $OUT; mkdir -p $OUT
cd /root/repo
python bench.py --steps 60 --warmup 10 --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/plain.json 2> $OUT/plain.err
PLAIN=$(tail -1 $OUT/plain.json | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/bench.py --steps 8 --warmup 2 --no-graphs --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/trace_bench.json 2> $OUT/trace.err
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-graphs --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/fetch_bench.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-graphs --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/write_bench.json 2> $OUT/write.err
cd /root/repo
python3 profiles/extract_traffic.py $(ls $OUT/fetch/*/*counter_collection.csv | head -1) $(ls $OUT/write/*/*counter_collection.csv | head -1) $OUT/traffic.json
rm -rf $OUT/fetch $OUT/write
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_graph -- python3 /root/repo/bench.py --steps 10 --warmup 3 --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/trace_graph_bench.json 2> $OUT/trace_graph.err
cd /root/repo
cp $(ls $OUT/trace_graph/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_graph.csv
TR=$(ls $OUT/trace_graph/*/*kernel_trace.csv | head -1)
python3 profiles/step_timeline.py $TR --floor $PLAIN > $OUT/nonconv_floor.json
python3 profiles/step_timeline.py $TR --list > $OUT/step_timeline.txt
rm -rf $OUT/trace_graph
# SQ counters of the dominant kernel (conv3x3_pp<bf16,2>, 128 -> 128 @128^2, B = 20), three passes
bash scratch/pmc_pp.sh r5_pp 1 128 128 128 20 > $OUT/pmc_pp.log 2>&1
cp gpurun_out/pmc_r5_pp/summary.txt $OUT/pmc_conv3x3_pp_2_128x128_128px.txt
ls -la $OUT

# BASELINE.json config 4 on one GPU: kernel stats of the replayed step at channel_factor 0.5 and 2 (round-4 VERDICT, next #7)
cd /tmp
for CF in 0.5 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cf$CF -- python3 /root/repo/bench.py --channel-factor $CF --steps 8 --warmup 3 --no-sub-records --no-cpu-baseline --no-kernel-probe > $OUT/trace_cf${CF}_bench.json 2> $OUT/trace_cf$CF.err
  cp $(ls $OUT/trace_cf$CF/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_cf$CF.csv
  rm -rf $OUT/trace_cf$CF
done
cd /root/repo
ls -la $OUT
