#!/bin/bash
# HBM bytes of the split-K linear kernel (FETCH_SIZE / WRITE_SIZE in their own passes): gpurun_out/pmc_linear/pass{1,2}
mkdir -p /root/repo/gpurun_out/pmc_linear
cd /tmp && export TMPDIR=/tmp
i=1
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /root/repo/gpurun_out/pmc_linear/pass$i -- python3 /root/repo/scratch/bench_linear.py > /root/repo/gpurun_out/pmc_linear/log$i.txt 2>&1
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
for i in (1, 2):
    fs = glob.glob('/root/repo/gpurun_out/pmc_linear/pass%d/*/*counter_collection.csv' % i)
    if not fs: print("no csv", i); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'linear_mfma' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:60], r['Counter_Name'], r.get('Grid_Size', ''))].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()): print(i, k, "avg %.4g" % (sum(v) / len(v)), "max %.4g" % max(v), len(v))
PY
