import csv, glob, collections, sys
tag, pat = sys.argv[1], sys.argv[2]
for i in (1, 2, 3, 4):
    fs = glob.glob('/root/repo/gpurun_out/pmc_%s/pass%d/*/*counter_collection.csv' % (tag, i))
    if not fs: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if pat in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()): print(i, k, "%.4g" % (sum(v) / len(v)), len(v))
