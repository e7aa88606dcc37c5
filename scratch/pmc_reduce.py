"""mean of every counter over the LAST 20 dispatches of kernels matching a pattern (3 passes) + mean duration from the kernel trace"""
import collections, csv, glob, sys
root, pat = sys.argv[1:3]
for i in (1, 2, 3):
    fs = glob.glob('%s/pass%d/*/*counter_collection.csv' % (root, i))
    if not fs: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if pat in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    ks = glob.glob('%s/pass%d/*/*kernel_trace.csv' % (root, i))
    dur = []
    if ks:
        for r in csv.DictReader(open(ks[0])):
            if pat in r['Kernel_Name']:
                dur.append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
    print("pass %d: %d dispatches, duration last-20 mean %.2f us" % (i, len(dur), sum(dur[-20:]) / max(1, len(dur[-20:]))))
    for k, v in sorted(agg.items()):
        print("  %-28s %.6e" % (k, sum(v[-20:]) / len(v[-20:])))
