import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total %.2f ms/step" % (tot / steps / 1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print("%-70s x%6.1f  avg %7.1f us  %6.3f ms/step" % (n[:70], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / steps / 1e6))
