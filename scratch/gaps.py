import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# the last replayed steps: take the final 40 % of the trace (graph replays are back to back there)
n = len(ev)
tail = ev[int(n * 0.6):]
span = tail[-1][1] - tail[0][0]
busy = sum(e - s for s, e, _ in tail)
gaps = [tail[i + 1][0] - max(t[1] for t in tail[max(0, i - 3):i + 1]) for i in range(len(tail) - 1)]
pos = [g for g in gaps if g > 0]
neg = [g for g in gaps if g <= 0]
print("kernels %d  span %.2f ms  sum of kernel durations %.2f ms (%.1f %%)  positive gaps: %d, %.2f ms (%.1f %%), median %.2f us  overlapping starts: %d"
      % (len(tail), span / 1e6, busy / 1e6, 100.0 * busy / span, len(pos), sum(pos) / 1e6, 100.0 * sum(pos) / span,
         sorted(pos)[len(pos) // 2] / 1e3 if pos else 0, len(neg)))
hist = collections.Counter(min(int(g / 500), 20) for g in pos)
print("gap histogram (0.5 us bins):", sorted(hist.items()))
big = sorted(((tail[i + 1][0] - tail[i][1], tail[i][2][:60], tail[i + 1][2][:60]) for i in range(len(tail) - 1)), reverse=True)[:12]
for g, a, b in big: print("%8.1f us  after %-60s before %s" % (g / 1e3, a, b))
