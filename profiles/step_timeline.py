"""Timeline of ONE replayed training step from a rocprofv3 --kernel-trace CSV: the kernels in start order, with the idle gap in
front of each, summed per phase of the step (the phases are cut at the two adam_multi launches: discriminator step | generator
step) and per kernel symbol.  Usage: python profiles/step_timeline.py <kernel_trace.csv> [--list]"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
    adam = [i for i, e in enumerate(ev) if "adam_multi" in e[2]]
    if len(adam) < 6:
        sys.exit("need at least three steps in the trace")
    # the last complete step: from just after the G-adam of step n-2 to the G-adam of step n-1 (adam launches alternate D, G)
    a0, a1, a2 = adam[-3], adam[-2], adam[-1]
    step = ev[a0 + 1:a2 + 1]
    t0 = ev[a0][1]
    total = (step[-1][1] - t0) / 1e3
    busy = sum(e[1] - e[0] for e in step) / 1e3
    print("step: %d launches, %.1f us wall, %.1f us kernel time, %.1f us idle between kernels" % (len(step), total, busy, total - busy))
    prev = t0
    per = collections.defaultdict(lambda: [0, 0.0, 0.0])
    phase_of = {}
    for i, e in enumerate(step):
        ph = "D step" if a0 + 1 + i <= a1 else "G step"
        k = per[(ph, short(e[2]))]
        k[0] += 1
        k[1] += (e[1] - e[0]) / 1e3
        k[2] += max(0, e[0] - prev) / 1e3
        if "--list" in sys.argv:
            print("%9.1f %7.1f gap %5.1f  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, (e[0] - prev) / 1e3, short(e[2])))
        prev = max(prev, e[1])
    for ph in ("D step", "G step"):
        items = [(k[1], v) for k, v in per.items() if k[0] == ph]
        print("%s: %d launches, %.1f us kernel, %.1f us gaps" % (ph, sum(v[0] for _, v in items), sum(v[1] for _, v in items), sum(v[2] for _, v in items)))
    print("%-8s %-70s %5s %9s %8s" % ("phase", "kernel", "n", "us", "gap us"))
    for (ph, name), v in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("%-8s %-70s %5d %9.1f %8.1f" % (ph, name, v[0], v[1], v[2]))


if __name__ == "__main__":
    main()
