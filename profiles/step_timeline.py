"""Timeline of ONE replayed training step from a rocprofv3 --kernel-trace CSV: the kernels in start order, with the idle gap in
front of each, summed per phase of the step (the phases are cut at the two adam_multi launches: discriminator step | generator
step) and per kernel symbol.  Usage: python profiles/step_timeline.py <kernel_trace.csv> [--list]

    python profiles/step_timeline.py <kernel_trace.csv> --floor PLAIN_STEP_MS > profiles/roundN_nonconv_floor.json

writes the record bench.py's probe check reads: the kernel time of everything that is not a probed convolution launch, per step,
over the last three replayed steps of the trace.  PLAIN_STEP_MS = ms_per_step of the same command on the same box WITHOUT the
profiler: under --kernel-trace every launch is ~3 us longer than in the replayed graph (a launch that computes nothing shows as
4.7 us; 100 of them added to the replayed step cost 1.57 us each - profiles/README.md), so the profiled total exceeds the plain step
although the GPU is never idle in either.  The excess, spread evenly over the launches, is taken off the non-convolution total."""
import collections
import csv
import json
import re
import sys

# the kernels behind sp_conv2d_igemm / sp_conv2d_wgrad* (the launches bench.py brackets with events), incl. their finalize / reduce passes
CONV = re.compile(r"conv3x3_|conv1x1_|conv_igemm|conv_finalize|conv_wgrad|wgrad1x1_|wgrad3x3_|wgrad_reduce|conv_wgrad_rows_reduce")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
    adam = [i for i, e in enumerate(ev) if "adam_multi" in e[2]]
    if len(adam) < 8:
        sys.exit("need at least four steps in the trace")
    if "--floor" in sys.argv:
        plain_ms = float(sys.argv[sys.argv.index("--floor") + 1])
        nsteps = 3
        span = ev[adam[-1 - 2 * nsteps] + 1:adam[-1] + 1]           # the last three steps (two adam launches each)
        conv = [e for e in span if CONV.search(e[2])]
        other = [e for e in span if not CONV.search(e[2])]
        conv_ms = sum(e[1] - e[0] for e in conv) / 1e6 / nsteps
        other_ms = sum(e[1] - e[0] for e in other) / 1e6 / nsteps
        launches = len(span) / nsteps
        inflation_us = max(0.0, (conv_ms + other_ms - plain_ms) * 1e3 / launches)
        print(json.dumps({"source": "rocprofv3 --kernel-trace of `bench.py` (graph replay), last %d steps" % nsteps, "steps": nsteps,
                          "plain_step_ms": plain_ms, "profiled_kernel_ms_per_step": round(conv_ms + other_ms, 4),
                          "profiler_inflation_us_per_launch": round(inflation_us, 3),
                          "launches_per_step": round(launches, 1), "nonconv_launches_per_step": round(len(other) / nsteps, 1),
                          "conv_ms_per_step_profiled": round(conv_ms, 4), "nonconv_ms_per_step_profiled": round(other_ms, 4),
                          "nonconv_ms_per_step": round(other_ms - len(other) / nsteps * inflation_us / 1e3, 4)}, indent=1))
        return
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
