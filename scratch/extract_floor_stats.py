#!/usr/bin/env python3
"""Kernel time per training step of everything that is NOT a convolution launch bench.py's probe brackets with events, from a
rocprofv3 --kernel-trace --stats summary of `bench.py --no-graphs ...` (the *_kernel_stats.csv):

    python profiles/extract_floor.py profiles/round4_kernel_stats.csv STEPS > profiles/round4_nonconv_floor.json

bench.py reads the result for its sanity check (conv time + this floor must fit into the measured step)."""
import csv
import json
import re
import sys

# the kernels behind sp_conv2d_igemm / sp_conv2d_wgrad* (probed launches), incl. their split-K finalize / slab-reduce passes
CONV = re.compile(r"conv3x3_|conv1x1_|conv_igemm|conv_finalize|conv_wgrad|wgrad1x1_|wgrad3x3_|wgrad_reduce|conv_wgrad_rows_reduce")


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    conv_ns = other_ns = 0
    conv_calls = other_calls = 0
    for row in csv.DictReader(open(path)):
        ns, calls = int(row["TotalDurationNs"]), int(row["Calls"])
        if CONV.search(row["Name"]):
            conv_ns += ns; conv_calls += calls
        else:
            other_ns += ns; other_calls += calls
    print(json.dumps({"source": path, "steps": steps, "nonconv_ms_per_step": round(other_ns / steps / 1e6, 4),
                      "conv_ms_per_step": round(conv_ns / steps / 1e6, 4), "launches_per_step": round((conv_calls + other_calls) / steps, 1),
                      "nonconv_launches_per_step": round(other_calls / steps, 1)}, indent=1))


if __name__ == "__main__":
    main()
