#!/usr/bin/env python3
"""Microbenchmark of sp_conv2d_igemm on a list of 3x3 shapes under tuning switches (same box, same process):

    python3 tools/time_conv.py "28=0;28=1" [n,cin,cout,h,w[,ksize] ...]

first argument: ';'-separated tuning settings, each a ','-separated list of key=value (sp_set_tuning keys, include/sempyr.h); every
shape is timed under every setting (median of 40 launches behind 5 warm-up launches, events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_pyramid_for_image_generation_amd import ops

DEFAULT = [(20, 512, 512, 32, 32), (40, 512, 512, 32, 32), (20, 256, 256, 64, 64), (40, 256, 128, 64, 64), (40, 256, 256, 32, 32),
           (20, 128, 128, 128, 128), (20, 256, 256, 32, 32), (20, 128, 64, 128, 128), (20, 512, 512, 16, 16), (40, 512, 512, 16, 16)]


def main():
    settings = [[tuple(int(v) for v in kv.split("=")) for kv in s.split(",") if kv] for s in (sys.argv[1] if len(sys.argv) > 1 else "").split(";")]
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or DEFAULT
    dt = torch.bfloat16
    for shp in shapes:
        n, cin, cout, h, w = shp[:5]
        ks = shp[5] if len(shp) > 5 else 3
        x = ops.nhwc_empty(n, cin, h, w, dt, "cuda").normal_()
        wt = (torch.randn(cout * ks * ks * cin, device="cuda") * 0.05).to(dt)
        bias = torch.randn(cout, device="cuda")
        y = ops.nhwc_empty(n, cout, h, w, dt, "cuda")
        out = []
        for st in settings:
            for k, v in st:
                ops.set_tuning(k, v)
            try:
                def launch():
                    ops._conv_launch(x, wt.data_ptr(), bias, y, None, None, None, 0.2, n, h, w, cin, cout, cout, ks, 1, dt)
                for _ in range(5):
                    launch()
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
                torch.cuda.synchronize()
                for a, b in ev:
                    a.record(); launch(); b.record()
                torch.cuda.synchronize()
                t = sorted(a.elapsed_time(b) for a, b in ev)[20] * 1e3
                from semantic_pyramid_for_image_generation_amd import _lib as L
                out.append((t, L.lib().sp_last_route().decode()))
            finally:
                for k, _ in st:
                    ops.set_tuning(k, -1)
        gf = 2.0 * n * h * w * cin * cout * ks * ks / 1e9
        mb = n * h * w * (cin + cout) * 2 / 1e6
        print("n=%d %d->%d @%dx%d k%d  " % (n, cin, cout, h, w, ks) + "  |  ".join("%7.1f us %5.0f TF %4.1f TB/s %s" % (t, gf / t * 1e-3, mb / t * 1e-6 * 1e0, r[:28]) for t, r in out))


if __name__ == "__main__":
    main()
