"""Where a wave of conv3x3_tall_kernel spends its cycles (debug build scratch/libsempyr_timing.so: s_memtime stamps per stage):
vmcnt wait, barrier wait, DMA issue, stage (LDS reads + MFMAs), between stages (epilogue)."""
import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
B = 20
dt = torch.bfloat16
for cin, cout, hw in [(128, 128, 128), (256, 256, 64), (512, 512, 32), (64, 64, 256)]:
    x = ops.nhwc_empty(B, cin, hw, hw, dt, 'cuda'); x.normal_()
    w = (torch.randn(cout * 9 * cin, device='cuda') * 0.05).to(dt)
    bias = torch.randn(cout, device='cuda')
    y = ops.nhwc_empty(B, cout, hw, hw, dt, 'cuda')
    dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device='cuda')
    def launch(ws):
        p = L.SpConvParams()
        p.x, p.w, p.bias, p.y = x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr()
        p.n, p.h, p.w_, p.cin_p, p.cout, p.ldy, p.ksize, p.act, p.dtype = B, hw, hw, cin, cout, cout, 3, 1, L.SP_BF16
        if ws is not None:
            p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel() * 8
        import ctypes
        L.call("sp_conv2d_igemm", ctypes.byref(p), ops.stream())
    for _ in range(3): launch(None)
    launch(dbg)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(256, 8, 8).astype(np.float64)
    live = d[:, :, 6] > 0
    tot = d[:, :, 6][live].mean()
    names = ["vmcnt wait", "barrier wait", "DMA issue", "stage (LDS+MFMA)", "between stages", "stages"]
    print("%d->%d @%d: cycles per wave %.0f (%d waves)" % (cin, cout, hw, tot, live.sum()))
    for half, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        dd = d[:, sl, :]
        lv = dd[:, :, 6] > 0
        t = dd[:, :, 6][lv].mean()
        print("   %s: " % half + ", ".join("%s %.1f%%" % (names[k], 100 * dd[:, :, k][lv].mean() / t) for k in range(5)) +
              ", stages %.0f, cycles/stage %.0f" % (dd[:, :, 5][lv].mean(), t / dd[:, :, 5][lv].mean()))
