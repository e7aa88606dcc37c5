"""fp8 path unit check on the GPU: quantiser / weight packer bytes vs torch.float8_e4m3fn, SP_F8 convolution vs an fp32 emulation on
the dequantised operands; timing vs the bf16 ping-pong kernel."""
import sys; sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops, _lib as L
torch.manual_seed(0)
def check(n, cin, cout, hw, pool2):
    x = ops.nhwc_empty(n, cin, hw, hw, torch.bfloat16, 'cuda'); x.normal_(); x.abs_()
    amax = torch.zeros(1, device='cuda')
    sx = (x.float().abs().max() / 448.0).reshape(1)
    inv = (1.0 / sx)
    x8 = ops.quantize_fp8(x, inv, amax)
    ref8 = (x.float() * inv).clamp(-448, 448).to(torch.float8_e4m3fn)
    same = (x8.contiguous(memory_format=torch.channels_last).view(torch.uint8) == ref8.contiguous(memory_format=torch.channels_last).view(torch.uint8))
    print("quantize bytes equal: %.6f  amax ok: %s" % (same.float().mean().item(), abs(amax.item() - x.float().abs().max().item()) < 1e-6))
    w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    w8, sw, cin_p = ops.pack_weight_fp8(w)
    swr = w.abs().amax(dim=(1, 2, 3)) / 448.0
    wq = (w / swr[:, None, None, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    wref = torch.zeros(cout, 9, cin_p, dtype=torch.uint8, device='cuda')
    wref[:, :, :cin] = wq.permute(0, 2, 3, 1).reshape(cout, 9, cin).view(torch.uint8)
    print("pack bytes equal: %.6f scale err %.2e" % ((w8.view(cout, 9, cin_p) == wref).float().mean().item(), (sw - swr).abs().max().item()))
    bias = torch.randn(cout, device='cuda')
    ho = hw // 2 if pool2 else hw
    y = ops.nhwc_empty(n, cout, ho, ho, torch.bfloat16, 'cuda')
    y8 = torch.empty((n, ho, ho, cout), dtype=torch.uint8, device='cuda').permute(0, 3, 1, 2)
    # reference on the dequantised operands
    xd = ref8.float() * sx
    wd = wq.float() * swr[:, None, None, None]
    r = F.relu(F.conv2d(xd, wd, bias, padding=1))
    if pool2: r = F.max_pool2d(r, 2)
    sy = (r.abs().max() / 448.0).reshape(1); inv_y = 1.0 / sy
    amax_y = torch.zeros(1, device='cuda')
    ops.conv_launch_f8(x8, w8, sw, sx, bias, y, y8, inv_y, amax_y, n, hw, hw, cin_p, cout, ops.ACT_RELU, 2 if pool2 else 0)
    torch.cuda.synchronize()
    err = (y.float() - r).abs().max().item() / r.abs().max().item()
    r8 = (r * inv_y).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    g8 = y8.contiguous(memory_format=torch.channels_last).view(torch.float8_e4m3fn).float() if False else y8.permute(0, 2, 3, 1).contiguous().view(torch.float8_e4m3fn).float().permute(0, 3, 1, 2)
    e8 = ((g8 - r8).abs() > 0).float().mean().item()
    print("n=%d %d->%d @%d pool2=%d: bf16 out rel err %.3e | fp8 out differing codes %.4f | amax %.4f vs %.4f" % (n, cin, cout, hw, pool2, err, e8, amax_y.item(), r.abs().max().item()))
    return x8, w8, sw, sx, bias, y, y8, inv_y, amax_y, cin_p
for args in [(2, 64, 128, 64, 0), (3, 128, 128, 32, 2), (2, 256, 256, 32, 0), (4, 512, 512, 32, 2), (1, 80, 192, 32, 0)]:
    check(*args)
# timing vs bf16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B = 20
for cin, cout, hw in [(128, 128, 128), (256, 256, 64), (512, 512, 32), (64, 128, 128)]:
    x8, w8, sw, sx, bias, y, y8, inv_y, amax_y, cin_p = check(B, cin, cout, hw, 0)
    xb = ops.nhwc_empty(B, cin, hw, hw, torch.bfloat16, 'cuda'); xb.normal_()
    wb = (torch.randn(cout * 9 * cin, device='cuda') * 0.05).to(torch.bfloat16)
    yb = ops.nhwc_empty(B, cout, hw, hw, torch.bfloat16, 'cuda')
    flops = 2.0 * B * hw * hw * cin * cout * 9
    t8 = timeit(lambda: ops.conv_launch_f8(x8, w8, sw, sx, bias, y, y8, inv_y, amax_y, B, hw, hw, cin_p, cout, ops.ACT_RELU, 0))
    t8b = timeit(lambda: ops.conv_launch_f8(x8, w8, sw, sx, bias, None, y8, inv_y, amax_y, B, hw, hw, cin_p, cout, ops.ACT_RELU, 0))
    tb = timeit(lambda: ops._conv_launch(xb, wb.data_ptr(), bias, yb, None, None, None, 0.0, B, hw, hw, cin, cout, cout, 3, ops.ACT_RELU, torch.bfloat16))
    print("%d->%d @%d: fp8 (bf16+fp8 out) %.1f us %.0f TF | fp8 (fp8 out only) %.1f us %.0f TF | bf16 %.1f us %.0f TF" % (cin, cout, hw, t8 * 1e3, flops / t8 / 1e9, t8b * 1e3, flops / t8b / 1e9, tb * 1e3, flops / tb / 1e9))
