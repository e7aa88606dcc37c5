"""Host cost of the graph-replayed step, phase by phase (no synchronisation inside), and what the GPU needs for the same steps."""
import sys, time
sys.path.insert(0, '.')
import torch
import semantic_pyramid_for_image_generation_amd as sp
from semantic_pyramid_for_image_generation_amd import ops, params, synthetic
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
G = sp.Generator().cuda(); D = sp.Discriminator().cuda(); V = sp.VGG16(); V.load_state_dict(params.synth_state_dict(V.state_dict(), 2)); V.cuda().eval()
og = torch.optim.Adam(G.parameters(), lr=1e-5); od = torch.optim.Adam(D.parameters(), lr=1e-5)
mw = sp.ModelWrapper(G, D, None, None, vgg16=V, generator_optimizer=og, discriminator_optimizer=od, save_data_path=None)
im, lb, mk = synthetic.synthetic_batch(20, 1234); im, lb, mk = im.cuda(), lb.cuda(), [m.cuda() for m in mk]
for _ in range(3): mw.train_step(im, lb, mk)
mw.capture_graphs(im, lb, mk)
for _ in range(5): mw.train_step_graphed()
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N): mw.train_step_graphed()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('graphed: host enqueue %.2f ms/step, wall %.2f ms/step' % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
# phase by phase
st = mw._graph_state
acc = {}
def tick(name, f):
    a = time.perf_counter(); f(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - a
for _ in range(N):
    tick('noise_d', lambda: st["noise_d"].normal_())
    tick('replay gd', lambda: st["gd"].replay())
    tick('noise_g', lambda: st["noise_g"].normal_())
    tick('replay gf', lambda: st["gf"].replay())
    tick('adam d', lambda: mw.discriminator_optimizer.step())
    tick('replay gg', lambda: st["gg"].replay())
    tick('adam g', lambda: mw.generator_optimizer.step())
torch.cuda.synchronize()
for k, v in acc.items(): print('  %-10s %.3f ms/step (host)' % (k, v / N * 1e3))
# the same kernels with the device kept saturated: queue 10 steps, time on the device
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(N): mw.train_step_graphed()
e1.record(); torch.cuda.synchronize()
print('device time between events: %.2f ms/step' % (e0.elapsed_time(e1) / N))
