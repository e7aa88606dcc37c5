import sys, time, os
sys.path.insert(0, '/root/repo')
import torch
from oracle import sempyr_oracle as O
from semantic_pyramid_for_image_generation_amd import params, synthetic
for th in (16, 32, 64):
    torch.set_num_threads(th)
    G = O.make_state(params.synth_state_dict(O.layout_template(O.generator_layout(1)), 0))
    D = O.make_state(params.synth_state_dict(O.layout_template(O.discriminator_layout(1)), 1))
    V = O.make_state(params.synth_state_dict(O.layout_template(O.vgg16_layout()), 2), frozen=True)
    og = torch.optim.Adam(O.trainable(G), lr=1e-5); od = torch.optim.Adam(O.trainable(D), lr=1e-5)
    images, labels, masks = synthetic.synthetic_batch(2, 0)
    g = torch.Generator().manual_seed(2)
    ts = []
    for i in range(3):
        nd, ng = torch.randn(2, 128, generator=g), torch.randn(2, 128, generator=g)
        t0 = time.time(); O.train_step(G, D, V, og, od, images, labels, masks, nd, ng, skip_dead_d_wgrad=True); ts.append(time.time() - t0)
        if time.time() - t0 > 60: break
    print(th, "threads: steps", [round(t, 2) for t in ts], "-> %.2f img/s" % (2 / min(ts[1:] or ts)), flush=True)
