"""CPU oracle for the Semantic-Pyramid GAN adversarial train step.

TEST INFRASTRUCTURE ONLY.  This file is a plain torch-CPU fp32 *restatement* of the
reference's algorithm for the hot path named in BASELINE.json (one discriminator step +
one generator step, /root/reference/model_wrapper.py:131-190).  It exists so the HIP
path can be checked for result parity; it is never the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the unmodified
reference (with stubs for the three absent third-party packages kornia / torchvision /
skimage) in the build container, runs seeded steps and commits the resulting vectors
under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against those vectors.  The reference itself ships no tests or golden vectors
(SURVEY.md section 4), so the committed vectors embed torch-2.10 CPU semantics.

Design: the reference is a tree of ``nn.Module`` objects.  This restatement is
deliberately *functional*: a network is a flat ``dict`` name -> tensor that uses the
reference's ``state_dict`` key names, and every layer is a function of that dict and a
key prefix.  That keeps the oracle independent code (nothing is copied) while letting a
reference ``state_dict`` be dropped in unchanged.

Third-party arithmetic restated here (absent from /root/reference, unpinned in its
requirements.txt; semantics are those of the torch 2.10 wheel in this image):
  * torch.nn.utils.spectral_norm (legacy hook API): one power iteration per
    training-mode forward, in place, sigma = u . (W v), W / sigma.
  * torchvision VGG-16 (configuration "D") topology; kornia.normalize affine.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]

LRELU_SLOPE = 0.2
# VGG-16 "D": conv widths, 'M' = 2x2/2 max-pool (torchvision.models.vgg16 topology).
VGG_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


# --------------------------------------------------------------------------------------
# storage-noise model (tests only; default OFF = the exact fp32 restatement above all else)
#
# The HIP path's throughput modes keep activations, activation gradients and the packed weights W / sigma in a 16-bit storage
# type and accumulate in fp32.  ``set_storage(torch.bfloat16)`` makes THIS oracle round the same quantities to that type - every
# layer output and its gradient, every normalised weight - while all arithmetic stays fp32: a noise model of 16-bit storage that is
# independent of any kernel.  tests/test_gpu_step.py derives the bounds of the bf16 / fp16 modes from it (the error of this
# model against the fp32 goldens, times a stated factor) instead of from a past measurement of the kernels themselves.
# --------------------------------------------------------------------------------------
_STORAGE: List[Optional[torch.dtype]] = [None]
_GRAD_SCALE = [1.0]


def set_storage(dtype: Optional[torch.dtype], grad_scale: float = 1.0) -> None:
    """None: exact fp32 (the oracle proper).  torch.bfloat16 / torch.float16: the storage-noise model; grad_scale = the loss scale the
    gradients carry while they are stored (the fp16 mode's 2^16: without it they underflow)."""
    _STORAGE[0] = dtype
    _GRAD_SCALE[0] = float(grad_scale)


class _StoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype, gscale):
        ctx.dtype, ctx.gscale = dtype, gscale
        return x.to(dtype).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return (g * ctx.gscale).to(ctx.dtype).to(torch.float32) / ctx.gscale, None, None


def st_(x: torch.Tensor) -> torch.Tensor:
    """x as the 16-bit storage of the noise model holds it (identity in the oracle proper)."""
    if _STORAGE[0] is None or not x.is_floating_point():
        return x
    return _StoreFn.apply(x, _STORAGE[0], _GRAD_SCALE[0])


# --------------------------------------------------------------------------------------
# spectral normalisation  (torch.nn.utils.spectral_norm, call sites models.py:28,34,55,...)
# --------------------------------------------------------------------------------------
def sn_weight(S: State, prefix: str, training: bool) -> torch.Tensor:
    """weight = weight_orig / sigma with one in-place power iteration when training.

    Follows torch/nn/utils/spectral_norm.py (compute_weight): u, v are updated under
    no_grad even inside an outer no_grad forward, then cloned; sigma is differentiated
    through W only.  eps = 1e-12, n_power_iterations = 1, dim = 0.
    """
    w = S[prefix + ".weight_orig"]
    u = S[prefix + ".weight_u"]
    v = S[prefix + ".weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=1e-12))
        u = u.clone()
        v = v.clone()
    sigma = torch.dot(u, torch.mv(wm, v))
    return st_(w / sigma)


def sn_linear(S: State, prefix: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    return st_(F.linear(x, sn_weight(S, prefix, training), S[prefix + ".bias"]))


def sn_conv(S: State, prefix: str, x: torch.Tensor, training: bool, padding: int) -> torch.Tensor:
    return st_(F.conv2d(x, sn_weight(S, prefix, training), S[prefix + ".bias"], stride=1, padding=padding))


def lrelu(x: torch.Tensor) -> torch.Tensor:
    return st_(F.leaky_relu(x, LRELU_SLOPE))


def upsample2(x: torch.Tensor) -> torch.Tensor:
    # nn.UpsamplingBilinear2d(scale_factor=2) == bilinear, align_corners=True (models.py:52,298,308)
    return st_(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True))


# --------------------------------------------------------------------------------------
# class-conditional batch norm (models.py:469-506)
# --------------------------------------------------------------------------------------
def batch_norm(S: State, prefix: str, x: torch.Tensor, training: bool, momentum: float,
               affine: bool) -> torch.Tensor:
    rm, rv = S[prefix + ".running_mean"], S[prefix + ".running_var"]
    if training:
        S[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv,
                        S[prefix + ".weight"] if affine else None,
                        S[prefix + ".bias"] if affine else None,
                        training, momentum, 1e-5)


def conditional_batch_norm(S: State, prefix: str, x: torch.Tensor, class_id: torch.Tensor,
                           training: bool) -> torch.Tensor:
    """BatchNorm2d(affine=False, momentum=0.001) then per-sample (scale, bias) gathered from
    Embedding(classes, 2C)[argmax(class_id)] (models.py:484-505)."""
    y = batch_norm(S, prefix + ".batch_norm", x, training, 0.001, affine=False)
    emb = S[prefix + ".embedding.weight"][class_id.argmax(dim=-1)]
    c = x.shape[1]
    scale, bias = emb[:, :c], emb[:, c:]
    return st_(scale[:, :, None, None] * y + bias[:, :, None, None])


# --------------------------------------------------------------------------------------
# SAGAN self-attention (models.py:219-275)
# --------------------------------------------------------------------------------------
def self_attention(S: State, prefix: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    b, c, h, w = x.shape
    pooled = F.max_pool2d(x, 2, 2)
    q = sn_conv(S, prefix + ".query_convolution", x, training, 0)
    k = sn_conv(S, prefix + ".key_convolution", pooled, training, 0)
    v = sn_conv(S, prefix + ".value_convolution", pooled, training, 0)
    q = q.reshape(b, c // 8, h * w).transpose(1, 2)          # (B, HW, C/8)
    k = k.reshape(b, c // 8, h * w // 4)                      # (B, C/8, HW/4)
    v = v.reshape(b, c // 2, h * w // 4)                      # (B, C/2, HW/4)
    attn = torch.softmax(torch.bmm(q, k), dim=-1)             # no 1/sqrt(d) scale (models.py:266)
    o = st_(torch.bmm(v, attn.transpose(1, 2)).reshape(b, c // 2, h, w))
    o = sn_conv(S, prefix + ".attention_convolution", o, training, 0)
    return st_(S[prefix + ".gamma"] * o + x)


# --------------------------------------------------------------------------------------
# generator (models.py:10-99, blocks :278-375)
# --------------------------------------------------------------------------------------
def linear_block(S: State, prefix: str, x: torch.Tensor, masked_feature: torch.Tensor,
                 training: bool) -> torch.Tensor:
    main = sn_linear(S, prefix + ".main_block.1", lrelu(x), training)
    return st_(main + sn_linear(S, prefix + ".masked_feature_mapping", masked_feature, training))


def generator_residual_block(S: State, prefix: str, x: torch.Tensor, feature: torch.Tensor,
                             class_id: torch.Tensor, training: bool) -> torch.Tensor:
    m = conditional_batch_norm(S, prefix + ".main_block.0", x, class_id, training)
    m = sn_conv(S, prefix + ".main_block.3", upsample2(lrelu(m)), training, 1)
    m = conditional_batch_norm(S, prefix + ".main_block.4", m, class_id, training)
    m = sn_conv(S, prefix + ".main_block.6", lrelu(m), training, 1)
    r = sn_conv(S, prefix + ".residual_mapping.1", upsample2(x), training, 0)
    f = sn_conv(S, prefix + ".masked_feature_mapping", feature, training, 1)
    return st_((m + r) + f)


def generator_forward(S: State, z: torch.Tensor, features: Sequence[torch.Tensor],
                      masks: Sequence[torch.Tensor], class_id: torch.Tensor,
                      training: bool = True) -> torch.Tensor:
    """models.py:65-99.  ``class_id`` is the float one-hot (model_wrapper.py:151)."""
    d = len(features) - 1
    x = sn_linear(S, "linear_layer", z, training)
    x = linear_block(S, "linear_block_1", x, features[d] * masks[d], training); d -= 1
    x = linear_block(S, "linear_block_2", x, features[d] * masks[d], training); d -= 1
    x = x.reshape(x.shape[0], -1, 4, 4)
    x = sn_conv(S, "convolution_layer.1", lrelu(x), training, 0)
    for i in range(6):
        p = "main_path.%d" % i
        if i == 3:
            x = self_attention(S, p, x, training)
        else:
            f = torch.cat((features[d] * masks[d], masks[d]), dim=1)
            x = generator_residual_block(S, p, x, f, class_id, training)
            d -= 1
    x = upsample2(x)
    x = st_(batch_norm(S, "final_block.1", x, training, 0.1, affine=True))
    x = sn_conv(S, "final_block.3", lrelu(x), training, 1)
    x = sn_conv(S, "final_block.5", lrelu(x), training, 0)
    return st_(torch.tanh(x))


# --------------------------------------------------------------------------------------
# discriminator (models.py:102-155, blocks :378-466)
# --------------------------------------------------------------------------------------
def discriminator_input_block(S: State, p: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    """DiscriminatorInputResidualBlock.forward (models.py:408-419)."""
    m = sn_conv(S, p + ".main_block.0", x, training, 1)
    m = sn_conv(S, p + ".main_block.2", lrelu(m), training, 1)
    r = sn_conv(S, p + ".residual_mapping", st_(F.avg_pool2d(x, 2)), training, 0)
    return st_(F.avg_pool2d(m, 2) + r)


def discriminator_residual_block(S: State, p: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    """DiscriminatorResidualBlock.forward (models.py:453-466)."""
    m = sn_conv(S, p + ".main_block.1", lrelu(x), training, 1)
    m = sn_conv(S, p + ".main_block.3", lrelu(m), training, 1)
    r = sn_conv(S, p + ".residual_mapping", x, training, 0)
    return st_(F.avg_pool2d(m + r, 2))


def discriminator_forward(S: State, x: torch.Tensor, class_id: torch.Tensor,
                          training: bool = True) -> torch.Tensor:
    """Returns the (B, B, 128) tensor the reference produces (models.py:151-155 quirk)."""
    x = discriminator_input_block(S, "layers.0", x, training)
    for i in (1, 2, 3, 4, 5, 6, 7):
        p = "layers.%d" % i
        x = self_attention(S, p, x, training) if i == 3 else discriminator_residual_block(S, p, x, training)
    x = lrelu(x)
    x = st_(F.adaptive_avg_pool2d(x, 1).flatten(1))
    x = lrelu(sn_linear(S, "layers.11", x, training))                       # (B, 128)
    emb_w = sn_weight(S, "embedding", training)                             # SN on the embedding too
    emb = emb_w[class_id.argmax(dim=-1, keepdim=True)]                      # (B, 1, 128)
    out_emb = x * emb                                                       # (B,128)*(B,1,128) -> (B,B,128)
    out = sn_linear(S, "classification", x, training)                       # (B, 1)
    return out + out_emb


# --------------------------------------------------------------------------------------
# frozen VGG-16 feature pyramid (models.py:158-216)
# --------------------------------------------------------------------------------------
def vgg16_forward(S: State, x: torch.Tensor) -> List[torch.Tensor]:
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype)[None, :, None, None]
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype)[None, :, None, None]
    x = st_((x - mean) / std)
    feats = []
    idx = 0
    for v in VGG_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            feats.append(x)
            idx += 1
        else:
            p = "vgg16.features.%d" % idx
            x = st_(F.relu(F.conv2d(x, st_(S[p + ".weight"]), S[p + ".bias"], padding=1)))
            idx += 2
    x = st_(F.adaptive_avg_pool2d(x, (7, 7)).flatten(1))
    x = st_(F.relu(F.linear(x, st_(S["vgg16.classifier.0.weight"]), S["vgg16.classifier.0.bias"])))
    # models.py:212 appends the output of classifier[3] (the Linear), but classifier[4] is torchvision's
    # nn.ReLU(inplace=True) and eval-mode Dropout returns its input, so the tapped tensor is overwritten
    # in place: the 4096-d feature the generator and the reconstruction loss see is POST-ReLU.
    # (Pinned by golden step_cf1_b2_seed0 iteration 1, whose sample 0 unmasks this level.)
    x = st_(F.relu(F.linear(x, st_(S["vgg16.classifier.3.weight"]), S["vgg16.classifier.3.bias"])))
    feats.append(x)
    x = st_(F.linear(x, st_(S["vgg16.classifier.6.weight"]), S["vgg16.classifier.6.bias"]))
    feats.append(x)
    return feats


# --------------------------------------------------------------------------------------
# losses (lossfunction.py)
# --------------------------------------------------------------------------------------
def lsgan_generator_loss(pred_fake: torch.Tensor) -> torch.Tensor:
    return 0.5 * torch.mean((pred_fake - 1.0) ** 2)                         # lossfunction.py:137


def lsgan_discriminator_loss(pred_real: torch.Tensor, pred_fake: torch.Tensor):
    return 0.5 * torch.mean((pred_real - 1.0) ** 2), 0.5 * torch.mean(pred_fake ** 2)   # :164


def semantic_reconstruction_loss(feats_real, feats_fake, masks) -> torch.Tensor:
    """lossfunction.py:31-68: max-pool(2) real, fake AND mask, then mean |(r-f)*m| per level."""
    loss = torch.zeros(1, dtype=torch.float32)
    for r, f, m in zip(feats_real, feats_fake, masks):
        if f.dim() == 4:
            r, f, m = F.max_pool2d(r, 2), F.max_pool2d(f, 2), F.max_pool2d(m, 2)
        else:
            r, f, m = (F.max_pool1d(t.unsqueeze(1), 2) for t in (r, f, m))
        loss = loss + torch.mean(torch.abs((r - f) * m))
    return loss


def diversity_loss(images_fake: torch.Tensor, latents: torch.Tensor) -> torch.Tensor:
    """lossfunction.py:92-110."""
    assert images_fake.shape[0] > 1
    h = images_fake.shape[0] // 2
    hz = latents.shape[0] // 2
    num = torch.mean(torch.abs(latents[:hz] - latents[hz:]))
    den = torch.mean(torch.abs(images_fake[:h] - images_fake[h:]))
    return num / (den + 1e-8)


# --------------------------------------------------------------------------------------
# one full D + G training step (model_wrapper.py:131-190)
# --------------------------------------------------------------------------------------
def trainable(S: State) -> List[torch.Tensor]:
    """Parameters in the reference's ``parameters()`` order (insertion order of the dict)."""
    return [t for t in S.values() if t.requires_grad]


def make_state(state_dict: Dict[str, torch.Tensor], frozen: bool = False) -> State:
    """Deep-copies a (reference-keyed) state_dict into an oracle state; float tensors that are
    parameters (not ``*_u``/``*_v``/running stats) become leaves that require grad."""
    S: State = {}
    for k, t in state_dict.items():
        t = t.detach().clone()
        is_buffer = (k.endswith("weight_u") or k.endswith("weight_v") or "running_" in k
                     or k.endswith("num_batches_tracked"))
        if not frozen and not is_buffer and t.is_floating_point():
            t.requires_grad_(True)
        S[k] = t
    return S


def zero_grads(S: State) -> None:
    for t in S.values():
        t.grad = None


def train_step(G: State, D: State, V: State, opt_g, opt_d, images_real: torch.Tensor,
               labels: torch.Tensor, masks: Sequence[torch.Tensor], noise_d: torch.Tensor,
               noise_g: torch.Tensor, w_rec: float = 0.1, w_div: float = 0.1,
               skip_dead_d_wgrad: bool = False) -> Dict[str, torch.Tensor]:
    """One iteration of the reference loop body.  ``noise_d`` / ``noise_g`` replace the two
    ``torch.randn`` draws (model_wrapper.py:147,168) so CPU and GPU runs see identical latents.
    ``labels`` is the int64 one-hot of data.py:58-59."""
    out: Dict[str, torch.Tensor] = {}
    # ---- discriminator step (model_wrapper.py:136-162)
    zero_grads(G); zero_grads(D)
    with torch.no_grad():
        feats_real = vgg16_forward(V, images_real)
        fake = generator_forward(G, noise_d, feats_real, masks, labels.float(), training=True)
    out["images_fake_d"] = fake
    pred_real = discriminator_forward(D, images_real, labels, training=True)
    pred_fake = discriminator_forward(D, fake, labels, training=True)
    l_real, l_fake = lsgan_discriminator_loss(pred_real, pred_fake)
    (l_real + l_fake).backward()
    out["grads_d"] = [p.grad.detach().clone() for p in trainable(D)]
    opt_d.step()
    # ---- generator step (model_wrapper.py:165-190)
    zero_grads(G); zero_grads(D)
    if skip_dead_d_wgrad:
        d_params = trainable(D)
        for p in d_params:
            p.requires_grad_(False)
    fake = generator_forward(G, noise_g, feats_real, masks, labels.float(), training=True)
    pred_fake = discriminator_forward(D, fake, labels, training=True)
    l_g = lsgan_generator_loss(pred_fake)
    l_div = w_div * diversity_loss(fake, noise_g)
    feats_fake = vgg16_forward(V, fake)
    l_rec = w_rec * semantic_reconstruction_loss(feats_real, feats_fake, masks)
    (l_g + l_rec + l_div).backward()
    if skip_dead_d_wgrad:
        for p in d_params:
            p.requires_grad_(True)
    out["grads_g"] = [p.grad.detach().clone() for p in trainable(G)]
    opt_g.step()
    out.update(images_fake_g=fake.detach(), loss_d_real=l_real.detach(), loss_d_fake=l_fake.detach(),
               loss_g=l_g.detach(), loss_rec=l_rec.detach().reshape(()), loss_div=l_div.detach())
    return out


# --------------------------------------------------------------------------------------
# state layouts: ordered name -> (shape, dtype) in the reference's state_dict order
# (checked against the key lists recorded from the reference in tests/golden/*.json)
# --------------------------------------------------------------------------------------
def _sn(spec, prefix, w_shape):
    spec[prefix + ".bias"] = ((w_shape[0],), torch.float32)
    spec[prefix + ".weight_orig"] = (tuple(w_shape), torch.float32)
    rest = 1
    for s in w_shape[1:]:
        rest *= s
    spec[prefix + ".weight_u"] = ((w_shape[0],), torch.float32)
    spec[prefix + ".weight_v"] = ((rest,), torch.float32)


def _bn(spec, prefix, c, affine):
    if affine:
        spec[prefix + ".weight"] = ((c,), torch.float32)
        spec[prefix + ".bias"] = ((c,), torch.float32)
    spec[prefix + ".running_mean"] = ((c,), torch.float32)
    spec[prefix + ".running_var"] = ((c,), torch.float32)
    spec[prefix + ".num_batches_tracked"] = ((), torch.int64)


def _attention(spec, prefix, c):
    spec[prefix + ".gamma"] = ((1,), torch.float32)
    _sn(spec, prefix + ".query_convolution", (c // 8, c, 1, 1))
    _sn(spec, prefix + ".key_convolution", (c // 8, c, 1, 1))
    _sn(spec, prefix + ".value_convolution", (c // 2, c, 1, 1))
    _sn(spec, prefix + ".attention_convolution", (c, c // 2, 1, 1))


def generator_layout(channels_factor=1, latent=128, classes=365, out_channels=3):
    ch = lambda n: int(n // channels_factor)      # models.py:34: the factor DIVIDES
    spec = {}
    _sn(spec, "linear_layer", (latent, latent))
    for name, i, o, f in (("linear_block_1", latent, 365, 365), ("linear_block_2", 365, 2048, 4096)):
        _sn(spec, name + ".main_block.1", (o, i))
        _sn(spec, name + ".masked_feature_mapping", (o, f))
    _sn(spec, "convolution_layer.1", (ch(512), 128, 1, 1))
    blocks = ((ch(512), ch(512), 513), (ch(512), ch(512), 513), (ch(512), ch(256), 257), None,
              (ch(256), ch(128), 129), (ch(128), ch(64), 65))
    for i, b in enumerate(blocks):
        p = "main_path.%d" % i
        if b is None:
            _attention(spec, p, ch(256))
            continue
        ci, co, cf_ = b
        _bn(spec, p + ".main_block.0.batch_norm", ci, False)
        spec[p + ".main_block.0.embedding.weight"] = ((classes, 2 * ci), torch.float32)
        _sn(spec, p + ".main_block.3", (co, ci, 3, 3))
        _bn(spec, p + ".main_block.4.batch_norm", co, False)
        spec[p + ".main_block.4.embedding.weight"] = ((classes, 2 * co), torch.float32)
        _sn(spec, p + ".main_block.6", (co, co, 3, 3))
        _sn(spec, p + ".residual_mapping.1", (co, ci, 1, 1))
        _sn(spec, p + ".masked_feature_mapping", (co, cf_, 3, 3))
    _bn(spec, "final_block.1", ch(64), True)
    _sn(spec, "final_block.3", (ch(64), ch(64), 3, 3))
    _sn(spec, "final_block.5", (out_channels, ch(64), 1, 1))
    return spec


def discriminator_layout(channel_factor=1, classes=365, in_channels=3):
    ch = lambda n: int(n // channel_factor)
    spec = {}
    _sn(spec, "layers.0.main_block.0", (ch(64), in_channels, 3, 3))
    _sn(spec, "layers.0.main_block.2", (ch(64), ch(64), 3, 3))
    _sn(spec, "layers.0.residual_mapping", (ch(64), in_channels, 1, 1))
    widths = {1: (ch(64), ch(128)), 2: (ch(128), ch(256)), 4: (ch(256), ch(256)), 5: (ch(256), ch(256)),
              6: (ch(256), ch(512)), 7: (ch(512), ch(768))}
    for i in range(1, 8):
        p = "layers.%d" % i
        if i == 3:
            _attention(spec, p, ch(256))
            continue
        ci, co = widths[i]
        _sn(spec, p + ".main_block.1", (co, ci, 3, 3))
        _sn(spec, p + ".main_block.3", (co, co, 3, 3))
        _sn(spec, p + ".residual_mapping", (co, ci, 1, 1))
    _sn(spec, "layers.11", (128, ch(768)))
    _sn(spec, "classification", (1, 128))
    spec["embedding.weight_orig"] = ((classes, 128), torch.float32)
    spec["embedding.weight_u"] = ((classes,), torch.float32)
    spec["embedding.weight_v"] = ((128,), torch.float32)
    return spec


def vgg16_layout(classes=365):
    spec = {}
    c, idx = 3, 0
    for v in VGG_CFG:
        if v == "M":
            idx += 1
            continue
        spec["vgg16.features.%d.weight" % idx] = ((v, c, 3, 3), torch.float32)
        spec["vgg16.features.%d.bias" % idx] = ((v,), torch.float32)
        c = v
        idx += 2
    for i, (o, k) in ((0, (4096, 25088)), (3, (4096, 4096)), (6, (classes, 4096))):
        spec["vgg16.classifier.%d.weight" % i] = ((o, k), torch.float32)
        spec["vgg16.classifier.%d.bias" % i] = ((o,), torch.float32)
    return spec


def layout_template(spec) -> Dict[str, torch.Tensor]:
    """Zero tensors with the layout's shapes/dtypes (meta-like template for params.synth_state_dict)."""
    return {k: torch.empty(shape, dtype=dt) for k, (shape, dt) in spec.items()}


# --------------------------------------------------------------------------------------
# training masks (row f1): restatement of csrc/eltwise.hip::training_masks_kernel, i.e. of
# misc.get_masks_for_training (/root/reference/misc.py:13-68) on a counter-based integer generator
# --------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def _mix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def _draw(seed: int, sample: int, k: int) -> int:
    return _mix64((_mix64((seed ^ (sample << 32)) & _M64) + k) & _M64)


def training_mask_decisions(batch: int, seed: int, p_random_mask: float = 0.3):
    """(stage, spatial) of every sample - the decisions of misc.py:28-34."""
    tab = [0, 1, 2, 3, 4, 5, 6, 0, 1]
    thresh = int(float(torch.tensor(p_random_mask, dtype=torch.float32)) * 16777216.0)
    out = []
    for b in range(batch):
        stage = tab[_draw(seed, b, 0) % 9]
        spatial = (_draw(seed, b, 1) >> 40) < thresh and 0 < stage < 6
        out.append((stage, spatial))
    return out


def training_masks(batch: int, seed: int, p_random_mask: float = 0.3) -> List[torch.Tensor]:
    """The seven masks (reference list order: 128^2 ... 365) the device generator must reproduce bit for bit."""
    sides = [1, 1, 8, 16, 32, 64, 128]
    numel = [365, 4096, 64, 256, 1024, 4096, 16384]
    levels = [torch.zeros(batch, n) for n in numel]
    for b, (stage, spatial) in enumerate(training_mask_decisions(batch, seed, p_random_mask)):
        base = sides[min(stage + 1, 6)]
        lo = min(8, base // 2)
        rects = []
        if spatial:
            for k in range(1 + _draw(seed, b, 2) % 4):
                h = lo + _draw(seed, b, 3 + 4 * k) % (base - lo + 1)
                w = lo + _draw(seed, b, 4 + 4 * k) % (base - lo + 1)
                y0 = _draw(seed, b, 5 + 4 * k) % (base - h + 1)
                x0 = _draw(seed, b, 6 + 4 * k) % (base - w + 1)
                rects.append((y0, x0, h, w, _draw(seed, b, 19 + k) % 4))
        for idx in range(7):
            if idx == stage:
                levels[idx][b] = 1.0
            elif idx > stage and spatial:
                side = sides[idx]
                shape_map = torch.ones(base, base)
                for y0, x0, h, w, kind in rects:
                    # kinds of skimage.draw.random_shapes (misc.py:37): 0 rectangle, 1 circle, 2 triangle, 3 ellipse - inside the box,
                    # on pixel centres, integer arithmetic (csrc/eltwise.hip: training_masks_kernel)
                    dy = torch.arange(h, dtype=torch.int64)[:, None]
                    dx = torch.arange(w, dtype=torch.int64)[None, :]
                    ey, ex = 2 * dy + 1 - h, 2 * dx + 1 - w
                    if kind == 0:
                        hit = torch.ones(h, w, dtype=torch.bool)
                    elif kind == 1:
                        hit = ey * ey + ex * ex <= min(h, w) ** 2
                    elif kind == 2:
                        hit = ex.abs() * 2 * h <= w * (2 * dy + 1)
                    else:
                        hit = ey * ey * w * w + ex * ex * h * h <= h * h * w * w
                    box = shape_map[y0:y0 + h, x0:x0 + w]
                    box[hit] = 0.0
                src = (torch.arange(side) * base) // side
                levels[idx][b] = shape_map[src][:, src].reshape(-1)
    out = []
    for idx in (6, 5, 4, 3, 2):
        s = sides[idx]
        out.append(levels[idx].reshape(batch, 1, s, s))
    out += [levels[1], levels[0]]
    return out
