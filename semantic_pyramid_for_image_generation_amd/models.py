"""Generator / Discriminator / VGG16 with the reference's constructor signatures, module tree and
``state_dict`` layout (/root/reference/models.py), computing through the libsempyr.so HIP kernels.

Drop-in contract: same class names, same keyword arguments, same parameter names,
shapes (OIHW) and order, so ``torch.optim.Adam(generator.parameters())``, reference checkpoints and the
fine-tuned VGG-16 file load unchanged; ``torch.manual_seed(s)`` followed by construction consumes the RNG
in the reference's order and yields bit-identical initial parameters (tests/test_host_models.py).

Differences in mechanism, not in results: spectral normalisation of a whole network is one batched call
per forward (ops.SpectralNormBank) instead of 32/28 forward-pre-hooks; activations are NHWC in the
compute dtype; LeakyReLU / residual adds / tanh are fused into the producing kernels.
"""
from __future__ import annotations

from typing import List, Optional, Union

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .config import CFG
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH


# --------------------------------------------------------------------------------------------------
# parameter containers for spectral-normalised layers (legacy torch.nn.utils.spectral_norm layout)
# --------------------------------------------------------------------------------------------------
class _SpectralNormMixin:
    def _register_sn(self, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> None:
        # parameter order of the reference: bias first (registered by the wrapped layer), then weight_orig
        if bias is not None:
            self.bias = nn.Parameter(bias)
        self.weight_orig = nn.Parameter(weight)
        rows = weight.shape[0]
        cols = weight[0].numel()
        # same RNG draws, in the same order, as torch.nn.utils.spectral_norm.apply
        u = F.normalize(weight.new_empty(rows).normal_(0, 1), dim=0, eps=1e-12)
        v = F.normalize(weight.new_empty(cols).normal_(0, 1), dim=0, eps=1e-12)
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)


class SNConv2d(nn.Module, _SpectralNormMixin):
    """spectral_norm(nn.Conv2d(..., stride 1, padding k//2, bias=True)) - e.g. models.py:34,299,394."""
    _sn_kind = "conv"

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int) -> None:
        super().__init__()
        ref = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, padding=kernel_size // 2, bias=True)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self._register_sn(ref.weight.data, ref.bias.data)

    def extra_repr(self) -> str:
        return "%d, %d, kernel_size=%d, spectral_norm" % (self.in_channels, self.out_channels, self.kernel_size)

    def forward(self, x, act: int = ACT_NONE, res1=None, res2=None, premasked: bool = False, mask_input: bool = False,
                pool2: bool = False, dest=None):
        return ops.sn_conv2d(x, self, self.kernel_size, act, res1, res2, premasked, mask_input, pool2, dest)

    def pool2_ok(self, x) -> bool:
        """True if the 2x2 average pooling behind this convolution can ride in its epilogue (ops.conv_pool2_ok)."""
        return _FUSE_POOL2 and ops.conv_pool2_ok(x.shape[2], x.shape[3], self.out_channels, self.kernel_size)


class SNLinear(nn.Module, _SpectralNormMixin):
    """spectral_norm(nn.Linear(..., bias=True)) - e.g. models.py:28,128,356."""
    _sn_kind = "linear"

    def __init__(self, in_features: int, out_features: int) -> None:
        super().__init__()
        ref = nn.Linear(in_features, out_features, bias=True)
        self.in_features, self.out_features = in_features, out_features
        self._register_sn(ref.weight.data, ref.bias.data)

    def extra_repr(self) -> str:
        return "%d, %d, spectral_norm" % (self.in_features, self.out_features)

    def forward(self, x, act: int = ACT_NONE, res=None):
        return ops.sn_linear(x, self, act, res)


class SNEmbedding(nn.Module, _SpectralNormMixin):
    """spectral_norm(nn.Embedding(...)) (models.py:135); consumed by the discriminator head kernel."""
    _sn_kind = "plain"

    def __init__(self, num_embeddings: int, embedding_dim: int) -> None:
        super().__init__()
        ref = nn.Embedding(num_embeddings, embedding_dim)
        self._register_sn(ref.weight.data, None)


_FUSE_LRELU_BWD = CFG.fuse_lrelu_bwd   # A/B switch (profiles/README.md)
# The 1x1 residual convolutions commute with the linear resampling next to them (bilinear x2 with align_corners and 2x2
# average pooling are convex combinations of pixels, so the bias passes through unchanged): conv1x1(upsample(x)) =
# upsample(conv1x1(x)) and avgpool(conv1x1(x) + m) = conv1x1(avgpool(x)) + avgpool(m).  Running the convolution on the
# low-resolution side does a quarter of its forward / input-gradient / weight-gradient work; results agree with the
# reference order to fp32 rounding (tests/test_gpu_step.py).  SP_COMMUTE_1X1=0 restores the reference order.
_COMMUTE_1X1 = CFG.commute_1x1
# The 2x2 average pooling behind the second convolution of a discriminator block (models.py:407-417, 452-462) is computed in
# that convolution's epilogue from the fp32 accumulators (one rounding instead of two; the full-resolution tensor never
# reaches HBM).  SP_FUSE_POOL2=0 runs the separate pooling kernel.
_FUSE_POOL2 = CFG.fuse_pool2
# The input of a discriminator block is read by LeakyReLU -> conv and by AvgPool -> 1x1 conv: one kernel produces both
# (ops.act_avgpool2), and one backward kernel replaces activation backward + pooling backward + the autograd sum.
_FUSE_ACT_POOL = CFG.fuse_act_pool
# CBN -> LeakyReLU -> UpsamplingBilinear2d of a generator block in one pass (ops.batch_norm(..., upsample=True)): measured
# SLOWER than the two kernels (897 vs 932 img/s: every source pixel is normalised by four output pixels and the class-gathered
# affine is re-read per output vector), so it is off by default; the operator stays for A/B runs.
_FUSE_BN_UPSAMPLE = CFG.fuse_bn_upsample
# The generator's final block (models.py:52-54): UpsamplingBilinear2d -> BatchNorm2d -> LeakyReLU at 256 x 256 computed from the
# 128 x 128 tensor, whose expansion is never written (6 instead of 17 low-resolution tensor volumes through HBM per forward).
_FUSE_UPSAMPLE_BN = CFG.fuse_upsample_bn
# The generator's masked-feature mappings on a side stream (Generator._map_features_ahead; a parallel branch of the captured graph):
# built in round 4 and measured SLOWER - 1 139 vs 1 159 img/s, same box, alternating runs: the convolution kernels are persistent
# (one block per CU walks the work items), so a second kernel on some CUs delays those blocks and with them the whole launch.
# Off by default; SP_SIDE_FEATURES=1 for A/B runs.
_SIDE_FEATURES = CFG.side_features
# The second generator forward of a step derives its masked-feature mappings from the first one's (Generator._feature_maps_of_the_step).
_REUSE_FEATURE_MAPS = CFG.reuse_feature_maps
# VGG-16 pass with gradient: pooled output + window positions instead of the unpooled tensor (_VGGPyramidFn, sp_conv_params.pool_idx).
_VGG_POOL_IDX = CFG.vgg_pool_idx


def init_weights(module: nn.Module) -> None:
    """Xavier-uniform weights / zero biases for linear and convolution layers (models.py:509-519)."""
    if isinstance(module, (SNConv2d, SNLinear)):
        nn.init.xavier_uniform_(module.weight_orig)
        module.bias.data.fill_(0.0)


def _class_index(class_id: torch.Tensor) -> torch.Tensor:
    """one-hot (B, classes) -> int64 (B,) like class_id.argmax(dim=-1) (models.py:151,501); indices pass through."""
    if class_id.dim() == 2:
        return class_id.argmax(dim=-1)
    return class_id.to(torch.int64)


def _collect_sn(root: nn.Module, no_dgrad=()):
    specs = []
    for name, m in root.named_modules():
        if isinstance(m, (SNConv2d, SNLinear, SNEmbedding)):
            specs.append((m, m._sn_kind, not any(name.endswith(s) for s in no_dgrad)))
    return specs


def _non_sn_params(root: nn.Module):
    """Parameters outside the spectral-normalised layers (conditional-BatchNorm embeddings, BatchNorm affine, attention gates):
    the bank gives them slots at the tail of its flat gradient buffer (ops.SpectralNormBank.collect_extra)."""
    sn = set()
    for m in root.modules():
        if isinstance(m, (SNConv2d, SNLinear, SNEmbedding)):
            sn.update(id(p) for p in m.parameters(recurse=False))
    return [p for p in root.parameters() if id(p) not in sn]


# --------------------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------------------
# nn.BatchNorm2d.num_batches_tracked += 1 per training forward (torch semantics): the generator bumps the counters of all its
# BatchNorm layers with ONE launch (they are views of one int64 tensor, Generator._tick_batch_counters) and raises this flag
# for the duration of its forward; a layer used on its own bumps its own counter.
_COUNTERS_TICKED = [False]


class _Duo:
    """A tensor of a two-group generator pass (Generator.forward_pair): `all` = both groups' images [g | d] as plain memory (what the
    launches over 2n images read and write, no autograd), `g` = the first group's images as a tensor of the autograd graph (an alias
    of the first half of `all`)."""
    __slots__ = ("all", "g")

    def __init__(self, all_, g):
        self.all, self.g = all_, g


class _GeneratorPair:
    """The two generator forwards of one training iteration (model_wrapper.py:144-151 without gradient - group d; :165-172 with -
    group g) as ONE pass over [g | d]: every convolution, resampling and attention launch of the stages below 256 x 256 runs once over
    2n images (at batch 20 these launches have 80 - 640 work items for 256 CUs), what is per forward - the BatchNorm statistics and
    their running averages (d first: the reference runs that forward first), the linear trunk, the 256 x 256 final block - runs per
    group.  The two forwards see the same weight_orig and differ in sigma (every forward advances the power iteration): launches use
    forward g's packing W / sigma_g with the per-group accumulator scales {1, sigma_g / sigma_d} (ops.conv_two_groups).  Group g's
    autograd graph is the one a forward of its own would have built: every operator's node is created around its slice of the joint
    result (ops.Dest), so the backward pass is untouched."""

    def __init__(self, bank, call_d, call_g, handles_g, scales, n: int):
        self.bank, self.call_d, self.call_g, self.handles_g, self.scales, self.n = bank, call_d, call_g, handles_g, scales, n
        self.cls2 = None                       # the class indices of both groups (the same samples twice), made on first use

    def as_d(self):
        self.bank.use_call(self.call_d, None)

    def as_g(self):
        self.bank.use_call(self.call_g, self.handles_g)

    def per_group(self, fn, x: _Duo, shape=None) -> _Duo:
        """fn(tensor, dest) -> tensor, applied to group d without autograd (first: forward #1 of the iteration) and to group g with; both
        write their half of one buffer."""
        n = self.n
        a = x.all
        shape = shape if shape is not None else (a.shape[1], a.shape[2], a.shape[3])
        out = ops.nhwc_empty(2 * n, shape[0], shape[1], shape[2], a.dtype, a.device)
        self.as_d()
        with torch.no_grad():
            fn(a[n:], ops.Dest(out[n:]))
        self.as_g()
        g = fn(x.g, ops.Dest(out[:n]))
        return _Duo(out, g)

    def batch_norm(self, module: "ConditionalBatchNorm", x: _Duo, cls: torch.Tensor, act: int, upsample: bool = False) -> _Duo:
        """A conditional BatchNorm (+ activation) over both groups, each normalised by ITS OWN batch statistics, in one launch set
        (sp_bn_stats_pair / sp_bn_apply_pair): the running statistics take group d's batch first (the forward the reference runs
        first), then group g's; group g's autograd node is built around its half."""
        n, a = self.n, x.all
        c, h, w = a.shape[1], a.shape[2], a.shape[3]
        bn = module.batch_norm
        dev, sd = a.device, ops.sp_dtype(a.dtype)
        if self.cls2 is None:
            self.cls2 = torch.cat([cls, cls])
        sums = torch.empty(1024 * 2 * c, dtype=torch.float32, device=dev)
        stat = torch.empty(4 * c, dtype=torch.float32, device=dev)                 # [mean g | mean d | invstd g | invstd d]
        mean2, invstd2 = stat[:2 * c], stat[2 * c:]
        ops.L.call("sp_bn_stats_pair", ops.ptr(a), 2 * n, n, h * w, c, ops.ptr(sums), bn.eps, bn.momentum, ops.ptr(bn.running_mean),
                   ops.ptr(bn.running_var), 1, ops.ptr(mean2), ops.ptr(invstd2), sd, ops.stream())
        if upsample:
            # ... and the bilinear x2 that follows in the same pass (sp_bn_apply_upsample2_pair): the normalised low-resolution tensor is
            # neither written nor read back (re-measured in round 5 with the row-per-block bilinear kernels: +0.25 % on two forwards)
            y = ops.nhwc_empty(2 * n, c, 2 * h, 2 * w, a.dtype, dev)
            ops.L.call("sp_bn_apply_upsample2_pair", ops.ptr(a), ops.ptr(y), 2 * n, n, h, w, c, ops.ptr(mean2), ops.ptr(invstd2), None, None,
                       ops.ptr(module.embedding.weight), ops.ptr(self.cls2), act, sd, ops.stream())
        else:
            y = ops.nhwc_empty(2 * n, c, h, w, a.dtype, dev)
            ops.L.call("sp_bn_apply_pair", ops.ptr(a), ops.ptr(y), 2 * n, n, h * w, c, ops.ptr(mean2), ops.ptr(invstd2), None, None,
                       ops.ptr(module.embedding.weight), ops.ptr(self.cls2), act, sd, ops.stream())
        g = module(x.g, cls, act, upsample, dest=ops.Dest(y[:n], True, (mean2[:c], invstd2[:c])))
        return _Duo(y, g)

    def conv(self, module, x: _Duo, act: int = ACT_NONE, res1: Optional[_Duo] = None, res2: Optional[_Duo] = None) -> _Duo:
        n = self.n
        pl = self.call_g.layers[module._sn_slot]
        y = ops.conv_two_groups(x.all, module, pl, self.scales.data_ptr() + 8 * pl.slot, n, act,
                                res1.all if res1 is not None else None, res2.all if res2 is not None else None)
        self.as_g()
        g = module(x.g, act, res1.g if res1 is not None else None, res2.g if res2 is not None else None, dest=ops.Dest(y[:n], True))
        return _Duo(y, g)

    def upsample2(self, x: _Duo) -> _Duo:
        n, a = self.n, x.all
        y = ops.nhwc_empty(2 * n, a.shape[1], 2 * a.shape[2], 2 * a.shape[3], a.dtype, a.device)
        ops.L.call("sp_upsample2_fwd", ops.ptr(a), ops.ptr(y), 2 * n, a.shape[2], a.shape[3], a.shape[1], ops.sp_dtype(a.dtype), ops.stream())
        return _Duo(y, ops.upsample2(x.g, ops.Dest(y[:n], True)))

    def maxpool2(self, x: _Duo) -> _Duo:
        n, a = self.n, x.all
        y = ops.nhwc_empty(2 * n, a.shape[1], a.shape[2] // 2, a.shape[3] // 2, a.dtype, a.device)
        ops.L.call("sp_maxpool2_fwd", ops.ptr(a), ops.ptr(y), 2 * n, a.shape[2], a.shape[3], a.shape[1], 0, ops.sp_dtype(a.dtype), ops.stream())
        return _Duo(y, ops.maxpool2(x.g, ops.Dest(y[:n], True)))

    def attention(self, q: _Duo, k: _Duo, v: _Duo) -> _Duo:
        n = self.n
        o, lse = ops.attention_raw(q.all, k.all, v.all)
        return _Duo(o, ops.attention_core(q.g, k.g, v.g, ops.Dest(o[:n], True, lse[:n])))

    def scale_add(self, a: _Duo, b: _Duo, gamma) -> _Duo:
        n = self.n
        y = torch.empty_like(a.all)
        ops.L.call("sp_scale_add", ops.ptr(a.all), ops.ptr(b.all), ops.ptr(gamma), ops.ptr(y), a.all.numel(), ops.sp_dtype(y.dtype), ops.stream())
        return _Duo(y, ops.scale_add(a.g, b.g, gamma, ops.Dest(y[:n], True)))


class ConditionalBatchNorm(nn.Module):
    """models.py:469-506."""

    def __init__(self, num_features: int, number_of_classes: int = 365) -> None:
        super().__init__()
        self.batch_norm = nn.BatchNorm2d(num_features=num_features, momentum=0.001, affine=False)
        self.embedding = nn.Embedding(num_embeddings=number_of_classes, embedding_dim=num_features * 2)
        self.embedding.weight.data[:, :num_features].fill_(1.0)
        self.embedding.weight.data[:, num_features:].zero_()

    def forward(self, input: torch.Tensor, class_id: torch.Tensor, act: int = ACT_NONE, upsample: bool = False, dest=None) -> torch.Tensor:
        bn = self.batch_norm
        if self.training and not _COUNTERS_TICKED[0]:
            bn.num_batches_tracked.add_(1)
        return ops.batch_norm(input, None, None, self.embedding.weight, _class_index(class_id), bn.running_mean, bn.running_var,
                              bn.momentum, bn.eps, self.training, act, upsample, dest)


class SelfAttention(nn.Module):
    """models.py:219-275."""

    def __init__(self, channels: int) -> None:
        super().__init__()
        self.query_convolution = SNConv2d(channels, channels // 8, 1)
        self.key_convolution = SNConv2d(channels, channels // 8, 1)
        self.value_convolution = SNConv2d(channels, channels // 2, 1)
        self.attention_convolution = SNConv2d(channels // 2, channels, 1)
        self.max_pooling = nn.MaxPool2d(kernel_size=2, stride=2, padding=0)
        self.gamma = nn.Parameter(torch.ones(1, dtype=torch.float32))

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        pooled = ops.maxpool2(input)
        q = self.query_convolution(input)
        k = self.key_convolution(pooled)
        v = self.value_convolution(pooled)
        o = ops.attention_core(q, k, v)
        o = self.attention_convolution(o)
        return ops.scale_add(o, input, self.gamma)

    def forward_pair(self, x: _Duo, pp: _GeneratorPair) -> _Duo:
        """forward() over the two groups of a generator pair pass (every launch over 2n images)."""
        pooled = pp.maxpool2(x)
        q = pp.conv(self.query_convolution, x)
        k = pp.conv(self.key_convolution, pooled)
        v = pp.conv(self.value_convolution, pooled)
        o = pp.conv(self.attention_convolution, pp.attention(q, k, v))
        return pp.scale_add(o, x, self.gamma)


class GeneratorResidualBlock(nn.Module):
    """models.py:278-339."""

    def __init__(self, in_channels: int, out_channels: int, feature_channels: int, number_of_classes: int = 365) -> None:
        super().__init__()
        self.main_block = nn.ModuleList([
            ConditionalBatchNorm(num_features=in_channels, number_of_classes=number_of_classes),
            nn.LeakyReLU(negative_slope=0.2),
            nn.UpsamplingBilinear2d(scale_factor=2),
            SNConv2d(in_channels, out_channels, 3),
            ConditionalBatchNorm(num_features=out_channels, number_of_classes=number_of_classes),
            nn.LeakyReLU(negative_slope=0.2),
            SNConv2d(out_channels, out_channels, 3)])
        self.residual_mapping = nn.Sequential(nn.UpsamplingBilinear2d(scale_factor=2), SNConv2d(in_channels, out_channels, 1))
        self.masked_feature_mapping = SNConv2d(feature_channels, out_channels, 3)

    def forward(self, input: torch.Tensor, masked_features: torch.Tensor, class_id: torch.Tensor, mapped=None) -> torch.Tensor:
        """mapped: the output of masked_feature_mapping computed ahead of time (Generator's side stream), as a zero-argument
        callable that makes it visible to the current stream and returns it."""
        cls = _class_index(class_id)
        if _FUSE_BN_UPSAMPLE:
            h = self.main_block[3](self.main_block[0](input, cls, ACT_LRELU, upsample=True))     # CBN + LeakyReLU + bilinear x2 in one pass
        else:
            h = self.main_block[0](input, cls, ACT_LRELU)             # CBN + LeakyReLU fused
            h = self.main_block[3](ops.upsample2(h))
        h = self.main_block[4](h, cls, ACT_LRELU)
        if _COMMUTE_1X1:
            r = ops.upsample2(self.residual_mapping[1](input))
        else:
            r = self.residual_mapping[1](ops.upsample2(input))
        f = mapped() if mapped is not None else self.masked_feature_mapping(masked_features)
        return self.main_block[6](h, ACT_NONE, r, f)                  # (main + residual) + features in the epilogue

    def forward_pair(self, x: _Duo, f: _Duo, cls: torch.Tensor, pp: _GeneratorPair) -> _Duo:
        """forward() over the two groups of a generator pair pass: the conditional BatchNorms per group (their statistics belong to one
        forward), everything else once over 2n images.  f: this block's masked-feature mapping of both groups."""
        if CFG.bn_pair:
            if CFG.bn_pair_upsample:
                h = pp.conv(self.main_block[3], pp.batch_norm(self.main_block[0], x, cls, ACT_LRELU, upsample=True))
            else:
                h = pp.conv(self.main_block[3], pp.upsample2(pp.batch_norm(self.main_block[0], x, cls, ACT_LRELU)))
            h = pp.batch_norm(self.main_block[4], h, cls, ACT_LRELU)
        else:
            h = pp.per_group(lambda t, dest: self.main_block[0](t, cls, ACT_LRELU, dest=dest), x)
            h = pp.conv(self.main_block[3], pp.upsample2(h))
            h = pp.per_group(lambda t, dest: self.main_block[4](t, cls, ACT_LRELU, dest=dest), h)
        if _COMMUTE_1X1:
            r = pp.upsample2(pp.conv(self.residual_mapping[1], x))
        else:
            r = pp.conv(self.residual_mapping[1], pp.upsample2(x))
        return pp.conv(self.main_block[6], h, ACT_NONE, r, f)


class LinearBlock(nn.Module):
    """models.py:342-375.  ``act_out`` fuses the LeakyReLU that the only consumer applies first."""

    def __init__(self, in_features: int, out_features: int, feature_size: int) -> None:
        super().__init__()
        self.main_block = nn.Sequential(nn.LeakyReLU(negative_slope=0.2), SNLinear(in_features, out_features))
        self.masked_feature_mapping = SNLinear(feature_size, out_features)

    def forward(self, input: torch.Tensor, masked_features: torch.Tensor, act_out: int = ACT_NONE,
                input_is_activated: bool = False, mapped=None) -> torch.Tensor:
        if not input_is_activated:
            input = ops.activation(input, ACT_LRELU)
        mapped = mapped() if mapped is not None else self.masked_feature_mapping(masked_features)
        return self.main_block[1](input, act_out, mapped)


class DiscriminatorInputResidualBlock(nn.Module):
    """models.py:378-419."""

    def __init__(self, in_channels: int, out_channels: int) -> None:
        super().__init__()
        self.main_block = nn.Sequential(SNConv2d(in_channels, out_channels, 3), nn.LeakyReLU(negative_slope=0.2),
                                        SNConv2d(out_channels, out_channels, 3))
        self.residual_mapping = SNConv2d(in_channels, out_channels, 1)
        self.downsampling = nn.AvgPool2d(kernel_size=(2, 2))

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        m = self.main_block[0](input, ACT_LRELU, premasked=_FUSE_LRELU_BWD)      # LeakyReLU backward rides in main_block[2]'s dgrad epilogue
        if self.main_block[2].pool2_ok(m):
            mp = self.main_block[2](m, mask_input=_FUSE_LRELU_BWD, pool2=True)
        else:
            mp = ops.avgpool2(self.main_block[2](m, mask_input=_FUSE_LRELU_BWD))
        return self.residual_mapping(ops.avgpool2(input), ACT_NONE, mp)


class DiscriminatorResidualBlock(nn.Module):
    """models.py:422-466."""

    def __init__(self, in_channels: int, out_channels: int) -> None:
        super().__init__()
        self.main_block = nn.Sequential(nn.LeakyReLU(negative_slope=0.2), SNConv2d(in_channels, out_channels, 3),
                                        nn.LeakyReLU(negative_slope=0.2), SNConv2d(out_channels, out_channels, 3))
        self.residual_mapping = SNConv2d(in_channels, out_channels, 1)
        self.downsampling = nn.AvgPool2d(kernel_size=(2, 2))

    def forward(self, input: torch.Tensor, input_activated: Optional[torch.Tensor] = None, act_out: int = ACT_NONE):
        pooled = None
        if input_activated is None:
            if _COMMUTE_1X1 and _FUSE_ACT_POOL:
                input_activated, pooled = ops.act_avgpool2(input, ACT_LRELU)     # both consumers of `input` in one pass
            else:
                input_activated = ops.activation(input, ACT_LRELU)
        m = self.main_block[1](input_activated, ACT_LRELU, premasked=_FUSE_LRELU_BWD)
        if _COMMUTE_1X1:
            if self.main_block[3].pool2_ok(m):
                sp = self.main_block[3](m, ACT_NONE, None, mask_input=_FUSE_LRELU_BWD, pool2=True)
            else:
                sp = ops.avgpool2(self.main_block[3](m, ACT_NONE, None, mask_input=_FUSE_LRELU_BWD))
            if pooled is None:
                pooled = ops.avgpool2(input)
            out = self.residual_mapping(pooled, ACT_NONE, sp)                               # pooled residual + pooled main
            return out if act_out == ACT_NONE else (out, ops.activation(out, act_out))
        r = self.residual_mapping(input)
        s = self.main_block[3](m, ACT_NONE, r, mask_input=_FUSE_LRELU_BWD)
        return ops.avgpool2(s, act_out)       # act_out != NONE -> (pooled, lrelu(pooled)) for the next block


# --------------------------------------------------------------------------------------------------
# generator
# --------------------------------------------------------------------------------------------------
class Generator(nn.Module):
    """models.py:10-99."""

    def __init__(self, out_channels: int = 3, latent_dimensions: int = 128, channels_factor: Union[int, float] = 1,
                 number_of_classes: int = 365) -> None:
        super().__init__()
        self.latent_dimensions = latent_dimensions
        ch = lambda n: int(n // channels_factor)       # noqa: E731  (the factor DIVIDES, models.py:34)
        self.linear_layer = SNLinear(latent_dimensions, latent_dimensions)
        self.linear_block_1 = LinearBlock(in_features=latent_dimensions, out_features=365, feature_size=365)
        self.linear_block_2 = LinearBlock(in_features=365, out_features=2048, feature_size=4096)
        self.convolution_layer = nn.Sequential(nn.LeakyReLU(negative_slope=0.2), SNConv2d(128, ch(512), 1))
        self.main_path = nn.ModuleList([
            GeneratorResidualBlock(ch(512), ch(512), 513, number_of_classes),
            GeneratorResidualBlock(ch(512), ch(512), 513, number_of_classes),
            GeneratorResidualBlock(ch(512), ch(256), 257, number_of_classes),
            SelfAttention(channels=ch(256)),
            GeneratorResidualBlock(ch(256), ch(128), 129, number_of_classes),
            GeneratorResidualBlock(ch(128), ch(64), 65, number_of_classes)])
        self.final_block = nn.Sequential(
            nn.UpsamplingBilinear2d(scale_factor=2),
            nn.BatchNorm2d(ch(64)),
            nn.LeakyReLU(negative_slope=0.2),
            SNConv2d(ch(64), ch(64), 3),
            nn.LeakyReLU(negative_slope=0.2),
            SNConv2d(ch(64), out_channels, 1))
        self.apply(init_weights)
        self._bank = ops.SpectralNormBank(_collect_sn(self, no_dgrad=("masked_feature_mapping",)), _non_sn_params(self))
        self._bn_list = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        self._nbt_flat = None
        self._side_streams = {}
        self.map_mode = None               # "stash" / "reuse": set by ModelWrapper for the next forward (_feature_maps_of_the_step)
        self._map_stash = None

    def _feature_maps_of_the_step(self, features, masks):
        """The seven masked-feature mappings (models.py:78-94) see the SAME pyramid, masks and weight_orig in both generator forwards
        of a training step (model_wrapper.py:147-151 without gradient, :168-172 with): only sigma differs, because every forward
        advances the power iteration.  ModelWrapper announces the first forward (map_mode = "stash": the masked inputs, the outputs
        and the packed layers are kept) and the second (map_mode = "reuse"): conv(x, W / sigma_2) + b = (conv(x, W / sigma_1) + b - b)
        * sigma_1 / sigma_2 + b - one elementwise pass per mapping instead of five 3x3 convolutions, two linear layers and the seven
        masking passes; the weight gradients of the second forward are taken from the stashed inputs as usual (ops._ReusedLayerFn).
        Anything that does not match the stash (other tensors, shapes, dtype) falls back to computing the mappings."""
        mode, self.map_mode = self.map_mode, None
        if mode not in ("stash", "reuse") or not _REUSE_FEATURE_MAPS:
            self._map_stash = None
            return {}
        blocks = [(self.linear_block_1, True), (self.linear_block_2, True)] + [(m, False) for m in self.main_path
                                                                               if isinstance(m, GeneratorResidualBlock)]
        key = (ops.compute_dtype(),) + tuple((f.data_ptr(), tuple(f.shape)) for f in features) + tuple((m.data_ptr(), tuple(m.shape)) for m in masks)
        depth = len(features) - 1
        out = {}
        if mode == "reuse":
            stash, self._map_stash = self._map_stash, None
            if stash is None or stash["key"] != key:
                return {}
            for (block, _), (src, f, pl) in zip(blocks, stash["maps"]):
                out[depth] = (lambda block=block, src=src, f=f, pl=pl: ops.reused_layer(block.masked_feature_mapping, src, f, pl))
                depth -= 1
            return out
        maps = []
        for block, is_linear in blocks:
            src = ops.mask_mul_2d(features[depth], masks[depth]) if is_linear else ops.mask_concat(features[depth], masks[depth])
            layer = block.masked_feature_mapping
            f = layer(src)
            maps.append((src, f.detach(), ops.packed_layer(layer, layer.training, f.dtype, f.device)))
            out[depth] = (lambda f=f: f)
            depth -= 1
        self._map_stash = {"key": key, "maps": maps}
        return out

    def _map_features_ahead(self, features, masks):
        """The seven masked-feature mappings (models.py:78-94: mask * feature [cat mask] -> spectral-norm linear / 3x3 convolution)
        depend on the frozen VGG-16 pyramid only, not on the activations of the generator: they are enqueued on a SIDE STREAM at the
        start of the forward pass and joined (an event per mapping) where a block adds its mapping in.  Their kernels - and, since
        autograd runs a node's backward on the stream of its forward, their weight-gradient kernels - then overlap the low-resolution
        stages of the main path, whose launches are too small to fill 256 CUs on their own (4x4 ... 32x32 maps).  Under HIP-graph
        capture the side stream becomes a parallel branch of the graph.  Returns {pyramid depth: callable -> mapped tensor}."""
        dev = features[0].device
        if dev.type != "cuda":
            return {}
        main = torch.cuda.current_stream(dev)
        side = self._side_streams.get(dev)
        if side is None:
            side = self._side_streams[dev] = torch.cuda.Stream(dev)
        side.wait_stream(main)                       # the features / masks / packed weights were produced on the main stream
        out = {}
        depth = len(features) - 1
        with torch.cuda.stream(side):
            jobs = [(self.linear_block_1, True), (self.linear_block_2, True)] + [(m, False) for m in self.main_path
                                                                                 if isinstance(m, GeneratorResidualBlock)]
            for block, is_linear in jobs:
                src = ops.mask_mul_2d(features[depth], masks[depth]) if is_linear else ops.mask_concat(features[depth], masks[depth])
                f = block.masked_feature_mapping(src)
                ev = torch.cuda.Event()
                ev.record(side)

                def visible(f=f, ev=ev, main=main):
                    torch.cuda.current_stream(f.device).wait_event(ev)
                    f.record_stream(torch.cuda.current_stream(f.device))     # allocated on the side stream, consumed (and freed) on this one
                    return f
                out[depth] = visible
                depth -= 1
        return out

    def _tick_batch_counters(self) -> None:
        """num_batches_tracked += 1 for every BatchNorm layer with one launch: the counters are 0-dim views of one int64
        tensor (re-linked whenever .to() / load_state_dict(assign=True) replaced the buffers)."""
        bns, flat = self._bn_list, self._nbt_flat
        linked = flat is not None and all(b.num_batches_tracked.data_ptr() == flat.data_ptr() + 8 * i for i, b in enumerate(bns))
        if not linked:
            flat = torch.stack([b.num_batches_tracked.detach().reshape(()) for b in bns])
            for i, b in enumerate(bns):
                b._buffers["num_batches_tracked"] = flat[i]
            self._nbt_flat = flat
        flat.add_(1)

    def forward(self, input: torch.Tensor, features: List[torch.Tensor], masks: List[torch.Tensor] = None,
                class_id: torch.Tensor = None) -> torch.Tensor:
        dt = ops.compute_dtype()
        ops.require_gpu(input)
        self._bank.begin(self.training, dt, input.device)
        ticked_before = _COUNTERS_TICKED[0]
        try:
            if self.training:
                self._tick_batch_counters()
                _COUNTERS_TICKED[0] = True
            cls = _class_index(class_id)
            depth = len(features) - 1
            mapped = self._map_features_ahead(features, masks) if _SIDE_FEATURES else self._feature_maps_of_the_step(features, masks)
            # the latent's requires_grad (model_wrapper.py:148) is a dead gradient: detach
            x = self.linear_layer(ops.as_rows(input.detach(), dt), ACT_LRELU)
            x = self.linear_block_1(x, None if depth in mapped else ops.mask_mul_2d(features[depth], masks[depth]), ACT_LRELU, True,
                                    mapped.get(depth))
            depth -= 1
            x = self.linear_block_2(x, None if depth in mapped else ops.mask_mul_2d(features[depth], masks[depth]), ACT_LRELU, True,
                                    mapped.get(depth))
            depth -= 1
            x = ops.rows_to_nhwc(x, x.shape[1] // 16, 4, 4)            # view(B, -1, 4, 4) of the NCHW reference
            x = self.convolution_layer[1](x)
            for layer in self.main_path:
                if isinstance(layer, SelfAttention):
                    x = layer(x)
                else:
                    x = layer(x, None if depth in mapped else ops.mask_concat(features[depth], masks[depth]), cls, mapped.get(depth))
                    depth -= 1
            return self._final_block(x)
        finally:
            _COUNTERS_TICKED[0] = ticked_before
            self._bank.end()

    def _final_block(self, x: torch.Tensor) -> torch.Tensor:
        """models.py:51-61 on the 128 x 128 tensor of one forward."""
        fb = self.final_block
        bn = fb[1]
        if _FUSE_UPSAMPLE_BN:
            # UpsamplingBilinear2d -> BatchNorm2d -> LeakyReLU on the expansion of x without writing it (sp_bn_*_up2)
            x = ops.batch_norm(x, bn.weight, bn.bias, None, None, bn.running_mean, bn.running_var, bn.momentum, bn.eps, self.training,
                               ACT_LRELU, "before")
        else:
            x = ops.batch_norm(ops.upsample2(x), bn.weight, bn.bias, None, None, bn.running_mean, bn.running_var, bn.momentum,
                               bn.eps, self.training, ACT_LRELU)
        if ops.conv_tail_ok(x, fb[3], fb[5]):
            try:
                return ops.sn_conv2d_tail(x, fb[3], ACT_LRELU, fb[5], ACT_TANH)  # no-grad pass: the 1x1 + tanh ride in the 3x3's epilogue
            except ops.L.SempyrError:
                # the library's own admission test is stricter than conv_tail_ok (its A/B tuning keys, operands of 1 GiB or more -
                # a no-grad batch of 128 images): the two layers then run one after the other, as below (round-4 ADVICE)
                pass
        if ops.conv_tail_ok(x, fb[3], fb[5], with_grad=True):
            # the pass WITH autograd (round 5): the same single launch, which here also stores the 64-channel tensor (the 1x1 layer's
            # weight gradient and the LeakyReLU's derivative read it); the two layers' autograd nodes are built around the results
            try:
                mid, img = ops.sn_conv2d_tail(x, fb[3], ACT_LRELU, fb[5], ACT_TANH, keep_mid=True)
            except ops.L.SempyrError:
                mid = None
            if mid is not None:
                h = fb[3](x, ACT_LRELU, premasked=_FUSE_LRELU_BWD, dest=ops.Dest(mid, True))
                return fb[5](h, ACT_TANH, mask_input=_FUSE_LRELU_BWD, dest=ops.Dest(img, True))
        # (the LeakyReLU between the two convolutions: its backward rides in the 1x1's input-gradient epilogue - the separate pass
        # read and wrote the 64-channel 256 x 256 gradient once more: 84 us per step)
        x = fb[3](x, ACT_LRELU, premasked=_FUSE_LRELU_BWD)
        return fb[5](x, ACT_TANH, mask_input=_FUSE_LRELU_BWD)

    def forward_pair(self, input_g: torch.Tensor, input_d: torch.Tensor, features: List[torch.Tensor], masks: List[torch.Tensor],
                     class_id: torch.Tensor):
        """``(self(input_g, ...), [no_grad] self(input_d, ...))`` where the reference calls self(input_d) FIRST (the discriminator step's
        fake images, model_wrapper.py:144-151) and self(input_g) second (the generator step, :165-172): both forwards of one training
        iteration in one pass (_GeneratorPair).  Returns (images of group g with their autograd graph, images of group d without).
        Spectral-norm u / v advance twice, the BatchNorm running statistics take forward d's batch first, then forward g's, and
        num_batches_tracked counts both - as two calls would leave them.  Training mode, 16-bit or fp32 storage."""
        dt = ops.compute_dtype()
        ops.require_gpu(input_g)
        if not self.training or input_g.shape != input_d.shape:
            raise ops.L.SempyrError("Generator.forward_pair: training mode and two latent batches of one shape")
        n = input_g.shape[0]
        bank = self._bank
        dev = input_g.device
        if getattr(self, "_pair_slots", None) is None:
            # the convolutions of the joint stages take forward g's packing (+ the per-group scale): forward d does not pack them.  It
            # keeps what runs per group on its own sigma: the linear trunk, the masked-feature mappings, the final block
            joint = [self.convolution_layer[1]]
            for m in self.main_path:
                joint += ([m.query_convolution, m.key_convolution, m.value_convolution, m.attention_convolution] if isinstance(m, SelfAttention)
                          else [m.main_block[3], m.main_block[6], m.residual_mapping[1]])
            ids = {id(m) for m in joint}
            self._pair_slots = frozenset(i for i, (m, _, _) in enumerate(bank.specs) if id(m) in ids)
        with torch.no_grad():
            call_d = bank.begin(True, dt, dev, skip_pack=self._pair_slots if CFG.sn_skip_pack else None)   # forward #1: own power iteration, no autograd handles
        call_g = bank.begin(True, dt, dev)                     # forward #2
        handles_g = bank.handles
        scales = torch.empty(2 * len(bank.specs), dtype=torch.float32, device=dev)
        # per layer {1, sigma_g / sigma_d}: the launches use g's packing, group d's accumulators are rescaled to its own sigma
        ops.L.call("sp_sn_pair_scales", ops.ptr(bank.table_dev), len(bank.specs), ops.ptr(call_g.scratch), ops.ptr(call_d.scratch), ops.ptr(scales),
                   ops.stream())
        pp = _GeneratorPair(bank, call_d, call_g, handles_g, scales, n)
        ticked_before = _COUNTERS_TICKED[0]
        try:
            self._tick_batch_counters()
            self._tick_batch_counters()
            _COUNTERS_TICKED[0] = True
            self.map_mode, self._map_stash = None, None
            cls = _class_index(class_id)
            # the seven masked-feature mappings (models.py:78-94): group d's from the layer (sigma_d), group g's derived from them
            # (ops.reused_layer: only sigma differs; its weight gradients come from the shared masked input)
            maps = {}
            depth = len(features) - 1
            blocks = [(self.linear_block_1, True), (self.linear_block_2, True)] + [(m, False) for m in self.main_path
                                                                                   if isinstance(m, GeneratorResidualBlock)]
            for block, is_linear in blocks:
                layer = block.masked_feature_mapping
                src = ops.mask_mul_2d(features[depth], masks[depth]) if is_linear else ops.mask_concat(features[depth], masks[depth])
                pp.as_d()
                with torch.no_grad():
                    if is_linear:
                        f_d = layer(src)
                        both = None
                    else:
                        both = ops.nhwc_empty(2 * n, layer.out_channels, src.shape[2], src.shape[3], dt, dev)
                        f_d = layer(src, dest=ops.Dest(both[n:]))
                    pl_d = ops.packed_layer(layer, True, dt, dev)
                pp.as_g()
                f_g = ops.reused_layer(layer, src, f_d, pl_d, None if is_linear else ops.Dest(both[:n]))
                maps[depth] = (f_d, f_g) if is_linear else _Duo(both, f_g)
                depth -= 1
            depth = len(features) - 1
            # the linear trunk per group (its launches stream weights; batch rows cost nothing there)
            rows = []
            for z, first, grad in ((input_d, True, False), (input_g, False, True)):
                (pp.as_g if grad else pp.as_d)()
                with torch.set_grad_enabled(grad):
                    pick = (lambda t: t[1]) if grad else (lambda t: t[0])
                    x = self.linear_layer(ops.as_rows(z.detach(), dt), ACT_LRELU)
                    x = self.linear_block_1(x, None, ACT_LRELU, True, (lambda d=depth: pick(maps[d])))
                    x = self.linear_block_2(x, None, ACT_LRELU, True, (lambda d=depth - 1: pick(maps[d])))
                    rows.append(x)
            depth -= 2
            c4 = rows[0].shape[1] // 16
            x_all = ops.nhwc_empty(2 * n, c4, 4, 4, dt, dev)
            with torch.no_grad():
                ops.rows_to_nhwc(rows[0], c4, 4, 4, ops.Dest(x_all[n:]))
            x = _Duo(x_all, ops.rows_to_nhwc(rows[1], c4, 4, 4, ops.Dest(x_all[:n])))
            x = pp.conv(self.convolution_layer[1], x)
            for layer in self.main_path:
                if isinstance(layer, SelfAttention):
                    x = layer.forward_pair(x, pp)
                else:
                    x = layer.forward_pair(x, maps[depth], cls, pp)
                    depth -= 1
            # the 256 x 256 final block per group: whole rounds of work items at either batch size, and group d (no autograd) takes its
            # 1x1 + tanh in the 3x3's epilogue
            pp.as_d()
            with torch.no_grad():
                fake_d = self._final_block(x.all[n:])
            pp.as_g()
            fake_g = self._final_block(x.g)
            return fake_g, fake_d
        finally:
            _COUNTERS_TICKED[0] = ticked_before
            bank.end()


# --------------------------------------------------------------------------------------------------
# discriminator
# --------------------------------------------------------------------------------------------------
class Discriminator(nn.Module):
    """models.py:102-155.  Returns the (B, B, 128) tensor the reference returns."""

    def __init__(self, in_channels: int = 3, channel_factor: Union[int, float] = 1, number_of_classes: int = 365):
        super().__init__()
        ch = lambda n: int(n // channel_factor)        # noqa: E731
        self.layers = nn.Sequential(
            DiscriminatorInputResidualBlock(in_channels, ch(64)),
            DiscriminatorResidualBlock(ch(64), ch(128)),
            DiscriminatorResidualBlock(ch(128), ch(256)),
            SelfAttention(channels=ch(256)),
            DiscriminatorResidualBlock(ch(256), ch(256)),
            DiscriminatorResidualBlock(ch(256), ch(256)),
            DiscriminatorResidualBlock(ch(256), ch(512)),
            DiscriminatorResidualBlock(ch(512), ch(768)),
            nn.LeakyReLU(negative_slope=0.2),
            nn.AdaptiveAvgPool2d(output_size=(1, 1)),
            nn.Flatten(start_dim=1),
            SNLinear(ch(768), 128),
            nn.LeakyReLU(negative_slope=0.2))
        self.classification = SNLinear(128, 1)
        self.classification._sn_kind = "plain"          # consumed as an fp32 vector by the head kernel
        self.embedding = SNEmbedding(number_of_classes, 128)
        self.apply(init_weights)
        self._bank = ops.SpectralNormBank(_collect_sn(self), _non_sn_params(self))

    def forward(self, input: torch.Tensor, class_id: torch.Tensor) -> torch.Tensor:
        dt = ops.compute_dtype()
        ops.require_gpu(input)
        self._bank.begin(self.training, dt, input.device)
        try:
            L = self.layers
            x = L[0](ops.ingest_image(input, dt))
            if _COMMUTE_1X1 and _FUSE_ACT_POOL:
                # every block forms lrelu(x) and avgpool(x) of its own input in one pass (ops.act_avgpool2)
                x = L[3](L[2](L[1](x)))
                x = L[7](L[6](L[5](L[4](x))))
                x = ops.adaptive_avgpool(x, 1, 1, ACT_LRELU).flatten(1)
                x = L[11](x, ACT_LRELU)
                return ops.discriminator_head(x, self.embedding, self.classification, _class_index(class_id))
            x, xa = L[1](x, None, ACT_LRELU)
            x = L[2](x, xa, ACT_NONE)                    # raw output feeds the attention block
            x = L[3](x)
            x, xa = L[4](x, None, ACT_LRELU)
            x, xa = L[5](x, xa, ACT_LRELU)
            x, xa = L[6](x, xa, ACT_LRELU)
            x = L[7](x, xa, ACT_NONE)
            x = ops.adaptive_avgpool(x, 1, 1, ACT_LRELU).flatten(1)    # LeakyReLU -> AdaptiveAvgPool2d(1) -> Flatten
            x = L[11](x, ACT_LRELU)
            return ops.discriminator_head(x, self.embedding, self.classification, _class_index(class_id))
        finally:
            self._bank.end()


    def forward_pair(self, input_a: torch.Tensor, input_b: torch.Tensor, class_id: torch.Tensor):
        """``(self(input_a, class_id), self(input_b, class_id))`` - the discriminator step's D(real), D(fake)
        (model_wrapper.py:153-155) - with the convolution trunk run ONCE over the 2B images: the two forwards share weight_orig
        and differ only in the spectral-norm sigma (every forward advances the power iteration), which a per-group accumulator
        scale in the convolution epilogue absorbs (ops.PairPass).  Twice the work items per launch (B = 20: 640 -> 1 280 items
        of the 256-channel layers on 256 CUs - whole rounds instead of 2.5), half the launches.  The layers behind the trunk
        (768 -> 128 linear, the (B, B, 128) head, which couples the samples of ONE forward) run per group.  Results equal two
        separate calls up to fp32 rounding of the scale; no image gradients (the D step needs none)."""
        dt = ops.compute_dtype()
        ops.require_gpu(input_a)
        if not (_COMMUTE_1X1 and _FUSE_ACT_POOL):
            return self(input_a, class_id), self(input_b, class_id)
        bank = self._bank
        na = input_a.shape[0]
        if getattr(self, "_trunk_slots", None) is None:      # every convolution runs on forward a's packing in the two-group pass
            self._trunk_slots = frozenset(i for i, (m, _, _) in enumerate(bank.specs) if isinstance(m, SNConv2d))
        pair = bank.begin_pair(self.training, dt, input_a.device, na, self._trunk_slots)
        try:
            L = self.layers
            x = L[0](ops.ingest_image_pair(input_a, input_b, dt))
            x = L[3](L[2](L[1](x)))
            x = L[7](L[6](L[5](L[4](x))))
            x = ops.adaptive_avgpool(x, 1, 1, ACT_LRELU).flatten(1)
            xa, xb = ops.split_rows(x, na)
            cls = _class_index(class_id)
            out = []
            for rows, call, handles in ((xa, pair.call_a, pair.handles_a), (xb, pair.call_b, pair.handles_b)):
                bank.use_call(call, handles)
                h = L[11](rows, ACT_LRELU)
                out.append(ops.discriminator_head(h, self.embedding, self.classification, cls))
            return out[0], out[1]
        finally:
            bank.end()


# --------------------------------------------------------------------------------------------------
# frozen VGG-16 feature pyramid
# --------------------------------------------------------------------------------------------------
_VGG_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
_IMAGENET_MEAN = (0.485, 0.456, 0.406)
_IMAGENET_STD = (0.229, 0.224, 0.225)


class _VGG16Topology(nn.Module):
    """Parameter container with torchvision's vgg16 attribute layout (features / avgpool / classifier)."""

    def __init__(self, num_classes: int) -> None:
        super().__init__()
        layers, c = [], 3
        for v in _VGG_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.ModuleList(layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.ModuleList([nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 4096),
                                         nn.ReLU(True), nn.Dropout(), nn.Linear(4096, num_classes)])


class _VGGPyramidFn(torch.autograd.Function):
    """Whole frozen pyramid as one autograd node: forward keeps the post-ReLU activations, backward is a pure
    input-gradient chain (no weight gradients exist, model_wrapper.py:67-68) with the ReLU derivative folded
    into each dgrad epilogue (mask_src) and into the max-pool backward."""

    @staticmethod
    def forward(ctx, img, packs, dtype, img_b=None):
        """img_b (a second batch that needs NO gradient, e.g. the next iteration's real images): the pyramid runs ONCE over [img | img_b]
        - the 16 x 16 / 32 x 32 stages fill 31 - 62 % of the CUs at batch 20 - and returns the seven features of img followed by the
        seven of img_b; the backward walks only the first img.shape[0] images of every saved tensor (a prefix: same base addresses)."""
        import ctypes
        Lb = ops.L
        dev = img.device
        # (backward() tests every incoming gradient for None: without this, autograd materialises a zero tensor for each of the seven
        # non-differentiable features of the second group in every backward pass - seven fill launches per step, round-5 timeline)
        ctx.set_materialize_grads(False)
        ops.PROBE_NET[0] = "vgg"          # bench.py's launch probe: everything until the return below belongs to the frozen pyramid
        n_grad = img.shape[0]
        n = n_grad + (img_b.shape[0] if img_b is not None else 0)
        scale = tuple(1.0 / s for s in _IMAGENET_STD)
        shift = tuple(-m / s for m, s in zip(_IMAGENET_MEAN, _IMAGENET_STD))
        with torch.no_grad():
            if img_b is None:
                x = ops.ingest_image(img.detach(), dtype, scale, shift)
            else:
                x = ops.ingest_image_pair(img, img_b, dtype, scale, shift)
        acts = [x]                 # conv inputs / pool inputs, in order
        feats = []
        h = w = img.shape[2]
        ci = 0
        trace = []                 # ('conv', idx_in_acts_of_input, pack) | ('pool', idx_of_input) | ('poolidx', idx_of_pooled_output, slot, h, w)
        pool_idx = []              # window positions recorded by the fused ReLU + MaxPool epilogues of the pass with gradient
        # no-gradient pass (features of the real images, model_wrapper.py:139-141): the unpooled output of a stage's last
        # convolution is never looked at again, so its ReLU + MaxPool ride in the convolution's epilogue (pool2 = 2)
        fuse_pool = _FUSE_POOL2 and not ctx.needs_input_grad[0]
        skip_pool = False
        # BASELINE.json config 5 (ops.set_vgg_fp8): layers of the fp8 chain take e4m3 operands on the fp8 MFMA.  x8 = the e4m3 copy
        # of x (slot = index of its scale), produced by the previous layer's epilogue or by sp_quantize_fp8 where the producer is
        # a bf16 kernel; bf16 outputs are still written wherever the pyramid taps / the backward pass need them.
        f8 = packs.get("f8")
        if f8 is not None and ctx.needs_input_grad[0]:
            f8 = None                                        # the pass with gradient stays 16-bit (ops.set_vgg_fp8)
        calibrating = f8 is not None and not f8["calibrated"]
        x8 = None

        def in_chain(conv_idx, hh, ww):                      # the fp8 kernel's shapes (include/sempyr.h, SP_F8)
            return f8 is not None and conv_idx in f8["w"] and hh % 8 == 0 and ww % 32 == 0

        def to_fp8(t, slot):
            if calibrating:                                  # first call: bf16 pass; record max|x| of every chain input once
                f8["amax"][slot:slot + 1].copy_(t.detach().float().abs().amax().reshape(1))
                return None
            return ops.quantize_fp8(t, f8["inv"][slot:slot + 1], f8["amax"][slot:slot + 1])
        for k, v in enumerate(_VGG_CFG):
            if v == "M" and skip_pool:
                skip_pool = False
                h, w = h // 2, w // 2
                feats.append(x)
                continue
            if v == "M":
                y = ops.nhwc_empty(n, x.shape[1], h // 2, w // 2, dtype, dev)
                Lb.call("sp_maxpool2_fwd", ops.ptr(x), ops.ptr(y), n, h, w, x.shape[1], 0, ops.sp_dtype(dtype), ops.stream())
                trace.append(("pool", len(acts) - 1))
                h, w = h // 2, w // 2
                feats.append(y)
                x8 = None                                                # pooled by the bf16 kernel: the next fp8 layer quantises it
            elif in_chain(ci, h, w):
                pk = packs["conv"][ci]
                slot = ci                                                # one scale slot per convolution input
                nxt_is_pool = k + 1 < len(_VGG_CFG) and _VGG_CFG[k + 1] == "M"
                nh, nw = (h // 2, w // 2) if nxt_is_pool else (h, w)
                nslot = ci + 1 if in_chain(ci + 1, nh, nw) else None    # an fp8 layer behind this one (possibly across the pool)?
                if x8 is None:
                    x8 = to_fp8(x, slot)
                ci += 1
                if calibrating:                                          # plain bf16 layer this once (the recorded maxima set the scales)
                    y = ops.nhwc_empty(n, v, h, w, dtype, dev)
                    ops.conv_launch(x, pk["fwd"].data_ptr(), pk["bias"], y, None, None, None, 0.0, n, h, w, x.shape[1], v, v, 3, ACT_RELU, dtype, k_real=pk["cin"])
                    x8 = None
                    trace.append(("conv", len(acts) - 1, pk))
                else:
                    w8, w_scale, cin_p8 = f8["w"][ci - 1]
                    pool_here = fuse_pool and nxt_is_pool
                    oh, ow = (h // 2, w // 2) if pool_here else (h, w)
                    want_next8 = nslot is not None and (pool_here or not nxt_is_pool)
                    need_bf16 = ctx.needs_input_grad[0] or nxt_is_pool or nslot is None
                    y = ops.nhwc_empty(n, v, oh, ow, dtype, dev) if need_bf16 else None
                    y8 = torch.empty((n, oh, ow, v), dtype=torch.uint8, device=dev).permute(0, 3, 1, 2) if want_next8 else None
                    ops.conv_launch_f8(x8, w8, w_scale, f8["scale"][slot:slot + 1], pk["bias"], y, y8,
                                       f8["inv"][nslot:nslot + 1] if want_next8 else None, f8["amax"][nslot:nslot + 1] if want_next8 else None,
                                       n, h, w, cin_p8, v, ACT_RELU, 2 if pool_here else 0)
                    x8 = y8
                    if pool_here:
                        skip_pool = True
                    trace.append(("conv", len(acts) - 1, pk))
                    if y is None:                                        # only the e4m3 copy exists: nothing downstream reads bf16
                        x = None
                        acts.append(None)
                        continue
            else:
                pk = packs["conv"][ci]
                ci += 1
                x8 = None
                if (ctx.needs_input_grad[0] and _VGG_POOL_IDX and _FUSE_POOL2 and k + 1 < len(_VGG_CFG) and _VGG_CFG[k + 1] == "M"
                        and ops.conv_pool_idx_ok(n, h, w, x.shape[1], v, dtype)):
                    # pass WITH gradient (the fake images, model_wrapper.py:179; both groups of a two-group pass), last convolution of a
                    # stage: ReLU + MaxPool ride in the epilogue and the window position of every maximum is recorded (2 bits per pooled
                    # element, sp_conv_params.pool_idx) - the unpooled tensor (168 MB at 64 x 256 x 256, batch 20) is neither written,
                    # nor read by a pooling pass, nor read again by the pooling's backward (sp_maxpool2_bwd_idx)
                    y = ops.nhwc_empty(n, v, h // 2, w // 2, dtype, dev)
                    pidx = torch.empty(n * (h // 2) * (w // 2) * (v // 16), dtype=torch.int32, device=dev)
                    ops.conv_launch(x, pk["fwd"].data_ptr(), pk["bias"], y, None, None, None, 0.0, n, h, w, x.shape[1], v, v, 3, ACT_RELU,
                                    dtype, pool2=2, k_real=pk["cin"], pool_idx=pidx)
                    trace.append(("conv", len(acts) - 1, pk))
                    acts.append(None)                                    # (the unpooled output does not exist)
                    trace.append(("poolidx", len(acts), len(pool_idx), h, w))
                    pool_idx.append(pidx)
                    skip_pool = True
                    x = y
                    acts.append(x)
                    continue
                if fuse_pool and k + 1 < len(_VGG_CFG) and _VGG_CFG[k + 1] == "M" and ops.conv_pool2_ok(h, w, v, 3):
                    y = ops.nhwc_empty(n, v, h // 2, w // 2, dtype, dev)
                    ops.conv_launch(x, pk["fwd"].data_ptr(), pk["bias"], y, None, None, None, 0.0, n, h, w, x.shape[1], v, v, 3, ACT_RELU,
                                    dtype, pool2=2, k_real=pk["cin"])
                    skip_pool = True
                elif (n > n_grad and _FUSE_POOL2 and v <= 128 and k + 1 < len(_VGG_CFG) and _VGG_CFG[k + 1] == "M"
                      and ops.conv_pool2_ok(h, w, v, 3)):
                    # two-group pass, last convolution of the first two stages (64 / 128 channels at 256^2 / 128^2: whole rounds of work
                    # items at either batch size, so one launch over 2B images gains nothing there): the group WITH gradient keeps its
                    # unpooled output for the backward pass, the group without takes ReLU + MaxPool in the convolution's epilogue -
                    # its full-resolution tensor (168 / 84 MB) is neither written nor read back by a pooling pass
                    yf = ops.nhwc_empty(n_grad, v, h, w, dtype, dev)
                    ops.conv_launch(x[:n_grad], pk["fwd"].data_ptr(), pk["bias"], yf, None, None, None, 0.0, n_grad, h, w, x.shape[1], v, v, 3,
                                    ACT_RELU, dtype, k_real=pk["cin"])
                    trace.append(("conv", len(acts) - 1, pk))
                    acts.append(yf)
                    y = ops.nhwc_empty(n, v, h // 2, w // 2, dtype, dev)
                    Lb.call("sp_maxpool2_fwd", ops.ptr(yf), ops.ptr(y), n_grad, h, w, v, 0, ops.sp_dtype(dtype), ops.stream())
                    trace.append(("pool", len(acts) - 1))
                    ops.conv_launch(x[n_grad:], pk["fwd"].data_ptr(), pk["bias"], y[n_grad:], None, None, None, 0.0, n - n_grad, h, w, x.shape[1],
                                    v, v, 3, ACT_RELU, dtype, pool2=2, k_real=pk["cin"])
                    skip_pool = True
                    x = y
                    acts.append(x)
                    continue
                else:
                    y = ops.nhwc_empty(n, v, h, w, dtype, dev)
                    ops.conv_launch(x, pk["fwd"].data_ptr(), pk["bias"], y, None, None, None, 0.0, n, h, w, x.shape[1], v, v, 3, ACT_RELU, dtype, k_real=pk["cin"])
                trace.append(("conv", len(acts) - 1, pk))
            x = y
            acts.append(x)
        p7 = ops.nhwc_empty(n, 512, 7, 7, dtype, dev)
        Lb.call("sp_adaptive_avgpool_fwd", ops.ptr(x), ops.ptr(p7), n, h, w, 512, 7, 7, ACT_NONE, ops.sp_dtype(dtype), ops.stream())
        flat = p7.permute(0, 2, 3, 1).reshape(n, 7 * 7 * 512)        # NHWC order; FC1 columns were permuted at pack time
        fcs = packs["fc"]

        def lin(inp, pk, nout, act):
            out = torch.empty((n, nout), dtype=dtype, device=dev)
            step = n if (n <= 64 and CFG.vgg_fc_joint) else n_grad   # (the split-K MFMA form takes up to 64 rows: both groups of a two-batch
            for lo in range(0, n, step):                         # pass in one launch - the weights are streamed once; beyond: per group)
                rows = min(step, n - lo)
                ops.linear_launch(inp[lo:lo + rows], pk["fwd"].data_ptr(), pk["kp"], pk["bias"], None, out[lo:lo + rows], rows, inp.shape[1],
                                  nout, act)
            return out
        h1 = lin(flat, fcs[0], 4096, ACT_RELU)
        # tap 5 is POST-ReLU: torchvision's classifier[4] is ReLU(inplace=True) and overwrites the tensor appended at
        # models.py:212-213 (pinned by tests/golden step_cf1_b2_seed0, see oracle/sempyr_oracle.py vgg16_forward)
        h2 = lin(h1, fcs[1], 4096, ACT_RELU)
        logits = lin(h2, fcs[2], fcs[2]["n"], ACT_NONE)
        feats += [h2, logits]
        if f8 is not None:                                   # delayed scaling: this pass's maxima set the next pass's scales
            Lb.call("sp_fp8_update_scales", ops.ptr(f8["amax"]), ops.ptr(f8["scale"]), ops.ptr(f8["inv"]), f8["amax"].numel(),
                    float(f8["margin"]), ops.stream())
            f8["calibrated"] = True
        ctx.trace, ctx.packs, ctx.dtype, ctx.hw_last = trace, packs, dtype, (h, w)
        ctx.img_meta = (img.shape, img.dtype)
        ctx.save_for_backward(*acts, p7, h1, h2, *pool_idx)
        ctx.n_acts = len(acts)
        ctx.n_grad = n_grad
        ops.PROBE_NET[0] = "sn"
        if img_b is None:
            return tuple(feats)
        second = tuple(f[n_grad:] for f in feats)                # batch-outermost layouts: both halves are dense
        ctx.mark_non_differentiable(*second)
        return tuple(f[:n_grad] for f in feats) + second

    @staticmethod
    def backward(ctx, *dfeats):
        Lb = ops.L
        ops.PROBE_NET[0] = "vgg"
        saved = ctx.saved_tensors
        acts = saved[:ctx.n_acts]
        p7, h1, h2 = saved[ctx.n_acts:ctx.n_acts + 3]
        pool_idx = saved[ctx.n_acts + 3:]
        dtype, packs = ctx.dtype, ctx.packs
        dev = h1.device
        n = ctx.n_grad                    # (two-group pass: the first n images of every saved tensor)
        dfeats = dfeats[:7]
        h1, h2 = h1[:n], h2[:n]
        fcs = packs["fc"]
        sd = ops.sp_dtype(dtype)

        def lin_dgrad(dz, pk, k):
            out = torch.empty((n, k), dtype=dtype, device=dev)
            ops.linear_launch(dz, pk["dgrad"].data_ptr(), pk["np"], None, None, out, n, dz.shape[1], k, ACT_NONE)
            return out
        d_h2 = None
        if dfeats[6] is not None:
            d_h2 = lin_dgrad(ops.as_rows(dfeats[6], dtype), fcs[2], 4096)
        if dfeats[5] is not None:
            g5 = ops.as_rows(dfeats[5], dtype)
            d_h2 = g5 if d_h2 is None else d_h2 + g5
        g = None
        if d_h2 is not None:
            dz2 = ops.act_backward(d_h2, h2, ACT_RELU)
            dz1 = ops.act_backward(lin_dgrad(dz2, fcs[1], 4096), h1, ACT_RELU)
            dflat = lin_dgrad(dz1, fcs[0], 7 * 7 * 512)
            dp7 = dflat.reshape(n, 7, 7, 512).permute(0, 3, 1, 2)
            h, w = ctx.hw_last
            g = ops.nhwc_empty(n, 512, h, w, dtype, dev)
            Lb.call("sp_adaptive_avgpool_bwd", ops.ptr(dp7), None, ops.ptr(g), n, h, w, 512, 7, 7, ACT_NONE, sd, ops.stream())
        tap = 4
        folded = False              # the tap gradient of the coming pool step already rode in the dgrad launch that produced g (res1)
        for step in reversed(ctx.trace):
            if step[0] == "poolidx":
                _, yi, slot, h, w = step
                ypool = acts[yi]
                c = ypool.shape[1]
                gt = dfeats[tap] if not folded else None
                folded = False
                tap -= 1
                if gt is not None:
                    gt = ops.as_nhwc(gt, dtype)
                    g = gt if g is None else g + gt
                if g is None:
                    continue
                dx = ops.nhwc_empty(n, c, h, w, dtype, dev)
                Lb.call("sp_maxpool2_bwd_idx", ops.ptr(g), ops.ptr(ypool), ops.ptr(pool_idx[slot]), ops.ptr(dx), n, h, w, c, sd, ops.stream())
                g = dx
            elif step[0] == "pool":
                xin = acts[step[1]]
                _, c, h, w = xin.shape
                gt = dfeats[tap] if not folded else None
                folded = False
                tap -= 1
                if gt is not None:
                    gt = ops.as_nhwc(gt, dtype)
                    g = gt if g is None else g + gt
                if g is None:
                    continue
                dx = ops.nhwc_empty(n, c, h, w, dtype, dev)
                # relu=1: the pooled tensor is post-ReLU; zero where the window maximum is not positive
                Lb.call("sp_maxpool2_bwd", ops.ptr(g), ops.ptr(xin), ops.ptr(dx), n, h, w, c, 1, sd, ops.stream())
                g = dx
            else:
                if g is None:
                    continue
                _, idx, pk = step
                xin = acts[idx]
                _, cin_p, h, w = xin.shape
                first = idx == 0
                dx = (ops.nhwc_zeros if first and pk["cin"] != cin_p else ops.nhwc_empty)(n, cin_p, h, w, dtype, dev)
                # the producer of this conv's input is a ReLU (another conv) unless it is a pool output (already masked
                # by the pool backward) or the image: fold that ReLU's derivative in via mask_src = the input itself
                producer_is_conv = (not first) and ctx.trace[idx - 1][0] == "conv"
                # the input of this convolution is a pyramid tap (a pool output): the tap's own gradient (reconstruction loss) joins the
                # chain in this launch's epilogue (res1) instead of a separate elementwise addition
                res1 = None
                if (not first) and ctx.trace[idx - 1][0] in ("pool", "poolidx") and dfeats[tap] is not None and pk["cin"] == cin_p:
                    res1 = ops.as_nhwc(dfeats[tap], dtype)
                    folded = True
                ops.conv_launch(g, pk["dgrad"].data_ptr(), None, dx, res1, None, xin if producer_is_conv else None, 0.0, n, h, w,
                                g.shape[1], pk["cin"], cin_p, 3, ACT_NONE, dtype, family="dgrad")
                g = dx
        ops.PROBE_NET[0] = "sn"
        if g is None:
            return None, None, None, None
        shape, src_dtype = ctx.img_meta
        dimg = ops.nhwc_empty(n, 3, shape[2], shape[3], dtype, dev)
        import ctypes
        sc = (ctypes.c_float * 3)(*[1.0 / s for s in _IMAGENET_STD])
        Lb.call("sp_ingest_image_bwd", ops.ptr(g), g.shape[1], ops.ptr(dimg), 3, n * shape[2] * shape[3], ctypes.cast(sc, ctypes.c_void_p),
                sd, ops.stream())
        if dimg.dtype != src_dtype:
            dimg = dimg.to(src_dtype)
        return dimg, None, None, None


class VGG16(nn.Module):
    """models.py:158-216: 7-level feature pyramid of a (frozen, eval-mode) VGG-16 fine-tuned on Places365."""

    def __init__(self, path_to_pre_trained_model: Optional[str] = None, return_output: Optional[bool] = False) -> None:
        super().__init__()
        self.return_output = return_output
        if path_to_pre_trained_model is not None:
            # the reference unpickles a whole torchvision model here (models.py:174); its parameters are copied
            loaded = torch.load(path_to_pre_trained_model, weights_only=False)
            self.vgg16 = _VGG16Topology(num_classes=loaded.classifier[-1].out_features)
            self.vgg16.load_state_dict(loaded.state_dict())
        else:
            self.vgg16 = _VGG16Topology(num_classes=365)
        self._packs = None
        self._pack_key = None

    def _packed(self, dtype, device):
        params = list(self.vgg16.parameters())
        key = (dtype, str(device), ops.vgg_fp8()) + tuple((p.data_ptr(), p._version) for p in params)
        if key == self._pack_key:
            return self._packs
        e = ops.chunk_elems(dtype)
        packs = {"conv": [], "fc": []}
        sd = ops.sp_dtype(dtype)
        with torch.no_grad():
            for m in self.vgg16.features:
                if not isinstance(m, nn.Conv2d):
                    continue
                w = m.weight.detach().to(device=device, dtype=torch.float32).contiguous()
                o, i = w.shape[0], w.shape[1]
                cin_p, cout_p = ops.pad_to(i, e), ops.pad_to(o, e)
                fwd = torch.empty(o * 9 * cin_p, dtype=dtype, device=device)
                dg = torch.empty(i * 9 * cout_p, dtype=dtype, device=device)
                ops.L.call("sp_pack_weight", ops.ptr(w), o, i * 9, i, 9, cin_p, cout_p, 0, 0, ops.ptr(fwd), ops.ptr(dg), sd, ops.stream())
                packs["conv"].append({"fwd": fwd, "dgrad": dg, "cin": i, "cin_p": cin_p, "cout_p": cout_p,
                                      "bias": m.bias.detach().to(device=device, dtype=torch.float32).contiguous()})
            for j, m in enumerate(x for x in self.vgg16.classifier if isinstance(x, nn.Linear)):
                w = m.weight.detach().to(device=device, dtype=torch.float32).contiguous()
                o, k = w.shape
                kp, np_ = ops.pad_to(k, 8), ops.pad_to(o, 8)
                fwd = torch.empty(o * kp, dtype=dtype, device=device)
                dg = torch.empty(k * np_, dtype=dtype, device=device)
                chw = (512, 49) if j == 0 else (0, 0)     # FC1 consumes the NHWC-flattened 7x7x512 pool output
                ops.L.call("sp_pack_weight", ops.ptr(w), o, k, k, 1, kp, np_, chw[0], chw[1], ops.ptr(fwd), ops.ptr(dg), sd, ops.stream())
                packs["fc"].append({"fwd": fwd, "dgrad": dg, "kp": kp, "np": np_, "n": o,
                                    "bias": m.bias.detach().to(device=device, dtype=torch.float32).contiguous()})
            if ops.vgg_fp8() > 0 and ops.is_16bit(dtype):
                # BASELINE.json config 5: e4m3 filters (one scale per output channel) for every 3x3 layer the fp8 kernel can take
                # (Cout > 64, Cin a multiple of 16); activation scales start uncalibrated (the first forward runs in bf16 and records them)
                f8w, ci = {}, 0
                for m in self.vgg16.features:
                    if not isinstance(m, nn.Conv2d):
                        continue
                    if m.out_channels > 64 and m.in_channels % 16 == 0:
                        f8w[ci] = ops.pack_weight_fp8(m.weight.detach().to(device=device, dtype=torch.float32))
                    ci += 1
                packs["f8"] = {"w": f8w, "scale": torch.ones(ci + 1, dtype=torch.float32, device=device),
                               "inv": torch.ones(ci + 1, dtype=torch.float32, device=device),
                               "amax": torch.zeros(ci + 1, dtype=torch.float32, device=device), "calibrated": False, "margin": 1.5,
                               "mode": ops.vgg_fp8()}
        self._packs, self._pack_key = packs, key
        return packs

    def forward(self, input: torch.Tensor) -> List[torch.Tensor]:
        if self.training:
            raise ops.L.SempyrError("VGG16 is used frozen in eval mode (model_wrapper.py:114); call .eval()")
        ops.require_gpu(input)
        if input.shape[1] == 1:
            input = input.repeat_interleave(3, dim=1)
        dt = ops.compute_dtype()
        feats = _VGGPyramidFn.apply(input, self._packed(dt, input.device), dt)
        if self.return_output:
            return feats[-1]
        return list(feats)

    def forward_pair(self, input: torch.Tensor, input_no_grad: torch.Tensor):
        """The pyramid of two batches in ONE pass: `input` (may need its image gradient - the generator step's fake images,
        model_wrapper.py:179) and `input_no_grad` (a batch of real images, model_wrapper.py:141).  Returns (features of input, features of
        input_no_grad), each as forward() would - the network is frozen and in eval mode, so the two results do not depend on each
        other; only the tile counts of the deep stages do (80 -> 160 work items on 256 CUs at batch 20)."""
        if self.training:
            raise ops.L.SempyrError("VGG16 is used frozen in eval mode (model_wrapper.py:114); call .eval()")
        ops.require_gpu(input)
        if input.shape[1:] != input_no_grad.shape[1:] or input.shape[1] != 3:
            raise ops.L.SempyrError("VGG16.forward_pair: two RGB batches of one image size are expected")
        if ops.vgg_fp8() > 0:
            raise ops.L.SempyrError("VGG16.forward_pair: the fp8 slice (ops.set_vgg_fp8) runs in the no-gradient pass only")
        dt = ops.compute_dtype()
        feats = _VGGPyramidFn.apply(input, self._packed(dt, input.device), dt, input_no_grad.detach())
        return list(feats[:7]), [f.detach() for f in feats[7:]]
