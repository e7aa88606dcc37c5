"""ModelWrapper with the reference's constructor and ``train()`` signature (/root/reference/model_wrapper.py).

The loop body (model_wrapper.py:131-190) is ``train_step``: one discriminator step then one generator step.
What differs from the reference, without changing any observable result:
  * the discriminator's weight gradients of the GENERATOR step are not computed - the reference computes
    them and zeroes them unused at the top of the next iteration (model_wrapper.py:136-137,174,188);
  * gradients w.r.t. the real images, the masks and the latent vector (all dead) are not computed;
  * the five loss scalars are fetched with ONE device->host transfer per iteration instead of ten ``.item()``
    calls (model_wrapper.py:192-202);
  * multi-GPU is one process per GPU with an RCCL all-reduce of the gradients (``distributed.GradientReducer``)
    instead of single-process nn.DataParallel (main.py:91-94).
Validation (FID with a downloaded Inception-v3) and the sample-grid plots are outside the hot path
(SURVEY.md section 2, rows 8-9): ``validate`` returns nan; ``inference`` writes the reference's 7 x 7 sample grid as a PNG
(misc.save_image_grid - torchvision is not a dependency).
"""
from __future__ import annotations

import os
from datetime import datetime
from typing import Dict, Optional, Union

import torch
import torch.nn as nn

from . import misc, ops, profiling
from .config import CFG
from .lossfunction import DiversityLoss, LSGANDiscriminatorLoss, LSGANGeneratorLoss, SemanticReconstructionLoss
from .models import VGG16, Discriminator, Generator, _class_index


def _unwrap(module):
    return module.module if hasattr(module, "module") and isinstance(module.module, nn.Module) else module


class ModelWrapper(object):
    def __init__(self,
                 generator: Union[Generator, nn.DataParallel],
                 discriminator: Union[Discriminator, nn.DataParallel],
                 training_dataset,
                 validation_dataset,
                 vgg16: Union[VGG16, nn.DataParallel] = None,
                 generator_optimizer: torch.optim.Optimizer = None,
                 discriminator_optimizer: torch.optim.Optimizer = None,
                 generator_loss: nn.Module = None,
                 discriminator_loss: nn.Module = None,
                 semantic_reconstruction_loss: nn.Module = None,
                 diversity_loss: nn.Module = None,
                 save_data_path: Optional[str] = 'saved_data',
                 gradient_reducer=None) -> None:
        # nn.DataParallel wrappers are unwrapped: data parallelism is one process per GPU here
        self.generator = _unwrap(generator)
        self.discriminator = _unwrap(discriminator)
        self.training_dataset = training_dataset
        self.validation_dataset_fid = validation_dataset
        self.vgg16 = _unwrap(vgg16) if vgg16 is not None else VGG16()
        self.generator_optimizer = generator_optimizer
        self.discriminator_optimizer = discriminator_optimizer
        self.generator_loss = generator_loss if generator_loss is not None else LSGANGeneratorLoss()
        self.discriminator_loss = discriminator_loss if discriminator_loss is not None else LSGANDiscriminatorLoss()
        self.semantic_reconstruction_loss = (semantic_reconstruction_loss if semantic_reconstruction_loss is not None
                                             else SemanticReconstructionLoss())
        self.diversity_loss = diversity_loss if diversity_loss is not None else DiversityLoss()
        self.latent_dimensions = self.generator.latent_dimensions
        self.gradient_reducer = gradient_reducer
        for parameter in self.vgg16.parameters():            # model_wrapper.py:67-68
            parameter.requires_grad = False
        self.logger = misc.Logger()
        self.path_save_models = self.path_save_plots = self.path_save_metrics = None
        if save_data_path is not None:
            stamp = str(datetime.now())
            self.path_save_models = os.path.join(save_data_path, 'models_' + stamp)
            self.path_save_plots = os.path.join(save_data_path, 'plots_' + stamp)
            self.path_save_metrics = os.path.join(save_data_path, 'metrics_' + stamp)
            for p in (self.path_save_models, self.path_save_plots, self.path_save_metrics):
                os.makedirs(p, exist_ok=True)
        for name in ('generator', 'discriminator', 'vgg16', 'generator_optimizer', 'discriminator_optimizer',
                     'generator_loss', 'discriminator_loss', 'diversity_loss', 'semantic_reconstruction_loss'):
            self.logger.hyperparameter[name] = str(getattr(self, name))
        self._d_params = [p for p in self.discriminator.parameters()]
        self._g_params = [p for p in self.generator.parameters()]
        # every gradient of a network lives in ONE flat fp32 buffer (ops.SpectralNormBank.flat; param.grad are views of it):
        # D runs twice per backward (real, fake) and its gradients meet there instead of in 59 autograd sums, and the
        # data-parallel reducer all-reduces ranges of the buffer in place.  SP_DIRECT_GRADS=0: plain autograd gradients.
        self._banks = {}
        if CFG.direct_grads:
            for key, net, passes in (("d", self.discriminator, 2), ("g", self.generator, 1)):
                bank = getattr(net, "_bank", None)
                if bank is not None:
                    bank.direct_grads = True
                    bank.expected_passes = passes
                    bank.set_groups(4 if gradient_reducer is not None else 1)
                    self._banks[key] = bank
        self._graph_state = None
        self.graph_after_iterations = CFG.graph_after       # train(): eager iterations before the step is captured (0 = never)
        self._eager_run, self._eager_sig, self._graph_failed = 0, None, False
        # (real images tensor, its ._version, its VGG pyramid) where the previous iteration computed it ahead (train_step: next_images_real)
        self._vgg_ahead = None
        # (fake images of the generator step with their autograd graph, their latents) where the discriminator phase has already taken
        # that forward together with its own (_d_phase: Generator.forward_pair)
        self._fake_ahead = None
        self._capturing = False
        self._fired = set()
        self.iterations = 0

    # ------------------------------------------------------------------------------------------
    # data parallelism: flat, in-place, overlapped gradient reduction (distributed.GradientReducer)
    # ------------------------------------------------------------------------------------------
    def _arm_reducer(self, key: str) -> None:
        """Before a backward pass launched eagerly: the bank's layer groups hand their flat range to the reducer the moment
        their batched spectral-norm backward has been enqueued - while the rest of the backward pass still runs."""
        bank = self._banks.get(key)
        if bank is None:
            return
        bank.on_group_done = None
        self._fired = set()
        # (fp16 mode: the loss scale comes off the whole flat buffer after the backward pass - ranges go to the wire only then)
        if self.gradient_reducer is not None and not self._capturing and self.gradient_reducer.active() and ops.loss_scale() == 1.0:
            red = self.gradient_reducer

            def done(start, stop, bank=bank):
                self._fired.add((start, stop))
                bucket = max(1, red.bucket_bytes // 4)
                for a in range(start, stop, bucket):
                    red.reduce_range(bank.flat, a, min(stop, a + bucket))
            bank.on_group_done = done

    def _finish_backward(self, key: str) -> None:
        """After .backward(): non-SN gradients move into the flat buffer (inside the captured graph too)."""
        bank = self._banks.get(key)
        if bank is not None:
            bank.on_group_done = None
            bank.collect_extra()
            if bank.flat is not None:
                # fp16 mode: the loss scale has come off the spectral-normalised layers' gradients inside their batched
                # backward; the tail of the buffer (gradients that arrived through autograd) loses it here (16-byte aligned offset)
                ops.unscale_(bank.flat, bank.sn_floats)
        elif ops.loss_scale() != 1.0:
            raise ops.L.SempyrError("the fp16 storage mode needs the flat gradient buffers (config.CFG.direct_grads) for its loss scale")

    def _start_reduce(self, key: str, params, eager: bool) -> None:
        """Enqueues (side stream) whatever of the network's gradients has not been handed over by the group hooks."""
        red = self.gradient_reducer
        if red is None or not red.active():
            return
        bank = self._banks.get(key)
        if bank is None or bank.flat is None:
            red.reduce(params)                        # loose gradients: flatten / all-reduce / scatter back
            return
        fired = self._fired if eager else set()
        todo = []
        # replayed graphs: the whole buffer is final here - no collective below 8 MB (SpectralNormBank.flat_ranges); eager launches: the
        # groups have handed their own ranges over from inside the backward pass, what is left is matched against them group by group
        for a, b in bank.flat_ranges(max(1, red.bucket_bytes // 4), **({"min_floats": 0} if eager else {})):
            if not any(fa <= a and b <= fb for fa, fb in fired):
                todo.append((a, b))
        red.reduce_flat(bank.flat, todo)
        # safety net: a gradient that does not live in the flat buffer (a parameter the bank does not know) is reduced on its own
        lo = bank.flat.data_ptr()
        hi = lo + 4 * bank.flat.numel()
        loose = [p for p in params if p.grad is not None and not (lo <= p.grad.data_ptr() < hi)]
        if loose:
            red.reduce(loose)

    def _join_reduce(self, tag: str = "") -> None:
        if self.gradient_reducer is not None:
            self.gradient_reducer.join(tag)

    def _optimizer_step(self, key: str, optimizer) -> None:
        """optimizer.step() - in the fp16 storage mode behind the overflow guard of the dynamic loss scale (ops.LossScaler): the
        network's flat gradient buffer (already averaged over the ranks, so every rank decides alike) is screened for inf / NaN, the
        step is skipped where one is found, and the scale is backed off / grown - all on the device for this package's Adam
        (sp_adam_multi_guarded); a foreign optimizer costs one host sync per step for the same decision."""
        bank = self._banks.get(key)
        sc = ops.loss_scaler(bank.flat.device) if bank is not None and bank.flat is not None else None
        if sc is None:
            optimizer.step()
            return
        from . import optim
        sc.check(bank.flat)
        if isinstance(optimizer, optim.Adam):
            optimizer.step(found_inf=sc.found_ptr)
        elif not sc.values()["found"]:
            optimizer.step()
        sc.update()

    # ------------------------------------------------------------------------------------------
    def _g_pair_ok(self) -> bool:
        """Both generator forwards of the iteration in one pass (config.CFG.g_pair): this package's generator in training mode.  (Under
        data parallelism the discriminator's gradient all-reduce then hides behind _g_features instead of the generator forward.)"""
        return CFG.g_pair and isinstance(self.generator, Generator) and self.generator.training

    def _d_phase(self, images_real, labels, labels_f, masks, noise_d, features_real=None, noise_g=None):
        """model_wrapper.py:136-160: forward passes and backward of the discriminator step (everything but Adam).  features_real: the
        pyramid of images_real where an earlier generator step has already computed it (_g_rest, next_images_real).  noise_g: the
        latents of the generator step, for the case that its forward is taken in the same pass as this step's (_g_pair_ok)."""
        G, D, V = self.generator, self.discriminator, self.vgg16
        G.zero_grad()
        D.zero_grad()
        self._fake_ahead = None
        with torch.no_grad():
            if features_real is None:
                features_real = V(images_real)
            if noise_d is None:
                noise_d = torch.randn((images_real.shape[0], self.latent_dimensions), dtype=torch.float32, device=images_real.device)
        if self._g_pair_ok():
            if noise_g is None:                       # (the reference draws it later, model_wrapper.py:168 - nothing else draws in between)
                noise_g = torch.randn((images_real.shape[0], self.latent_dimensions), dtype=torch.float32, device=images_real.device)
            images_fake_g, images_fake = G.forward_pair(noise_g, noise_d, features_real, masks, labels_f)
            self._fake_ahead = (images_fake_g, noise_g)
        else:
          with torch.no_grad():
            if hasattr(G, "map_mode"):
                G.map_mode = "stash"                  # this forward and the G step's see the same pyramid, masks and weights
            images_fake = G(input=noise_d, features=features_real, masks=masks, class_id=labels_f)
        if CFG.d_pair and hasattr(D, "forward_pair") and D.training and images_fake.shape == images_real.shape:
            prediction_real, prediction_fake = D.forward_pair(images_real, images_fake, labels)     # one trunk pass over 2B images
        else:
            prediction_real = D(images_real, labels)
            prediction_fake = D(images_fake, labels)
        loss_d_real, loss_d_fake = self.discriminator_loss(prediction_real, prediction_fake)
        self._arm_reducer("d")
        # d(real + fake): one seed per loss instead of a sum kernel and autograd's own seed (model_wrapper.py:158-160)
        torch.autograd.backward([loss_d_real, loss_d_fake], ops.backward_seeds([loss_d_real, loss_d_fake]))
        self._finish_backward("d")
        return features_real, loss_d_real, loss_d_fake

    def _g_forward(self, images_real, labels_f, masks, features_real, noise_g):
        """model_wrapper.py:165-172: the generator forward of the G step.  It does not read the discriminator, so D's gradient
        all-reduce and D's Adam step may still be in flight / pending while it runs (train_step)."""
        G = self.generator
        G.zero_grad()
        ahead, self._fake_ahead = self._fake_ahead, None
        if ahead is not None:                         # taken in the discriminator phase's pass (Generator.forward_pair)
            return ahead
        if noise_g is None:
            noise_g = torch.randn((images_real.shape[0], self.latent_dimensions), dtype=torch.float32, device=images_real.device)
        if hasattr(G, "map_mode"):
            G.map_mode = "reuse"                      # the masked-feature mappings come from the D step's forward (other sigma only)
        return G(input=noise_g, features=features_real, masks=masks, class_id=labels_f), noise_g

    def _vgg_pair_ok(self, images_fake, next_images_real) -> bool:
        """One VGG pass over [fake | next real] pays where a batch alone leaves CUs idle in the pyramid's deep stages (batch 20 at
        256 x 256: 80 work items on 256 CUs in the 16 x 16 stage, 320 in the 32 x 32 stage): same box, bf16 15.73 -> 15.58 ms, fp32
        103.5 -> 102.2 ms per step; at batch 32 the stages are full already and the pass costs 0.7 % (the second half cannot pool in
        the convolution epilogue) - so: up to 24 images of 256 x 256 (n x (H / 64) x (W / 64) <= 384)."""
        if not (CFG.vgg_pair and next_images_real is not None and hasattr(self.vgg16, "forward_pair") and ops.vgg_fp8() == 0
                and next_images_real.is_cuda and tuple(next_images_real.shape) == tuple(images_fake.shape)):
            return False
        n, _, h, w = images_fake.shape
        return n * (h // 64) * (w // 64) <= 384

    def _g_features(self, images_fake, noise_g, w_div, next_images_real=None):
        """The part of the generator step that does NOT read the discriminator (model_wrapper.py:179-186): the diversity loss and the
        frozen VGG-16's pass over the fake images.  train_step runs it BEFORE the discriminator's optimizer step: under data parallelism
        this pass (1.7 ms at batch 20) - together with the generator forward, where that has not already ridden in the discriminator
        phase (_g_pair_ok) - is what the discriminator's gradient all-reduce hides behind.  next_images_real: the real images of the NEXT
        iteration, whose pyramid (model_wrapper.py:141 of that iteration; the network is frozen) is taken in the same pass
        (VGG16.forward_pair).  Returns (diversity loss, features of the fake images, features of the next real images or None)."""
        # the weights ride inside this package's loss kernels; a caller's own loss module is weighted the reference's way
        if isinstance(self.diversity_loss, DiversityLoss):
            loss_div = self.diversity_loss(images_fake, noise_g, weight=w_div)
        else:
            loss_div = w_div * self.diversity_loss(images_fake, noise_g)
        features_next = None
        if self._vgg_pair_ok(images_fake, next_images_real):
            features_fake, features_next = self.vgg16.forward_pair(images_fake, next_images_real)
        else:
            features_fake = self.vgg16(images_fake)
        return loss_div, features_fake, features_next

    def _g_rest(self, images_fake, noise_g, labels, masks, features_real, w_rec, w_div, next_images_real=None, features_next_out=None,
                ahead=None):
        """model_wrapper.py:174-188: D(fake), the generator and reconstruction losses, backward (everything but Adam).  ahead: what
        _g_features has already computed for these fake images (train_step: in front of the discriminator's optimizer step); without
        it that part runs here.  features_next_out: static tensors to copy the next batch's pyramid into after the backward pass
        (captured graphs); it is kept in self._vgg_ahead."""
        D = self.discriminator
        D.zero_grad()
        if ahead is None:
            ahead = self._g_features(images_fake, noise_g, w_div, next_images_real)
        loss_div, features_fake, features_next = ahead
        for p in self._d_params:                               # dead D weight gradients are skipped
            p.requires_grad_(False)
        try:
            prediction_fake = D(images_fake, labels)
            loss_g = self.generator_loss(prediction_fake)
            if isinstance(self.semantic_reconstruction_loss, SemanticReconstructionLoss):
                loss_rec = self.semantic_reconstruction_loss(features_real, features_fake, masks, weight=w_rec)
            else:
                loss_rec = w_rec * self.semantic_reconstruction_loss(features_real, features_fake, masks)
            self._arm_reducer("g")
            # backward of (loss_g + loss_rec + loss_div) (model_wrapper.py:187-188): one seed per term
            terms = [loss_g, loss_rec, loss_div]
            torch.autograd.backward(terms, ops.backward_seeds(terms))
            self._finish_backward("g")
            self._vgg_ahead = None
            if features_next is not None:
                if features_next_out is not None:                # (after the backward pass: it still read this iteration's features)
                    with torch.no_grad():                        # one multi-tensor launch instead of seven copies
                        torch._foreach_copy_(list(features_next_out), list(features_next))
                    features_next = features_next_out
                self._vgg_ahead = (next_images_real, next_images_real._version, features_next, self._mode_key())
        finally:
            for p in self._d_params:
                p.requires_grad_(True)
        return loss_g, loss_rec, loss_div

    def _features_ahead(self, images_real):
        """The pyramid of images_real if the previous iteration computed it ahead (same tensor object, not written since)."""
        ahead, self._vgg_ahead = self._vgg_ahead, None
        if ahead is not None and ahead[0] is images_real and ahead[1] == images_real._version and ahead[3] == self._mode_key():
            return ahead[2]
        return None

    @staticmethod
    def _mode_key():
        """What a pyramid computed ahead depends on beside the images: the storage type and the fp8 slice (round-4 ADVICE)."""
        return (ops.compute_dtype(), ops.vgg_fp8())

    def train_step(self, images_real: torch.Tensor, labels: torch.Tensor, masks, w_rec: float = 0.1, w_div: float = 0.1,
                   noise_d: Optional[torch.Tensor] = None, noise_g: Optional[torch.Tensor] = None,
                   next_images_real: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """One iteration of model_wrapper.py:131-190 on tensors already on the device.  ``noise_d`` / ``noise_g``
        replace the two ``torch.randn`` draws (model_wrapper.py:147,168) for parity runs.  Returns the loss
        scalars as device tensors (no host sync).

        ``next_images_real`` (optional): the real images of the NEXT call (the same tensor object must then be passed as its
        ``images_real``).  The frozen VGG-16 sees them in the pass it makes over this iteration's fake images - one pass over 2B
        images instead of two over B (config.CFG.vgg_pair) - and the next call finds its pyramid computed.  Results do not change:
        the network is frozen and in eval mode.

        Order of work (results identical to the reference's order): D phase (with both generator forwards where _g_pair_ok); [D gradients
        -> side stream]; what the G phase can do without the discriminator - the generator forward if it is still due, the diversity
        loss, the VGG-16 pass over the fake images; join; Adam(D); rest of the G phase; [G gradients -> side stream]; join; Adam(G)."""
        # the one-hot labels become class indices ONCE per step (the reference's modules take the argmax in every forward,
        # models.py:151,501: five reductions and a float copy per step; our modules pass indices through)
        # - only for this package's modules: a caller's own module may take the argmax of what it is given (round-4 ADVICE)
        raw_labels = labels
        idx = _class_index(labels)
        labels = idx if isinstance(self.discriminator, Discriminator) else raw_labels
        labels_f = idx if isinstance(self.generator, Generator) else raw_labels
        with profiling.range("D phase"):
            features_real, loss_d_real, loss_d_fake = self._d_phase(images_real, labels, labels_f, masks, noise_d,
                                                                      self._features_ahead(images_real), noise_g)
            self._start_reduce("d", self._d_params, eager=True)
        with profiling.range("G forward"):
            images_fake, noise_g = self._g_forward(images_real, labels_f, masks, features_real, noise_g)
            ahead = self._g_features(images_fake, noise_g, w_div, next_images_real)
        with profiling.range("Adam(D)"):
            self._join_reduce("d")
            self._optimizer_step("d", self.discriminator_optimizer)
        with profiling.range("G rest"):
            loss_g, loss_rec, loss_div = self._g_rest(images_fake, noise_g, labels, masks, features_real, w_rec, w_div, next_images_real,
                                                      ahead=ahead)
            self._start_reduce("g", self._g_params, eager=True)
        with profiling.range("Adam(G)"):
            self._join_reduce("g")
            self._optimizer_step("g", self.generator_optimizer)
        self.iterations += 1
        return {"loss_discriminator_real": loss_d_real.detach(), "loss_discriminator_fake": loss_d_fake.detach(),
                "loss_generator": loss_g.detach(), "loss_generator_semantic_reconstruction": loss_rec.detach().reshape(()),
                "loss_generator_diversity": loss_div.detach(), "images_fake": images_fake.detach()}

    # ------------------------------------------------------------------------------------------
    def capture_graphs(self, images_real: torch.Tensor, labels: torch.Tensor, masks, w_rec: float = 0.1, w_div: float = 0.1) -> None:
        """Records train_step as THREE HIP graphs over static copies of the inputs - the D phase, the generator forward of the G
        phase, the rest of the G phase (~1000 kernel launches together); train_step_graphed() then costs three graph launches
        plus the two optimizer steps, so the step time no longer depends on how fast the host can enqueue.  The cut after the
        generator forward is where a multi-GPU run hides the discriminator's gradient all-reduce (distributed.py).  The shapes
        are static (fixed batch, 256x256); call after a few eager steps (lazy state: packed VGG weights, kernel
        attributes).  The two latent draws stay eager (two tiny launches into static buffers), so the device RNG is
        consumed exactly as in train_step()."""
        import gc
        for bank in self._banks.values():
            if bank.flat is None:
                raise RuntimeError("capture_graphs(): run at least one eager train_step() first (the flat gradient buffers and the "
                                   "packed VGG weights are allocated lazily and must not be born inside a graph's memory pool)")
        self.generator.zero_grad()
        self.discriminator.zero_grad()
        gc.collect()                       # no autograd nodes of earlier (eager-stream) iterations may survive into the capture
        torch.cuda.synchronize()
        st = self._graph_state = {}
        st["flat_ptrs"] = self._flat_ptrs()
        st["images"], st["labels"], st["masks"] = images_real.clone(), labels.clone(), [m.clone() for m in masks]
        st["w"] = (w_rec, w_div)
        st["mode"] = self._mode_key()
        zdim = (images_real.shape[0], self.latent_dimensions)
        st["noise_d"] = torch.zeros(zdim, dtype=torch.float32, device=images_real.device)
        st["noise_g"] = torch.zeros(zdim, dtype=torch.float32, device=images_real.device)
        # The pyramid of the real images is computed AHEAD (config.CFG.vgg_pair): the generator-step graph takes the next batch's real
        # images (st["images_next"]) through the VGG pass it makes over the fake images and leaves their features in st["feats_real"],
        # which the discriminator-step graph of the next replay reads.  Eager once here, so that the first replay finds them.
        st["feats_real"] = None
        if self._vgg_pair_ok(st["images"], st["images"]):
            with torch.no_grad():
                st["feats_real"] = [f.detach().clone() for f in self.vgg16(st["images"])]
            st["images_next"] = st["images"].clone()
            st["next_is_resident"] = True      # st["images_next"] holds what st["images"] holds (until a caller streams batches in)
            st["resident_ok"] = True           # st["feats_real"] is the pyramid of what st["images"] holds
            st["announced"] = None             # (tensor, version) whose pyramid st["feats_real"] holds, if a caller announced it
        handed_over = self._vgg_ahead      # the eager iteration in front of this capture announced its successor
        self._vgg_ahead = None
        # the zero-at-rest counters of the convolution kernels' K-split (ops._split_sync) are per stream: create the capture stream's
        # area BEFORE the capture, so that its zero fill is not a node of every replay
        if torch.cuda.graph.default_capture_stream is None:
            torch.cuda.graph.default_capture_stream = torch.cuda.Stream()
        with torch.cuda.stream(torch.cuda.graph.default_capture_stream):
            ops._split_sync(images_real.device)
        torch.cuda.synchronize()
        self._capturing = True
        try:
            gd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gd):
                cls = st["cls"] = _class_index(st["labels"])          # recomputed by every replay of this graph; the other two read it
                lab_d = cls if isinstance(self.discriminator, Discriminator) else st["labels"]      # (a caller's own module gets what it was written for)
                lab_g = cls if isinstance(self.generator, Generator) else st["labels"]
                feats, l_real, l_fake = self._d_phase(st["images"], lab_d, lab_g, st["masks"], st["noise_d"], st["feats_real"], st["noise_g"])
            st["d_grads"] = [p.grad for p in self._d_params]
            # second graph: what the generator step can do in front of the discriminator's optimizer step - its forward, unless that rode
            # in the discriminator phase's pass (Generator.forward_pair), the diversity loss and the VGG-16 pass over the fake images
            gf = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gf, pool=gd.pool()):
                fake, _ = self._g_forward(st["images"], lab_g, st["masks"], feats, st["noise_g"])
                ahead = self._g_features(fake, st["noise_g"], w_div, st.get("images_next"))
            st["noise_g_early"] = self._g_pair_ok()             # both latents feed the FIRST graph then
            gg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gg, pool=gd.pool()):
                l_g, l_rec, l_div = self._g_rest(fake, st["noise_g"], lab_d, st["masks"], feats, w_rec, w_div, st.get("images_next"),
                                                 st["feats_real"], ahead=ahead)
            st["g_grads"] = [p.grad for p in self._g_params]
        except BaseException:
            ops.drop_wgrad_reduce()            # queued slab reductions of the abandoned capture point into its (released) pool
            self._graph_state = None
            raise
        finally:
            self._capturing = False
            self._vgg_ahead = None
            self._fake_ahead = None
        if handed_over is not None and st["feats_real"] is not None and handed_over[0]._version == handed_over[1]:
            with torch.no_grad():                               # ... the first replay finds that batch's pyramid as the eager loop would
                for dst, src in zip(st["feats_real"], handed_over[2]):
                    dst.copy_(src)
            st["announced"] = (handed_over[0], handed_over[1])
            st["resident_ok"] = handed_over[0] is images_real       # (the resident batch announced as its own successor: bench.py)
        st["gd"], st["gf"], st["gg"], st["feats"] = gd, gf, gg, feats
        st["out"] = {"loss_discriminator_real": l_real.detach(), "loss_discriminator_fake": l_fake.detach(),
                     "loss_generator": l_g.detach(), "loss_generator_semantic_reconstruction": l_rec.detach().reshape(()),
                     "loss_generator_diversity": l_div.detach(), "images_fake": fake.detach()}

    def _flat_ptrs(self):
        """Addresses of the two networks' flat gradient buffers (None where not allocated): what captured graphs are tied to."""
        out = []
        for net in (self.generator, self.discriminator):
            bank = getattr(net, "_bank", None)
            flat = getattr(bank, "flat", None) if bank is not None else None
            out.append(flat.data_ptr() if flat is not None else None)
        return tuple(out)

    def train_step_graphed(self, images_real: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None, masks=None,
                           noise_d: Optional[torch.Tensor] = None, noise_g: Optional[torch.Tensor] = None,
                           next_images_real: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """train_step() through the captured graphs (capture_graphs() first).  images / labels / masks: new batch to copy into
        the graphs' static inputs, or None to reuse the resident one.  next_images_real: as in train_step() - the next call's real
        images, whose VGG pyramid this call computes ahead; without it (and outside the resident mode) the next call computes its
        pyramid eagerly in front of its first graph.  The returned tensors are the graphs' static outputs: they are overwritten by
        the next call."""
        st = self._graph_state
        if st is None or st.get("flat_ptrs") != self._flat_ptrs():
            self._graph_state = None
            raise RuntimeError("train_step_graphed(): the networks' flat gradient buffers were re-allocated after capture_graphs() "
                               "(parameters or compute dtype changed) - capture again")
        if noise_d is None:
            st["noise_d"].normal_()
        else:
            st["noise_d"].copy_(noise_d)
        if st["noise_g_early"]:                                 # both generator forwards run in the first graph: it needs both draws
            if noise_g is None:
                st["noise_g"].normal_()
            else:
                st["noise_g"].copy_(noise_g)
        ahead = st.get("feats_real") is not None
        feats_ok = ahead and st["resident_ok"]                  # resident batch: its pyramid is what the last replay left behind
        if images_real is not None and images_real is not st["images"]:
            st["images"].copy_(images_real, non_blocking=True)
            if ahead:
                ann = st["announced"]
                feats_ok = ann is not None and ann[0] is images_real and ann[1] == images_real._version
                st["next_is_resident"] = False
        if ahead:
            if next_images_real is not None:
                st["images_next"].copy_(next_images_real, non_blocking=True)
                st["next_is_resident"] = False
                st["announced"] = (next_images_real, next_images_real._version)
            else:
                st["announced"] = None
                if images_real is None and not st["next_is_resident"]:
                    # back in the resident mode after streamed batches: the batch to take ahead is the resident one again (round-4
                    # ADVICE: the flag stayed False for good and every later resident replay recomputed its pyramid eagerly)
                    st["images_next"].copy_(st["images"])
                    st["next_is_resident"] = True
            # after this call st["feats_real"] holds the pyramid of st["images_next"]: the resident batch's only while both buffers agree
            st["resident_ok"] = st["next_is_resident"]
            if not feats_ok:                                    # nobody computed this batch's pyramid ahead: now, eagerly
                with torch.no_grad():
                    for dst, src in zip(st["feats_real"], self.vgg16(st["images"])):
                        dst.copy_(src)
        if labels is not None and labels is not st["labels"]:
            st["labels"].copy_(labels, non_blocking=True)
        if masks is not None and masks is not st["masks"]:
            for dst, src in zip(st["masks"], masks):
                dst.copy_(src, non_blocking=True)
        with profiling.range("D phase"):
            st["gd"].replay()
            for p, g in zip(self._d_params, st["d_grads"]):
                p.grad = g
            self._start_reduce("d", self._d_params, eager=False)
        if not st["noise_g_early"]:
            if noise_g is None:
                st["noise_g"].normal_()
            else:
                st["noise_g"].copy_(noise_g)
        with profiling.range("G forward"):
            st["gf"].replay()                                   # (generator forward,) diversity loss, VGG pass: overlap the D gradient all-reduce
        with profiling.range("Adam(D)"):
            self._join_reduce("d")
            self._optimizer_step("d", self.discriminator_optimizer)
        with profiling.range("G rest"):
            st["gg"].replay()
            for p, g in zip(self._g_params, st["g_grads"]):
                p.grad = g
            for p in self._d_params:
                p.grad = None
            self._start_reduce("g", self._g_params, eager=False)
        with profiling.range("Adam(G)"):
            self._join_reduce("g")
            self._optimizer_step("g", self.generator_optimizer)
        self.iterations += 1
        return st["out"]

    # ------------------------------------------------------------------------------------------
    def _batch_signature(self, images_real, labels, masks):
        return (tuple(images_real.shape), images_real.dtype, tuple(labels.shape), labels.dtype,
                tuple((tuple(m.shape), m.dtype) for m in masks), str(images_real.device))

    def _train_iteration(self, images_real, labels, masks, w_rec: float, w_div: float, next_images_real=None) -> Dict[str, torch.Tensor]:
        """One iteration of train(): the first ``graph_after_iterations`` batches (config.CFG.graph_after, default 3; 0 = never)
        run eagerly, then - the batch shapes being static, as with the reference's drop_last loader (main.py:80-88) - the step
        is captured once and every further batch of the same shapes REPLAYS the three HIP graphs (its tensors are copied into
        the graphs' static inputs).  A batch of other shapes, or any failure to capture, runs eagerly.  Replay and eager
        launches execute the same kernels in the same order on the same RNG stream: the logged metrics are identical
        (tests/test_gpu_frontdoor.py).  An iteration without an announced successor (next_images_real is None: an epoch's last) runs
        eagerly even after the capture: the captured generator-step graph always takes the NEXT batch's VGG pyramid in its pass
        over the fake images."""
        after = self.graph_after_iterations
        sig = self._batch_signature(images_real, labels, masks)
        st = self._graph_state
        if st is not None and st.get("flat_ptrs") != self._flat_ptrs():
            if self._reducer_active():
                # one rank replaying while another launches eagerly would pair different sequences of collectives (or hang); a
                # re-allocation is deterministic program state, so it either happens on every rank or is a bug: fail loudly
                raise RuntimeError("ModelWrapper.train(): the flat gradient buffers were re-allocated under a captured graph in a "
                                   "data-parallel job - re-create the ModelWrapper on every rank")
            st = self._graph_state = None                       # gradient buffers re-allocated: eager steps, then a fresh capture
            self._eager_run = 0
        if st is not None and st.get("sig") == sig and st.get("w") == (w_rec, w_div):
            # the captured generator-step graph takes the next batch's VGG pyramid in its pass over the fake images; an iteration
            # without a successor (an epoch's last) runs eagerly instead - the same kernels as the eager loop, one iteration per epoch
            if st.get("feats_real") is None or next_images_real is not None:
                return self.train_step_graphed(images_real, labels, masks, next_images_real=next_images_real)
            ann = st.get("announced")
            if ann is not None and ann[0] is images_real and ann[1] == images_real._version:
                self._vgg_ahead = (images_real, ann[1], st["feats_real"], st.get("mode"))       # computed ahead by the last replay
            st["announced"] = None
        out = self.train_step(images_real, labels, masks, w_rec=w_rec, w_div=w_div, next_images_real=next_images_real)
        if st is not None and st.get("sig") == sig and st.get("w") == (w_rec, w_div) and next_images_real is None:
            return out                                          # (an epoch's last iteration under a valid capture: nothing to re-capture)
        if st is not None and st.get("sig") == sig and st.get("w") != (w_rec, w_div):
            st = self._graph_state = None                       # other loss weights than the captured ones: capture again (round-4 ADVICE)
            self._eager_run = 0
        if after and after > 0 and images_real.is_cuda and not self._graph_failed:
            self._eager_run = self._eager_run + 1 if sig == self._eager_sig else 1
            self._eager_sig = sig
            if self._eager_run >= after:
                try:
                    self.capture_graphs(images_real, labels, masks, w_rec=w_rec, w_div=w_div)
                    self._graph_state["sig"] = sig
                except Exception as exc:                  # a caller's module the capture cannot hold: stay eager, say so once
                    self._graph_state = None
                    self._graph_failed = True
                    import warnings
                    warnings.warn("ModelWrapper.train(): HIP-graph capture failed (%s); continuing with eager launches" % (exc,))
                if self._reducer_active() and not self._ranks_agree(self._graph_state is not None):
                    # data parallelism: every rank replays or none does (bench.py does the same) - the eager path hands its
                    # gradient ranges over group by group, the replay path bucket by bucket: mixed ranks would mismatch
                    self._graph_state = None
                    self._graph_failed = True
        return out

    def _reducer_active(self) -> bool:
        return self.gradient_reducer is not None and self.gradient_reducer.active()

    def _ranks_agree(self, ok: bool) -> bool:
        """all-reduce(MIN) of a per-rank flag over the reducer's process group (one host sync, once per capture attempt)."""
        import torch.distributed as dist
        red = self.gradient_reducer
        dev = next(self.generator.parameters()).device
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if dist.get_backend(red.group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=red.group)
        return bool(int(flag.item()))

    def _device_batches(self, device):
        """One pass over the training loader: (images, labels, masks, next batch's images or None), every batch moved to the device when
        it is FETCHED - one batch ahead where config.CFG.vgg_pair is on, so that an iteration can hand the next one's real images to the
        frozen VGG-16 together with its own fake images (train_step: next_images_real).  The tensors are those of model_wrapper.py:
        133-135; a next batch of another shape (a last, smaller batch) is simply not announced."""
        def fetch(it):
            batch = next(it, None)
            if batch is None:
                return None
            images_real, labels, masks = batch
            return images_real.detach().to(device), labels.to(device), [m.detach().to(device) for m in masks]
        it = iter(self.training_dataset)
        cur = fetch(it)
        while cur is not None:
            nxt = fetch(it) if CFG.vgg_pair else None
            ahead = nxt[0] if nxt is not None and nxt[0].shape == cur[0].shape else None
            yield cur[0], cur[1], cur[2], ahead
            cur = nxt if CFG.vgg_pair else fetch(it)

    def train(self, epochs: int = 20, validate_after_n_iterations: int = 100000, device: str = 'cuda',
              save_model_after_n_epochs: int = 1, w_rec: float = 0.1, w_div: float = 0.1) -> None:
        """model_wrapper.py:93-228."""
        from tqdm import tqdm
        self.logger.hyperparameter['w_rec'] = str(w_rec)
        self.logger.hyperparameter['w_div'] = str(w_div)
        bs = self.training_dataset.batch_size
        validate_after_n_iterations = max(bs, (validate_after_n_iterations // bs) * bs)
        self.generator.train()
        self.discriminator.train()
        self.vgg16.eval()
        self.generator.to(device)
        self.discriminator.to(device)
        self.vgg16.to(device)
        self.progress_bar = tqdm(total=epochs * len(self.training_dataset.dataset), dynamic_ncols=True)
        self.progress_bar.set_description('Validation')
        self.inference(device=device)
        fid = self.validate()
        names = ("loss_generator_diversity", "loss_generator_semantic_reconstruction", "loss_generator",
                 "loss_discriminator_fake", "loss_discriminator_real")
        for epoch in range(epochs):
            self.generator.train()
            self.discriminator.train()
            self.vgg16.eval()
            for images_real, labels, masks, next_images_real in self._device_batches(device):
                self.progress_bar.update(n=images_real.shape[0])
                out = self._train_iteration(images_real, labels, masks, w_rec, w_div, next_images_real)
                vals = torch.stack([out[n].float().reshape(()) for n in names]).tolist()      # the single host sync
                l_div, l_rec, l_g, l_df, l_dr = vals
                self.progress_bar.set_description(
                    'FID={:.4f}, Loss Div={:.4f}, Loss Rec={:.4f}, Loss G={:.4f}, Loss D={:.4f}'.format(
                        fid, l_div, l_rec, l_g, l_df + l_dr))
                self.logger.log(metric_name='loss_discriminator_real', value=l_dr)
                self.logger.log(metric_name='loss_discriminator_fake', value=l_df)
                self.logger.log(metric_name='loss_generator', value=l_g)
                self.logger.log(metric_name='loss_generator_semantic_reconstruction', value=l_rec)
                self.logger.log(metric_name='loss_generator_diversity', value=l_div)
                self.logger.log(metric_name='iterations', value=self.progress_bar.n)
                self.logger.log(metric_name='epoch', value=epoch)
                if self.progress_bar.n % validate_after_n_iterations == 0:
                    self.progress_bar.set_description('Validation')
                    fid = self.validate()
                    self.inference(device=device)
                    self.logger.log(metric_name='fid', value=fid)
                    self.logger.log(metric_name='iterations_fid', value=self.progress_bar.n)
                    if self.path_save_metrics is not None:
                        self.logger.save_metrics(self.path_save_metrics)
            if epoch % save_model_after_n_epochs == 0 and self.path_save_models is not None:
                torch.save({"generator": self.generator.state_dict(),
                            "discriminator": self.discriminator.state_dict(),
                            "generator_optimizer": self.generator_optimizer.state_dict(),
                            "discriminator_optimizer": self.discriminator_optimizer.state_dict()},
                           os.path.join(self.path_save_models, 'checkpoint_{}.pt'.format(str(epoch).zfill(3))))
            self.inference(device=device)
            if self.path_save_metrics is not None:
                self.logger.save_metrics(self.path_save_metrics)
        self.progress_bar.close()

    @torch.no_grad()
    def validate(self) -> float:
        """FID needs a downloaded Inception-v3 (frechet_inception_distance.py:22) - outside the hot path."""
        return float('nan')

    @torch.no_grad()
    def inference(self, device: str = 'cuda') -> None:
        """model_wrapper.py:247-296: 7 validation images x the 7 single-stage mask sets (misc.get_masks_for_inference), one fake image
        each from a fresh latent, generator in eval mode, saved as the reference's 7 x 7 grid ``predictions_<n>.png`` (every image
        scaled to [0, 1] by its own range, misc.normalize_0_1_batch); the generator goes back to training mode.  The draws follow the
        reference's order: np.random.choice over range(len(validation loader)), then one torch.randn per (image, stage)."""
        if self.validation_dataset_fid is None or self.path_save_plots is None:
            return
        from .data import image_label_list_of_masks_collate_function
        import numpy as np
        self.generator.to(device)
        self.vgg16.to(device)
        self.generator.eval()
        try:
            idx = np.random.choice(range(len(self.validation_dataset_fid)), replace=False, size=7)
            images, labels, _ = image_label_list_of_masks_collate_function([self.validation_dataset_fid.dataset[i] for i in idx])
            masks_levels = [misc.get_masks_for_inference(stage, add_batch_size=True, device=device) for stage in range(7)]
            fakes = torch.empty((7 ** 2,) + tuple(images.shape[1:]), dtype=torch.float32, device=device)
            counter = 0
            for image, label in zip(images, labels):
                image, label = image.detach().to(device)[None], label.to(device)[None]
                feats = self.vgg16(image)                      # (frozen, eval mode: the reference recomputes the same pyramid per stage)
                for masks in masks_levels:
                    z = torch.randn(1, self.latent_dimensions, dtype=torch.float32, device=device)
                    fakes[counter] = self.generator(input=z, features=feats, masks=masks, class_id=label.float()).float()[0]
                    counter += 1
            n = getattr(self, "progress_bar", None)
            misc.save_image_grid(misc.normalize_0_1_batch(fakes), os.path.join(self.path_save_plots, 'predictions_{}.png'.format(n.n if n is not None else 0)),
                                 nrow=7)
        finally:
            self.generator.train()
