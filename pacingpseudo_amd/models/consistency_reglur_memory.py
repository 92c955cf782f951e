"""ConsistencyRegulr: the siamese PacingPseudo model, MI355X-native.

Drop-in for the reference's ``models/consistency_reglur_memory.py``:
``ConsistencyRegulr(kwargs_unet, kwargs_aux_path, args_parser)`` and
``model(names_to_data, mode, step) -> dict`` keep the reference's signature, dictionary keys, assertion /
ValueError behaviour and ``state_dict`` layout (consistency_reglur_memory.py:15-102).  Behind that boundary the
whole step is ONE autograd node: the forward runs the static HIP plan of ``StepEngine`` (both views of the
siamese pair in the same launches), and ``loss.backward()`` runs the hand-derived backward plan, depositing
weight gradients into a flat slab (``model.flat``) that ``pacingpseudo_amd.optim.FusedAdam`` and the RCCL
all-reduce consume in place.

Reference behaviours kept on purpose:
  * BatchNorm mode follows ``self.training`` (train_chaos.py:370 switches to eval after epoch 0 and never back);
  * the auxiliary path reads the features of the LAST backbone pass (the strong view when
    ``--do_decoder_consistency``), because the reference's UNet mutates one ``end_points`` dict in place;
  * only batch sample 0 updates the memory bank;
  * the weak probabilities are a differentiable target of the consistency loss unless ``--detach_weak_cr``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .unet import UNet
from .aux_path_memory import AuxPath
from ..engine import StepEngine
from ..flat import FlatSlab

_LOSS_KEYS = ('loss_pce', 'loss_ent', 'loss_cr', 'loss_aux_cls', 'loss_memory')


class _StepFunction(torch.autograd.Function):
    """One autograd node for the whole step: outputs the losses (differentiable) and logits (not)."""

    @staticmethod
    def forward(ctx, model, batch, mode, step, anchor):
        out = model.engine.forward_step(batch, mode, step, need_grad=True)
        ctx.model = model
        ctx.state = model.engine.last          # this forward's record: a later forward must not be differentiated for it
        ctx.loss_names = [k for k in _LOSS_KEYS if k in out]
        ctx.other_names = [k for k in out if k not in ctx.loss_names]
        others = [out[k] for k in ctx.other_names]
        ctx.mark_non_differentiable(*others)
        # no zero tensors for the outputs nobody differentiated: autograd would otherwise fill a logits-sized gradient (42 MB at the
        # benchmark shape) for each of the three non-differentiable outputs at the start of every backward pass
        ctx.set_materialize_grads(False)
        model._last_names = ctx.loss_names + ctx.other_names
        return tuple(out[k] for k in model._last_names)

    @staticmethod
    def backward(ctx, *gouts):
        model = ctx.model
        g = {name: gouts[i] for i, name in enumerate(ctx.loss_names)}
        model._run_backward(g, ctx.state)
        return None, None, None, None, None


class ConsistencyRegulr(nn.Module):
    def __init__(self, kwargs_unet, kwargs_aux_path=None, args_parser=None):
        super().__init__()
        self.kwargs_unet = kwargs_unet
        self.kwargs_aux_path = kwargs_aux_path
        self.args = args_parser
        self.backbone = UNet(**kwargs_unet)
        self.aux_path = AuxPath(**kwargs_aux_path)
        self.engine = StepEngine(self.backbone, self.aux_path, self.args)
        self.flat = None
        self._reducer = None              # pacingpseudo_amd.parallel.GradReducer when data-parallel
        # a loaded checkpoint changes every weight: forward-only plans must re-pack their kernel-side layouts
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.engine.invalidate_packed())

    # ---- flat parameter / gradient slabs -------------------------------------------------------
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)          # .cuda() / .to(): parameters get new storage
        self.flat = None
        if self.backbone.final_conv.weight.is_cuda:
            self._flatten()
        return out

    def _flatten(self):
        segs = [('backbone', [p for p in self.backbone.parameters() if p.requires_grad]),
                ('aux_path', [p for p in self.aux_path.parameters() if p.requires_grad])]
        self.flat = FlatSlab(segs)
        return self.flat

    def _ensure_flat(self):
        if self.flat is None or not self.flat.owns(self.backbone.final_conv.weight):
            self._flatten()
        return self.flat

    # ---- forward / backward --------------------------------------------------------------------
    def forward(self, names_to_data, mode=None, step=None):
        assert mode in ['train', 'val', None]
        if not self.backbone.final_conv.weight.is_cuda:
            raise RuntimeError('pacingpseudo_amd runs on the GPU only: call model.cuda() first '
                               '(there is no CPU fallback for the HIP path)')
        self._ensure_flat()
        if torch.is_grad_enabled() and mode == 'train':
            outs = _StepFunction.apply(self, names_to_data, mode, step, self.backbone.final_conv.weight)
            got = dict(zip(self._last_names, outs))
            return {k: got[k] for k in self._expected_keys(mode)}      # the reference's key order
        with torch.no_grad():
            return self.engine.forward_step(names_to_data, mode, step, need_grad=False)

    def _expected_keys(self, mode):
        """Keys of net_outputs in the order StepEngine.forward_step inserts them."""
        a = self.args
        keys = ['segmentation/logits', 'loss_pce']
        if mode == 'train':
            if a.do_loss_ent:
                keys.append('loss_ent')
            if a.do_decoder_consistency:
                keys += ['loss_cr', 'segmentation/logits_strong']
            if a.do_aux_path:
                keys += ['logits_aux_cls', 'loss_aux_cls']
                if a.do_memory:
                    keys.append('loss_memory')
        return keys

    def _run_backward(self, g, state):
        flat = self._ensure_flat()
        active = ['backbone'] + (['aux_path'] if state['do_aux'] else [])
        red = self._reducer
        self.engine.bucket_hook = (lambda tag: red.bucket_ready(flat, tag)) if red is not None else None
        self.engine.backward_step(g, flat.grad_views, state)
        flat.publish_grads(active)
        if red is not None:
            red.reduce(flat, active)
        self.engine.unscale_grads(state, [flat.segment(n, 'grads') for n in active], flat)
