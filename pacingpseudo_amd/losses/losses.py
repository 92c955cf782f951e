"""Loss functions of PacingPseudo with the reference's names and call signatures (losses/losses.py:9-116),
evaluated by the fused HIP loss kernels of libpacingpseudo_hip.so.

Inside ``ConsistencyRegulr`` the five losses of a step are produced by ONE fused forward and ONE fused backward
launch sequence (pacingpseudo_amd/engine.py).  The functions below expose the same kernels one loss at a time for
code that calls the reference's functional API directly; each is a ``torch.autograd.Function`` whose backward
runs the matching HIP gradient kernel.  Inputs are NCHW float32 CUDA tensors, exactly as in the reference.

Difference to note: ``soft_label_cross_entropy_loss`` / ``l1_loss`` / ``l2_loss`` receive the target as
probabilities; here the target is treated as a constant (no gradient flows into it).  The in-model path keeps the
reference's differentiable target (consistency_reglur_memory.py:53-54).
"""
from __future__ import annotations

import torch

from .._lib import lib, stream_ptr

_VARIANT = {'ce_loss': 1, 'l1_loss': 2, 'l2_loss': 3, 'kl_loss': 4}


def _check(t, name):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4):
        raise TypeError(f'{name} must be a 4-D float32 CUDA tensor (N,C,H,W)')
    return t.contiguous()


class _SegLoss(torch.autograd.Function):
    """which: 0 = partial CE, 1 = entropy, 2 = consistency(variant)."""

    @staticmethod
    def forward(ctx, zw, zs, target, mask, ignore_index, which, variant, detach_weak):
        N, K, H, W = zw.shape
        st = stream_ptr()
        dev = zw.device
        sums = torch.zeros(6, device=dev, dtype=torch.float64)
        nws = lib.pp_seg_losses_workspace(N, H * W)
        ws = torch.empty(nws + 64, device=dev, dtype=torch.uint8)
        if target is None:
            target = torch.full((N, H, W), ignore_index, device=dev, dtype=torch.int64)
        lib.pp_seg_losses_fwd(zw.data_ptr(), zs.data_ptr() if zs is not None else None, target.data_ptr(),
                              mask.data_ptr() if mask is not None else None, N, K, H * W, ignore_index,
                              1 if which == 1 else 0, variant if which == 2 else 0, sums.data_ptr(), ws.data_ptr(), nws, st)
        out = torch.empty((), device=dev, dtype=torch.float32)
        ptrs = [None, None, None]
        ptrs[which] = out.data_ptr()
        lib.pp_losses_finalize(sums.data_ptr(), 1 if mask is not None else 0, *ptrs, st)
        ctx.save_for_backward(zw, zs if zs is not None else zw, target, mask if mask is not None else zw, sums)
        ctx.cfg = (zs is not None, mask is not None, ignore_index, which, variant, detach_weak)
        return out

    @staticmethod
    def backward(ctx, g):
        zw, zs, target, mask, sums = ctx.saved_tensors
        has_s, has_m, ignore_index, which, variant, detach_weak = ctx.cfg
        N, K, H, W = zw.shape
        g = g.to(torch.float32).contiguous()
        gp = [None, None, None]
        gp[which] = g.data_ptr()
        dzw = torch.empty_like(zw)
        dzs = torch.empty_like(zs) if has_s else None
        lib.pp_seg_losses_bwd(zw.data_ptr(), zs.data_ptr() if has_s else None, target.data_ptr(),
                              mask.data_ptr() if has_m else None, N, K, H * W, ignore_index, 1 if which == 1 else 0,
                              variant if which == 2 else 0, 1 if detach_weak else 0, sums.data_ptr(), gp[0], gp[1], gp[2],
                              1.0, dzw.data_ptr(), dzs.data_ptr() if has_s else None, stream_ptr())
        return dzw, dzs, None, None, None, None, None, None


def partial_cross_entropy_loss(input, target, ignore_index):
    """F.cross_entropy(input, target, ignore_index): mean over labelled pixels (losses/losses.py:35-43)."""
    zw = _check(input, 'input')
    t = target.to(torch.int64).contiguous()
    return _SegLoss.apply(zw, None, t, None, int(ignore_index), 0, 0, False)


def cross_entropy_loss(input, target):
    """Plain mean cross entropy on (N,C,H,W) logits / (N,H,W) targets or (N,C) / (N) (losses/losses.py:26-33)."""
    if input.dim() == 2:
        input = input.t().contiguous()[None, :, :, None].contiguous()      # (1,C,N,1)
        target = target[None, :, None]
    return partial_cross_entropy_loss(input, target, -100)


def entropy_minimization_loss(input, valid_mask=None):
    """Masked mean of the per-pixel soft-max entropy (losses/losses.py:9-24)."""
    zw = _check(input, 'input')
    m = _check(valid_mask, 'valid_mask') if valid_mask is not None else None
    return _SegLoss.apply(zw, None, None, m, -100, 1, 0, False)


def _consistency(strong_logits, weak_logits, valid_mask, variant, detach_weak):
    zs, zw = _check(strong_logits, 'input'), _check(weak_logits, 'target')
    m = _check(valid_mask, 'valid_mask') if valid_mask is not None else None
    return _SegLoss.apply(zw, zs, None, m, -100, 2, _VARIANT[variant], detach_weak)


def _logits_of(prob):
    return torch.log(prob.detach().clamp_min(1e-38))


def soft_label_cross_entropy_loss(input, target, valid_mask=None):
    """-sum_c target_c * log_softmax(input)_c, masked mean (losses/losses.py:45-62).  `target`: probabilities."""
    return _consistency(input, _logits_of(target), valid_mask, 'ce_loss', True)


def l1_loss(input, target, valid_mask=None):
    """sum_c |input_c - target_c| on probabilities, masked mean (losses/losses.py:64-79)."""
    return _consistency(_logits_keep_grad(input), _logits_of(target), valid_mask, 'l1_loss', True)


def l2_loss(input, target, valid_mask=None):
    """sum_c (input_c - target_c)^2 on probabilities, masked mean (losses/losses.py:81-96)."""
    return _consistency(_logits_keep_grad(input), _logits_of(target), valid_mask, 'l2_loss', True)


def _logits_keep_grad(prob):
    # softmax(log p) == p for a normalised p, so the probability-space losses reuse the logit-space kernels
    return torch.log(prob.clamp_min(1e-38))


def kl_loss(input, target, valid_mask=None):
    """KL(softmax(target) || softmax(input)) element-wise, masked mean; both arguments are logits and both
    receive gradients (losses/losses.py:98-116)."""
    return _consistency(input, target, valid_mask, 'kl_loss', False)


class _DiceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, label):
        N, K, H, W = z.shape
        st = stream_ptr()
        sums = torch.empty((N, K, 3), device=z.device, dtype=torch.float64)
        nws = lib.pp_dice_loss_workspace(N, K)
        ws = torch.empty(nws, device=z.device, dtype=torch.uint8)
        out = torch.empty((), device=z.device, dtype=torch.float32)
        lib.pp_dice_loss_fwd(z.data_ptr(), label.data_ptr(), N, K, H * W, sums.data_ptr(), out.data_ptr(), ws.data_ptr(), nws, st)
        ctx.save_for_backward(z, label, sums)
        return out

    @staticmethod
    def backward(ctx, g):
        z, label, sums = ctx.saved_tensors
        N, K, H, W = z.shape
        g = g.to(torch.float32).contiguous()
        dz = torch.empty_like(z)
        lib.pp_dice_loss_bwd(z.data_ptr(), label.data_ptr(), N, K, H * W, sums.data_ptr(), g.data_ptr(), 1.0, dz.data_ptr(), 0,
                             stream_ptr())
        return dz, None


def dice_loss_fn(input, target):
    """-mean_{n,c} 2 sum(p t) / (sum p + sum t + 1e-5) on softmax(input) vs the one-hot target (losses/losses.py:147-162;
    the fully-supervised trainer upper_bound_chaos.py:165 adds it to the cross entropy)."""
    return _DiceLoss.apply(_check(input, 'input'), _check(target, 'target'))
