"""Loss functions of PacingPseudo with the reference's names and call signatures (losses/losses.py:9-116),
evaluated by the fused HIP loss kernels of libpacingpseudo_hip.so.

Inside ``ConsistencyRegulr`` the five losses of a step are produced by ONE fused forward and ONE fused backward
launch sequence (pacingpseudo_amd/engine.py).  The functions below expose the same kernels one loss at a time for
code that calls the reference's functional API directly; each is a ``torch.autograd.Function`` whose backward
runs the matching HIP gradient kernel.  Inputs are NCHW float32 CUDA tensors, exactly as in the reference.

``soft_label_cross_entropy_loss`` / ``l1_loss`` / ``l2_loss`` receive the target as probabilities and, like the
reference's (losses/losses.py:45-96), differentiate through it: the gradient with respect to ``input`` comes from the
HIP kernel, the one with respect to ``target`` -- element-wise, -mask * log_softmax(input) / D and its L1 / L2
counterparts -- is attached by ``_TargetGrad`` when the target requires one (round 4; before, the target was silently
a constant).  The in-model path evaluates both gradients inside the fused kernels (consistency_reglur_memory.py:53-54).
The targets are expected to be normalised distributions (they pass through log / soft-max on their way to the kernels).
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


class _TargetGrad(torch.autograd.Function):
    """A zero-valued term that carries d loss / d target for the probability-target losses: forward returns 0, backward
    returns g * dLdt, where dLdt is the closed-form element-wise derivative (no reduction: plain tensor plumbing)."""

    @staticmethod
    def forward(ctx, target, dLdt):
        ctx.save_for_backward(dLdt)
        return torch.zeros((), device=target.device, dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        (dLdt,) = ctx.saved_tensors
        return g * dLdt, None


def _with_target_grad(loss, target, make_dLdt, valid_mask, per_pixel: bool):
    """loss + 0 * (term whose gradient w.r.t. `target` is make_dLdt() * mask / denominator).  Denominators as in the
    reference: max(sum(mask), 1e-8) when masked; else the element count of the un-reduced loss tensor -- (N,C,H,W) for the
    soft-label CE, (N,1,H,W) for L1 / L2 (losses/losses.py:56-61, 74-78, 91-95)."""
    if not (torch.is_tensor(target) and target.requires_grad and torch.is_grad_enabled()):
        return loss
    with torch.no_grad():
        d = make_dLdt()
        if valid_mask is not None:
            d = d * valid_mask / valid_mask.sum().clamp_min(1e-8)
        else:
            N, C, H, W = target.shape
            d = d / float(N * H * W * (1 if per_pixel else C))
    return loss + _TargetGrad.apply(target, d)


def soft_label_cross_entropy_loss(input, target, valid_mask=None):
    """-sum_c target_c * log_softmax(input)_c, masked mean (losses/losses.py:45-62).  `target`: probabilities; gradients
    flow into both arguments."""
    loss = _consistency(input, _logits_of(target), valid_mask, 'ce_loss', True)
    return _with_target_grad(loss, target, lambda: -torch.log_softmax(input.detach(), 1), valid_mask, per_pixel=False)


def l1_loss(input, target, valid_mask=None):
    """sum_c |input_c - target_c| on probabilities, masked mean (losses/losses.py:64-79)."""
    loss = _consistency(_logits_keep_grad(input), _logits_of(target), valid_mask, 'l1_loss', True)
    return _with_target_grad(loss, target, lambda: -torch.sign(input.detach() - target.detach()), valid_mask, per_pixel=True)


def l2_loss(input, target, valid_mask=None):
    """sum_c (input_c - target_c)^2 on probabilities, masked mean (losses/losses.py:81-96)."""
    loss = _consistency(_logits_keep_grad(input), _logits_of(target), valid_mask, 'l2_loss', True)
    return _with_target_grad(loss, target, lambda: -2.0 * (input.detach() - target.detach()), valid_mask, per_pixel=True)


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


class _WeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, *terms):
        import ctypes
        n = len(terms)
        for t in terms:
            if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.numel() == 1):
                raise TypeError('weighted_loss_sum: every term must be a one-element float32 CUDA tensor')
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in terms])
        w = (ctypes.c_float * n)(*[float(x) for x in weights])
        out = torch.empty((), device=terms[0].device, dtype=torch.float32)
        lib.pp_weighted_sum_fwd(ptrs, w, n, out.data_ptr(), stream_ptr())
        ctx.w, ctx.n = w, n
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.to(torch.float32).contiguous()
        gout = torch.empty(ctx.n, device=g.device, dtype=torch.float32)
        lib.pp_weighted_sum_bwd(g.data_ptr(), ctx.w, ctx.n, gout.data_ptr(), stream_ptr())
        return (None,) + tuple(gout[i] for i in range(ctx.n))


def weighted_loss_sum(terms, weights):
    """total = terms[0] * weights[0] + terms[1] * weights[1] + ... -- the loss assembly of the training loop
    (train_chaos.py:273-310) as ONE launch forward and ONE backward instead of a chain of element-wise launches on 0-dim
    tensors.  Same arithmetic as that chain (fp32 products and sums, left to right): bit-identical totals and gradients.
    terms: 0-dim float32 CUDA tensors (at most 8); weights: Python floats."""
    terms = list(terms)
    if len(terms) != len(weights) or not 1 <= len(terms) <= 8:
        raise ValueError('weighted_loss_sum: 1..8 terms, one weight each')
    return _WeightedSum.apply(tuple(float(w) for w in weights), *terms)

