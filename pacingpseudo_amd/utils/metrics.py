"""Validation Dice (utils/metrics.py:7-34) with the counting done on the GPU.

``compute_dice(prob, target)`` keeps the reference's per-sample signature (C x H x W arrays/tensors, returns a list
of per-class Dice values with NaN when prediction and target are both empty).  ``batch_dice`` is the batched
form the training driver uses: one HIP launch produces |P&T|, |P|, |T| for every (sample, class), so the
per-sample ``.cpu().numpy()`` round trips of train_chaos.py:388 disappear."""
import numpy as np
import torch

from .._lib import lib, stream_ptr


def batch_dice(logits_or_prob: torch.Tensor, target_onehot: torch.Tensor) -> np.ndarray:
    """(N,C,H,W) scores (soft-max or logits: arg-max is the same) and one-hot labels -> (N,C) Dice, NaN = skip."""
    x = logits_or_prob.contiguous().float()
    t = target_onehot.contiguous().float()
    assert x.shape == t.shape and x.is_cuda and t.is_cuda
    N, C, H, W = x.shape
    counts = torch.empty((N, C, 3), device=x.device, dtype=torch.float32)
    lib.pp_dice_counts(x.data_ptr(), t.data_ptr(), N, C, H * W, counts.data_ptr(), stream_ptr())
    c = counts.double().cpu().numpy()
    inter, ps, ts = c[..., 0], c[..., 1], c[..., 2]
    with np.errstate(invalid='ignore'):
        dice = 2 * inter / (ps + ts + 1e-5)
    dice[(ps == 0) & (ts == 0)] = np.nan
    return dice


def compute_dice(input, target):
    """Per-class Dice of one sample: input C x H x W soft-max values, target C x H x W one-hot."""
    assert tuple(input.shape) == tuple(target.shape)
    dev = torch.device('cuda', torch.cuda.current_device())
    x = torch.as_tensor(np.asarray(input) if not torch.is_tensor(input) else input, dtype=torch.float32, device=dev)
    t = torch.as_tensor(np.asarray(target) if not torch.is_tensor(target) else target, dtype=torch.float32, device=dev)
    return list(batch_dice(x[None], t[None])[0])
