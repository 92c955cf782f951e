"""Validation Dice (utils/metrics.py:7-34) with the counting done on the GPU.

``compute_dice(prob, target)`` keeps the reference's per-sample signature (C x H x W arrays/tensors, returns a list
of per-class Dice values with NaN when prediction and target are both empty).  ``batch_dice`` is the batched
form the training driver uses: one HIP launch produces |P&T|, |P|, |T| for every (sample, class), so the
per-sample ``.cpu().numpy()`` round trips of train_chaos.py:388 disappear."""
import numpy as np
import torch

from .._lib import lib, stream_ptr


def dice_counts_device(logits_or_prob: torch.Tensor, target_onehot: torch.Tensor) -> torch.Tensor:
    """(N,C,H,W) scores and one-hot labels -> (N,C,3) float32 device tensor |P & T|, |P|, |T| of the arg-max prediction."""
    x = logits_or_prob.contiguous().float()
    t = target_onehot.contiguous().float()
    assert x.shape == t.shape and x.is_cuda and t.is_cuda
    N, C, H, W = x.shape
    counts = torch.empty((N, C, 3), device=x.device, dtype=torch.float32)
    lib.pp_dice_counts(x.data_ptr(), t.data_ptr(), N, C, H * W, counts.data_ptr(), stream_ptr())
    return counts


def batch_dice_counts(logits_or_prob: torch.Tensor, target_onehot: torch.Tensor) -> np.ndarray:
    """(N,C,H,W) scores and one-hot labels -> (N,C,3) = |P & T|, |P|, |T| of the arg-max prediction, one launch."""
    return dice_counts_device(logits_or_prob, target_onehot).double().cpu().numpy()


class ValAccumulator:
    """The validation meters of train_chaos.py:371-395 kept ON THE DEVICE: per class the sum and the count of the per-sample
    Dice values that are not NaN (``AvgMeter.update`` per sample and class, :388-392) and the n-weighted loss (:383).
    ``update`` enqueues work only; ``result`` is the one host sync of a validation epoch.  Under data-parallel runs every
    rank scores its share of the validation set and ``result(comm)`` sums the accumulators over the ranks first."""

    def __init__(self, num_classes: int, device):
        self.K = int(num_classes)
        self.acc = torch.zeros(2 * self.K + 2, device=device, dtype=torch.float64)   # [sum dice | count | sum loss*n, n]

    def update(self, logits: torch.Tensor, target_onehot: torch.Tensor, loss_pce=None):
        c = dice_counts_device(logits, target_onehot).double()
        inter, ps, ts = c[..., 0], c[..., 1], c[..., 2]
        valid = ~((ps == 0) & (ts == 0))                        # utils/metrics.py:29-31: both empty -> NaN -> skipped
        dice = torch.where(valid, 2 * inter / (ps + ts + 1e-5), torch.zeros_like(inter))
        K, n = self.K, logits.shape[0]
        self.acc[:K] += dice.sum(0)
        self.acc[K:2 * K] += valid.double().sum(0)
        if loss_pce is not None:
            self.acc[2 * K] += loss_pce.detach().double() * n
        self.acc[2 * K + 1] += n

    def result(self, all_reduce=None):
        """-> (per-class mean Dice (numpy, NaN for a class no sample had), mean loss, samples)."""
        if all_reduce is not None:
            all_reduce(self.acc)
        a = self.acc.cpu().numpy()
        K = self.K
        with np.errstate(invalid='ignore', divide='ignore'):
            avg = np.where(a[K:2 * K] > 0, a[:K] / a[K:2 * K], 0.0)      # AvgMeter.avg of an empty meter is 0
        return avg, float(a[2 * K] / max(a[2 * K + 1], 1)), int(a[2 * K + 1])


def batch_dice(logits_or_prob: torch.Tensor, target_onehot: torch.Tensor) -> np.ndarray:
    """(N,C,H,W) scores (soft-max or logits: arg-max is the same) and one-hot labels -> (N,C) Dice, NaN = skip."""
    c = batch_dice_counts(logits_or_prob, target_onehot)
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


def batch_hd95(pred_hard: torch.Tensor, label: torch.Tensor, num_classes: int, spacing=(1.0, 1.0)) -> np.ndarray:
    """95 % Hausdorff distance per (sample, class) of hard class maps (N,H,W) -- the batched form of the reference's
    ``inference.py:_compute_95hd`` (``medpy.metric.binary.hd95(pred == k, label == k, spacing, 1)`` per class).
    Surfaces and both directed surface-distance sets are computed on the GPU (pp_hd95_surface_distances); the 95th
    percentile of the joined sets (``numpy.percentile``, linear interpolation, as medpy) is taken on the host.
    NaN where prediction or label is empty or fills the image (inference.py:231-232)."""
    p = pred_hard.contiguous().to(torch.int64)
    t = label.contiguous().to(torch.int64)
    assert p.shape == t.shape and p.dim() == 3 and p.is_cuda and t.is_cuda
    N, H, W = p.shape
    K = int(num_classes)
    sy, sx = (float(spacing), float(spacing)) if np.isscalar(spacing) else (float(spacing[0]), float(spacing[1]))
    dist = torch.empty((N * K, 2, H * W), device=p.device, dtype=torch.float32)
    counts = torch.empty((N * K, 4), device=p.device, dtype=torch.int32)
    nws = lib.pp_hd95_workspace(N, K, H, W)
    ws = torch.empty(nws, device=p.device, dtype=torch.uint8)
    lib.pp_hd95_surface_distances(p.data_ptr(), t.data_ptr(), N, K, H, W, sy, sx, dist.data_ptr(), counts.data_ptr(),
                                  ws.data_ptr(), nws, stream_ptr())
    cnt = counts.cpu().numpy()
    d = dist.cpu().numpy()
    out = np.full((N, K), np.nan)
    for i in range(N * K):
        na, nb, ta, tb = (int(v) for v in cnt[i])
        if ta == 0 or tb == 0 or ta == H * W or tb == H * W:
            continue
        out[i // K, i % K] = np.percentile(np.hstack((d[i, 0, :na], d[i, 1, :nb])), 95)
    return out


def compute_95hd(pred_hard, label, num_classes, spacing):
    """Per-class HD95 of one sample (inference.py:217-237): pred_hard / label (H,W) class maps."""
    dev = torch.device('cuda', torch.cuda.current_device())
    p = torch.as_tensor(np.asarray(pred_hard), device=dev)[None]
    t = torch.as_tensor(np.asarray(label), device=dev)[None]
    return list(batch_hd95(p, t, num_classes, spacing)[0])
