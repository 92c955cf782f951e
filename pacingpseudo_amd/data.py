"""Minimal input side of the training driver.

The reference's CPU augmentation pipeline (datasets/augmentations.py, scipy/skimage/cv2) is OUTSIDE the hot path
this package rebuilds ("next").  What is here is only what the driver needs to run:

  * ``NpzSlices``: reads the reference's on-disk format (``.npz`` with ``uid/img/lab/scb``,
    datasets/chaos/chaos_dataset.py:92-105), applies MeanStdNorm (augmentations.py:11-21), one-hot encodes label and
    scribble (augmentations.py:448-461) and centre-crops / zero-pads to the network size with a matching
    ``valid_mask``.  The strong view is a per-sample brightness / contrast jitter (the colour part of
    chaos_aug_configs.py:71-78).  This CPU path does no geometric augmentation: the reference's full two-stream
    pipeline runs on the GPU (``augment.DeviceAugmenter``, ``--gpu_augment``), fed by ``raw=True`` datasets.
  * ``SyntheticPhantoms``: ellipse "organs" with skeleton-like scribbles, used when no dataset is on disk
    (CHAOS / ACDC / LVSC are external downloads, README.md:9-11) and by the Dice-parity runs.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset


def _one_hot(lab: np.ndarray, n: int) -> np.ndarray:
    out = np.zeros((n,) + lab.shape, dtype=np.float32)
    for c in range(n):
        out[c][lab == c] = 1
    return out


def _fit(a: np.ndarray, size: int, fill=0):
    """Centre-crop / pad a (H,W) array to (size,size); returns the array and the validity mask."""
    out = np.full((size, size), fill, dtype=a.dtype)
    valid = np.zeros((size, size), dtype=np.float32)
    h, w = a.shape
    ch, cw = min(h, size), min(w, size)
    sy, sx = (h - ch) // 2, (w - cw) // 2
    dy, dx = (size - ch) // 2, (size - cw) // 2
    out[dy:dy + ch, dx:dx + cw] = a[sy:sy + ch, sx:sx + cw]
    valid[dy:dy + ch, dx:dx + cw] = 1
    return out, valid


def _strong(image: np.ndarray, rng: np.random.Generator, strength: float) -> np.ndarray:
    a = 1.0 + strength * rng.uniform(-0.8, 0.8)
    b = strength * rng.uniform(-0.8, 0.8)
    return (image * a + b).astype(np.float32)


class NpzSlices(Dataset):
    def __init__(self, file_ls, num_classes, size=256, do_strong=False, strength=1.0, train=True, seed=1, raw=False):
        self.files, self.K, self.size = list(file_ls), num_classes, size
        self.do_strong, self.strength, self.train = do_strong, strength, train
        self.seed, self.epoch, self._index = int(seed), 0, 0
        self.rng = None             # a caller may pin the strong-view generator (tests/studies/dice_study.py does, per sample)
        self.raw = raw              # un-augmented {'img', 'lab', 'scb'} arrays for augment.DeviceAugmenter (collate_raw)

    def __len__(self):
        return len(self.files)

    def _sample(self, img, lab, scb):
        if self.raw:
            return {'img': img.astype(np.float32), 'lab': lab.astype(np.int32), 'scb': scb.astype(np.int32)}
        img = img.astype(np.float32)
        img = (img - img.mean()) / (img.std() + 1e-8)                 # MeanStdNorm
        img, valid = _fit(img, self.size)
        lab, _ = _fit(lab.astype(np.int64), self.size)
        scb, _ = _fit(scb.astype(np.int64), self.size, fill=self.K)   # outside the slice = ignored
        d = {'image': torch.from_numpy(img[None]), 'label': torch.from_numpy(_one_hot(lab, self.K)),
             'scribble': torch.from_numpy(_one_hot(scb, self.K + 1))}
        if self.train:
            d['valid_mask'] = torch.from_numpy(valid[None])
            if self.do_strong:
                # one generator per (seed, epoch, sample): DataLoader workers are forked copies of this object, a shared
                # generator would replay the same jitter sequence in every worker and every epoch
                rng = self.rng if self.rng is not None else np.random.default_rng([self.seed, self.epoch, self._index])
                d['image_strong'] = torch.from_numpy(_strong(img, rng, self.strength)[None])
                d['label_strong'] = d['label']
        return d

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __getitem__(self, i):
        self._index = int(i)
        z = np.load(self.files[i])
        return self._sample(z['img'], z['lab'], z['scb'])


class SyntheticPhantoms(NpzSlices):
    """`n` deterministic slices: K-1 ellipses on noise; scribbles = a short stroke inside each structure."""

    def __init__(self, n, num_classes, size=256, do_strong=False, strength=1.0, train=True, seed=1, raw=False):
        super().__init__([None] * n, num_classes, size, do_strong, strength, train, seed, raw)
        self.base_seed = seed + (0 if train else 10_000)

    def __getitem__(self, i):
        self._index = int(i)
        rng = np.random.default_rng(self.base_seed * 100_003 + i)
        S, K = self.size, self.K
        yy, xx = np.mgrid[0:S, 0:S].astype(np.float32)
        lab = np.zeros((S, S), np.int64)
        scb = np.full((S, S), K, np.int64)
        img = rng.normal(0, 0.3, (S, S)).astype(np.float32)
        for c in range(1, K):
            cy, cx = rng.uniform(0.25, 0.75, 2) * S
            ry, rx = rng.uniform(0.06, 0.16, 2) * S
            m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
            lab[m] = c
            img[m] += 0.6 + 0.35 * c
        for c in range(K):
            ys, xs = np.nonzero(lab == c)
            if len(ys) == 0:
                continue
            j = rng.integers(len(ys))
            y0, x0 = int(ys[j]), int(xs[j])
            for t in range(int(0.08 * S)):                           # a short horizontal stroke
                x = min(S - 1, x0 + t)
                if lab[y0, x] == c:
                    scb[y0, x] = c
        return self._sample(img, lab, scb)
