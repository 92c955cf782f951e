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

from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset


def default_args(**over) -> SimpleNamespace:
    """The argparse defaults of train_chaos.py:23-179 that the hot path reads (bench.py, smoke, tests and the CPU oracle all
    build their flag namespace here)."""
    a = dict(
        input_ch=1, init_ch=32, max_ch=512, num_classes=5, output_stride=8,
        ignored_index=5, epoch=400, lr=1e-4, wd=3e-4, lr_decay='poly',
        do_loss_ent=False, loss_ent_weight=1.0, ramp_up_loss_ent=True, ramp_up_scale=8.0,
        do_decoder_consistency=False, ramp_up_loss_cr=True, detach_weak_cr=False,
        loss_cr_variants='ce_loss', loss_cr_weight=1.0,
        do_aux_path=False, feat_stage=['encoder/stage6', 'encoder/stage5'], feat_ch=[512, 512],
        loss_aux_weight=0.01, hid_ch=64, aux_drop_prob=0.0,
        do_memory=False, loss_memory_weight=1.0, update_momentum=0.9,
        ensemble_mode='cosine_similarity',
    )
    a.update(over)
    return SimpleNamespace(**a)


def full_flags(**over) -> SimpleNamespace:
    """README.md:63 'Experiment' flags: ent + decoder consistency + aux path + memory."""
    d = dict(do_loss_ent=True, do_decoder_consistency=True, do_aux_path=True, do_memory=True)
    d.update(over)
    return default_args(**d)


def synthetic_batch(B: int, H: int, W: int, num_classes: int = 5, seed: int = 0, keep: float = 0.02):
    """The synthetic batch of SURVEY.md 8(d) (bench.py, smoke, parity tests, the oracle's cpu_baseline): image ~ N(0, 1),
    strong view = per-sample a x + b, blob labels, ~2 % of the pixels keep their label as scribble, all-ones valid mask."""
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 1, H, W, generator=g)
    a = torch.rand(B, 1, 1, 1, generator=g) * 1.6 + 0.2
    b = torch.rand(B, 1, 1, 1, generator=g) * 1.6 - 0.8
    image_strong = image * a + b
    coarse = torch.randint(0, num_classes, (B, 1, max(H // 16, 1), max(W // 16, 1)), generator=g).float()
    label = F.interpolate(coarse, size=(H, W), mode='nearest').long().squeeze(1)
    kept = torch.rand(B, H, W, generator=g) < keep
    scb = torch.where(kept, label, torch.full_like(label, num_classes))
    scribble = F.one_hot(scb, num_classes + 1).permute(0, 3, 1, 2).float().contiguous()
    label_1h = F.one_hot(label, num_classes).permute(0, 3, 1, 2).float().contiguous()
    return dict(image=image, image_strong=image_strong, scribble=scribble,
                valid_mask=torch.ones(B, 1, H, W), label=label_1h)


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
    def __init__(self, file_ls, num_classes, size=256, do_strong=False, strength=1.0, train=True, seed=1, raw=False,
                 native=False, compact=False):
        self.files, self.K, self.size = list(file_ls), num_classes, size
        self.do_strong, self.strength, self.train = do_strong, strength, train
        self.seed, self.epoch, self._index = int(seed), 0, 0
        self.rng = None             # a caller may pin the strong-view generator (tests/studies/dice_study.py does, per sample)
        self.raw = raw              # un-augmented {'img', 'lab', 'scb'} arrays for augment.DeviceAugmenter (collate_raw)
        # evaluation as the reference does it (train_chaos.py:235-241, inference.py:125-133: base_transforms = [MeanStdNorm()]
        # and nothing else): the slice at its NATIVE size, never cropped or padded -- every pixel is scored
        self.native = bool(native) and not train
        # evaluation items with the two label maps as uint8 class indices ('label_idx', 'scribble_idx') instead of one-hot fp32
        # planes: 6 bytes per pixel through the loader's shared memory, its pinned copy and the upload instead of 48 (K = 5) --
        # the drivers expand them on the device (expand_compact).  The validation loop was bound by exactly that copy (round 5).
        self.compact = bool(compact) and self.native

    def __len__(self):
        return len(self.files)

    def _sample(self, img, lab, scb):
        if self.raw:
            return {'img': img.astype(np.float32), 'lab': lab.astype(np.int32), 'scb': scb.astype(np.int32)}
        img = img.astype(np.float32)
        img = (img - img.mean()) / (img.std() + 1e-8)                 # MeanStdNorm
        if self.compact:
            return {'image': torch.from_numpy(img[None]), 'label_idx': torch.from_numpy(lab.astype(np.uint8)),
                    'scribble_idx': torch.from_numpy(scb.astype(np.uint8))}
        if self.native:
            return {'image': torch.from_numpy(img[None]), 'label': torch.from_numpy(_one_hot(lab.astype(np.int64), self.K)),
                    'scribble': torch.from_numpy(_one_hot(scb.astype(np.int64), self.K + 1))}
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

    # Mixup's second image comes from the WHOLE file list (datasets/augmentations.py:66 `np.load(np.random.choice(file_ls))`,
    # handed in by CHAOSTwoStream.__getitem__, chaos_dataset.py:73-74): with `mix_partner = True` a raw item carries one
    # candidate partner slice, drawn here, which the device pipeline uses when its own draw lets Mixup fire for the sample
    mix_partner = False

    def __getitem__(self, i):
        self._index = int(i)
        z = np.load(self.files[i])
        d = self._sample(z['img'], z['lab'], z['scb'])
        if self.raw and self.mix_partner:
            rng = np.random.default_rng([self.seed, self.epoch, self._index, 77])
            d['mix'] = np.load(self.files[int(rng.integers(len(self.files)))])['img'].astype(np.float32)
        return d


# The reference ships one dataset class per data set (datasets/{chaos,acdc,lvsc}/*_dataset.py: CHAOSDataset / CHAOSTwoStream,
# ACDCDataset / ACDCTwoStream, LVSCDataset / LVSCTwoStream).  They read the same .npz layout (uid / img / lab / scb) and differ
# only in their `classnames` table and class count; the two-stream transform lists live in augment.py (device) here.
class CHAOSDataset(NpzSlices):
    """datasets/chaos/chaos_dataset.py:17-41"""
    classnames = {0: 'background', 1: 'liver', 2: 'right kidney', 3: 'left kidney', 4: 'spleen', 5: 'unknown'}
    num_classes, ignored_index, input_size = 5, 5, (256, 256)          # chaos_aug_configs.py:9-11


class ACDCDataset(NpzSlices):
    """datasets/acdc/acdc_dataset.py:12-36"""
    classnames = {0: 'background', 1: 'right ventricle', 2: 'myocardium', 3: 'left ventricle', 4: 'unknown'}
    num_classes, ignored_index, input_size = 4, 4, (224, 224)          # acdc_aug_configs.py:9-11


class LVSCDataset(NpzSlices):
    """datasets/lvsc/lvsc_dataset.py:16-40"""
    classnames = {0: 'background', 1: 'myo', 2: 'unknown'}
    num_classes, ignored_index, input_size = 2, 2, (224, 224)          # lvsc_aug_configs.py:9-13


DATASET_CLASSES = {'chaos': CHAOSDataset, 'acdc': ACDCDataset, 'lvsc': LVSCDataset}


def dataset_class(name: str):
    """The reader class of a data set name (--dataset); unknown names get the generic reader."""
    return DATASET_CLASSES.get(name, NpzSlices)


_LOADER_CTX = None


def loader_context(num_workers: int):
    """``multiprocessing_context`` for the DataLoaders of the drivers (round 6): worker processes come from a FORK SERVER that has
    imported torch and this package but never touched the GPU.  torch's default on Linux forks the training process itself --
    with a HIP context and tens of GB of device mappings in it that took 15 - 30 s per worker on the MI355X box (the driver test
    with two loaders of two workers spent 90 - 150 s of the GPU suite's wall clock there) and is not a supported thing to do to a
    process with an initialised GPU runtime.  The workers only build CPU tensors; pinning and uploads stay in the main process."""
    global _LOADER_CTX
    if num_workers <= 0:
        return None
    if _LOADER_CTX is None:
        import multiprocessing as mp
        ctx = mp.get_context('forkserver')
        ctx.set_forkserver_preload(['numpy', 'torch', 'pacingpseudo_amd.data', 'pacingpseudo_amd.augment'])
        _LOADER_CTX = ctx
    return _LOADER_CTX


def collate_by_shape(items):
    """Evaluation collate for native-size slices: the reference's default collate needs equal sizes in a batch (and its
    inference driver runs batch size 1, writing one row per slice in file-list order, inference.py:159-190); here a batch is
    cut at every change of shape into RUNS of consecutive same-shape slices, so the rows a caller appends group by group
    stay in data-set order (round 3 grouped by shape across the whole batch and permuted them).  Returns a LIST of batch
    dictionaries."""
    runs = []
    for it in items:
        if runs and tuple(runs[-1][0]['image'].shape) == tuple(it['image'].shape):
            runs[-1].append(it)
        else:
            runs.append([it])
    return [{k: torch.stack([it[k] for it in g]) for k in g[0]} for g in runs]


def expand_compact(batch, num_classes, device):
    """An evaluation batch on the device.  Items of a `compact` data set carry uint8 class maps: they are uploaded as they are and
    expanded to the one-hot fp32 planes the model and the Dice meters take (pp_aug_onehot: the same values `_one_hot` writes on the
    host); other batches are uploaded unchanged."""
    if 'label_idx' not in batch:
        return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
    from ._lib import lib, stream_ptr
    K = int(num_classes)
    img = batch['image'].to(device, non_blocking=True)
    out = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()
           if k not in ('image', 'label_idx', 'scribble_idx')}
    out['image'] = img
    st = stream_ptr()
    for key, name, planes in (('label_idx', 'label', K), ('scribble_idx', 'scribble', K + 1)):
        idx = batch[key].to(device, non_blocking=True).to(torch.int32).contiguous()
        B, H, W = idx.shape
        hot = torch.empty((B, planes, H, W), device=device, dtype=torch.float32)
        lib.pp_aug_onehot(idx.data_ptr(), hot.data_ptr(), B, planes, H * W, st)
        out[name] = hot
    return out


class SyntheticPhantoms(NpzSlices):
    """`n` deterministic slices: K-1 ellipses on noise; scribbles = a short stroke inside each structure."""

    def __init__(self, n, num_classes, size=256, do_strong=False, strength=1.0, train=True, seed=1, raw=False, native=False,
                 compact=False):
        super().__init__([None] * n, num_classes, size, do_strong, strength, train, seed, raw, native, compact)
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
