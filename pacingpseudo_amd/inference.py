"""Post-training evaluation on the GPU: the reference's ``inference.py`` (``main_interface`` :97-194, ``main`` :256-320).

What it does, as the reference: build a bare ``UNet``, load a checkpoint written by either trainer (a ``ConsistencyRegulr``
checkpoint is reduced to its ``backbone.`` entries, :138-146), run the test fold in eval mode, report per-slice / per-class
Dice (:196-215) and 95 % Hausdorff distance in millimetres (:217-237, ``medpy.metric.hd95`` with the data set's pixel
spacing), write ``eval_data.npz`` (``dicearr``, ``hd95arr``: slices x classes, NaN = class absent) and the summary line.

How it differs: slices are evaluated in batches -- soft-max / arg-max, the Dice counts (``pp_dice_counts``) and both
directed surface-distance sets of HD95 (``pp_hd95_surface_distances``) are computed on the GPU for the whole batch, the
per-slice ``.cpu().numpy()`` round trips and medpy's scipy distance transforms are gone.  Like the reference, every slice
is fed at its NATIVE size with MeanStdNorm only (:125-133) -- never cropped or padded, so every pixel is scored; a loader
batch of differently sized slices is split into same-shape groups, and a slice whose size is not a multiple of the encoder
stride is an error here as it is in the reference (its skip concatenation fails).  ``--synthetic N`` evaluates phantom
slices when no data set is on disk."""
from __future__ import annotations

import argparse
import logging
import os
import random
import shutil
import sys
from collections import OrderedDict

import numpy as np
import torch

SPACING = {'acdc': (1.51, 1.51), 'chaost1': (1.62, 1.62), 'chaost2': (1.62, 1.62), 'lvsc': (1.48, 1.48)}   # mm, inference.py:52-57
CLASSES = {'acdc': 4, 'chaost1': 5, 'chaost2': 5, 'lvsc': 2}                                                # :59-64
CROP = {'acdc': 224, 'chaost1': 256, 'chaost2': 256, 'lvsc': 224}

parser = argparse.ArgumentParser(description='evaluate a checkpoint: Dice + HD95 per slice and class (flags of the reference inference.py)')
parser.add_argument('--gpu', type=str, default='1')
parser.add_argument('--seed', type=int, default=1)
parser.add_argument('--root', type=str, default='./outputs', help='outputs go to <root>/<session>/<dataset>/<checkpoint name>/')
parser.add_argument('--session', type=str, default='Inference')
parser.add_argument('--fold', type=int, required=True, help='test fold; must appear as "fold<k>" in --checkpoint_file')
parser.add_argument('--checkpoint_file', type=str, required=True,
                    help='run directory of a training session (ckps/ckp_399.pth, or ckp_39.pth for lvsc, is taken) or a .pth file')
parser.add_argument('--best_ckp', action='store_true', default=False, help='take best_ckp.pth of the run directory instead')
parser.add_argument('--dataset', type=str, default='acdc', choices=['acdc', 'chaost1', 'chaost2', 'lvsc'])
parser.add_argument('--num_workers', type=int, default=4)
parser.add_argument('--batch_size', type=int, default=1, help='slices per forward pass (metrics stay per slice)')
parser.add_argument('--input_ch', type=int, default=1)
parser.add_argument('--init_ch', type=int, default=32)
parser.add_argument('--max_ch', type=int, default=512)
parser.add_argument('--output_stride', type=int, default=8, choices=[32, 16, 8])
parser.add_argument('--is_stride_conv', type=bool, default=False)
parser.add_argument('--is_trans_conv', type=bool, default=False)
parser.add_argument('--elab_end_points', type=bool, default=False)
# ---- additions of this implementation
parser.add_argument('--image_size', type=int, default=0, help='size of the --synthetic phantoms (0 = the data set\'s training crop); real slices are evaluated at their native size')
parser.add_argument('--synthetic', type=int, default=0, help='evaluate N phantom slices instead of ./data')


def load_backbone(model, state_dict):
    """inference.py:138-146: a full-model checkpoint is reduced to its `backbone.` entries."""
    try:
        model.load_state_dict(state_dict)
    except RuntimeError:
        stripped = OrderedDict((k.partition('.')[-1], v) for k, v in state_dict.items() if 'backbone' in k)
        model.load_state_dict(stripped)
    return model


def evaluate(model, loader, num_classes, spacing, device):
    """-> (dicearr, hd95arr), both (slices, classes) float32 with NaN where the reference skips a class."""
    from .data import expand_compact
    from .utils.metrics import batch_dice_counts, batch_hd95
    dice_rows, hd_rows = [], []
    model.eval()
    for groups in loader:
        for batch in (groups if isinstance(groups, list) else [groups]):   # same-shape groups (data.collate_by_shape)
            batch = expand_compact(batch, num_classes, device)          # uint8 class maps -> one-hot planes, on the device
            image, label = batch['image'], batch['label']
            with torch.no_grad():
                logits = model(image)['segmentation/logits']
            c = batch_dice_counts(logits, label)                           # |P & T|, |P|, |T| per (slice, class), one launch
            inter, ps, ts = c[..., 0], c[..., 1], c[..., 2]
            with np.errstate(invalid='ignore', divide='ignore'):
                dice = 2.0 * inter / np.maximum(ps + ts, 1e-8)             # inference.py:211-213 (no smoothing term here)
            dice[(ps == 0) & (ts == 0)] = np.nan                           # :208-209
            dice_rows.extend(dice.tolist())
            hd_rows.extend(batch_hd95(logits.argmax(1), label.argmax(1), num_classes, spacing).tolist())
    return np.array(dice_rows, np.float32), np.array(hd_rows, np.float32)


def main_interface(args):
    from .data import NpzSlices, SyntheticPhantoms, collate_by_shape, loader_context
    from .models import UNet
    from .utils import AvgMeter
    num_classes, spacing = CLASSES[args.dataset], SPACING[args.dataset]
    size = args.image_size or CROP[args.dataset]          # only the size of --synthetic phantoms; real slices keep theirs
    logging.info(f'Number of classes: {num_classes}')
    logging.info(f'Spacing: {spacing}')
    device = torch.device('cuda', 0)
    model = UNet(input_ch=args.input_ch, init_ch=args.init_ch, max_ch=args.max_ch, num_classes=num_classes,
                 output_stride=args.output_stride, is_stride_conv=args.is_stride_conv, is_trans_conv=args.is_trans_conv,
                 elab_end_points=args.elab_end_points).to(device)
    if args.synthetic:
        test_dataset = SyntheticPhantoms(args.synthetic, num_classes, size=size, train=False, seed=args.seed, native=True, compact=True)
    else:
        test_dataset = NpzSlices(args.test_ls, num_classes, size=size, train=False, seed=args.seed, native=True, compact=True)
    loader = torch.utils.data.DataLoader(test_dataset, batch_size=args.batch_size, shuffle=False,
                                         num_workers=args.num_workers, drop_last=False, collate_fn=collate_by_shape,
                                         multiprocessing_context=loader_context(args.num_workers))
    logging.info('Length {}'.format(len(loader)))
    load_backbone(model, torch.load(args.checkpoint_file, map_location=device))
    dicearr, hd95arr = evaluate(model, loader, num_classes, spacing, device)
    np.savez(os.path.join(args.child, 'eval_data'), dicearr=dicearr, hd95arr=hd95arr)
    meter_dice = [AvgMeter() for _ in range(num_classes)]
    meter_hd95 = [AvgMeter() for _ in range(num_classes)]
    for drow, hrow in zip(dicearr, hd95arr):
        for cls in range(num_classes):
            if not np.isnan(drow[cls]):
                meter_dice[cls].update(float(drow[cls]))
            if not np.isnan(hrow[cls]):
                meter_hd95[cls].update(float(hrow[cls]))
    logging.info('Dataset: {}'.format(args.dataset))
    logging.info('Number of clases: {}'.format(num_classes))
    foldavgdice = np.mean([meter_dice[_].avg for _ in range(1, num_classes)])
    foldavghd95 = np.mean([meter_hd95[_].avg for _ in range(1, num_classes)])
    logging.info('Fold {}, overall Dice: {:.4f}, overall HD95: {:.2f}'.format(args.fold, foldavgdice, foldavghd95))
    logging.info('Shape of the Dice array: {}'.format(dicearr.shape))
    logging.info('Shape of the HD95 array: {}'.format(hd95arr.shape))
    return dicearr, hd95arr


def main(argv=None):
    args = parser.parse_args(argv)
    if 'LOCAL_RANK' not in os.environ:
        os.environ['CUDA_VISIBLE_DEVICES'] = args.gpu
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    assert f'fold{args.fold}' in args.checkpoint_file, 'the checkpoint must come from the same fold (inference.py:264)'
    args.child = os.path.join(args.root, args.session, args.dataset, os.path.basename(args.checkpoint_file.rstrip('/')))
    os.makedirs(args.child, exist_ok=True)
    if os.path.isdir(args.checkpoint_file):                    # a run directory: pick the file the reference picks (:274-284)
        run = args.checkpoint_file
        if args.best_ckp:
            cand = [os.path.join(run, 'ckps', 'best_ckp.pth'), os.path.join(run, 'best_ckp.pth')]
        else:
            cand = [os.path.join(run, 'ckps', 'ckp_39.pth' if args.dataset == 'lvsc' else 'ckp_399.pth')]
        found = [c for c in cand if os.path.isfile(c)]
        if not found:
            raise FileNotFoundError(f'none of {cand} exists')
        args.checkpoint_file = found[0]
    if os.path.isfile(sys.argv[0]):
        shutil.copy(sys.argv[0], os.path.join(args.child, os.path.basename(sys.argv[0])))
    log = logging.getLogger()
    log.setLevel(logging.INFO)
    for h in list(log.handlers):
        log.removeHandler(h)
    fh = logging.FileHandler(args.child + '/log.txt', mode='w')
    fh.setFormatter(logging.Formatter('[%(asctime)s.%(msecs)03d] %(message)s', datefmt='%H:%M:%S'))
    log.addHandler(fh)
    log.addHandler(logging.StreamHandler(sys.stdout))
    logging.info(''.join(f'{k}={v}\n' for k, v in args._get_kwargs()))
    if not args.synthetic:
        from .train import split_dir                              # the same table the trainers read their fold lists from
        ds, mod = {'acdc': ('acdc', ''), 'lvsc': ('lvsc', ''), 'chaost1': ('chaos', 't1'), 'chaost2': ('chaos', 't2')}[args.dataset]
        data_root, base = split_dir(ds, mod)
        with open(f'{base}/test_fold{args.fold}.txt', 'r') as f:
            args.test_ls = [(data_root + '/' + p).rstrip('\n') for p in f.readlines()]
    return main_interface(args)


if __name__ == '__main__':
    main()
