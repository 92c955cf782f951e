"""Fully-supervised upper bound: the reference's ``upper_bound_chaos.py`` (bare UNet trained with cross entropy on the
FULL label + soft Dice loss, upper_bound_chaos.py:110-244) on the MI355X path.

Kept from the reference: flag names / types / defaults (upper_bound_chaos.py:23-108), the run-directory layout
``{root}/{modality}/{session}/{session}-{time}-fold{k}-{tag}/`` with ``ckps/``, ``log.txt``, ``valdice.npz``, the log
lines (:173-174, :213-218, :241-243), per-epoch poly LR, ``model.eval()`` after the first epoch and never back (:180),
``ckp_{epoch}.pth`` / ``best_ckp.pth`` holding the bare UNet's ``state_dict()``.
The step runs through ``UNet.forward`` (one autograd node over the engine's static plan) and the HIP loss kernels
``partial_cross_entropy_loss`` / ``dice_loss_fn``; the training set goes through the reference's WEAK augmentation list
(upper_bound_chaos.py:132-137) on the GPU (``augment.DeviceAugmenter(do_strong=False)``; ``--cpu_input`` selects the minimal CPU
path); validation at the native slice size with device-side meters; widened ``choices`` and the --synthetic / --image_size /
--max_iters additions are those of ``pacingpseudo_amd.train``.
"""
from __future__ import annotations

import argparse
import logging
import os
import random
import shutil
import sys
import time

import numpy as np
import torch

parser = argparse.ArgumentParser()
parser.add_argument('--gpu', type=str, default='1')
parser.add_argument('--seed', type=int, default=1)
parser.add_argument('--dataset', type=str, default='chaos')
parser.add_argument('--root', type=str, default='./outputs/chaos')
parser.add_argument('--session', type=str, default='Upperbound')
parser.add_argument('--tag', type=str, required=True)
parser.add_argument('--fold', type=int, default=1, choices=[0, 1, 2, 3, 4])
parser.add_argument('--modality', type=str, default='t1', choices=['t1', 't2'])
parser.add_argument('--num_classes', type=int, default=None, help='default: the --dataset preset (5 for chaos)')
parser.add_argument('--num_workers', type=int, default=4)
parser.add_argument('--augmentation_configs', type=str, default='datasets.chaos.chaos_aug_configs')
parser.add_argument('--augmentations', type=str, default='TransformsColor', choices=['TransformsColor'])
parser.add_argument('--input_ch', type=int, default=1)
parser.add_argument('--init_ch', type=int, default=32)
parser.add_argument('--max_ch', type=int, default=512)
parser.add_argument('--output_stride', type=int, default=8, choices=[32, 16, 8])
parser.add_argument('--is_stride_conv', type=bool, default=False)
parser.add_argument('--is_trans_conv', type=bool, default=False)
parser.add_argument('--elab_end_points', type=bool, default=True)
parser.add_argument('--loss_dice', action='store_true', default=True)
parser.add_argument('--ignored_index', type=int, default=None, help='default: the --dataset preset (5 for chaos)')
parser.add_argument('--epoch', type=int, default=400)
parser.add_argument('--batch_size', type=int, default=12)
parser.add_argument('--optimizer', type=str, default='adam', choices=['adam'])
parser.add_argument('--momentum', type=float, default=0.9)
parser.add_argument('--lr', type=float, default=0.0001)
parser.add_argument('--lr_decay', type=str, default='poly', choices=['linear', 'poly', 'cosine'])
parser.add_argument('--wd', type=float, default=0.0003)
parser.add_argument('--ckp_interval', type=int, default=10000)
# ---- additions of this implementation (same meaning as in pacingpseudo_amd.train)
parser.add_argument('--synthetic', type=int, default=0)
parser.add_argument('--image_size', type=int, default=None, help='training crop size (default: the --dataset preset); validation runs at the native size')
parser.add_argument('--max_iters', type=int, default=0)
parser.add_argument('--cpu_input', action='store_true',
                    help='minimal CPU input path of data.py (MeanStdNorm + centre crop only) instead of the weak augmentation '
                         'pipeline on the GPU')
parser.add_argument('--gpu_augment', action='store_true', help='accepted for compatibility: the GPU pipeline is the default')


def train_interface(args):
    from .augment import AugConfig, DeviceAugmenter, collate_raw
    from .data import SyntheticPhantoms, collate_by_shape, dataset_class, expand_compact, loader_context
    from .losses.losses import dice_loss_fn, partial_cross_entropy_loss
    from .models import UNet
    from .optim import FusedAdam
    from .train import _class_names
    from .utils import cosine_lr_decay, linear_lr_decay, poly_lr_decay
    from .utils.metrics import ValAccumulator
    from .utils.scalars import ScalarLog

    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(device)
    best_avg, best_epoch, best_avg_class = 0, 0, []
    model = UNet(input_ch=args.input_ch, init_ch=args.init_ch, max_ch=args.max_ch, num_classes=args.num_classes,
                 output_stride=args.output_stride, is_stride_conv=args.is_stride_conv, is_trans_conv=args.is_trans_conv,
                 elab_end_points=args.elab_end_points).cuda()
    logging.info(model)
    optimizer = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
    ds_kw = dict(num_classes=args.num_classes, size=args.image_size, seed=args.seed)
    # the reference trains the upper bound with the whole WEAK pipeline (upper_bound_chaos.py:132-137: base_transforms =
    # Scaling, Elastic, Rotation, Mirroring, GaussianNoise, RandomCrop; no strong view): the same device pipeline as
    # train_chaos.py with do_strong=False, fed by raw slices
    gpu_aug = not args.cpu_input
    augmenter = DeviceAugmenter(AugConfig(num_classes=args.num_classes, crop_size=(args.image_size, args.image_size),
                                          do_strong=False), device=device, seed=args.seed) if gpu_aug else None
    if args.synthetic:
        train_dataset = SyntheticPhantoms(args.synthetic, do_strong=False, train=True, raw=gpu_aug, **ds_kw)
        val_dataset = SyntheticPhantoms(max(args.synthetic // 4, 1), train=False, native=True, compact=True, **ds_kw)
    else:
        train_dataset = dataset_class(args.dataset)(args.train_ls, do_strong=False, train=True, raw=gpu_aug, **ds_kw)
        val_dataset = dataset_class(args.dataset)(args.val_ls, train=False, native=True, compact=True, **ds_kw)
    train_loader = torch.utils.data.DataLoader(train_dataset, batch_size=args.batch_size, shuffle=True,
                                               num_workers=args.num_workers, drop_last=True,
                                               collate_fn=collate_raw if gpu_aug else None,
                                               persistent_workers=bool(gpu_aug and args.num_workers > 0), pin_memory=gpu_aug,
                                               multiprocessing_context=loader_context(args.num_workers))
    val_loader = torch.utils.data.DataLoader(val_dataset, batch_size=args.batch_size, shuffle=False,
                                             num_workers=args.num_workers, drop_last=False, collate_fn=collate_by_shape,
                                             persistent_workers=args.num_workers > 0, pin_memory=True,
                                             multiprocessing_context=loader_context(args.num_workers))
    names = _class_names(args.num_classes, args.dataset)
    scalars = ScalarLog(os.path.join(args.child, 'tb_summary', 'scalars.jsonl'))
    decay = {'poly': poly_lr_decay, 'cosine': cosine_lr_decay, 'linear': linear_lr_decay}
    if args.lr_decay not in decay:
        raise ValueError('Unimplemented learning rate decay policy.')
    valdice = np.zeros(args.epoch)
    aug_stream = torch.cuda.Stream() if (augmenter is not None and os.environ.get('PP_AUG_STREAM', '1') != '0') else None
    for curr_epoch in range(args.epoch):
        epoch_tic = time.time()
        optimizer, new_lr = decay[args.lr_decay](optimizer, curr_epoch, args.epoch, args.lr)
        acc = torch.zeros(3, device=device, dtype=torch.float64)        # sum ce*n, sum dice*n, n (read once per epoch)
        for idx, batch in enumerate(train_loader):
            if args.max_iters and idx >= args.max_iters:
                break
            if augmenter is not None:
                batch = augmenter.ahead(aug_stream, batch['img'], batch['lab'], batch['scb'], batch['sizes'])   # beside the previous step (train.py)
            image, label = batch['image'].to(device, non_blocking=True), batch['label'].to(device, non_blocking=True)
            n = image.shape[0]
            logits = model(image)['segmentation/logits']
            target = torch.argmax(label, dim=1).long()
            loss_ce = partial_cross_entropy_loss(logits, target, args.ignored_index)
            loss = loss_ce
            acc[0] += loss_ce.detach() * n
            if args.loss_dice:
                loss_dice = dice_loss_fn(logits, label)
                loss = loss + loss_dice
                acc[1] += loss_dice.detach() * n
            acc[2] += n
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        a = acc.cpu().numpy()
        cnt = max(a[2], 1)
        logging.info("epoch: {:03d}, lr: {:.6f}, loss_ce: {:.6f}, loss_dice: {:.6f}, {:.2f} s/epoch".format(
            curr_epoch, new_lr, a[0] / cnt, a[1] / cnt, time.time() - epoch_tic))

        model.eval()                                   # upper_bound_chaos.py:180, never undone
        tic = time.time()
        meters = ValAccumulator(args.num_classes, device)              # Dice meters + n-weighted loss_ce, on the device
        dsum = torch.zeros(1, device=device, dtype=torch.float64)       # n-weighted loss_dice
        for groups in val_loader:
            for batch in groups:
                batch = expand_compact(batch, args.num_classes, device)      # uint8 class maps -> one-hot planes, on the device
                image, label = batch['image'], batch['label']
                with torch.no_grad():
                    logits = model(image)['segmentation/logits']
                    target = torch.argmax(label, dim=1).long()
                    meters.update(logits, label, partial_cross_entropy_loss(logits, target, args.ignored_index))
                    dsum += dice_loss_fn(logits, label).double() * image.shape[0]
        dsc, val_ce, n_val = meters.result()                            # the host sync of the validation epoch
        val_dice = float(dsum) / max(n_val, 1)
        avg_all = np.mean([dsc[_] for _ in range(1, args.num_classes)])
        logging.info("val: {:03d}, loss_ce: {:.6f}, loss_dice: {:.6f}, {:.2f} s/epoch".format(
            curr_epoch, val_ce, val_dice, time.time() - tic))
        logging.info("[" + ", ".join("{}: {:.4f}".format(nm, dsc[i]) for i, nm in enumerate(names))
                     + ", All: {:.4f}]".format(avg_all))
        # scalar tags of upper_bound_chaos.py:176-178, :216-224
        scalars.add('losses/loss_ce_train', a[0] / cnt, curr_epoch)
        scalars.add('losses/loss_dice_train', a[1] / cnt, curr_epoch)
        scalars.add('lr/current_lr', new_lr, curr_epoch)
        scalars.add('losses/loss_ce_val', val_ce, curr_epoch)
        scalars.add('losses/loss_dice_val', val_dice, curr_epoch)
        for i, nm in enumerate(names):
            scalars.add(f'DSC/{nm}', dsc[i], curr_epoch)
        scalars.add('DSC/All', avg_all, curr_epoch)
        scalars.add('DSC/Best', max(best_avg, avg_all), curr_epoch)
        valdice[curr_epoch] = avg_all
        if curr_epoch + 1 == args.epoch or (curr_epoch + 1) % args.ckp_interval == 0:
            torch.save(model.state_dict(), os.path.join(args.child, 'ckps', 'ckp_{:d}.pth'.format(curr_epoch)))
        if avg_all > best_avg:
            best_epoch, best_avg = curr_epoch, avg_all
            best_avg_class = [dsc[_] for _ in range(1, args.num_classes)]
            torch.save(model.state_dict(), args.child + '/best_ckp.pth')
    logging.info("The best at epoch: {:d}, ".format(best_epoch)
                 + ", ".join("{}: {:.4f}".format(nm, v) for nm, v in zip(names[1:], best_avg_class))
                 + ", All: {:.4f}".format(best_avg))
    np.savez(os.path.join(args.child, 'valdice'), valdice=valdice)
    return valdice


def train_main(argv=None):
    from .train import DATASETS, apply_dataset_preset, split_dir
    args = apply_dataset_preset(parser.parse_args(argv))
    if 'LOCAL_RANK' not in os.environ:
        os.environ['CUDA_VISIBLE_DEVICES'] = args.gpu
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    sub = DATASETS.get(args.dataset, DATASETS['chaos'])['split_subdir'].format(modality=args.modality)
    args.child = os.path.join(os.path.join(args.root, sub) if sub else args.root, args.session,
                              f'{args.session}-{time.strftime("%H-%M-%S-%m%d")}-fold{args.fold}-{args.tag}')
    os.makedirs(args.child, exist_ok=False)
    os.makedirs(os.path.join(args.child, 'ckps'), exist_ok=True)
    os.makedirs(os.path.join(args.child, 'tb_summary'), exist_ok=True)
    if os.path.isfile(sys.argv[0]):
        shutil.copy(sys.argv[0], os.path.join(args.child, os.path.basename(sys.argv[0])))
    log = logging.getLogger()
    log.setLevel(logging.INFO)
    fh = logging.FileHandler(args.child + "/log.txt")
    fh.setFormatter(logging.Formatter('[%(asctime)s.%(msecs)03d] %(message)s', datefmt='%H:%M:%S'))
    log.addHandler(fh)
    log.addHandler(logging.StreamHandler(sys.stdout))
    logging.info(''.join(f'{k}={v}\n' for k, v in args._get_kwargs()))
    if not args.synthetic:
        data_root, base = split_dir(args.dataset, args.modality)
        with open(f'{base}/train_fold{args.fold}.txt', 'r') as f:
            args.train_ls = [(data_root + '/' + p).rstrip('\n') for p in f.readlines()]
        with open(f'{base}/test_fold{args.fold}.txt', 'r') as f:
            args.val_ls = [(data_root + '/' + p).rstrip('\n') for p in f.readlines()]
    return train_interface(args)


if __name__ == '__main__':
    train_main()
