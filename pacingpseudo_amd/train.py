"""Training driver with the flag surface, run-directory layout, log lines and checkpoint files of the reference's
``train_chaos.py`` (train_chaos.py:23-179 flags, :431-463 run dir, :318,:396-399,:426 log lines, :405-413 checkpoints),
running the step on MI355X through ``pacingpseudo_amd``.

Kept from the reference: every flag name / type / default; ``{root}/{modality}/{session}/{session}-{time}-fold{k}-{tag}/``
with ``ckps/``, ``log.txt`` and ``valdice.npz``; per-epoch poly LR (utils.poly_lr_decay); the loss assembly of
train_chaos.py:273-310; ``model.eval()`` after the first epoch's validation and never back (train_chaos.py:370);
``ckps/ckp_{last}.pth`` + ``best_ckp.pth`` holding ``model.state_dict()`` (165 entries, reference key layout).

Changed on purpose: ``choices`` of --batch_size / --epoch / --lr / --wd / --init_ch / --max_ch are widened (the
reference rejects the benchmark's batch 32, train_chaos.py:93); loss meters stay on the GPU and are read once per
epoch instead of five ``.item()`` syncs per iteration; validation runs at the native slice size with its meters on the
GPU (one host sync per epoch, slices sharded over the ranks); the TensorBoard scalar tags (train_chaos.py:362-367,
:416-423) go to ``tb_summary/scalars.jsonl`` (tensorboard is not installed); its figures are dropped.
New flags: --synthetic N (phantom slices when no dataset is on disk), --image_size, --max_iters, and the
data-parallel launch is picked up from torch.distributed.run's environment (one process per GPU, RCCL).
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

parser = argparse.ArgumentParser(description='PacingPseudo training on MI355X (flag surface of the reference driver)')


def _flags(title, ref, table):
    """One argparse group per block of the reference's flag list; `ref` = its line range in train_chaos.py.
    Names, types, defaults and choices are the reference's (tests/golden/flags.json); the help texts say what the
    flag selects in THIS implementation."""
    grp = parser.add_argument_group(f'{title} (train_chaos.py:{ref})')
    for name, kw in table:
        grp.add_argument('--' + name, **kw)


_on = dict(action='store_true')
_flags('run', '26-41', [
    ('gpu', dict(type=str, default='0', help='CUDA_VISIBLE_DEVICES of a single-process run (ignored under torch.distributed.run)')),
    ('seed', dict(type=int, default=1, help='python / numpy / torch seed; with the same seed the initial weights equal the reference\'s')),
    ('dataset', dict(type=str, default='chaos', help='chaos | acdc | lvsc: preset of class count, ignored index, crop size, class names')),
    ('root', dict(type=str, default='./outputs/chaos', help='run directories are created below <root>/<modality>/<session>/')),
    ('session', dict(type=str, default='Control', choices=['Control', 'Experiment'], help='first level of the run directory')),
    ('tag', dict(type=str, required=True, help='suffix of the run directory')),
])
_flags('data', '43-61', [
    ('fold', dict(type=int, default=1, choices=[0, 1, 2, 3, 4], help='which five-fold split files to read')),
    ('modality', dict(type=str, default='t1', choices=['t1', 't2'], help='sub-directory of the split files and of the outputs')),
    ('num_classes', dict(type=int, default=None, help='segmentation classes incl. background, excl. the ignored label (default: the --dataset preset, 5 for chaos)')),
    ('num_workers', dict(type=int, default=4, help='DataLoader worker processes')),
    ('augmentation_configs', dict(type=str, default='datasets.chaos.chaos_aug_configs',
                                  help='kept for compatibility; the transform list of all three data sets is built into augment.py')),
    ('augmentations', dict(type=str, default='TransformsColor',
                           choices=['TransformsColor', 'TransformsColorBlur', 'TransformsColorMixup', 'TransformsColorLow'],
                           help='strong-view recipe (colour transforms, + GaussianBlur / Mixup / SimulationLowRes)')),
])
_flags('network', '63-84', [
    ('input_ch', dict(type=int, default=1, help='image channels')),
    ('init_ch', dict(type=int, default=32, help='width of the first encoder stage; doubles per stage')),
    ('max_ch', dict(type=int, default=512, help='cap of the stage width')),
    ('output_stride', dict(type=int, default=8, choices=[32, 16, 8], help='encoder stride; deeper stages dilate instead of pooling')),
    ('is_stride_conv', dict(type=bool, default=False, help='stride-2 first convolution instead of max-pooling (models/unet.py:113; needs --is_trans_conv too)')),
    ('is_trans_conv', dict(type=bool, default=False, help='ConvTranspose2d instead of bilinear up-sampling (models/unet.py:140; needs --is_stride_conv too)')),
    ('elab_end_points', dict(type=bool, default=True, help='expose per-stage features (the aux path needs them)')),
])
_flags('optimisation', '86-112', [
    ('ignored_index', dict(type=int, default=None, help='label value of unlabelled pixels (= the extra scribble plane; default: the --dataset preset, 5 for chaos)')),
    ('epoch', dict(type=int, default=400, help='epochs; also the horizon of the LR schedule and of the memory momentum')),
    ('batch_size', dict(type=int, default=12, help='images per GPU and step')),
    ('optimizer', dict(type=str, default='adam', choices=['adam', 'momentum'], help='FusedAdam or FusedSGD (one kernel over the flat parameter slab)')),
    ('momentum', dict(type=float, default=0.9, help='momentum of --optimizer momentum')),
    ('lr', dict(type=float, default=0.0001, help='learning rate at epoch 0')),
    ('lr_decay', dict(type=str, default='poly', choices=['linear', 'poly', 'cosine'], help='per-epoch schedule')),
    ('wd', dict(type=float, default=0.0003, help='L2 weight decay, added to the gradient')),
    ('ckp_interval', dict(type=int, default=10000, help='write ckps/ckp_<epoch>.pth every this many epochs (and at the end)')),
])
_flags('entropy minimisation', '114-126', [
    ('do_loss_ent', dict(default=False, help='add the entropy of the weak-view prediction', **_on)),
    ('loss_ent_weight', dict(type=float, default=1., help='its weight')),
    ('ramp_up_loss_ent', dict(default=True, help='Gaussian ramp-up of that weight over the epochs', **_on)),
    ('ramp_up_scale', dict(type=float, default=8., choices=[5., 8., 10.], help='sharpness of the ramp-ups')),
])
_flags('decoder consistency', '128-145', [
    ('do_decoder_consistency', dict(default=False, help='second (strong-view) pass + consistency loss between the two predictions', **_on)),
    ('ramp_up_loss_cr', dict(default=True, help='Gaussian ramp-up of its weight', **_on)),
    ('detach_weak_cr', dict(default=False, help='treat the weak-view probabilities as a constant target', **_on)),
    ('loss_cr_variants', dict(type=str, default='ce_loss', choices=['ce_loss', 'l1_loss', 'l2_loss', 'kl_loss'],
                              help='form of the consistency loss (all four in pp_seg_losses_*)')),
    ('strength', dict(type=float, default=1., choices=[0.125, 0.25, 0.5, 1.], help='range of the strong view\'s colour transforms')),
    ('loss_cr_weight', dict(type=float, default=1., help='weight of the consistency loss')),
])
_flags('auxiliary path', '147-169', [
    ('do_aux_path', dict(default=False, help='low-resolution classification head on the encoder output', **_on)),
    ('feat_stage', dict(type=list, default=['encoder/stage6', 'encoder/stage5'], help='encoder features it concatenates')),
    ('feat_ch', dict(type=list, default=[512, 512], help='their channel counts')),
    ('loss_aux_weight', dict(type=float, default=0.01, choices=[1., 0.01, 0.001], help='weight of its partial cross-entropy')),
    ('hid_ch', dict(type=int, default=64, choices=[256, 128, 64], help='embedding width = width of a memory-bank row')),
    ('aux_drop_prob', dict(type=float, default=0., choices=[0., 0.5, 0.8], help='Dropout2d on the embedding')),
])
_flags('memory bank', '171-179', [
    ('do_memory', dict(default=False, help='class prototypes updated from labelled pixels + their classification loss', **_on)),
    ('loss_memory_weight', dict(type=float, default=1, choices=[1., 0.01], help='weight of that loss')),
    ('update_momentum', dict(type=float, default=0.9, help='start of the EMA momentum schedule of the prototypes')),
    ('ensemble_mode', dict(type=str, default='cosine_similarity', choices=['cosine_similarity', 'mean'],
                           help='how labelled pixel embeddings are pooled into a prototype')),
])
# ---- additions of this implementation
parser.add_argument('--synthetic', type=int, default=0,
                    help='train on N synthetic phantom slices (and N//4 validation slices) instead of ./data')
parser.add_argument('--gpu_augment', action='store_true',
                    help='accepted for compatibility: the two-stream augmentation pipeline of datasets/augmentations.py on the GPU '
                         '(augment.py) is the default input path')
parser.add_argument('--cpu_input', action='store_true',
                    help='minimal CPU input path of data.py (MeanStdNorm, centre crop, colour jitter only: no Scaling / Elastic / '
                         'Rotation / Mirroring / Noise / RandomCrop) instead of the GPU pipeline')
parser.add_argument('--image_size', type=int, default=None, help='network input size of the TRAINING crops (default: the --dataset preset, 256 for chaos); validation runs at the native slice size')
parser.add_argument('--max_iters', type=int, default=0, help='stop each epoch after this many iterations (0 = all)')
parser.add_argument('--precision', type=str, default='fp32', choices=['fp32', 'fp16'],
                    help='fp32: every matrix product at fp32 grade (three fp16 MFMA products of split operands); fp16: the forward / '
                         'data-gradient products of the halo-tile and Winograd kernels use fp16 operands with fp32 accumulation '
                         '(mixed precision, BASELINE config 5; so does the Winograd weight-gradient GEMM; tensors in HBM stay fp32 unless --storage fp16)')
parser.add_argument('--storage', type=str, default='fp32', choices=['fp32', 'fp16', 'bf16'],
                    help='fp16 / bf16: activations and activation gradients of the TRAINING step live in HBM as IEEE fp16 / bfloat16 '
                         '(16-bit storage, BASELINE config 5 -- which names bf16: half the activation traffic; fp32 accumulation, '
                         'statistics, weights and parameter gradients; static loss scale).  fp16 keeps 11 significand bits, bf16 8 '
                         '(tests/test_gpu_h16.py states both tolerances).  Validation / inference stay fp32')
parser.add_argument('--graph_step', action='store_true',
                    help='replay the iteration (forward, losses, backward, optimizer) from a hipGraph captured once per epoch '
                         '(pacingpseudo_amd/graph.py): ONE host call per step instead of ~300 launches -- for hosts that cannot keep '
                         'ahead of the GPU; bit-identical to the eager step (single process or RCCL)')
parser.add_argument('--sync_bn', action='store_true',
                    help='data-parallel runs: BatchNorm batch statistics over the GLOBAL batch during epoch 0 '
                         '(= the single-process step on the concatenated batch); default: per-rank statistics, '
                         'running buffers averaged before the switch to eval mode')


# The reference ships ONE driver (train_chaos.py) and three dataset packages that differ only in class tables and
# crop size (datasets/{chaos,acdc,lvsc}/*_dataset.py:17-24, *_aug_configs.py:9-13).  --dataset selects the preset; the
# explicit --num_classes / --ignored_index / --image_size flags still win when they are given.
# `split_subdir`: where the five-fold split files live below ./data/<dataset>/train_test_split/five_fold_split/ and which
# sub-directory of --root the runs go to: CHAOS has one per modality, ACDC / LVSC have none (reference inference.py:301-318).
DATASETS = {
    'chaos': dict(num_classes=5, ignored_index=5, image_size=256, names=['BG', 'Liver', 'R-Kidney', 'L-Kidney', 'Spleen'],
                  split_subdir='{modality}'),
    'acdc': dict(num_classes=4, ignored_index=4, image_size=224, names=['BG', 'RV', 'Myo', 'LV'], split_subdir=''),
    'lvsc': dict(num_classes=2, ignored_index=2, image_size=224, names=['BG', 'Myo'], split_subdir=''),
}


def split_dir(dataset: str, modality: str = 't1'):
    """(data root, directory of the train_fold<k>.txt / test_fold<k>.txt lists) -- shared with inference.py."""
    sub = DATASETS[dataset]['split_subdir'].format(modality=modality)
    root = f'./data/{dataset}'
    return root, os.path.join(root, 'train_test_split', 'five_fold_split', sub).rstrip('/')


def _class_names(n, dataset='chaos'):
    names = list(DATASETS.get(dataset, DATASETS['chaos'])['names'])
    return names[:n] + [f'C{i}' for i in range(len(names), n)]


def apply_dataset_preset(args, argv=None):
    """Fill num_classes / ignored_index / image_size from the --dataset preset unless the flag was given: the three flags
    default to None, so whatever argparse accepted (abbreviations included) counts as given."""
    preset = DATASETS.get(args.dataset, DATASETS['chaos'])
    for key in ('num_classes', 'ignored_index', 'image_size'):
        if getattr(args, key, None) is None:
            setattr(args, key, preset[key])
    return args


def train_interface(args):
    from . import parallel
    from .data import SyntheticPhantoms, collate_by_shape, dataset_class, expand_compact, loader_context
    from .models import ConsistencyRegulr
    from .optim import FusedAdam, FusedSGD
    from .utils import AvgMeter, cosine_lr_decay, gaussian_ramp_up, linear_lr_decay, poly_lr_decay
    from .losses.losses import weighted_loss_sum
    from .utils.metrics import ValAccumulator
    from .utils.scalars import ScalarLog

    world, rank, local_rank = parallel.init_from_env('nccl')
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    best_avg, best_epoch, best_avg_class = 0, 0, []
    from ._lib import lib
    lib.pp_set_matrix_products(1 if getattr(args, 'precision', 'fp32') == 'fp16' else 3)

    model = ConsistencyRegulr(
        kwargs_unet=dict(input_ch=args.input_ch, init_ch=args.init_ch, max_ch=args.max_ch,
                         num_classes=args.num_classes, output_stride=args.output_stride,
                         is_stride_conv=args.is_stride_conv, is_trans_conv=args.is_trans_conv,
                         elab_end_points=args.elab_end_points),
        kwargs_aux_path=dict(num_classes=args.num_classes, feat_stage=args.feat_stage, feat_ch=args.feat_ch,
                             hid_ch=args.hid_ch, aux_drop_prob=args.aux_drop_prob, do_memory=args.do_memory,
                             max_step=args.epoch, update_momentum=args.update_momentum,
                             ensemble_mode=args.ensemble_mode),
        args_parser=args).cuda()
    if world > 1:
        parallel.attach(model, sync_bn=args.sync_bn)
    if rank == 0:
        logging.info(model)

    if args.optimizer == 'adam':
        optimizer = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
    elif args.optimizer == 'momentum':
        optimizer = FusedSGD(model.parameters(), lr=args.lr, momentum=args.momentum, weight_decay=args.wd)
    else:
        raise ValueError('Unimplemented optimizer')

    args.gpu_augment = not args.cpu_input          # the reference's recipe is the default; --cpu_input opts out
    if args.augmentations != 'TransformsColor' and not args.gpu_augment:
        raise NotImplementedError(f'--augmentations {args.augmentations} is not available with --cpu_input: the minimal CPU '
                                  'input path of data.py only has the colour jitter')
    ds_kw = dict(num_classes=args.num_classes, size=args.image_size, strength=args.strength, seed=args.seed)
    augmenter, collate = None, None
    if args.gpu_augment:
        # the reference's two-stream pipeline (chaos_dataset.py:58-90) on the device: the loader hands over raw slices
        from .augment import AugConfig, DeviceAugmenter, collate_raw
        augmenter = DeviceAugmenter(AugConfig(num_classes=args.num_classes, crop_size=(args.image_size, args.image_size),
                                              strength=args.strength, do_strong=args.do_decoder_consistency,
                                              recipe=args.augmentations),
                                    device=device, seed=args.seed + 7919 * rank)
        collate = collate_raw
    if args.synthetic:
        train_dataset = SyntheticPhantoms(args.synthetic, do_strong=args.do_decoder_consistency, train=True,
                                          raw=args.gpu_augment, **ds_kw)
        val_dataset = SyntheticPhantoms(max(args.synthetic // 4, 1), train=False, native=True, compact=True, **ds_kw)
    else:
        train_dataset = dataset_class(args.dataset)(args.train_ls, do_strong=args.do_decoder_consistency, train=True, raw=args.gpu_augment,
                                  **ds_kw)
        val_dataset = dataset_class(args.dataset)(args.val_ls, train=False, native=True, compact=True, **ds_kw)
        # Mixup blends with a slice drawn from the whole training list (datasets/augmentations.py:66): the loader draws it
        train_dataset.mix_partner = bool(args.gpu_augment and args.augmentations == 'TransformsColorMixup')
    sampler = torch.utils.data.distributed.DistributedSampler(train_dataset, world, rank, shuffle=True,
                                                              seed=args.seed, drop_last=True) if world > 1 else None
    train_loader = torch.utils.data.DataLoader(train_dataset, batch_size=args.batch_size, shuffle=sampler is None,
                                               sampler=sampler, num_workers=args.num_workers, drop_last=True,
                                               collate_fn=collate,
                                               # raw slices carry no per-epoch state: keep the workers alive across epochs
                                               # (re-forking them cost a third of a 1.5 s epoch in the r02 end-to-end run)
                                               persistent_workers=bool(args.gpu_augment and args.num_workers > 0),
                                               multiprocessing_context=loader_context(args.num_workers),
                                               pin_memory=bool(args.gpu_augment))
    # validation as train_chaos.py:235-241 runs it (MeanStdNorm only, native slice size), but: every rank scores its share of
    # the slices (strided, no padding duplicates), the workers stay alive across epochs, the meters live on the device
    val_subset = torch.utils.data.Subset(val_dataset, list(range(rank, len(val_dataset), world))) if world > 1 else val_dataset
    val_loader = torch.utils.data.DataLoader(val_subset, batch_size=args.batch_size, shuffle=False,
                                             num_workers=args.num_workers, drop_last=False, collate_fn=collate_by_shape,
                                             persistent_workers=args.num_workers > 0, pin_memory=True,
                                             multiprocessing_context=loader_context(args.num_workers))
    names = _class_names(args.num_classes, args.dataset)
    scalars = ScalarLog(os.path.join(args.child, 'tb_summary', 'scalars.jsonl')) if rank == 0 else None
    valdice = np.zeros(args.epoch)
    graph_step = None                            # --graph_step: pacingpseudo_amd.graph.GraphedStep, built on first use
    aug_stream = torch.cuda.Stream() if (augmenter is not None and os.environ.get('PP_AUG_STREAM', '1') != '0') else None
    for curr_epoch in range(args.epoch):
        epoch_tic = time.time()
        if sampler is not None:
            sampler.set_epoch(curr_epoch)
        train_dataset.set_epoch(curr_epoch)           # strong-view jitter differs per epoch (workers re-fork per epoch)
        if args.lr_decay == 'poly':
            optimizer, new_lr = poly_lr_decay(optimizer, curr_epoch, args.epoch, args.lr)
        elif args.lr_decay == 'cosine':
            optimizer, new_lr = cosine_lr_decay(optimizer, curr_epoch, args.epoch, args.lr)
        elif args.lr_decay == 'linear':
            optimizer, new_lr = linear_lr_decay(optimizer, curr_epoch, args.epoch, args.lr)
        else:
            raise ValueError('Unimplemented learning rate decay policy.')

        # loss meters live on the GPU: [sum pce*n, sum ent*n, sum cr*n, sum aux*n, sum mem, n, iterations]
        acc = torch.zeros(7, device=device, dtype=torch.float64)
        n_img = 0
        for idx, batch in enumerate(train_loader):
            if args.max_iters and idx >= args.max_iters:
                break
            if augmenter is not None:
                # upload + augmentation on their own stream: the host runs a step ahead of the GPU, so the 25 MB host-to-device copy
                # and the ~1.4 ms of augmentation kernels of THIS batch execute beside the previous training step instead of in
                # front of this one (round 5: 1,045 -> ~1,130 images/s through the driver in the reference's steady state)
                batch = augmenter.ahead(aug_stream, batch['img'], batch['lab'], batch['scb'], batch['sizes'], batch.get('mix'), batch.get('mix_sizes'))
            batch.pop('label', None)
            batch.pop('label_strong', None)
            batch = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            n = batch['image'].shape[0]
            # loss assembly of train_chaos.py:273-310; the ramp-up weights depend on the epoch only
            w_ent = gaussian_ramp_up(t=curr_epoch, base_value=args.loss_ent_weight, scale=args.ramp_up_scale) if args.ramp_up_loss_ent else 1.0
            w_cr = gaussian_ramp_up(t=curr_epoch, base_value=args.loss_cr_weight, scale=args.ramp_up_scale) if args.ramp_up_loss_cr else 1.0

            def assemble(out, _epoch=None):
                # train_chaos.py:273-310: pce + ent * w_ent + cr * w_cr + aux * w_aux + memory * w_mem -- one launch each way
                # (weighted_loss_sum: the same fp32 products and sums in the same order as the chain of torch operations)
                terms, weights = [out['loss_pce']], [1.0]
                if args.do_loss_ent:
                    terms.append(out['loss_ent']); weights.append(w_ent)
                if args.do_decoder_consistency:
                    terms.append(out['loss_cr']); weights.append(w_cr)
                if args.do_aux_path:
                    terms.append(out['loss_aux_cls']); weights.append(args.loss_aux_weight)
                    if args.do_memory:
                        terms.append(out['loss_memory']); weights.append(args.loss_memory_weight)
                return weighted_loss_sum(terms, weights)
            if args.graph_step and n == args.batch_size:      # (a ragged last batch runs eagerly: another shape, another plan)
                if graph_step is None:
                    from .graph import GraphedStep
                    graph_step = GraphedStep(model, optimizer, assemble, warmup=2)
                graph_step.loss_fn = assemble
                _, net_outputs = graph_step(batch, curr_epoch)
            else:
                net_outputs = model(batch, mode='train', step=curr_epoch)
                loss = assemble(net_outputs)
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
            acc[0] += net_outputs['loss_pce'].detach() * n
            if args.do_loss_ent:
                acc[1] += net_outputs['loss_ent'].detach() * (w_ent * n)
            if args.do_decoder_consistency:
                acc[2] += net_outputs['loss_cr'].detach() * (w_cr * n)
            if args.do_aux_path:
                acc[3] += net_outputs['loss_aux_cls'].detach() * (args.loss_aux_weight * n)
                if args.do_memory:
                    acc[4] += net_outputs['loss_memory'].detach() * args.loss_memory_weight
            acc[5] += n
            acc[6] += 1
            n_img += n
        a = acc.cpu().numpy()                      # the one host sync of the epoch
        epoch_toc = time.time()
        cnt, its = max(a[5], 1), max(a[6], 1)
        # 16-bit storage: optimizer steps the overflow guard skipped this epoch (a static loss scale that keeps
        # overflowing used to skip every update silently).  Read with the epoch's one host sync; when MOST of the epoch's updates
        # were skipped the scale is halved for the following epochs (a power of two: exact to remove), never silently.
        flat = getattr(model, 'flat', None)
        if flat is not None and getattr(flat, 'guard_on', False):
            total_skipped = int(flat.guard[1])
            skipped = total_skipped - getattr(flat, '_skipped_logged', 0)
            flat._skipped_logged = total_skipped
            if skipped:
                msg = "epoch: {:03d}: {} of {} optimizer steps skipped (non-finite gradients at loss scale {:g})".format(
                    curr_epoch, skipped, int(its), model.engine.loss_scale)
                if skipped * 2 > its:
                    new_scale = model.engine.set_loss_scale(model.engine.loss_scale / 2.0)
                    msg += "; loss scale halved to {:g}".format(new_scale)
                if rank == 0:
                    logging.warning(msg)
                if skipped >= its and model.engine.loss_scale < 1.0:
                    raise FloatingPointError('every optimizer step of the epoch was skipped even at loss scale < 1: the gradients '
                                             'themselves are not finite')
        if rank == 0:
            logging.info("epoch: {:03d}, lr: {:.6f}, loss_pce: {:.6f}, loss_ent: {:.6f}, loss_cr: {:.6f}, loss_aux_cls: {:.6f}, "
                         "loss_memory: {:.6f}, {:.2f} s/epoch".format(curr_epoch, new_lr, a[0] / cnt, a[1] / cnt, a[2] / cnt,
                                                                     a[3] / cnt, a[4] / its, epoch_toc - epoch_tic))
            logging.info("throughput: {:.1f} images/sec ({} GPU)".format(n_img * world / max(epoch_toc - epoch_tic, 1e-9), world))

        # ---- validation (train_chaos.py:369-399); model.eval() is never undone, as in the reference
        if world > 1 and model.training and not args.sync_bn:
            parallel.sync_bn_buffers(model)        # per-rank epoch-0 statistics -> one set of running buffers
        model.eval()
        meters = ValAccumulator(args.num_classes, device)
        tic = time.time()
        for groups in val_loader:
            for batch in groups:                   # same-shape groups of one loader batch (collate_by_shape)
                batch = expand_compact(batch, args.num_classes, device)      # uint8 class maps -> one-hot planes, on the device
                with torch.no_grad():
                    net_outputs = model(batch, mode='val')
                meters.update(net_outputs['segmentation/logits'], batch['label'], net_outputs['loss_pce'])
        dsc, loss_pce_val, _ = meters.result(parallel.all_reduce_sum if world > 1 else None)   # the one host sync
        toc = time.time()
        avg_all = np.mean([dsc[_] for _ in range(1, args.num_classes)])
        if rank == 0:
            logging.info("val: {:03d}, loss_pce: {:.6f}, time: {:.2f} s/epoch".format(curr_epoch, loss_pce_val, toc - tic))
            logging.info("[" + ", ".join("{}: {:.4f}".format(nm, dsc[i]) for i, nm in enumerate(names))
                         + ", All: {:.4f}]".format(avg_all))
            # the scalar tags of train_chaos.py:362-367 and :416-423 (TensorBoard there; one JSON line per scalar here)
            scalars.add('losses/loss_pce_train', a[0] / cnt, curr_epoch)
            scalars.add('losses/loss_cr', a[2] / cnt, curr_epoch)
            scalars.add('losses/loss_ent', a[1] / cnt, curr_epoch)
            scalars.add('losses/loss_aux_cls', a[3] / cnt, curr_epoch)
            scalars.add('losses/loss_memory', a[4] / its, curr_epoch)
            scalars.add('lr/current_lr', new_lr, curr_epoch)
            scalars.add('losses/loss_pce_val', loss_pce_val, curr_epoch)
            for i, nm in enumerate(names):
                scalars.add(f'DSC/{nm}', dsc[i], curr_epoch)
            scalars.add('DSC/All', avg_all, curr_epoch)
            scalars.add('DSC/Best', max(best_avg, avg_all), curr_epoch)
        valdice[curr_epoch] = avg_all
        if rank == 0:
            if curr_epoch + 1 == args.epoch or (curr_epoch + 1) % args.ckp_interval == 0:
                torch.save(model.state_dict(), os.path.join(args.child, 'ckps', 'ckp_{:d}.pth'.format(curr_epoch)))
            if avg_all > best_avg:
                best_epoch, best_avg = curr_epoch, avg_all
                best_avg_class = [dsc[_] for _ in range(1, args.num_classes)]
                torch.save(model.state_dict(), args.child + '/best_ckp.pth')
    if rank == 0:
        logging.info("The best at epoch: {:d}, ".format(best_epoch)
                     + ", ".join("{}: {:.4f}".format(nm, v) for nm, v in zip(names[1:], best_avg_class))
                     + ", All: {:.4f}".format(best_avg))
        np.savez(os.path.join(args.child, 'valdice'), valdice=valdice)
    return valdice


def train_main(argv=None):
    args = apply_dataset_preset(parser.parse_args(argv), argv)
    if 'LOCAL_RANK' not in os.environ:
        os.environ['CUDA_VISIBLE_DEVICES'] = args.gpu
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    rank = int(os.environ.get('RANK', '0'))
    sub = DATASETS.get(args.dataset, DATASETS['chaos'])['split_subdir'].format(modality=args.modality)
    args.child = os.path.join(os.path.join(args.root, sub) if sub else args.root, args.session,
                              f'{args.session}-{time.strftime("%H-%M-%S-%m%d")}-fold{args.fold}-{args.tag}')
    if rank == 0:
        os.makedirs(args.child, exist_ok=False)
        os.makedirs(os.path.join(args.child, 'ckps'), exist_ok=True)
        os.makedirs(os.path.join(args.child, 'tb_summary'), exist_ok=True)        # train_chaos.py:184-185
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
            train_ls = f.readlines()
        with open(f'{base}/test_fold{args.fold}.txt', 'r') as f:
            val_ls = f.readlines()
        args.train_ls = [(data_root + '/' + p).rstrip('\n') for p in train_ls]
        args.val_ls = [(data_root + '/' + p).rstrip('\n') for p in val_ls]
    return train_interface(args)


if __name__ == '__main__':
    train_main()
