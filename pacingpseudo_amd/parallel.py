"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI via torch.distributed.

The reference is single-GPU; this module adds the sharding the north_star asks for.  The batch
dimension is split across ranks (each rank holds a full model replica and B_local samples).  Three couplings keep
the result equal to the single-process step on the concatenated batch:

  * loss denominators (#labelled pixels, sum of valid_mask) are all-reduced BEFORE the losses are finalised and
    back-propagated (`Comm.allreduce_sums`), so every rank scales its local gradient by the GLOBAL count and the
    gradient all-reduce is a plain SUM -- no post-division, no mean-of-means bias when counts differ per rank;
  * the memory bank is updated on rank 0 from ITS sample 0 (= global sample 0, aux_path_memory.py:116) and
    broadcast; the bank-classification gradient is identical on all ranks and therefore scaled by 1/world;
  * weight gradients live in one flat slab and are summed bucket by bucket while the backward pass is still
    running: a bucket is handed to RCCL as soon as the last layer writing into it has been launched, on RCCL's own
    stream, so the 81 MB all-reduce (about 1 ms at xGMI ring rates) hides under the encoder's backward.

BatchNorm while it is in train mode (the reference's epoch 0 only; from epoch 1 on BN runs in eval mode,
train_chaos.py:370): with ``attach(model, sync_bn=True)`` every BatchNorm call all-reduces its packed per-channel sums
(forward: sum / sum of squares, backward: sum g / sum g*xhat), so the statistics, the running buffers and the
gradients equal those of ONE process on the concatenated batch (models/unet.py:189 semantics).  Without it the
statistics are per rank (B_local samples) and the running buffers drift apart during epoch 0; ``sync_bn_buffers`` then
averages them (and aligns num_batches_tracked) before the switch to eval mode, so that every rank normalises with the
same statistics from epoch 1 on and rank 0's checkpoint reflects all shards.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


class Comm:
    """Two communicators over the same ranks: `group` carries the gradient buckets (81 MB per step, asynchronous, overlapped
    with backward), `small` the blocking few-hundred-byte exchanges the step cannot proceed without -- loss denominators,
    the memory-bank row, SyncBN sums.  On one RCCL communicator (= one stream) a BatchNorm backward of epoch 0 would wait
    behind the 81 MB decoder bucket that was enqueued just before it; with its own communicator it does not."""

    def __init__(self, group=None, separate_small: bool = True):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.small = group
        if separate_small and (self.world > 1 or os.environ.get('PP_FORCE_DIST') == '1'):
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            self.small = dist.new_group(ranks=ranks, backend=dist.get_backend(group))     # collective: every rank calls it

    def allreduce_sums(self, t: torch.Tensor) -> None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.small)

    def broadcast_bank(self, bank: torch.Tensor) -> None:
        src = dist.get_global_rank(self.small, 0) if self.small is not None else 0
        dist.broadcast(bank.data, src=src, group=self.small)


def all_reduce_sum(t: torch.Tensor) -> None:
    """In-place SUM over the default process group (validation meters, utils.metrics.ValAccumulator.result)."""
    dist.all_reduce(t, op=dist.ReduceOp.SUM)


def backbone_buckets(model) -> List[Tuple[str, List[torch.nn.Parameter]]]:
    """Gradient buckets in the order the backward plan completes them (engine.backward_step)."""
    bb = model.backbone

    def ps(*mods):
        return [p for m in mods for p in m.parameters() if p.requires_grad]
    return [
        ('decoder_upper', ps(bb.dec_block4, bb.dec_block3, bb.dec_block2, bb.dec_block1, bb.final_conv)),
        ('dec5', ps(bb.dec_block5)),
        ('aux', ps(model.aux_path)),
        ('enc6', ps(bb.enc_block6)),
        ('enc5', ps(bb.enc_block5)),
        ('enc_rest', ps(bb.enc_block1, bb.enc_block2, bb.enc_block3, bb.enc_block4)),
    ]


class GradReducer:
    """Sums the flat gradient slab across ranks, one contiguous bucket at a time, overlapped with backward."""

    def __init__(self, model, comm: Comm):
        self.comm = comm
        self.model = model
        self.pending = []
        self.ranges: Dict[str, Tuple[int, int]] = {}

    def _range(self, flat, tag) -> Tuple[int, int]:
        r = self.ranges.get((id(flat), tag))
        if r is None:
            params = dict(backbone_buckets(self.model))[tag]
            offs = sorted((flat.offsets[p], p.numel()) for p in params)
            lo, hi = offs[0][0], offs[-1][0] + offs[-1][1]
            if sum(n for _, n in offs) != hi - lo:
                raise RuntimeError(f'bucket {tag} is not contiguous in the gradient slab')
            r = (lo, hi)
            self.ranges[(id(flat), tag)] = r
        return r

    def bucket_ready(self, flat, tag: str) -> None:
        """Called by the engine right after the launches that complete bucket `tag` were enqueued."""
        lo, hi = self._range(flat, tag)
        if dist.get_backend(self.comm.group) == 'nccl':
            # RCCL: enqueued on its own stream behind the launches above, overlapped with the rest of the backward pass
            self.pending.append(dist.all_reduce(flat.grads[lo:hi], op=dist.ReduceOp.SUM, group=self.comm.group,
                                                async_op=True))
        else:
            # gloo (CPU tests, one-GPU rehearsals): several asynchronous collectives in flight deadlocked with four
            # ranks (worker threads picking them up in different orders); run them one by one at the end of backward
            self.pending.append((lo, hi))

    def reduce(self, flat, active: Sequence[str]) -> None:
        """Finish the step: wait (stream-side) for every bucket launched during backward."""
        for w in self.pending:
            if isinstance(w, tuple):
                dist.all_reduce(flat.grads[w[0]:w[1]], op=dist.ReduceOp.SUM, group=self.comm.group)
            else:
                w.wait()
        self.pending = []


def sync_bn_buffers(model, group=None) -> None:
    """Average running_mean / running_var over the ranks and take rank 0's num_batches_tracked (what
    torch.nn.SyncBatchNorm / DDP(broadcast_buffers) leave behind, here as one flat all-reduce).  Call it before
    ``model.eval()`` when BatchNorm ran with per-rank statistics; a no-op outside a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    floats = [b for n, b in model.named_buffers() if n.endswith('running_mean') or n.endswith('running_var')]
    if floats:
        flat = torch.cat([b.reshape(-1) for b in floats])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= world
        o = 0
        for b in floats:
            b.copy_(flat[o:o + b.numel()].view_as(b))
            o += b.numel()
    for n, b in model.named_buffers():
        if n.endswith('num_batches_tracked'):
            dist.broadcast(b, src=0, group=group)


def attach(model, group=None, sync_bn: bool = False) -> Comm:
    """Make `model` (ConsistencyRegulr) data-parallel over the default process group.  sync_bn: BatchNorm batch
    statistics over the global batch while BN is in train mode (one packed all-reduce per BN call, fwd and bwd)."""
    comm = Comm(group)
    eng = model.engine
    eng.comm, eng.world, eng.rank = comm, comm.world, comm.rank
    eng.sync_bn = bool(sync_bn)
    model._reducer = GradReducer(model, comm)
    # identical initial weights everywhere
    flat = model._ensure_flat()
    dist.broadcast(flat.params, src=0, group=group)
    for b in model.buffers():
        dist.broadcast(b, src=0, group=group)
    dist.broadcast(model.aux_path.memory_bank.data, src=0, group=group)
    # the broadcast wrote the weights through the slab: a forward-only plan that packed its kernel-side weight layouts before
    # attach() (a validation pass on a rank other than 0) must pack again
    flat.version += 1
    eng.invalidate_packed()
    return comm


def init_from_env(backend: str = 'nccl'):
    """torch.distributed.run contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # rehearsal switch: PP_DIST_BACKEND=gloo PP_SHARE_GPU=1 runs N ranks of the real driver / bench code on ONE GPU (RCCL
    # refuses two ranks on one device); the product default is RCCL with one GPU per rank
    backend = os.environ.get('PP_DIST_BACKEND', backend)
    if os.environ.get('PP_SHARE_GPU') == '1':
        local_rank = 0
    # PP_FORCE_DIST=1: build the process group for ONE rank as well (a one-rank RCCL communicator on a one-GPU box exercises the
    # library calls of the N > 1 path: init with device_id, the second communicator, all-reduce / broadcast / barrier)
    forced = os.environ.get('PP_FORCE_DIST') == '1'
    if forced:
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if (world > 1 or forced) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    return world, rank, local_rank
