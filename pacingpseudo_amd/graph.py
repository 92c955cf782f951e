"""GraphedStep: one whole training iteration captured into a hipGraph and replayed with ONE host call per step.

The iteration body of the reference (train_chaos.py:263-315: forward of ``ConsistencyRegulr``, loss assembly, ``zero_grad`` /
``backward`` / ``step``) is, here, a fixed sequence of ~300 C-ABI launches on two HIP streams that the Python host enqueues one
by one (3 ms of host time per 31 ms step; ``bench.py`` reports how far the host runs ahead as ``host.lead_ms``).  On a slow or
contended host -- eight ranks sharing the cores of one node -- that enqueue can become the limiter.  ``GraphedStep`` removes
it: after ``warmup`` ordinary (eager) calls it captures the next call into a ``torch.cuda.CUDAGraph`` (= hipGraph on ROCm) --
both streams, the fork / join events between them, the fused optimizer -- and every later call is ``graph.replay()``.

What makes the step capturable (round 5):
  * nothing a kernel reads per step is a host value: Adam's step counts and the learning rate live on the device
    (``optim.py``), BatchNorm's ``num_batches_tracked`` always did; values that change per EPOCH only -- the ramp-up weights of
    the loss assembly, the memory-bank momentum (aux_path_memory.py:118-120) -- are part of the capture key, so a new epoch
    captures a new graph (one eager-cost call per epoch);
  * every backward pass starts with both dz slots free and joins the second stream before it returns (``engine.py``), so a
    captured step never waits on an event recorded outside the capture;
  * the library allocates nothing and synchronises nothing inside its launch functions (include/pacingpseudo_hip.h).
The replayed kernels are the eager kernels on the same buffers in the same order: parameters after N replayed steps equal the
eager run's bit for bit (tests/test_gpu_graph.py).  Not capturable: a gloo process group (CPU collectives); RCCL is.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch


class GraphedStep:
    def __init__(self, model, optimizer, loss_fn: Callable[[dict, int], torch.Tensor], warmup: int = 2):
        """loss_fn(net_outputs, epoch) -> scalar loss tensor (the loss assembly of train_chaos.py:273-310)."""
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.warmup = max(1, int(warmup))      # >= 1: the eager calls build the plan, the optimizer state and the launch attributes
        self.calls = 0
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.key = None
        self.static_batch: Dict[str, torch.Tensor] = {}
        self.static_out: Optional[dict] = None
        self.static_loss: Optional[torch.Tensor] = None
        self.captures = 0
        self.replays = 0

    # ------------------------------------------------------------------------------------------------------------
    def _eager(self, batch, epoch):
        out = self.model(batch, mode='train', step=epoch)
        loss = self.loss_fn(out, epoch)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss, out

    def _key(self, batch, epoch):
        m = self.model
        eng = m.engine
        return (int(epoch), bool(m.backbone.training), bool(m.aux_path.training),
                tuple((k, tuple(v.shape), v.dtype, v.device.index) for k, v in sorted(batch.items()) if torch.is_tensor(v)),
                float(getattr(eng, 'loss_scale', 1.0)), str(getattr(eng, 'storage', 'fp32')), id(eng.comm), bool(eng.sync_bn),
                tuple((float(g.get('weight_decay', 0.0)), tuple(g.get('betas', ())), float(g.get('momentum', 0.0)))
                      for g in self.opt.param_groups),
                # device objects baked into a capture: the engine's training plans (their buffers), the learning-rate
                # scalars and the parameter slab -- a rebuilt plan, a re-created scalar or a re-flattened model captures again
                tuple(id(p) for p in eng.plans.values() if p.trainable), self.opt.hyper_ptrs() if hasattr(self.opt, 'hyper_ptrs') else (),
                id(getattr(m, 'flat', None)))

    def _capture(self, batch, epoch, key):
        comm = self.model.engine.comm
        if comm is not None:
            import torch.distributed as dist
            if dist.get_backend(comm.group) != 'nccl':
                raise RuntimeError('GraphedStep: only RCCL (backend nccl) collectives can be captured into a hipGraph')
        # a capture that raises must not leave the previous epoch's graph behind under a stale key
        self.graph, self.key, self.static_loss, self.static_out = None, None, None, None
        self.static_batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        self._static_nontensor = {k: v for k, v in batch.items() if not torch.is_tensor(v)}
        self.opt.sync_hyper()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):      # (loader threads may allocate pinned memory meanwhile)
            loss, out = self._eager(self.static_batch, epoch)
        self.graph, self.key = g, key
        self.static_loss, self.static_out = loss, out
        self.captures += 1

    def __call__(self, batch, epoch):
        """One iteration on `batch` (dict of device tensors) at `epoch`; returns (loss, net_outputs).  From the first replay on the
        returned tensors are the graph's static outputs: read them before the next call overwrites them."""
        self.calls += 1
        if self.calls <= self.warmup:
            return self._eager(batch, epoch)
        key = self._key(batch, epoch)
        if key != self.key:
            self._capture(batch, epoch, key)          # records the step without executing it; the replay below runs it
        else:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    if v.data_ptr() != self.static_batch[k].data_ptr():
                        self.static_batch[k].copy_(v, non_blocking=True)
                elif self._static_nontensor.get(k) != v:
                    raise ValueError(f'GraphedStep: batch[{k!r}] is not a tensor and changed since the capture ({self._static_nontensor.get(k)!r} '
                                     f'-> {v!r}): non-tensor entries are frozen into the captured step')
        self.opt.sync_hyper()                         # lr may have changed on the host (poly_lr_decay): refresh the device scalar
        self.graph.replay()
        self.replays += 1
        flat = getattr(self.model, 'flat', None)
        if flat is not None:
            flat.version += 1                         # the replayed optimizer wrote the weights (forward-only plans must re-pack)
        return self.static_loss, self.static_out
