"""FusedAdam: ``torch.optim.Adam(params, lr, weight_decay)`` semantics in one HIP kernel per slab segment.

Reference call site: train_chaos.py:218-223 (Adam, lr 1e-4, L2-coupled weight decay 3e-4) and
train_chaos.py:313-315 (zero_grad / backward / step).  The class keeps the ``torch.optim.Optimizer`` protocol
(``param_groups[i]['lr']`` is what utils.poly_lr_decay writes, ``zero_grad()``, ``step()``, ``state_dict()``),
but the update runs over the model's flat slabs (pacingpseudo_amd.flat.FlatSlab): p, g, m, v are read once and
p, m, v written once, 28 B per parameter, instead of one foreach launch chain per tensor list.
"""
from __future__ import annotations

import torch

from ._lib import lib, stream_ptr


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0:
            raise ValueError('invalid Adam hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._slabs = {}      # id(FlatSlab) -> dict(m, v, steps{segment: int})

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError('closures are not supported')
        st = stream_ptr()
        for group in self.param_groups:
            todo = {}
            for p in group['params']:
                if p.grad is None:
                    continue                      # torch.optim.Adam skips parameters without a gradient
                flat = getattr(p, '_pp_flat', None)
                if flat is None or not flat.owns(p) or p.grad.data_ptr() != flat.grad_views[p].data_ptr():
                    raise RuntimeError('FusedAdam only updates parameters that live in a pacingpseudo_amd FlatSlab '
                                       '(build the model with pacingpseudo_amd and call .cuda() before training)')
                todo.setdefault(id(flat), (flat, set()))[1].add(p)
            for flat, have in todo.values():
                state = self._slabs.setdefault(id(flat), dict(
                    m=torch.zeros_like(flat.params), v=torch.zeros_like(flat.params), steps={}))
                b1, b2 = group['betas']
                for name, (a, b) in flat.segments.items():
                    ps = flat.seg_params[name]
                    n_have = sum(1 for p in ps if p in have)
                    if n_have == 0:
                        continue
                    if n_have != len(ps):
                        raise RuntimeError(f'segment {name}: only {n_have}/{len(ps)} parameters carry a gradient')
                    t = state['steps'].get(name, 0) + 1
                    state['steps'][name] = t
                    lib.pp_adam_step(flat.params.data_ptr() + 4 * a, flat.grads.data_ptr() + 4 * a,
                                     state['m'].data_ptr() + 4 * a, state['v'].data_ptr() + 4 * a, b - a,
                                     float(group['lr']), float(b1), float(b2), float(group['eps']),
                                     float(group['weight_decay']), t, st)
                flat.version += 1
        return None
