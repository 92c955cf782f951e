"""FusedAdam / FusedSGD: ``torch.optim.Adam`` / ``torch.optim.SGD`` semantics in one HIP kernel per slab segment.

Reference call site: train_chaos.py:218-223 (Adam, lr 1e-4, L2-coupled weight decay 3e-4) and
train_chaos.py:313-315 (zero_grad / backward / step).  The class keeps the ``torch.optim.Optimizer`` protocol
(``param_groups[i]['lr']`` is what utils.poly_lr_decay writes, ``zero_grad()``, ``step()``, ``state_dict()``),
but the update runs over the model's flat slabs (pacingpseudo_amd.flat.FlatSlab): p, g, m, v are read once and
p, m, v written once, 28 B per parameter, instead of one foreach launch chain per tensor list.
"""
from __future__ import annotations

import torch

from ._lib import lib, prof_range, stream_ptr


class _SlabOptimizer(torch.optim.Optimizer):
    """Shared plumbing: which FlatSlab segments carry gradients this step, and the optimiser state as slab-shaped
    tensors (exposed through state_dict / load_state_dict so a run can be resumed; the reference saves none,
    train_chaos.py:405-413)."""
    STATE_KEYS: tuple = ()

    def _init_state(self):
        self._slabs = {}      # id(FlatSlab) -> dict(<STATE_KEYS tensors>, steps{segment: int}, flat=FlatSlab)
        self._pending = None  # state handed to load_state_dict before the slab exists

    def _segments_with_grads(self, group):
        todo = {}
        for p in group['params']:
            if p.grad is None:
                continue                      # torch.optim skips parameters without a gradient
            flat = getattr(p, '_pp_flat', None)
            if flat is None or not flat.owns(p) or p.grad.data_ptr() != flat.grad_views[p].data_ptr():
                raise RuntimeError(f'{type(self).__name__} only updates parameters that live in a pacingpseudo_amd FlatSlab '
                                   '(build the model with pacingpseudo_amd and call .cuda() before training)')
            todo.setdefault(id(flat), (flat, set()))[1].add(p)
        for flat, have in todo.values():
            if self._slabs and id(flat) not in self._slabs:
                raise RuntimeError('the model was re-flattened (.cuda()/.to() after the first optimiser step): the '
                                   'optimiser state belongs to the old parameter slab; rebuild the optimiser or carry '
                                   'the state over with state_dict() / load_state_dict()')
            state = self._slabs.get(id(flat))
            if state is None:
                state = {k: torch.zeros_like(flat.params) for k in self.STATE_KEYS}
                state['steps'] = {}
                if self._pending is not None:
                    for k in self.STATE_KEYS:
                        state[k].copy_(self._pending[k].to(flat.params.device))
                    state['steps'] = dict(self._pending['steps'])
                    self._pending = None
                self._slabs[id(flat)] = state
            active = []
            for name, (a, b) in flat.segments.items():
                ps = flat.seg_params[name]
                n_have = sum(1 for p in ps if p in have)
                if n_have == 0:
                    continue
                if n_have != len(ps):
                    raise RuntimeError(f'segment {name}: only {n_have}/{len(ps)} parameters carry a gradient')
                t = state['steps'].get(name, 0) + 1
                state['steps'][name] = t
                active.append((name, a, b, t))
            yield flat, state, active

    def state_dict(self):
        sd = dict(param_groups=[{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups])
        sd['slabs'] = [dict({k: st[k].detach().cpu() for k in self.STATE_KEYS}, steps=dict(st['steps']))
                       for st in self._slabs.values()]
        return sd

    def load_state_dict(self, sd):
        for g, saved in zip(self.param_groups, sd['param_groups']):
            g.update(saved)
        if sd.get('slabs'):
            if len(sd['slabs']) != 1:
                raise ValueError('expected the state of exactly one parameter slab')
            st = sd['slabs'][0]
            if self._slabs:
                cur = next(iter(self._slabs.values()))
                for k in self.STATE_KEYS:
                    cur[k].copy_(st[k].to(cur[k].device))
                cur['steps'] = dict(st['steps'])
            else:
                self._pending = st


class FusedAdam(_SlabOptimizer):
    STATE_KEYS = ('m', 'v')

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0:
            raise ValueError('invalid Adam hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._init_state()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError('closures are not supported')
        st = stream_ptr()
        with prof_range('optimizer: Adam'):
            for group in self.param_groups:
                b1, b2 = group['betas']
                for flat, state, active in self._segments_with_grads(group):
                    skip = flat.guard.data_ptr() if getattr(flat, 'guard_on', False) else None      # 16-bit storage: overflow guard
                    for name, a, b, t in active:
                        lib.pp_adam_step_guard(flat.params.data_ptr() + 4 * a, flat.grads.data_ptr() + 4 * a,
                                               state['m'].data_ptr() + 4 * a, state['v'].data_ptr() + 4 * a, b - a,
                                               float(group['lr']), float(b1), float(b2), float(group['eps']),
                                               float(group['weight_decay']), t, skip, st)
                    flat.version += 1
        return None


class FusedSGD(_SlabOptimizer):
    """``torch.optim.SGD(params, lr, momentum, weight_decay)`` (train_chaos.py:220-221, ``--optimizer momentum``):
    L2-coupled decay, dampening 0, no Nesterov, momentum buffer initialised with the first gradient."""
    STATE_KEYS = ('momentum_buffer',)

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError('invalid SGD hyper-parameter')
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._init_state()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError('closures are not supported')
        st = stream_ptr()
        for group in self.param_groups:
            for flat, state, active in self._segments_with_grads(group):
                skip = flat.guard.data_ptr() if getattr(flat, 'guard_on', False) else None
                for name, a, b, t in active:
                    lib.pp_sgd_momentum_step_guard(flat.params.data_ptr() + 4 * a, flat.grads.data_ptr() + 4 * a,
                                                   state['momentum_buffer'].data_ptr() + 4 * a, b - a, float(group['lr']),
                                                   float(group['momentum']), float(group['weight_decay']), t, skip, st)
                flat.version += 1
        return None
