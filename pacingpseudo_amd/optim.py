"""FusedAdam / FusedSGD: ``torch.optim.Adam`` / ``torch.optim.SGD`` semantics in one HIP kernel per slab segment.

Reference call site: train_chaos.py:218-223 (Adam, lr 1e-4, L2-coupled weight decay 3e-4) and
train_chaos.py:313-315 (zero_grad / backward / step).  The class keeps the ``torch.optim.Optimizer`` protocol
(``param_groups[i]['lr']`` is what utils.poly_lr_decay writes, ``zero_grad()``, ``step()``, ``state_dict()``),
but the update runs over the model's flat slabs (pacingpseudo_amd.flat.FlatSlab): p, g, m, v are read once and
p, m, v written once, 28 B per parameter, instead of one foreach launch chain per tensor list.

Step counts live ON THE DEVICE (round 5): one int32 per slab segment, advanced by the library after an update
that really happened.  A step the 16-bit storage mode's overflow guard skipped therefore does not advance Adam's bias
corrections (``torch.cuda.amp.GradScaler.step`` does not count skipped steps either), is counted ONCE in ``flat.guard[1]``
however many segments the step had, and nothing the kernels read per step is a host value -- which is what lets a captured
hipGraph of the whole iteration replay correctly.  The learning rate is mirrored into a device scalar for the same reason
(``sync_hyper``: refreshed whenever ``param_groups[i]['lr']`` changed, i.e. once per epoch).
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
        self._slabs = {}      # id(FlatSlab) -> dict(<STATE_KEYS tensors>, steps_dev int32[segments], names, flat=FlatSlab)
        self._pending = None  # state handed to load_state_dict before the slab exists
        self._lr_dev = {}     # (index of the param group, device) -> [device scalar, the host value it holds]

    # ---- hyper-parameters the kernels read from the device ------------------------------------------------------
    def lr_scalar(self, group, device) -> torch.Tensor:
        """Device scalar holding ``group['lr']`` (written only when the host value changed: poly_lr_decay sets it per epoch)."""
        # keyed by the group's POSITION: load_state_dict / add_param_group may replace the dict objects, and a
        # captured hipGraph keeps reading the scalar it was captured with -- the scalar of group i must stay the same tensor
        gi = next((i for i, g in enumerate(self.param_groups) if g is group), None)
        if gi is None:
            raise ValueError('lr_scalar: not one of this optimizer\'s param_groups')
        key = (gi, device.index)
        ent = self._lr_dev.get(key)
        lr = float(group['lr'])
        if ent is None:
            ent = [torch.full((1,), lr, device=device, dtype=torch.float32), lr]
            self._lr_dev[key] = ent
        elif ent[1] != lr:
            ent[0].fill_(lr)
            ent[1] = lr
        return ent[0]

    def hyper_ptrs(self) -> tuple:
        """Addresses of the device scalars the step kernels read (part of GraphedStep's capture key: a replay is only valid while
        the captured kernels' scalars are the ones sync_hyper refreshes)."""
        return tuple(sorted((k, v[0].data_ptr()) for k, v in self._lr_dev.items()))

    def sync_hyper(self) -> None:
        """Refresh the device copies of the host-side hyper-parameters (call OUTSIDE a graph capture, before a replay)."""
        for group in self.param_groups:
            for p in group['params']:
                if p.is_cuda:
                    self.lr_scalar(group, p.device)
                    break

    def _steps_host(self, state) -> dict:
        vals = state['steps_dev'].tolist()                       # one host sync (state_dict / tests only)
        return {name: int(vals[i]) for i, name in enumerate(state['names']) if vals[i] > 0}

    def _set_steps(self, state, steps: dict) -> None:
        vals = [int(steps.get(name, 0)) for name in state['names']]
        state['steps_dev'].copy_(torch.tensor(vals, dtype=torch.int32))

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
                state['names'] = list(flat.segments)
                state['steps_dev'] = torch.zeros(len(state['names']), device=flat.params.device, dtype=torch.int32)
                if self._pending is not None:
                    for k in self.STATE_KEYS:
                        state[k].copy_(self._pending[k].to(flat.params.device))
                    self._set_steps(state, self._pending['steps'])
                    self._pending = None
                self._slabs[id(flat)] = state
            active = []
            for i, (name, (a, b)) in enumerate(flat.segments.items()):
                ps = flat.seg_params[name]
                n_have = sum(1 for p in ps if p in have)
                if n_have == 0:
                    continue
                if n_have != len(ps):
                    raise RuntimeError(f'segment {name}: only {n_have}/{len(ps)} parameters carry a gradient')
                active.append((name, a, b, state['steps_dev'].data_ptr() + 4 * i))
            yield flat, state, active

    def state_dict(self):
        sd = dict(param_groups=[{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups])
        sd['slabs'] = [dict({k: st[k].detach().cpu() for k in self.STATE_KEYS}, steps=self._steps_host(st))
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
                self._set_steps(cur, st['steps'])
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
                    lr_dev = self.lr_scalar(group, flat.params.device).data_ptr()
                    for j, (name, a, b, step_ptr) in enumerate(active):
                        lib.pp_adam_step_dev(flat.params.data_ptr() + 4 * a, flat.grads.data_ptr() + 4 * a,
                                             state['m'].data_ptr() + 4 * a, state['v'].data_ptr() + 4 * a, b - a,
                                             float(group['lr']), lr_dev, float(b1), float(b2), float(group['eps']),
                                             float(group['weight_decay']), step_ptr, skip, 1 if j == 0 else 0, st)
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
                lr_dev = self.lr_scalar(group, flat.params.device).data_ptr()
                for j, (name, a, b, step_ptr) in enumerate(active):
                    lib.pp_sgd_momentum_step_dev(flat.params.data_ptr() + 4 * a, flat.grads.data_ptr() + 4 * a,
                                                 state['momentum_buffer'].data_ptr() + 4 * a, b - a, float(group['lr']), lr_dev,
                                                 float(group['momentum']), float(group['weight_decay']), step_ptr, skip,
                                                 1 if j == 0 else 0, st)
                flat.version += 1
        return None
