"""Where does the HOST spend an iteration of the training driver?  Monkey-patches the pieces of the loop with wall-clock timers (no
device synchronisation added) and runs train_chaos.py's main for two epochs:   python scripts/driver_loop_timing.py [extra driver flags]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pacingpseudo_amd import train as T, augment as A
from pacingpseudo_amd.models import consistency_reglur_memory as M
from pacingpseudo_amd import optim as OPT
acc = collections.defaultdict(lambda: [0.0, 0])
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[name]; e[0] += time.perf_counter() - t; e[1] += 1
    return w
A.DeviceAugmenter.__call__ = timed('augmenter', A.DeviceAugmenter.__call__)
A.DeviceAugmenter.draw = timed('  aug.draw (host RNG)', A.DeviceAugmenter.draw)
A.DeviceAugmenter._up = timed('  aug._up (H2D of parameters)', A.DeviceAugmenter._up)
A.DeviceAugmenter.strong = timed('  aug.strong', A.DeviceAugmenter.strong)
A.pack_params = timed('  aug.pack_params', A.pack_params)
torch.Tensor.to = timed('  Tensor.to', torch.Tensor.to)
torch.Tensor.clone = timed('  Tensor.clone', torch.Tensor.clone)
torch.Tensor.contiguous = timed('  Tensor.contiguous', torch.Tensor.contiguous)
torch.Tensor.pin_memory = timed('  Tensor.pin_memory', torch.Tensor.pin_memory)
_empty, _empty_like = torch.empty, torch.empty_like
torch.empty = timed('  torch.empty', _empty)
torch.empty_like = timed('  torch.empty_like', _empty_like)
_fwd = M.ConsistencyRegulr.forward
M.ConsistencyRegulr.forward = lambda self, *a, **k: timed(f'{phase[0]}: model forward (enqueue)', _fwd)(self, *a, **k)
M.ConsistencyRegulr._run_backward = timed('backward (enqueue)', M.ConsistencyRegulr._run_backward)
OPT.FusedAdam.step = timed('optimizer.step', OPT.FusedAdam.step)
from pacingpseudo_amd.utils import metrics as MET
phase = ['train']
_vinit, _vres = MET.ValAccumulator.__init__, MET.ValAccumulator.result
def vinit(self, *a, **k):
    phase[0] = 'val'
    return _vinit(self, *a, **k)
def vres(self, *a, **k):
    r = timed('val: meters.result (host sync)', _vres)(self, *a, **k)
    phase[0] = 'train'
    return r
MET.ValAccumulator.__init__, MET.ValAccumulator.result = vinit, vres
MET.ValAccumulator.update = timed('val: meters.update (enqueue)', MET.ValAccumulator.update)
T.ValAccumulator = MET.ValAccumulator
_next = torch.utils.data.dataloader._BaseDataLoaderIter.__next__
def nxt(self):
    return timed(f'{phase[0]}: loader next()', _next)(self)
torch.utils.data.dataloader._BaseDataLoaderIter.__next__ = nxt
t0 = time.perf_counter()
T.train_main(['--session', 'Experiment', '--tag', 'timing', '--root', '/tmp/timing_root', '--synthetic', '1024', '--epoch', '2', '--batch_size', '32',
              '--image_size', '256', '--num_workers', '4', '--do_loss_ent', '--do_decoder_consistency', '--do_aux_path', '--do_memory'] + sys.argv[1:])
print(f'total {time.perf_counter() - t0:.2f} s')
for k, (s, n) in acc.items():
    print(f'{k:34s} {n:5d} calls  {s * 1e3 / max(n, 1):8.2f} ms each  {s:7.2f} s total')
