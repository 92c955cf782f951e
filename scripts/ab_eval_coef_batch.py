import sys, time, json
sys.path.insert(0, '/root/repo')
import torch, bench
from pacingpseudo_amd import engine as E
from pacingpseudo_amd.data import full_flags, synthetic_batch
from pacingpseudo_amd.optim import FusedAdam
dev = torch.device('cuda', 0)
a = full_flags()
m = bench.build(a, dev); o = FusedAdam(m.parameters(), lr=a.lr, weight_decay=a.wd)
batch = {k: v.to(dev) for k, v in synthetic_batch(32, 256, 256, a.num_classes, seed=0).items() if k != 'label'}
m.train()
for _ in range(3): bench.train_iteration(m, o, batch, a, 0)
m.eval()
def run(n=20):
    for _ in range(3): bench.train_iteration(m, o, batch, a, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): bench.train_iteration(m, o, batch, a, 1)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for r in range(3):
    for flag in (True, False):
        E.COEF_BATCH = flag
        print('round', r, 'COEF_BATCH', flag, 'eval-mode step %.3f ms' % run(), flush=True)
