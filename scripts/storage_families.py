"""Per-family kernel time of one training step in each activation-storage mode (fp32 / fp16 / bf16), one stream, HIP events around
every launch (pp_prof_*):   python scripts/storage_families.py [modes ...]   -> one line per mode"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from pacingpseudo_amd import engine as E  # noqa: E402
from pacingpseudo_amd._lib import lib, prof_collect  # noqa: E402
from pacingpseudo_amd.data import full_flags, synthetic_batch  # noqa: E402
from pacingpseudo_amd.optim import FusedAdam  # noqa: E402

dev = torch.device('cuda', 0)
E.WGRAD_STREAM = False
batch = None
for mode in (sys.argv[1:] or ['fp32', 'fp16', 'bf16']):
    a = full_flags()
    a.storage = mode
    m = bench.build(a, dev)
    o = FusedAdam(m.parameters(), lr=a.lr, weight_decay=a.wd)
    if batch is None:
        batch = {k: v.to(dev) for k, v in synthetic_batch(32, 256, 256, a.num_classes, seed=0).items() if k != 'label'}
    m.train()
    for _ in range(3):
        bench.train_iteration(m, o, batch, a, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        bench.train_iteration(m, o, batch, a, 0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    lib.pp_prof_select((1 << 64) - 1)
    lib.pp_prof_enable(1)
    prof_collect()
    for _ in range(2):
        bench.train_iteration(m, o, batch, a, 0)
    torch.cuda.synchronize()
    lib.pp_prof_enable(0)
    p = prof_collect()
    print(json.dumps(dict(storage=mode, one_stream_ms_per_step=round(ms, 3),
                          families={k: round(v['ms'] / 2, 3) for k, v in p.items() if v['launches']})), flush=True)
    del m, o
    torch.cuda.empty_cache()
