"""Which thread count gives the CPU oracle its best throughput on this box (fair cpu_baseline)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pacing_oracle as O
a = O.full_flags()
for nt in (16, 32, 64, 128):
    torch.set_num_threads(nt)
    sd = O.init_state(a, seed=1); batch = O.synthetic_batch(8, 256, 256, 5, seed=0); adam = O.AdamState()
    O.train_step(sd, batch, 0, a, True, adam)
    t0 = time.perf_counter(); O.train_step(sd, batch, 0, a, True, adam); dt = time.perf_counter() - t0
    print(f'threads {nt}: {8 / dt:.2f} images/sec ({dt:.2f} s/step)', flush=True)
