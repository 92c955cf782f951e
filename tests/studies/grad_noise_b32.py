"""How much of the batch-32 gradient difference between the HIP path and the fp32 CPU oracle is the ORACLE's own fp32 summation
noise?  One full-flags train-mode-BN step at bench.py's configuration (32 images per view, 256x256), three sets of parameter
gradients with the LeakyReLU / max-pool choices aligned to the device's:
    hip   -- the HIP path;   o32 -- the oracle in fp32 (what the parity test compares with);   o64 -- the oracle in fp64.
Prints, per parameter, max-norm relative errors hip:o32, hip:o64, o32:o64 (worst first) and writes them to
gpurun_out/grad_noise_b32.json.  Run on the GPU box:  python tests/studies/grad_noise_b32.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pacing_oracle as O  # noqa: E402
from tests import _golden as G  # noqa: E402
from tests.test_gpu_step import build_model, device_masks, device_pool_winners, iteration  # noqa: E402


def main():
    from pacingpseudo_amd.optim import FusedAdam
    B = int(os.environ.get('B', '32'))
    args = O.full_flags()
    torch.manual_seed(1)
    model = build_model(args)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batch = O.synthetic_batch(B, 256, 256, seed=0)
    opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    rec, grads = iteration(model, opt, batch, args, 0)
    O.MASKS, O.POOLS = device_masks(model), device_pool_winners(model)
    try:
        _, g32, _ = O.train_step({k: v.clone() for k, v in sd.items()}, batch, 0, args, True)
        sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
        _, g64, _ = O.train_step(sd64, b64, 0, args, True)
    finally:
        O.MASKS = O.POOLS = None
    rows = []
    for k, v in g64.items():
        if v is None or G.is_bias_before_bn(k):
            continue
        h = grads[k].double().cpu().numpy()
        rows.append(dict(key=k, hip_o32=G.rel_err(h, g32[k].double().numpy()), hip_o64=G.rel_err(h, v.numpy()),
                         o32_o64=G.rel_err(g32[k].double().numpy(), v.numpy())))
    rows.sort(key=lambda r: -r['hip_o32'])
    for r in rows[:16]:
        print(f"{r['key']:58s} hip:o32 {r['hip_o32']:.2e}  hip:o64 {r['hip_o64']:.2e}  o32:o64 {r['o32_o64']:.2e}")
    print('max over parameters: hip:o32 %.2e  hip:o64 %.2e  o32:o64 %.2e' % (max(r['hip_o32'] for r in rows),
          max(r['hip_o64'] for r in rows), max(r['o32_o64'] for r in rows)))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(dict(batch_per_view=B, lazy_bn=os.environ.get('PP_LAZY_BN', '1'), rows=rows),
              open(os.path.join(ROOT, 'gpurun_out', 'grad_noise_b32.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
