import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pacing_oracle as O
from tests import _golden as G
from tests.test_gpu_step import build_model, iteration
from pacingpseudo_amd import engine as E
from pacingpseudo_amd.optim import FusedAdam
args = O.full_flags()
batch = O.synthetic_batch(2, 128, 128, seed=7, keep=0.05)
def run(flags):
    E.LAZY_BN, E.LAZY_WINO, E.LAZY_BILINEAR = flags
    torch.manual_seed(1)
    model = build_model(args)
    opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
    rec, grads = iteration(model, opt, batch, args, 0)
    return rec, grads, sorted(k for k, v in model.engine.last_plan.lazy_out.items() if v)
ref = run((False, False, False))
for flags in [(True, False, False), (True, True, False), (True, False, True), (True, True, True)]:
    rec, grads, lz = run(flags)
    errs = sorted(((G.rel_err(grads[k].double().cpu().numpy(), v.double().cpu().numpy()), k) for k, v in ref[1].items()
                   if v is not None and not G.is_bias_before_bn(k)), reverse=True)
    print(flags, 'lazy layers:', [l.replace('.conv_block.conv_layer', '.c').replace('_block', '') for l in lz])
    print('   worst:', [(f'{e:.1e}', k.replace('backbone.', '').replace('.conv_block.conv_layer', '.c')) for e, k in errs[:4]])
