"""Per-tensor gradient error of the full-width model vs the kink-aligned oracle (GPU box)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pacing_oracle as O
from tests import _golden as G
from tests.test_gpu_step import build_model, iteration, oracle_with_device_branches
from pacingpseudo_amd.optim import FusedAdam
flags = sys.argv[1] if len(sys.argv) > 1 else 'control'
over = dict(do_loss_ent=True, do_decoder_consistency=True, do_aux_path=True, do_memory=True) if flags == 'full' else {}
args = O.default_args(**over)
torch.manual_seed(1)
model = build_model(args)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
batch = O.synthetic_batch(2, 64, 64, seed=3, keep=0.05)
opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
for step in range(2):
    sd0 = {k: v.clone() for k, v in sd.items()}
    rec, grads = iteration(model, opt, batch, args, 0)
    out, og, _ = oracle_with_device_branches(model, sd0, batch, 0, args, True)
    errs = []
    for k, v in og.items():
        if v is None or G.is_bias_before_bn(k):
            continue
        errs.append((G.rel_err(grads[k].double().cpu().numpy(), v.numpy()), k))
    errs.sort(reverse=True)
    print(f'step {step}: logits err {G.rel_err(rec["segmentation/logits"].double().cpu().numpy(), out["segmentation/logits"].numpy()):.2e}; '
          f'worst grads: ' + ', '.join(f'{e:.1e} {k.replace("backbone.", "").replace(".conv_block.conv_layer", ".c")[-28:]}' for e, k in errs[:6]),
          f'| median {np.median([e for e, _ in errs]):.1e}')
    sd.update({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
