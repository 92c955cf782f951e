"""HIP path against the round-3 reference vectors (tests/golden/upper_bound.npz, captured from the reference by
tests/golden/make_golden_r3.py): the soft Dice loss kernels (losses/losses.py:147-162) and the fully supervised trainer's
iteration (upper_bound_chaos.py:156-171) -- logits, both losses, every gradient, and Adam's first update."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import _golden as G  # noqa: E402
from tests.test_gpu_step import TOL_OUT, check_grads  # noqa: E402


@pytest.fixture(scope='module')
def ub():
    return G.load('upper_bound')


@pytest.mark.parametrize('i', [0, 1, 2])
def test_dice_loss_kernels_against_reference_values(ub, i):
    from pacingpseudo_amd.losses.losses import dice_loss_fn
    z = torch.from_numpy(ub[f'dice{i}/logits']).cuda().requires_grad_(True)
    loss = dice_loss_fn(z, torch.from_numpy(ub[f'dice{i}/onehot']).cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(ub[f'dice{i}/loss'])) < 2e-6
    ref = ub[f'dice{i}/grad']
    got = z.grad.cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-5 * float(np.abs(ref).max()))


def test_upper_bound_iterations_against_reference_vectors(ub):
    from pacingpseudo_amd.losses.losses import dice_loss_fn, partial_cross_entropy_loss
    from pacingpseudo_amd.models import UNet
    from pacingpseudo_amd.optim import FusedAdam
    from pacingpseudo_amd.utils import poly_lr_decay
    for it, ep in enumerate([0, 3]):
        net = UNet(input_ch=1, init_ch=4, max_ch=32, num_classes=5, output_stride=8, is_stride_conv=False,
                   is_trans_conv=False, elab_end_points=True)
        sd = G.sub(ub, 'ub/init/') if it == 0 else G.sub(ub, f'ub/step{it - 1}/post/')
        net.load_state_dict(G.to_state(sd))
        net = net.cuda()
        opt = FusedAdam(net.parameters(), lr=1e-4, weight_decay=3e-4)
        opt, lr = poly_lr_decay(opt, ep, 400, 1e-4)
        assert abs(lr - float(ub[f'ub/step{it}/lr'])) < 1e-12
        image = torch.from_numpy(ub[f'ub/step{it}/in/image']).cuda()
        label = torch.from_numpy(ub[f'ub/step{it}/in/label']).cuda()
        logits = net(image)['segmentation/logits']
        loss_ce = partial_cross_entropy_loss(logits, label.argmax(1).long(), 5)
        loss_dice = dice_loss_fn(logits, label)
        opt.zero_grad()
        (loss_ce + loss_dice).backward()
        assert G.rel_err(logits.detach().double().cpu().numpy(), ub[f'ub/step{it}/logits']) < TOL_OUT
        assert abs(float(loss_ce.detach()) - float(ub[f'ub/step{it}/loss_ce'])) < 1e-5
        assert abs(float(loss_dice.detach()) - float(ub[f'ub/step{it}/loss_dice'])) < 1e-5
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        # loose: raw reference gradients (LeakyReLU kink / pool tie choices may differ for single elements)
        worst = check_grads(grads, G.sub(ub, f'ub/step{it}/grad/'), True, tag=f'upper bound step {it} ', tol=1e-1)
        print(f'upper-bound step {it}: worst gradient rel err vs the reference {worst}')
        if it == 0:
            # Adam's first update with the reference's gradients written into the slab: post-step weights to rounding
            for k, p in net.named_parameters():
                p.grad.copy_(torch.from_numpy(np.array(ub[f'ub/step0/grad/{k}'])).cuda())
            opt.step()
            torch.cuda.synchronize()
            for k, p in net.named_parameters():
                np.testing.assert_allclose(p.detach().cpu().numpy(), ub[f'ub/step0/post/{k}'], rtol=0, atol=3e-7, err_msg=k)
