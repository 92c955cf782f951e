"""Pin the CPU oracle against vectors captured from the reference itself (tests/golden/*.npz)."""
import numpy as np
import pytest
import torch

from oracle import pacing_oracle as O
from tests import _golden as G

TOL_OUT = 2e-5      # relative to max |ref| of the tensor
TOL_GRAD = 2e-4


@pytest.mark.parametrize('name', list(G.CASES))
def test_sequence_matches_reference(name):
    torch.set_num_threads(4)
    d = G.load(name)
    args = G.case_args(name)
    _, epochs = G.CASES[name]
    sd = G.to_state(G.sub(d, 'init/'))
    assert sorted(sd) == sorted(O.init_state(args)), 'state_dict key layout'
    training = True
    slim_adam = O.AdamState()
    prev = epochs[0]
    full_post = f'step0/post/{O.conv_layer_prefixes(args)[3]}.conv.weight' in d
    for i, ep in enumerate(epochs):
        if ep != prev:
            training = False            # model.eval() after the first epoch, never undone
        prev = ep
        assert int(d[f'step{i}/bn_training']) == int(training)
        lr = O.lr_at('poly', ep, args.epoch, args.lr)
        assert abs(lr - float(d[f'step{i}/lr'])) < 1e-12
        if i > 0:
            # restart every iteration from the reference's own post-step state: Adam's first updates
            # are sign(g)*lr, so rounding noise in near-zero gradients must not be allowed to drift
            assert full_post
            sd = G.to_state(G.sub(d, f'step{i - 1}/post/'))
        out, grads, total = O.train_step(sd, G.batch_of(d, i), ep, args, training,
                                         None if full_post else slim_adam, lr)
        ref_out = G.sub(d, f'step{i}/out/')
        for k, v in ref_out.items():
            if k == 'total_loss':
                assert abs(total - float(v)) <= TOL_OUT * max(1.0, abs(float(v))), k
            else:
                assert k in out, k
                assert G.rel_err(out[k].numpy(), v) < TOL_OUT, (k, G.rel_err(out[k].numpy(), v))
        for k, v in G.sub(d, f'step{i}/grad/').items():
            assert grads[k] is not None, k
            if training and G.is_bias_before_bn(k):
                # mathematically zero (train-mode BN removes the mean): both sides hold rounding noise
                assert float(grads[k].abs().max()) < 2e-5 and np.max(np.abs(v)) < 2e-5, k
            else:
                assert G.rel_err(grads[k].numpy(), v) < TOL_GRAD, (k, G.rel_err(grads[k].numpy(), v))
        # buffers the forward pass mutates: BN running stats, num_batches_tracked, memory bank
        for k, v in G.sub(d, f'step{i}/post/').items():
            if k in grads:
                continue            # trained weights: see test_adam_restatement
            got = sd[k].detach().numpy()
            if got.dtype.kind == 'i':
                assert np.array_equal(got, v), k
            else:
                assert G.rel_err(got, v) < TOL_OUT, (k, G.rel_err(got, v))
    # validation forward (train_chaos.py:370-392)
    sd0 = G.to_state(G.sub(d, f'step{len(epochs) - 1}/post/')) if full_post else sd
    with torch.no_grad():
        vo = O.consistency_forward(sd0, G.batch_of(d, 0), 'val', None, args, training=False)
    assert sorted(k for k in vo if not k.startswith('_')) == sorted(d['val/keys'].tolist())
    # slim fixtures do not carry the post-Adam weights: there the oracle's own Adam ran, and its
    # sign(g)*lr first step on noise-level gradients may differ from the reference's -> looser bound
    tol_val = TOL_OUT if full_post else 5e-3
    assert G.rel_err(vo['segmentation/logits'].numpy(), d['val/logits']) < tol_val
    assert G.rel_err(vo['loss_pce'].numpy(), d['val/loss_pce']) < tol_val
    if not full_post:
        return
    sm = torch.softmax(vo['segmentation/logits'], 1).numpy()
    lab = G.batch_of(d, 0)['label'].numpy()
    dice = np.asarray([O.compute_dice(sm[n], lab[n]) for n in range(sm.shape[0])])
    assert np.allclose(dice, d['val/dice'], atol=1e-12, equal_nan=True)


def test_argmax_pseudo_labels_bit_exact():
    d = G.load('full_seq')
    args = G.case_args('full_seq')
    sd = G.to_state(G.sub(d, 'init/'))
    with torch.no_grad():
        out = O.consistency_forward(sd, G.batch_of(d, 0), 'train', 0, args, training=True)
    # same aten ops -> the arg-max masks of the reference logits and of the oracle logits must agree
    ref = np.argmax(d['step0/out/segmentation/logits'], 1)
    got = out['segmentation/logits'].argmax(1).numpy()
    assert np.array_equal(ref, got)


def test_memory_bank_quirks():
    """Only sample 0 updates the bank; a class without scribble in sample 0 keeps its row."""
    d = G.load('full_seq')
    bank0 = d['step0/post/aux_path.memory_bank'][:, :, 0, 0]
    assert np.all(bank0[4] == 0) and np.all(np.abs(bank0[:4]).sum(1) > 0)
    bank1 = d['step1/post/aux_path.memory_bank'][:, :, 0, 0]
    assert np.all(bank1[4] == 0) and not np.allclose(bank0[:4], bank1[:4])


@pytest.mark.parametrize('name', ['full_seq', 'control_seq', 'variant_kl', 'strideconv8'])
def test_adam_restatement(name):
    """Feed the reference's own gradients through the oracle's Adam: post-step weights must match
    torch.optim.Adam(lr, weight_decay) to rounding (train_chaos.py:219,313-315)."""
    d = G.load(name)
    args = G.case_args(name)
    _, epochs = G.CASES[name]
    sd = G.to_state(G.sub(d, 'init/'))
    adam = O.AdamState()
    for i, ep in enumerate(epochs):
        g = {k: torch.from_numpy(np.array(v)) for k, v in G.sub(d, f'step{i}/grad/').items()}
        keys = O.trainable_keys(sd)
        adam.step(sd, {k: g.get(k) for k in keys}, float(d[f'step{i}/lr']), args.wd)
        for k in keys:
            ref = d[f'step{i}/post/{k}']
            assert np.max(np.abs(sd[k].numpy() - ref)) <= 2e-7 * max(1.0, np.max(np.abs(ref))) + 1e-9, (i, k)
    if name == 'control_seq':
        # aux-path weights never receive a gradient in the Control session -> Adam leaves them alone
        assert np.array_equal(sd['aux_path.fc_cls.1.weight'].numpy(), d['init/aux_path.fc_cls.1.weight'])
