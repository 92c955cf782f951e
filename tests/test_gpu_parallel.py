"""The real data-parallel engine path on the GPU box: two ranks share the one MI355X (gloo transport, so RCCL's
one-rank-per-device rule does not apply) and must reproduce the single-process step on the concatenated batch.
BatchNorm in eval mode (the reference's state from epoch 1 on), full flags including the rank-0 memory bank."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import pacing_oracle as O  # noqa: E402
from tests import _golden as G  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, out_path, bn_train=False, backend='gloo', storage='fp32'):
    dev = rank if backend == 'nccl' else 0
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(dev), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from pacingpseudo_amd import parallel
    from tests.test_gpu_step import build_model
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == 'nccl':
            parallel.init_from_env('nccl')
        else:
            dist.init_process_group('gloo')
    args = O.full_flags(init_ch=8, max_ch=64, hid_ch=16, feat_ch=[64, 64])
    size = 64
    if storage == 'fp16':                  # the 16-bit kernels need the real channel widths (>= 32 outputs per layer)
        args = O.full_flags()
        args.storage = 'fp16'
        size = 128
    torch.manual_seed(1)
    model = build_model(args)
    sd = model.state_dict()
    g = torch.Generator().manual_seed(7)
    for k, v in sd.items():                         # non-trivial running statistics and a visited memory bank
        if k.endswith('running_mean'):
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
        elif k.endswith('running_var'):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
        elif k.endswith('memory_bank'):
            v.copy_(torch.randn(v.shape, generator=g))
    if world > 1:
        parallel.attach(model, sync_bn=bn_train)
    if not bn_train:
        model.eval()
    full = O.synthetic_batch(4, size, size, seed=11, keep=0.06)
    full['valid_mask'][0, :, :9] = 0
    nloc = 4 // world
    batch = {k: v[rank * nloc:(rank + 1) * nloc].cuda() for k, v in full.items() if k != 'label'}
    out = model(batch, mode='train', step=37)
    w = O.loss_weights(args, 37)
    total = sum(out[k] * wt for k, wt in w.items())
    total.backward()
    torch.cuda.synchronize()
    res = dict(losses={k: float(out[k]) for k in w}, grads=model.flat.grads.cpu(),
               bank=model.state_dict()['aux_path.memory_bank'].cpu(),
               buffers={k: v.cpu() for k, v in model.state_dict().items() if 'running' in k or 'num_batches' in k})
    if rank == 0:
        torch.save(res, out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch(world, out_path, bn_train=False, backend='gloo', storage='fp32'):
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, out_path, bn_train, backend, storage)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f'rank process exit code {p.exitcode}'
    return torch.load(out_path)


@pytest.mark.timeout(1500)
def test_two_ranks_equal_one_process(tmp_path):
    one = _launch(1, str(tmp_path / 'one.pt'))
    two = _launch(2, str(tmp_path / 'two.pt'))
    for k, v in one['losses'].items():
        assert abs(two['losses'][k] - v) < 1e-5 * max(1.0, abs(v)), (k, two['losses'][k], v)
    assert G.rel_err(two['bank'].numpy(), one['bank'].numpy()) < 1e-6
    e = G.rel_err(two['grads'].numpy(), one['grads'].numpy())
    assert e < 2e-4, f'all-reduced gradient slab differs from the single-process one: {e:.3e}'


@pytest.mark.timeout(1500)
def test_two_ranks_equal_one_process_in_16_bit_storage(tmp_path):
    """`--storage fp16` data-parallel: the loss-scaled fp32 gradient slabs are all-reduced, THEN divided by the scale
    (ConsistencyRegulr._run_backward).  Eval-mode BatchNorm makes every sample's fp16 activations independent of how the batch
    is split, so two ranks must reproduce one process up to the summation order of the fp32 reductions."""
    one = _launch(1, str(tmp_path / 'one.pt'), storage='fp16')
    two = _launch(2, str(tmp_path / 'two.pt'), storage='fp16')
    for k, v in one['losses'].items():
        assert abs(two['losses'][k] - v) < 1e-5 * max(1.0, abs(v)), (k, two['losses'][k], v)
    assert G.rel_err(two['bank'].numpy(), one['bank'].numpy()) < 1e-6
    e = G.rel_err(two['grads'].numpy(), one['grads'].numpy())
    assert e < 2e-4, f'all-reduced gradient slab differs from the single-process one: {e:.3e}'
    assert float(one['grads'].abs().max()) < 1e3            # the loss scale is gone from the slab


@pytest.mark.timeout(1500)
def test_sync_bn_two_ranks_equal_one_process_in_16_bit_storage(tmp_path):
    """Train-mode BatchNorm with global statistics (`attach(sync_bn=True)`) in the 16-bit storage mode: the split statistics /
    backward-sum kernels of the `_h16` build, all-reduced in fp64, must give two ranks the single process's step."""
    one = _launch(1, str(tmp_path / 'one.pt'), bn_train=True, storage='fp16')
    two = _launch(2, str(tmp_path / 'two.pt'), bn_train=True, storage='fp16')
    for k, v in one['losses'].items():
        assert abs(two['losses'][k] - v) < 1e-4 * max(1.0, abs(v)), (k, two['losses'][k], v)
    for k, v in one['buffers'].items():
        if 'num_batches' in k:
            assert torch.equal(two['buffers'][k], v), k
        else:
            assert G.rel_err(two['buffers'][k].numpy(), v.numpy()) < 1e-4, k
    # The gradients are compared as directions, not element-wise: the single process takes its statistics from the fp32
    # accumulators of the convolution epilogues, the ranks from the stored fp16 z (pp_bn_stats_sums) -- 1e-4-relative
    # differences in mean / variance that move fp16 roundings and LeakyReLU branch decisions downstream (tests/test_gpu_h16.py
    # explains the size: sqrt(flip fraction) per layer); a wiring error would show as a cosine far from 1.
    a, b = one['grads'].double(), two['grads'].double()
    cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
    rel = float((a - b).norm() / a.norm())
    print('sync-BN 16-bit storage: gradient slab cosine', cos, 'relative L2', rel)
    assert cos > 0.95 and rel < 0.3, (cos, rel)          # measured 0.981 / 0.197


@pytest.mark.timeout(1500)
def test_sync_bn_two_ranks_equal_one_process(tmp_path):
    """BatchNorm in TRAIN mode (the reference's epoch 0) with attach(sync_bn=True): batch statistics, running buffers,
    losses and the summed gradient slab of two ranks equal ONE process on the concatenated batch
    (models/unet.py:189 normalises over the whole batch; SURVEY.md 8(e) coupling A)."""
    one = _launch(1, str(tmp_path / 'one.pt'), bn_train=True)
    two = _launch(2, str(tmp_path / 'two.pt'), bn_train=True)
    for k, v in one['losses'].items():
        assert abs(two['losses'][k] - v) < 2e-5 * max(1.0, abs(v)), (k, two['losses'][k], v)
    for k, v in one['buffers'].items():
        if 'num_batches' in k:
            assert torch.equal(two['buffers'][k], v), k
        else:
            assert G.rel_err(two['buffers'][k].numpy(), v.numpy()) < 1e-5, k
    assert G.rel_err(two['bank'].numpy(), one['bank'].numpy()) < 1e-5
    e = G.rel_err(two['grads'].numpy(), one['grads'].numpy())
    assert e < 5e-4, f'sync-BN gradient slab differs from the single-process one: {e:.3e}'


@pytest.mark.timeout(1500)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one GPU per rank (this box has one)')
def test_two_ranks_rccl(tmp_path):
    """The same equalities over RCCL (backend nccl), one rank per GPU: runs wherever two devices are visible."""
    one = _launch(1, str(tmp_path / 'one.pt'))
    two = _launch(2, str(tmp_path / 'two.pt'), backend='nccl')
    for k, v in one['losses'].items():
        assert abs(two['losses'][k] - v) < 1e-5 * max(1.0, abs(v)), (k, two['losses'][k], v)
    assert G.rel_err(two['grads'].numpy(), one['grads'].numpy()) < 2e-4
    sync1 = _launch(1, str(tmp_path / 'one_t.pt'), bn_train=True)
    sync2 = _launch(2, str(tmp_path / 'two_t.pt'), bn_train=True, backend='nccl')
    assert G.rel_err(sync2['grads'].numpy(), sync1['grads'].numpy()) < 5e-4


def test_bench_contract_with_two_ranks(tmp_path):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one JSON line from rank 0).
    Two ranks share the one GPU through gloo (PP_DIST_BACKEND / PP_SHARE_GPU: RCCL refuses two ranks on one device), so
    everything but the collective library itself is the code an 8-GPU run executes.  Regression: a collective inside a
    rank-0-only block hung every multi-rank run (r02)."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PP_DIST_BACKEND='gloo', PP_SHARE_GPU='1', PP_HANG_DUMP='240')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(root, 'bench.py'),
                        '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2', '--size', '64'],
                       env=env, cwd=root, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                     # rank 0 only
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['rccl_world_size'] == 2 and j['collective_backend'] == 'gloo'
    assert j['config']['global_batch'] == 4 and j['scaling'] == 'weak' and j['value'] > 0
    assert 'cpu_baseline' not in j                             # rank 0 at N = 1 only


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it (the way the driver calls `--gpus 1`): bench.py must start the
    two ranks itself (a torch.distributed.run child, before the parent touches the GPU), relay rank 0's single JSON line and
    the exit code -- never print a 1-GPU number under an `n_gpus: 2` request.  Same one-GPU rehearsal
    transport as above (gloo, both ranks on device 0)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(PP_DIST_BACKEND='gloo', PP_SHARE_GPU='1', PP_HANG_DUMP='240')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--batch', '2', '--size', '64'], env=env, cwd=root, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['rccl_world_size'] == 2 and j['config']['global_batch'] == 4 and j['value'] > 0
    assert len(j['step_ms']['all']) == 2 and j['step_ms']['min'] > 0
    # without the rehearsal switch a request for more GPUs than the box has must fail loudly, not shrink
    env.pop('PP_SHARE_GPU')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(torch.cuda.device_count() + 1), '--steps', '1',
                        '--warmup', '0'], env=env, cwd=root, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'visible' in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith('{')]


@pytest.mark.parametrize('sync_bn', [False, True])
def test_bench_on_a_one_rank_rccl_group(sync_bn):
    """RCCL itself, on the one GPU this box has: PP_FORCE_DIST=1 makes bench.py build its process group for ONE rank (backend nccl =
    RCCL, `device_id` init, the second communicator for the small exchanges) and run every collective of the N > 1 step through
    the library -- the packed loss-denominator all-reduce, the bank broadcast, the six gradient buckets as asynchronous
    all-reduces joined before the optimizer, SyncBN's packed sums (second case), the barriers and the MAX reduction of the
    timing.  With one rank each collective is an identity, so the line must equal a plain run's; what this covers is that the
    calls are legal for RCCL (dtypes, devices, stream / async semantics) before an 8-GPU node ever sees them."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'PP_DIST_BACKEND', 'PP_SHARE_GPU')}
    env.update(PP_FORCE_DIST='1', MASTER_PORT=str(_free_port()), PP_HANG_DUMP='240')
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '2', '--size', '64',
           '--no-cpu-baseline', '--no-bn-eval'] + (['--sync-bn'] if sync_bn else [])
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert j['collective_backend'] == 'nccl' and j['rccl_world_size'] == 1 and j['n_gpus'] == 1
    env.pop('PP_FORCE_DIST')
    r0 = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=280)
    assert r0.returncode == 0, r0.stdout[-2000:] + r0.stderr[-4000:]
    j0 = json.loads([l for l in r0.stdout.splitlines() if l.startswith('{')][-1])
    assert j0['collective_backend'] is None
    # (SyncBN takes its statistics from a separate pass over z instead of the convolution epilogue: another summation order, and
    # the line's loss is read after eight Adam steps on a 2-image batch, which turn a 1e-6 difference into 1e-3 -- DESIGN.md
    # section 4 "Chaos, not bias"; the one-step equality of SyncBN with the single process is test_sync_bn_two_ranks_*)
    tol = 2e-2 if sync_bn else 1e-5
    assert abs(j['final_loss'] - j0['final_loss']) <= tol * max(1.0, abs(j0['final_loss'])), (j['final_loss'], j0['final_loss'])
