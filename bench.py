#!/usr/bin/env python
"""Headline benchmark: training images/sec of the PacingPseudo step (256x256, 5 classes) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: PacingPseudo full flags (entropy + decoder consistency + aux path + memory),
synthetic 256x256x1 slices, 5 classes, batch 32 per GPU (weak scaling: global batch 32*N), fp32, random-init
weights (seed 1).  A "step" is one full iteration of train_chaos.py:263-315: both siamese passes, all five losses,
backward, gradient all-reduce (N>1) and the Adam update; the batch is resident in HBM before the clock starts.
BatchNorm runs in train mode (batch statistics + running-stat updates), i.e. the state the reference trains in
during its first epoch and the more expensive of its two modes; `bn_eval_images_per_sec` reports the eval-mode
step the reference uses from epoch 1 on (train_chaos.py:370).

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around every launch of the dominant
matrix-core kernel families during the timed steps; `cpu_baseline` times the CPU oracle
(the reference path restated on PyTorch-CPU) on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 256 FLOP/clk x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2516.6     # v_mfma_f32_32x32x16_f16 dense: 256 CUs x 4096 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md: ~2.5 PF)
FLOP_PER_IMAGE_FULL = 348.2e9     # 3 x (2 x 57.437 + 1.208) GFLOP, conv MACs only
BYTES_PER_IMAGE_FULL = 1.10e9     # algorithmic HBM bytes per image, full flags


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--session', default='Experiment', choices=['Experiment', 'Control'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-bn-eval', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--cpu-steps', type=int, default=5)
    ap.add_argument('--prof-timed', default='dominant', choices=['dominant', 'matrix', 'none'],
                    help='which kernel families get HIP events around every launch INSIDE the timed region: the dominant matrix family '
                         '(chosen in a profiled warm-up step; default), all matrix families (rounds 2-4), or none')
    ap.add_argument('--bn-mode', default='train', choices=['train', 'eval'],
                    help="BatchNorm mode of the TIMED step: train (default, the headline: the reference's epoch 0, the more expensive mode) or "
                         "eval (running statistics: the reference's state in 399 of its 400 epochs, train_chaos.py:370) -- the profile "
                         'scripts use eval to collect the steady-state kernel statistics; the line then says so in config.workload')
    ap.add_argument('--sync-bn', action='store_true',
                    help='N > 1: train-mode BatchNorm statistics over the GLOBAL batch (one packed all-reduce per BN call, as the '
                         "reference's single-process batch would see them); default: per-rank statistics (stated in the JSON line)")
    return ap.parse_args()


def build(args_model, device):
    import torch
    from pacingpseudo_amd.models import ConsistencyRegulr
    torch.manual_seed(1)
    a = args_model
    model = ConsistencyRegulr(
        kwargs_unet=dict(input_ch=a.input_ch, init_ch=a.init_ch, max_ch=a.max_ch, num_classes=a.num_classes,
                         output_stride=a.output_stride, is_stride_conv=False, is_trans_conv=False,
                         elab_end_points=True),
        kwargs_aux_path=dict(num_classes=a.num_classes, feat_stage=a.feat_stage, feat_ch=a.feat_ch, hid_ch=a.hid_ch,
                             aux_drop_prob=a.aux_drop_prob, do_memory=a.do_memory, max_step=a.epoch,
                             update_momentum=a.update_momentum, ensemble_mode=a.ensemble_mode),
        args_parser=a)
    return model.to(device)


def assemble_loss(out, a, epoch):
    """The loss assembly of train_chaos.py:273-310."""
    from pacingpseudo_amd.losses.losses import weighted_loss_sum
    from pacingpseudo_amd.utils import gaussian_ramp_up
    terms, weights = [out['loss_pce']], [1.0]
    if a.do_loss_ent:
        terms.append(out['loss_ent']); weights.append(gaussian_ramp_up(epoch, a.loss_ent_weight, scale=a.ramp_up_scale))
    if a.do_decoder_consistency:
        terms.append(out['loss_cr']); weights.append(gaussian_ramp_up(epoch, a.loss_cr_weight, scale=a.ramp_up_scale))
    if a.do_aux_path:
        terms.append(out['loss_aux_cls']); weights.append(a.loss_aux_weight)
        if a.do_memory:
            terms.append(out['loss_memory']); weights.append(a.loss_memory_weight)
    return weighted_loss_sum(terms, weights)        # as pacingpseudo_amd/train.py assembles it: one launch each way


def train_iteration(model, opt, batch, a, epoch):
    """train_chaos.py:272-315 (meters kept on the device: no per-loss .item() host sync)."""
    out = model(batch, mode='train', step=epoch)
    loss = assemble_loss(out, a, epoch)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


def traffic_per_launch(names):
    """HBM bytes per launch of EXACTLY the named kernels (template arguments ignored) from the newest committed
    rocprofv3 PMC summary (profiles/*hbm_traffic_per_launch.json, written by scripts/profile_bench.sh: FETCH_SIZE x2 +
    WRITE_SIZE with the gfx950 corrections).  None when the profile has no row for them -- never a neighbour's."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', '*hbm_traffic_per_launch.json'))
                   if '_h16_' not in f and '_evalbn_' not in f)        # the train-mode fp32 step's profile (what `value` times)
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    n = b = 0.0
    for k, v in d.items():
        if k.split('<')[0].strip() in names:
            n += v['launches']
            b += v['launches'] * v['hbm_bytes_per_launch']
    return (round(b / n) if n else None), os.path.basename(files[-1])


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _time_cpu_steps(O, a, B, size, threads, warm, steps):
    import torch
    torch.set_num_threads(threads)
    sd = O.init_state(a, seed=1)
    batch = O.synthetic_batch(B, size, size, a.num_classes, seed=0)
    adam = O.AdamState()
    for _ in range(warm):
        O.train_step(sd, batch, 0, a, True, adam)             # warm-up (oneDNN primitive creation, allocator)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        O.train_step(sd, batch, 0, a, True, adam)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]                                    # median step time


def cpu_baseline(a, B, size, steps):
    """The oracle (CPU restatement of the reference path, kind "port") timed on this box's host cores, as SURVEY.md
    8(d) plans it: 2 warm-up + >= 5 timed steps, median; full flags and Control (BASELINE config 1) at batch 8; the
    thread count that is best on this box (sweep recorded below) and a 1-thread figure on a smaller sample."""
    from oracle import pacing_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # thread sweep on the MI355X box (2 x EPYC 9575F, tests/studies/cpu_threads.py, r01): 16 -> 2.08, 32 -> 2.24,
    # 64 -> 1.41, 128 -> 0.70 images/s; the oracle gets its best setting
    cores = max(1, min(avail, 32))
    full = _time_cpu_steps(O, a, B, size, cores, 2, steps)
    ctl = _time_cpu_steps(O, O.default_args(), B, size, cores, 2, steps)
    one = _time_cpu_steps(O, a, 2, size, 1, 1, 1)              # one core: 1 warm-up + 1 timed step of batch 2
    b32 = _time_cpu_steps(O, a, 32, size, cores, 1, 1) if a.do_aux_path else None   # the benchmark's own batch: 1 + 1 steps
    return dict(value=round(B / full, 3), unit='images/sec', cores=cores, kind='port', cpu_model=_cpu_model(),
                cores_available=avail,
                sample=f'median of {steps} full training steps (fwd+losses+bwd+Adam) of batch {B} at {size}x{size}, '
                       f'{"full flags" if a.do_aux_path else "Control"}, after 2 warm-up steps; {full:.2f} s/step',
                batch32_images_per_sec=round(32 / b32, 3) if b32 else None,
                batch32_sample=(f'same flags at the benchmark batch of 32: 1 timed step after 1 warm-up; {b32:.1f} s/step' if b32 else None),
                control_batch8_images_per_sec=round(B / ctl, 3),
                control_sample=f'same protocol, --session=Control (BASELINE.json configs[0]: UNet + partial CE, batch {B}); {ctl:.2f} s/step',
                one_thread_images_per_sec=round(2 / one, 4),
                one_thread_sample=f'1 thread, full flags, batch 2, 1 timed step after 1 warm-up; {one:.1f} s/step')


# matrix-core kernel families: (profiler kind, kernels it times, peak of the MFMA instruction it issues, what)
FAMILIES = [
    ('wino_gemm_f16x3', ('wino_gemm_psp_kernel', 'wino_gemm_ps_kernel'), PEAK_F16_MFMA_TFLOPS, 'fwd + dgrad, Winograd-domain GEMM on pre-split fp16 operands, persistent over its tiles'),
    ('conv_halo_f16x3', ('conv3x3_halo2_f16x3_kernel', 'conv3x3_halo_f16x3_kernel'), PEAK_F16_MFMA_TFLOPS, 'fwd + dgrad of the narrow layers, persistent halo tiles, split-fp16 operands'),
    ('conv_f16x3', ('conv3x3_igemm_f16x3_kernel',), PEAK_F16_MFMA_TFLOPS, 'fwd + dgrad, direct implicit GEMM, split-fp16 operands'),
    ('wino_wgrad_f16x3', ('wino_wgrad_gemm_f16x3_kernel',), PEAK_F16_MFMA_TFLOPS, 'weight gradient, Winograd domain, split-fp16 operands'),
    ('conv_wgrad_f16x3', ('conv3x3_wgrad_halo_mp_f16x3_kernel', 'conv3x3_wgrad_halo_f16x3_kernel'), PEAK_F16_MFMA_TFLOPS, 'weight gradient, direct (narrow layers), split-fp16 operands'),
    ('wino_gemm', ('wino_gemm_kernel',), PEAK_F32_MFMA_TFLOPS, 'fwd + dgrad, Winograd-domain GEMM, fp32 MFMA'),
    ('conv_igemm', ('conv3x3_igemm_kernel', 'conv3x3_halo_kernel', 'conv3x3_c4_fwd_kernel'), PEAK_F32_MFMA_TFLOPS, 'fwd + dgrad, direct, fp32 MFMA'),
    ('conv_wgrad', ('conv3x3_wgrad9_kernel', 'conv3x3_wgrad_kernel', 'conv3x3_c4_wgrad_kernel'), PEAK_F32_MFMA_TFLOPS, 'weight gradient, direct, fp32 MFMA'),
    ('wino_wgrad', ('wino_wgrad_gemm_kernel',), PEAK_F32_MFMA_TFLOPS, 'weight gradient, Winograd domain, fp32 MFMA'),
]

def family_table(prof, steps, traffic=True):
    """Rows of the matrix-core families of one event-profiled run of `steps` steps (pp_prof_* sums), most time first."""
    table = []
    for kind, names, peak, what in FAMILIES:
        v = prof.get(kind)
        if not v or not v['launches'] or v['ms'] <= 0:
            continue
        ex = v['flops'] / (v['ms'] * 1e-3) / 1e12
        table.append({'family': kind, 'what': what, 'kernels': list(names), 'ms_per_step': round(v['ms'] / steps, 3),
                      'launches_per_step': v['launches'] / steps, 'avg_launch_ms': round(v['ms'] / v['launches'], 4),
                      'executed_tflops': round(ex, 2), 'peak_tflops': peak, 'frac': round(ex / peak, 4),
                      'algorithmic_tflops': round(v['alg_flops'] / (v['ms'] * 1e-3) / 1e12, 2),
                      'algorithmic_bytes_per_launch': round(v['bytes'] / v['launches']),
                      'algorithmic_frac': round(v['alg_flops'] / (v['ms'] * 1e-3) / 1e12 / peak, 4),
                      'traffic': traffic_per_launch(names)[0] if traffic else None})
    table.sort(key=lambda r: -r['ms_per_step'])
    return table


def launch_ranks(cli) -> int:
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start N ranks, one per GPU, under
    torch.distributed.run -- as a CHILD process, before this process has touched the GPU (it never does: it only relays the
    exit code; rank 0's JSON line goes straight to the inherited stdout)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()          # counts devices without initialising HIP
    if have < cli.gpus and os.environ.get('PP_SHARE_GPU') != '1':
        print(f'[bench] --gpus {cli.gpus} but only {have} GPU(s) are visible; refusing to report a smaller run under that label '
              '(rehearsal on one GPU: PP_DIST_BACKEND=gloo PP_SHARE_GPU=1)', file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(cli.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f'[bench] --gpus {cli.gpus} without a launcher: starting {cli.gpus} ranks: {" ".join(cmd[1:9])} ...', file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def _dist(xs):
    """min / median / p90 / max of a list of per-step times (ms)."""
    v = sorted(xs)
    n = len(v)
    return dict(min=round(v[0], 3), median=round(v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2]), 3),
                p90=round(v[min(n - 1, int(0.9 * n))], 3), max=round(v[-1], 3))


def main():
    cli = parse()
    if cli.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(cli))
    if os.environ.get('PP_HANG_DUMP'):                 # debugging aid: dump every thread's stack after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['PP_HANG_DUMP']), exit=True)
    import torch
    import torch.distributed as dist
    from pacingpseudo_amd.data import default_args, full_flags, synthetic_batch
    from pacingpseudo_amd import parallel
    from pacingpseudo_amd._lib import lib, prof_collect
    from pacingpseudo_amd.optim import FusedAdam

    world, rank, local_rank = parallel.init_from_env('nccl')
    if world != cli.gpus and rank == 0:
        print(f'[bench] note: --gpus {cli.gpus} but WORLD_SIZE={world}; using WORLD_SIZE (n_gpus in the line = {world})', file=sys.stderr)
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    # the collective code path runs for N > 1 -- and for N = 1 under PP_FORCE_DIST=1 (a one-rank RCCL group: every collective the
    # 8-GPU run issues, on the real library, on a one-GPU box)
    dist_on = dist.is_available() and dist.is_initialized()

    a = full_flags() if cli.session == 'Experiment' else default_args()
    model = build(a, device)
    if dist_on:
        parallel.attach(model, sync_bn=cli.sync_bn)
    opt = FusedAdam(model.parameters(), lr=a.lr, weight_decay=a.wd)
    B, S = cli.batch, cli.size
    batch = {k: v.to(device) for k, v in synthetic_batch(B, S, S, a.num_classes, seed=rank).items() if k != 'label'}

    def sync():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    import gc
    from pacingpseudo_amd._lib import PROF_KINDS
    matrix_kinds = ('conv_igemm', 'conv_wgrad', 'wino_gemm', 'wino_wgrad', 'conv_f16x3', 'wino_gemm_f16x3',
                    'wino_wgrad_f16x3', 'conv_wgrad_f16x3', 'conv_halo_f16x3')
    matrix_mask = sum(1 << PROF_KINDS.index(k) for k in matrix_kinds)
    eval_main = cli.bn_mode == 'eval'
    ep_main = 1 if eval_main else 0                # the reference switches to eval-mode BatchNorm behind epoch 0 (train_chaos.py:370)
    model.eval() if eval_main else model.train()
    # HIP events (recorded inside the library on the launch stream) around the launches of the DOMINANT matrix-core family
    # during the timed region: that is what `roofline` is computed from.  Which family that is, is measured in the last warm-up
    # step (all matrix families timed there).  Every other family is timed in two extra, untimed steps afterwards (`kernels`,
    # `roofline.matrix_families`): two event records per launch are host work and queue packets, and rounds 2-4 paid for
    # ~160 of them per step inside the timed region (--prof-timed matrix restores that).
    timed_mask, dom_kind = 0, None
    for i in range(cli.warmup):
        last = i == cli.warmup - 1 and cli.prof_timed != 'none'
        if last:
            sync()
            lib.pp_prof_select(matrix_mask)
            lib.pp_prof_enable(1)
            prof_collect()
        train_iteration(model, opt, batch, a, ep_main)
        if last:
            sync()
            lib.pp_prof_enable(0)
            pw = prof_collect()
            # dominant = most event time among the families of the MAIN stream.  The weight-gradient families run on the second
            # stream under a CU budget, where a launch's event time includes the time it waits for CUs behind the critical chain
            # -- not a kernel duration (`single_stream` lists them alone on the chip).  Families within 10 % of the leader count
            # as tied and the tie goes to the Winograd GEMM: ONE kernel symbol (the top row of the committed rocprofv3 kernel
            # stats), where the halo family's time is the sum of two kernels
            side_kinds = ('conv_wgrad', 'wino_wgrad', 'wino_wgrad_f16x3', 'conv_wgrad_f16x3')
            from pacingpseudo_amd import engine as _eng0
            cand = [k for k in matrix_kinds if not (_eng0.WGRAD_STREAM and k in side_kinds)]
            top = max(pw[k]['ms'] for k in cand)
            tied = [k for k in cand if pw[k]['ms'] >= 0.9 * top]
            dom_kind = 'wino_gemm_f16x3' if 'wino_gemm_f16x3' in tied else max(tied, key=lambda k: pw[k]['ms'])
            per_step = sum(pw[k]['launches'] for k in (matrix_kinds if cli.prof_timed == 'matrix' else (dom_kind,)))
            timed_mask = matrix_mask if cli.prof_timed == 'matrix' else (1 << PROF_KINDS.index(dom_kind))
            lib.pp_prof_reserve(2 * per_step * cli.steps + 64)      # no hipEventCreate inside the timed region
    # one event per step on the main stream + the host clock after each step's enqueue: `step_ms` says whether a slow run was
    # one stall or twenty slow steps, `host_lead_ms` whether the GPU ever waited for the host
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(cli.steps + 1)]
    host_t = [0.0] * cli.steps
    gc.collect()
    gc.disable()                                   # no collector pause between the launches of a step
    sync()
    if timed_mask:
        lib.pp_prof_select(timed_mask)
        lib.pp_prof_enable(1)
        prof_collect()
    t0 = time.perf_counter()
    step_ev[0].record()
    for i in range(cli.steps):
        loss = train_iteration(model, opt, batch, a, ep_main)
        step_ev[i + 1].record()
        host_t[i] = time.perf_counter()
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    lib.pp_prof_enable(0)
    prof = prof_collect()
    gpu_t = [step_ev[0].elapsed_time(step_ev[i + 1]) for i in range(cli.steps)]            # ms since the start, per step END
    step_ms = [gpu_t[0]] + [gpu_t[i] - gpu_t[i - 1] for i in range(1, cli.steps)]
    enq_ms = [(host_t[0] - t0) * 1e3] + [(host_t[i] - host_t[i - 1]) * 1e3 for i in range(1, cli.steps)]
    # how far the host was ahead when it finished enqueueing step i: the GPU reached that point (end of step i) this much later
    lead_ms = [gpu_t[i] - (host_t[i] - t0) * 1e3 for i in range(cli.steps)]
    step_stats = dict(_dist(step_ms), slowest_step=int(max(range(cli.steps), key=lambda i: step_ms[i])),
                      all=[round(x, 3) for x in step_ms],
                      what='GPU time between consecutive per-step HIP events on the main stream inside the timed region')
    host_stats = dict(enqueue_ms=_dist(enq_ms), lead_ms_min=round(min(lead_ms), 3), lead_ms_median=_dist(lead_ms)['median'],
                      what='enqueue_ms: host wall time to enqueue one step; lead_ms: GPU completion of step i minus the host clock '
                           'when its enqueue finished (<= 0 would mean the GPU waited for the host)')
    lib.pp_prof_select((1 << 64) - 1)
    lib.pp_prof_enable(1)
    for _ in range(2):
        train_iteration(model, opt, batch, a, ep_main)
    sync()
    lib.pp_prof_enable(0)
    prof_all = prof_collect()
    if cli.prof_timed != 'matrix':       # families not timed inside the timed region: from the two untimed steps (per-step scale)
        for k in matrix_kinds:
            if not (timed_mask >> PROF_KINDS.index(k)) & 1:
                v = prof_all[k]
                prof[k] = dict(launches=v['launches'] * cli.steps / 2, ms=v['ms'] * cli.steps / 2, flops=v['flops'] * cli.steps / 2,
                               bytes=v['bytes'] * cli.steps / 2, alg_flops=v['alg_flops'] * cli.steps / 2, untimed=True)
    # The timed region runs the weight gradients on a second HIP stream beside the critical chain, so a kernel's event time
    # there includes the kernels it shares the chip with.  Two more untimed steps with the second stream off give every
    # family's time when it has the chip to itself (`single_stream` in the line; `roofline` itself stays the timed region's).
    from pacingpseudo_amd import engine as _engine
    prof_single = None
    if _engine.WGRAD_STREAM:
        _engine.WGRAD_STREAM = False
        try:
            train_iteration(model, opt, batch, a, ep_main)
            sync()
            lib.pp_prof_enable(1)
            prof_collect()
            for _ in range(2):
                train_iteration(model, opt, batch, a, ep_main)
            sync()
            lib.pp_prof_enable(0)
            prof_single = prof_collect()
        finally:
            _engine.WGRAD_STREAM = True
    final_loss = float(loss.detach())
    if dist_on:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = dt / cli.steps * 1e3
    value = B * world * cli.steps / dt

    bn_eval = None
    if not cli.no_bn_eval and not eval_main:
        # The reference's steady state: BatchNorm with running statistics from epoch 1 on (train_chaos.py:370, never undone) --
        # 399 of its 400 epochs.  Timed like the headline (same batch, whole iteration), then two event-profiled steps for the
        # family times, and two more with the weight gradients back on the main stream (every family alone on the chip).
        model.eval()
        for _ in range(2):
            train_iteration(model, opt, batch, a, 1)
        sync()
        n_eval = max(3, cli.steps // 2)
        t1 = time.perf_counter()
        for _ in range(n_eval):
            train_iteration(model, opt, batch, a, 1)
        sync()
        dte = time.perf_counter() - t1
        if dist_on:
            t = torch.tensor([dte], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dte = float(t)
        lib.pp_prof_select((1 << 64) - 1)
        lib.pp_prof_enable(1)
        prof_collect()
        for _ in range(2):
            train_iteration(model, opt, batch, a, 1)
        sync()
        lib.pp_prof_enable(0)
        pe2 = prof_collect()
        pe1 = None
        if _engine.WGRAD_STREAM:
            _engine.WGRAD_STREAM = False
            try:
                train_iteration(model, opt, batch, a, 1)
                sync()
                lib.pp_prof_enable(1)
                prof_collect()
                for _ in range(2):
                    train_iteration(model, opt, batch, a, 1)
                sync()
                lib.pp_prof_enable(0)
                pe1 = prof_collect()
            finally:
                _engine.WGRAD_STREAM = True
        tab_e = family_table(pe1 if pe1 is not None else pe2, 2, traffic=False)
        dom_e = tab_e[0] if tab_e else None
        bn_eval = dict(images_per_sec=round(B * world * n_eval / dte, 2), ms_per_step=round(dte / n_eval * 1e3, 3), steps=n_eval,
                       workload='the same step with BatchNorm in eval mode (running statistics): the reference from epoch 1 on, '
                                'train_chaos.py:370',
                       dominant_family=(dict(family=dom_e['family'], kernel=dom_e['kernels'][0], ms_per_step=dom_e['ms_per_step'],
                                             executed_tflops=dom_e['executed_tflops'], peak_tflops=dom_e['peak_tflops'],
                                             frac=dom_e['frac'], algorithmic_frac=dom_e['algorithmic_frac'],
                                             timed='alone on the chip (one stream)' if pe1 is not None else 'two streams')
                                        if dom_e else None),
                       families_ms_per_step={k: round(v['ms'] / 2, 3) for k, v in pe2.items() if v['launches']},
                       single_stream_families_ms_per_step=({k: round(v['ms'] / 2, 3) for k, v in pe1.items() if v['launches']}
                                                           if pe1 is not None else None),
                       profile='profiles/r06_evalbn_{kernel_stats.csv,hbm_traffic_per_launch.json,roofline_table.md} '
                               '(scripts/profile_bench.sh r06_evalbn --bn-mode eval)')
        model.train()

    # BASELINE configs[4] names a mixed-precision mode: the same step with fp16 OPERANDS in the forward / data-gradient products
    # of the halo-tile and Winograd kernels and in the Winograd weight-gradient GEMM (fp32 accumulation, fp32 tensors in HBM, fp32-grade
    # direct weight gradients).  Reported
    # beside the headline, never as `value`: it has no 1e-4 parity claim (tests/test_gpu_round3.py states its tolerance).
    mixed = None
    if not cli.no_bn_eval:
        model.train()
        lib.pp_set_matrix_products(1)
        try:
            for _ in range(2):
                train_iteration(model, opt, batch, a, 0)
            sync()
            n_mx = max(3, cli.steps // 2)
            t1 = time.perf_counter()
            for _ in range(n_mx):
                train_iteration(model, opt, batch, a, 0)
            sync()
            dtm = time.perf_counter() - t1
        finally:
            lib.pp_set_matrix_products(3)
        if dist_on:
            t = torch.tensor([dtm], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtm = float(t)
        mixed = dict(images_per_sec=round(B * world * n_mx / dtm, 2), ms_per_step=round(dtm / n_mx * 1e3, 3), steps=n_mx,
                     dtype='fp16 operands (hi parts only) in the forward / data-gradient products of the halo-tile and Winograd '
                           'kernels and in the Winograd weight-gradient GEMM, fp32 accumulation, fp32 tensors in HBM, split-fp16 (fp32-grade) direct weight gradients',
                     batchnorm='train mode', note='not the headline: no 1e-4 parity claim for this mode')

    # BASELINE configs[4] again, as a STORAGE mode (`--storage fp16` and, round 6, `--storage bf16` -- the type the config names):
    # activations and activation gradients live in HBM in 16 bits (half the activation traffic; a 16-bit activation has no low
    # part, so the split-operand products drop to two per fp32 product forward / one in the weight gradients), fp32 accumulation
    # / statistics / weights / parameter gradients, static loss scale.  A second model with the same initial weights per mode;
    # reported beside the headline, never as `value` (tests/test_gpu_h16.py states the tolerances).  `with_fp16_operands`: the
    # same plus `--precision fp16`.
    storage16 = storage_bf16 = None

    def storage_leg(kind, with_operands=True):
        """One 16-bit storage mode (`--storage fp16|bf16`): a second model with the same initial weights, timed next to the fp32-storage
        step.  Never raises: an extra leg must not cost the headline line."""
        m16 = o16 = None
        try:
            a16 = full_flags()
            a16.storage = kind
            m16 = build(a16, device)
            if dist_on:
                parallel.attach(m16, sync_bn=cli.sync_bn)
            o16 = FusedAdam(m16.parameters(), lr=a16.lr, weight_decay=a16.wd)
            m16.train()

            def run16(n):
                for _ in range(3):
                    train_iteration(m16, o16, batch, a16, 0)
                sync()
                t1 = time.perf_counter()
                for _ in range(n):
                    loss16 = train_iteration(m16, o16, batch, a16, 0)
                sync()
                dt16 = time.perf_counter() - t1
                if dist_on:
                    t = torch.tensor([dt16], device=device, dtype=torch.float64)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt16 = float(t)
                return dt16, float(loss16)
            n16 = max(5, cli.steps // 2)
            # the fp32-storage step timed again, directly in front of the 16-bit legs: the chip is power-managed and runs the
            # later legs of a bench call a few per cent slower than the first, so the ratio is taken between neighbours
            model.train()
            for _ in range(2):
                train_iteration(model, opt, batch, a, 0)
            sync()
            t1 = time.perf_counter()
            for _ in range(n16):
                train_iteration(model, opt, batch, a, 0)
            sync()
            dt32 = time.perf_counter() - t1
            if dist_on:
                t = torch.tensor([dt32], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt32 = float(t)
            dt16, l16 = run16(n16)
            dt16x = None
            if with_operands:
                lib.pp_set_matrix_products(1)
                try:
                    dt16x, _ = run16(n16)
                finally:
                    lib.pp_set_matrix_products(3)
            m16.eval()                              # the reference's steady state: BatchNorm with running statistics
            dt16e, _ = run16(n16)
            m16.train()
            what = ('IEEE fp16 (11 significand bits; an fp16 activation IS the high part the split-fp16 matrix kernels consume)' if kind == 'fp16' else
                    'bfloat16 -- the type BASELINE.json configs[4] names (8 significand bits, fp32 exponent range; converts exactly into the fp16 '
                    'hi operand of the matrix kernels)')
            return dict(images_per_sec=round(B * world * n16 / dt16, 2), ms_per_step=round(dt16 / n16 * 1e3, 3), steps=n16,
                        speedup_over_fp32_storage=round(dt32 / dt16, 3), fp32_storage_ms_per_step_adjacent=round(dt32 / n16 * 1e3, 3),
                        with_fp16_operands=(dict(images_per_sec=round(B * world * n16 / dt16x, 2), ms_per_step=round(dt16x / n16 * 1e3, 3),
                                                 speedup_over_fp32_storage=round(dt32 / dt16x, 3)) if dt16x else None),
                        bn_eval_images_per_sec=round(B * world * n16 / dt16e, 2),
                        final_loss=round(l16, 6), loss_scale=m16.engine.loss_scale,
                        dtype=f'{kind} storage: activations / activation gradients in HBM as {what}; fp32 accumulation, BatchNorm '
                              'statistics, weights, logits, parameter gradients, optimizer', batchnorm='train mode',
                        note='not the headline: stated tolerance in tests/test_gpu_h16.py, Dice rows in DESIGN.md')
        except Exception as e:                 # noqa: BLE001 -- reported, not raised
            return dict(error=f'{type(e).__name__}: {e}')
        finally:
            del m16, o16
            torch.cuda.empty_cache()

    if world == 1 and not cli.no_bn_eval and cli.session == 'Experiment':      # single process only: a second attached model is not part of the scaling runs
        storage16 = storage_leg('fp16')
        storage_bf16 = storage_leg('bf16', with_operands=False)

    # The same step replayed from a hipGraph (pacingpseudo_amd/graph.py: both streams, their fork / join events and the fused
    # optimizer captured once, ONE host call per step afterwards), timed next to the eager step: what the step costs when the
    # Python host is out of it.  Beside the headline, never `value` (the contract's per-launch HIP events cannot be recorded
    # inside a replayed graph).  Single process only.
    graphed = None
    if world == 1 and not dist_on and not cli.no_bn_eval:
        try:
            from pacingpseudo_amd.graph import GraphedStep
            model.train()
            n_g = max(5, cli.steps // 2)
            for _ in range(2):
                train_iteration(model, opt, batch, a, 0)
            sync()
            t1 = time.perf_counter()
            for _ in range(n_g):
                train_iteration(model, opt, batch, a, 0)
            sync()
            dt_e = time.perf_counter() - t1
            gs = GraphedStep(model, opt, lambda o, ep: assemble_loss(o, a, ep), warmup=1)
            for _ in range(3):                       # 1 eager call, the capture (+ its first replay), one more replay
                gs(batch, 0)
            sync()
            host_g = 0.0
            t1 = time.perf_counter()
            for _ in range(n_g):
                h0 = time.perf_counter()
                lg, _ = gs(batch, 0)
                host_g += time.perf_counter() - h0
            sync()
            dt_g = time.perf_counter() - t1
            graphed = dict(images_per_sec=round(B * n_g / dt_g, 2), ms_per_step=round(dt_g / n_g * 1e3, 3), steps=n_g,
                           eager_ms_per_step_adjacent=round(dt_e / n_g * 1e3, 3), host_ms_per_replay=round(host_g / n_g * 1e3, 3),
                           captures=gs.captures, replays=gs.replays, final_loss=round(float(lg), 6),
                           what='the whole iteration (forward, losses, backward on two streams, fused Adam) as ONE hipGraph replay per step')
            del gs
        except Exception as e:                 # noqa: BLE001 -- reported, not raised
            graphed = dict(error=f'{type(e).__name__}: {e}'[:400])

    # BASELINE.json configs[0] on the GPU: --session=Control (UNet + partial CE, one backbone pass), batch 8 -- the case the
    # cpu_baseline leg times as `control_batch8_images_per_sec`.  Single process only (it is the reference's CPU-runnable case).
    control = None
    if world == 1 and cli.session == 'Experiment' and not cli.no_bn_eval:
        m_ctl = o_ctl = b_ctl = None
        try:                                   # an extra leg must never cost the headline line
            a_ctl = default_args()
            m_ctl = build(a_ctl, device)
            o_ctl = FusedAdam(m_ctl.parameters(), lr=a_ctl.lr, weight_decay=a_ctl.wd)
            b_ctl = {k: v.to(device) for k, v in synthetic_batch(cli.cpu_batch, S, S, a_ctl.num_classes, seed=0).items() if k != 'label'}
            m_ctl.train()
            for _ in range(3):
                train_iteration(m_ctl, o_ctl, b_ctl, a_ctl, 0)
            sync()
            n_ctl = max(5, cli.steps)
            t1 = time.perf_counter()
            for _ in range(n_ctl):
                train_iteration(m_ctl, o_ctl, b_ctl, a_ctl, 0)
            sync()
            dtc = time.perf_counter() - t1
            control = dict(images_per_sec=round(cli.cpu_batch * n_ctl / dtc, 2), ms_per_step=round(dtc / n_ctl * 1e3, 3), steps=n_ctl,
                           batch=cli.cpu_batch, workload=f'--session=Control (UNet + partial CE), synthetic {S}x{S}x1 5-class, batch '
                           f'{cli.cpu_batch}, BatchNorm train mode (BASELINE.json configs[0])')
        except Exception as e:                 # noqa: BLE001 -- reported, not raised
            control = dict(error=f'{type(e).__name__}: {e}')
        finally:
            del m_ctl, o_ctl, b_ctl
            torch.cuda.empty_cache()

    # the GPU input pipeline, timed on its own: NOT part of `value` (inputs are resident in HBM there)
    aug_rate = None
    if rank == 0 and not cli.no_bn_eval:
        try:                                   # an extra leg must never cost the headline line
            from pacingpseudo_amd.augment import AugConfig, DeviceAugmenter
            aug = DeviceAugmenter(AugConfig(num_classes=a.num_classes, crop_size=(S, S)), device, seed=1)
            g = torch.Generator().manual_seed(0)
            raw_img = torch.randn(B, S, S, generator=g).to(device)
            raw_lab = torch.randint(0, a.num_classes, (B, S, S), generator=g, dtype=torch.int32).to(device)
            for _ in range(3):
                aug(raw_img, raw_lab, raw_lab)
            torch.cuda.synchronize()                       # rank 0 only: no collective in this block
            t1 = time.perf_counter()
            for _ in range(20):
                aug(raw_img, raw_lab, raw_lab)
            torch.cuda.synchronize()
            aug_rate = B * 20 / (time.perf_counter() - t1)
        except Exception as e:                 # noqa: BLE001 -- reported, not raised
            print(f'[bench] input-pipeline leg failed: {type(e).__name__}: {e}', file=sys.stderr)

    if rank == 0:
        fams = FAMILIES
        table = family_table(prof, cli.steps)
        single = None
        if prof_single is not None:
            single = {}
            for kind, names, peak, what in fams:
                v = prof_single.get(kind)
                if v and v['launches'] and v['ms'] > 0:
                    ex = v['flops'] / (v['ms'] * 1e-3) / 1e12
                    single[kind] = {'ms_per_step': round(v['ms'] / 2, 3), 'executed_tflops': round(ex, 2), 'frac': round(ex / peak, 4),
                                    'algorithmic_frac': round(v['alg_flops'] / (v['ms'] * 1e-3) / 1e12 / peak, 4)}
            sm = sum(r['ms_per_step'] for r in single.values())
            single = {'note': 'two untimed steps with the weight gradients back on the main stream: every family alone on the chip',
                      'families': single, 'matrix_ms_per_step': round(sm, 3),
                      'matrix_pipe_utilisation_time_weighted': round(sum(r['frac'] * r['ms_per_step'] for r in single.values()) / sm, 4) if sm else None,
                      'all_families_ms_per_step': {k: round(v['ms'] / 2, 3) for k, v in prof_single.items() if v['launches']}}
        # `roofline` describes the family that was event-timed INSIDE the timed region (the dominant one of the profiled warm-up step)
        dom = next((r for r in table if r['family'] == dom_kind), table[0]) if cli.prof_timed == 'dominant' else table[0]
        mfma_ms = sum(r['ms_per_step'] for r in table)
        # time-weighted utilisation of the matrix pipes over all of these launches
        util = sum(r['frac'] * r['ms_per_step'] for r in table) / mfma_ms if mfma_ms > 0 else 0.0
        # all families, from the two untimed steps that follow the timed region (every launch bracketed by events)
        kernels = {k: dict(launches_per_step=v['launches'] / 2, ms_per_step=round(v['ms'] / 2, 3),
                           tflops=round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) if v['ms'] > 0 and v['flops'] else None,
                           alg_gbps=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] > 0 else None)
                   for k, v in prof_all.items() if v['launches']}
        line = {
            'metric': 'training images/sec (256x256, 5-class)', 'value': round(value, 2), 'unit': 'images/sec',
            'n_gpus': world, 'steps': cli.steps, 'warmup': cli.warmup, 'ms_per_step': round(ms_per_step, 3),
            'step_ms': step_stats, 'host': host_stats,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (matrix products as 3 fp16 MFMA products of split operands with fp32 accumulation, or fp32 MFMA)',
            'data': 'synthetic',
            'config': {'workload': f'PacingPseudo {"full flags (ent + decoder-consistency + aux-path + memory)" if a.do_aux_path else "Control (pCE only)"}, '
                                   f'synthetic {S}x{S}x1 5-class, batch {B}/GPU, BatchNorm {"eval mode (epoch >= 1, running statistics)" if eval_main else "train mode"}',
                       'global_batch': B * world, 'image': [S, S], 'parallelism': f'dp{world}'},
            'roofline': {
                'kernel': f"{dom['kernels'][0]} ({dom['what']}): the matrix-core family with the most time per step",
                'bound': 'mfma', 'achieved': dom['executed_tflops'], 'peak': dom['peak_tflops'], 'unit': 'TFLOP/s',
                'frac': dom['frac'], 'algorithmic_frac': dom['algorithmic_frac'], 'traffic': dom['traffic'],
                'traffic_source': traffic_per_launch(dom['kernels'])[1],
                'flops_counted': 'EXECUTED MFMA flops of that kernel (split-fp16 kernels issue 3 fp16 products per fp32 '
                                 'product, Winograd GEMMs 4.5 [F(4x4)] or 8 [F(2x2)] flop per pixel*cin*cout where the direct '
                                 'form needs 18); algorithmic_tflops prices the same time with the direct-form fp32 count of '
                                 'SURVEY.md 8(d)',
                'algorithmic_tflops': dom['algorithmic_tflops'],
                'algorithmic_bytes_per_launch': dom['algorithmic_bytes_per_launch'],
                'launches_per_step': dom['launches_per_step'], 'avg_launch_ms': dom['avg_launch_ms'],
                'matrix_families': table,
                'matrix_ms_per_step': round(mfma_ms, 3),
                'matrix_pipe_utilisation_time_weighted': round(util, 4),
                'timed_by_events_in_timed_region': ([dom_kind] if cli.prof_timed == 'dominant' and dom_kind else
                                                    (list(matrix_kinds) if cli.prof_timed == 'matrix' else [])),
                'other_families_from': 'two untimed steps after the timed region (HIP events around every launch)',
                'streams': ('2: weight gradients run beside the data-gradient / BatchNorm chain, event times include the co-running '
                            'kernels' if prof_single is not None else '1'),
                'single_stream': single,
            },
            'whole_step': {'algorithmic_tflops': round(FLOP_PER_IMAGE_FULL * value / world / 1e12, 2) if a.do_aux_path else None,
                           'algorithmic_over_f32_peak': round(FLOP_PER_IMAGE_FULL * value / world / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if a.do_aux_path else None,
                           'frac_of_hbm_roofline': round(BYTES_PER_IMAGE_FULL * value / world / 8.0e12, 4) if a.do_aux_path else None},
            'kernels': kernels,
            'rccl_world_size': (dist.get_world_size() if dist_on else 1),
            'collective_backend': (dist.get_backend() if dist_on else None),
            'batchnorm': {'mode': ('eval (running statistics: the reference from epoch 1 on)' if eval_main else
                                   'train (batch statistics; the reference runs this mode in epoch 0 and eval mode from epoch 1 on)'),
                          'sync_bn': bool(world > 1 and cli.sync_bn),
                          'statistics': ('one process: the whole batch' if world == 1 else
                                         ('global batch (packed all-reduce per BatchNorm call)' if cli.sync_bn else
                                          f'per rank ({B} images): pass --sync-bn for the reference\'s whole-batch statistics'))},
            'bn_eval': bn_eval,
            'bn_eval_images_per_sec': bn_eval['images_per_sec'] if bn_eval else None,
            'control_images_per_sec': control['images_per_sec'] if control else None,
            'control': control,
            'graph_replay': graphed,
            'mixed_precision': mixed,
            'storage_fp16': storage16,
            'storage_bf16': storage_bf16,
            'input_pipeline_images_per_sec': round(aug_rate, 1) if aug_rate else None,
            'final_loss': round(final_loss, 6),
        }
        if world == 1 and not cli.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline(a, cli.cpu_batch, S, cli.cpu_steps)
                line['gpu_over_cpu'] = round(value / line['cpu_baseline']['value'], 1)
            except Exception as e:             # noqa: BLE001 -- the GPU line must still be printed
                line['cpu_baseline'] = dict(error=f'{type(e).__name__}: {e}'[:400], kind='port')
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
