#!/usr/bin/env python
"""Headline benchmark: ConvLSTM-UNet training frames/s (seq_len*batch per optimiser step) on
synthetic 256x256 clips, BASELINE.json config-2 per GPU (T=8, B=4 slots/GPU, Params.py widths:
5x5 ConvLSTM @128/256/256/512, 3x3 encoder/decoder, fp32), weak-scaled data-parallel over N GPUs.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under
                                                          torch.distributed.run, one rank per GPU, backend nccl = RCCL)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The line carries `dp` = what the collective layer actually saw (backend, dist.get_world_size(), the device of every rank,
gradient all-reduce launches per step and their stand-alone duration).  A mismatch between --gpus, WORLD_SIZE, the process
group and the visible devices is an ERROR (exit code 2), never a one-GPU number under an N-GPU label.

A step = forward(training) + weighted CE + backward (BPTT inside the window) + Adam + recurrent
state mask [+ bucketed RCCL gradient all-reduce].  Inputs are resident in HBM before the timed
region.  Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (the MFMA kernel class
with the largest share of the step -- fused ConvLSTM step / dgrad convs / weight gradients are each about a
third -- fp32 MFMA bound, algorithmic FLOPs over HIP-event time on the launch stream) and
`cpu_baseline` (the torch-CPU oracle of the same network on a bounded sample, N=1 only).
"""
import argparse
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from lu_native.profile import PEAK_FP32_MFMA_TFLOPS, PEAK_HBM_GBS, PEAK_BF16_MFMA_TFLOPS  # noqa: E402,F401


def conv_flops(k, cin, cout, h, w):
    return 2.0 * k * k * cin * cout * h * w


def step_flops(net, H, W, B, T, cin=1):
    """Algorithmic FLOPs of one training step: 3x forward, minus the data-image dgrad,
    minus the t=0 recurrent dgrad of every ConvLSTM (no gradient into the carried state)."""
    from lu_native.plan import make_plan
    plan = make_plan(net, cin)
    fwd = skip_first = rec = 0.0
    h, w, first = H, W, True
    for blk in plan['down']:
        for l in blk['lstm']:
            fx = conv_flops(l['k'], l['cin'], 4 * l['f'], h, w)
            fh = conv_flops(l['k'], l['f'], 4 * l['f'], h, w)
            fwd += fx + fh
            rec += fh
            if first:
                skip_first = fx
                first = False
        for l in blk['conv']:
            if l['stride'] == 2:
                h, w = -(-h // 2), -(-w // 2)
            fwd += conv_flops(l['k'], l['cin'], l['cout'], h, w)
    for blk in plan['up']:
        if blk['up_factor'] == 2:
            h, w = 2 * h, 2 * w
        for l in blk['conv']:
            fwd += conv_flops(l['k'], l['cin'], l['cout'], h, w)
    per_frame = 3 * fwd - skip_first - rec / T
    return per_frame * B * T, fwd


def lstm_step_flops(net, H, W, B, cin=1):
    """[(level, flops of ONE fused ConvLSTM step launch)]"""
    from lu_native.plan import make_plan
    plan = make_plan(net, cin)
    out, h, w = [], H, W
    for blk in plan['down']:
        for l in blk['lstm']:
            out.append(B * (conv_flops(l['k'], l['cin'], 4 * l['f'], h, w) + conv_flops(l['k'], l['f'], 4 * l['f'], h, w)))
        if blk['stride'] == 2:
            h, w = -(-h // 2), -(-w // 2)
    return out


NETS = ('params', 'default5', 'lstm3')


def net_by_name(name):
    """params   = Params.py:49-69, what train2D.py trains by default (3x3 convs, 5x5 ConvLSTM): the headline;
    default5 = Networks.DEFAULT_NET_DOWN_PARAMS (Networks.py:12-32): 5x5 everywhere, 2 543.8 GFLOP/frame of training at 256^2;
    lstm3    = north_star's wording, "one 3x3 conv producing i/f/o/g": 3x3 everywhere, 915.8 GFLOP/frame."""
    import Params
    import Networks
    return {'params': Params.CTCParams.net_kernel_params, 'default5': Networks.DEFAULT_NET_DOWN_PARAMS,
            'lstm3': Params.lstm_unet_kernels(3, 3)}[name]


NET_WORDS = {'params': 'Params.py widths (5x5 ConvLSTM 128/256/256/512, 3x3 convs)',
             'default5': 'Networks.DEFAULT_NET_DOWN_PARAMS (5x5 ConvLSTM AND 5x5 convs, same widths)',
             'lstm3': '3x3-ConvLSTM variant of north_star (3x3 ConvLSTM 128/256/256/512, 3x3 convs)'}


def annotate_x3(rows):
    """precision 'bf16x3': the bf16 classes execute SIX bf16 MFMA products per fp32 product.  `achieved` / `frac` stay what the matrix
    pipe does (executed FLOPs against the bf16 peak: the roofline the kernel lives under); the fp32-equivalent rate rides along."""
    for r_ in rows:
        if 'bf16' in r_['kernel']:
            r_['executed_over_algorithmic'] = 6
            r_['algorithmic_fp32_tflops'] = round(r_['achieved'] / 6.0, 2)


def x3_summary(rows, sec, flops, frames):
    """precision 'bf16x3', whole step, priced against the pipe that executes it: the bf16 MFMA FLOPs actually issued (6 x the
    algorithmic FLOPs of the layers on split operands = sum over the bf16 kernel classes of rate x time) over the step time, against the
    bf16 peak.  No fraction of the fp32 peak is reported: the mode does not run on that pipe (round-5 verdict, weak #6)."""
    ex = sum(r_['achieved'] * r_['ms_per_step'] for r_ in rows if 'bf16' in r_['kernel']) / (1e3 * sec)      # TFLOP/s
    return {'executed_bf16_tflops': round(ex, 1), 'frac_of_bf16_peak': round(ex / PEAK_BF16_MFMA_TFLOPS, 4), 'bf16_peak': PEAK_BF16_MFMA_TFLOPS,
            'algorithmic_fp32_tflops': round(flops / 1e12 / sec, 2), 'fp32_equivalent_frames_per_s': round(frames / sec, 3)}


def measure_variant(net, precision, batches, B, T, H, W, dp, steps, warmup):
    """One more trainer on the same resident batches: K timed steps + one step under per-class HIP events."""
    import Params
    import train2D
    from lu_native import ops
    from lu_native.profile import summarize_events
    tr = train2D.Trainer(Params.CTCParams.net_model, net, 'NCHW', Params.CTCParams.class_weights,
                         Params.CTCParams.learning_rate, dp=dp, seed=0, precision=precision)

    def step(i):
        img, seg, keep = batches[i % len(batches)]
        tr.train_step(img, seg, want_outputs=True)
        tr.model.reset_states_per_batch(keep)

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / steps
    ops.EVENT_LOG = []
    step(warmup + steps)
    torch.cuda.synchronize()
    ev, ops.EVENT_LOG = ops.EVENT_LOG, None
    rows, _ = summarize_events(ev)
    if precision == 'bf16x3':
        annotate_x3(rows)
    flops, fwd = step_flops(net, H, W, B, T)
    peak = PEAK_BF16_MFMA_TFLOPS if precision == 'bf16' else PEAK_FP32_MFMA_TFLOPS
    del tr
    torch.cuda.empty_cache()
    out = {'ms_per_step': round(1e3 * sec, 3), 'frames_per_s': round(B * T / sec, 3),
           'train_gflop_per_frame': round(flops / (B * T) / 1e9, 1), 'step_tflops_achieved': round(flops / 1e12 / sec, 2),
           'frac_of_peak': round(flops / 1e12 / sec / peak, 4), 'peak': peak, 'steps': steps,
           'mfma_kernels': [{k_: r_[k_] for k_ in ('kernel', 'achieved', 'frac', 'launches_per_step', 'ms_per_step', 'algorithmic_fp32_tflops')
                             if k_ in r_} for r_ in rows]}
    if precision == 'bf16x3':      # priced against the pipe it runs on: executed bf16 FLOPs over the bf16 peak, never a fraction of the fp32 peak
        for k_ in ('frac_of_peak', 'peak', 'step_tflops_achieved'):
            out.pop(k_)
        out.update(x3_summary(rows, sec, flops, B * T))
        out['dtype'] = 'f32 results via 3 x bf16 split'
    return out


def synthetic_batches(n, B, T, H, W, rank, device):
    from DataHandeling import SyntheticSequence2D
    prov = SyntheticSequence2D(image_crop_size=(H, W), unroll_len=T, batch_size=B, data_format='NCHW',
                               seed=1234, rank=rank)
    out = []
    for i in range(n):
        img, seg, _, keep = prov.get_batch()
        keep[:] = 1.0
        if i % 4 == 3:
            keep[i % B] = 0.0     # one slot's clip ends every 4th step
        out.append((torch.from_numpy(img).to(device), torch.from_numpy(seg).to(device), torch.from_numpy(keep).to(device)))
    return out


def cpu_baseline(net, budget_s=60.0):
    """the torch-CPU restatement of the TF2 path (oracle/torch_oracle.py, fp32; TensorFlow is not installed) on
    the GPU box's host cores, two bounded samples: BASELINE config-1 timed fully (128x128, T=4, B=1, 32-channel 3x3 net) and
    a T=2 slice of config-2 at the full 256x256 frame size (B=1, Params.py widths).  `value` = the config-2 slice."""
    from oracle import np_oracle as npo
    from oracle import torch_oracle as tho
    host_cores = os.cpu_count() or 1
    t_start = time.time()
    # PROBE, don't assert (SURVEY §8c last row): is there a TensorFlow on this box?  If so, pin the oracle
    # (tools/tf_pin.py -> gpurun_out/tf_pin/) and time a build-owned tf.keras step beside the port; if not, say what the import
    # actually raised.
    import importlib.util
    spec = importlib.util.spec_from_file_location('lu_tf_pin', os.path.join(ROOT, 'tools', 'tf_pin.py'))
    tf_tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tf_tool)
    tf_mod, tf_what = tf_tool.probe_tensorflow()
    tf_line = None
    if tf_mod is not None:
        try:
            rc = tf_tool.main(out=os.path.join(ROOT, 'gpurun_out', 'tf_pin'))
            step_tf = tf_tool.keras_train_step_timer(tf_mod, net, 256, 256, 1, 2)
            step_tf()                                             # trace + warm-up
            t_tf = min(step_tf(), step_tf())
            tf_line = {'value': round(2.0 / t_tf, 4), 'unit': 'frames/s', 'kind': 'tf', 'tensorflow': tf_what, 'tf_pin_rc': rc,
                       'cores': host_cores, 'sample': 'tf.keras (build-owned model, NHWC) config-2 slice: 256x256, B=1, T=2, '
                                                      'best of 2 steps after the trace: %.2f s/step' % t_tf}
        except Exception as exc:
            tf_line = {'value': None, 'kind': 'tf', 'tensorflow': tf_what, 'sample': 'failed: %r' % (exc,)}

    def make(net_, H, W, B, T):
        p = npo.init_params(net_, 1, seed=0)
        m = tho.TorchULSTM(net_, 1, p, dtype=torch.float32)
        rng = np.random.default_rng(0)
        x = rng.standard_normal((B, T, H, W, 1)).astype(np.float32)
        gt = rng.integers(-1, 3, size=(B, T, H, W)).astype(np.float32)

        def timed_step():
            t0 = time.time()
            m.train_step(x, gt, [0.15, 0.25, 0.6])
            m.reset_states_per_batch(np.ones(B))
            return time.time() - t0
        return timed_step

    # config-1: also the probe for the thread count.  torch's CPU convolutions collapse when oversubscribed (256 threads:
    # > 100 s/step on the GPU box), so walk the count up and keep the fastest; `cores` reports the threads actually used.
    c1 = {'down_conv_kernels': [[(3, 32), (3, 32)]] * 4, 'lstm_kernels': [[(3, 32)]] * 4,
          'up_conv_kernels': [[(3, 32), (3, 32)]] * 3 + [[(3, 32), (3, 32), (1, 3)]]}
    step1 = make(c1, 128, 128, 1, 4)
    best = (float('inf'), 1)
    for nt in [c for c in (4, 8, 16, 32, 64, 128) if c <= host_cores] or [host_cores]:
        torch.set_num_threads(nt)
        step1()                           # warm-up at this thread count
        dt = min(step1(), step1())
        if dt < best[0]:
            best = (dt, nt)
        elif dt > 1.3 * best[0]:
            break
    threads = best[1]
    torch.set_num_threads(threads)
    n1 = 5
    t1 = sum(step1() for _ in range(n1)) / n1
    # config-2 slice: full-size frames, full-width net, T = 2, B = 1 (4.5 TFLOP per step)
    step2 = make(net, 256, 256, 1, 2)
    step2()                               # warm-up
    n2, e2 = 0, 0.0
    while n2 < 2 and (n2 == 0 or time.time() - t_start + e2 / n2 < budget_s):
        e2 += step2()
        n2 += 1
    t2 = e2 / n2
    return {'value': round(2.0 / t2, 4), 'unit': 'frames/s', 'cores': threads, 'host_cores': host_cores, 'kind': 'port',
            'tensorflow_probe': tf_what if tf_mod is not None else '`import tensorflow` on this box raised %s' % tf_what,
            'tf': tf_line,
            'sample': 'config-2 slice: Params.py-width net, 256x256 full frames, B=1, T=2, %d training step(s) after 1 warm-up: '
                      '%.2f s/step; torch CPU fp32 restatement of the TF2 path, %d threads (best of a '
                      'walk-up probe) on a %d-core host' % (n2, t2, threads, host_cores),
            'config1': {'value': round(4.0 / t1, 3), 'unit': 'frames/s', 's_per_step': round(t1, 4),
                        'sample': 'BASELINE config-1 timed fully: 128x128, T=4, B=1, 32-channel 3x3 ConvLSTM-UNet, '
                                  '%d training steps after warm-up, %d threads' % (n1, threads)}}


def _die(msg):
    print('bench.py: ' + msg, file=sys.stderr, flush=True)
    sys.exit(2)


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start N ranks, one per GPU, under
    torch.distributed.run (rendezvous on 127.0.0.1, a free port) and hand back its exit code.  LU_DP_BACKEND=gloo lets the
    ranks share devices (control-flow checks on a 1-GPU box); the default backend nccl (= RCCL) needs a device per rank."""
    import socket
    import subprocess
    backend = os.environ.get('LU_DP_BACKEND') or 'nccl'
    n_dev = torch.cuda.device_count()
    if n_dev < 1 or (backend == 'nccl' and n_dev < n):
        _die('--gpus %d needs %d visible GPUs for the %s backend, this node shows %d' % (n, n, backend, n_dev))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('bench.py: launching %d ranks: %s' % (n, ' '.join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))))


def dry_run_plan(args):
    """`--gpus N --dry-run`: everything a multi-GPU launch depends on, checked and printed WITHOUT touching RCCL or building a model
    -- visible devices against N and the backend, the launcher environment (WORLD_SIZE / RANK / LOCAL_RANK / MASTER_* when one
    is set), the dmabuf-IPC switch RCCL needs on this driver, a free rendezvous port, the exact command the self-launch would
    run, the per-rank plan (device, clip slots, frames per step) and the gradient buckets the engine will hand to the collective
    layer (ranges of the flat gradient buffer in backward-completion order, merged by DataParallel.bucket_ready exactly as in a
    step).  ONE JSON line on stdout; exit code 0 when a real launch would start, 2 (and `problems`) when it would be refused."""
    import socket
    import Params
    from lu_native.dp import DataParallel
    from lu_native.plan import make_plan, param_specs
    n = args.gpus
    backend = os.environ.get('LU_DP_BACKEND') or 'nccl'
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    problems = []
    if n_dev < 1:
        problems.append('no GPU visible: the training path has no CPU fallback')
    elif backend == 'nccl' and n_dev < n:
        problems.append('--gpus %d on RCCL needs one device per rank, %d visible (LU_DP_BACKEND=gloo shares devices: control flow only)' % (n, n_dev))
    env = {k: os.environ.get(k) for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'LU_DP_BACKEND',
                                          'HSA_ENABLE_IPC_MODE_LEGACY', 'HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'NCCL_DEBUG')}
    if env['WORLD_SIZE'] is not None and int(env['WORLD_SIZE']) != n:
        problems.append('--gpus %d but the launcher environment says WORLD_SIZE=%s' % (n, env['WORLD_SIZE']))
    if env['WORLD_SIZE'] is not None and n > 1:
        for k in ('RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
            if env[k] is None:
                problems.append('launcher environment lacks %s' % k)
    if env['HSA_ENABLE_IPC_MODE_LEGACY'] not in (None, '0') and backend == 'nccl' and n > 1:
        problems.append('HSA_ENABLE_IPC_MODE_LEGACY=%s: this host driver only supports dmabuf IPC; RCCL fails with '
                        'hipIpcGetMemHandle: invalid argument unless it is 0 (the self-launch sets 0 when it is unset)' % env['HSA_ENABLE_IPC_MODE_LEGACY'])
    port = None
    try:
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', int(env['MASTER_PORT']) if (env['WORLD_SIZE'] is not None and env['MASTER_PORT']) else 0))
            port = sk.getsockname()[1]
    except OSError as exc:
        problems.append('rendezvous port on 127.0.0.1 not available: %s' % exc)
    passthrough = [a for a in sys.argv[1:] if a != '--dry-run']
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + passthrough
    H, W = (args.hw if args.hw else (args.size, args.size))
    B, T = args.batch, args.unroll
    ranks = [{'rank': r, 'local_rank': r, 'device': 'cuda:%d' % (r % max(n_dev, 1)), 'global_slots': list(range(r * B, (r + 1) * B)),
              'frames_per_step': B * T} for r in range(n)]
    # gradient buckets: the engine's segments (one per block, decoder first) through the collective layer's own merging rule
    plan = make_plan(net_by_name(args.net), 1)
    total, members = 0, {}
    for name, shape, _ in param_specs(plan):
        cnt = int(np.prod(shape))
        blk = '.'.join(name.split('.')[:2])
        lo, hi = members.get(blk, (total, total))
        members[blk] = (min(lo, total), total + (cnt + 3) // 4 * 4)
        total += (cnt + 3) // 4 * 4
    order = ['up.%d' % i for i in reversed(range(len(plan['up'])))] + ['down.%d' % i for i in reversed(range(len(plan['down'])))]
    launched = []
    sim = DataParallel.solo()
    sim.world_size, sim.collectives = max(n, 2), True      # (the merging rule only runs with collectives on; nothing here issues one)
    sim._launch = lambda s_, e_: launched.append((s_, e_))
    for blk in order:
        sim.bucket_ready(*members[blk])
    if sim._carry is not None:
        launched.append(sim._carry)
    line = {'dry_run': True, 'ok': not problems, 'problems': problems, 'n_gpus': n, 'backend': backend, 'visible_devices': n_dev,
            'device_names': [torch.cuda.get_device_name(i) for i in range(n_dev)], 'environment': env,
            'rendezvous': {'addr': '127.0.0.1', 'port_probed_free': port}, 'self_launch_command': ' '.join(cmd),
            'self_launch_sets': {'HSA_ENABLE_IPC_MODE_LEGACY': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')},
            'config': {'workload': '%dx%d, seq_len=%d, batch=%d per GPU, %s, %s' % (H, W, T, B, NET_WORDS[args.net], args.precision),
                       'global_batch': n * B, 'scaling': 'weak'},
            'ranks': ranks, 'parameters': total,
            'gradient_buckets': [{'start': s_, 'end': e_, 'mbytes': round(4 * (e_ - s_) / 1e6, 1)} for s_, e_ in launched],
            'collectives_per_step': {'gradient_all_reduce': len(launched), 'loss_sums_all_reduce': 1,
                                     'sync_bn_all_reduce': (2 * sum(len(b['conv']) for b in plan['down'] + plan['up']) - 2) if args.sync_bn else 0},
            'next': 'python bench.py --gpus %d --check   (self-check, timing, overlap trace)' % n}
    print(json.dumps(line), flush=True)
    sys.exit(0 if not problems else 2)


def dp_self_check(dp, dev):
    """`--check` (N > 1): before anything is timed, one data-parallel training step with SyncBN on N ranks x 1 slot must equal the
    single-process step on the N-slot batch -- loss to 1e-5, the all-reduced PRE-ADAM gradients to 2e-6 of the largest gradient
    (the comparison of tests/test_dp_equivalence.py::test_dpN_syncbn_over_rccl_equals_single_process, inline).  Every rank
    learns the verdict (one flag all-reduce); a failure ends the run with exit code 2 and no JSON line."""
    import Networks
    import train2D
    from lu_native.dp import DataParallel
    n = dp.world_size
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 3, 1, 24, 32)).astype(np.float32)
    gt = rng.integers(-1, 3, size=(n, 3, 1, 24, 32)).astype(np.float32)
    net = {'down_conv_kernels': [[(3, w), (3, w)] for w in (32, 16, 16, 32)], 'lstm_kernels': [[(3, w)] for w in (32, 16, 16, 32)],
           'up_conv_kernels': [[(3, 16), (3, 16)], [(3, 8), (3, 8)], [(3, 8), (3, 8)], [(3, 8), (3, 8), (1, 3)]]}
    tr = train2D.Trainer(Networks.ULSTMnet2D, net, 'NCHW', [0.15, 0.25, 0.6], 1e-3, dp=dp, sync_bn=True, seed=3)
    sl = slice(dp.rank, dp.rank + 1)
    _, _, loss = tr.train_step(x[sl], gt[sl])
    grads = tr.engine.flat_grads.clone()
    res = {'ranks': n, 'slots': n, 'loss_err': None, 'grad_err_over_max': None, 'ok': False}
    flag = torch.zeros(1, device=dev)
    if dp.rank == 0:
        ref = train2D.Trainer(Networks.ULSTMnet2D, net, 'NCHW', [0.15, 0.25, 0.6], 1e-3, dp=DataParallel.solo(), seed=3)
        _, _, loss_ref = ref.train_step(x, gt)
        g_ref = ref.engine.flat_grads
        res['loss_err'] = abs(float(loss) - float(loss_ref))
        res['grad_err_over_max'] = float((grads - g_ref).abs().max() / g_ref.abs().max())
        res['ok'] = res['loss_err'] <= 1e-5 and res['grad_err_over_max'] <= 2e-6
        flag += 0.0 if res['ok'] else 1.0
        print('bench.py --check: dp%d + SyncBN vs single process: loss err %.3e, pre-Adam gradient err / max|g| %.3e -> %s' %
              (n, res['loss_err'], res['grad_err_over_max'], 'OK' if res['ok'] else 'FAILED'), file=sys.stderr, flush=True)
    torch.distributed.all_reduce(flag)
    if float(flag.item()) > 0:
        _die('--check failed: the data-parallel step does not reproduce the single-process step; nothing was timed')
    del tr
    torch.cuda.empty_cache()
    return res


def dp_report(dp, dev_index, engine, steps, launched, sync_bn):
    """What the collective layer saw, measured on the live process group (every rank calls this): backend, group size, the
    device of each rank, gradient all-reduce launches per step, and -- timed stand-alone after the timed region with HIP
    events, on scratch buffers of the real bucket sizes -- how long the gradient buckets and one SyncBN-sized all-reduce
    take when nothing overlaps them."""
    import torch.distributed as dist
    if not dp.collectives:
        return {'backend': None, 'world_size': 1, 'devices': [{'rank': 0, 'device': dev_index,
                                                               'name': torch.cuda.get_device_name(dev_index)}]}
    mine = {'rank': dp.rank, 'local_rank': dp.local_rank, 'device': dev_index, 'name': torch.cuda.get_device_name(dev_index),
            'pid': os.getpid()}
    everyone = [None] * dp.world_size
    dist.all_gather_object(everyone, mine)
    dev = torch.device('cuda', dev_index)
    sizes = [e - s_ for (s_, e) in dp.last_ranges] or [engine.n_flat]
    scratch = torch.zeros(max(sizes), device=dev, dtype=torch.float32)
    small = torch.zeros(1024, device=dev, dtype=torch.float32)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / reps], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    grad_ms = timed(lambda: [dist.all_reduce(scratch[:n_]) for n_ in sizes], 3)
    small_ms = timed(lambda: dist.all_reduce(small), 20)
    n_bn = sum(1 for name in engine.P if name.endswith('.gamma'))
    return {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'devices': everyone,
            'distinct_devices': len({d['device'] for d in everyone}),
            'allreduce_launches_per_step': round(launched / max(steps, 1), 2),
            'gradient_bucket_bytes': [4 * n_ for n_ in sizes],
            'allreduce_ms_per_step': round(grad_ms, 3),
            'allreduce_what': 'the gradient buckets of one step all-reduced back to back on scratch buffers, nothing else '
                              'running (max over ranks); inside the step they overlap the encoder backward',
            'small_allreduce_ms': round(small_ms, 4),
            'syncbn_allreduces_per_step': 2 * n_bn if sync_bn else 0,
            'syncbn_ms_per_step_est': round(2 * n_bn * small_ms, 3) if sync_bn else 0.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--hw', type=int, nargs=2, default=None, help='non-square frames, e.g. 832 992 (config-4)')
    ap.add_argument('--batch', type=int, default=4, help='clip slots per GPU')
    ap.add_argument('--unroll', type=int, default=8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--by-shape', default=None, metavar='FILE',
                    help='also write the HIP-event timings grouped by (kernel class, work per launch) -- one row per layer shape')
    ap.add_argument('--no-infer', action='store_true', help='skip the secondary streaming-inference measurement')
    ap.add_argument('--no-bf16', action='store_true', help='skip the secondary bf16-mode measurement of the same step')
    ap.add_argument('--sync-bn', action='store_true')
    ap.add_argument('--no-wgrad-overlap', action='store_true', help='A/B: weight gradients in line instead of on the side stream')
    ap.add_argument('--wgrad-flags', type=int, default=0, help='A/B: LU_WGRAD_F_* bits OR-ed into every weight-gradient descriptor')
    ap.add_argument('--conv-flags', type=int, default=0, help='A/B: LU_CONV_F_* bits OR-ed into every convolution descriptor')
    ap.add_argument('--precision', choices=['fp32', 'bf16', 'bf16x3'], default='fp32',
                    help="bf16: BASELINE config-5 mixed precision (bf16 MFMA operands, fp32 everything else); bf16x3: fp32 arithmetic "
                         "on the bf16 MFMA (ConvLSTM convolutions on the exact three-way bf16 split of their fp32 operands)")
    ap.add_argument('--no-x3', action='store_true', help='skip the secondary bf16x3-mode measurement of the same step')
    ap.add_argument('--wgrad-overlap', action='store_true', help='A/B: weight gradients on the side stream (the default in bf16 mode only)')
    ap.add_argument('--net', choices=list(NETS), default='params',
                    help='kernel-size variant (SURVEY D1): params = train2D.py default = the headline; default5 = 5x5 everywhere; '
                         'lstm3 = 3x3 ConvLSTM (north_star wording)')
    ap.add_argument('--check', action='store_true',
                    help='N > 1: run the DP + SyncBN == single-process comparison inline first; refuse to print a line if it fails')
    ap.add_argument('--force-collectives', action='store_true',
                    help='N = 1: initialise the process group anyway (RCCL: backend nccl, one rank) and send the gradient buckets, the loss '
                         'sums and -- with --sync-bn -- the BatchNorm statistics through real all-reduce calls (LU_DP_FORCE=1, lu_native/dp.py)')
    ap.add_argument('--no-variants', action='store_true',
                    help='skip the `variants` block (lstm3 / default5 in fp32 and bf16, N = 1, headline net only)')
    ap.add_argument('--lib', default=None, metavar='SO',
                    help='A/B: another build of the kernel library (same ABI), e.g. the previous commit\'s for a same-box comparison')
    ap.add_argument('--dry-run', action='store_true',
                    help='validate devices / environment / rendezvous port for --gpus N and print the rank plan and gradient buckets as '
                         'one JSON line, without touching RCCL or building a model (exit code 2 if a real launch would be refused)')
    args = ap.parse_args()
    if args.gpus < 1:
        _die('--gpus must be >= 1')
    if args.dry_run:
        dry_run_plan(args)                   # does not return
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args.gpus)               # does not return
    if int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
        _die('--gpus %d but the launcher started WORLD_SIZE=%s ranks' % (args.gpus, os.environ.get('WORLD_SIZE')))
    if not torch.cuda.is_available():
        _die('no GPU visible: the training path has no CPU fallback')
    # stdout carries ONE JSON line and nothing else: native libraries write there too (gloo: "[Gloo] Rank 0 is connected to 1 peer
    # ranks" on some builds; RCCL with NCCL_DEBUG set) -- file descriptor 1 points at stderr until the line is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    backend_wanted = os.environ.get('LU_DP_BACKEND') or 'nccl'
    if args.gpus > 1 and backend_wanted == 'nccl' and torch.cuda.device_count() < args.gpus:
        _die('--gpus %d on RCCL needs one device per rank, %d visible' % (args.gpus, torch.cuda.device_count()))

    import Params
    from lu_native import build as lu_build
    from lu_native import ops
    from lu_native.dp import DataParallel
    import train2D

    if args.lib:
        ops.LIB_PATH = lu_build.LIB = os.path.abspath(args.lib)      # (build_id on the line is then this file's)
    if args.force_collectives:
        os.environ['LU_DP_FORCE'] = '1'
    dp = DataParallel()
    if dp.collectives and torch.distributed.get_world_size() != args.gpus:
        _die('process group has %d ranks, --gpus %d' % (torch.distributed.get_world_size(), args.gpus))
    dev_index = dp.local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    self_check = dp_self_check(dp, dev) if (args.check and dp.world_size > 1) else None
    net = net_by_name(args.net)
    H = W = args.size
    if args.hw:
        H, W = args.hw
    B, T = args.batch, args.unroll
    trainer = train2D.Trainer(Params.CTCParams.net_model, net, 'NCHW', Params.CTCParams.class_weights,
                              Params.CTCParams.learning_rate, dp=dp, sync_bn=args.sync_bn, seed=0,
                              precision=args.precision)
    batches = synthetic_batches(4, B, T, H, W, dp.rank, dev)
    if args.no_wgrad_overlap:
        trainer.engine.overlap_wgrad = False
    if args.wgrad_overlap:
        trainer.engine.overlap_wgrad = True
    ops.WGRAD_FLAGS |= args.wgrad_flags
    ops.CONV_FLAGS |= args.conv_flags

    def one_step(i):
        img, seg, keep = batches[i % len(batches)]
        trainer.train_step(img, seg, want_outputs=True)    # (softmax, predictions, loss) as train2D.py:87-103 returns
        trainer.model.reset_states_per_batch(keep)

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    dp.barrier()
    launched0 = dp.launched
    ms0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize()
    dp.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if dp.collectives:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    ms1 = torch.cuda.memory_stats(dev)
    # what the caching allocator did INSIDE the timed region: a retry = hipFree of every cached block + hipMalloc again, each a
    # device synchronisation (config-4 at 192 GB resident is where this matters: round-4 verdict, "find the 0.8 s")
    allocator = {k: int(ms1.get(k, 0) - ms0.get(k, 0)) for k in ('num_alloc_retries', 'num_ooms', 'num_device_alloc', 'num_device_free')}
    allocator['reserved_gb'] = round(ms1.get('reserved_bytes.all.current', 0) / 2 ** 30, 2)
    allocator['per'] = '%d timed steps' % args.steps
    dp_info = dp_report(dp, dev_index, trainer.engine, args.steps, dp.launched - launched0, args.sync_bn)
    if dp.collectives:
        dp_info['forced_world_of_one'] = dp.world_size == 1
        # proof of overlap: two more steps with per-bucket time stamps (outside the timed region: the stamps are device events,
        # but a traced step is not the step that is reported)
        dp.trace = []
        one_step(args.warmup + args.steps)
        one_step(args.warmup + args.steps + 1)
        torch.cuda.synchronize()
        traced = dp.trace_report()
        dp.trace = None
        dp_info['bucket_trace'] = traced[-1] if traced else None
        dp_info['bucket_trace_what'] = ('per gradient bucket of one step, rank 0: handed to the collective layer this long BEFORE '
                                        'backward ended (the remaining backward hides it), compute stream free again this long '
                                        'AFTER backward ended (the exposed part; compare allreduce_ms_per_step = stand-alone)')
        dp_info['exposed_allreduce_ms'] = traced[-1][-1]['done_ms_after_backward_end'] if traced and traced[-1] else None
        dp_info['self_check'] = self_check
    ms_per_step = 1e3 * elapsed / args.steps
    frames_per_s = dp.world_size * B * T * args.steps / elapsed

    # ---- roofline: the three MFMA kernel classes timed live with HIP events on the launch stream; the one with the
    # ---- largest share of the step is reported as `roofline`, all three under roofline.all_mfma_kernels
    roofline = None
    if dp.rank == 0:
        ops.EVENT_LOG = []
    one_step(args.warmup + args.steps)      # every rank takes part (the step contains collectives)
    torch.cuda.synchronize()
    if dp.rank == 0:
        ev, ops.EVENT_LOG = ops.EVENT_LOG, None
        traffic_db, traffic_src, traffic_build = {}, None, None
        build_id = lu_build.build_id()
        # which workload's counter tables: config-2 (no suffix), config-4 (_c4), the config-5 per-GPU shape (_c5shape); the net
        # variants carry their name (round 5: the round-4 tables covered config-2 / params only)
        wl_sfx = {(256, 256, 8, 4): '', (832, 992, 16, 2): '_c4', (512, 512, 8, 2): '_c5shape'}.get((H, W, T, B))
        if wl_sfx is not None and args.net != 'params':
            wl_sfx = '_' + args.net + wl_sfx
        try:     # L2-fabric bytes per launch from the committed rocprofv3 --pmc passes: a STATIC table
            if wl_sfx is not None:      # collected by tools/gpu/pmc_traffic.sh, not measured in this run
                sfx = ('' if args.precision == 'fp32' else '_' + args.precision) + wl_sfx
                rounds = ('r06', 'r05', 'r04', 'r03', 'r02', 'r01') if wl_sfx == '' else ('r06', 'r05')
                name = next(n for n in ['%s_pmc_traffic%s.json' % (r_, sfx) for r_ in rounds]
                            if os.path.exists(os.path.join(ROOT, 'profiles', n)))
                with open(os.path.join(ROOT, 'profiles', name)) as fh:
                    blob = json.load(fh)
                traffic_db = blob['kernels']
                traffic_build = blob.get('build_id')
                traffic_src = 'profiles/%s (static table from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, %s)' % (
                    name, blob.get('collected', 'an earlier binary'))
        except (OSError, KeyError, ValueError, StopIteration):
            traffic_db = {}

        def class_pattern(kind):
            """event class -> regular expression over rocprof's kernel names, e.g. the class
            conv_halo_frag_kernel<5,LU_EPI_LSTM,*,bf16> covers rocprof's conv_halo_frag2_kernel<5, 1, 8, true> and <5, 1, 4, true>
            (and the first-generation conv_halo_frag_kernel<5, 1, 8, false, ...> of older tables)."""
            name = kind.split(' ')[0]
            m = re.match(r'(\w+)<(\d)(?:,LU_EPI_(LSTM|BIAS))?', name)
            if name.startswith('conv_halo_frag_kernel<') and m:
                return r'conv_halo_frag[23]?_kernel<%s, %d, ' % (m.group(2), 1 if m.group(3) == 'LSTM' else 0)
            if name.startswith('wgrad_row_bf16_kernel<') and m:      # rocprof: wgrad_row_bf16_kernel<5, 128, true, true, 1, 64>
                return r'wgrad_row_bf16_kernel<%s, ' % m.group(2)
            if name.startswith('wgrad_row_kernel<') and m:           # rocprof: wgrad_row_kernel<5> (older tables) or <5, false>
                return r'wgrad_row_kernel<%s[,>]' % m.group(2)
            if m and m.group(3):                                     # conv_halo_kernel<5,LU_EPI_LSTM> -> <5, 1> (older tables), <5, 1, true / false>
                return re.escape('%s<%s, %d' % (m.group(1), m.group(2), 1 if m.group(3) == 'LSTM' else 0)) + '[,>]'
            return re.escape(name) + (r'[<(]' if '<' not in name else '')

        def traffic_of(kind):
            """launch-weighted mean over the rocprof kernel names of this event class"""
            pat = class_pattern(kind)
            hit = [v for k_, v in traffic_db.items() if re.search(pat, k_)]
            n = sum(v['launches'] for v in hit)
            return round(sum(v['traffic_bytes_per_launch'] * v['launches'] for v in hit) / n) if n else None

        # shader clock and MFMA-busy share per kernel class: a STATIC table as well (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES
        # GRBM_GUI_ACTIVE, tools/pmc_mfma.py), time-weighted over the class -- so that a power-throttled kernel is visible on the line
        # itself: frac_of_clocked_peak = achieved / (peak x clock / 2400 MHz)
        util_db, util_src, util_build = {}, None, None
        try:
            if wl_sfx is not None:
                rounds = ('r06', 'r05', 'r04', 'r03', 'r02') if wl_sfx == '' else ('r06', 'r05')
                name = next(n for n in ['%s_pmc_mfma_util%s.json' % (r_, wl_sfx) for r_ in rounds]
                            if os.path.exists(os.path.join(ROOT, 'profiles', n)))
                with open(os.path.join(ROOT, 'profiles', name)) as fh:
                    blob = json.load(fh)
                util_db, util_build, util_src = blob.get(args.precision, {}), blob.get('build_id'), 'profiles/' + name
        except (OSError, KeyError, ValueError, StopIteration):
            util_db = {}

        def clock_of(kind):
            pat = class_pattern(kind)
            hit = [v for k_, v in util_db.items() if re.search(pat, k_)]
            wt = sum(v['launches'] * v['avg_duration_us'] for v in hit)
            if not wt:
                return None, None
            clk = sum(v['shader_clock_ghz'] * v['launches'] * v['avg_duration_us'] for v in hit) / wt
            busy = sum(v['mfma_utilisation'] * v['launches'] * v['avg_duration_us'] for v in hit) / wt
            return clk, busy

        from lu_native.profile import summarize_events
        rows, hbm_rows = summarize_events(ev)
        if args.by_shape:
            from lu_native.profile import by_shape
            with open(args.by_shape, 'w') as fh:
                json.dump({'build_id': build_id, 'precision': args.precision, 'rows': by_shape(ev)}, fh, indent=1)
        if args.precision == 'bf16x3':
            annotate_x3(rows)
        for r_ in rows:
            r_['traffic'] = traffic_of(r_['kernel'])
            clk, busy = clock_of(r_['kernel'])
            if clk:
                r_['clock_mhz'] = round(1e3 * clk)
                r_['mfma_busy'] = round(busy, 4)
                r_['frac_of_clocked_peak'] = round(r_['achieved'] / (r_['peak'] * clk / 2.4), 4)
        if rows:
            roofline = dict(rows[0])            # the dominant kernel class = largest share of the step
            roofline['traffic_unit'] = ('L2-fabric bytes per launch INCLUDING Infinity-Cache hits (an upper bound on HBM bytes): '
                                        'rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes')
            roofline['traffic_source'] = traffic_src
            roofline['traffic_build_id'] = traffic_build
            roofline['traffic_stale'] = bool(traffic_db) and traffic_build != build_id      # table from another binary
            roofline['clock_source'] = util_src and ('%s (static table from a separate rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES '
                                                     'GRBM_GUI_ACTIVE pass; peak quoted at 2.4 GHz)' % util_src)
            roofline['clock_build_id'] = util_build
            roofline['clock_stale'] = bool(util_db) and util_build != build_id
            roofline['all_mfma_kernels'] = rows
            roofline['hbm_kernels'] = hbm_rows
    # ---- secondary metric: streaming inference (Inference2D.py:45-62: B=1, T=1, pad_image=True, stateful) ----
    infer = None
    if dp.rank == 0 and dp.world_size == 1 and not args.no_infer:
        import Networks
        m = Networks.ULSTMnet2D(net, 'NCHW', True, seed=0, precision=args.precision)
        frames_in = [torch.randn(1, 1, 1, H, W, device=dev) for _ in range(4)]
        for i in range(3):
            m(frames_in[i % 4], training=False)
        torch.cuda.synchronize()
        t_inf = time.perf_counter()
        n_inf = 40
        for i in range(n_inf):
            _, sm_ = m(frames_in[i % 4], training=False)
        torch.cuda.synchronize()
        infer = {'frames_per_s': round(n_inf / (time.perf_counter() - t_inf), 2), 'frames': n_inf,
                 'what': 'streaming forward, B=1 T=1, %dx%d (+reflect pad to %dx%d), softmax returned per frame' %
                         (H, W, H + 16, W + 16)}
        # ... and the whole per-frame path of Inference2D.py:45-131: forward + post-processing to the uint16 instance map
        # (GPU connected components / hole fill / edge absorption / relabel, label map copied to the host).  The model is
        # random-init, so its own softmax holds almost no cells: the post-processing is timed on a synthetic softmax with
        # ~60 cells per 256x256 frame (scaled with the area) swapped in behind the forward.
        import Inference2D
        from DataHandeling import SyntheticSequence2D
        prov = SyntheticSequence2D(image_crop_size=(H, W), unroll_len=1, batch_size=1, data_format='NCHW', seed=7, rank=0)
        seg = prov.get_batch()[1][0, 0, 0]                        # {-1,0,1,2} class map of synthetic cells
        seg = np.where(seg < 0, 0, seg).astype(np.int64)
        fake = torch.from_numpy(np.eye(3, dtype=np.float32)[seg].transpose(2, 0, 1) * 0.9 + 0.03).to(dev).contiguous()
        lab = Inference2D.postprocess(fake, 2, 10, 10 ** 6)
        torch.cuda.synchronize()
        # as Inference2D.inference() runs it: the post-processing of frame t on a side stream while frame t + 1's forward runs
        pipe = Inference2D.PostPipeline(2, 10, 10 ** 6)
        pipe.push(-2, fake)                  # first use: side stream, two processors, their pinned host buffers
        pipe.push(-1, fake)
        pipe.flush()
        torch.cuda.synchronize()
        runs = []          # three runs of n_inf frames: the first one after start-up reads low now and then (one process = one sample)
        for rep in range(3):
            t_pp = time.perf_counter()
            for i in range(n_inf):
                _, sm_ = m(frames_in[i % 4], training=False)
                for (_, lab, _) in pipe.push(i, fake):
                    pass
            for (_, lab, _) in pipe.flush():
                pass
            torch.cuda.synchronize()
            runs.append(round(n_inf / (time.perf_counter() - t_pp), 2))
        infer['frames_per_s_with_postprocess'] = sorted(runs)[1]      # median of three
        infer['frames_per_s_with_postprocess_runs'] = runs
        infer['postprocess_objects'] = int(lab.max())
        del m
    total_flops, _ = step_flops(net, H, W, B, T)
    # ---- secondary: the same step in the bf16 mixed-precision mode (BASELINE config-5 arithmetic), N = 1 only ----
    mixed = None
    if args.precision == 'fp32' and dp.world_size == 1 and not args.no_bf16:
        del trainer
        torch.cuda.empty_cache()
        tr16 = train2D.Trainer(Params.CTCParams.net_model, net, 'NCHW', Params.CTCParams.class_weights,
                               Params.CTCParams.learning_rate, dp=dp, sync_bn=args.sync_bn, seed=0, precision='bf16')

        def step16(i):
            img, seg, keep = batches[i % len(batches)]
            tr16.train_step(img, seg, want_outputs=True)
            tr16.model.reset_states_per_batch(keep)

        try:
            for i in range(args.warmup):
                step16(i)
            torch.cuda.synchronize()
            t16 = time.perf_counter()
            for i in range(args.steps):
                step16(args.warmup + i)
            torch.cuda.synchronize()
            e16 = time.perf_counter() - t16
            mixed = {'frames_per_s': round(B * T * args.steps / e16, 3), 'ms_per_step': round(1e3 * e16 / args.steps, 3),
                     'step_tflops_achieved': round(total_flops / 1e12 / (e16 / args.steps), 2), 'steps': args.steps,
                     'what': 'same workload with --precision bf16: bf16 MFMA operands (v_mfma_f32_32x32x16_bf16) on the '
                             'convolutions and weight gradients, fp32 accumulate / master weights / state / optimiser'}
        except Exception as exc:      # a secondary measurement: never lose the headline line over it
            mixed = {'failed': repr(exc)}
        del tr16
    # ---- secondary: the same step with precision 'bf16x3' -- fp32 ARITHMETIC on the bf16 MFMA (exact three-way bf16 split of the
    # ---- ConvLSTM operands, six bf16 products per fp32 product, fp32 accumulation; lu_native/engine.py), N = 1 only ----
    split3 = None
    if args.precision == 'fp32' and dp.world_size == 1 and not args.no_x3:
        try:
            del trainer
        except NameError:
            pass
        torch.cuda.empty_cache()
        try:
            split3 = measure_variant(net, 'bf16x3', batches, B, T, H, W, dp, args.steps, args.warmup)
        except Exception as exc:      # a secondary measurement: never lose the headline line over it
            split3 = {'failed': repr(exc)}
            torch.cuda.empty_cache()
    if split3 is not None and 'failed' not in split3:
        split3['what'] = ("same workload with --precision bf16x3: every ConvLSTM convolution (93 % of the FLOPs) and the wide stride-1 Conv2D units "
                          "on v_mfma_f32_32x32x16_bf16 "
                          "over the exact three-way bf16 split of their fp32 operands -- x = hi + mid + lo, six bf16 products per fp32 product, "
                          "each exact in the fp32 accumulator, the dropped ones below 2^-26 -- everything else on the fp32 kernels; nothing "
                          "is stored rounded.  'executed_bf16_tflops' / 'frac_of_bf16_peak': the bf16 MFMA FLOPs the pipe executes (6x the algorithmic "
                          "FLOPs on the split layers) against the 2.5 PFLOP/s bf16 peak; 'algorithmic_fp32_tflops' / 'fp32_equivalent_frames_per_s': "
                          "the step's fp32 FLOPs and frames over the same time")
    # ---- the kernel-size variants SURVEY D1 asks to report beside the headline net (N = 1, default headline run only) ----
    variants = None
    if dp.world_size == 1 and args.net == 'params' and not args.no_variants:
        try:
            del trainer
        except NameError:
            pass
        torch.cuda.empty_cache()
        variants = {}
        for vname in ('lstm3', 'default5'):
            variants[vname] = {'net': NET_WORDS[vname]}
            for prec in ('fp32', 'bf16', 'bf16x3'):
                try:
                    variants[vname][prec] = measure_variant(net_by_name(vname), prec, batches, B, T, H, W, dp, args.steps, args.warmup)
                except Exception as exc:      # (secondary measurements: the headline line survives them)
                    variants[vname][prec] = {'failed': repr(exc)}
                    torch.cuda.empty_cache()
    cpu = None
    if dp.rank == 0 and dp.world_size == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(Params.CTCParams.net_kernel_params)
        except Exception as exc:  # the baseline is informational; never lose the GPU line over it
            cpu = {'value': None, 'unit': 'frames/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': 'failed: %r' % (exc,)}
    if dp.rank == 0:
        line = {
            'metric': 'training frames/sec (seq_len*batch) at %dx%d' % (H, W),
            'value': round(frames_per_s, 3), 'unit': 'frames/s', 'n_gpus': dp.world_size, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': {'fp32': 'f32', 'bf16': 'bf16', 'bf16x3': 'f32 results via 3 x bf16 split'}[args.precision],
            'data': 'synthetic',
            'config': {'workload': ('BASELINE config-2 per GPU: ' if (H, W, T, B, args.net) == (256, 256, 8, 4, 'params') else
                                    ('BASELINE config-4: ' if (H, W, T, B) == (832, 992, 16, 2) else '')) +
                                   '%dx%d, seq_len=%d, batch=%d slots/GPU, ConvLSTM-UNet %s, %s, random-init' %
                                   (H, W, T, B, NET_WORDS[args.net], 'fp32' if args.precision == 'fp32' else
                                    'fp32 arithmetic, ConvLSTM convolutions as six bf16-MFMA products of the exact three-way bf16 split' if args.precision == 'bf16x3' else
                                    'bf16 MFMA operands on the wide stride-1 convs (fp32 master weights / accumulate / wgrad)'),
                       'net': args.net, 'global_batch': B * dp.world_size, 'seq_len': T,
                       'parallelism': 'dp%d' % dp.world_size, 'sync_bn': bool(args.sync_bn)},
            'build_id': lu_build.build_id(),      # sha256[:16] of liblstmunet_hip.so
            'dp': dp_info,
            'step_tflop_per_gpu': round(total_flops / 1e12, 2),
            'step_tflops_achieved_per_gpu': round(total_flops / 1e12 / (ms_per_step * 1e-3), 2),
            'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            'allocator': allocator,
            'inference': infer,
            'bf16_mode': mixed,
            'bf16x3_mode': split3,
            'bf16x3_summary': x3_summary(rows, ms_per_step * 1e-3, total_flops, B * T) if args.precision == 'bf16x3' else None,
            'variants': variants,
            'roofline': roofline, 'cpu_baseline': cpu,
        }
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
    if dp.collectives:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
