"""Whole-path parity: the product's engine / public API vs the oracle (numpy fp64 forward, torch fp64
autograd for gradients and the Adam update).

Gradient tolerances.  The loss is only piecewise smooth (hard-sigmoid and LeakyReLU kinks, BatchNorm over
as few as 1024 pixels), so its gradient is ill-conditioned: perturbing the WEIGHTS of the fp64 oracle by a
relative 1e-6 / 1e-5 moves its own gradients by up to 8e-3 / 4e-2 of a tensor's max at config-1 size
(measured with tests/diag/, notes in DESIGN.md §9).  fp32 arithmetic therefore cannot agree with fp64 better than
that on the big case (and the second step starts from fp32-drifted weights/state); the small cases (few kink
crossings) are held to 2e-3, config-1 to a stated multiple of the torch-fp32 oracle's own error on the same inputs (round 5,
see GRAD_TOL below: 5x / 3x, floors 5e-3 / 2e-3 -- measured 2.6e-3 / 1.1e-3 against torch-fp32's 1.1e-3 / 7e-4), and
`test_layerwise_backward_consistency` checks every backward kernel of the big case against an fp64
evaluation FROM THE SAME DEVICE INPUTS to 1e-6 (no chaos in that comparison).  'hip' = real gfx950 library (`-m gpu`); 'emu' = same
host code on the host-emulated kernels (CPU, small shapes) to validate tape/backward plumbing.

Stated tolerances: logits |d| <= 1e-3*max(1,|ref|) after T steps; gradients compared
relative to the largest reference gradient of that tensor; argmax maps bit-exact outside a top-2-gap
< 2e-3 tie band; SEG within 1e-3."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as npo
from oracle import torch_oracle as tho
from conftest import tiny_net, c1_net
from engine_backend import engine_backend

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def dev(request):
    with engine_backend(request.param) as d:
        yield d


def perturbed_params(net, cin, seed):
    rng = np.random.default_rng(seed)
    p = npo.init_params(net, cin, seed=seed, dtype=np.float32)
    for k in p:
        if k.endswith(('gamma', 'moving_var')):
            p[k] = (p[k] + 0.2 * rng.random(p[k].shape)).astype(np.float32)
        elif k.endswith(('beta', 'bias', 'moving_mean')):
            p[k] = (p[k] + 0.1 * rng.standard_normal(p[k].shape)).astype(np.float32)
    return p


def make_engine(net, p, cin, dev, pad_image=False):
    from lu_native.engine import Engine
    e = Engine(net, pad_image=pad_image)
    e.build(cin, dev)
    e.load_params(p)
    return e


def to_tb(x):   # [B,T,H,W,C] -> time-major frames
    B, T = x.shape[:2]
    return np.ascontiguousarray(np.swapaxes(x, 0, 1)).reshape((T * B,) + x.shape[2:])


def from_tb(y, B, T):
    return np.swapaxes(y.reshape((T, B) + y.shape[1:]), 0, 1)


def rel_err(a, b, floor=0.0):
    """max|a-b| relative to max|b|; `floor` keeps tensors whose true gradient is ~0 (conv biases in
    front of BatchNorm: the mean subtraction cancels them exactly) from dividing by rounding noise."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max(), floor))


def grad_floor(grads):
    return 1e-3 * max(float(np.abs(np.asarray(g)).max()) for g in grads.values())


CASES = [  # name, net, cin, B, T, H, W, pad_image
    ('unfused-k3', tiny_net(3), 1, 2, 2, 16, 16, False),
    ('fused-k3-pad', tiny_net(3, (32, 8, 8, 32), (8, 8, 8, 8)), 1, 1, 2, 13, 16, True),
    ('fused-train', tiny_net(3, (8, 32, 8, 32), (8, 8, 8, 8)), 1, 1, 3, 16, 16, False),
    # 2 levels, stacked ConvLSTMs per block with different kernel sizes, 3 convs in a block, 1x1 conv, 2 input channels
    ('two-level-stacked-lstm', {'down_conv_kernels': [[(3, 8), (3, 8), (3, 4)], [(1, 8)]],
                                'lstm_kernels': [[(3, 8), (5, 4)], [(3, 12)]],
                                'up_conv_kernels': [[(3, 8)], [(3, 8), (1, 3)]]}, 2, 2, 2, 10, 12, False),
]
GPU_CASES = [
    ('c1', c1_net(), 1, 1, 4, 128, 128, False),
    ('k5-odd', tiny_net(5, (32, 64, 32, 64), (32, 16, 16, 8)), 3, 2, 3, 35, 35, True),
]
# config-1 end-to-end against the fp64 oracle.  Rounds 2-3 stated "2x what the HIP path measured" (0.055 / 0.02, from 2.7e-2 / 6.3e-3 on
# the kernels of that time).  Round 5: the tolerance is a stated multiple of what an INDEPENDENT fp32 implementation -- the torch
# oracle run in fp32 on the same inputs, weights and carried state, inline in the test -- is away from the same fp64 gradients:
#     worst tensor, max-abs / tensor-max:  HIP <= C1_MAXABS_X * torch-fp32's worst   (floor 5e-3)
#     worst tensor, L2-relative:           HIP <= C1_L2_X     * torch-fp32's worst   (floor 2e-3)
# MEASURED on the MI355X (profiles/r05_c1_three_way.log, tests/diag/diag3_gpu.py): step 0 torch-fp32 1.10e-3 / 7.1e-4, HIP 2.58e-3 /
# 1.14e-3 (medians 2.3e-4 / 2.1e-4 vs 2.9e-4 / 3.1e-4); step 1 torch-fp32 1.76e-3 / 7.9e-4, HIP 1.10e-3 / 5.4e-4.  The product's
# forward noise at level 0 is 3x torch's (h: 1.8e-7 vs 6.3e-8 max, the 15-instruction tanh of DESIGN 1.2: 1.5e-7 absolute), 1.2-1.4x
# below; every backward kernel of this case is held to 1e-5 from the same device inputs (test_layerwise_backward_consistency,
# test_lstm_bptt_backward_consistency).
GRAD_TOL = {'c1': None}      # name -> None: calibrated in the test (GPU_CASES with an inline torch-fp32 pass)
C1_MAXABS_X, C1_L2_X, C1_MAXABS_FLOOR, C1_L2_FLOOR = 5.0, 3.0, 5e-3, 2e-3


def _all_cases(request_dev):
    return CASES + (GPU_CASES if request_dev.type == 'cuda' else [])


@pytest.fixture(autouse=True)
def _force_fused_path(monkeypatch):
    """Small test shapes would take the tile-starved (unfused, K-split) ConvLSTM route; force the fused epilogue
    kernel wherever F %% 32 == 0 so both routes are covered (the unfused one by the F < 32 layers)."""
    from lu_native import ops
    monkeypatch.setattr(ops, 'FUSED_MIN_TILES', 0)


def test_forward_and_inference_parity(dev):
    for name, net, cin, B, T, H, W, pad in _all_cases(dev):
        rng = np.random.default_rng(1)
        x = rng.standard_normal((B, T, H, W, cin)).astype(np.float32)
        p = perturbed_params(net, cin, 3)
        for training in (True, False):
            e = make_engine(net, p, cin, dev, pad)
            ref = npo.model_forward(net, p, x, training=training, pad_image=pad, update_moving=True)
            lg = e.forward(torch.from_numpy(to_tb(x)).to(dev), T, B, training)
            got = from_tb(lg.cpu().numpy(), B, T)
            tol = 1e-3 * max(1.0, float(np.abs(ref['logits']).max()))
            assert np.abs(got - ref['logits']).max() <= tol, (name, training)
            # carried state after the window
            for bi, blk in enumerate(ref['states']):
                for li, (h, c) in enumerate(blk):
                    assert np.abs(e.states[bi][li][0].cpu().numpy() - h).max() <= 1e-3, (name, 'h', bi)
                    assert np.abs(e.states[bi][li][1].cpu().numpy() - c).max() <= 1e-3, (name, 'c', bi)
            if training:
                for k, v in ref['moving'].items():
                    assert np.abs(e.S[k].cpu().numpy() - v).max() <= 1e-4, (name, k)
            # second window continues from the carried state, after a keep-mask
            keep = np.array([1.0, 0.0][:B], np.float32)
            e.tape = None
            e.reset_states_per_batch(keep)
            st2 = npo.reset_states_per_batch(ref['states'], keep)
            ref2 = npo.model_forward(net, p if not training else {**p, **ref['moving']}, x[:, ::-1], states=st2,
                                     training=training, pad_image=pad)
            lg2 = e.forward(torch.from_numpy(to_tb(x[:, ::-1])).to(dev), T, B, training)
            assert np.abs(from_tb(lg2.cpu().numpy(), B, T) - ref2['logits']).max() <= tol, (name, 'window2')
            # argmax label maps: bit-exact outside the tie band
            top2 = np.sort(ref2['logits'], -1)
            band = (top2[..., -1] - top2[..., -2]) < 2e-3
            agree = from_tb(lg2.cpu().numpy(), B, T).argmax(-1) == ref2['logits'].argmax(-1)
            assert np.all(agree | band), name


def test_train_step_parity(dev):
    """loss, every gradient tensor, Adam-updated weights and carried state of one optimiser step
    (train2D.py:87-95), then a second step from the carried state."""
    from lu_native.engine import Adam
    from lu_native import ops
    cw = [0.15, 0.25, 0.6]
    for name, net, cin, B, T, H, W, pad in _all_cases(dev):
        if pad:
            continue    # training uses pad_image=False (train2D.py:46)
        rng = np.random.default_rng(5)
        p = perturbed_params(net, cin, 7)
        e = make_engine(net, p, cin, dev, False)
        opt = Adam(e, lr=1e-3)
        tm = tho.TorchULSTM(net, cin, p, dtype=torch.float64)
        # calibration (round 5): the torch oracle in fp32 on the same inputs, weights and carried state -- an independent fp32
        # implementation whose own distance from the fp64 gradients is printed next to the product's
        t32 = tho.TorchULSTM(net, cin, p, dtype=torch.float32) if name in GRAD_TOL else None
        cwt = torch.tensor(cw, dtype=torch.float32, device=dev)
        for step in range(2):
            x = rng.standard_normal((B, T, H, W, cin)).astype(np.float32)
            gt = rng.integers(-1, 3, size=(B, T, H, W)).astype(np.float32)
            loss_ref, logits_ref, grads_ref = tm.train_step(x, gt, cw, lr=1e-3)
            if t32 is not None:
                _, _, g32 = t32.train_step(x, gt, cw, lr=1e-3, apply=False)
                fl32 = grad_floor({k: v.numpy() for k, v in grads_ref.items()})
                w32 = max((rel_err(g32[k].numpy(), grads_ref[k].numpy(), fl32), k) for k in grads_ref)
                l32 = max((float(np.linalg.norm(g32[k].numpy().astype(np.float64) - grads_ref[k].numpy()) /
                                 max(np.linalg.norm(grads_ref[k].numpy()), fl32 * (3.0 if '.conv.' in k and k.endswith('.bias') else 1.0))),
                           k) for k in grads_ref)
                print('train_step_parity %s step %d: torch-fp32 oracle vs fp64: worst max-rel %.3e (%s), worst L2-rel %.3e (%s)' %
                      (name, step, w32[0], w32[1], l32[0], l32[1]))
            lg = e.forward(torch.from_numpy(to_tb(x)).to(dev), T, B, True)
            g = torch.from_numpy(to_tb(gt[..., None])).to(dev).view(-1)
            sums, _ = ops.wce_forward(lg.view(-1, 3), g, cwt, False)
            dl = ops.wce_backward(lg.view(-1, 3), g, cwt, sums, 1.0)
            e.backward(dl.view(lg.shape))
            loss = float(ops.wce_loss(sums).cpu()[0])
            assert abs(loss - float(loss_ref)) <= 1e-4 * max(1.0, abs(float(loss_ref))), (name, step)
            fl = grad_floor({k: v.numpy() for k, v in grads_ref.items()})
            worst = max((rel_err(e.G[k].cpu().numpy(), grads_ref[k].numpy(), fl), k) for k in grads_ref)
            tol_max = 2e-3 if t32 is None else max(C1_MAXABS_X * w32[0], C1_MAXABS_FLOOR)
            tol_l2 = 2e-3 if t32 is None else max(C1_L2_X * l32[0], C1_L2_FLOOR)
            assert worst[0] <= tol_max, (name, step, worst, tol_max)
            # (a conv bias in front of BatchNorm has an exactly-zero true gradient -- the mean subtraction cancels it -- so the
            # product's value there is pure fp32 rounding noise: its floor is wider)
            l2 = max((float(np.linalg.norm(e.G[k].cpu().numpy().astype(np.float64) - grads_ref[k].numpy()) /
                            max(np.linalg.norm(grads_ref[k].numpy()), fl * (3.0 if '.conv.' in k and k.endswith('.bias') else 1.0))),
                      k) for k in grads_ref)
            assert l2[0] <= tol_l2, (name, step, l2, tol_l2)
            print('train_step_parity %s step %d: worst max-rel %.3e (%s), worst L2-rel %.3e (%s)' %
                  (name, step, worst[0], worst[1], l2[0], l2[1]))
            opt.apply_gradients()
            perr = max(float(np.abs(e.P[k].cpu().numpy() - tm.P[k].numpy()).max()) for k in grads_ref)
            # Adam's first steps move every weight by ~lr; sign flips of tiny gradients can cost up to 2*lr
            n_bad = sum(int((np.abs(e.P[k].cpu().numpy() - tm.P[k].numpy()) > 2e-4).sum()) for k in grads_ref)
            n_all = sum(int(np.prod(tm.P[k].shape)) for k in grads_ref)
            assert perr <= 2.5e-3 and n_bad <= (1e-3 if name not in GRAD_TOL else 2e-2) * n_all, \
                (name, step, perr, n_bad, n_all)
            keep = np.ones(B, np.float32)
            keep[-1] = 0.0
            e.reset_states_per_batch(keep)
            tm.reset_states_per_batch(keep)
            # keep the two trajectories glued: continue the oracle from the product's weights
            for k in grads_ref:
                tm.P[k] = torch.tensor(e.P[k].cpu().numpy(), dtype=torch.float64)
                if t32 is not None:
                    t32.P[k] = torch.tensor(e.P[k].cpu().numpy(), dtype=torch.float32)
            if t32 is not None:
                t32.reset_states_per_batch(keep)


def test_head_depth_other_than_three(dev):
    """Networks.py:201-206: `last_depth` is whatever the last up-block kernel says and the model's second output is the softmax
    over that many classes (the reference's loss hard-codes 3 classes; the model does not)."""
    import Networks
    net = tiny_net(3)
    net['up_conv_kernels'][-1][-1] = (1, 5)
    rng = np.random.default_rng(11)
    x = rng.standard_normal((1, 2, 1, 16, 16)).astype(np.float32)
    m = Networks.ULSTMnet2D(net, 'NCHW', False, seed=2)
    assert m.last_depth == 5
    logits, sm = m(x, training=False)
    assert tuple(logits.shape) == (1, 2, 5, 16, 16) and tuple(sm.shape) == (1, 2, 5, 16, 16)
    lg = logits.cpu().numpy().astype(np.float64)
    want = np.exp(lg - lg.max(2, keepdims=True))
    want /= want.sum(2, keepdims=True)
    assert np.abs(sm.cpu().numpy() - want).max() <= 1e-6
    p = m.engine.export_params()
    ref = npo.model_forward(net, p, np.transpose(x, (0, 1, 3, 4, 2)), training=False, pad_image=False)
    assert np.abs(np.transpose(lg, (0, 1, 3, 4, 2)) - ref['logits']).max() <= 1e-3 * max(1.0, np.abs(ref['logits']).max())


def test_public_api_and_autograd_path(dev):
    import Networks
    import losses
    net = tiny_net(3)
    p = perturbed_params(net, 1, 2)
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 2, 16, 16, 1)).astype(np.float32)
    gt = rng.integers(-1, 3, size=(2, 2, 16, 16, 1)).astype(np.float32)
    ref = npo.model_forward(net, p, x, training=True, pad_image=False)
    out = {}
    for fmt in ('NCHW', 'NHWC'):
        m = Networks.ULSTMnet2D(net, fmt, False)
        assert m.get_states()[0][0] == [None, None]
        m.engine.build(1, dev)
        m.engine.load_params(p)
        xin = np.transpose(x, (0, 1, 4, 2, 3)) if fmt == 'NCHW' else x
        gin = np.transpose(gt, (0, 1, 4, 2, 3)) if fmt == 'NCHW' else gt
        logits, sm = m(xin, True)
        assert tuple(logits.shape) == ((2, 2, 3, 16, 16) if fmt == 'NCHW' else (2, 2, 16, 16, 3))
        lcl = logits.detach().cpu().numpy()
        lcl = np.transpose(lcl, (0, 1, 3, 4, 2)) if fmt == 'NCHW' else lcl
        assert np.abs(lcl - ref['logits']).max() <= 1e-3
        smc = sm.cpu().numpy()
        smc = np.transpose(smc, (0, 1, 3, 4, 2)) if fmt == 'NCHW' else smc
        assert np.abs(smc - ref['softmax']).max() <= 1e-4
        ce = losses.WeightedCELoss(2 if fmt == 'NCHW' else 4, [0.15, 0.25, 0.6])
        loss = ce(gin, logits)
        assert abs(float(loss) - npo.weighted_ce(gt[..., 0], ref['logits'], [0.15, 0.25, 0.6])) <= 1e-4
        loss.backward()
        out[fmt] = m.parameters()[0].grad.cpu().numpy().copy()
        st = m.get_states()
        assert st[0][0][0].shape == (2, 16, 16, 8)
        m.set_states(st)
        m.reset_states_per_batch(np.array([0.0, 1.0], np.float32))
        assert float(np.abs(m.get_states()[0][0][0][0]).max()) == 0.0
    assert np.abs(out['NCHW'] - out['NHWC']).max() <= 1e-6
    tm = tho.TorchULSTM(net, 1, p, dtype=torch.float64)
    _, _, grads = tm.train_step(x, gt[..., 0], [0.15, 0.25, 0.6], apply=False)
    eng = m.engine
    fl = grad_floor({k: v.numpy() for k, v in grads.items()})
    worst = max(rel_err(eng.G[k].cpu().numpy(), grads[k].numpy(), fl) for k in grads)
    assert worst <= 2e-3


BF16_NET = {'down_conv_kernels': [[(3, 72), (3, 72)], [(3, 8)]], 'lstm_kernels': [[(3, 32)], [(5, 32)]],
            'up_conv_kernels': [[(3, 72)], [(3, 72), (1, 3)]]}


# wide first level: F = 96 > 64 puts the recurrent gradient, the hoisted weight gradients and the gate backward on the
# bf16-tape kernels (bf16 h sequence / dz as MFMA operands); the narrow nets above exercise the fp32 fall-backs
BF16_NET_WIDE = {'down_conv_kernels': [[(3, 72)], [(3, 8)]], 'lstm_kernels': [[(3, 96)], [(3, 32)]],
                 'up_conv_kernels': [[(3, 72)], [(3, 72), (1, 3)]]}


@pytest.mark.parametrize('which', ['narrow', 'wide'])
def test_bf16_precision_mode(dev, monkeypatch, which):
    """Engine(precision='bf16') (BASELINE config 5): the wide stride-1 convolutions run on the bf16-MFMA kernel and the
    step stays close to the fp32 step.  Stated tolerances: logits within 3e-2 * max|logit| of the fp32 engine, loss within
    2e-2 relative, label maps equal outside a 5e-2 top-2 tie band, every gradient tensor within 0.1 L2-relative."""
    from lu_native import calls, ops
    from lu_native.engine import Engine
    seen = []
    real = calls.conv2d
    monkeypatch.setattr(calls, 'conv2d', lambda *a, **k: (seen.append(k.get('precision', 0)), real(*a, **k))[1])
    net, cin, B, T, H, W = (BF16_NET, 1, 2, 3, 16, 32) if which == 'narrow' else (BF16_NET_WIDE, 1, 1, 3, 8, 32)
    rng = np.random.default_rng(21)
    p = perturbed_params(net, cin, 4)
    x = rng.standard_normal((B, T, H, W, cin)).astype(np.float32)
    gt = rng.integers(-1, 3, size=(B, T, H, W)).astype(np.float32)
    cwt = torch.tensor([0.15, 0.25, 0.6], dtype=torch.float32, device=dev)
    res = {}
    for prec in ('fp32', 'bf16'):
        del seen[:]
        e = Engine(net, pad_image=False, precision=prec)
        e.build(cin, dev)
        e.load_params(p)
        lg = e.forward(torch.from_numpy(to_tb(x)).to(dev), T, B, True)
        g = torch.from_numpy(to_tb(gt[..., None])).to(dev).view(-1)
        sums, _ = ops.wce_forward(lg.view(-1, 3), g, cwt, False)
        e.backward(ops.wce_backward(lg.view(-1, 3), g, cwt, sums, 1.0).view(lg.shape))
        res[prec] = (lg.cpu().numpy().astype(np.float64), float(ops.wce_loss(sums).cpu()[0]),
                     {k: v.cpu().numpy().astype(np.float64) for k, v in e.G.items()})
        n_bf16 = sum(seen)
        assert (n_bf16 == 0) if prec == 'fp32' else (n_bf16 >= 2 * T + 6), (prec, n_bf16, len(seen))
    (l32, loss32, g32), (l16, loss16, g16) = res['fp32'], res['bf16']
    assert np.abs(l16 - l32).max() <= 3e-2 * np.abs(l32).max()
    assert np.abs(l16 - l32).max() > 0.0
    assert abs(loss16 - loss32) <= 2e-2 * abs(loss32)
    top2 = np.sort(l32, -1)
    band = (top2[..., -1] - top2[..., -2]) < 5e-2 * np.abs(l32).max()
    assert np.all((l16.argmax(-1) == l32.argmax(-1)) | band)
    fl = grad_floor(g32)
    worst = max((float(np.linalg.norm(g16[k] - g32[k]) / max(np.linalg.norm(g32[k]), fl)), k) for k in g32)
    assert worst[0] <= 0.1, worst
    ref = npo.model_forward(net, p, x, training=True, pad_image=False)           # and both sit next to the oracle
    assert np.abs(from_tb(l16, B, T) - ref['logits']).max() <= 3e-2 * np.abs(ref['logits']).max()


ACT16_NET = {'down_conv_kernels': [[(3, 64), (3, 64)], [(3, 72), (3, 72)], [(3, 40)]],
             'lstm_kernels': [[(5, 32)], [(3, 32)], [(3, 24)]],
             'up_conv_kernels': [[(3, 64), (3, 64)], [(3, 64), (3, 40)], [(3, 32), (3, 32), (1, 3)]]}


def test_bf16_activations_stored_as_bf16_change_no_value(dev):
    """bf16 mode stores the activations whose every consumer rounds them to bf16 MFMA operands AS bf16 (BatchNorm'd outputs
    inside / between blocks, the up-sampled decoder inputs; Engine.act_bf16) -- and, by the same argument, the gradient a
    BatchNorm backward hands to its convolution when that layer's weight gradient and input gradient both run on bf16 operands
    (Engine.grad_bf16).  The claim is that no kernel then computes with a
    different value: logits, loss and EVERY gradient tensor of a training step must be bit-identical to the same step with fp32
    storage.  The net covers: bf16 block outputs feeding a ConvLSTM, a skip convolution and (last block) the first up block;
    up-sampled bf16 tensors next to a bf16 skip and next to the 1-channel image (padded to 8 bf16 channels); N = 32 / 40 / 64 / 72
    units, a 40-channel tensor (C % 8 == 0) and the 1x1 logits convolution, whose input stays fp32."""
    from lu_native import ops
    from lu_native.engine import Engine
    net, cin, B, T, H, W = ACT16_NET, 1, 2, 2, 16, 128      # (widths 128 / 64 / 32: every level inside the bf16 weight gradient's domain)
    rng = np.random.default_rng(33)
    p = perturbed_params(net, cin, 5)
    x = rng.standard_normal((B, T, H, W, cin)).astype(np.float32)
    gt = rng.integers(-1, 3, size=(B, T, H, W)).astype(np.float32)
    cwt = torch.tensor([0.15, 0.25, 0.6], dtype=torch.float32, device=dev)
    res, n16, g16 = {}, {}, {}
    real, real_bwd = ops.bn_lrelu_apply, ops.bn_lrelu_bwd_apply
    modes = [(True, True), (True, False), (False, False)]       # (activations, BatchNorm-backward gradients) stored as bf16
    if dev.type != 'cuda':
        modes = [modes[0], modes[2]]       # (the emulator run keeps the CPU suite short: the mixed mode on the MI355X only)
    for mode in modes:
        count, countg = [0], [0]

        def spy(*a, **k):
            count[0] += int(bool(k.get('out_bf16')))
            return real(*a, **k)

        def spy_bwd(*a, **k):
            countg[0] += int(bool(k.get('out_bf16')))
            return real_bwd(*a, **k)
        ops.bn_lrelu_apply, ops.bn_lrelu_bwd_apply = spy, spy_bwd
        try:
            e = Engine(net, pad_image=False, precision='bf16')
            e.act_bf16, e.grad_bf16 = mode
            e.overlap_wgrad = False
            e.build(cin, dev)
            e.load_params(p)
            lg = e.forward(torch.from_numpy(to_tb(x)).to(dev), T, B, True)
            g = torch.from_numpy(to_tb(gt[..., None])).to(dev).view(-1)
            sums, _ = ops.wce_forward(lg.view(-1, 3), g, cwt, False)
            e.backward(ops.wce_backward(lg.view(-1, 3), g, cwt, sums, 1.0).view(lg.shape))
        finally:
            ops.bn_lrelu_apply, ops.bn_lrelu_bwd_apply = real, real_bwd
        n16[mode], g16[mode] = count[0], countg[0]
        res[mode] = (lg.cpu().numpy(), float(ops.wce_loss(sums).cpu()[0]), {k: v.cpu().numpy() for k, v in e.G.items()})
    assert n16[(True, True)] >= 6 and n16[(False, False)] == 0, n16          # the switches do something
    assert g16[(True, True)] >= 6 and g16.get((True, False), 0) == 0 and g16[(False, False)] == 0, g16
    lb, lossb, gb = res[(False, False)]
    # a conv bias in front of a BatchNorm: its true gradient is zero (the mean subtraction cancels it), what fp32 storage
    # computes is the rounding noise of a column sum
    bn_bias = {k for k in gb if k.endswith('.bias') and k.replace('.conv.', '.bn.').replace('.bias', '.gamma') in gb}
    assert len(bn_bias) >= 10
    for mode in modes[:-1]:
        la, lossa, ga = res[mode]
        assert np.array_equal(la, lb) and lossa == lossb
        bad = [k for k in ga if not np.array_equal(ga[k], gb[k]) and not (mode[1] and k in bn_bias)]
        assert not bad, (mode, bad)
        # ... fp32 storage: noise (<= 1e-5 of the largest gradient); bf16 storage: the exact value, zero, wherever the layer
        # qualifies (a column sum of ROUNDED values would be noise that Adam turns into full-size steps)
        scale = max(float(np.abs(v).max()) for v in gb.values())
        assert all(float(np.abs(ga[k]).max()) <= 1e-5 * scale for k in bn_bias), mode
        if mode[1]:
            assert sum(1 for k in bn_bias if not ga[k].any()) >= 6


@pytest.mark.gpu
def test_bf16_mode_ragged_inference_shapes():
    """bf16 mode on frame sizes that are not multiples of anything (35 x 37, reflect-padded to 48 x 48 inside): streaming
    forward with carried state, every frame within 3e-2 * max|logit| of the fp32 engine, labels equal outside the tie band."""
    from lu_native.engine import Engine
    net, cin = BF16_NET, 1
    rng = np.random.default_rng(8)
    p = perturbed_params(net, cin, 5)
    frames = [rng.standard_normal((1, 35, 37, cin)).astype(np.float32) for _ in range(3)]
    outs = {}
    for prec in ('fp32', 'bf16'):
        e = Engine(net, pad_image=True, precision=prec)
        e.build(cin, torch.device('cuda', 0))
        e.load_params(p)
        outs[prec] = [e.forward(torch.from_numpy(f).cuda(), 1, 1, False).cpu().numpy().astype(np.float64) for f in frames]
    for a, b in zip(outs['fp32'], outs['bf16']):
        assert a.shape == (1, 35, 37, 3)
        assert np.abs(a - b).max() <= 3e-2 * np.abs(a).max()
        top2 = np.sort(a, -1)
        band = (top2[..., -1] - top2[..., -2]) < 5e-2 * np.abs(a).max()
        assert np.all((a.argmax(-1) == b.argmax(-1)) | band)


@pytest.mark.gpu
def test_layerwise_backward_consistency():
    """Config-1 backward on the GPU: every Conv->BN->LeakyReLU unit's backward (BN sums, input gradient
    of the BN, weight gradient) is re-evaluated in fp64 torch ON THE DEVICE from the very tensors the
    kernels consumed -- a well-conditioned check of each backward kernel at real layer shapes."""
    import torch.nn.functional as F
    from lu_native import ops, engine as eng_mod
    from lu_native.calls import same_pad
    with engine_backend('hip') as dev:
        net, cin, B, T, H, W = c1_net(), 1, 1, 4, 128, 128
        rng = np.random.default_rng(5)
        p = perturbed_params(net, cin, 7)
        x = rng.standard_normal((B, T, H, W, cin)).astype(np.float32)
        gt = rng.integers(-1, 3, size=(B, T, H, W)).astype(np.float32)
        cwt = torch.tensor([0.15, 0.25, 0.6], dtype=torch.float32, device=dev)
        e = make_engine(net, p, cin, dev, False)
        orig = eng_mod.Engine._conv_unit_backward
        report = []

        def patched(self, rec, dz, need_dx):
            if not rec['bn']:
                return orig(self, rec, dz, need_dx)
            y = rec['y']
            d64, y64 = dz.double(), y.double()
            sc, sh, mean, inv = [rec[k].double() for k in ('scale', 'shift', 'mean', 'invstd')]
            dzp = d64 * torch.where(y64 * sc + sh > 0, 1.0, 0.3)
            xhat = (y64 - mean) * inv
            s0, s1 = dzp.sum(dim=(0, 1, 2)), (dzp * xhat).sum(dim=(0, 1, 2))
            n = y.numel() // y.shape[-1]
            dy_ref = sc * (dzp - s0 / n - xhat * s1 / n)
            srcs = rec['srcs']
            prefix, ci, stride = rec['prefix'], rec['ci'], rec['spec']['stride']
            out = orig(self, rec, dz, need_dx)           # dz now holds dy (in place)
            err_dy = float((dz.double() - dy_ref).abs().max() / dy_ref.abs().max())
            gw = self.G[f'{prefix}.conv.{ci}.kernel']
            w = self.P[f'{prefix}.conv.{ci}.kernel']
            k = gw.shape[0]
            errs_w, errs_x = [], []
            for (xin, co, cs), dx in zip(srcs, out):
                x64 = xin.double()
                _, pt, pb = same_pad(x64.shape[1], k, stride)
                _, pl, pr = same_pad(x64.shape[2], k, stride)
                xn = F.pad(x64.permute(0, 3, 1, 2), (pl, pr, pt, pb)).requires_grad_(True)
                w64 = w[:, :, co:co + cs, :].double().permute(3, 2, 0, 1).contiguous().requires_grad_(True)
                yy = F.conv2d(xn, w64, None, stride=stride)
                gx, gwr = torch.autograd.grad(yy, [xn, w64], dy_ref.permute(0, 3, 1, 2).contiguous())
                gwr = gwr.permute(2, 3, 1, 0)
                errs_w.append(float((gw[:, :, co:co + cs, :].double() - gwr).abs().max() / gwr.abs().max()))
                if dx is not None:
                    gx = gx[:, :, pt:pt + x64.shape[1], pl:pl + x64.shape[2]].permute(0, 2, 3, 1)
                    errs_x.append(float((dx.double() - gx).abs().max() / gx.abs().max()))
            report.append((f'{prefix}.{ci}', err_dy, max(errs_w), max(errs_x) if errs_x else 0.0))
            return out

        eng_mod.Engine._conv_unit_backward = patched
        try:
            lg = e.forward(torch.from_numpy(to_tb(x)).to(dev), T, B, True)
            g = torch.from_numpy(to_tb(gt[..., None])).to(dev).view(-1)
            sums, _ = ops.wce_forward(lg.view(-1, 3), g, cwt, False)
            e.backward(ops.wce_backward(lg.view(-1, 3), g, cwt, sums, 1.0).view(lg.shape))
            torch.cuda.synchronize()
        finally:
            eng_mod.Engine._conv_unit_backward = orig
        assert len(report) == 16
        for name, e_dy, e_w, e_x in report:
            assert e_dy <= 1e-5 and e_w <= 1e-5 and e_x <= 1e-5, (name, e_dy, e_w, e_x)


def _bptt_case(name):
    if name == 'c1':
        return c1_net(), 1, 1, 4, 128, 128
    import Params
    return Params.CTCParams.net_kernel_params, 1, 2, 3, 64, 64     # Params.py widths (5x5 ConvLSTM 128/256/256/512)


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['c1', 'params-width-64'])
def test_lstm_bptt_backward_consistency(case):
    """The ConvLSTM BPTT chain at real layer shapes, re-evaluated in fp64 FROM THE SAME DEVICE TENSORS the kernels
    consumed (saved post-activation gates, c_all, h_all, x, dh_seq): per step dz_t (gate backward -> recurrent dgrad ->
    next gate backward, i.e. the whole chain t = T-1..0, linear in dh_seq once the saved gates fix the hard-sigmoid
    branches), then -- from the DEVICE dz -- the three hoisted gradients (kernel, recurrent_kernel, bias) and the
    input gradient of the layer.  Well-conditioned (no kink can flip between the two evaluations), so held to 1e-5 of
    each tensor's maximum.  Reference semantics: Keras ConvLSTM2D cell as constructed at Networks.py:48-50, BPTT
    truncated at the window edge (stateful=True), train2D.py:89-93."""
    import torch.nn.functional as F
    from lu_native import ops, engine as eng_mod
    with engine_backend('hip') as dev:
        net, cin, B, T, H, W = _bptt_case(case)
        rng = np.random.default_rng(11)
        p = perturbed_params(net, cin, 7)
        x = rng.standard_normal((B, T, H, W, cin)).astype(np.float32)
        gt = rng.integers(-1, 3, size=(B, T, H, W)).astype(np.float32)
        cwt = torch.tensor([0.15, 0.25, 0.6], dtype=torch.float32, device=dev)
        e = make_engine(net, p, cin, dev, False)
        orig = eng_mod.Engine._lstm_backward
        report = []

        def conv64(inp, w, k):      # SAME stride-1 cross-correlation, channels-last in / out, fp64
            return F.conv2d(inp.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), None, padding=(k - 1) // 2).permute(0, 2, 3, 1)

        def patched(self, rec, dh_seq, need_dx):
            bi, li, spec, T_, B_ = rec['bi'], rec['li'], rec['spec'], rec['T'], rec['B']
            pre = f'down.{bi}.lstm.{li}'
            k = spec['k']
            gates_dev = rec['gates']                       # dz is written in place of the saved gates
            g64 = gates_dev.double().clone()
            c64, h64, x64 = rec['c_all'].double(), rec['h_all'].double(), rec['x'].double()
            Fh = h64.shape[-1]
            dh64 = dh_seq.double().clone().view(T_, B_, h64.shape[2], h64.shape[3], Fh)
            Wk, Wr = self.P[pre + '.kernel'].double(), self.P[pre + '.recurrent_kernel'].double()
            dx = orig(self, rec, dh_seq, need_dx)
            dz_dev = gates_dev.double()
            hsg = lambda a: torch.where((a > 0) & (a < 1), 0.2, 0.0).to(a.dtype)     # noqa: E731
            dz_ref = torch.empty_like(g64)
            dc_next, dh_rec = None, None
            for t in reversed(range(T_)):
                gi, gf, gg, go = g64[t].split(Fh, dim=-1)
                dh = dh64[t] if dh_rec is None else dh64[t] + dh_rec
                tc = torch.tanh(c64[t + 1])
                dc = dh * go * (1 - tc * tc)
                if dc_next is not None:
                    dc = dc + dc_next
                dz_ref[t] = torch.cat([dc * gg * hsg(gi), dc * c64[t] * hsg(gf), dc * gi * (1 - gg * gg),
                                       dh * tc * hsg(go)], -1)
                dc_next = dc * gf
                if t > 0:
                    hp = torch.zeros_like(h64[t], requires_grad=True)
                    (dh_rec,) = torch.autograd.grad(conv64(hp, Wr, k), hp, dz_ref[t])
            scale = float(dz_ref.abs().max())
            err_dz = float((dz_dev - dz_ref).abs().max()) / scale
            # hoisted gradients from the device dz
            dzf = dz_dev.view((T_ * B_,) + tuple(dz_dev.shape[2:]))
            wr = Wr.clone().requires_grad_(True)
            (gr,) = torch.autograd.grad(conv64(h64[:T_].reshape((T_ * B_,) + tuple(h64.shape[2:])), wr, k), wr, dzf)
            wk = Wk.clone().requires_grad_(True)
            xin = x64.clone().requires_grad_(True)
            gk, gx = torch.autograd.grad(conv64(xin, wk, k), [wk, xin], dzf)
            gb = dzf.sum(dim=(0, 1, 2))
            rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())      # noqa: E731
            report.append((pre, err_dz, rel(self.G[pre + '.recurrent_kernel'], gr), rel(self.G[pre + '.kernel'], gk),
                           rel(self.G[pre + '.bias'], gb), rel(dx, gx) if dx is not None else 0.0))
            return dx

        eng_mod.Engine._lstm_backward = patched
        try:
            lg = e.forward(torch.from_numpy(to_tb(x)).to(dev), T, B, True)
            g = torch.from_numpy(to_tb(gt[..., None])).to(dev).view(-1)
            sums, _ = ops.wce_forward(lg.view(-1, 3), g, cwt, False)
            e.backward(ops.wce_backward(lg.view(-1, 3), g, cwt, sums, 1.0).view(lg.shape))
            torch.cuda.synchronize()
        finally:
            eng_mod.Engine._lstm_backward = orig
        assert len(report) == 4
        print('lstm bptt consistency (%s): %s' % (case, report))
        for row in report:
            assert max(row[1:]) <= 1e-5, row


def test_block_and_layer_views_share_the_model_parameters(dev):
    """model.DownLayers[i] / model.UpLayers[i] are callable DownBlock2D / UpBlock2D views over the model's own flat
    parameter buffer and recurrent state (reference Networks.py:195-205 holds the very layer objects the model calls), their
    .ConvLSTM / .Conv / .BN / .LReLU entries are callable layer views; trainable_variables lists every tensor by name;
    WeightedCELoss accepts host logits."""
    import Networks
    import losses
    net = tiny_net(3)
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 2, 16, 16, 1)).astype(np.float32)
    m = Networks.ULSTMnet2D(net, 'NHWC', False, seed=6)
    with pytest.raises(RuntimeError):
        m.DownLayers[1](x, False)                                   # the shared variables do not exist yet
    logits, _ = m(x, False)
    assert isinstance(m.DownLayers[0], Networks.DownBlock2D) and isinstance(m.UpLayers[0], Networks.UpBlock2D)
    names = [v.name for v in m.trainable_variables]
    assert len(names) == len(m.engine.P) and 'down.0.lstm.0.recurrent_kernel' in names and 'up.3.conv.2.bias' in names
    assert len(m.variables) == len(names) + len(m.engine.S)
    # the blocks, chained by hand on a twin model, reproduce the model bit for bit (same kernels, same weights, same state)
    m2 = Networks.ULSTMnet2D(net, 'NHWC', False, seed=6)
    m2.engine.build(1, dev)
    skips, seq = [], torch.from_numpy(x).to(dev)
    flat = seq.reshape((4,) + tuple(seq.shape[2:]))
    for blk in m2.DownLayers:
        skips.append(flat)
        seq, flat = blk(seq, False)
    up = flat
    for blk, skip in zip(m2.UpLayers, skips[::-1]):
        up = blk((up, skip), False)
    got = up.reshape((2, 2) + tuple(up.shape[1:]))
    assert torch.equal(got, logits)
    assert m2.DownLayers[2].get_states()[0][0].shape == m2.get_states()[2][0][0].shape
    # layer views alias the flat buffer
    w = m.DownLayers[0].Conv[0].weights[0]
    assert w.data_ptr() == m.engine.P['down.0.conv.0.kernel'].data_ptr()
    assert [tuple(t.shape) for t in m.DownLayers[0].ConvLSTM[0].weights] == [(3, 3, 1, 32), (3, 3, 8, 32), (32,)]
    assert len(m.UpLayers[3].BN[2].weights) == 0                     # constructed, never called: owns no variables
    y = m.DownLayers[0].Conv[1](torch.from_numpy(rng.standard_normal((1, 8, 8, 8)).astype(np.float32)).to(dev))
    assert tuple(y.shape) == (1, 8, 8, 8)
    z = m.DownLayers[0].LReLU[0](torch.tensor([[[[-1.0, 2.0]]]]))
    assert np.allclose(z.cpu().numpy().ravel(), [-0.3, 2.0])
    # get_weights / set_weights round trip
    ws = m.get_weights()
    m2.set_weights(ws)
    assert torch.equal(m2.engine.flat_params, m.engine.flat_params)
    # host logits are accepted by the loss (moved to the device; the arithmetic stays in the HIP kernels)
    gt = rng.integers(-1, 3, size=(2, 2, 16, 16, 1)).astype(np.float32)
    ce = losses.WeightedCELoss(4, [0.15, 0.25, 0.6])
    host = logits.detach().cpu().numpy()
    assert abs(float(ce(gt, host)) - float(ce(gt, logits))) <= 1e-7


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_training_windows_carry_state_lazily_without_changing_a_bit(dev, precision):
    """A training window leaves the state as views of its tape, reset_states_per_batch only records its mask, and the next
    training window applies both in one pass (ops.state_begin); a reader in between (`states`, get_states, inference) makes
    the state own its tensors and applies the mask in place.  Both routes must give the same bits, the derived weight images
    refreshed in a batch (lu_native/wbank.py) the same bits as freshly built ones."""
    from lu_native.engine import Adam, Engine
    name, net, cin, B, T, H, W, pad = CASES[3]
    if precision == 'bf16':      # wide enough for the bf16 kernels: packed forward kernels and packed flips in the bank
        net, cin, B, T, H, W, pad = ACT16_NET, 1, 2, 2, 16, 32, False
    rng = np.random.default_rng(5)
    xs = [rng.standard_normal((B, T, H, W, cin)).astype(np.float32) for _ in range(3)]
    keeps = [np.array([1.0, 0.0][:B], np.float32), np.array([0.0, 1.0][:B], np.float32)]
    p = perturbed_params(net, cin, 4)
    dl = rng.standard_normal((T * B, H, W, 3)).astype(np.float32) * 1e-2
    outs = []
    for lazy in (True, False):
        e = Engine(net, pad_image=pad, precision=precision)
        e.build(cin, dev)
        e.load_params(p)
        opt = Adam(e, lr=1e-3)
        got = []
        n_win = 2 if (precision == 'bf16' and dev.type != 'cuda') else 3      # (emulator: keep the CPU suite short)
        for w in range(n_win):
            lg = e.forward(torch.from_numpy(to_tb(xs[w])).to(dev), T, B, True)
            e.backward(torch.from_numpy(dl).to(dev))
            opt.apply_gradients()
            if not lazy:
                e.bank = type(e.bank)()                        # fresh weight images every step instead of the batch refresh
                assert len(e._alias) and e.states is not None and not e._alias      # reading `states` made them own tensors
            if w < n_win - 1:
                e.reset_states_per_batch(keeps[w])
                if lazy:
                    assert e._keep is not None and e._alias    # nothing applied yet
                else:
                    e.get_states()
                    assert e._keep is None
            got.append(lg.cpu().numpy())
        if lazy:
            assert e.bank.refreshes >= n_win - 1 and len(e.bank._flips) and (precision == 'fp32' or len(e.bank._packs))
        got.append(e.flat_params.cpu().numpy())
        st = e.get_states()
        got += [t for blk in st for l in blk for t in l]
        outs.append(got)
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


# precision 'bf16x3': two stacked ConvLSTMs at W = 32 -- the first reads the thin image (fp32 kernel gradient, zero-padded
# split blocks), the second a 64-channel input (split kernel gradient, split input gradient); level 1 (F = 8) stays fp32
# ... and two wide Conv2D units on split operands: down.0.conv.1 (32 -> 96, stride 1, behind the stride-2 fp32 layer) and up.0.conv.0 (two
# sources: 32 + 96 channels -> 96)
X3_NET = {'down_conv_kernels': [[(3, 32), (3, 96)], [(3, 32)]], 'lstm_kernels': [[(5, 64), (3, 64)], [(3, 8)]],
          'up_conv_kernels': [[(3, 96)], [(3, 16), (1, 3)]]}


@pytest.mark.parametrize('W', [32, 24])
def test_bf16x3_precision_mode_is_fp32_arithmetic(dev, monkeypatch, W):
    """Engine(precision='bf16x3'): the ConvLSTM convolutions run the bf16-MFMA kernels on the exact three-way bf16 split of
    their fp32 operands (six bf16 products per fp32 product, fp32 accumulation: 2^-26 per product).  The claim is fp32
    ARITHMETIC, so the test is the fp32 engine's own: logits and every gradient tensor are compared with the fp64 oracle and
    must sit where the fp32 engine sits (<= 2x its error + 1e-6), and the two engines agree to 2e-5 / 2e-4 -- two hundred
    times closer than the bf16 mode's contract.  W = 24: a width outside the bf16 kernel-row weight gradient (W % 32 != 0, config-4's
    coarse levels) -- the convolutions run split as they are, the weight gradients on zero-padded copies of the split tensors."""
    from lu_native import calls, ops
    from lu_native.engine import Engine
    if W == 24 and dev.type != 'cuda':
        pytest.skip('the W % 32 != 0 case runs on the MI355X only (-m gpu): another minute of host emulation for the same host code')
    seen = []
    real = calls.conv2d
    monkeypatch.setattr(calls, 'conv2d', lambda *a, **k: (seen.append(k.get('precision', 0)), real(*a, **k))[1])
    net, cin, B, T, H = X3_NET, 1, 1, 2, 4      # (sized for the host emulator: the split layers run 6x the channels)
    n_win = 2 if W == 32 else 1
    rng = np.random.default_rng(33)
    p = perturbed_params(net, cin, 6)
    x = rng.standard_normal((B, T, H, W, cin)).astype(np.float32)
    gt = rng.integers(-1, 3, size=(B, T, H, W)).astype(np.float32)
    cwt = torch.tensor([0.15, 0.25, 0.6], dtype=torch.float32, device=dev)
    res = {}
    split = 'bf16x3' if W == 32 else 'bf16x3-lean'      # (W = 24: the step-by-step route alone -- config-4's combination, and a third less emulator time)
    for prec in (('fp32', 'bf16x3', 'bf16x3-lean') if W == 32 else ('fp32', 'bf16x3-lean')):
        del seen[:]
        e = Engine(net, pad_image=False, precision=prec.split('-')[0])
        if prec.endswith('lean'):
            e.x3_lean_bytes = 0.0      # every split layer forms its gradients step by step (the config-4 route)
        e.build(cin, dev)
        e.load_params(p)
        outs = []
        for win in range(n_win):      # second window: carried state through state_begin + its split copy
            lg = e.forward(torch.from_numpy(to_tb(x)).to(dev), T, B, True)
            g = torch.from_numpy(to_tb(gt[..., None])).to(dev).view(-1)
            sums, _ = ops.wce_forward(lg.view(-1, 3), g, cwt, False)
            e.backward(ops.wce_backward(lg.view(-1, 3), g, cwt, sums, 1.0).view(lg.shape))
            outs.append((lg.cpu().numpy().astype(np.float64), {k: v.cpu().numpy().astype(np.float64) for k, v in e.G.items()}))
            e.reset_states_per_batch(np.ones(B, np.float32))
        inf = e.forward(torch.from_numpy(to_tb(x)).to(dev), T, B, False).cpu().numpy().astype(np.float64)      # inference route
        res[prec] = (outs, inf)
        n_bf16 = sum(seen)
        # fused steps 2 layers x T x windows (+ T inference) + recurrent gradients 2 x (T - 1) x windows + one input gradient x windows
        assert (n_bf16 == 0) if prec == 'fp32' else (n_bf16 >= 2 * T * (n_win + 1) + (2 * (T - 1) + 1) * n_win), (prec, n_bf16)
    for win in range(n_win if W == 32 else 0):      # the step-by-step route: the same products, weight gradients summed over t in dw instead of inside the slabs
        (la, ga), (lb, gb) = res['bf16x3'][0][win], res['bf16x3-lean'][0][win]
        assert np.array_equal(la, lb)
        fl = grad_floor(ga)
        worst = max((float(np.abs(gb[k] - ga[k]).max() / max(np.abs(ga[k]).max(), fl)), k) for k in ga)
        assert worst[0] <= 2e-5, (win, worst)
    for win in range(n_win):
        (l32, g32), (l3, g3) = res['fp32'][0][win], res[split][0][win]
        assert np.abs(l3 - l32).max() <= 2e-5 * np.abs(l32).max(), (win, np.abs(l3 - l32).max() / np.abs(l32).max())
        fl = grad_floor(g32)
        # (a conv bias in front of a BatchNorm has an exactly-zero true gradient: what either engine holds there is its own
        # rounding noise -- 2e-6 of the largest gradient here -- and is left to the oracle comparison below, which has a floor)
        def noise_only(k):
            return '.conv.' in k and k.endswith('.bias') and k.replace('.conv.', '.bn.').replace('.bias', '.gamma') in g32
        worst = max((float(np.linalg.norm(g3[k] - g32[k]) / max(np.linalg.norm(g32[k]), fl)), k) for k in g32 if not noise_only(k))
        assert worst[0] <= 2e-4, (win, worst)
    assert np.abs(res[split][1] - res['fp32'][1]).max() <= 2e-5 * np.abs(res['fp32'][1]).max()
    # against the fp64 oracle (first window: zero state): both engines at fp32 rounding distance
    o = npo.model_forward(net, p, x, training=True, pad_image=False)['logits']
    e32 = np.abs(from_tb(res['fp32'][0][0][0], B, T) - o).max()
    e3 = np.abs(from_tb(res[split][0][0][0], B, T) - o).max()
    assert e3 <= 2.0 * e32 + 1e-6 * np.abs(o).max(), (e3, e32)
    # ... and every gradient tensor of the first window against fp64 autograd (oracle/torch_oracle.py): the split engine's worst
    # tensor within 2x the fp32 engine's worst (+ 1e-5 of the tensor scale)
    tm = tho.TorchULSTM(net, cin, p, dtype=torch.float64)
    _, _, gref = tm.train_step(x, gt, [0.15, 0.25, 0.6], lr=1e-3, apply=False)
    gref = {k: v.numpy() for k, v in gref.items()}
    fl = grad_floor(gref)
    w32 = max(rel_err(res['fp32'][0][0][1][k], gref[k], fl) for k in gref)
    w3 = max((rel_err(res[split][0][0][1][k], gref[k], fl), k) for k in gref)
    print('bf16x3 vs fp64 oracle: logits %.3e (fp32 engine %.3e), worst gradient tensor %.3e %s (fp32 engine %.3e)' %
          (e3 / np.abs(o).max(), e32 / np.abs(o).max(), w3[0], w3[1], w32))
    assert w3[0] <= 2.0 * w32 + 1e-5, (w3, w32)


def test_bf16x3_streaming_inference_on_ragged_frames(dev):
    """precision 'bf16x3', the Inference2D.py:45-62 call pattern (B = 1, T = 1, pad_image: 13 x 21 frames reflect-padded to 32 x 40 inside (emulator: 5 x 9 -> 24 x 32) --
    no weight gradient, so any width takes the split route): three frames with carried state within 2e-5 * max|logit| of the fp32
    engine, labels identical outside a 1e-4 tie band."""
    from lu_native import calls
    from lu_native.engine import Engine
    net, cin = X3_NET, 1
    rng = np.random.default_rng(9)
    p = perturbed_params(net, cin, 5)
    fh, fw = (13, 21) if dev.type == 'cuda' else (5, 9)      # (the emulator gets 24 x 32 padded frames)
    frames = [rng.standard_normal((1, fh, fw, cin)).astype(np.float32) for _ in range(3 if dev.type == 'cuda' else 2)]
    outs, seen = {}, []
    real = calls.conv2d
    calls.conv2d = lambda *a, **k: (seen.append(k.get('precision', 0)), real(*a, **k))[1]
    try:
        for prec in ('fp32', 'bf16x3'):
            del seen[:]
            e = Engine(net, pad_image=True, precision=prec)
            e.build(cin, dev)
            e.load_params(p)
            outs[prec] = [e.forward(torch.from_numpy(f).to(dev), 1, 1, False).cpu().numpy().astype(np.float64) for f in frames]
            assert (sum(seen) == 0) if prec == 'fp32' else (sum(seen) == 4 * len(frames)), (prec, sum(seen))      # two split ConvLSTM layers + two split Conv2D units per frame
    finally:
        calls.conv2d = real
    for a, b in zip(outs['fp32'], outs['bf16x3']):
        assert a.shape == (1, fh, fw, 3)
        assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max(), np.abs(a - b).max() / np.abs(a).max()
        top2 = np.sort(a, -1)
        band = (top2[..., -1] - top2[..., -2]) < 1e-4 * np.abs(a).max()
        assert np.all((a.argmax(-1) == b.argmax(-1)) | band)
