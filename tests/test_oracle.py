"""CPU tests of the oracle itself: hand-computable KATs, reference-derived goldens, and the
numpy-fp64 vs torch restatement cross-check."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import np_oracle as npo
from oracle import torch_oracle as tho
from conftest import tiny_net


# ---------------- KATs ----------------
def test_same_pad_geometry():
    assert npo.tf_same_pad(256, 3, 2) == (128, 0, 1)      # even in, k3 s2: pad 0/1
    assert npo.tf_same_pad(256, 5, 2) == (128, 1, 2)
    assert npo.tf_same_pad(7, 3, 2) == (4, 1, 1)          # odd in: symmetric
    assert npo.tf_same_pad(50, 5, 1) == (50, 2, 2)
    assert npo.tf_same_pad(4, 1, 1) == (4, 0, 0)


def test_conv_stride2_same_4x4_kat():
    x = np.arange(16, dtype=np.float64).reshape(1, 4, 4, 1)
    w = np.ones((3, 3, 1, 1))
    y = npo.conv2d_same(x, w, None, 2)
    # windows start at rows/cols 0 and 2, zero pad only at bottom/right
    exp = np.array([[x[0, 0:3, 0:3].sum(), x[0, 0:3, 2:4].sum()],
                    [x[0, 2:4, 0:3].sum(), x[0, 2:4, 2:4].sum()]])
    assert np.allclose(y[0, :, :, 0], exp)


def test_conv_matches_naive_loops():
    rng = np.random.default_rng(0)
    for (h, w, c, o, k, s) in [(5, 7, 3, 4, 3, 1), (6, 6, 2, 3, 5, 1), (7, 8, 3, 2, 3, 2), (8, 8, 1, 2, 5, 2),
                               (4, 4, 3, 3, 1, 1)]:
        x = rng.standard_normal((2, h, w, c))
        wt = rng.standard_normal((k, k, c, o))
        b = rng.standard_normal(o)
        assert np.allclose(npo.conv2d_same(x, wt, b, s), npo.conv2d_same_naive(x, wt, b, s), atol=1e-12)


def test_identity_kernel():
    x = np.random.default_rng(1).standard_normal((1, 6, 6, 2))
    w = np.zeros((5, 5, 2, 2))
    w[2, 2] = np.eye(2)
    assert np.allclose(npo.conv2d_same(x, w), x)


def test_hard_sigmoid_breakpoints():
    assert npo.hard_sigmoid(np.array([-2.5, 2.5, 0.0, -3.0, 3.0, 1.0])).tolist() == [0, 1, 0.5, 0, 1, 0.7]


def test_bilinear_2x2_kat():
    x = np.array([[0., 1.], [0., 1.]]).reshape(1, 2, 2, 1)
    y = npo.resize_bilinear(x, 2, 'half_pixel')[0, 0, :, 0]
    assert np.allclose(y, [0, .25, .75, 1])                      # tf.image.resize v2: half-pixel centres
    y = npo.resize_bilinear(x, 2, 'tf2.0')[0, 0, :, 0]
    assert np.allclose(y, [0, .5, 1, 1])                         # legacy v1 op (TF 2.0 / 2.1 Keras): src = o / 2, edge clamp
    assert np.allclose(npo.resize_bilinear(x, 1), x)


def test_bn_constant_channel_and_moving():
    x = np.ones((2, 3, 3, 2))
    x[..., 1] = np.arange(18).reshape(2, 3, 3)
    y, mean, var = npo.batchnorm_train(x, np.ones(2), np.zeros(2))
    assert np.allclose(y[..., 0], 0) and np.allclose(mean, [1, 8.5])
    mm, mv = npo.batchnorm_moving_update(np.zeros(2), np.ones(2), mean, var, 18)
    assert np.allclose(mm, 0.01 * mean) and np.allclose(mv, 0.99 + 0.01 * var * 18 / 17)


def test_ce_uniform_logits():
    gt = np.array([[[[0., 1.], [2., -1.]]]])
    lg = np.zeros((1, 1, 2, 2, 3))
    w = [0.15, 0.25, 0.6]
    assert np.isclose(npo.weighted_ce(gt, lg, w), np.log(3) * sum(w) / (3 + 1e-5))


def test_reflect_pad_and_model_pads():
    a = np.arange(5.).reshape(1, 1, 5, 1)
    p = npo.reflect_pad_hw(a, (0, 0), (2, 1))
    assert p[0, 0, :, 0].tolist() == [2, 1, 0, 1, 2, 3, 4, 3]
    assert npo.model_pads(35, 35, 8, True) == ((8, 13), (8, 13))     # ULSTMnet2D.unit_test: 35 -> 56
    assert npo.model_pads(256, 256, 8, False) == ((0, 0), (0, 0))


def test_adam_tf_form():
    p, m, v = npo.adam_step(np.array([1.0]), np.array([0.5]), np.zeros(1), np.zeros(1), 1, lr=1e-2)
    # first step of Adam moves by ~lr regardless of gradient scale
    assert np.isclose(p[0], 1.0 - 1e-2 * 0.05 * np.sqrt(0.001) / 0.1 / (np.sqrt(0.00025) + 1e-7), rtol=1e-12)


# ---------------- goldens captured from the reference ----------------
def test_seg_unit_fixture(golden_dir):
    d = np.load(os.path.join(golden_dir, 'seg_unit_fixture.npz'))
    assert np.float32(d['seg']) == np.float32(0.59999996)
    got = npo.seg_measure(d['gt'][..., 0], d['logits'])
    assert abs(got - float(d['seg'])) < 1e-6


def test_seg_random_goldens(golden_dir):
    d = np.load(os.path.join(golden_dir, 'seg_random.npz'))
    for gt, lg, exp in zip(d['gt'], d['logits'], d['seg']):
        got = npo.seg_measure(gt, lg)
        if np.isnan(exp):
            assert np.isnan(got)
        else:
            assert abs(got - float(exp)) < 1e-6


def test_edge_rule_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, 'edge_rule.npz'))
    for inst, cls in zip(d['inst'], d['classes']):
        assert np.array_equal(npo.instances_to_classes(inst), cls)


def test_default_params_golden(golden_dir):
    with open(os.path.join(golden_dir, 'default_params.json')) as f:
        d = json.load(f)
    plan = npo.net_plan(d['CTCParams.net_kernel_params'], 1)
    assert plan['total_stride'] == 8 and plan['last_depth'] == 3
    assert [b['conv'][0]['cin'] for b in plan['up']] == [768, 512, 256, 65]   # SURVEY §8a layer table
    p = npo.init_params(d['CTCParams.net_kernel_params'], 1, seed=0)
    n_train = sum(int(np.prod(p[k].shape)) for k in npo.trainable_names(p))
    assert n_train == 74606531


def test_net_plan_validation():
    bad = tiny_net()
    bad['lstm_kernels'] = bad['lstm_kernels'][:-1]
    with pytest.raises(ValueError):
        npo.net_plan(bad, 1)


# ---------------- shape contracts of the reference smokes ----------------
def test_unit_test_shape_contract():
    net = tiny_net(3, (4, 4, 4, 4), (4, 4, 4, 4))
    x = np.random.default_rng(0).standard_normal((2, 2, 35, 35, 3))
    p = npo.init_params(net, 3, seed=1)
    out = npo.model_forward(net, p, x, training=True, pad_image=True)
    assert out['logits'].shape == (2, 2, 35, 35, 3)
    assert out['taps']['down.3.out'].shape == (4, 7, 7, 4)        # 56 / 8


# ---------------- numpy fp64 vs torch restatement ----------------
@pytest.mark.parametrize('k_lstm,pad_image,hw', [(3, False, (16, 24)), (5, True, (13, 18))])
def test_numpy_vs_torch_forward(k_lstm, pad_image, hw):
    net = tiny_net(k_lstm)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 3) + hw + (1,))
    p = npo.init_params(net, 1, seed=2, dtype=np.float64)
    for k in p:  # non-trivial BN affine / moving stats / biases
        if k.endswith(('gamma', 'moving_var')):
            p[k] = p[k] + 0.2 * rng.random(p[k].shape)
        elif k.endswith(('beta', 'bias', 'moving_mean')):
            p[k] = p[k] + 0.1 * rng.standard_normal(p[k].shape)
    for training in (True, False):
        ref = npo.model_forward(net, p, x, training=training, pad_image=pad_image, update_moving=True)
        tm = tho.TorchULSTM(net, 1, p, dtype=torch.float64, pad_image=pad_image)
        lg = tm.forward(torch.tensor(x), training=training).numpy()
        assert np.abs(lg - ref['logits']).max() < 1e-9
        # second window: carried state
        ref2 = npo.model_forward(net, p, x[:, ::-1], states=ref['states'], training=training, pad_image=pad_image)
        lg2 = tm.forward(torch.tensor(x[:, ::-1].copy()), training=training, update_moving=False).numpy()
        assert np.abs(lg2 - ref2['logits']).max() < 1e-9
        if training:
            for k, v in ref['moving'].items():
                assert np.abs(tm.P[k].numpy() - v).max() < 1e-12


def test_torch_gradient_vs_finite_difference():
    """Pins the gradient oracle: autograd of the torch restatement == numerical derivative
    of the numpy fp64 forward + loss."""
    net = tiny_net(3, (3, 3, 4, 4), (4, 3, 3, 3))
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 2, 8, 8, 1))
    gt = rng.integers(-1, 3, size=(1, 2, 8, 8)).astype(np.float64)
    cw = [0.15, 0.25, 0.6]
    p = npo.init_params(net, 1, seed=4, dtype=np.float64)
    tm = tho.TorchULSTM(net, 1, p, dtype=torch.float64)
    loss, _, grads = tm.train_step(x, gt, cw, apply=False)
    l0 = npo.weighted_ce(gt, npo.model_forward(net, p, x, training=True)['logits'], cw)
    assert abs(float(loss) - l0) < 1e-12
    eps = 1e-6
    for name in ['down.0.lstm.0.recurrent_kernel', 'down.1.lstm.0.kernel', 'down.0.lstm.0.bias',
                 'down.2.conv.0.kernel', 'up.1.conv.0.kernel', 'up.3.conv.2.bias', 'down.1.bn.1.gamma']:
        g = grads[name].numpy()
        idxs = [tuple(rng.integers(0, s) for s in p[name].shape) for _ in range(3)]
        for idx in idxs:
            pp = {k: v.copy() for k, v in p.items()}
            pp[name][idx] += eps
            lp = npo.weighted_ce(gt, npo.model_forward(net, pp, x, training=True)['logits'], cw)
            pp[name][idx] -= 2 * eps
            lm = npo.weighted_ce(gt, npo.model_forward(net, pp, x, training=True)['logits'], cw)
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - g[idx]) < 1e-6 * max(1.0, abs(fd)) + 1e-8, (name, idx, fd, g[idx])
