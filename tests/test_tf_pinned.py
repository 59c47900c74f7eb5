"""Oracle vs a REAL TensorFlow, from fixtures written by tools/tf_pin.py on a machine that has TensorFlow 2.x
(tests/golden/tf_*.npz + tf_pin.json).  No TensorFlow exists in the build container or on the GPU box, so the fixtures
are absent there and these tests SKIP with the status line "parity unpinned at the TensorFlow boundary"; once the fixtures
are committed the same tests turn the status into "pinned" (tolerance 1e-5 on fp32 TensorFlow outputs)."""
import json
import os

import numpy as np
import pytest

from oracle import np_oracle as npo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _fixture(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip('parity unpinned at the TensorFlow boundary: %s not generated (run tools/tf_pin.py where TensorFlow '
                    'is installed)' % name)
    return path


def parity_status():
    return 'pinned (TensorFlow %s)' % json.load(open(os.path.join(GOLDEN, 'tf_pin.json')))['tensorflow'] \
        if os.path.exists(os.path.join(GOLDEN, 'tf_pin.json')) else 'parity unpinned at the TensorFlow boundary'


def test_status_line():
    print('TF parity status:', parity_status())
    assert parity_status().startswith(('pinned', 'parity unpinned'))


@pytest.mark.parametrize('name', ['convlstm_k5', 'convlstm_k3'])
def test_convlstm_against_tensorflow(name):
    f = np.load(_fixture('tf_%s.npz' % name))
    y1, h, c = npo.convlstm_seq(f['x1'], f['kernel'], f['recurrent_kernel'], f['bias'])
    assert np.abs(y1 - f['y1']).max() <= 1e-5
    y2, h, c = npo.convlstm_seq(f['x2'], f['kernel'], f['recurrent_kernel'], f['bias'], h, c)      # stateful=True
    assert np.abs(y2 - f['y2']).max() <= 1e-5 and np.abs(h - f['h']).max() <= 1e-5 and np.abs(c - f['c']).max() <= 1e-5


def test_conv2d_same_against_tensorflow():
    f = np.load(_fixture('tf_conv2d.npz'))
    for tag, stride in [('k3s2_even', 2), ('k3s2_odd', 2), ('k5s2_even', 2), ('k3s1', 1), ('k1', 1)]:
        y = npo.conv2d_same(f[tag + '_x'], f[tag + '_kernel'], f[tag + '_bias'], stride)
        assert y.shape == f[tag + '_y'].shape and np.abs(y - f[tag + '_y']).max() <= 1e-5, tag


def test_batchnorm_lrelu_against_tensorflow():
    f = np.load(_fixture('tf_bn_lrelu.npz'))
    eps, mom = float(f['eps']), float(f['momentum'])
    y1, mean, var = npo.batchnorm_train(f['x1'], f['gamma'], f['beta'], eps)
    assert np.abs(y1 - f['y1']).max() <= 1e-5
    n = f['x1'].size // f['x1'].shape[-1]
    mm, mv = npo.batchnorm_moving_update(np.zeros(3), np.ones(3), mean, var, n, mom)
    assert np.abs(mm - f['mm1']).max() <= 1e-6 and np.abs(mv - f['mv1']).max() <= 1e-6
    assert np.abs(npo.leaky_relu(f['lrelu_x']) - f['lrelu_y']).max() <= 1e-6
    yi = npo.batchnorm_infer(f['x2'], f['gamma'], f['beta'], f['mm2'], f['mv2'], eps)
    assert np.abs(yi - f['y_infer']).max() <= 1e-5


def test_resize_and_reflect_pad_against_tensorflow():
    f = np.load(_fixture('tf_resize_pad.npz'))
    info = json.load(open(_fixture('tf_pin.json')))
    conv = info['resize_images_bilinear']
    assert conv in ('tf2.0', 'half_pixel'), 'keras.backend.resize_images matches neither restated convention'
    assert np.abs(npo.resize_bilinear(f['x'], 2, conv) - f['y']).max() <= 1e-5
    assert np.array_equal(npo.reflect_pad_hw(f['pad_x'], (2, 3), (1, 4)).astype(np.float32), f['pad_y'])


def test_loss_and_adam_against_tensorflow():
    f = np.load(_fixture('tf_loss_adam.npz'))
    assert abs(npo.weighted_ce(f['gt'], f['logits'], f['class_weights']) - float(f['loss'])) <= 1e-5
    p, m, v = f['p0'].astype(np.float64), np.zeros((5, 4)), np.zeros((5, 4))
    p, m, v = npo.adam_step(p, f['g1'], m, v, 1, lr=float(f['lr']), eps=float(f['eps']))
    assert np.abs(p - f['p1']).max() <= 1e-6
    p, m, v = npo.adam_step(p, f['g2'], m, v, 2, lr=float(f['lr']), eps=float(f['eps']))
    assert np.abs(p - f['p2']).max() <= 1e-6
