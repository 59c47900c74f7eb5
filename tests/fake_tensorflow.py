"""TEST INFRASTRUCTURE: a minimal stand-in for the `tensorflow` module, just large enough to execute tools/tf_pin.py.

It is NOT TensorFlow and pins nothing: every layer returns the ORACLE's own output (oracle/np_oracle.py), so fixtures written
through it are the oracle talking to itself.  Its only purpose is to prove -- in the build container, where TensorFlow cannot
be installed -- that tools/tf_pin.py runs end to end and writes exactly the fixture schema tests/test_tf_pinned.py reads, so
that on the day a TensorFlow 2.x machine exists the pin is one command (tests/test_tf_pin_tool.py).

Tensors wrap torch float64 tensors so that the few `tf.*` ops the script uses for the loss are differentiable
(tf.GradientTape -> torch.autograd)."""
import sys
import types

import numpy as np
import torch

from oracle import np_oracle as npo


class _Shape(list):
    def as_list(self):
        return list(self)

    def __getitem__(self, i):
        r = list.__getitem__(self, i)
        return _Shape(r) if isinstance(i, slice) else r


class Tensor(object):
    def __init__(self, value, name='t'):
        self.t = value if torch.is_tensor(value) else torch.as_tensor(np.asarray(value))
        self.name = name

    def numpy(self):
        return self.t.detach().numpy()

    @property
    def shape(self):
        return _Shape(self.t.shape)

    def _bin(self, other, fn):
        o = other.t if isinstance(other, Tensor) else torch.as_tensor(np.asarray(other))
        return Tensor(fn(self.t, o))

    def __mul__(self, o):
        return self._bin(o, lambda a, b: a * b)

    __rmul__ = __mul__

    def __add__(self, o):
        return self._bin(o, lambda a, b: a + b)

    __radd__ = __add__

    def __truediv__(self, o):
        return self._bin(o, lambda a, b: a / b)


def _np(x):
    return x.numpy() if isinstance(x, Tensor) else np.asarray(x)


class Variable(Tensor):
    def __init__(self, value, name='var'):
        Tensor.__init__(self, torch.tensor(np.asarray(_np(value)), dtype=torch.float64, requires_grad=True), name)
        self.dtype = np.asarray(_np(value)).dtype

    def numpy(self):
        return self.t.detach().numpy().astype(self.dtype)

    def assign(self, value):
        with torch.no_grad():
            self.t.copy_(torch.as_tensor(np.asarray(value), dtype=torch.float64))


class GradientTape(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def gradient(self, loss, var):
        return Tensor(torch.autograd.grad(loss.t, var.t)[0].to(torch.float32))


def _weight(a, name):
    return Variable(np.asarray(a, np.float32), name)


class _Layer(object):
    def set_weights(self, ws):
        for v, a in zip(self.weights, ws):
            v.assign(a)


class ConvLSTM2D(_Layer):
    def __init__(self, filters, kernel_size, strides=1, padding='valid', data_format=None, return_sequences=False,
                 stateful=False):
        assert padding == 'same' and return_sequences and stateful and strides == 1
        self.f, self.k = filters, kernel_size
        self.weights, self.states = [], [None, None]

    def _build(self, cin):
        rng = np.random.default_rng(5)
        bias = np.zeros(4 * self.f, np.float32)
        bias[self.f:2 * self.f] = 1.0                     # unit_forget_bias
        self.weights = [_weight(rng.standard_normal((self.k, self.k, cin, 4 * self.f)) * 0.1, 'conv_lst_m2d/kernel:0'),
                        _weight(rng.standard_normal((self.k, self.k, self.f, 4 * self.f)) * 0.1, 'conv_lst_m2d/recurrent_kernel:0'),
                        _weight(bias, 'conv_lst_m2d/bias:0')]

    def reset_states(self):
        self.states = [None, None]

    def __call__(self, x):
        x = _np(x)
        if not self.weights:
            self._build(x.shape[-1])
        w = [v.numpy() for v in self.weights]
        h0 = None if self.states[0] is None else self.states[0].numpy()
        c0 = None if self.states[1] is None else self.states[1].numpy()
        y, h, c = npo.convlstm_seq(x, w[0], w[1], w[2], h0, c0)
        self.states = [Tensor(h.astype(np.float32)), Tensor(c.astype(np.float32))]
        return Tensor(y.astype(np.float32))


class Conv2D(_Layer):
    def __init__(self, filters, kernel_size, strides=1, use_bias=True, data_format=None, padding='valid'):
        assert padding == 'same' and use_bias
        self.f, self.k, self.s = filters, kernel_size, strides
        self.weights = []

    def __call__(self, x):
        x = _np(x)
        if not self.weights:
            rng = np.random.default_rng(6)
            self.weights = [_weight(rng.standard_normal((self.k, self.k, x.shape[-1], self.f)) * 0.1, 'conv2d/kernel:0'),
                            _weight(np.zeros(self.f), 'conv2d/bias:0')]
        return Tensor(npo.conv2d_same(x, self.weights[0].numpy(), self.weights[1].numpy(), self.s).astype(np.float32))


class BatchNormalization(_Layer):
    def __init__(self, axis=-1):
        self.epsilon, self.momentum = 1e-3, 0.99
        self.weights = []

    def _build(self, c):
        self.gamma, self.beta = _weight(np.ones(c), 'bn/gamma:0'), _weight(np.zeros(c), 'bn/beta:0')
        self.moving_mean, self.moving_variance = _weight(np.zeros(c), 'bn/moving_mean:0'), _weight(np.ones(c), 'bn/moving_variance:0')
        self.weights = [self.gamma, self.beta, self.moving_mean, self.moving_variance]

    def __call__(self, x, training=False):
        x = _np(x)
        if not self.weights:
            self._build(x.shape[-1])
        g, b = self.gamma.numpy(), self.beta.numpy()
        if training:
            y, mean, var = npo.batchnorm_train(x, g, b, self.epsilon)
            n = x.size // x.shape[-1]
            mm, mv = npo.batchnorm_moving_update(self.moving_mean.numpy(), self.moving_variance.numpy(), mean, var, n,
                                                 self.momentum)
            self.moving_mean.assign(mm)
            self.moving_variance.assign(mv)
        else:
            y = npo.batchnorm_infer(x, g, b, self.moving_mean.numpy(), self.moving_variance.numpy(), self.epsilon)
        return Tensor(np.asarray(y, np.float32))


class LeakyReLU(object):
    def __call__(self, x):
        return Tensor(npo.leaky_relu(_np(x)).astype(np.float32))


class Adam(object):
    def __init__(self, lr):
        self.lr, self.epsilon, self.step, self.slots = float(lr), 1e-7, 0, {}

    def apply_gradients(self, pairs):
        self.step += 1
        for g, var in pairs:
            m, v = self.slots.get(id(var), (0.0, 0.0))
            p, m, v = npo.adam_step(var.numpy().astype(np.float64), _np(g), m, v, self.step, lr=self.lr, eps=self.epsilon)
            self.slots[id(var)] = (m, v)
            var.assign(p)


_VAR_NAMES = {ConvLSTM2D: ['cell/kernel', 'cell/recurrent_kernel', 'cell/bias'], Conv2D: ['kernel', 'bias'],
              BatchNormalization: ['gamma', 'beta', 'moving_mean', 'moving_variance']}


class Model(object):
    def __call__(self, x, training=None):
        return self.call(x, training)

    def _walk(self, obj, path, out):
        if isinstance(obj, (list, tuple)):
            for i, o in enumerate(obj):
                self._walk(o, path + [str(i)], out)
        elif isinstance(obj, Model):
            for key, val in vars(obj).items():
                self._walk(val, path + [key], out)
        elif type(obj) in _VAR_NAMES:
            for name, var in zip(_VAR_NAMES[type(obj)], obj.weights):
                out['/'.join(path + [name])] = var.numpy()

    def save_weights(self, prefix, save_format='tf'):
        import tf_bundle as tb
        found = {}
        for key, val in vars(self).items():
            self._walk(val, [key], found)
        tb.write_bundle(prefix, {k + tb.SUFFIX: v for k, v in found.items()})


class _Reader(object):
    def __init__(self, prefix):
        import tf_bundle as tb
        self.data = tb.read_bundle(prefix)

    def get_tensor(self, key):
        return self.data[key]


def install():
    """Put the stand-in into sys.modules as `tensorflow` (+ tensorflow.python.keras); returns the names to remove again."""
    tf = types.ModuleType('tensorflow')
    tf.__version__ = '2.0.0-standin'
    tf.float32, tf.int32 = torch.float64, torch.int64
    tf.constant = lambda x: Tensor(np.asarray(x))
    tf.Variable = Variable
    tf.GradientTape = GradientTape
    tf.cast = lambda x, dt: Tensor((x.t if isinstance(x, Tensor) else torch.as_tensor(np.asarray(x))).to(dt))
    tf.greater = lambda x, v: Tensor(torch.as_tensor(_np(x)) > v)
    tf.maximum = lambda x, v: Tensor(torch.clamp(torch.as_tensor(_np(x)), min=v))
    tf.one_hot = lambda idx, depth: Tensor(torch.nn.functional.one_hot(torch.clamp(idx.t, min=0), depth).to(torch.float64) *
                                           (idx.t >= 0).unsqueeze(-1))
    tf.reduce_sum = lambda x, axis=None: Tensor(x.t.sum() if axis is None else x.t.sum(axis))
    tf.reshape = lambda x, shape: Tensor(np.asarray(_np(x)).reshape(shape))
    tf.pad = lambda x, pads, mode: Tensor(np.pad(_np(x), pads, mode='reflect'))
    nn = types.ModuleType('tensorflow.nn')
    nn.sparse_softmax_cross_entropy_with_logits = lambda labels, logits: Tensor(
        torch.logsumexp(logits.t, -1) - torch.gather(logits.t, -1, labels.t.unsqueeze(-1)).squeeze(-1))
    tf.nn = nn
    train = types.ModuleType('tensorflow.train')
    train.load_checkpoint = _Reader
    tf.train = train
    keras = types.ModuleType('tensorflow.python.keras')
    keras.Model = Model
    layers = types.ModuleType('tensorflow.python.keras.layers')
    layers.ConvLSTM2D, layers.Conv2D, layers.BatchNormalization, layers.LeakyReLU = ConvLSTM2D, Conv2D, BatchNormalization, LeakyReLU
    keras.layers = layers
    backend = types.ModuleType('tensorflow.python.keras.backend')
    backend.resize_images = lambda x, fh, fw, fmt, interpolation='nearest': Tensor(
        npo.resize_bilinear(_np(x), fh, 'tf2.0').astype(np.float32))
    keras.backend = backend
    optimizers = types.ModuleType('tensorflow.python.keras.optimizers')
    optimizers.Adam = Adam
    keras.optimizers = optimizers
    python = types.ModuleType('tensorflow.python')
    python.keras = keras
    tf.python = python
    tf.keras = keras
    mods = {'tensorflow': tf, 'tensorflow.python': python, 'tensorflow.python.keras': keras, 'tensorflow.nn': nn,
            'tensorflow.train': train}
    sys.modules.update(mods)
    return list(mods)
