"""MI355X-native ConvLSTM-UNet model builders with the public surface of the reference's
Networks.py (arbellea/LSTM-UNet): DEFAULT_NET_DOWN_PARAMS (Networks.py:12-32), DownBlock2D
(:35-119), UpBlock2D (:122-175), ULSTMnet2D (:178-291).

Same constructor signatures, call conventions (inputs [B,T,C,H,W] for 'NCHW', [B,T,H,W,C]
otherwise; data_format decided by data_format[1] == 'C'), return tuples, ValueError texts and
stateful-recurrence API (reset_states_per_batch / get_states / set_states).  The arithmetic runs in
hand-written gfx950 kernels through lu_native (no TensorFlow, no MIOpen, no CPU fallback).

Documented deviation: the reference's Softmax axis is wrong for NHWC
(`Softmax(self.channel_axis + 1)` = batch axis); here softmax is always over the class axis.
"""
from typing import List

import numpy as np
import torch

from lu_native import ops
from lu_native import plan as plan_mod
from lu_native.engine import Engine

__all__ = ['DEFAULT_NET_DOWN_PARAMS', 'DownBlock2D', 'UpBlock2D', 'ULSTMnet2D']

_W = (128, 256, 256, 512)
DEFAULT_NET_DOWN_PARAMS = {
    'down_conv_kernels': [[(5, w), (5, w)] for w in _W],
    'lstm_kernels': [[(5, w)] for w in _W],
    'up_conv_kernels': [[(5, 256), (5, 256)], [(5, 128), (5, 128)], [(5, 64), (5, 64)],
                        [(5, 32), (5, 32), (1, 3)]],
}


def _is_nchw(data_format):
    return data_format[1] == 'C'


def _device():
    if not torch.cuda.is_available():
        raise ops.NativeError('no HIP device visible: this framework only runs on MI355X-class GPUs '
                              '(there is no CPU execution path)')
    return torch.device('cuda', torch.cuda.current_device())


def _to_internal(x, nchw, device):
    """public [B,T,C,H,W] / [B,T,H,W,C] -> time-major channels-last frames [T*B,H,W,C]."""
    x = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
    x = x.to(device=device, dtype=torch.float32)
    if x.dim() != 5:
        raise ValueError('expected a 5-D [batch, time, ...] tensor, got shape %s' % (tuple(x.shape),))
    x = x.permute(1, 0, 3, 4, 2) if nchw else x.permute(1, 0, 2, 3, 4)
    T, B = x.shape[0], x.shape[1]
    return x.contiguous().view(T * B, x.shape[2], x.shape[3], x.shape[4]), T, B


def _from_internal(y, T, B, nchw):
    """[T*B,H,W,C] -> public [B,T,C,H,W] / [B,T,H,W,C]."""
    y = y.view(T, B, y.shape[1], y.shape[2], y.shape[3])
    y = y.permute(1, 0, 4, 2, 3) if nchw else y.permute(1, 0, 2, 3, 4)
    return y.contiguous()


def _flat_from_internal(y, T, B, nchw):
    """[T*B,H,W,C] -> the reference's 4-D [B*T, ...] skip layout."""
    y = _from_internal(y, T, B, nchw)
    return y.view((B * T,) + tuple(y.shape[2:]))


class _Descr(object):
    """Stands in for the Keras layer objects the reference keeps in its .ConvLSTM/.Conv/.BN/.LReLU lists."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    def __repr__(self):
        return '%s(%s)' % (self.kind, ', '.join('%s=%r' % kv for kv in self.__dict__.items() if kv[0] != 'kind'))


class DownBlock2D(object):
    """N x ConvLSTM2D (stateful, return_sequences) -> M x (Conv2D -> BN -> LeakyReLU), first conv strided."""

    def __init__(self, conv_kernels: List[tuple], lstm_kernels: List[tuple], stride=2, data_format='NCHW'):
        self.data_format = data_format
        self._nchw = _is_nchw(data_format)
        self.ConvLSTM = [_Descr('ConvLSTM2D', kernel_size=k, filters=f) for k, f in lstm_kernels]
        self.Conv, self.BN, self.LReLU = [], [], []
        self.total_stride = 1
        for l_ind, (kxy, kout) in enumerate(conv_kernels):
            _stride = stride if l_ind == 0 else 1
            self.total_stride *= _stride
            self.Conv.append(_Descr('Conv2D', kernel_size=kxy, filters=kout, strides=_stride))
            self.BN.append(_Descr('BatchNormalization', momentum=0.99, epsilon=1e-3))
            self.LReLU.append(_Descr('LeakyReLU', alpha=0.3))

        def plan_fn(cin):
            blk, c = plan_mod.down_block(conv_kernels, lstm_kernels, stride, cin)
            return {'down': [blk], 'up': [], 'total_stride': self.total_stride, 'in_channels': cin, 'last_depth': c}

        self._engine = Engine(None, pad_image=False, plan_fn=plan_fn)

    def call(self, inputs, training=None, mask=None):
        x, T, B = _to_internal(inputs, self._nchw, _device())
        e = self._engine
        e.build(x.shape[-1], x.device)
        blk = e.plan['down'][0]
        seq = x
        for li, l in enumerate(blk['lstm']):
            seq = e._lstm_forward(0, li, l, seq, T, B, None)
        for ci, l in enumerate(blk['conv']):
            seq = e._conv_unit('down.0', ci, l, [(seq, 0, l['cin'])], True, bool(training), [] if training else None)
        return _from_internal(seq, T, B, self._nchw), _flat_from_internal(seq, T, B, self._nchw)

    __call__ = call

    def reset_states_per_batch(self, is_last_batch):
        self._engine.reset_states_per_batch(is_last_batch)

    def get_states(self):
        st = self._engine.get_states()
        return [[None, None] for _ in self.ConvLSTM] if st is None else st[0]

    def set_states(self, states):
        self._engine.set_states([states])

    @classmethod
    def unit_test(cls):
        model = cls([(3, 16), (3, 32), (3, 64)], [(3, 16), (3, 32), (3, 64)], 2, 'NHWC')
        for i in range(4):
            out = model(np.random.randn(2, 3, 50, 50, 3).astype(np.float32), True)
            print(i, tuple(out[0].shape), tuple(out[1].shape))


class UpBlock2D(object):
    """bilinear resize x up_factor -> concat [x, skip] on channels -> M x (Conv2D -> BN -> LeakyReLU)."""

    def __init__(self, kernels: List[tuple], up_factor=2, data_format='NCHW', return_logits=False):
        if up_factor not in (1, 2):
            raise ValueError('up_factor must be 1 or 2 (got %r)' % (up_factor,))
        self.data_format = data_format
        self._nchw = _is_nchw(data_format)
        self.up_factor = up_factor
        self.channel_axis = 1 if self._nchw else -1
        self.return_logits = return_logits
        self.Conv = [_Descr('Conv2D', kernel_size=k, filters=f, strides=1) for k, f in kernels]
        self.BN = [_Descr('BatchNormalization', momentum=0.99, epsilon=1e-3) for _ in kernels]
        self.LReLU = [_Descr('LeakyReLU', alpha=0.3) for _ in kernels]
        self._kernels = list(kernels)
        self._engine = None

    def call(self, inputs, training=None, mask=None):
        input_sequence, skip = inputs
        dev = _device()

        def to4(x):
            x = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(device=dev, dtype=torch.float32)
            return (x.permute(0, 2, 3, 1) if self._nchw else x).contiguous()

        x, s = to4(input_sequence), to4(skip)
        if self._engine is None:
            c_up, c_skip = x.shape[-1], s.shape[-1]

            def plan_fn(cin):
                blk, c = plan_mod.up_block(self._kernels, self.up_factor, self.return_logits, c_up, c_skip)
                return {'down': [], 'up': [blk], 'total_stride': 1, 'in_channels': cin, 'last_depth': c}

            self._engine = Engine(None, pad_image=False, plan_fn=plan_fn)
            self._engine.build(c_skip, dev)
        e = self._engine
        blk = e.plan['up'][0]
        u = ops.upsample2x(x) if self.up_factor == 2 else x
        n = len(blk['conv'])
        a = None
        for ci, l in enumerate(blk['conv']):
            last = self.return_logits and ci == n - 1
            srcs = [(u, 0, blk['c_up']), (s, blk['c_up'], blk['c_skip'])] if ci == 0 else [(a, 0, l['cin'])]
            a = e._conv_unit('up.0', ci, l, srcs, not last, bool(training), [] if training else None)
        return (a.permute(0, 3, 1, 2) if self._nchw else a).contiguous()

    __call__ = call

    @classmethod
    def unit_test(cls):
        model = cls([(3, 16), (3, 32), (3, 64)], 2, 'NHWC')
        for i in range(4):
            out = model((np.random.randn(6, 50, 50, 3).astype(np.float32),
                         np.random.randn(6, 100, 100, 3).astype(np.float32)), True)
            print(i, tuple(out.shape))


class _ModelFn(torch.autograd.Function):
    """Autograd bridge: lets `loss.backward()` drive the engine's hand-written backward."""

    @staticmethod
    def forward(ctx, flat_params, model, x_tb, T, B):
        ctx.model = model
        logits = model._engine.forward(x_tb, T, B, True)
        ctx.meta = (T, B)
        return _from_internal(logits, T, B, model._nchw)

    @staticmethod
    def backward(ctx, dlogits_pub):
        model = ctx.model
        T, B = ctx.meta
        d = dlogits_pub.permute(1, 0, 3, 4, 2) if model._nchw else dlogits_pub.permute(1, 0, 2, 3, 4)
        d = d.contiguous().view(T * B, d.shape[2], d.shape[3], d.shape[4])
        model._engine.backward(d)
        return model._engine.flat_grads, None, None, None, None


class ULSTMnet2D(object):
    """ConvLSTM encoder / conv decoder U-Net (reference Networks.py:178-291)."""

    def __init__(self, net_params=DEFAULT_NET_DOWN_PARAMS, data_format='NCHW', pad_image=True, seed=0, dp=None,
                 sync_bn=False, precision='fp32'):
        self.data_format = data_format
        self._nchw = _is_nchw(data_format)
        self.data_format_keras = 'channels_first' if self._nchw else 'channels_last'
        self.channel_axis = 1 if self._nchw else -1
        self.pad_image = pad_image
        self.net_params = net_params
        self.DownLayers, self.UpLayers = [], []
        self.total_stride = 1
        if not len(net_params['down_conv_kernels']) == len(net_params['lstm_kernels']):
            raise ValueError('Number of layers in down path ({}) do not match number of LSTM layers ({})'.format(
                len(net_params['down_conv_kernels']), len(net_params['lstm_kernels'])))
        if not len(net_params['down_conv_kernels']) == len(net_params['up_conv_kernels']):
            raise ValueError('Number of layers in down path ({}) do not match number of layers in up path ({})'.format(
                len(net_params['down_conv_kernels']), len(net_params['up_conv_kernels'])))
        n = len(net_params['down_conv_kernels'])
        for i, (cf, lf) in enumerate(zip(net_params['down_conv_kernels'], net_params['lstm_kernels'])):
            stride = 2 if i < n - 1 else 1
            blk = _Descr('DownBlock2D', conv_kernels=cf, lstm_kernels=lf, stride=stride, total_stride=stride)
            self.DownLayers.append(blk)
            self.total_stride *= stride
        for i, cf in enumerate(net_params['up_conv_kernels']):
            self.UpLayers.append(_Descr('UpBlock2D', kernels=cf, up_factor=2 if i > 0 else 1,
                                        return_logits=i + 1 == n))
            self.last_depth = cf[-1][1]
        self._engine = Engine(net_params, pad_image=bool(pad_image), seed=seed, dp=dp, sync_bn=sync_bn,
                              precision=precision)
        self._flat_param = None

    # -- torch-style parameter access (one flat leaf: all weights live in one HBM buffer) --
    def parameters(self):
        if self._engine.plan is None:
            raise RuntimeError('parameters are created at the first call (Keras-style lazy build)')
        if self._flat_param is None:
            self._flat_param = self._engine.flat_params.requires_grad_(True)
        return [self._flat_param]

    @property
    def engine(self):
        return self._engine

    def call(self, inputs, training=None, mask=None):
        x_tb, T, B = _to_internal(inputs, self._nchw, _device())
        e = self._engine
        if training and torch.is_grad_enabled():
            e.build(x_tb.shape[-1], x_tb.device)
            logits = _ModelFn.apply(self.parameters()[0], self, x_tb, T, B)
            sm_src = logits.detach()
        else:
            with torch.no_grad():
                logits = _from_internal(e.forward(x_tb, T, B, bool(training)), T, B, self._nchw)
            if training:
                e.tape = None
            sm_src = logits
        # softmax over the class axis
        cl = sm_src.permute(0, 1, 3, 4, 2).contiguous() if self._nchw else sm_src.contiguous()
        if cl.shape[-1] != 3:
            raise NotImplementedError('the softmax / loss kernels are written for the 3-class (bg, cell, edge) head')
        sm = ops.softmax3(cl)
        sm = sm.permute(0, 1, 4, 2, 3).contiguous() if self._nchw else sm
        return logits, sm

    __call__ = call

    def reset_states_per_batch(self, is_last_batch):
        self._engine.reset_states_per_batch(is_last_batch)

    def get_states(self):
        st = self._engine.get_states()
        if st is None:
            return [[[None, None] for _ in lf] for lf in self.net_params['lstm_kernels']]
        return st

    def set_states(self, states):
        self._engine.set_states(states)

    # weights I/O (own format; TF tensor-bundle reader is SURVEY §8f-3, out of scope this round)
    def save_weights(self, path):
        torch.save({k: torch.from_numpy(v) for k, v in self._engine.export_params().items()}, path)

    def load_weights(self, path, in_channels=1):
        blob = torch.load(path, map_location='cpu')
        self._engine.build(in_channels, _device())
        self._engine.load_params({k: v.numpy() for k, v in blob.items()})

    @classmethod
    def unit_test(cls):
        model = cls(DEFAULT_NET_DOWN_PARAMS, 'NHWC', True)
        for i in range(4):
            out = model(np.random.randn(2, 2, 35, 35, 3).astype(np.float32), True)
            print(i, tuple(out[0].shape))


if __name__ == '__main__':
    ULSTMnet2D.unit_test()
