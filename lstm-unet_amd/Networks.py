"""MI355X-native ConvLSTM-UNet model builders with the public surface of the reference's
Networks.py (arbellea/LSTM-UNet): DEFAULT_NET_DOWN_PARAMS (Networks.py:12-32), DownBlock2D
(:35-119), UpBlock2D (:122-175), ULSTMnet2D (:178-291).

Same constructor signatures, call conventions (inputs [B,T,C,H,W] for 'NCHW', [B,T,H,W,C]
otherwise; data_format decided by data_format[1] == 'C'), return tuples, ValueError texts and
stateful-recurrence API (reset_states_per_batch / get_states / set_states).  The arithmetic runs in
hand-written gfx950 kernels through lu_native (no TensorFlow, no MIOpen, no CPU fallback).

Documented deviation: the reference's Softmax axis is wrong for NHWC
(`Softmax(self.channel_axis + 1)` = batch axis); here softmax is always over the class axis.

TensorFlow-version dependence: `k.backend.resize_images(x, f, f, fmt, interpolation='bilinear')` (Networks.py:143) samples
at src = o / f in TF 2.0 / 2.1 (the v1 resize_bilinear op without half-pixel centres -- the release the reference's README
pins, 2.0.0a0, and the one its pretrained models come from) and at half-pixel centres in later releases.  `resize='tf2.0'`
(default) / `resize='half_pixel'` on ULSTMnet2D / UpBlock2D select the convention.
"""
from typing import List

import numpy as np
import torch

from lu_native import ops
from lu_native import plan as plan_mod
from lu_native.engine import Engine

__all__ = ['DEFAULT_NET_DOWN_PARAMS', 'DownBlock2D', 'UpBlock2D', 'ULSTMnet2D', 'Variable']

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
    # (through a flat view: size-1 dimensions keep arbitrary strides under .contiguous(), the kernels want canonical ones)
    return x.contiguous().reshape(-1).view(T * B, x.shape[2], x.shape[3], x.shape[4]), T, B


def _from_internal(y, T, B, nchw):
    """[T*B,H,W,C] -> public [B,T,C,H,W] / [B,T,H,W,C]."""
    y = y.view(T, B, y.shape[1], y.shape[2], y.shape[3])
    y = y.permute(1, 0, 4, 2, 3) if nchw else y.permute(1, 0, 2, 3, 4)
    return y.contiguous()


def _flat_from_internal(y, T, B, nchw):
    """[T*B,H,W,C] -> the reference's 4-D [B*T, ...] skip layout."""
    y = _from_internal(y, T, B, nchw)
    return y.view((B * T,) + tuple(y.shape[2:]))


class Variable(object):
    """One named tensor of the model, tf.Variable-style: .name, .shape, .numpy(), .assign(value); `.tensor` is the device
    view into the engine's flat buffer (no copy -- assigning through it changes the model)."""

    def __init__(self, name, tensor, on_assign=None):
        self.name, self.tensor, self._on_assign = name, tensor, on_assign

    @property
    def shape(self):
        return tuple(self.tensor.shape)

    def numpy(self):
        return self.tensor.detach().cpu().numpy()

    def __array__(self, dtype=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def assign(self, value):
        with torch.no_grad():
            self.tensor.copy_(torch.as_tensor(np.asarray(value), dtype=torch.float32).reshape(self.tensor.shape))
        if self._on_assign is not None:
            self._on_assign()
        return self

    def __repr__(self):
        return '<Variable %s shape=%s>' % (self.name, self.shape)


class _Layer(object):
    """Callable view of one Keras layer object of the reference's .ConvLSTM / .Conv / .BN / .LReLU lists
    (Networks.py:44-58,130-139): it owns no storage -- weights are views into the owning engine's flat parameter buffer,
    so calling a layer, a block or the whole model always uses the same tensors.  Channels-last 4-D / 5-D device tensors
    in and out (the blocks do the NCHW <-> channels-last boundary work)."""

    def __init__(self, kind, owner, prefix=None, index=None, **kw):
        self.kind, self._owner, self._prefix, self._index = kind, owner, prefix, index
        self.__dict__.update(kw)

    def __repr__(self):
        skip = ('kind', '_owner', '_prefix', '_index')
        return '%s(%s)' % (self.kind, ', '.join('%s=%r' % kv for kv in self.__dict__.items() if kv[0] not in skip))

    def _engine(self):
        e = self._owner._engine_built()
        return e, self._owner._prefix_of(self)

    @property
    def weights(self):
        """[kernel, (recurrent_kernel,) bias] / [gamma, beta, moving_mean, moving_variance] as device tensor views."""
        e, pre = self._engine()
        if self.kind == 'ConvLSTM2D':
            names = [f'{pre}.lstm.{self._index}.{n}' for n in ('kernel', 'recurrent_kernel', 'bias')]
        elif self.kind == 'Conv2D':
            names = [f'{pre}.conv.{self._index}.{n}' for n in ('kernel', 'bias')]
        elif self.kind == 'BatchNormalization':
            names = [f'{pre}.bn.{self._index}.{n}' for n in ('gamma', 'beta', 'moving_mean', 'moving_var')]
        else:
            return []
        return [e.P[n] if n in e.P else e.S[n] for n in names if n in e.P or n in e.S]

    def get_weights(self):
        return [w.detach().cpu().numpy() for w in self.weights]

    def __call__(self, x, training=None):
        e, pre = self._engine()
        x = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(device=e.device, dtype=torch.float32)
        if self.kind == 'LeakyReLU':
            Cc = x.shape[-1]
            one, zero = torch.ones(Cc, device=e.device), torch.zeros(Cc, device=e.device)
            return ops.bn_lrelu_apply(x.contiguous(), one, zero, self.alpha)
        if self.kind == 'BatchNormalization':      # alpha = 1: the fused kernel's LeakyReLU becomes the identity
            if f'{pre}.bn.{self._index}.gamma' not in e.P:
                raise RuntimeError('this BatchNormalization layer is constructed but never called by the model '
                                   '(Networks.py:138,148-149): it owns no variables')
            gamma, beta = e.P[f'{pre}.bn.{self._index}.gamma'], e.P[f'{pre}.bn.{self._index}.beta']
            mm, mv = e.S[f'{pre}.bn.{self._index}.moving_mean'], e.S[f'{pre}.bn.{self._index}.moving_var']
            from lu_native.engine import BN_EPS, BN_MOMENTUM
            y = x.contiguous()
            if training:
                sums = ops.bn_stats(y)
                scale, shift, _, _ = ops.bn_finalize_train(sums, y.numel() // y.shape[-1], gamma, beta, BN_EPS, BN_MOMENTUM,
                                                           mm, mv)
            else:
                scale, shift = ops.bn_finalize_infer(gamma, beta, mm, mv, BN_EPS)
            return ops.bn_lrelu_apply(y, scale, shift, 1.0)
        if self.kind == 'Conv2D':
            w, b = e.P[f'{pre}.conv.{self._index}.kernel'], e.P[f'{pre}.conv.{self._index}.bias']
            return ops.conv2d([(x.contiguous(), w)], b, self.strides)
        if self.kind == 'ConvLSTM2D':              # [B,T,H,W,C] channels-last sequence in and out, stateful
            if x.dim() != 5:
                raise ValueError('ConvLSTM2D expects a 5-D [batch, time, H, W, C] tensor')
            B, T = x.shape[0], x.shape[1]
            seq = x.permute(1, 0, 2, 3, 4).contiguous().view((T * B,) + tuple(x.shape[2:]))
            bi = int(pre.split('.')[1])
            spec = e.plan['down'][bi]['lstm'][self._index]
            h = e._lstm_forward(bi, self._index, spec, seq, T, B, None)
            return h.view((T, B) + tuple(h.shape[1:])).permute(1, 0, 2, 3, 4).contiguous()
        raise TypeError(self.kind)


class DownBlock2D(object):
    """N x ConvLSTM2D (stateful, return_sequences) -> M x (Conv2D -> BN -> LeakyReLU), first conv strided."""

    def __init__(self, conv_kernels: List[tuple], lstm_kernels: List[tuple], stride=2, data_format='NCHW', _parent=None):
        """_parent = (engine, block index): this block is model.DownLayers[i] -- a view that runs on the model's own
        parameters and recurrent state (reference Networks.py:195-199 keeps the very layer objects the model calls)."""
        self.data_format = data_format
        self._nchw = _is_nchw(data_format)
        self.ConvLSTM = [_Layer('ConvLSTM2D', self, index=i, kernel_size=k, filters=f) for i, (k, f) in enumerate(lstm_kernels)]
        self.Conv, self.BN, self.LReLU = [], [], []
        self.total_stride = 1
        for l_ind, (kxy, kout) in enumerate(conv_kernels):
            _stride = stride if l_ind == 0 else 1
            self.total_stride *= _stride
            self.Conv.append(_Layer('Conv2D', self, index=l_ind, kernel_size=kxy, filters=kout, strides=_stride))
            self.BN.append(_Layer('BatchNormalization', self, index=l_ind, momentum=0.99, epsilon=1e-3))
            self.LReLU.append(_Layer('LeakyReLU', self, index=l_ind, alpha=0.3))

        def plan_fn(cin):
            blk, c = plan_mod.down_block(conv_kernels, lstm_kernels, stride, cin)
            return {'down': [blk], 'up': [], 'total_stride': self.total_stride, 'in_channels': cin, 'last_depth': c}

        if _parent is None:
            self._engine, self._bi = Engine(None, pad_image=False, plan_fn=plan_fn), 0
        else:
            self._engine, self._bi = _parent

    def _engine_built(self):
        if self._engine.plan is None:
            raise RuntimeError('layer variables are created at the first call (Keras-style lazy build): call the block / '
                               'the model once first')
        return self._engine

    def _prefix_of(self, layer):
        return f'down.{self._bi}'

    def _build_for(self, cin, device):
        e = self._engine
        if e.plan is None:
            if e.net_params is not None and self._bi != 0:
                raise RuntimeError('model.DownLayers[%d] shares the model\'s parameters: call the model (or DownLayers[0]) '
                                   'once so that they exist' % self._bi)
            e.build(cin, device)
        return e

    def call(self, inputs, training=None, mask=None):
        x, T, B = _to_internal(inputs, self._nchw, _device())
        e = self._build_for(x.shape[-1], x.device)
        bi = self._bi
        if e.batch is None:
            e.batch = B
        blk = e.plan['down'][bi]
        seq = x
        for li, l in enumerate(blk['lstm']):
            seq = e._lstm_forward(bi, li, l, seq, T, B, None)
        for ci, l in enumerate(blk['conv']):
            seq = e._conv_unit(f'down.{bi}', ci, l, [(seq, 0, l['cin'])], True, bool(training), [] if training else None)
        return _from_internal(seq, T, B, self._nchw), _flat_from_internal(seq, T, B, self._nchw)

    __call__ = call

    def reset_states_per_batch(self, is_last_batch):
        e = self._engine
        if e.states is None:
            return
        keep = torch.as_tensor(np.asarray(is_last_batch), dtype=torch.float32).reshape(-1).to(e.device)
        e.invalidate_state_copies(self._bi)
        for st in e.states[self._bi]:
            if st is not None:
                ops.scale_frames(st[0], keep)
                ops.scale_frames(st[1], keep)

    def get_states(self):
        st = self._engine.get_states()
        return [[None, None] for _ in self.ConvLSTM] if st is None else st[self._bi]

    def set_states(self, states):
        e = self._engine
        if e.states is None:
            if e.net_params is not None:
                raise RuntimeError('call the model once before setting the states of one of its blocks')
            e.set_states([states])
            return
        full = [[None if s is None else [s[0], s[1]] for s in blk] for blk in e.states]
        full[self._bi] = states
        e.set_states(full)

    @classmethod
    def unit_test(cls):
        model = cls([(3, 16), (3, 32), (3, 64)], [(3, 16), (3, 32), (3, 64)], 2, 'NHWC')
        for i in range(4):
            out = model(np.random.randn(2, 3, 50, 50, 3).astype(np.float32), True)
            print(i, tuple(out[0].shape), tuple(out[1].shape))


class UpBlock2D(object):
    """bilinear resize x up_factor -> concat [x, skip] on channels -> M x (Conv2D -> BN -> LeakyReLU)."""

    def __init__(self, kernels: List[tuple], up_factor=2, data_format='NCHW', return_logits=False, _parent=None,
                 resize='tf2.0'):
        if up_factor not in (1, 2):
            raise ValueError('up_factor must be 1 or 2 (got %r)' % (up_factor,))
        self.data_format = data_format
        self._nchw = _is_nchw(data_format)
        self.up_factor = up_factor
        self.channel_axis = 1 if self._nchw else -1
        self.return_logits = return_logits
        self.Conv = [_Layer('Conv2D', self, index=i, kernel_size=k, filters=f, strides=1) for i, (k, f) in enumerate(kernels)]
        self.BN = [_Layer('BatchNormalization', self, index=i, momentum=0.99, epsilon=1e-3) for i in range(len(kernels))]
        self.LReLU = [_Layer('LeakyReLU', self, index=i, alpha=0.3) for i in range(len(kernels))]
        self._kernels = list(kernels)
        self._resize = resize
        self._engine, self._bi = (None, 0) if _parent is None else _parent
        self._shared = _parent is not None

    def _engine_built(self):
        if self._engine is None or self._engine.plan is None:
            raise RuntimeError('layer variables are created at the first call (Keras-style lazy build): call the block / '
                               'the model once first')
        return self._engine

    def _prefix_of(self, layer):
        return f'up.{self._bi}'

    def call(self, inputs, training=None, mask=None):
        input_sequence, skip = inputs
        dev = _device()

        def to4(x):
            x = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(device=dev, dtype=torch.float32)
            return (x.permute(0, 2, 3, 1) if self._nchw else x).contiguous()

        x, s = to4(input_sequence), to4(skip)
        if self._shared:
            self._engine_built()
        elif self._engine is None:
            c_up, c_skip = x.shape[-1], s.shape[-1]

            def plan_fn(cin):
                blk, c = plan_mod.up_block(self._kernels, self.up_factor, self.return_logits, c_up, c_skip)
                return {'down': [], 'up': [blk], 'total_stride': 1, 'in_channels': cin, 'last_depth': c}

            self._engine = Engine(None, pad_image=False, plan_fn=plan_fn, resize=self._resize)
            self._engine.build(c_skip, dev)
        e = self._engine
        bi = self._bi
        blk = e.plan['up'][bi]
        if (x.shape[-1], s.shape[-1]) != (blk['c_up'], blk['c_skip']):
            raise ValueError('UpBlock2D was built for %d + %d input channels, got %d + %d' %
                             (blk['c_up'], blk['c_skip'], x.shape[-1], s.shape[-1]))
        u = ops.upsample2x(x, e.resize) if self.up_factor == 2 else x
        n = len(blk['conv'])
        a = None
        for ci, l in enumerate(blk['conv']):
            last = self.return_logits and ci == n - 1
            srcs = [(u, 0, blk['c_up']), (s, blk['c_up'], blk['c_skip'])] if ci == 0 else [(a, 0, l['cin'])]
            a = e._conv_unit(f'up.{bi}', ci, l, srcs, not last, bool(training), [] if training else None)
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
                 sync_bn=False, precision='fp32', resize='tf2.0'):
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
        self._engine = Engine(net_params, pad_image=bool(pad_image), seed=seed, dp=dp, sync_bn=sync_bn,
                              precision=precision, resize=resize)
        # DownLayers / UpLayers are callable block views over THIS model's parameters and recurrent state
        # (reference Networks.py:195-205 keeps the layer objects the model itself calls)
        for i, (cf, lf) in enumerate(zip(net_params['down_conv_kernels'], net_params['lstm_kernels'])):
            stride = 2 if i < n - 1 else 1
            self.DownLayers.append(DownBlock2D(cf, lf, stride, data_format, _parent=(self._engine, i)))
            self.total_stride *= stride
        for i, cf in enumerate(net_params['up_conv_kernels']):
            self.UpLayers.append(UpBlock2D(cf, 2 if i > 0 else 1, data_format, return_logits=i + 1 == n,
                                           _parent=(self._engine, i)))
            self.last_depth = cf[-1][1]
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

    @property
    def trainable_variables(self):
        """Per-tensor views of the flat parameter buffer, each carrying a `.name` (k.Model.trainable_variables of the
        reference, train2D.py:92): 12 ConvLSTM + 34 Conv2D + 32 BatchNormalization tensors for the default network."""
        if self._engine.plan is None:
            raise RuntimeError('variables are created at the first call (Keras-style lazy build)')
        e = self._engine
        return [Variable(name, t.detach(), e.weights_changed) for name, t in e.P.items()]

    @property
    def variables(self):
        """trainable_variables + the BatchNormalization moving statistics."""
        e = self._engine
        return self.trainable_variables + [Variable(name, t.detach(), e.weights_changed) for name, t in e.S.items()]

    def get_weights(self):
        return [v.numpy() for v in self.variables]

    def set_weights(self, weights):
        names = [v.name for v in self.variables]
        if len(weights) != len(names):
            raise ValueError('expected %d arrays, got %d' % (len(names), len(weights)))
        self._engine.load_params(dict(zip(names, weights)))

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
        sm = ops.softmax_last(cl)      # (any head depth, Networks.py:205-206; losses.WeightedCELoss is 3-class like the reference's)
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

    # weights I/O: the reference's `save_weights(path, save_format='tf')` / `load_weights(path)` pair (train2D.py:235,
    # Inference2D.py:34).  save_format 'tf' writes / reads a TensorFlow tensor bundle (<path>.index +
    # <path>.data-00000-of-00001, tf_bundle.py: no TensorFlow needed) with the object-graph keys tf.keras gives this model;
    # 'pt' (default) is a torch.save blob under exactly `path`.
    def save_weights(self, path, save_format=None):
        if save_format in ('tf', 'tensorflow'):
            import tf_bundle
            tf_bundle.save_model_weights(self, path)
        elif save_format in (None, 'pt', 'h5'):
            torch.save({k: torch.from_numpy(v) for k, v in self._engine.export_params().items()}, path)
        else:
            raise ValueError('unknown save_format %r' % (save_format,))

    def load_weights(self, path, in_channels=1):
        import os
        self._engine.build(in_channels, _device())
        if os.path.exists(path + '.index'):
            import tf_bundle
            self._engine.load_params(tf_bundle.load_model_weights(self, path))
            return
        blob = torch.load(path, map_location='cpu')
        self._engine.load_params({k: v.numpy() for k, v in blob.items()})

    @classmethod
    def unit_test(cls):
        model = cls(DEFAULT_NET_DOWN_PARAMS, 'NHWC', True)
        for i in range(4):
            out = model(np.random.randn(2, 2, 35, 35, 3).astype(np.float32), True)
            print(i, tuple(out[0].shape))


if __name__ == '__main__':
    ULSTMnet2D.unit_test()
