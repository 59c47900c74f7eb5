#!/usr/bin/env python
"""Pin the oracle to a real TensorFlow, wherever one is importable (SURVEY §8c last row, §8f-3b).

TensorFlow is NOT installed in the build container or on the GPU box, so every Keras-layer semantic in
oracle/np_oracle.py is restated from its published behaviour and the TF boundary is "parity unpinned".  This build-owned
script (no reference file is read) closes that gap on any machine that has TensorFlow 2.x:

    python tools/tf_pin.py            # writes tests/golden/tf_*.npz + tests/golden/tf_pin.json

It constructs the tf.keras layers EXACTLY as the reference constructs them --
    ConvLSTM2D(filters, kernel_size, strides=1, padding='same', data_format=fmt, return_sequences=True, stateful=True)
                                                                                          (Networks.py:48-50)
    Conv2D(filters, kernel_size, strides, use_bias=True, data_format=fmt, padding='same')  (Networks.py:55-56,135-136)
    BatchNormalization(axis=channel_axis), LeakyReLU()                                      (Networks.py:57-58,138-139)
    k.backend.resize_images(x, 2, 2, fmt, interpolation='bilinear')                         (Networks.py:143)
    tf.pad(x, pads, 'REFLECT')                                                              (Networks.py:232)
    tf.nn.sparse_softmax_cross_entropy_with_logits + class weights                          (losses.py:13-27)
    tf.keras.optimizers.Adam(lr) one apply_gradients                                        (train2D.py:61,93)
-- feeds seeded inputs, and dumps weights, inputs and outputs.  tests/test_tf_pinned.py consumes the fixtures when they
exist (oracle vs TensorFlow at 1e-5) and reports "parity unpinned" when they do not.  It also
  * records which bilinear convention this TensorFlow's keras.backend.resize_images uses (legacy src = o/2 of TF 2.0 / 2.1
    vs half-pixel centres), the BatchNormalization moving-variance rule and the Adam epsilon placement;
  * cross-checks lstm-unet_amd/tf_bundle.py against a real tensor bundle: a tf.keras model with the reference's attribute
    structure is saved with save_weights(..., save_format='tf') and read back with tf_bundle (names + values), and a
    bundle written by tf_bundle is read with tf.train.load_checkpoint.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def probe_tensorflow():
    """-> (module or None, version string or the text of the exception `import tensorflow` actually raised here)."""
    try:
        import tensorflow as tf
        return tf, getattr(tf, '__version__', '?')
    except Exception as exc:      # ImportError normally; a broken install raises other things
        return None, '%s: %s' % (type(exc).__name__, exc)


def keras_train_step_timer(tf, net, H, W, B, T, class_weights=(0.15, 0.25, 0.6), lr=1e-5):
    """Build-owned tf.keras model (no reference file is read) with the layers constructed as the reference constructs them
    (Networks.py:44-58,130-139,195-205; channels_last: stock CPU builds of TensorFlow run Conv2D in NHWC only), the loss of
    losses.py:13-27 and the step of train2D.py:87-95 -> a callable that runs ONE optimiser step on seeded synthetic data and
    returns its wall time.  Used by bench.py's cpu_baseline when TensorFlow is importable on the GPU box (kind "tf")."""
    import time
    try:
        from tensorflow.python import keras as k
    except ImportError:
        from tensorflow import keras as k
    L = k.layers

    class Down(k.Model):
        def __init__(self, convs, lstms, stride):
            super().__init__()
            self.ConvLSTM = [L.ConvLSTM2D(filters=f, kernel_size=ks, strides=1, padding='same', data_format='channels_last',
                                          return_sequences=True, stateful=True) for ks, f in lstms]
            self.Conv = [L.Conv2D(filters=f, kernel_size=ks, strides=(stride if i == 0 else 1), use_bias=True,
                                  data_format='channels_last', padding='same') for i, (ks, f) in enumerate(convs)]
            self.BN = [L.BatchNormalization(axis=-1) for _ in convs]
            self.LReLU = [L.LeakyReLU() for _ in convs]

        def call(self, x, training=None):
            for layer in self.ConvLSTM:
                x = layer(x)
            bt = x.shape[0] * x.shape[1]
            y = tf.reshape(x, [bt] + x.shape[2:].as_list())
            for c, b, a in zip(self.Conv, self.BN, self.LReLU):
                y = a(b(c(y), training=training))
            return tf.reshape(y, [x.shape[0], x.shape[1]] + y.shape[1:].as_list()), y

    class Up(k.Model):
        def __init__(self, convs, factor, logits):
            super().__init__()
            self.factor, self.logits = factor, logits
            self.Conv = [L.Conv2D(filters=f, kernel_size=ks, strides=1, use_bias=True, data_format='channels_last',
                                  padding='same') for ks, f in convs]
            self.BN = [L.BatchNormalization(axis=-1) for _ in convs]
            self.LReLU = [L.LeakyReLU() for _ in convs]

        def call(self, xs, training=None):
            x, skip = xs
            x = k.backend.resize_images(x, self.factor, self.factor, 'channels_last', interpolation='bilinear')
            y = tf.concat([x, skip], -1)
            for i, (c, b, a) in enumerate(zip(self.Conv, self.BN, self.LReLU)):
                y = c(y)
                if not (self.logits and i == len(self.Conv) - 1):
                    y = a(b(y, training=training))
            return y

    n = len(net['down_conv_kernels'])
    downs = [Down(net['down_conv_kernels'][i], net['lstm_kernels'][i], 2 if i < n - 1 else 1) for i in range(n)]
    ups = [Up(net['up_conv_kernels'][j], 1 if j == 0 else 2, j == n - 1) for j in range(n)]
    opt = k.optimizers.Adam(lr)
    cw = tf.constant(np.asarray(class_weights, np.float32))
    rng = np.random.default_rng(0)
    x = tf.constant(rng.standard_normal((B, T, H, W, 1)).astype(np.float32))
    gt = tf.constant(rng.integers(-1, 3, size=(B, T, H, W)).astype(np.float32))

    def forward(training):
        skips, seq = [], x
        flat = tf.reshape(x, [B * T, H, W, 1])
        for d in downs:
            skips.append(flat)
            seq, flat = d(seq, training=training)
        y = flat
        for u, sk in zip(ups, skips[::-1]):
            y = u((y, sk), training=training)
        return tf.reshape(y, [B, T, H, W, 3])

    @tf.function
    def step():
        with tf.GradientTape() as tape:
            logits = forward(True)
            valid = tf.cast(tf.greater(gt, -1), tf.float32)
            pix_w = tf.reduce_sum(tf.one_hot(tf.cast(gt, tf.int32), 3) * cw, -1)
            ce = tf.nn.sparse_softmax_cross_entropy_with_logits(tf.cast(tf.maximum(gt, 0), tf.int32), logits)
            loss = tf.reduce_sum(ce * pix_w * valid) / (tf.reduce_sum(valid) + 0.00001)
        variables = [v for m in downs + ups for v in m.trainable_variables]
        opt.apply_gradients(zip(tape.gradient(loss, variables), variables))
        return loss

    def timed():
        t0 = time.time()
        float(step().numpy())
        return time.time() - t0
    return timed


def main(out=None):
    global GOLDEN
    if out is None and '--out' in sys.argv:
        out = sys.argv[sys.argv.index('--out') + 1]
    if out:
        os.makedirs(out, exist_ok=True)
        GOLDEN = out
    tf, why = probe_tensorflow()
    if tf is None:
        print('tf_pin: TensorFlow is not importable here (%s) -- nothing written; the oracle stays "parity unpinned".' % why)
        print('tf_pin: python %s, executable %s' % (sys.version.split()[0], sys.executable))
        return 1
    if int(tf.__version__.split('.')[0]) != 2:
        print('tf_pin: the reference requires TensorFlow 2.x (train2D.py:25-26), found', tf.__version__)
        return 1
    try:
        from tensorflow.python import keras as k          # the reference's import (Networks.py:5-8)
    except ImportError:
        from tensorflow import keras as k
    rng = np.random.default_rng(2024)
    info = {'tensorflow': tf.__version__, 'keras_module': k.__name__}
    fmt = 'channels_last'                                  # stock CPU builds of TF run Conv2D in NHWC only

    # ---- ConvLSTM2D: two stateful calls (carried state), 5x5 and 3x3 -------------------------------------------------
    for name, (ksz, cin, f, hw) in {'convlstm_k5': (5, 3, 8, (9, 11)), 'convlstm_k3': (3, 1, 4, (8, 8))}.items():
        layer = k.layers.ConvLSTM2D(filters=f, kernel_size=ksz, strides=1, padding='same', data_format=fmt,
                                    return_sequences=True, stateful=True)
        x1 = rng.standard_normal((2, 3) + hw + (cin,)).astype(np.float32)
        x2 = rng.standard_normal((2, 3) + hw + (cin,)).astype(np.float32)
        y1 = layer(tf.constant(x1)).numpy()
        w = [v.numpy() for v in layer.weights[:3]]
        # perturb so that nothing is pinned only at the initialiser's special values (zero bias, unit forget bias)
        w = [a + 0.1 * rng.standard_normal(a.shape).astype(np.float32) for a in w]
        layer.set_weights(w + [v.numpy() for v in layer.weights[3:]])
        layer.reset_states()
        y1 = layer(tf.constant(x1)).numpy()
        y2 = layer(tf.constant(x2)).numpy()
        states = [s.numpy() for s in layer.states]
        np.savez(os.path.join(GOLDEN, 'tf_%s.npz' % name), kernel=w[0], recurrent_kernel=w[1], bias=w[2], x1=x1, x2=x2,
                 y1=y1, y2=y2, h=states[0], c=states[1], weight_names=np.array([v.name for v in layer.weights]))
    # ---- Conv2D 'same': stride 1 / 2, even and odd inputs ------------------------------------------------------------
    out = {}
    for tag, (ksz, stride, hw) in {'k3s2_even': (3, 2, (8, 10)), 'k3s2_odd': (3, 2, (7, 9)), 'k5s2_even': (5, 2, (8, 8)),
                                   'k3s1': (3, 1, (6, 7)), 'k1': (1, 1, (5, 5))}.items():
        layer = k.layers.Conv2D(filters=5, kernel_size=ksz, strides=stride, use_bias=True, data_format=fmt, padding='same')
        x = rng.standard_normal((2,) + hw + (3,)).astype(np.float32)
        layer(tf.constant(x))
        w = [v.numpy() + 0.1 * rng.standard_normal(v.shape).astype(np.float32) for v in layer.weights]
        layer.set_weights(w)
        out.update({tag + '_x': x, tag + '_kernel': w[0], tag + '_bias': w[1], tag + '_y': layer(tf.constant(x)).numpy()})
    np.savez(os.path.join(GOLDEN, 'tf_conv2d.npz'), **out)
    # ---- BatchNormalization (train twice, then infer) + LeakyReLU -----------------------------------------------------
    bn = k.layers.BatchNormalization(axis=-1)
    x1 = rng.standard_normal((4, 6, 5, 3)).astype(np.float32) * 2 + 1
    x2 = rng.standard_normal((4, 6, 5, 3)).astype(np.float32) - 0.5
    bn(tf.constant(x1), training=True)
    bn.set_weights([np.array([1.5, 0.5, 1.0], np.float32), np.array([0.1, -0.2, 0.0], np.float32),
                    np.zeros(3, np.float32), np.ones(3, np.float32)])
    y1 = bn(tf.constant(x1), training=True).numpy()
    mm1, mv1 = bn.moving_mean.numpy().copy(), bn.moving_variance.numpy().copy()
    y2 = bn(tf.constant(x2), training=True).numpy()
    yi = bn(tf.constant(x2), training=False).numpy()
    lr = k.layers.LeakyReLU()
    np.savez(os.path.join(GOLDEN, 'tf_bn_lrelu.npz'), x1=x1, x2=x2, y1=y1, y2=y2, y_infer=yi, mm1=mm1, mv1=mv1,
             mm2=bn.moving_mean.numpy(), mv2=bn.moving_variance.numpy(), gamma=bn.gamma.numpy(), beta=bn.beta.numpy(),
             lrelu_x=x1, lrelu_y=lr(tf.constant(x1)).numpy(), eps=np.float32(bn.epsilon), momentum=np.float32(bn.momentum))
    n = x1.size // 3
    var = x1.reshape(-1, 3).var(0)
    info['bn_moving_variance_rule'] = 'unbiased' if np.allclose(mv1, 0.99 + 0.01 * var * n / (n - 1), atol=1e-5) else (
        'biased' if np.allclose(mv1, 0.99 + 0.01 * var, atol=1e-5) else 'unknown')
    # ---- bilinear resize convention -----------------------------------------------------------------------------------
    x = rng.standard_normal((2, 4, 5, 3)).astype(np.float32)
    y = k.backend.resize_images(tf.constant(x), 2, 2, fmt, interpolation='bilinear').numpy()
    from oracle import np_oracle as npo
    info['resize_images_bilinear'] = 'tf2.0' if np.allclose(y, npo.resize_bilinear(x, 2, 'tf2.0'), atol=1e-5) else (
        'half_pixel' if np.allclose(y, npo.resize_bilinear(x, 2, 'half_pixel'), atol=1e-5) else 'unknown')
    xr = rng.standard_normal((1, 5, 6, 2)).astype(np.float32)
    yr = tf.pad(tf.constant(xr), [[0, 0], [2, 3], [1, 4], [0, 0]], 'REFLECT').numpy()
    np.savez(os.path.join(GOLDEN, 'tf_resize_pad.npz'), x=x, y=y, pad_x=xr, pad_y=yr)
    # ---- weighted CE (losses.py:13-27) and one Adam step --------------------------------------------------------------
    logits = rng.standard_normal((2, 3, 6, 7, 3)).astype(np.float32)
    gt = rng.integers(-1, 3, size=(2, 3, 6, 7)).astype(np.float32)
    cw = np.array([0.15, 0.25, 0.6], np.float32)
    var = tf.Variable(logits)
    with tf.GradientTape() as tape:
        valid = tf.cast(tf.greater(gt, -1), tf.float32)
        onehot = tf.one_hot(tf.cast(gt, tf.int32), 3)
        pix_w = tf.reduce_sum(onehot * cw, -1)
        ce = tf.nn.sparse_softmax_cross_entropy_with_logits(tf.cast(tf.maximum(gt, 0), tf.int32), var)
        loss = tf.reduce_sum(ce * pix_w * valid) / (tf.reduce_sum(valid) + 0.00001)
    g = tape.gradient(loss, var)
    opt = k.optimizers.Adam(1e-3)
    p0 = rng.standard_normal((5, 4)).astype(np.float32)
    pv = tf.Variable(p0)
    g1 = rng.standard_normal((5, 4)).astype(np.float32)
    g2 = rng.standard_normal((5, 4)).astype(np.float32)
    opt.apply_gradients([(tf.constant(g1), pv)])
    p1 = pv.numpy().copy()
    opt.apply_gradients([(tf.constant(g2), pv)])
    np.savez(os.path.join(GOLDEN, 'tf_loss_adam.npz'), logits=logits, gt=gt, class_weights=cw, loss=loss.numpy(),
             dlogits=g.numpy(), p0=p0, g1=g1, g2=g2, p1=p1, p2=pv.numpy(), lr=np.float32(1e-3),
             eps=np.float32(getattr(opt, 'epsilon', 1e-7)))
    # ---- tensor-bundle cross-check -------------------------------------------------------------------------------------
    import tempfile
    import tf_bundle as tb

    class Blk(k.Model):
        def __init__(self):
            super().__init__()
            self.ConvLSTM = [k.layers.ConvLSTM2D(4, 3, padding='same', return_sequences=True, stateful=True)]
            self.Conv = [k.layers.Conv2D(4, 3, padding='same')]
            self.BN = [k.layers.BatchNormalization()]

        def call(self, x, training=None):
            y = self.ConvLSTM[0](x)
            y = tf.reshape(y, [-1] + y.shape[2:].as_list())
            return self.BN[0](self.Conv[0](y), training)

    class Net(k.Model):
        def __init__(self):
            super().__init__()
            self.DownLayers = [Blk(), Blk()]

        def call(self, x, training=None):
            for b in self.DownLayers:
                y = b(x, training)
            return y

    net = Net()
    net(tf.constant(rng.standard_normal((1, 2, 6, 6, 2)).astype(np.float32)), True)
    with tempfile.TemporaryDirectory() as tmp:
        prefix = os.path.join(tmp, 'model.ckpt')
        net.save_weights(prefix, save_format='tf')
        keys = sorted(kk for kk in tb.list_bundle(prefix) if kk.endswith(tb.SUFFIX))
        info['tf_checkpoint_keys_sample'] = keys[:6]
        mine = tb.read_bundle(prefix)
        theirs = tf.train.load_checkpoint(prefix)
        info['tf_bundle_reads_tf_checkpoint'] = all(np.array_equal(mine[kk], theirs.get_tensor(kk)) for kk in keys)
        tb.write_bundle(prefix + '.mine', {kk: mine[kk] for kk in keys})
        back = tf.train.load_checkpoint(prefix + '.mine')
        info['tf_reads_tf_bundle_checkpoint'] = all(np.array_equal(mine[kk], back.get_tensor(kk)) for kk in keys)
        info['attribute_path_keys'] = any(kk.startswith('DownLayers/0/ConvLSTM/0/cell/kernel') for kk in keys)
    with open(os.path.join(GOLDEN, 'tf_pin.json'), 'w') as fh:
        json.dump(info, fh, indent=1, sort_keys=True)
    print(json.dumps(info, indent=1, sort_keys=True))
    return 0


if __name__ == '__main__':
    sys.exit(main())
