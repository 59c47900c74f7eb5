"""Kernel SOURCES vs the oracle, on CPU: lstm-unet_amd/csrc/*.hip compiled for the host SIMT
emulator (tests/emu, test infrastructure) and driven through the same C ABI the GPU build exports.
These catch index / fragment-layout / masking mistakes without a GPU; the `-m gpu` tests repeat the
comparisons on the real gfx950 build."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import np_oracle as npo
from oracle import torch_oracle as tho
import emu_np as E
from emu_np import f32, ptr
from lu_native import calls


@pytest.fixture(scope='module')
def lib():
    return E.emu_lib()


RNG = np.random.default_rng(11)


def rnd(*shape, scale=1.0):
    return f32(RNG.standard_normal(shape) * scale)


def close(a, b, tol):
    err = float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
    assert np.isfinite(err) and err <= tol, err


CONV_CASES = [  # frames, H, W, C, N, k, stride
    (1, 8, 8, 16, 32, 3, 1), (2, 9, 7, 8, 12, 3, 1), (1, 10, 12, 20, 40, 5, 1), (2, 8, 10, 1, 8, 5, 1),
    (1, 9, 9, 3, 70, 3, 2), (1, 8, 8, 24, 130, 3, 2), (1, 6, 6, 8, 3, 1, 1), (3, 5, 5, 4, 33, 5, 2),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd(lib, case):
    fr, H, W, Cc, N, k, s = case
    x, w, b = rnd(fr, H, W, Cc), rnd(k, k, Cc, N, scale=0.2), rnd(N)
    close(E.conv2d(lib, [x], [w], b, k, s), npo.conv2d_same(x, w, b, s), 5e-5)


def test_conv_two_sources_and_strided_views(lib):
    """UpBlock2D concat([up, skip]) (Networks.py:145) as two sources reading channel slices."""
    xa, xb = rnd(2, 6, 7, 12), rnd(2, 6, 7, 1)
    w = rnd(3, 3, 13, 20, scale=0.2)
    b = rnd(20)
    ref = npo.conv2d_same(np.concatenate([xa, xb], -1), w, b, 1)
    wa, wb = w[:, :, :12, :], w[:, :, 12:, :]   # views: tap stride stays 13*20
    srcs = [calls.conv_src(ptr(xa), xa.strides[0] // 4, 12, 12, wa.ctypes.data, 13 * 20, 20),
            calls.conv_src(ptr(xb), xb.strides[0] // 4, 1, 1, wb.ctypes.data, 13 * 20, 20)]
    out = np.full((2, 6, 7, 20), np.nan, np.float32)
    calls.conv2d(lib, None, srcs, 2, 6, 7, 6, 7, 3, 1, 1, 1, 1, 20, ptr(b), ptr(out), 6 * 7 * 20, 20)
    close(out, ref, 5e-5)
    # a source that is a channel slice of a wider activation tensor (pix_stride > C)
    wide = rnd(2, 6, 7, 24)
    w2 = rnd(3, 3, 8, 16, scale=0.2)
    srcs = [calls.conv_src(wide[..., 8:16].ctypes.data, wide.strides[0] // 4, 24, 8, ptr(w2), 8 * 16, 16)]
    out = np.full((2, 6, 7, 16), np.nan, np.float32)
    calls.conv2d(lib, None, srcs, 2, 6, 7, 6, 7, 3, 1, 1, 1, 1, 16, None, ptr(out), 6 * 7 * 16, 16)
    close(out, npo.conv2d_same(wide[..., 8:16], w2, None, 1), 5e-5)


def _torch_conv_grads(x, w, dy, stride):
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    y = tho.conv2d_same(xt, wt, None, stride)
    gx, gw = torch.autograd.grad(y, [xt, wt], torch.tensor(dy, dtype=torch.float64))
    return gx.numpy(), gw.numpy()


@pytest.mark.parametrize('case', [(2, 8, 9, 8, 12, 3, 1), (1, 7, 7, 4, 8, 5, 1), (2, 8, 8, 8, 16, 3, 2),
                                  (1, 9, 7, 12, 8, 3, 2), (1, 8, 8, 4, 4, 5, 2), (1, 6, 6, 3, 5, 3, 1),
                                  (1, 6, 5, 8, 3, 1, 1)])
def test_conv_dgrad_wgrad(lib, case):
    fr, H, W, Cc, N, k, s = case
    x, w = rnd(fr, H, W, Cc), rnd(k, k, Cc, N, scale=0.3)
    Ho, Wo = calls.same_pad(H, k, s)[0], calls.same_pad(W, k, s)[0]
    dy = rnd(fr, Ho, Wo, N)
    gx, gw = _torch_conv_grads(x, w, dy, s)
    close(E.conv2d_dgrad(lib, dy, w, (H, W), s), gx, 5e-5)
    close(E.conv2d_wgrad(lib, x, dy, k, s, splits=1), gw, 1e-4)
    close(E.conv2d_wgrad(lib, x, dy, k, s, splits=3), gw, 1e-4)


def test_wgrad_wide_channels_and_beta(lib):
    x, dy = rnd(1, 6, 6, 132), rnd(1, 6, 6, 136)
    _, gw = _torch_conv_grads(x, rnd(3, 3, 132, 136), dy, 1)
    dw0 = rnd(3, 3, 132, 136)
    got = E.conv2d_wgrad(lib, x, dy, 3, 1, splits=2, dw=dw0.copy(), beta=1.0)
    close(got, gw + dw0, 2e-4)
    x, dy = rnd(1, 6, 6, 40), rnd(1, 6, 6, 8)
    _, gw = _torch_conv_grads(x, rnd(3, 3, 40, 8), dy, 1)
    close(E.conv2d_wgrad(lib, x, dy, 3, 1), gw, 1e-4)


@pytest.mark.parametrize('k,cin', [(3, 8), (5, 1)])
def test_convlstm_fused_step(lib, k, cin):
    """Fused two-source conv + gate epilogue == Keras ConvLSTM2D cell step."""
    F = 32
    x, h, c = rnd(2, 6, 7, cin), rnd(2, 6, 7, F, scale=0.5), rnd(2, 6, 7, F)
    ker, rec, b = rnd(k, k, cin, 4 * F, scale=0.3), rnd(k, k, F, 4 * F, scale=0.1), rnd(4 * F, scale=0.5)
    h1, c1 = npo.convlstm_step(x, h, c, ker, rec, b)
    hg, cg, gates = E.convlstm_step_fused(lib, x, h, c, ker, rec, b)
    close(hg, h1, 2e-5)
    close(cg, c1, 2e-5)
    z = npo.conv2d_same(x, ker, b) + npo.conv2d_same(h, rec)
    close(gates[..., :F], npo.hard_sigmoid(z[..., :F]), 2e-5)
    close(gates[..., 2 * F:3 * F], np.tanh(z[..., 2 * F:3 * F]), 2e-5)


def test_lstm_gates_pointwise_fwd_bwd(lib):
    fr, H, W, F = 2, 4, 5, 6
    z, c0 = rnd(fr, H, W, 4 * F, scale=2.0), rnd(fr, H, W, F)
    c1 = np.full_like(c0, np.nan)
    hseq = np.full((fr, 3, H, W, F), np.nan, np.float32)   # h written into slot t=1 of [B,T,...]
    gates = np.full_like(z, np.nan)
    calls.check(lib, lib.lu_lstm_gates_fwd(ptr(z), ptr(c0), ptr(c1), hseq[:, 1].ctypes.data, ptr(gates), fr, H * W, F,
                                           3 * H * W * F, None), 'gates_fwd')
    zt = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    ct = torch.tensor(c0, dtype=torch.float64, requires_grad=True)
    i, f, g, o = [zt[..., j * F:(j + 1) * F] for j in range(4)]
    cn = tho.hard_sigmoid(f) * ct + tho.hard_sigmoid(i) * torch.tanh(g)
    hn = tho.hard_sigmoid(o) * torch.tanh(cn)
    close(c1, cn.detach().numpy(), 1e-5)
    close(hseq[:, 1], hn.detach().numpy(), 1e-5)
    dh, dh2, dc = rnd(fr, H, W, F), rnd(fr, H, W, F), rnd(fr, H, W, F)
    gz, gc = torch.autograd.grad([hn, cn], [zt, ct], [torch.tensor(dh + dh2, dtype=torch.float64),
                                                      torch.tensor(dc, dtype=torch.float64)])
    dz, dcp = np.full_like(z, np.nan), np.full_like(c0, np.nan)
    calls.check(lib, lib.lu_lstm_gates_bwd(ptr(gates), ptr(c0), ptr(c1), ptr(dh), H * W * F, ptr(dh2), ptr(dc),
                                           ptr(dz), ptr(dcp), fr, H * W, F, None), 'gates_bwd')
    close(dz, gz.numpy(), 1e-5)
    close(dcp, gc.numpy(), 1e-5)


def test_bn_lrelu_fwd_bwd(lib):
    rows, Cc = 300, 20
    x = rnd(rows, Cc, scale=2.0) + 0.5
    gamma, beta = f32(1 + 0.2 * RNG.random(Cc)), rnd(Cc, scale=0.3)
    mm, mv = rnd(Cc, scale=0.1), f32(1 + RNG.random(Cc))
    ws = np.empty(lib.lu_colreduce_workspace_bytes(rows, Cc) // 8 + 1, np.float64)
    sums = np.empty(2 * Cc, np.float64)
    calls.check(lib, lib.lu_bn_stats(ptr(x), rows, Cc, ptr(sums), ptr(ws), None), 'stats')
    close(sums[:Cc], x.astype(np.float64).sum(0), 1e-3)
    scale, shift, smean, sinv = [np.empty(Cc, np.float32) for _ in range(4)]
    mm2, mv2 = mm.copy(), mv.copy()
    calls.check(lib, lib.lu_bn_finalize_train(ptr(sums), float(rows), ptr(gamma), ptr(beta), 1e-3, 0.99, ptr(mm2),
                                              ptr(mv2), ptr(scale), ptr(shift), ptr(smean), ptr(sinv), Cc, None), 'fin')
    y = np.empty_like(x)
    calls.check(lib, lib.lu_bn_lrelu_apply(ptr(x), ptr(y), ptr(scale), ptr(shift), 0.3, rows, Cc, None), 'apply')
    xr = x.reshape(1, 1, rows, Cc)
    yr, mean, var = npo.batchnorm_train(xr, gamma, beta)
    close(y, npo.leaky_relu(yr).reshape(rows, Cc), 2e-5)
    emm, emv = npo.batchnorm_moving_update(mm, mv, mean, var, rows)
    close(mm2, emm, 1e-6)
    close(mv2, emv, 1e-5)
    # backward vs torch autograd
    dy = rnd(rows, Cc)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    gt_, bt_ = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (gamma, beta)]
    z, _, _ = tho.bn_train(xt.reshape(1, 1, rows, Cc), gt_, bt_)
    out = torch.nn.functional.leaky_relu(z, 0.3).reshape(rows, Cc)
    gx, gg, gb = torch.autograd.grad(out, [xt, gt_, bt_], torch.tensor(dy, dtype=torch.float64))
    bs = np.empty(2 * Cc, np.float64)
    calls.check(lib, lib.lu_bn_lrelu_bwd_reduce(ptr(x), ptr(dy), ptr(scale), ptr(shift), ptr(smean), ptr(sinv), 0.3,
                                                rows, Cc, ptr(bs), ptr(ws), None), 'bwd_reduce')
    dx, dg, db = np.empty_like(x), np.empty(Cc, np.float32), np.empty(Cc, np.float32)
    calls.check(lib, lib.lu_bn_lrelu_bwd_apply(ptr(x), ptr(dy), ptr(scale), ptr(shift), ptr(smean), ptr(sinv), 0.3,
                                               ptr(bs), float(rows), ptr(dx), ptr(dg), ptr(db), rows, Cc, None), 'bwd')
    close(dx, gx.numpy(), 2e-5)
    close(dg, gg.numpy(), 2e-4)
    close(db, gb.numpy(), 2e-4)
    # inference path
    calls.check(lib, lib.lu_bn_finalize_infer(ptr(gamma), ptr(beta), ptr(mm), ptr(mv), 1e-3, ptr(scale), ptr(shift),
                                              Cc, None), 'fin_inf')
    calls.check(lib, lib.lu_bn_lrelu_apply(ptr(x), ptr(y), ptr(scale), ptr(shift), 0.3, rows, Cc, None), 'apply')
    close(y, npo.leaky_relu(npo.batchnorm_infer(x, gamma, beta, mm, mv)), 2e-5)


def test_colsum(lib):
    x = rnd(777, 10)
    ws = np.empty(lib.lu_colreduce_workspace_bytes(777, 6) // 8 + 1, np.float64)
    out = f32(np.ones(6))
    calls.check(lib, lib.lu_colsum(x[:, 2:8].ctypes.data, 777, 6, 10, ptr(out), 1.0, ptr(ws), None), 'colsum')
    close(out, 1 + x[:, 2:8].astype(np.float64).sum(0), 1e-4)


def test_upsample_fwd_bwd(lib):
    fr, H, W, Cc = 2, 5, 4, 3
    x = rnd(fr, H, W, Cc)
    y = np.empty((fr, 2 * H, 2 * W, Cc), np.float32)
    calls.check(lib, lib.lu_upsample2x_fwd(ptr(x), ptr(y), fr, H, W, Cc, None), 'up')
    close(y, npo.resize_bilinear(x, 2), 1e-6)
    dyw = rnd(fr, 2 * H, 2 * W, Cc + 2)    # gradient arrives as a channel slice of a wider tensor
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    (gx,) = torch.autograd.grad(tho.resize_bilinear(xt, 2), [xt], torch.tensor(dyw[..., :Cc], dtype=torch.float64))
    dx = np.empty_like(x)
    calls.check(lib, lib.lu_upsample2x_bwd(ptr(dyw), Cc + 2, ptr(dx), fr, H, W, Cc, None), 'upb')
    close(dx, gx.numpy(), 1e-5)


def test_window_copy_reflect_crop_embed(lib):
    x = rnd(2, 5, 6, 2)
    y = np.empty((2, 5 + 2 + 3, 6 + 1 + 4, 2), np.float32)
    calls.check(lib, lib.lu_window_copy(ptr(x), 2, ptr(y), 2, 5, 6, 10, 11, 2, 2, 1, 1, 0.0, None), 'reflect')
    assert np.array_equal(y, npo.reflect_pad_hw(x, (2, 3), (1, 4)).astype(np.float32))
    crop = np.empty((2, 5, 6, 2), np.float32)
    calls.check(lib, lib.lu_window_copy(ptr(y), 2, ptr(crop), 2, 10, 11, 5, 6, 2, -2, -1, 0, 0.0, None), 'crop')
    assert np.array_equal(crop, x)
    emb = np.ones_like(y)
    calls.check(lib, lib.lu_window_copy(ptr(x), 2, ptr(emb), 2, 5, 6, 10, 11, 2, 2, 1, 0, 1.0, None), 'embed')
    exp = np.ones_like(y)
    exp[:, 2:7, 1:7] += x
    assert np.array_equal(emb, exp)


def test_softmax_wce(lib):
    rows = 1000
    lg = rnd(rows, 3, scale=2.0)
    gt = f32(RNG.integers(-1, 3, rows))
    cw = f32([0.15, 0.25, 0.6])
    ws = np.empty(lib.lu_wce_workspace_bytes(rows) // 8 + 1, np.float64)
    sums, sm, loss = np.empty(2, np.float64), np.empty_like(lg), np.empty(1, np.float32)
    calls.check(lib, lib.lu_softmax_wce_fwd(ptr(lg), ptr(gt), ptr(cw), ptr(sm), ptr(sums), rows, ptr(ws), None), 'f')
    calls.check(lib, lib.lu_wce_finalize(ptr(sums), ptr(loss), None), 'fin')
    assert abs(loss[0] - npo.weighted_ce(gt, lg, cw)) < 1e-5
    close(sm, npo.softmax(lg.astype(np.float64)), 1e-6)
    lt = torch.tensor(lg, dtype=torch.float64, requires_grad=True)
    (gl,) = torch.autograd.grad(tho.weighted_ce(torch.tensor(gt, dtype=torch.float64), lt, cw.tolist()), [lt])
    dl = np.empty_like(lg)
    calls.check(lib, lib.lu_softmax_wce_bwd(ptr(lg), ptr(gt), ptr(cw), ptr(sums), 1.0, ptr(dl), rows, None), 'b')
    close(dl, gl.numpy(), 1e-7)


def test_adam_scale_transpose_add(lib):
    n = 1003
    p, g = rnd(n), rnd(n)
    m, v = rnd(n, scale=0.1), f32(RNG.random(n) * 0.01)
    step, lr = 3, 1e-3
    alpha = lr * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
    ep, em, ev = npo.adam_step(p.astype(np.float64), 0.5 * g.astype(np.float64), m.astype(np.float64),
                               v.astype(np.float64), step, lr=lr)
    calls.check(lib, lib.lu_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), n, alpha, 0.9, 0.999, 1e-7, 0.5, None), 'adam')
    close(p, ep, 1e-6)
    close(m, em, 1e-6)
    close(v, ev, 1e-6)
    st = rnd(3, 40)
    keep = f32([1, 0, 1])
    exp = st * keep[:, None]
    calls.check(lib, lib.lu_scale_frames(ptr(st), ptr(keep), 3, 40, None), 'mask')
    assert np.array_equal(st, exp)
    a = rnd(2, 35, 3)
    b = np.empty((2, 3, 35), np.float32)
    calls.check(lib, lib.lu_transpose_inner(ptr(a), ptr(b), 2, 35, 3, None), 'tr')
    assert np.array_equal(b, a.transpose(0, 2, 1))
    y0, x0 = rnd(100), rnd(100)
    e = y0 + x0
    calls.check(lib, lib.lu_add_inplace(ptr(y0), ptr(x0), 100, None), 'add')
    assert np.array_equal(y0, e)
