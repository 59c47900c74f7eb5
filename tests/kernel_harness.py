"""TEST INFRASTRUCTURE: backend-neutral wrappers over the C ABI.
  * backend 'emu': kernel sources compiled for the host SIMT emulator (tests/emu), numpy buffers
  * backend 'hip': the real gfx950 library, torch device buffers (needs a GPU; `-m gpu`)
Mirrors what lu_native/ops.py does; every call goes through include/lstm_unet_hip.h's C ABI."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
from lu_native import cabi, calls  # noqa: E402


class EmuBackend(object):
    name = 'emu'

    def __init__(self):
        sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
        import build_emu
        self.lib = cabi.bind(build_emu.build())
        self.stream = None

    def dev(self, a, dtype=np.float32):
        return np.array(a, dtype=dtype, order="C", copy=True)

    def empty(self, shape, dtype=np.float32):
        if dtype == np.int16:
            return np.full(shape, 0x7fc0, np.int16)      # bf16 NaN pattern
        return np.full(shape, np.nan, dtype)

    def ptr(self, d, offset_elems=0):
        return None if d is None else d.ctypes.data + offset_elems * d.itemsize

    def host(self, d):
        return d


class HipBackend(object):
    name = 'hip'

    def __init__(self):
        import torch
        from lu_native import ops
        self.torch = torch
        self.lib = ops.lib()
        self.stream = None   # default stream

    def dev(self, a, dtype=np.float32):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()

    def empty(self, shape, dtype=np.float32):
        if dtype == np.int16:
            return self.torch.full(shape, 0x7fc0, dtype=self.torch.int16, device='cuda')      # bf16 NaN pattern
        t = self.torch.empty(shape, dtype=self.torch.float32 if dtype == np.float32 else self.torch.float64,
                             device='cuda')
        return t.fill_(float('nan'))

    def ptr(self, d, offset_elems=0):
        return None if d is None else d.data_ptr() + offset_elems * d.element_size()

    def host(self, d):
        self.torch.cuda.synchronize()
        return d.cpu().numpy()


_BACKENDS = {}


def backend(name):
    if name not in _BACKENDS:
        _BACKENDS[name] = EmuBackend() if name == 'emu' else HipBackend()
    return _BACKENDS[name]


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def bf16_round(a):
    """fp32 -> nearest-even bf16 -> fp32 (what the bf16 kernels do to their MFMA operands)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))


def bf16_bits(a):
    """fp32 -> nearest-even bf16 bit patterns (int16 array): a bf16 tensor as the C ABI sees it."""
    return (bf16_round(a).view(np.uint32) >> 16).astype(np.uint16).view(np.int16)


def bf16_values(bits):
    """int16 bf16 bit patterns -> fp32 values."""
    return (np.ascontiguousarray(bits).view(np.uint16).astype(np.uint32) << 16).view(np.float32)


def pack_bf16(be, wd, k, Cin, N):
    nbytes = be.lib.lu_pack_weights_bf16_bytes(k, Cin, N)
    out = be.empty((nbytes // 4 + 4,))
    calls.check(be.lib, be.lib.lu_pack_weights_bf16(be.ptr(wd), Cin * N, N, k, Cin, N, be.ptr(out), be.stream), 'pack')
    return out


def conv2d(be, srcs, ws, bias, k, stride=1, dil=1, pad=None, out_hw=None, N=None, splits=1, precision=0, flags=0,
           bf16_src=(), post=None, slabs=False):
    """srcs: [frames,H,W,C] numpy arrays; ws: [k,k,C,N] numpy arrays. Returns numpy.
    bf16_src: indices of sources handed over as bf16 tensors (precision 1).  post = (scale, shift, alpha) numpy / float.
    slabs=True: LU_CONV_F_SLABS_ONLY -> the [splits, frames*Hout*Wout, N] partial slabs."""
    frames, Hin, Win = srcs[0].shape[:3]
    if N is None:
        N = ws[0].shape[-1]
    if pad is None:
        Hout, pt, _ = calls.same_pad(Hin, k, stride)
        Wout, pl, _ = calls.same_pad(Win, k, stride)
    else:
        pt, pl = pad
        Hout, Wout = out_hw
    out = be.empty((frames, Hout, Wout, N))
    keep, cs = [], []
    for si, (x, w) in enumerate(zip(srcs, ws)):
        Cin = x.shape[3]
        b16 = si in bf16_src
        xd, wd = (be.dev(bf16_bits(x), np.int16) if b16 else be.dev(x)), be.dev(w)
        keep += [xd, wd]
        if precision:
            wd = pack_bf16(be, wd, k, Cin, N)
            keep.append(wd)
        cs.append(calls.conv_src(be.ptr(xd), Hin * Win * Cin, Cin, Cin, be.ptr(wd), Cin * N, N,
                                 dtype=cabi.LU_BF16 if b16 else cabi.LU_F32))
    bd = None if bias is None else be.dev(bias)
    wsb = be.empty((splits * frames * Hout * Wout * N,)) if splits > 1 else None
    pd = None
    if post is not None:
        keep += [be.dev(post[0]), be.dev(post[1])]
        pd = (be.ptr(keep[-2]), be.ptr(keep[-1]), float(post[2]))
    calls.conv2d(be.lib, be.stream, cs, frames, Hin, Win, Hout, Wout, k, stride, dil, pt, pl, N, be.ptr(bd),
                 be.ptr(out), Hout * Wout * N, N, splits=splits, workspace=be.ptr(wsb), precision=precision,
                 flags=flags | (cabi.LU_CONV_F_SLABS_ONLY if slabs else 0), post=pd)
    if slabs:
        return be.host(wsb).reshape(splits, frames * Hout * Wout, N)
    return be.host(out)


def flip_transpose(be, w, c_off=0, C_sub=None):
    k, _, Ct, N = w.shape
    C_sub = Ct - c_off if C_sub is None else C_sub
    wt = be.empty((k, k, N, C_sub))
    wd = be.dev(w)
    calls.check(be.lib, be.lib.lu_weight_flip_transpose(be.ptr(wd), be.ptr(wt), k, Ct, N, c_off, C_sub, be.stream),
                'flip')
    return be.host(wt)


def conv2d_dgrad_s2_parity(be, dy, w, in_hw):
    """Stride-2 input gradient as four parity-plane convolutions (mirrors lu_native.ops._conv2d_dgrad_stride2)."""
    k, _, Cc, N = w.shape
    Hin, Win = in_hw
    frames, Hd, Wd, _ = dy.shape
    _, pt, _ = calls.same_pad(Hin, k, 2)
    _, pl, _ = calls.same_pad(Win, k, 2)
    ks = (k + 1) // 2

    def axis(par, pad):
        offs = [(par + pad - kh) // 2 for kh in range(k) if (par + pad - kh) % 2 == 0]
        return (-min(offs), len(offs)) if offs else (0, 0)

    (py0, ny0), (py1, ny1) = axis(0, pt), axis(1, pt)
    (px0, nx0), (px1, nx1) = axis(0, pl), axis(1, pl)
    sub = be.empty(((ny0 + ny1) * (nx0 + nx1) * N * Cc,))          # four compact planes, back to back
    wd, dyd = be.dev(w), be.dev(dy)
    calls.check(be.lib, be.lib.lu_stride2_dgrad_weights(be.ptr(wd), be.ptr(sub), k, ks, Cc, N, pt, pl, py0, py1, px0, px1,
                                                        be.stream), 's2w')
    out = be.empty((frames, Hin, Win, Cc))
    off = 0
    for py, (pady, ny) in enumerate(((py0, ny0), (py1, ny1))):
        for px, (padx, nx) in enumerate(((px0, nx0), (px1, nx1))):
            taps = ny * nx
            Hs, Ws = (Hin - py + 1) // 2, (Win - px + 1) // 2
            if Hs > 0 and Ws > 0 and taps > 0:
                src = calls.conv_src(be.ptr(dyd), Hd * Wd * N, N, N, be.ptr(sub, off * N * Cc), N * Cc, Cc)
                calls.conv2d(be.lib, be.stream, [src], frames, Hd, Wd, Hs, Ws, nx, 1, 1, pady, padx, Cc, None,
                             be.ptr(out, (py * Win + px) * Cc), Hin * Win * Cc, 2 * Cc, out_row_stride=2 * Win * Cc, k_h=ny)
            elif Hs > 0 and Ws > 0:                                 # a parity class without taps: zero gradient
                o = be.host(out)
                o[:, py::2, px::2, :] = 0.0
            off += taps
    return be.host(out)


def conv2d_s2_fwd_bf16(be, x, w, b):
    """Stride-2 3x3 forward convolution on even extents from a bf16 tensor (lu_conv2d_s2_fwd_bf16)."""
    k, _, Cc, N = w.shape
    frames, H, W, _ = x.shape
    assert k == 3
    xd, wd = be.dev(bf16_bits(x), np.int16), be.dev(w)
    packed = pack_bf16(be, wd, 3, Cc, N)
    bd = be.dev(b) if b is not None else None
    out = be.empty((frames, H // 2, W // 2, N))
    calls.check(be.lib, be.lib.lu_conv2d_s2_fwd_bf16(be.ptr(xd), H * W * Cc, Cc, be.ptr(packed), be.ptr(bd) if bd is not None else None,
                                                     frames, H, W, Cc, N, be.ptr(out), be.stream), 's2 fwd')
    return be.host(out)


def conv2d_dgrad_s2_fused_bf16(be, dy, w):
    """Stride-2 3x3 input gradient on even extents, all four parity classes in one launch (lu_conv2d_s2_dgrad_bf16)."""
    k, _, Cc, N = w.shape
    frames, Hd, Wd, _ = dy.shape
    assert k == 3
    sub = be.empty((9 * N * Cc,))
    wd, dyd = be.dev(w), be.dev(dy)
    calls.check(be.lib, be.lib.lu_stride2_dgrad_weights(be.ptr(wd), be.ptr(sub), 3, 2, Cc, N, 0, 0, 1, 0, 1, 0, be.stream), 's2w')
    packed = be.empty((9 * -(-N // 32) * -(-Cc // 32) * 512,))       # 1024 bf16 per (tap, chunk, fragment)
    calls.check(be.lib, be.lib.lu_pack_weights_taps_bf16(be.ptr(sub), N * Cc, Cc, 9, N, Cc, be.ptr(packed), be.stream), 'pack')
    out = be.empty((frames, 2 * Hd, 2 * Wd, Cc))
    calls.check(be.lib, be.lib.lu_conv2d_s2_dgrad_bf16(be.ptr(dyd), Hd * Wd * N, N, be.ptr(packed), frames, Hd, Wd, N, Cc,
                                                       be.ptr(out), be.stream), 's2 dgrad')
    return be.host(out)


def conv2d_dgrad(be, dy, w, in_hw, stride):
    k = w.shape[0]
    Hin, Win = in_hw
    _, pt, _ = calls.same_pad(Hin, k, stride)
    _, pl, _ = calls.same_pad(Win, k, stride)
    wt = flip_transpose(be, w)
    return conv2d(be, [dy], [wt], None, k, 1, stride, pad=(k - 1 - pt, k - 1 - pl), out_hw=(Hin, Win))


def conv2d_wgrad(be, x, dy, k, stride, splits=1, dw0=None, beta=0.0, precision=0, dbias0=None, dbias_beta=0.0, flags=0,
                 x_bf16=False, dy_bf16=False, terms=None):
    """-> dw, or (dw, dbias) when dbias0 (initial contents of the bias-gradient buffer) is given.
    x_bf16 / dy_bf16: hand the operand over as a bf16 tensor (the bf16 BPTT tape).
    terms = (first, count): x / dy are split6 tensors [frames,H,W,6*C] / [.., 6*N] (bf16); the launch sums blocks first ..
    first + count - 1 of x against the same blocks of dy (lu_wgrad_desc.terms)."""
    if terms is not None:
        first, cnt = terms
        frames, Hin, Win, C6 = x.shape
        _, Hout, Wout, N6 = dy.shape
        Cin, N = C6 // 6, N6 // 6
        _, pt, _ = calls.same_pad(Hin, k, stride)
        _, pl, _ = calls.same_pad(Win, k, stride)
        dw = be.empty((k, k, Cin, N)) if dw0 is None else be.dev(dw0)
        xd, dyd = be.dev(bf16_bits(x), np.int16), be.dev(bf16_bits(dy), np.int16)
        db = None if dbias0 is None else be.dev(dbias0)
        d = calls.wgrad_desc(be.ptr(xd, first * Cin), Hin * Win * C6, C6, Cin, be.ptr(dyd, first * N), Hout * Wout * N6, N6, N, frames,
                             Hin, Win, Hout, Wout, k, stride, pt, pl, be.ptr(dw), Cin * N, N, splits, beta, precision=1,
                             dbias=be.ptr(db), dbias_beta=dbias_beta, x_dtype=cabi.LU_BF16, dy_dtype=cabi.LU_BF16, flags=flags,
                             terms=cnt, x_term_stride=Cin, dy_term_stride=N)
        ws = be.empty((be.lib.lu_conv2d_wgrad_workspace_bytes(C.byref(d)) // 4 + 4,))
        d.workspace = be.ptr(ws)
        calls.check(be.lib, be.lib.lu_conv2d_wgrad(C.byref(d), be.stream), 'wgrad terms')
        return be.host(dw) if db is None else (be.host(dw), be.host(db))
    frames, Hin, Win, Cin = x.shape
    _, Hout, Wout, N = dy.shape
    _, pt, _ = calls.same_pad(Hin, k, stride)
    _, pl, _ = calls.same_pad(Win, k, stride)
    dw = be.empty((k, k, Cin, N)) if dw0 is None else be.dev(dw0)
    xd = be.dev(bf16_bits(x), np.int16) if x_bf16 else be.dev(x)
    dyd = be.dev(bf16_bits(dy), np.int16) if dy_bf16 else be.dev(dy)
    db = None if dbias0 is None else be.dev(dbias0)
    d = calls.wgrad_desc(be.ptr(xd), Hin * Win * Cin, Cin, Cin, be.ptr(dyd), Hout * Wout * N, N, N, frames, Hin, Win,
                         Hout, Wout, k, stride, pt, pl, be.ptr(dw), Cin * N, N, splits, beta, precision=precision,
                         dbias=be.ptr(db), dbias_beta=dbias_beta, x_dtype=cabi.LU_BF16 if x_bf16 else cabi.LU_F32,
                         dy_dtype=cabi.LU_BF16 if dy_bf16 else cabi.LU_F32, flags=flags)
    ws = be.empty((be.lib.lu_conv2d_wgrad_workspace_bytes(C.byref(d)) // 4 + 4,))
    d.workspace = be.ptr(ws)
    calls.check(be.lib, be.lib.lu_conv2d_wgrad(C.byref(d), be.stream), 'wgrad')
    return be.host(dw) if db is None else (be.host(dw), be.host(db))


def convlstm_step_tape16(be, x_t, h, c, kernel, rec, bias, center=False, flags=0):
    """The fused bf16 ConvLSTM step as the bf16-tape engine drives it: h handed over as a bf16 tensor, h also written as
    bf16, gates written as bf16; center=True: x_t is a thin input, passed as its im2col image (lu_im2col_bf16) with the
    kernel packed as one tap (source order [h, image]).  -> h (fp32), c, gates (values of the bf16 tape), h16 (values)."""
    frames, H, W, Cin = x_t.shape
    F = rec.shape[2]
    k = rec.shape[0]
    c_out, h_out = be.empty((frames, H, W, F)), be.empty((frames, H, W, F))
    gates16, h16 = be.empty((frames, H, W, 4 * F), np.int16), be.empty((frames, H, W, F), np.int16)
    xd, cd, kd, rd, bd = [be.dev(a) for a in (x_t, c, kernel, rec, bias)]
    hd = be.dev(bf16_bits(h), np.int16)
    rp = pack_bf16(be, rd, k, F, 4 * F)
    src_h = calls.conv_src(be.ptr(hd), H * W * F, F, F, be.ptr(rp), 0, 0, dtype=cabi.LU_BF16)
    fl = flags | cabi.LU_CONV_F_GATES_BF16
    if center:
        x25 = be.empty((frames, H, W, 32), np.int16)
        calls.check(be.lib, be.lib.lu_im2col_bf16(be.ptr(xd), be.ptr(x25), frames, H, W, Cin, k, be.stream), 'im2col')
        kp = be.empty((-(-4 * F // 32) * 512 + 4,))
        calls.check(be.lib, be.lib.lu_pack_weights_taps_bf16(be.ptr(kd), 0, 4 * F, 1, k * k * Cin, 4 * F, be.ptr(kp),
                                                             be.stream), 'pack center')
        srcs = [src_h, calls.conv_src(be.ptr(x25), H * W * 32, 32, 32, be.ptr(kp), 0, 0, dtype=cabi.LU_BF16)]
        fl |= cabi.LU_CONV_F_SRC1_CENTER
    else:
        kp = pack_bf16(be, kd, k, Cin, 4 * F)
        x16 = be.dev(bf16_bits(x_t), np.int16)      # all sources of a launch share the element type
        srcs = [calls.conv_src(be.ptr(x16), H * W * Cin, Cin, Cin, be.ptr(kp), 0, 0, dtype=cabi.LU_BF16), src_h]
    p = (k - 1) // 2
    calls.conv2d(be.lib, be.stream, srcs, frames, H, W, H, W, k, 1, 1, p, p, 4 * F, be.ptr(bd), None, 0, 0,
                 lstm=(be.ptr(cd), H * W * F, be.ptr(c_out), H * W * F, be.ptr(h_out), H * W * F, be.ptr(gates16),
                       H * W * 4 * F), precision=1, flags=fl, h16=(be.ptr(h16), H * W * F))
    return be.host(h_out), be.host(c_out), bf16_values(be.host(gates16)), bf16_values(be.host(h16))


def lstm_gates_bwd_bf16(be, gates, c_prev, c_cur, dh_a, dh_b, dc_in):
    """-> dz (values of the bf16 tape after the in-place update), dc_prev."""
    frames, H, W, F = c_cur.shape
    gd = be.dev(bf16_bits(gates), np.int16)
    cp, cc, da = be.dev(c_prev), be.dev(c_cur), be.dev(dh_a)
    db = None if dh_b is None else be.dev(dh_b)
    di = None if dc_in is None else be.dev(dc_in)
    dcp = be.empty((frames, H, W, F))
    calls.check(be.lib, be.lib.lu_lstm_gates_bwd_bf16(be.ptr(gd), be.ptr(cp), be.ptr(cc), be.ptr(da), H * W * F, be.ptr(db),
                                                      be.ptr(di), be.ptr(dcp), frames, H * W, F, be.stream), 'gates bwd bf16')
    return bf16_values(be.host(gd)), be.host(dcp)


def convlstm_step_fused(be, x_t, h, c, kernel, rec, bias, precision=0, flags=0):
    frames, H, W, Cin = x_t.shape
    F = rec.shape[2]
    k = kernel.shape[0]
    c_out, h_out, gates = be.empty((frames, H, W, F)), be.empty((frames, H, W, F)), be.empty((frames, H, W, 4 * F))
    xd, hd, cd, kd, rd, bd = [be.dev(a) for a in (x_t, h, c, kernel, rec, bias)]
    srcs = [calls.conv_src(be.ptr(xd), H * W * Cin, Cin, Cin, be.ptr(kd), Cin * 4 * F, 4 * F),
            calls.conv_src(be.ptr(hd), H * W * F, F, F, be.ptr(rd), F * 4 * F, 4 * F)]
    if precision:
        pk = pack_bf16
        kp, rp = pk(be, kd, k, Cin, 4 * F), pk(be, rd, k, F, 4 * F)
        srcs = [calls.conv_src(be.ptr(xd), H * W * Cin, Cin, Cin, be.ptr(kp), 0, 0),
                calls.conv_src(be.ptr(hd), H * W * F, F, F, be.ptr(rp), 0, 0)]
    p = (k - 1) // 2
    calls.conv2d(be.lib, be.stream, srcs, frames, H, W, H, W, k, 1, 1, p, p, 4 * F, be.ptr(bd), None, 0, 0,
                 lstm=(be.ptr(cd), H * W * F, be.ptr(c_out), H * W * F, be.ptr(h_out), H * W * F, be.ptr(gates),
                       H * W * 4 * F), precision=precision, flags=flags)
    return be.host(h_out), be.host(c_out), be.host(gates)
