"""TEST INFRASTRUCTURE: numpy-facing wrappers over the C ABI, used with the host-emulated build of
the kernel sources (tests/emu).  Mirrors what lu_native/ops.py does with torch device tensors."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
from lu_native import cabi, calls  # noqa: E402

_LIB = None


def emu_lib():
    global _LIB
    if _LIB is None:
        sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
        import build_emu
        _LIB = cabi.bind(build_emu.build())
    return _LIB


def ptr(a):
    return None if a is None else a.ctypes.data


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def conv2d(lib, srcs, ws, bias, k, stride=1, dil=1, pad=None, out_hw=None, N=None):
    """srcs: list of [frames,H,W,C] arrays; ws: list of [k,k,C,N] arrays (or (array, tap_stride,row_stride, ptr_off))."""
    x0 = srcs[0]
    frames, Hin, Win = x0.shape[:3]
    if N is None:
        N = ws[0].shape[-1]
    if pad is None:
        Hout, pt, _ = calls.same_pad(Hin, k, stride)
        Wout, pl, _ = calls.same_pad(Win, k, stride)
    else:
        pt, pl = pad
        Hout, Wout = out_hw
    out = np.full((frames, Hout, Wout, N), np.nan, np.float32)
    cs = []
    for x, w in zip(srcs, ws):
        Cin = x.shape[3]
        cs.append(calls.conv_src(ptr(x), x.strides[0] // 4, x.strides[2] // 4, Cin, ptr(w), w.strides[1] // 4,
                                 w.strides[2] // 4))
    calls.conv2d(lib, None, cs, frames, Hin, Win, Hout, Wout, k, stride, dil, pt, pl, N, ptr(bias), ptr(out),
                 Hout * Wout * N, N)
    return out


def flip_transpose(lib, w, c_off=0, C_sub=None):
    k, _, Ct, N = w.shape
    C_sub = Ct - c_off if C_sub is None else C_sub
    wt = np.full((k, k, N, C_sub), np.nan, np.float32)
    calls.check(lib, lib.lu_weight_flip_transpose(ptr(w), ptr(wt), k, Ct, N, c_off, C_sub, None), 'flip')
    return wt


def conv2d_dgrad(lib, dy, w, in_hw, stride):
    """dX of conv2d_same(x, w, stride): conv of (zero-dilated) dy with the flipped/transposed kernel."""
    k = w.shape[0]
    Hin, Win = in_hw
    _, pt, _ = calls.same_pad(Hin, k, stride)
    _, pl, _ = calls.same_pad(Win, k, stride)
    wt = flip_transpose(lib, w)
    return conv2d(lib, [dy], [wt], None, k, 1, stride, pad=(k - 1 - pt, k - 1 - pl), out_hw=(Hin, Win))


def conv2d_wgrad(lib, x, dy, k, stride, splits=1, dw=None, beta=0.0):
    frames, Hin, Win, Cin = x.shape
    _, Hout, Wout, N = dy.shape
    _, pt, _ = calls.same_pad(Hin, k, stride)
    _, pl, _ = calls.same_pad(Win, k, stride)
    if dw is None:
        dw = np.full((k, k, Cin, N), np.nan, np.float32)
    d = calls.wgrad_desc(ptr(x), x.strides[0] // 4, x.strides[2] // 4, Cin, ptr(dy), dy.strides[0] // 4,
                         dy.strides[2] // 4, N, frames, Hin, Win, Hout, Wout, k, stride, pt, pl, ptr(dw),
                         dw.strides[1] // 4, dw.strides[2] // 4, splits, beta)
    ws = np.empty(lib.lu_conv2d_wgrad_workspace_bytes(C.byref(d)) // 4 + 4, np.float32)
    d.workspace = ptr(ws)
    calls.check(lib, lib.lu_conv2d_wgrad(C.byref(d), None), 'wgrad')
    return dw


def convlstm_step_fused(lib, x_t, h, c, kernel, rec, bias, save_gates=True):
    frames, H, W, Cin = x_t.shape
    F = rec.shape[2]
    k = kernel.shape[0]
    c_out = np.full((frames, H, W, F), np.nan, np.float32)
    h_out = np.full((frames, H, W, F), np.nan, np.float32)
    gates = np.full((frames, H, W, 4 * F), np.nan, np.float32) if save_gates else None
    srcs = [calls.conv_src(ptr(x_t), H * W * Cin, Cin, Cin, ptr(kernel), Cin * 4 * F, 4 * F),
            calls.conv_src(ptr(h), H * W * F, F, F, ptr(rec), F * 4 * F, 4 * F)]
    p = (k - 1) // 2
    calls.conv2d(lib, None, srcs, frames, H, W, H, W, k, 1, 1, p, p, 4 * F, ptr(bias), None, 0, 0,
                 lstm=(ptr(c), H * W * F, ptr(c_out), H * W * F, ptr(h_out), H * W * F, ptr(gates), H * W * 4 * F))
    return h_out, c_out, gates
