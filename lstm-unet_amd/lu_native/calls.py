"""Thin, pointer-level call helpers over the C ABI (descriptor packing + error checks).
Everything here works on raw addresses; ops.py feeds it torch device pointers."""
import ctypes as C

from . import cabi


class NativeError(RuntimeError):
    pass


def check(lib, rc, what):
    if rc != 0:
        msg = lib.lu_last_error()
        raise NativeError('%s failed (rc=%d): %s' % (what, rc, msg.decode() if msg else '?'))


def same_pad(n_in, k, s):
    """TF 'SAME' geometry (Conv2D padding='same', reference Networks.py:55-56)."""
    n_out = -(-n_in // s)
    total = max((n_out - 1) * s + k - n_in, 0)
    return n_out, total // 2, total - total // 2


def conv_src(x, frame_stride, pix_stride, Cin, w, w_tap_stride, w_row_stride, dtype=cabi.LU_F32):
    s = cabi.ConvSrc()
    s.x, s.w = x, w
    s.frame_stride, s.w_tap_stride = frame_stride, w_tap_stride
    s.pix_stride, s.C, s.w_row_stride = pix_stride, Cin, w_row_stride
    s.dtype = dtype
    return s


def conv_splits(frames, Hout, Wout, N, k, channels):
    """K-axis split for problems with too few 256x128 output tiles to occupy 256 CUs twice over."""
    tiles = -(-(frames * Hout * Wout) // 256) * -(-N // 128)
    n_it = k * k * -(-channels // 16)
    if tiles >= 384 or n_it < 64:
        return 1
    return int(max(1, min(512 // tiles, n_it // 32, 16)))


def conv2d(lib, stream, srcs, frames, Hin, Win, Hout, Wout, k, stride, dil, pad_t, pad_l, N, bias, out,
           out_frame_stride, out_pix_stride, lstm=None, splits=1, workspace=None, out_row_stride=0, precision=0, k_h=0,
           flags=0, h16=None):
    """h16 = (ptr, frame_stride): optional bf16 copy of h written by the fused ConvLSTM epilogue (precision 1)."""
    d = cabi.ConvDesc()
    d.n_src = len(srcs)
    for i, s in enumerate(srcs):
        d.src[i] = s
    d.frames, d.Hin, d.Win, d.Hout, d.Wout = frames, Hin, Win, Hout, Wout
    d.k, d.stride, d.dil, d.pad_t, d.pad_l, d.N = k, stride, dil, pad_t, pad_l, N
    d.bias, d.out, d.out_frame_stride, d.out_pix_stride = bias, out, out_frame_stride, out_pix_stride
    d.epilogue = cabi.LU_EPI_BIAS
    d.splits, d.workspace, d.out_row_stride, d.precision, d.k_h = splits, workspace, out_row_stride, precision, k_h
    d.flags = flags
    if h16 is not None:
        d.h16_out, d.h16_frame_stride = h16
    if lstm is not None:
        d.epilogue = cabi.LU_EPI_LSTM
        (d.c_prev, d.c_prev_frame_stride, d.c_out, d.c_out_frame_stride, d.h_out, d.h_frame_stride,
         d.gates_out, d.gates_frame_stride) = lstm
    check(lib, lib.lu_conv2d_fwd(C.byref(d), stream), 'lu_conv2d_fwd')


def wgrad_desc(x, x_fs, x_ps, Cin, dy, dy_fs, dy_ps, N, frames, Hin, Win, Hout, Wout, k, stride, pad_t, pad_l,
               dw, dw_tap_stride, dw_row_stride, splits, beta, precision=0, dbias=None, dbias_beta=0.0,
               x_dtype=cabi.LU_F32, dy_dtype=cabi.LU_F32, flags=0):
    d = cabi.WgradDesc()
    d.x, d.x_frame_stride, d.x_pix_stride, d.C = x, x_fs, x_ps, Cin
    d.dy, d.dy_frame_stride, d.dy_pix_stride, d.N = dy, dy_fs, dy_ps, N
    d.frames, d.Hin, d.Win, d.Hout, d.Wout = frames, Hin, Win, Hout, Wout
    d.k, d.stride, d.pad_t, d.pad_l = k, stride, pad_t, pad_l
    d.dw, d.dw_tap_stride, d.dw_row_stride, d.splits, d.beta = dw, dw_tap_stride, dw_row_stride, splits, beta
    d.precision = precision
    d.dbias, d.dbias_beta = dbias, dbias_beta
    d.x_dtype, d.dy_dtype, d.flags = x_dtype, dy_dtype, flags
    return d


def wgrad_splits_bf16_row(pixels, k, Cin, N, ct, rounds=5, cus=256):
    """Pixel-axis split of the bf16 kernel-row variant.  Its blocks are numbered XCD-aware (all tiles of a pixel slab on
    one XCD), one block per CU at 128-channel tiles (two at 64), all of equal length: pick a multiple of 8 slabs so that
    tiles x slabs is close to `rounds` full waves of the chip -- few fat slabs keep the slab write + re-read small
    (round 1: ~3000 blocks = 38 slabs of 26 MB at level 1 = 1 GB per launch; now 16 slabs)."""
    inner = k * -(-Cin // ct) * -(-N // 128)
    per_round = cus * (1 if ct == 128 else 2)
    s = max(8, int(round(rounds * per_round / float(inner) / 8.0)) * 8)
    while s > 8 and pixels // s < 2048:
        s -= 8
    if pixels // s < 1024:          # tiny problems (tests): any split count will do
        s = max(1, min(s, pixels // 512))
    return int(s)


def wgrad_splits(pixels, k, Cin, N, target_blocks=3072, row_variant=False, small3=False):
    """Pixel-axis split.  The wgrad kernel holds 4 workgroups per CU (1024 slots on 256 CUs); with only ~1000
    long-running blocks the slowest CU (4 blocks vs 3) sets the time, so aim for a few thousand shorter
    blocks (>= 2048 pixels = 128 pipeline stages each) and let the dispatcher balance them; beyond ~3000 the slab
    reduce grows faster than the balance improves (re-measured with the kernel-row variants: 3072 vs 6144 +1-2 %)."""
    if small3:           # all-taps kernel of the narrow layers: one block per pixel slab, ~2 blocks per CU
        return max(1, min(512, pixels // 2048))
    ct = max(1, -(-Cin // 128)) if Cin % 4 == 0 else -(-(k * k * Cin) // 32)
    taps = k * k if Cin % 4 == 0 else 1
    bn = 256 if (Cin % 4 == 0 and Cin > 64 and N >= 256 and N % 4 == 0) else 128
    tiles = taps * ct * max(1, -(-N // bn))
    if row_variant:      # kernel-row kernel: (k rows) x (64-channel tiles) x (128-column tiles)
        tiles = k * -(-Cin // 64) * -(-N // 128)
    s = max(1, min(target_blocks // max(tiles, 1), pixels // 2048))
    return max(1, min(s, 256))
