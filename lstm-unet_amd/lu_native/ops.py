"""Tensor-level wrappers over the gfx950 kernel library (C ABI in include/lstm_unet_hip.h).

torch is used for device memory, streams and torch.distributed only.  Every function here takes
CUDA(HIP) float32 tensors, hands raw pointers to the C ABI and enqueues on torch's current stream.
There is NO CPU / eager fallback: a missing library or a non-device tensor raises.
"""
import ctypes as C
import os

import torch

from . import cabi, calls
from .calls import NativeError, same_pad

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'csrc', 'liblstmunet_hip.so')
_lib = None
FUSED_MIN_TILES = None  # None: fp32 steps pick fused / K-split + gate kernel by modelled duration (calls.fused_step_cost_us);
                        # an integer: fused from that many 256x32ch tiles on (tests set 0 = always fused)
FUSED_MIN_TILES_BF16 = 0    # ... the bf16 kernel is better fused at every size (B = 1 streaming: 394 vs 384 frames/s)
EVENT_LOG = None   # bench.py: list collecting (kernel class, algorithmic FLOPs, start, end) around the MFMA launches
CONV_FLAGS = 0     # tests / A-B tools: cabi.LU_CONV_F_* kernel-variant overrides OR-ed into every lu_conv_desc
WGRAD_FLAGS = 0    # ... cabi.LU_WGRAD_F_* into every lu_wgrad_desc (the library itself never reads the environment)


class _timed(object):
    """HIP-event bracket on the launch stream (torch's current stream), active only while bench.py sets EVENT_LOG.
    `flops` is the launch's algorithmic work: FLOPs for the MFMA kernels, BYTES (kind prefixed 'hbm:') for the
    bandwidth-bound ones."""

    def __init__(self, kind, flops):
        self.kind, self.flops, self.ev = kind, flops, None

    def __enter__(self):
        if EVENT_LOG is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()
        return self

    def __exit__(self, *exc):
        if self.ev is not None:
            self.ev[1].record()
            EVENT_LOG.append((self.kind, self.flops, self.ev[0], self.ev[1]))
        return False


def lib():
    """The HIP kernel library; loud failure if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError('HIP kernel library missing: %s -- build it with `python -m lu_native.build` '
                              '(hipcc --offload-arch=gfx950); there is no CPU fallback' % LIB_PATH)
        _lib = cabi.bind(LIB_PATH)
        if _lib.lu_abi_version() != cabi.ABI_VERSION:
            raise NativeError('ABI version mismatch in %s' % LIB_PATH)
    return _lib


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """Raw handle of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object through
    three Python layers (~10 us; 31 launches per streaming frame = 0.3 ms of a 1.5 ms bf16 frame): ask the C layer directly."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NativeError('lu_native ops need device tensors (got %s); no CPU fallback exists' % t.device)
        if t.dtype not in (torch.float32, torch.float64, torch.int16, torch.bfloat16):    # int16: packed bf16 weight images
            raise NativeError('unexpected dtype %s' % t.dtype)


def _p(t):
    return None if t is None else t.data_ptr()


class PackedW(object):
    """MFMA-fragment-order bf16 image of a [k,k,C,N] kernel (lu_pack_weights_bf16) and the value of lu_conv_desc.precision that goes
    with it (1)."""
    __slots__ = ('data', 'shape', 'precision')

    def __init__(self, data, shape, precision=1):
        self.data, self.shape, self.precision = data, tuple(shape), precision


def pack_bf16(w):
    """fp32 [k,k,C,N] kernel (channel-slice views allowed) -> PackedW; fp32 stays the master copy."""
    _chk(w)
    assert w.dim() == 4 and w.stride(3) == 1 and w.stride(0) == w.shape[1] * w.stride(1)
    k, _, Cc, N = w.shape
    data = torch.empty(lib().lu_pack_weights_bf16_bytes(k, Cc, N) // 2, device=w.device, dtype=torch.int16)
    calls.check(lib(), lib().lu_pack_weights_bf16(w.data_ptr(), w.stride(1), w.stride(2), k, Cc, N, data.data_ptr(),
                                                  _stream()), 'lu_pack_weights_bf16')
    return PackedW(data, w.shape)


def pack_taps_bf16(w):
    """[k_h, k_w, C, N] dense tap window -> bf16 PackedW (the packer only sees a list of k_h*k_w taps)."""
    _chk(w)
    assert w.is_contiguous()
    kh, kw, Cc, N = w.shape
    taps = kh * kw
    data = torch.empty(taps * -(-Cc // 32) * -(-N // 32) * 1024, device=w.device, dtype=torch.int16)
    calls.check(lib(), lib().lu_pack_weights_taps_bf16(w.data_ptr(), Cc * N, N, taps, Cc, N, data.data_ptr(), _stream()),
                'lu_pack_weights_taps_bf16')
    return PackedW(data, w.shape)


def _src(x, w):
    """x: [frames,H,W,C] (channel-slice views allowed), w: [k,k,C,N] (channel-slice views allowed) or PackedW."""
    if isinstance(w, PackedW):
        # x may carry zero pad channels beyond the kernel's (thin inputs padded to 4): same number of 32-channel chunks
        assert x.dim() == 4 and x.stride(3) == 1 and x.stride(1) == x.shape[2] * x.stride(2)
        ck = 32 if w.precision == 1 else 16
        assert x.shape[3] >= w.shape[2] and -(-x.shape[3] // ck) == -(-w.shape[2] // ck), (x.shape, w.shape)
        assert x.dtype == torch.float32 or (x.dtype == torch.bfloat16 and w.precision == 1)
        return calls.conv_src(x.data_ptr(), x.stride(0), x.stride(2), x.shape[3], w.data.data_ptr(), 0, 0,
                              dtype=cabi.LU_BF16 if x.dtype == torch.bfloat16 else cabi.LU_F32)
    assert x.dtype == torch.float32, 'bf16 activations need packed bf16 weights'
    assert x.dim() == 4 and w.dim() == 4 and x.stride(3) == 1 and w.stride(3) == 1, (x.shape, x.stride(), w.shape)
    assert x.stride(1) == x.shape[2] * x.stride(2), 'rows of x must be dense'
    assert w.shape[2] == x.shape[3] and w.stride(0) == w.shape[1] * w.stride(1)
    return calls.conv_src(x.data_ptr(), x.stride(0), x.stride(2), x.shape[3], w.data_ptr(), w.stride(1), w.stride(2))


def conv_raw(pairs, frames, Hin, Win, Hout, Wout, k, stride, dil, pad_t, pad_l, N, bias, out, out_view=None, flops=None,
             k_h=0, post=None, slabs_only=False):
    """One lu_conv2d_fwd launch (bias epilogue); picks a K-split + workspace for tile-starved problems.
    out_view = (ptr, frame_stride, pix_stride, row_stride) overrides the dense addressing of `out`.
    post = (scale, shift, alpha): inference BatchNorm affine + LeakyReLU folded into the store / the slab reduce.
    slabs_only: return (workspace, splits) with the partial slabs instead of reducing them (None when no split applies)."""
    channels = sum(x.shape[3] for x, _ in pairs)
    prec = max([w.precision for _, w in pairs if isinstance(w, PackedW)] + [0])
    narrow = prec == 1 and N in (32, 64) and not (CONV_FLAGS & cabi.LU_CONV_F_NO_NARROW)      # narrow blocks of the bf16 halo kernel
    halo = stride == 1 and dil == 1 and k in (3, 5) and (N > 64 or narrow) and N % 4 == 0     # mirrors lu_conv2d_fwd's kernel choice
    extra_flags = 0
    if k_h:
        splits = 1
    elif prec == 0:
        # fp32: split count AND kernel (halo patches vs flattened pixel rows) by modelled duration
        splits, use_halo, _ = calls.conv_plan(frames, Hout, Wout, N, k, channels, halo and out_view is None)
        if halo and out_view is None and not use_halo:
            extra_flags = cabi.LU_CONV_F_NO_HALO
            halo = False
    else:
        splits = calls.conv_splits(frames, Hout, Wout, N, k, channels, halo and out_view is None)
    ws = None
    if splits > 1:
        ws = torch.empty(splits * frames * Hout * Wout * N, device=pairs[0][0].device, dtype=torch.float32)
    if slabs_only and splits <= 1:
        return None
    bf16 = prec == 1
    kind = ('conv_halo_kernel<%d,LU_EPI_BIAS> (recurrent / input dgrads, plain convs)' % k) if halo else \
        'conv_fwd_kernel (strided / dilated / narrow convs)'
    if bf16:
        kind = ('conv_halo_frag_kernel<%d,LU_EPI_BIAS,*,bf16> (bf16-MFMA recurrent / input dgrads, plain convs)' % k) \
            if (halo and out_view is None) else 'conv_gather_bf16_kernel (bf16-MFMA strided / narrow / parity-plane convs)'
    if slabs_only:
        optr, ofs, ops_, ors = None, Hout * Wout * N, N, 0
    else:
        optr, ofs, ops_, ors = out_view if out_view is not None else (out.data_ptr(), out.stride(0), out.stride(2), 0)
    if out_view is not None and not bf16:
        kind = 'conv_fwd_kernel (strided / dilated / narrow convs)'
    with _timed(kind, flops if flops is not None else 2.0 * k * k * channels * N * frames * Hout * Wout / (dil * dil)):
        calls.conv2d(lib(), _stream(), [_src(x, w) for x, w in pairs], frames, Hin, Win, Hout, Wout, k, stride, dil,
                     pad_t, pad_l, N, _p(bias), optr, ofs, ops_, splits=splits, workspace=_p(ws), out_row_stride=ors,
                     precision=prec, k_h=k_h, flags=CONV_FLAGS | extra_flags | (cabi.LU_CONV_F_SLABS_ONLY if slabs_only else 0),
                     post=None if post is None else (post[0].data_ptr(), post[1].data_ptr(), float(post[2])))
    return (ws, splits) if slabs_only else out


def conv2d(pairs, bias, stride=1, out=None, post=None):
    """SAME convolution summed over (activation, weight) pairs -> [frames,Ho,Wo,N].
    post = (scale, shift, alpha): -> lrelu(scale * conv + shift), the inference BN + LeakyReLU of the conv unit in the same pass."""
    x0, w0 = pairs[0]
    _chk(bias, out, *[t.data if isinstance(t, PackedW) else t for p in pairs for t in p])
    if post is not None:
        _chk(post[0], post[1])
        assert post[0].numel() == w0.shape[3] == post[1].numel() and post[0].dtype == post[1].dtype == torch.float32
    frames, Hin, Win = x0.shape[:3]
    k, N = w0.shape[0], w0.shape[3]
    Hout, pt, _ = same_pad(Hin, k, stride)
    Wout, pl, _ = same_pad(Win, k, stride)
    if out is None:
        out = torch.empty((frames, Hout, Wout, N), device=x0.device, dtype=torch.float32)
    return conv_raw(pairs, frames, Hin, Win, Hout, Wout, k, stride, 1, pt, pl, N, bias, out, post=post)


def conv2d_s2_fwd_bf16(x16, pw, bias):
    """Stride-2 3x3 SAME convolution of a bf16 tensor (even extents) with bf16-packed weights -> fp32 [frames, H/2, W/2, N]."""
    _chk(x16, pw.data, bias)
    assert x16.dtype == torch.bfloat16 and x16.dim() == 4 and x16.stride(3) == 1 and x16.stride(1) == x16.shape[2] * x16.stride(2)
    frames, H, W, Cc = x16.shape
    k, _, _, N = pw.shape
    assert k == 3 and H % 2 == 0 and W % 2 == 0 and pw.shape[2] == Cc
    out = torch.empty((frames, H // 2, W // 2, N), device=x16.device, dtype=torch.float32)
    with _timed('conv_s2_fwd_bf16_kernel (bf16-MFMA stride-2 forward convolution on the bf16 ConvLSTM output)',
                2.0 * 9 * Cc * N * frames * (H // 2) * (W // 2)):
        calls.check(lib(), lib().lu_conv2d_s2_fwd_bf16(x16.data_ptr(), x16.stride(0), x16.stride(2), pw.data.data_ptr(), _p(bias),
                                                       frames, H, W, Cc, N, out.data_ptr(), _stream()), 'lu_conv2d_s2_fwd_bf16')
    return out


def flip_transpose(w, c_off=0, c_sub=None):
    """dense [k,k,C,N] kernel, channels [c_off, c_off+c_sub) -> dense [k,k,N,c_sub] kernel of the
    input-gradient convolution (spatially flipped, channel-transposed)."""
    _chk(w)
    assert w.is_contiguous()
    k, _, Ctot, N = w.shape
    c_sub = Ctot - c_off if c_sub is None else c_sub
    wt = torch.empty((k, k, N, c_sub), device=w.device, dtype=torch.float32)
    calls.check(lib(), lib().lu_weight_flip_transpose(w.data_ptr(), wt.data_ptr(), k, Ctot, N, c_off, c_sub,
                                                      _stream()), 'lu_weight_flip_transpose')
    return wt


def conv2d_dgrad(dy, w, in_hw, stride, c_off=0, c_sub=None, out=None, bf16=False, bank=None):
    """Gradient w.r.t. (channels [c_off, c_off+c_sub) of) the input of conv2d(x, w, stride): a
    convolution of the (zero-dilated when stride == 2) dy with the flipped / transposed kernel."""
    _chk(dy, w)
    k = w.shape[0]
    Hin, Win = in_hw
    _, pt, _ = same_pad(Hin, k, stride)
    _, pl, _ = same_pad(Win, k, stride)
    bf16 = bf16 and dy.shape[3] % 4 == 0 and dy.is_contiguous()      # the bf16 kernels read 16-byte channel groups
    assert dy.dtype == torch.float32 or (bf16 and stride == 1), 'a bf16 dy needs the bf16 halo kernel'
    _chk(dy, w)
    if stride == 2 and c_off == 0 and (c_sub is None or c_sub == w.shape[2]) and out is None and k > 1:
        return _conv2d_dgrad_stride2(dy, w, Hin, Win, pt, pl, bf16)
    # bank (lu_native/wbank.py): the flipped (+ packed) kernel is kept across steps and refreshed in a batch by its owner
    wt = flip_transpose(w, c_off, c_sub) if bank is None else bank.flip(w, c_off, c_sub)
    frames, Hd, Wd, N = dy.shape
    Cs = wt.shape[3]
    if bf16 and stride == 1:      # (the zero-dilated form of a stride-2 layer's gradient has no bf16 kernel)
        wt = pack_bf16(wt) if bank is None else bank.pack(wt)
    if out is None:
        out = torch.empty((frames, Hin, Win, Cs), device=dy.device, dtype=torch.float32)
    return conv_raw([(dy, wt)], frames, Hd, Wd, Hin, Win, k, 1, stride, k - 1 - pt, k - 1 - pl, Cs, None, out)


def _conv2d_dgrad_stride2(dy, w, Hin, Win, pt, pl, bf16=False):
    """Input gradient of a stride-2 convolution as four stride-1 convolutions of dy, one per output parity class,
    each written in place into its (2a+py, 2b+px) plane -- no multiplications by the zeros of a dilated dy."""
    assert w.is_contiguous() and dy.is_contiguous()
    k, _, Cc, N = w.shape
    frames, Hd, Wd, _ = dy.shape
    ks = (k + 1) // 2

    def axis(par, pad):      # taps kh of parity class `par`: dY row = a + o, o = (par + pad - kh) / 2
        offs = [(par + pad - kh) // 2 for kh in range(k) if (par + pad - kh) % 2 == 0]
        return -min(offs), len(offs)

    (py0, ny0), (py1, ny1) = axis(0, pt), axis(1, pt)
    (px0, nx0), (px1, nx1) = axis(0, pl), axis(1, pl)
    sub = torch.empty(((ny0 + ny1) * (nx0 + nx1), N, Cc), device=w.device, dtype=torch.float32)    # four compact planes
    calls.check(lib(), lib().lu_stride2_dgrad_weights(w.data_ptr(), sub.data_ptr(), k, ks, Cc, N, pt, pl, py0, py1, px0,
                                                      px1, _stream()), 'lu_stride2_dgrad_weights')
    out = torch.empty((frames, Hin, Win, Cc), device=dy.device, dtype=torch.float32)
    if (bf16 and k == 3 and pt == 0 and pl == 0 and Hin == 2 * Hd and Win == 2 * Wd and N % 32 == 0 and
            not (CONV_FLAGS & cabi.LU_CONV_F_NO_NARROW)):
        # bf16 mode, even extents: all four parity classes in ONE launch (conv_s2_dgrad_bf16_kernel keeps the four classes'
        # accumulators of a dy tile; the nine tap matrices of `sub` are exactly its (class, tap) order)
        packed = pack_taps_bf16(sub.view(9, 1, N, Cc))
        with _timed('conv_s2_dgrad_bf16_kernel (bf16-MFMA stride-2 input gradient, four parity classes per block)',
                    2.0 * 9 * N * Cc * frames * Hd * Wd):
            calls.check(lib(), lib().lu_conv2d_s2_dgrad_bf16(dy.data_ptr(), dy.stride(0), dy.stride(2), packed.data.data_ptr(),
                                                             frames, Hd, Wd, N, Cc, out.data_ptr(), _stream()),
                        'lu_conv2d_s2_dgrad_bf16')
        return out
    off = 0
    for py, (pady, ny) in enumerate(((py0, ny0), (py1, ny1))):
        for px, (padx, nx) in enumerate(((px0, nx0), (px1, nx1))):
            wsub = sub[off:off + ny * nx].view(ny, nx, N, Cc)         # the plane's ny x nx tap window
            off += ny * nx
            Hs, Ws = (Hin - py + 1) // 2, (Win - px + 1) // 2
            if Hs <= 0 or Ws <= 0:
                continue
            if ny * nx == 0:          # (cannot happen for k >= 2; a class without taps has zero gradient)
                out[:, py::2, px::2].zero_()
                continue
            view = (out.data_ptr() + 4 * (py * Win + px) * Cc, Hin * Win * Cc, 2 * Cc, 2 * Win * Cc)
            if bf16:
                wsub = pack_taps_bf16(wsub)
            conv_raw([(dy, wsub)], frames, Hd, Wd, Hs, Ws, nx, 1, 1, pady, padx, Cc, None, out, out_view=view,
                     flops=2.0 * ny * nx * N * Cc * frames * Hs * Ws, k_h=ny)
    return out


def bf16_row_wgrad_ok(x, dy, k, stride):
    """True when lu_conv2d_wgrad takes its bf16 kernel-row variant for these operands (mirrors its kernel choice)."""
    frames, Hin, Win, Cin = x.shape
    _, Hout, Wout, N = dy.shape

    def vec(t, ch):
        q = 8 if t.dtype == torch.bfloat16 else 4
        return ch % q == 0 and t.stride(2) % q == 0 and t.stride(0) % q == 0 and t.data_ptr() % 16 == 0

    narrow = not (WGRAD_FLAGS & cabi.LU_WGRAD_F_NO_NARROW_BF16)      # C >= 32: masked tiles of the same kernel (decoder tail)
    small3 = k == 3 and Cin <= 64 and N <= 64 and x.dtype == torch.float32 and dy.dtype == torch.float32 and not narrow
    shape_ok = (k in (3, 5) and Cin >= (32 if narrow else 64) and not small3) or (k == 1 and Cin >= 32)
    if stride == 2:      # the stride-2 3x3 layers (first convolution of a down block)
        return (k in (3, 5) and Cin >= 64 and Wout % 32 == 0 and vec(x, Cin) and vec(dy, N) and Hout == (Hin + 1) // 2 and
                Wout == (Win + 1) // 2 and not (WGRAD_FLAGS & cabi.LU_WGRAD_F_NO_ROW))
    return (stride == 1 and shape_ok and Wout % 32 == 0 and vec(x, Cin) and vec(dy, N) and
            Hout == Hin and Wout == Win and not (WGRAD_FLAGS & cabi.LU_WGRAD_F_NO_ROW))


X3_PIECES = True      # precision 'bf16x3': the piece-aware weight-gradient kernel where it applies (False: two launches with the terms as frames, the round-5 form --
                      # what layers outside x3_pieces_ok still run; tools/wgbench_x3.py and the round's A/B set it from outside)


def x3_pieces_ok(x6, dy6, k):
    """lu_wgrad_desc.terms == 6 applies: split6 tensors on the device, stride-1 3x3 / 5x5, 128-channel tiles, 32-pixel runs inside a row."""
    return (X3_PIECES and x6.dtype == dy6.dtype == torch.bfloat16 and k in (3, 5) and x6.shape[3] % 6 == 0 and dy6.shape[3] % 6 == 0 and
            (x6.shape[3] // 6) % 128 == 0 and (dy6.shape[3] // 6) % 8 == 0 and x6.shape[2] % 32 == 0 and x6.shape[1:3] == dy6.shape[1:3] and
            x6.stride(2) % 8 == 0 and dy6.stride(2) % 8 == 0 and x6.stride(0) % 8 == 0 and dy6.stride(0) % 8 == 0 and
            x6.data_ptr() % 16 == 0 and dy6.data_ptr() % 16 == 0 and not (WGRAD_FLAGS & cabi.LU_WGRAD_F_NO_ROW))


def conv2d_wgrad(x, dy, dw, stride, beta=0.0, bf16=False, dbias=None, dbias_beta=0.0, terms=None, pieces=False):
    """dw ([k,k,C,N], may be a channel-slice view of a wider kernel gradient) = x (*) dy.
    dbias (optional [N]): the layer's bias gradient = column sums of dy -- summed on the side by the kernel-row wgrad
    kernels where they apply, by a separate lu_colsum pass elsewhere.
    x / dy may be bf16 tensors (the bf16 BPTT tape) when the bf16 kernel-row variant applies (bf16_row_wgrad_ok).
    terms = (first block, count) with x / dy split6 tensors (x in order A, dy in order B; precision 'bf16x3'): ONE launch sums
    block t of x against block t of dy for the `count` blocks from `first` on -- lu_wgrad_desc.terms; dbias then is the column sum
    of those dy blocks (the first three of order B are hi, mid, lo: exactly dy).  terms = (0, 6) with pieces=True: all six products
    in one pass of the piece-aware kernel (x3_pieces_ok, LU_WGRAD_F_PIECES3; round 6)."""
    _chk(x, dy, dw, dbias)
    n_terms, xts, yts = 0, 0, 0
    if terms is not None:
        first, n_terms = terms
        assert x.dtype == dy.dtype == torch.bfloat16 and x.shape[3] % 6 == 0 and dy.shape[3] % 6 == 0 and first + n_terms <= 6
        # the bias gradient is the column sum of dy = hi + mid + lo: blocks 0..2 of order B, each exactly once
        assert dbias is None or (first, n_terms) == (0, 3) or (pieces and (first, n_terms) == (0, 6)), \
            'dbias rides on terms (0, 3) or the piece-aware (0, 6) only'
        xts, yts = x.shape[3] // 6, dy.shape[3] // 6
        assert not pieces or ((first, n_terms) == (0, 6) and x3_pieces_ok(x, dy, dw.shape[0])), 'pieces: terms (0, 6) where x3_pieces_ok'
        x, dy = x[..., first * xts:(first + 1) * xts], dy[..., first * yts:(first + 1) * yts]
    frames, Hin, Win, Cin = x.shape
    _, Hout, Wout, N = dy.shape
    k = dw.shape[0]
    _, pt, _ = same_pad(Hin, k, stride)
    _, pl, _ = same_pad(Win, k, stride)
    xb, yb = x.dtype == torch.bfloat16, dy.dtype == torch.bfloat16
    aligned = (Cin % 4 == 0 and N % 4 == 0 and x.stride(2) % 4 == 0 and dy.stride(2) % 4 == 0 and x.stride(0) % 4 == 0 and
               dy.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0)
    ragged_w = (Wout % 16 != 0 and Wout >= 40 and not xb and not yb and not (WGRAD_FLAGS & cabi.LU_WGRAD_F_NO_RAGGED))
    row_variant = (aligned and stride == 1 and k in (3, 5) and (Wout % 16 == 0 or ragged_w) and Cin >= 64 and Hout == Hin and
                   Wout == Win and not (WGRAD_FLAGS & cabi.LU_WGRAD_F_NO_ROW))     # mirrors lu_conv2d_wgrad's kernel choice
    small3 = (aligned and not xb and not yb and stride == 1 and k == 3 and Cin <= 64 and N <= 64 and Wout % 16 == 0 and
              Hout == Hin and Wout == Win and
              not (WGRAD_FLAGS & cabi.LU_WGRAD_F_NO_SMALL3))      # all-taps kernel of the narrow decoder layers (takes precedence)
    bf16_row = bf16 and bf16_row_wgrad_ok(x, dy, k, stride)
    if bf16_row and stride == 1 and k in (3, 5):
        small3, row_variant = False, True      # (bf16 mode: the narrow layers ride the bf16 kernel-row variant as well)
    if small3:
        row_variant = False
    assert bf16_row or not (xb or yb), 'bf16 operands need the bf16 kernel-row weight gradient'
    if bf16_row and stride == 2:
        row_variant = True           # (the bias gradient rides on this launch as well)
    if bf16_row:
        ct = 64 if (k == 1 or (stride == 2 and k == 5) or WGRAD_FLAGS & cabi.LU_WGRAD_F_CT64) else 128 if (WGRAD_FLAGS & cabi.LU_WGRAD_F_CT128) else \
            (128 if (Cin % 128 == 0 or Cin > 256) else 64)
        all_taps = (k == 3 and stride == 1 and Cin >= 64 and      # mirrors lu_conv2d_wgrad: the all-taps form of the 3x3 layers
                    not (WGRAD_FLAGS & (cabi.LU_WGRAD_F_NO_TAPS9 | cabi.LU_WGRAD_F_CT64 | cabi.LU_WGRAD_F_CT128)))
        if pieces:      # piece-aware kernel: 128-channel kernel-row tiles; a pixel carries six products
            ct, all_taps = 128, False
        splits = calls.wgrad_splits_bf16_row(frames * max(1, n_terms) * Hout * Wout, k, Cin, N, ct, all_taps=all_taps)
    else:
        splits = calls.wgrad_splits(frames * Hout * Wout, k, Cin, N, row_variant=row_variant, small3=small3,
                                    target_blocks=3072)      # re-measured after the kernel-row variants: ~3000 blocks >= 6000
    d = calls.wgrad_desc(x.data_ptr(), x.stride(0), x.stride(2), Cin, dy.data_ptr(), dy.stride(0), dy.stride(2), N,
                         frames, Hin, Win, Hout, Wout, k, stride, pt, pl, dw.data_ptr(), dw.stride(1), dw.stride(2),
                         splits, beta, precision=1 if bf16 else 0,
                         dbias=dbias.data_ptr() if (dbias is not None and (row_variant or small3)) else None,
                         dbias_beta=dbias_beta, x_dtype=cabi.LU_BF16 if xb else cabi.LU_F32,
                         dy_dtype=cabi.LU_BF16 if yb else cabi.LU_F32, flags=WGRAD_FLAGS | (cabi.LU_WGRAD_F_PIECES3 if pieces else 0), terms=n_terms, x_term_stride=xts,
                         dy_term_stride=yts)
    assert n_terms == 0 or (bf16_row and stride == 1 and k in (3, 5)), 'terms: the bf16 kernel-row weight gradient, stride 1'
    if dbias is not None and not (row_variant or small3):
        bias_grad(dy, dbias, dbias_beta)
    nbytes = lib().lu_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
    d.workspace = ws.data_ptr()
    kind = ('wgrad_row_kernel<%d> (weight gradients hoisted over T)' % k) if row_variant else \
        ('wgrad_small3_kernel (narrow 3x3 layers, all nine taps per block)' if small3 else
         'wgrad_kernel (strided / thin / narrow layers)')
    if bf16_row:
        kind = 'wgrad_row_bf16_kernel<%d> (bf16-MFMA weight gradients hoisted over T)' % k
    if pieces:
        kind = 'wgrad_row_x3_kernel<%d> (bf16-MFMA weight gradients on the three pieces of the split operands, hoisted over T)' % k
    if EVENT_LOG is not None:      # bench.py's roofline pass: time the MFMA kernel alone, the slab reduce outside the bracket
        d.phase = 1
        with _timed(kind, 2.0 * k * k * Cin * N * frames * max(1, n_terms) * Hout * Wout):
            calls.check(lib(), lib().lu_conv2d_wgrad(C.byref(d), _stream()), 'lu_conv2d_wgrad')
        d.phase = 2
    calls.check(lib(), lib().lu_conv2d_wgrad(C.byref(d), _stream()), 'lu_conv2d_wgrad')
    return dw


def convlstm_step(x_t, h_prev, c_prev, kernel, rec, bias, h_out, c_out, gates_out, h16_out=None, x_center=False, h6_out=None):
    """One ConvLSTM2D cell step (reference Networks.py:48-50,62-63).  Fused two-source conv + gate
    epilogue when F % 32 == 0, otherwise conv -> pre-activations -> gate kernel.
    bf16 mode extras (fused bf16 kernel only): h_prev may be the bf16 copy of the previous hidden state, h16_out receives
    the bf16 copy of the new one, gates_out may be a bf16 tensor (the bf16 BPTT tape), and with x_center=True x_t is the
    im2col image of a thin input (ops.im2col_bf16) whose kernel was packed as ONE tap.
    precision 'bf16x3' (fused bf16 kernel on split6 operands): h6_out receives the split6 image of the new hidden state
    ([frames,H,W,6F] bf16, LU_CONV_F_H16_SPLIT) -- the next step's recurrent operand, written by the gate epilogue."""
    packed = isinstance(kernel, PackedW)
    bf16 = packed and kernel.precision == 1
    _chk(x_t, h_prev, c_prev, kernel.data if packed else kernel, rec.data if packed else rec, bias, h_out, c_out, gates_out,
         h16_out, h6_out)
    frames, H, W, _ = x_t.shape
    F = rec.shape[3] // 4      # (not rec.shape[2]: precision 'bf16x3' lays the recurrent kernel's rows out six times)
    k = rec.shape[0]
    p = (k - 1) // 2
    # The fused epilogue cannot take a K split, so tile-starved steps (streaming inference: B = 1) run the conv with
    # a split into pre-activations and the stand-alone gate kernel instead.
    if fused_step_applies(frames, H, W, F, bf16, k, x_t.shape[3] + h_prev.shape[3]):
        flags = CONV_FLAGS
        if gates_out is not None and gates_out.dtype == torch.bfloat16:
            flags |= cabi.LU_CONV_F_GATES_BF16
        if h6_out is not None:
            assert bf16 and h16_out is None and h6_out.dtype == torch.bfloat16 and h6_out.is_contiguous() and \
                h6_out.shape == (frames, H, W, 6 * F)
            flags |= cabi.LU_CONV_F_H16_SPLIT
            h16_out = h6_out
        cin_flops = kernel.shape[2] * (kernel.shape[0] * kernel.shape[1]) / float(k * k)
        if x_center:       # the hidden state goes first: the centre-tap image chunk is the last pipeline stage
            flags |= cabi.LU_CONV_F_SRC1_CENTER
            srcs = [_src(h_prev, rec), _src(x_t, kernel)]
        else:
            srcs = [_src(x_t, kernel), _src(h_prev, rec)]
        with _timed(('conv_halo_frag_kernel<%d,LU_EPI_LSTM,*,bf16> (fused bf16-MFMA ConvLSTM step)' % k) if bf16 else
                    'conv_halo_kernel<%d,LU_EPI_LSTM> (fused ConvLSTM step: two-source implicit GEMM + gate epilogue)' % k,
                    2.0 * k * k * (cin_flops + rec.shape[2]) * 4 * F * frames * H * W):
            calls.conv2d(lib(), _stream(), srcs, frames, H, W, H, W, k, 1, 1, p, p,
                         4 * F, _p(bias), None, 0, 0,
                         lstm=(c_prev.data_ptr(), c_prev.stride(0), c_out.data_ptr(), c_out.stride(0), h_out.data_ptr(),
                               h_out.stride(0), _p(gates_out), gates_out.stride(0) if gates_out is not None else 0),
                         precision=kernel.precision if packed else 0, flags=flags,
                         h16=None if h16_out is None else (h16_out.data_ptr(), h16_out.stride(0)))
    else:
        assert h16_out is None and h6_out is None and not x_center and h_prev.dtype == torch.float32 and \
            (gates_out is None or gates_out.dtype == torch.float32), 'the bf16 tape belongs to the fused bf16 step'
        assert c_prev.is_contiguous() and c_out.is_contiguous()
        # K-split steps hand their partial slabs straight to the gate kernel (one pass over z less than reduce + gates)
        slabs = conv_raw([(x_t, kernel), (h_prev, rec)], frames, H, W, H, W, k, 1, 1, p, p, 4 * F, None, None, slabs_only=True)
        if slabs is not None:
            calls.check(lib(), lib().lu_lstm_gates_fwd_slabs(slabs[0].data_ptr(), slabs[1], _p(bias), c_prev.data_ptr(),
                                                             c_out.data_ptr(), h_out.data_ptr(), _p(gates_out), frames,
                                                             H * W, F, h_out.stride(0), _stream()), 'lu_lstm_gates_fwd_slabs')
            return
        z = conv2d([(x_t, kernel), (h_prev, rec)], bias, 1)
        calls.check(lib(), lib().lu_lstm_gates_fwd(z.data_ptr(), c_prev.data_ptr(), c_out.data_ptr(), h_out.data_ptr(),
                                                   _p(gates_out), frames, H * W, F, h_out.stride(0), _stream()),
                    'lu_lstm_gates_fwd')


def fused_step_applies(frames, H, W, F, bf16, k=5, channels=None):
    """True when convlstm_step takes the fused kernel for this shape.  The fused epilogue cannot take a K split, so
    tile-starved fp32 steps (streaming inference: B = 1) may be faster as K-split conv + slab reduce + gate kernel."""
    if F % 32:
        return False
    tiles = -(-(frames * H * W) // 256) * (F // 32)
    if bf16:
        return tiles >= FUSED_MIN_TILES_BF16
    if FUSED_MIN_TILES is not None:
        return tiles >= FUSED_MIN_TILES
    if channels is None or tiles > 2048:
        return True
    unfused = calls.conv_plan(frames, H, W, 4 * F, k, channels)[2] + frames * H * W * F * 4 * 8 / 4e6 + 5
    return 0.95 * calls.fused_step_cost_us(frames, H, W, F, k, channels) <= unfused      # (margin: model error)


def to_bf16(x, out=None):
    """fp32 tensor -> bf16 tensor (round to nearest even), lu_convert_f32_bf16."""
    _chk(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    assert out.is_contiguous() and out.numel() == x.numel()
    calls.check(lib(), lib().lu_convert_f32_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), 'lu_convert_f32_bf16')
    return out


def to_f32(x, out=None):
    _chk(x)
    assert x.is_contiguous() and x.dtype == torch.bfloat16
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    calls.check(lib(), lib().lu_convert_bf16_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), 'lu_convert_bf16_f32')
    return out


def split6(x, lp=None, out=None, order=0):
    """fp32 activations [frames,H,W,C] -> the bf16 tensor [frames,H,W,6*lp] of precision 'bf16x3' (lu_split6; order 0 = A: blocks lo,
    mid, hi, mid, hi, hi of the exact three-way bf16 split x = hi + mid + lo; order 1 = B: hi, mid, lo, hi, mid, hi; channels
    [C, lp) of a block are zero).  Against weights laid out by pack_split6_bf16 in the OTHER order a bf16 convolution over the
    6*lp channels IS the fp32 convolution to 2^-26 per product; block t of an order-A tensor against block t of an order-B tensor
    is term t of ops.SPLIT_TERMS (conv2d_wgrad(..., terms=...))."""
    _chk(x, out)
    assert x.dtype == torch.float32 and x.dim() == 4 and x.stride(3) == 1 and x.stride(1) == x.shape[2] * x.stride(2) and \
        x.stride(0) == x.shape[1] * x.stride(1), 'split6: dense pixel rows'
    frames, H, W, Cc = x.shape
    lp = Cc if lp is None else lp
    if out is None:
        out = torch.empty((frames, H, W, 6 * lp), device=x.device, dtype=torch.bfloat16)
    assert out.is_contiguous() and out.dtype == torch.bfloat16 and out.shape == (frames, H, W, 6 * lp)
    with _timed('hbm:split6_kernel (bf16x3: three-way bf16 split of an activation, 1 fp32 read + 6 bf16 writes)',
                (4.0 * Cc + 12.0 * lp) * frames * H * W):
        calls.check(lib(), lib().lu_split6(x.data_ptr(), frames * H * W, Cc, x.stride(2), out.data_ptr(), 6 * lp, lp, int(order),
                                           cabi.LU_BF16, _stream()), 'lu_split6')
    return out


SPLIT_A = {'lo': 0, 'mid': 1, 'hi': 2}      # first block of an order-A tensor that holds each piece (order B: hi 0, mid 1, lo 2)
# the six products of precision 'bf16x3' as (piece of x, piece of dy), smallest first -- the order the channel blocks of a
# split6 / pack_split6_bf16 pair run through them
SPLIT_TERMS = (('lo', 'hi'), ('mid', 'mid'), ('hi', 'lo'), ('mid', 'hi'), ('hi', 'mid'), ('hi', 'hi'))


def split_piece(x6, piece):
    """Channel-slice VIEW [frames,H,W,lp] of one piece ('hi', 'mid', 'lo') of a split6 tensor."""
    lp = x6.shape[3] // 6
    b = SPLIT_A[piece]
    return x6[..., b * lp:(b + 1) * lp]


def pack_split6_bf16(w, cp=None, order=1):
    """fp32 kernel [k,k,C,N] (contiguous) -> the bf16 PackedW of its three-way split laid out six times along C ([k,k,6*cp,N], block
    order B by default; rows [C, cp) of a block zero): lu_split6 of the kernel along its row axis + lu_pack_weights_bf16 in one pass
    (lu_pack_weights_split6_bf16; channel-slice views allowed)."""
    _chk(w)
    assert w.dtype == torch.float32 and w.dim() == 4 and w.stride(3) == 1 and w.stride(0) == w.shape[1] * w.stride(1)
    k, _, Cc, N = w.shape
    cp = Cc if cp is None else cp
    data = torch.empty(lib().lu_pack_weights_bf16_bytes(k, 6 * cp, N) // 2, device=w.device, dtype=torch.int16)
    calls.check(lib(), lib().lu_pack_weights_split6_bf16(w.data_ptr(), w.stride(1), w.stride(2), k, Cc, cp, N, int(order), data.data_ptr(),
                                                         _stream()), 'lu_pack_weights_split6_bf16')
    return PackedW(data, (k, k, 6 * cp, N))


def im2col_bf16(x, k):
    """[frames,H,W,C] fp32 thin input (k*k*C <= 32) -> [frames,H,W,32] bf16 im2col image (lu_im2col_bf16)."""
    _chk(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    frames, H, W, Cc = x.shape
    y = torch.empty((frames, H, W, 32), device=x.device, dtype=torch.bfloat16)
    calls.check(lib(), lib().lu_im2col_bf16(x.data_ptr(), y.data_ptr(), frames, H, W, Cc, k, _stream()), 'lu_im2col_bf16')
    return y


def pack_center_bf16(w):
    """[k,k,C,N] kernel of a thin input (k*k*C <= 32) -> PackedW of ONE tap whose 'channels' are the k*k*C (tap, c) rows,
    matching im2col_bf16's channel order."""
    _chk(w)
    assert w.is_contiguous()
    k, _, Cc, N = w.shape
    rows = k * k * Cc
    assert rows <= 32
    data = torch.empty(-(-N // 32) * 1024, device=w.device, dtype=torch.int16)
    calls.check(lib(), lib().lu_pack_weights_taps_bf16(w.data_ptr(), 0, N, 1, rows, N, data.data_ptr(), _stream()),
                'lu_pack_weights_taps_bf16')
    return PackedW(data, (1, 1, rows, N))


def lstm_gates_bwd(gates, c_prev, c_cur, dh_a, dh_b, dc_in, dz, dc_prev_out):
    _chk(gates, c_prev, c_cur, dh_a, dh_b, dc_in, dz, dc_prev_out)
    frames, H, W, F = c_cur.shape
    for t in (gates, c_prev, c_cur, dz, dc_prev_out):
        assert t.is_contiguous()
    n_in = 4 + 2 + 1 + (dh_b is not None) + (dc_in is not None)       # gates, c_prev, c_cur, dh (+ dh_rec) (+ dc)
    with _timed('hbm:lstm_gates_bwd_kernel (BPTT gate backward: dz in place of the saved gates, dc)',
                4.0 * (n_in + 4 + 1) * F * frames * H * W):
        calls.check(lib(), lib().lu_lstm_gates_bwd(gates.data_ptr(), c_prev.data_ptr(), c_cur.data_ptr(), dh_a.data_ptr(),
                                                   dh_a.stride(0), _p(dh_b), _p(dc_in), dz.data_ptr(),
                                                   dc_prev_out.data_ptr(), frames, H * W, F, _stream()),
                    'lu_lstm_gates_bwd')


def lstm_gates_bwd_split(gates_dz, c_prev, c_cur, dh_a, dh_b, dc_in, dz6, dc_prev_out):
    """lstm_gates_bwd with dz written IN PLACE of the fp32 gates and, in the same pass, its split6 image (order B) into dz6
    [frames,H,W,24F] bf16 -- precision 'bf16x3' (lu_lstm_gates_bwd_split)."""
    _chk(gates_dz, c_prev, c_cur, dh_a, dh_b, dc_in, dz6, dc_prev_out)
    frames, H, W, F = c_cur.shape
    for t in (gates_dz, c_prev, c_cur, dz6, dc_prev_out):
        assert t.is_contiguous()
    assert dz6.dtype == torch.bfloat16 and dz6.shape == (frames, H, W, 24 * F) and gates_dz.dtype == torch.float32
    n_in = 4 + 2 + 1 + (dh_b is not None) + (dc_in is not None)
    with _timed('hbm:lstm_gates_bwd_split_kernel (bf16x3 gate backward: dz in place of the gates + its split image, dc)',
                (4.0 * (n_in + 4 + 1) + 2.0 * 24) * F * frames * H * W):
        calls.check(lib(), lib().lu_lstm_gates_bwd_split(gates_dz.data_ptr(), c_prev.data_ptr(), c_cur.data_ptr(), dh_a.data_ptr(),
                                                         dh_a.stride(0), _p(dh_b), _p(dc_in), dz6.data_ptr(),
                                                         dc_prev_out.data_ptr(), frames, H * W, F, _stream()),
                    'lu_lstm_gates_bwd_split')


def lstm_gates_bwd_bf16(gates_dz, c_prev, c_cur, dh_a, dh_b, dc_in, dc_prev_out):
    """BPTT gate backward on the bf16 tape: gates_dz holds the saved gates (bf16) and receives dz (bf16) in place."""
    _chk(gates_dz, c_prev, c_cur, dh_a, dh_b, dc_in, dc_prev_out)
    frames, H, W, F = c_cur.shape
    assert gates_dz.dtype == torch.bfloat16
    for t in (gates_dz, c_prev, c_cur, dc_prev_out):
        assert t.is_contiguous()
    n_in = 2 + 2 + 1 + (dh_b is not None) + (dc_in is not None)       # gates (bf16: 4 x 2 B = 2 words), c_prev, c_cur, dh ...
    with _timed('hbm:lstm_gates_bwd_bf16_kernel (BPTT gate backward on the bf16 tape: dz in place of the saved gates, dc)',
                4.0 * (n_in + 2 + 1) * F * frames * H * W):
        calls.check(lib(), lib().lu_lstm_gates_bwd_bf16(gates_dz.data_ptr(), c_prev.data_ptr(), c_cur.data_ptr(),
                                                        dh_a.data_ptr(), dh_a.stride(0), _p(dh_b), _p(dc_in),
                                                        dc_prev_out.data_ptr(), frames, H * W, F, _stream()),
                    'lu_lstm_gates_bwd_bf16')


def _colws(rows, Cc, device):
    n = lib().lu_colreduce_workspace_bytes(rows, Cc)
    return torch.empty((n + 7) // 8, device=device, dtype=torch.float64)


def colsum(x2d_rows, Cc, ld, ptr, out, beta, device):
    ws = _colws(x2d_rows, Cc, device)
    calls.check(lib(), lib().lu_colsum(ptr, x2d_rows, Cc, ld, out.data_ptr(), beta, ws.data_ptr(), _stream()),
                'lu_colsum')


def bias_grad(dy, out, beta=0.0):
    """out[N] = sum over all pixels of dy[..., N]."""
    _chk(dy, out)
    assert dy.is_contiguous()
    Cc = dy.shape[-1]
    colsum(dy.numel() // Cc, Cc, Cc, dy.data_ptr(), out, beta, dy.device)


def bn_stats(y):
    _chk(y)
    Cc = y.shape[-1]
    rows = y.numel() // Cc
    sums = torch.empty(2 * Cc, device=y.device, dtype=torch.float64)
    ws = _colws(rows, Cc, y.device)
    calls.check(lib(), lib().lu_bn_stats(y.data_ptr(), rows, Cc, sums.data_ptr(), ws.data_ptr(), _stream()),
                'lu_bn_stats')
    return sums


def bn_finalize_train(sums, count, gamma, beta, eps, momentum, mm, mv):
    Cc = gamma.numel()
    dev = gamma.device
    scale, shift, mean, invstd = (torch.empty(Cc, device=dev, dtype=torch.float32) for _ in range(4))
    calls.check(lib(), lib().lu_bn_finalize_train(sums.data_ptr(), float(count), gamma.data_ptr(), beta.data_ptr(),
                                                  eps, momentum, _p(mm), _p(mv), scale.data_ptr(), shift.data_ptr(),
                                                  mean.data_ptr(), invstd.data_ptr(), Cc, _stream()),
                'lu_bn_finalize_train')
    return scale, shift, mean, invstd


def bn_finalize_infer(gamma, beta, mm, mv, eps):
    Cc = gamma.numel()
    scale, shift = (torch.empty(Cc, device=gamma.device, dtype=torch.float32) for _ in range(2))
    calls.check(lib(), lib().lu_bn_finalize_infer(gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr(), eps,
                                                  scale.data_ptr(), shift.data_ptr(), Cc, _stream()),
                'lu_bn_finalize_infer')
    return scale, shift


def bn_lrelu_apply(y, scale, shift, alpha, out=None, out_bf16=False):
    """out_bf16: the activation is stored rounded to bf16 (every consumer is a bf16-operand convolution / weight gradient)."""
    _chk(y, scale, shift)
    Cc = y.shape[-1]
    if out_bf16:
        out = torch.empty(y.shape, device=y.device, dtype=torch.bfloat16)
        with _timed('hbm:bn_lrelu_apply_kernel (normalise + LeakyReLU: 1 read + 1 write)', 6.0 * y.numel()):
            calls.check(lib(), lib().lu_bn_lrelu_apply_bf16(y.data_ptr(), out.data_ptr(), scale.data_ptr(), shift.data_ptr(), alpha,
                                                            y.numel() // Cc, Cc, _stream()), 'lu_bn_lrelu_apply_bf16')
        return out
    if out is None:
        out = torch.empty_like(y)
    with _timed('hbm:bn_lrelu_apply_kernel (normalise + LeakyReLU: 1 read + 1 write)', 8.0 * y.numel()):
        calls.check(lib(), lib().lu_bn_lrelu_apply(y.data_ptr(), out.data_ptr(), scale.data_ptr(), shift.data_ptr(), alpha,
                                                   y.numel() // Cc, Cc, _stream()), 'lu_bn_lrelu_apply')
    return out


def bn_lrelu_bwd_reduce(y, dz, scale, shift, mean, invstd, alpha):
    Cc = y.shape[-1]
    rows = y.numel() // Cc
    sums = torch.empty(2 * Cc, device=y.device, dtype=torch.float64)
    ws = _colws(rows, Cc, y.device)
    calls.check(lib(), lib().lu_bn_lrelu_bwd_reduce(y.data_ptr(), dz.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                    mean.data_ptr(), invstd.data_ptr(), alpha, rows, Cc,
                                                    sums.data_ptr(), ws.data_ptr(), _stream()),
                'lu_bn_lrelu_bwd_reduce')
    return sums


def bn_lrelu_bwd_apply(y, dz, scale, shift, mean, invstd, alpha, sums, count, dgamma, dbeta, out=None, out_bf16=False):
    """out_bf16: a NEW bf16 tensor holds the result (every reader rounds it to bf16 MFMA operands: same values, half the bytes)."""
    Cc = y.shape[-1]
    if out_bf16:
        out = torch.empty(y.shape, device=y.device, dtype=torch.bfloat16)
        with _timed('hbm:bn_lrelu_bwd_apply_kernel (BN + LeakyReLU backward: 2 reads + 1 write)', 10.0 * y.numel()):
            calls.check(lib(), lib().lu_bn_lrelu_bwd_apply_bf16(y.data_ptr(), dz.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                                mean.data_ptr(), invstd.data_ptr(), alpha, sums.data_ptr(),
                                                                float(count), out.data_ptr(), _p(dgamma), _p(dbeta),
                                                                y.numel() // Cc, Cc, _stream()), 'lu_bn_lrelu_bwd_apply_bf16')
        return out
    if out is None:
        out = torch.empty_like(y)
    with _timed('hbm:bn_lrelu_bwd_apply_kernel (BN + LeakyReLU backward: 2 reads + 1 write)', 12.0 * y.numel()):
        calls.check(lib(), lib().lu_bn_lrelu_bwd_apply(y.data_ptr(), dz.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                       mean.data_ptr(), invstd.data_ptr(), alpha, sums.data_ptr(),
                                                       float(count), out.data_ptr(), _p(dgamma), _p(dbeta),
                                                       y.numel() // Cc, Cc, _stream()), 'lu_bn_lrelu_bwd_apply')
    return out


RESIZE_CONVENTIONS = ('tf2.0', 'half_pixel')     # legacy src = o/2 (TF 2.0 / 2.1 keras.backend.resize_images) | tf.image.resize v2


def _legacy(resize):
    if resize not in RESIZE_CONVENTIONS:
        raise ValueError('resize convention must be one of %s' % (RESIZE_CONVENTIONS,))
    return 1 if resize == 'tf2.0' else 0


def upsample2x(x, resize='tf2.0', out_bf16=False):
    """out_bf16: store the result rounded to bf16 (its only consumer is a bf16-operand convolution: same values, half the bytes)."""
    _chk(x)
    frames, H, W, Cc = x.shape
    if out_bf16:
        y = torch.empty((frames, 2 * H, 2 * W, Cc), device=x.device, dtype=torch.bfloat16)
        calls.check(lib(), lib().lu_upsample2x_fwd_bf16(x.data_ptr(), y.data_ptr(), frames, H, W, Cc, _legacy(resize), _stream()),
                    'lu_upsample2x_fwd_bf16')
        return y
    y = torch.empty((frames, 2 * H, 2 * W, Cc), device=x.device, dtype=torch.float32)
    calls.check(lib(), lib().lu_upsample2x_fwd(x.data_ptr(), y.data_ptr(), frames, H, W, Cc, _legacy(resize), _stream()),
                'lu_upsample2x_fwd')
    return y


def upsample2x_bwd(dy, in_hw, resize='tf2.0'):
    """dy: [frames,2H,2W,C] (channel-slice view allowed) -> [frames,H,W,C]."""
    _chk(dy)
    frames, _, _, Cc = dy.shape
    H, W = in_hw
    dx = torch.empty((frames, H, W, Cc), device=dy.device, dtype=torch.float32)
    calls.check(lib(), lib().lu_upsample2x_bwd(dy.data_ptr(), dy.stride(2), dx.data_ptr(), frames, H, W, Cc, _legacy(resize), _stream()),
                'lu_upsample2x_bwd')
    return dx


def window_copy(x, out_hw, off, mode, out=None, beta=0.0):
    """mode 1: REFLECT pad (reference Networks.py:232); mode 0: zero outside (crop / zero-embed)."""
    _chk(x, out)
    frames, Hx, Wx, Cc = x.shape
    Hy, Wy = out_hw
    if out is None:
        out = torch.empty((frames, Hy, Wy, Cc), device=x.device, dtype=torch.float32)
    calls.check(lib(), lib().lu_window_copy(x.data_ptr(), x.stride(2), out.data_ptr(), frames, Hx, Wx, Hy, Wy, Cc,
                                            off[0], off[1], mode, beta, _stream()), 'lu_window_copy')
    return out


def wce_forward(logits2d, gt1d, class_w, want_softmax):
    """-> (sums[2] double: [sum ce*w*valid, sum valid], softmax or None)."""
    _chk(logits2d, gt1d, class_w)
    rows = gt1d.numel()
    sums = torch.empty(2, device=logits2d.device, dtype=torch.float64)
    ws = torch.empty(lib().lu_wce_workspace_bytes(rows) // 8 + 1, device=logits2d.device, dtype=torch.float64)
    sm = torch.empty_like(logits2d) if want_softmax else None
    calls.check(lib(), lib().lu_softmax_wce_fwd(logits2d.data_ptr(), gt1d.data_ptr(), class_w.data_ptr(), _p(sm),
                                                sums.data_ptr(), rows, ws.data_ptr(), _stream()), 'lu_softmax_wce_fwd')
    return sums, sm


def wce_backward(logits2d, gt1d, class_w, sums, grad_scale=1.0):
    dl = torch.empty_like(logits2d)
    calls.check(lib(), lib().lu_softmax_wce_bwd(logits2d.data_ptr(), gt1d.data_ptr(), class_w.data_ptr(),
                                                sums.data_ptr(), grad_scale, dl.data_ptr(), gt1d.numel(), _stream()),
                'lu_softmax_wce_bwd')
    return dl


def softmax3(logits):
    _chk(logits)
    assert logits.is_contiguous() and logits.shape[-1] == 3
    out = torch.empty_like(logits)
    calls.check(lib(), lib().lu_softmax3(logits.data_ptr(), out.data_ptr(), logits.numel() // 3, _stream()),
                'lu_softmax3')
    return out


def softmax_last(logits):
    """Softmax over the last (class) axis for any head depth (Networks.py:205-206); 3 classes take the dedicated kernel."""
    if logits.shape[-1] == 3:
        return softmax3(logits)
    _chk(logits)
    assert logits.is_contiguous() and logits.dtype == torch.float32
    out = torch.empty_like(logits)
    c = int(logits.shape[-1])
    calls.check(lib(), lib().lu_softmax_rows(logits.data_ptr(), out.data_ptr(), logits.numel() // c, c, _stream()),
                'lu_softmax_rows')
    return out


def wce_loss(sums):
    loss = torch.empty(1, device=sums.device, dtype=torch.float32)
    calls.check(lib(), lib().lu_wce_finalize(sums.data_ptr(), loss.data_ptr(), _stream()), 'lu_wce_finalize')
    return loss


def adam_step(p, g, m, v, alpha, b1, b2, eps, grad_scale=1.0):
    _chk(p, g, m, v)
    with _timed('hbm:adam_kernel (tf.keras Adam on the flat buffers: 4 reads + 3 writes)', 28.0 * p.numel()):
        calls.check(lib(), lib().lu_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), alpha, b1,
                                              b2, eps, grad_scale, _stream()), 'lu_adam_step')


def scale_frames(x, keep):
    _chk(x, keep)
    assert x.is_contiguous()
    calls.check(lib(), lib().lu_scale_frames(x.data_ptr(), keep.data_ptr(), x.shape[0], x.numel() // x.shape[0],
                                             _stream()), 'lu_scale_frames')


def state_begin(dst, src, keep, dst16=None):
    """dst [B,...] = src * keep[b] (src None: zeros; keep None: copy), dst16 = its bf16 copy -- lu_state_begin."""
    _chk(dst, src, keep, dst16)
    assert dst.is_contiguous() and (src is None or (src.is_contiguous() and src.shape == dst.shape))
    assert dst16 is None or (dst16.is_contiguous() and dst16.shape == dst.shape and dst16.dtype == torch.bfloat16)
    assert keep is None or keep.numel() == dst.shape[0]
    calls.check(lib(), lib().lu_state_begin(dst.data_ptr(), _p(dst16), _p(src), _p(keep), dst.shape[0], dst.numel() // dst.shape[0],
                                            _stream()), 'lu_state_begin')


def transpose_inner(x, n, a, b):
    """[n,a,b] -> [n,b,a] on a contiguous tensor (NCHW <-> NHWC per frame)."""
    _chk(x)
    y = torch.empty(x.numel(), device=x.device, dtype=torch.float32)
    calls.check(lib(), lib().lu_transpose_inner(x.data_ptr(), y.data_ptr(), n, a, b, _stream()), 'lu_transpose_inner')
    return y


def add_(y, x):
    _chk(x, y)
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    calls.check(lib(), lib().lu_add_inplace(y.data_ptr(), x.data_ptr(), y.numel(), _stream()), 'lu_add_inplace')
    return y
