"""Thin, pointer-level call helpers over the C ABI (descriptor packing + error checks).
Everything here works on raw addresses; ops.py feeds it torch device pointers."""
import ctypes as C
import functools

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


SPLIT_CAP, SPLIT_MIN_IT = 32, 12      # most K splits / fewest k-steps per block conv_splits considers
FORCE_SPLITS = None   # A/B tools (tools/step_sweep.py): an integer forces that K split on every launch the plan would consider
IT_US = 1.89     # one k-step (16 channels of one tap) of a 256 x 128 fp32 tile, per block of a resident PAIR, microseconds
                 # (measured: 2048-block fused step, 620 k-steps, 9.39 ms = 8 block-times; a lone block needs 2.28)


def launch_rounds(blocks):
    """Block-times a launch of `blocks` equal tile blocks costs: 8 XCDs x 32 CUs x 2 resident blocks sharing one MFMA pipe.
    A partial round of <= 32 blocks per XCD runs one block per CU (a lone block reaches ~83 % of a pair's rate).  Fitted to
    tools/step_sweep.py on the B = 1 streaming shapes: 57 / 60 blocks per XCD ~2.0-2.3, 88 ~3.1, 95 ~3.1, 255 ~8.0."""
    per = -(-blocks // 8)
    full, rem = divmod(per, 64)
    if full == 0:
        return 1.2 if rem <= 32 else 2.0
    return 2.0 * full + (0.0 if rem == 0 else 1.1 if rem <= 32 else 2.0)


def conv_tiles(frames, Hout, Wout, N, k, halo=True):
    """256-pixel x 128-column tiles of a launch: 8 x 32-pixel patches where the halo kernel applies (stride-1 3x3 / 5x5,
    N > 64, <= 25 % of the patches wasted -- mirrors lu_conv2d_fwd), flattened pixel rows otherwise."""
    patches = frames * -(-Hout // 8) * -(-Wout // 32)
    if halo and k in (3, 5) and N > 64 and patches * 256 * 4 <= frames * Hout * Wout * 5:
        return patches * -(-N // 128)
    return -(-(frames * Hout * Wout) // 256) * -(-N // 128)


def _knobs():
    """The tunables the launch plans depend on, as a cache key: tools (tools/step_sweep.py) and tests change the module
    globals between runs, and a memoised plan must not outlive the knobs it was made with."""
    return (SPLIT_CAP, SPLIT_MIN_IT, IT_US, FORCE_SPLITS)


def clear_plan_cache():
    """For code that monkey-patches conv_splits / conv_cost_us themselves (the knob values are part of the cache key already)."""
    _conv_splits.cache_clear()
    _conv_plan.cache_clear()
    _fused_step_cost_us.cache_clear()


def chunk_splits(chunks, splits):
    """Non-empty slices when `chunks` chunks are dealt out ceil(chunks / splits) at a time (mirrors lu_conv_chunk_splits: the fp32
    halo kernel takes its compile-time tap sequence for a K split only if every slice gets a chunk)."""
    if chunks <= 0 or splits <= 0:
        return 0
    per = -(-chunks // splits)
    return -(-chunks // per)


def conv_cost_us(frames, Hout, Wout, N, k, channels, splits, halo=True, it_us=None):
    """Modelled duration of the fp32 conv kernels with a K split (+ the slab reduce), microseconds."""
    M = frames * Hout * Wout
    n_it = k * k * -(-channels // 16)
    per = n_it / float(splits)
    if halo and splits > 1 and chunk_splits(-(-channels // 16), splits) == splits:
        per = -(-(-(-channels // 16)) // splits) * k * k      # fp32 halo kernel: whole 16-channel chunks per slice (the longest one)
    t = launch_rounds(conv_tiles(frames, Hout, Wout, N, k, halo) * splits) * (per + 12) * (IT_US if it_us is None else it_us)
    if splits > 1:
        t += (2 * splits + 1) * M * N * 4 / 4e6 + 5        # slabs written + read, result written, at ~4 TB/s
    return t


# The plans are pure functions of the launch shape AND the knobs above; the streaming frame asks ~200 times per frame, so they are
# memoised -- on a key that carries the knob values (the public wrappers read the module globals at call time).
@functools.lru_cache(maxsize=4096)
def _conv_splits(frames, Hout, Wout, N, k, channels, halo, knobs):
    cap, min_it, it_us, force = knobs
    n_it = k * k * -(-channels // 16)
    if force is not None:
        return max(1, min(int(force), n_it))
    if conv_tiles(frames, Hout, Wout, N, k, halo) > 2048 or n_it < 64:
        return 1
    cands = [s for s in range(1, cap + 1) if n_it // s >= min_it] or [1]
    if halo:      # whole chunks per slice (round 5): only split counts that leave no slice empty
        nch = -(-channels // 16)
        cands = [s for s in cands if s == 1 or chunk_splits(nch, s) == s] or [1]
    return min(cands, key=lambda s: (conv_cost_us(frames, Hout, Wout, N, k, channels, s, halo, it_us), s))


def conv_splits(frames, Hout, Wout, N, k, channels, halo=True):
    """K-axis split for launches with too few 256x128 output tiles to fill 256 CUs: the split count with the smallest
    modelled duration (whole rounds of resident blocks matter more than the count itself: 152 tiles x 3 = 456 blocks is one
    round, x 4 = 608 is two)."""
    return _conv_splits(frames, Hout, Wout, N, k, channels, bool(halo), _knobs())


@functools.lru_cache(maxsize=4096)
def _conv_plan(frames, Hout, Wout, N, k, channels, halo_ok, knobs):
    it_us = knobs[2]
    s_g = _conv_splits(frames, Hout, Wout, N, k, channels, False, knobs)
    c_g = conv_cost_us(frames, Hout, Wout, N, k, channels, s_g, False, it_us)
    if not halo_ok or conv_tiles(frames, Hout, Wout, N, k, True) == conv_tiles(frames, Hout, Wout, N, k, False):
        return s_g, halo_ok, c_g
    s_h = _conv_splits(frames, Hout, Wout, N, k, channels, True, knobs)
    c_h = conv_cost_us(frames, Hout, Wout, N, k, channels, s_h, True, it_us)
    if c_g < 0.95 * c_h:
        return s_g, False, c_g
    return s_h, True, c_h


def conv_plan(frames, Hout, Wout, N, k, channels, halo_ok=True):
    """(splits, use_halo, modelled microseconds) of an fp32 LU_EPI_BIAS launch.  Where the halo kernel applies, its 8 x 32
    patches may hang over the image (136-pixel rows: 15 %); the general kernel tiles flattened pixel rows without waste
    and wins such launches when they are K-split anyway (B = 1 L1 step: 3.20 -> 2.93 ms measured).  The halo kernel is the
    faster one per k-step (134 vs 128 TFLOP/s in training), hence the 5 % margin."""
    return _conv_plan(frames, Hout, Wout, N, k, channels, bool(halo_ok), _knobs())


@functools.lru_cache(maxsize=4096)
def _fused_step_cost_us(frames, H, W, F, k, channels, knobs):
    blocks = conv_tiles(frames, H, W, 4 * F, k)
    return launch_rounds(blocks) * (k * k * -(-channels // 16) + 20) * knobs[2]


def fused_step_cost_us(frames, H, W, F, k, channels):
    """Modelled duration of the fused fp32 ConvLSTM step (8 x 32-pixel patches x 32 hidden channels x 4 gates per block)."""
    return _fused_step_cost_us(frames, H, W, F, k, channels, _knobs())


def conv2d(lib, stream, srcs, frames, Hin, Win, Hout, Wout, k, stride, dil, pad_t, pad_l, N, bias, out,
           out_frame_stride, out_pix_stride, lstm=None, splits=1, workspace=None, out_row_stride=0, precision=0, k_h=0,
           flags=0, h16=None, post=None):
    """h16 = (ptr, frame_stride): optional bf16 copy of h written by the fused ConvLSTM epilogue (precision 1).
    post = (scale_ptr, shift_ptr, alpha): inference BN affine + LeakyReLU folded into the store (LU_EPI_BIAS)."""
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
    if post is not None:
        d.post_scale, d.post_shift, d.post_alpha = post
    if h16 is not None:
        d.h16_out, d.h16_frame_stride = h16
    if lstm is not None:
        d.epilogue = cabi.LU_EPI_LSTM
        (d.c_prev, d.c_prev_frame_stride, d.c_out, d.c_out_frame_stride, d.h_out, d.h_frame_stride,
         d.gates_out, d.gates_frame_stride) = lstm
    check(lib, lib.lu_conv2d_fwd(C.byref(d), stream), 'lu_conv2d_fwd')


def wgrad_desc(x, x_fs, x_ps, Cin, dy, dy_fs, dy_ps, N, frames, Hin, Win, Hout, Wout, k, stride, pad_t, pad_l,
               dw, dw_tap_stride, dw_row_stride, splits, beta, precision=0, dbias=None, dbias_beta=0.0,
               x_dtype=cabi.LU_F32, dy_dtype=cabi.LU_F32, flags=0, terms=0, x_term_stride=0, dy_term_stride=0):
    d = cabi.WgradDesc()
    d.terms, d.x_term_stride, d.dy_term_stride = terms, x_term_stride, dy_term_stride
    d.x, d.x_frame_stride, d.x_pix_stride, d.C = x, x_fs, x_ps, Cin
    d.dy, d.dy_frame_stride, d.dy_pix_stride, d.N = dy, dy_fs, dy_ps, N
    d.frames, d.Hin, d.Win, d.Hout, d.Wout = frames, Hin, Win, Hout, Wout
    d.k, d.stride, d.pad_t, d.pad_l = k, stride, pad_t, pad_l
    d.dw, d.dw_tap_stride, d.dw_row_stride, d.splits, d.beta = dw, dw_tap_stride, dw_row_stride, splits, beta
    d.precision = precision
    d.dbias, d.dbias_beta = dbias, dbias_beta
    d.x_dtype, d.dy_dtype, d.flags = x_dtype, dy_dtype, flags
    return d


BF16_ROW_ROUNDS = 5      # (A/B: bench.py --wgrad-rounds)


def wgrad_splits_bf16_row(pixels, k, Cin, N, ct, rounds=None, cus=256, all_taps=False):
    """Pixel-axis split of the bf16 kernel-row variant.  Its blocks are numbered XCD-aware (all tiles of a pixel slab on
    one XCD), one block per CU at 128-channel tiles (two at 64), all of equal length: pick a multiple of 8 slabs so that
    tiles x slabs is close to `rounds` full waves of the chip -- few fat slabs keep the slab write + re-read small
    (round 1: ~3000 blocks = 38 slabs of 26 MB at level 1 = 1 GB per launch; now 16 slabs)."""
    rounds = BF16_ROW_ROUNDS if rounds is None else rounds
    inner = k * -(-Cin // ct) * -(-N // 128)
    per_round = cus * (1 if ct == 128 else 2)
    if all_taps:         # all-taps form of the 3x3 layers: (64-channel tiles) x (128-column tiles) blocks per slab, one block per CU
        inner = -(-Cin // 64) * -(-N // 128)
        per_round = cus
    if inner <= 4 and Cin <= 64 and (ct == 64 or all_taps):      # narrow layers (one channel tile, one column tile): 856 slabs of a 74 KB gradient made the slab REDUCE the
        rounds = 1       # long pole (74 blocks streaming 63 MB: 0.29 ms); one round of blocks = 170 slabs
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
    # thin / 1x1 layers (the image source of the last up block, the 32 -> 3 logits conv): a slab of a few KB costs nothing to
    # reduce, while 256 blocks of 256 threads cannot keep enough loads in flight to stream their 2 M pixels (0.5 ms measured
    # for 0.07 ms of HBM time): up to 1024 slabs
    cap = 1024 if k * k * Cin * N <= 65536 else 256
    return max(1, min(s, cap))
