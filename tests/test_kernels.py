"""Kernel-level parity against the oracle, through the C ABI of include/lstm_unet_hip.h.

Two backends run the SAME assertions:
  * 'emu' (CPU, not gpu): lstm-unet_amd/csrc/*.hip compiled for the host SIMT emulator in tests/emu
    (test infrastructure) -- catches index / fragment-layout / masking mistakes without a GPU;
  * 'hip' (`-m gpu`): the real gfx950 library on the MI355X.
Tolerances (fp32 kernels vs fp64 oracle): 5e-5 absolute for convolutions with |values| ~ O(1..10)
and K up to ~1200, 1e-4..2e-4 for weight gradients (long pixel reductions), ~1e-5 pointwise.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import np_oracle as npo
from oracle import torch_oracle as tho
import kernel_harness as KH
from kernel_harness import f32
from lu_native import cabi, calls

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


@pytest.fixture(scope='module', params=BACKENDS)
def be(request):
    return KH.backend(request.param)


RNG = np.random.default_rng(11)


def rnd(*shape, scale=1.0):
    return f32(RNG.standard_normal(shape) * scale)


def close(a, b, tol):
    err = float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
    assert np.isfinite(err) and err <= tol, err


def ck(be, rc, what):
    calls.check(be.lib, rc, what)


CONV_CASES = [  # frames, H, W, C, N, k, stride
    (1, 8, 8, 16, 32, 3, 1), (2, 9, 7, 8, 12, 3, 1), (1, 10, 12, 20, 40, 5, 1), (2, 8, 10, 1, 8, 5, 1),
    (1, 9, 9, 3, 70, 3, 2), (1, 8, 8, 24, 130, 3, 2), (1, 6, 6, 8, 3, 1, 1), (3, 5, 5, 4, 33, 5, 2),
    (2, 9, 9, 20, 136, 3, 1), (1, 7, 8, 36, 128, 5, 1), (1, 8, 8, 16, 72, 3, 2),     # wide tiles: LDS-DMA staged variant
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd(be, case):
    fr, H, W, Cc, N, k, s = case
    x, w, b = rnd(fr, H, W, Cc), rnd(k, k, Cc, N, scale=0.2), rnd(N)
    close(KH.conv2d(be, [x], [w], b, k, s), npo.conv2d_same(x, w, b, s), 5e-5)


def test_conv_ksplit_and_many_mtiles(be):
    """K-axis split (tile-starved coarse levels) and the XCD-aware tile order with > 8 m-tiles."""
    x, w, b = rnd(1, 16, 16, 48), rnd(3, 3, 48, 40, scale=0.2), rnd(40)
    ref = npo.conv2d_same(x, w, b, 1)
    for sp in (2, 3, 5):
        close(KH.conv2d(be, [x], [w], b, 3, 1, splits=sp), ref, 5e-5)
    x, w = rnd(3, 30, 31, 8), rnd(3, 3, 8, 8, scale=0.3)          # 2790 px -> 11 m-tiles (padded to 16)
    close(KH.conv2d(be, [x], [w], None, 3, 1), npo.conv2d_same(x, w, None, 1), 5e-5)
    xi, xh = rnd(2, 6, 6, 1), rnd(2, 6, 6, 32)                     # thin + vector source with a K split
    wi, wh = rnd(5, 5, 1, 24, scale=0.3), rnd(5, 5, 32, 24, scale=0.1)
    ref = npo.conv2d_same(xi, wi) + npo.conv2d_same(xh, wh)
    close(KH.conv2d(be, [xi, xh], [wi, wh], None, 5, 1, splits=4), ref, 5e-5)


def test_conv_halo_ksplit_ranges_start_anywhere_in_a_chunk(be):
    """conv_halo_kernel (N > 64) with a K split: the stage range of a slice may begin on ANY tap of a chunk -- in particular on the
    last one (the next halo image is then requested in the prologue) and on the last but one (requested in the first stage) --
    and may cross from the first source into the second; the weight tiles travel by LDS-DMA two stages ahead of their stage."""
    for k, splits in ((5, (2, 3, 6, 7)), (3, (2, 4, 5, 7))):
        xa, xb = rnd(2, 9, 35, 16), rnd(2, 9, 35, 32)            # 9 x 35: ragged patches (8 x 32 tiles hang over)
        wa, wb, b = rnd(k, k, 16, 128, scale=0.1), rnd(k, k, 32, 128, scale=0.1), rnd(128)
        ref = npo.conv2d_same(xa, wa, b, 1) + npo.conv2d_same(xb, wb, None, 1)
        for sp in splits:                                          # 5x5: 75 stages -> slices of 38 / 25 / 13 / 11; 3x3: 27 -> 14 / 7 / 6 / 4
            close(KH.conv2d(be, [xa, xb], [wa, wb], b, k, 1, splits=sp), ref, 5e-5)      # 3 chunks: sp = 2, 3 chunk-aligned, the others counted
        close(KH.conv2d(be, [xb], [wb], None, k, 1, splits=7 if k == 5 else 3), npo.conv2d_same(xb, wb, None, 1), 5e-5)      # 2 chunks, more slices: counted


def test_conv_halo_ksplit_whole_chunks_on_the_compile_time_tap_sequence(be):
    """Round 5 (ABI v10): a K split of the fp32 halo kernel deals out WHOLE 16-channel chunks -- ceil(chunks / splits) per slice --
    so that K-split launches run the compile-time tap sequence of the unsplit ones (conv_halo_kernel<K, BIAS, ST = true>).  Chunk
    counts that do not divide (7 chunks over 2 / 3 / 4 / 7 slices: 4+3, 3+3+1, 2+2+2+1, 1 x 7), slices that cross from the first
    source into the second, ragged patches, a ragged last chunk (C = 40: 16 + 16 + 8) -- against the oracle; 5 slices of 7 chunks would
    leave one empty, so the library keeps the counted loop there."""
    for k in (5, 3):
        xa, xb = rnd(2, 9, 35, 40), rnd(2, 9, 35, 64)
        wa, wb, b = rnd(k, k, 40, 128, scale=0.1), rnd(k, k, 64, 128, scale=0.1), rnd(128)
        ref = npo.conv2d_same(xa, wa, b, 1) + npo.conv2d_same(xb, wb, None, 1)      # 3 + 4 = 7 chunks
        for sp in (2, 3, 4, 7):
            close(KH.conv2d(be, [xa, xb], [wa, wb], b, k, 1, splits=sp), ref, 5e-5)
        close(KH.conv2d(be, [xa, xb], [wa, wb], b, k, 1, splits=5), ref, 5e-5)
    # the slab form (LU_CONV_F_SLABS_ONLY + lu_lstm_gates_fwd_slabs) rides on the same launch: covered by
    # test_conv_post_affine_lrelu_and_slab_gates, whose K-split launches take whole chunks as well


def test_conv_block_numbering_variants_are_bitwise_equal(be):
    """Few-tile launches number their blocks in equal runs of (tile, split) work items per XCD (weights-major or
    activations-major order, lu_block_tile); LU_CONV_F_NO_BALANCE keeps the m-tile-per-XCD numbering.  The numbering moves
    blocks between XCDs and nothing else: results must be bit-identical, with and without a K split, in the general
    and the halo kernel, with 5 / 11 / 3 m-tiles (not multiples of 8) and for both orders (k*k*N above / below M)."""
    nb = cabi.LU_CONV_F_NO_BALANCE
    for (fr, H, W, Cc, N, k, sp) in [(1, 34, 34, 32, 160, 5, 3), (3, 30, 31, 16, 24, 3, 1), (1, 20, 30, 32, 136, 3, 4),
                                     (2, 16, 30, 36, 128, 5, 2), (1, 17, 40, 16, 72, 5, 1)]:
        x, w, b = rnd(fr, H, W, Cc), rnd(k, k, Cc, N, scale=0.2), rnd(N)
        a = KH.conv2d(be, [x], [w], b, k, 1, splits=sp)
        close(a, npo.conv2d_same(x, w, b, 1), 5e-5)
        assert np.array_equal(a, KH.conv2d(be, [x], [w], b, k, 1, splits=sp, flags=nb)), (fr, H, W, Cc, N, k, sp)
    F = 32
    x, h, c = rnd(1, 16, 40, 8), rnd(1, 16, 40, F, scale=0.5), rnd(1, 16, 40, F)
    ker, rec, b = rnd(5, 5, 8, 4 * F, scale=0.3), rnd(5, 5, F, 4 * F, scale=0.1), rnd(4 * F, scale=0.5)
    fused = KH.convlstm_step_fused(be, x, h, c, ker, rec, b)
    for u, v in zip(fused, KH.convlstm_step_fused(be, x, h, c, ker, rec, b, flags=nb)):
        assert np.array_equal(u, v)


def test_conv_post_affine_lrelu_and_slab_gates(be):
    """ABI v5: (a) the inference BatchNorm affine + LeakyReLU folded into the conv's store or its slab reduce is the same
    arithmetic as lu_bn_lrelu_apply on the plain conv output (bit-identical), in the general, halo and bf16 kernels, with and
    without a K split; (b) LU_CONV_F_SLABS_ONLY + lu_lstm_gates_fwd_slabs == K-split conv + reduce + lu_lstm_gates_fwd."""
    for (fr, H, W, Cc, N, k, st, sp, prec) in [(1, 12, 14, 32, 24, 3, 1, 1, 0), (1, 12, 14, 32, 24, 3, 2, 3, 0),
                                               (1, 16, 32, 20, 136, 3, 1, 1, 0), (1, 16, 32, 20, 136, 5, 1, 4, 0),
                                               (1, 16, 32, 32, 136, 3, 1, 2, 1), (1, 12, 14, 32, 72, 3, 2, 1, 1)]:
        x, w, b = rnd(fr, H, W, Cc), rnd(k, k, Cc, N, scale=0.2), rnd(N)
        scale, shift = (1.0 + 0.3 * rnd(N)).astype(np.float32), rnd(N)
        plain = KH.conv2d(be, [x], [w], b, k, st, splits=sp, precision=prec)
        got = KH.conv2d(be, [x], [w], b, k, st, splits=sp, precision=prec, post=(scale, shift, 0.3))
        yd, sd, hd = be.dev(plain), be.dev(scale), be.dev(shift)
        want = be.empty(plain.shape)
        ck(be, be.lib.lu_bn_lrelu_apply(be.ptr(yd), be.ptr(want), be.ptr(sd), be.ptr(hd), 0.3, plain.size // N, N, be.stream),
           'bn_lrelu_apply')
        assert np.array_equal(got, be.host(want)), (fr, H, W, Cc, N, k, st, sp, prec)
        t = plain.astype(np.float64) * scale + shift
        close(got, np.where(t > 0, t, 0.3 * t), 1e-5)
    F = 8
    fr, H, W = 2, 6, 7
    xi, xh, c0 = rnd(fr, H, W, 4), rnd(fr, H, W, 32, scale=0.5), rnd(fr, H, W, F)
    wi, wh, b = rnd(5, 5, 4, 4 * F, scale=0.3), rnd(5, 5, 32, 4 * F, scale=0.1), rnd(4 * F, scale=0.5)
    z = KH.conv2d(be, [xi, xh], [wi, wh], b, 5, 1, splits=3)
    slabs = KH.conv2d(be, [xi, xh], [wi, wh], None, 5, 1, splits=3, slabs=True)
    close(slabs.sum(0).reshape(z.shape) + b, z, 1e-5)
    outs = []
    for route in ('z', 'slabs'):
        zd, sl, bd, c0d = be.dev(z), be.dev(slabs), be.dev(b), be.dev(c0)
        c1, h1, g1 = be.empty(c0.shape), be.empty(c0.shape), be.empty(z.shape)
        if route == 'z':
            ck(be, be.lib.lu_lstm_gates_fwd(be.ptr(zd), be.ptr(c0d), be.ptr(c1), be.ptr(h1), be.ptr(g1), fr, H * W, F,
                                            H * W * F, be.stream), 'gates_fwd')
        else:
            ck(be, be.lib.lu_lstm_gates_fwd_slabs(be.ptr(sl), 3, be.ptr(bd), be.ptr(c0d), be.ptr(c1), be.ptr(h1), be.ptr(g1),
                                                  fr, H * W, F, H * W * F, be.stream), 'gates_fwd_slabs')
        outs.append((be.host(c1), be.host(h1), be.host(g1)))
    for u, v in zip(*outs):
        assert np.array_equal(u, v)
    h_ref, c_ref = npo.convlstm_step(np.concatenate([xi, xh], -1), np.zeros_like(c0), c0,
                                     np.concatenate([wi, wh], 2), np.zeros((5, 5, F, 4 * F), np.float32), b)
    close(outs[1][0], c_ref, 2e-5)
    close(outs[1][1], h_ref, 2e-5)


def test_conv_halo_variant(be):
    """8x32-patch halo-reuse kernel (stride-1 3x3 / 5x5, N > 64): ragged patches, two sources incl. a thin one,
    K split, and the fused ConvLSTM epilogue on it."""
    for (fr, H, W, Cc, N, k) in [(1, 16, 32, 20, 136, 3), (2, 16, 30, 36, 128, 5), (1, 17, 40, 16, 72, 5)]:
        x, w, b = rnd(fr, H, W, Cc), rnd(k, k, Cc, N, scale=0.2), rnd(N)
        close(KH.conv2d(be, [x], [w], b, k, 1), npo.conv2d_same(x, w, b, 1), 5e-5)
    xi, xh = rnd(1, 16, 32, 1), rnd(1, 16, 32, 40)
    wi, wh = rnd(5, 5, 1, 72, scale=0.3), rnd(5, 5, 40, 72, scale=0.1)
    ref = npo.conv2d_same(xi, wi) + npo.conv2d_same(xh, wh)
    close(KH.conv2d(be, [xi, xh], [wi, wh], None, 5, 1), ref, 5e-5)
    close(KH.conv2d(be, [xi, xh], [wi, wh], None, 5, 1, splits=3), ref, 5e-5)
    F = 32
    x, h, c = rnd(1, 16, 32, 8), rnd(1, 16, 32, F, scale=0.5), rnd(1, 16, 32, F)
    ker, rec, b = rnd(5, 5, 8, 4 * F, scale=0.3), rnd(5, 5, F, 4 * F, scale=0.1), rnd(4 * F, scale=0.5)
    h1, c1 = npo.convlstm_step(x, h, c, ker, rec, b)
    hg, cg, gates = KH.convlstm_step_fused(be, x, h, c, ker, rec, b)
    close(hg, h1, 2e-5)
    close(cg, c1, 2e-5)
    z = npo.conv2d_same(x, ker, b) + npo.conv2d_same(h, rec)
    close(gates[..., 3 * F:], npo.hard_sigmoid(z[..., 3 * F:]), 2e-5)
    close(gates[..., F:2 * F], npo.hard_sigmoid(z[..., F:2 * F]), 2e-5)


@pytest.mark.parametrize('patch', ['8', '16', 'half'])
def test_conv_bf16_mfma_variant(be, patch):
    # 8 x 32 / 16 x 32 pixel patches (the latter: 5x5, >= 256 blocks) forced through lu_conv_desc.flags; 'half': 4-wave blocks
    # on 8 x 32 patches (WM = 1, two independent blocks per CU).  The wide layers run the third loop generation (round 5:
    # conv_halo_frag3_kernel, kernel-column-major taps); 3x3 on fp32 sources / 8-row patches and the narrow blocks the first.
    _conv_bf16_cases(be, {'16': cabi.LU_CONV_F_PATCH16, '8': cabi.LU_CONV_F_PATCH8, 'half': cabi.LU_CONV_F_HALF_BLOCK}[patch])


def test_conv_bf16_half_blocks_are_bitwise_equal_to_the_full_blocks(be):
    """LU_CONV_F_HALF_BLOCK only changes which waves own which patch rows: same taps, same channel order, same fp32 accumulation
    per output -- bias layers (3x3 / 5x5 on bf16 sources, two sources, ragged extents, a partial column tile, a K split) and the
    fused ConvLSTM step on the bf16 tape must give the same bits as the library's default kernels."""
    emu = be.name == 'emu'          # (the host emulator runs a subset: the CPU suite has to stay within minutes)
    # Third generation (the default): every block shape of the 5x5 kernel and the two 3x3 shapes that run it sum in the same order
    x5, w5, b5 = KH.bf16_round(rnd(2 - emu, 19, 33, 72)), rnd(5, 5, 72, 160, scale=0.1), rnd(160)
    o5 = [KH.conv2d(be, [x5], [w5], b5, 5, precision=1, flags=f_, bf16_src=(0,))
          for f_ in (cabi.LU_CONV_F_HALF_BLOCK, cabi.LU_CONV_F_PATCH16, cabi.LU_CONV_F_PATCH8)[:3 if not emu else 2]]
    assert all(np.array_equal(o5[0], o) for o in o5[1:])
    close(o5[0], npo.conv2d_same(x5, KH.bf16_round(w5), b5, 1), 5e-5)
    if not emu:
        x3, w3, b3 = KH.bf16_round(rnd(2, 20, 40, 40)), rnd(3, 3, 40, 136, scale=0.1), rnd(136)
        o3 = [KH.conv2d(be, [x3], [w3], b3, 3, precision=1, flags=f_, bf16_src=(0,))
              for f_ in (cabi.LU_CONV_F_HALF_BLOCK, cabi.LU_CONV_F_PATCH16, cabi.LU_CONV_F_PATCH8)]
        assert np.array_equal(o3[0], o3[1])                       # half blocks / 16-row patches: the third generation
        close(o3[0], o3[2], 5e-5)                                 # 8-row patches: the first generation's kernel (row-major taps)
        close(o3[0], npo.conv2d_same(x3, KH.bf16_round(w3), b3, 1), 5e-5)
    x5 = rnd(1, 16, 34, 36)                                      # fp32 sources: 5x5 only (the 3x3 halo needs bf16 pieces)
    w5, b5 = rnd(5, 5, 36, 128, scale=0.1), rnd(128)
    assert np.array_equal(KH.conv2d(be, [x5], [w5], b5, 5, precision=1, flags=cabi.LU_CONV_F_HALF_BLOCK),
                          KH.conv2d(be, [x5], [w5], b5, 5, precision=1))
    F = 32
    x, h, c = rnd(1, 16, 32, 8), rnd(1, 16, 32, F, scale=0.5), rnd(1, 16, 32, F)      # fp32 sources, fp32 gates out
    ker, rec, b = rnd(5, 5, 8, 4 * F, scale=0.3), rnd(5, 5, F, 4 * F, scale=0.1), rnd(4 * F, scale=0.5)
    for a_, b_ in zip(KH.convlstm_step_fused(be, x, h, c, ker, rec, b, precision=1, flags=cabi.LU_CONV_F_HALF_BLOCK),
                      KH.convlstm_step_fused(be, x, h, c, ker, rec, b, precision=1)):
        assert np.array_equal(a_, b_)
    if not emu:      # the fused step on the bf16 tape, 5x5 with and without the centre-tap image chunk: half blocks == 16-row patches
        for (k, cin, center) in [(5, 8, False), (5, 1, True)]:
            x, h, c = rnd(2, 18, 40, cin), rnd(2, 18, 40, F, scale=0.5), rnd(2, 18, 40, F)
            ker, rec, b = rnd(k, k, cin, 4 * F, scale=0.3), rnd(k, k, F, 4 * F, scale=0.1), rnd(4 * F, scale=0.5)
            want = KH.convlstm_step_tape16(be, x, h, c, ker, rec, b, center=center, flags=cabi.LU_CONV_F_PATCH16)
            got = KH.convlstm_step_tape16(be, x, h, c, ker, rec, b, center=center, flags=cabi.LU_CONV_F_HALF_BLOCK)
            for a_, b_ in zip(got, want):
                assert np.array_equal(a_, b_), (k, cin, center)


def _conv_bf16_cases(be, flags=0):
    """Mixed-precision halo kernel (precision = 1): operands rounded to bf16, fp32 accumulation.  Against the oracle
    on the SAME bf16-rounded operands only the summation order differs (tolerance as for the fp32 kernels); against
    the unrounded oracle the error is the bf16 operand rounding (2^-9 relative per operand)."""
    R = KH.bf16_round
    for (fr, H, W, Cc, N, k, sp) in [(1, 16, 32, 20, 136, 3, 1), (2, 16, 30, 36, 128, 5, 1), (1, 17, 40, 64, 72, 5, 3),
                                     (1, 8, 33, 100, 96, 3, 2)]:
        x, w, b = rnd(fr, H, W, Cc), rnd(k, k, Cc, N, scale=0.2), rnd(N)
        got = KH.conv2d(be, [x], [w], b, k, 1, splits=sp, precision=1, flags=flags)
        close(got, npo.conv2d_same(R(x), R(w), b, 1), 5e-5)
        full = npo.conv2d_same(x, w, b, 1)
        assert np.abs(got - full).max() <= 2.0 ** -7 * np.abs(full).max()
    xa, xb = rnd(1, 16, 32, 40), rnd(1, 16, 32, 24)               # two sources (UpBlock concat)
    wa, wb = rnd(3, 3, 40, 72, scale=0.1), rnd(3, 3, 24, 72, scale=0.1)
    ref = npo.conv2d_same(R(xa), R(wa)) + npo.conv2d_same(R(xb), R(wb))
    close(KH.conv2d(be, [xa, xb], [wa, wb], None, 3, 1, precision=1, flags=flags), ref, 5e-5)
    F = 32                                                         # fused ConvLSTM step
    x, h, c = rnd(1, 16, 32, 8), rnd(1, 16, 32, F, scale=0.5), rnd(1, 16, 32, F)
    ker, rec, b = rnd(5, 5, 8, 4 * F, scale=0.3), rnd(5, 5, F, 4 * F, scale=0.1), rnd(4 * F, scale=0.5)
    h1, c1 = npo.convlstm_step(R(x), R(h), c, R(ker), R(rec), b)
    hg, cg, _ = KH.convlstm_step_fused(be, x, h, c, ker, rec, b, precision=1, flags=flags)
    close(hg, h1, 3e-5)
    close(cg, c1, 3e-5)
    xi, xh = rnd(1, 16, 32, 1), rnd(1, 16, 32, 40)                 # 1-channel image (zero-padded to 4) + vector source
    wi, wh = rnd(5, 5, 1, 72, scale=0.3), rnd(5, 5, 40, 72, scale=0.1)
    ref = npo.conv2d_same(R(xi), R(wi)) + npo.conv2d_same(R(xh), R(wh))
    xi4 = np.concatenate([xi, np.zeros((1, 16, 32, 3), np.float32)], -1)
    wi4 = np.concatenate([wi, np.zeros((5, 5, 3, 72), np.float32)], 2)
    close(KH.conv2d(be, [xi4, xh], [wi4, wh], None, 5, 1, precision=1, flags=flags), ref, 5e-5)
    with pytest.raises(RuntimeError):                             # thin sources must be padded by the caller
        KH.conv2d(be, [xi, xh], [wi, wh], None, 5, 1, precision=1, flags=flags)
    # bf16 TENSORS as sources (the bf16 BPTT tape: dz for the recurrent / input gradients): same arithmetic, no conversion
    for (fr, H, W, Cc, N, k, sp) in [(2, 16, 30, 40, 128, 5, 1), (1, 9, 33, 96, 72, 3, 2)]:
        x, w, b = rnd(fr, H, W, Cc), rnd(k, k, Cc, N, scale=0.2), rnd(N)
        got = KH.conv2d(be, [x], [w], b, k, 1, splits=sp, precision=1, flags=flags, bf16_src=(0,))
        close(got, npo.conv2d_same(R(x), R(w), b, 1), 5e-5)
        if sp == 1 and k == 5:      # (3x3: fp32 sources stay on the first generation's kernel, another summation order)
            assert np.array_equal(got, KH.conv2d(be, [x], [w], b, k, 1, precision=1, flags=flags))
    close(KH.conv2d(be, [xa, xb], [wa, wb], None, 3, 1, precision=1, flags=flags, bf16_src=(0, 1)),
          npo.conv2d_same(R(xa), R(wa)) + npo.conv2d_same(R(xb), R(wb)), 5e-5)
    with pytest.raises(RuntimeError):                             # mixed element types in one launch are rejected
        KH.conv2d(be, [xa, xb], [wa, wb], None, 3, 1, precision=1, flags=flags, bf16_src=(1,))
    # fused step on the bf16 tape: h in as bf16, h / gates out as bf16 as well; then the thin image as a centre-tap chunk
    x, h, c = rnd(1, 16, 32, 8), rnd(1, 16, 32, F, scale=0.5), rnd(1, 16, 32, F)
    ker, rec, b = rnd(5, 5, 8, 4 * F, scale=0.3), rnd(5, 5, F, 4 * F, scale=0.1), rnd(4 * F, scale=0.5)
    h0, c0, g0 = KH.convlstm_step_fused(be, x, h, c, ker, rec, b, precision=1, flags=flags)
    hg, cg, g16, h16 = KH.convlstm_step_tape16(be, x, h, c, ker, rec, b, flags=flags)
    assert np.array_equal(hg, h0) and np.array_equal(cg, c0)
    assert np.array_equal(g16, R(g0)) and np.array_equal(h16, R(h0))
    for (k, cin) in [(5, 1), (3, 3)]:
        x, h, c = rnd(2, 16, 32, cin), rnd(2, 16, 32, F, scale=0.5), rnd(2, 16, 32, F)
        ker, rec, b = rnd(k, k, cin, 4 * F, scale=0.3), rnd(k, k, F, 4 * F, scale=0.1), rnd(4 * F, scale=0.5)
        h1, c1 = npo.convlstm_step(R(x), R(h), c, R(ker), R(rec), b)
        hg, cg, g16, h16 = KH.convlstm_step_tape16(be, x, h, c, ker, rec, b, center=True, flags=flags)
        close(hg, h1, 3e-5)
        close(cg, c1, 3e-5)
        assert np.array_equal(h16, R(hg))
        z = npo.conv2d_same(R(x), R(ker), b) + npo.conv2d_same(R(h), R(rec))
        assert np.abs(g16[..., 2 * F:3 * F] - np.tanh(z[..., 2 * F:3 * F])).max() <= 2.0 ** -8


def test_conv_bf16_narrow_blocks(be):
    """N = 32 / 64 stride-1 3x3 / 5x5 layers in bf16 mode (the decoder tail): narrow blocks of the halo fragment
    kernel -- 8 / 4 row groups of waves over 1 / 2 column fragments of the same 8 x 32 patch.  Against the oracle on
    bf16-rounded operands, and against the gather kernel (LU_CONV_F_NO_NARROW), which walks another (chunk, tap) order:
    summation-order noise only.  Ragged frames, several chunks, two sources, K split, bf16 tensors as sources."""
    R = KH.bf16_round
    for (fr, H, W, Cc, N, k, sp) in [(1, 16, 32, 32, 32, 3, 1), (2, 9, 37, 64, 32, 3, 1), (1, 17, 40, 100, 64, 3, 1),
                                     (1, 8, 33, 36, 64, 5, 1), (1, 11, 30, 20, 32, 5, 1), (1, 16, 32, 256, 64, 3, 2)]:
        x, w, b = rnd(fr, H, W, Cc), rnd(k, k, Cc, N, scale=0.2), rnd(N)
        got = KH.conv2d(be, [x], [w], b, k, 1, splits=sp, precision=1)
        close(got, npo.conv2d_same(R(x), R(w), b, 1), 5e-5)
        close(got, KH.conv2d(be, [x], [w], b, k, 1, splits=sp, precision=1, flags=cabi.LU_CONV_F_NO_NARROW), 5e-5)
        if Cc % 8 == 0:
            assert np.array_equal(got, KH.conv2d(be, [x], [w], b, k, 1, splits=sp, precision=1, bf16_src=(0,)))
    xa, xb = rnd(2, 12, 34, 40), rnd(2, 12, 34, 24)                # two sources (UpBlock concat), N = 64
    wa, wb = rnd(3, 3, 40, 64, scale=0.1), rnd(3, 3, 24, 64, scale=0.1)
    ref = npo.conv2d_same(R(xa), R(wa)) + npo.conv2d_same(R(xb), R(wb))
    close(KH.conv2d(be, [xa, xb], [wa, wb], None, 3, 1, precision=1), ref, 5e-5)


def test_conv_s2_fwd_bf16(be):
    """bf16 mode: the stride-2 3x3 forward convolution (Networks.py:52-56) reading the bf16 copy of the ConvLSTM output, input
    pixels staged by column parity.  Against the oracle on bf16-rounded operands and against the gather kernel it replaces;
    ragged tiles, several chunks, N beyond one 128-column tile, channel counts that are not multiples of 32."""
    R = KH.bf16_round
    for (fr, H, W, Cc, N) in [(1, 8, 64, 32, 32), (2, 10, 74, 64, 96), (1, 6, 32, 40, 160), (1, 16, 128, 128, 128)]:
        x, w, b = rnd(fr, H, W, Cc), rnd(3, 3, Cc, N, scale=0.2), rnd(N)
        got = KH.conv2d_s2_fwd_bf16(be, x, w, b)
        close(got, npo.conv2d_same(R(x), R(w), b, 2), 5e-5)
        close(got, KH.conv2d(be, [x], [w], b, 3, 2, precision=1), 5e-5)


def test_conv_s2_dgrad_fused_bf16(be):
    """bf16 mode: input gradient of the stride-2 3x3 layers (Networks.py:52-56) with all four output parity classes in one
    launch.  Against torch's gradient on bf16-rounded operands (summation order only), and against the four parity-plane
    launches it replaces.  Ragged tiles (Wd not a multiple of 32, odd Hd), several chunks, C beyond one 128-column tile."""
    R = KH.bf16_round
    for (fr, Hd, Wd, Cc, N) in [(1, 4, 32, 32, 32), (2, 5, 37, 64, 96), (1, 3, 16, 160, 64), (1, 8, 64, 128, 128)]:
        H, W = 2 * Hd, 2 * Wd
        w, dy = rnd(3, 3, Cc, N, scale=0.2), rnd(fr, Hd, Wd, N)
        gx, _ = _torch_conv_grads(rnd(fr, H, W, Cc), R(w), R(dy), 2)
        got = KH.conv2d_dgrad_s2_fused_bf16(be, dy, w)
        close(got, gx, 5e-5)
        full, _ = _torch_conv_grads(rnd(fr, H, W, Cc), w, dy, 2)
        close(KH.conv2d_dgrad_s2_parity(be, dy, w, (H, W)), full, 5e-5)      # (the fp32 four-plane form, for reference)
        assert np.abs(got - full).max() <= 2.0 ** -6 * np.abs(full).max()


def test_conv_bf16_gather_variant(be):
    """General bf16 kernel (precision = 1 outside the halo kernel's domain): stride 2 (TF-SAME asymmetric pads), 1x1 and
    7x7 kernels, narrow / ragged outputs, two sources, K split, strided output rows."""
    R = KH.bf16_round
    for (fr, H, W, Cc, N, k, s_, sp) in [(1, 9, 9, 4, 70, 3, 2, 1), (1, 8, 8, 24, 130, 3, 2, 1), (2, 6, 6, 8, 3, 1, 1, 1),
                                         (3, 5, 5, 4, 33, 5, 2, 1), (1, 7, 8, 36, 40, 3, 1, 2), (1, 11, 9, 8, 20, 7, 1, 1),
                                         (1, 16, 16, 48, 40, 3, 1, 3)]:
        x, w, b = rnd(fr, H, W, Cc), rnd(k, k, Cc, N, scale=0.2), rnd(N)
        got = KH.conv2d(be, [x], [w], b, k, s_, splits=sp, precision=1)
        close(got, npo.conv2d_same(R(x), R(w), b, s_), 5e-5)
    xa, xb = rnd(2, 6, 7, 12), rnd(2, 6, 7, 4)                     # two sources, narrow output
    wa, wb = rnd(3, 3, 12, 20, scale=0.2), rnd(3, 3, 4, 20, scale=0.2)
    ref = npo.conv2d_same(R(xa), R(wa)) + npo.conv2d_same(R(xb), R(wb))
    close(KH.conv2d(be, [xa, xb], [wa, wb], None, 3, 1, precision=1), ref, 5e-5)
    with pytest.raises(RuntimeError):                             # input dilation (the zero-dilated dgrad form) stays fp32
        KH.conv2d(be, [rnd(1, 8, 8, 8)], [rnd(3, 3, 8, 8)], None, 3, 1, dil=2, pad=(1, 1), out_hw=(8, 8), precision=1)


def test_conv_two_sources_and_strided_views(be):
    """UpBlock2D concat([up, skip]) (Networks.py:145) as two sources reading channel slices."""
    xa, xb = rnd(2, 6, 7, 12), rnd(2, 6, 7, 1)
    w = rnd(3, 3, 13, 20, scale=0.2)
    b = rnd(20)
    ref = npo.conv2d_same(np.concatenate([xa, xb], -1), w, b, 1)
    xad, xbd, wd, bd = be.dev(xa), be.dev(xb), be.dev(w), be.dev(b)
    srcs = [calls.conv_src(be.ptr(xad), 6 * 7 * 12, 12, 12, be.ptr(wd), 13 * 20, 20),
            calls.conv_src(be.ptr(xbd), 6 * 7, 1, 1, be.ptr(wd, 12 * 20), 13 * 20, 20)]
    out = be.empty((2, 6, 7, 20))
    calls.conv2d(be.lib, be.stream, srcs, 2, 6, 7, 6, 7, 3, 1, 1, 1, 1, 20, be.ptr(bd), be.ptr(out), 6 * 7 * 20, 20)
    close(be.host(out), ref, 5e-5)
    # a source that is a channel slice of a wider activation tensor (pix_stride > C)
    wide = rnd(2, 6, 7, 24)
    w2 = rnd(3, 3, 8, 16, scale=0.2)
    wided, w2d = be.dev(wide), be.dev(w2)
    srcs = [calls.conv_src(be.ptr(wided, 8), 6 * 7 * 24, 24, 8, be.ptr(w2d), 8 * 16, 16)]
    out = be.empty((2, 6, 7, 16))
    calls.conv2d(be.lib, be.stream, srcs, 2, 6, 7, 6, 7, 3, 1, 1, 1, 1, 16, None, be.ptr(out), 6 * 7 * 16, 16)
    close(be.host(out), npo.conv2d_same(wide[..., 8:16], w2, None, 1), 5e-5)


def _torch_conv_grads(x, w, dy, stride):
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    y = tho.conv2d_same(xt, wt, None, stride)
    gx, gw = torch.autograd.grad(y, [xt, wt], torch.tensor(dy, dtype=torch.float64))
    return gx.numpy(), gw.numpy()


@pytest.mark.parametrize('case', [(2, 8, 9, 8, 12, 3, 1), (1, 7, 7, 4, 8, 5, 1), (2, 8, 8, 8, 16, 3, 2),
                                  (1, 9, 7, 12, 8, 3, 2), (1, 8, 8, 4, 4, 5, 2), (1, 6, 6, 3, 5, 3, 1),
                                  (1, 6, 5, 8, 3, 1, 1), (1, 7, 9, 72, 24, 3, 1), (1, 8, 8, 132, 8, 3, 2), (2, 7, 9, 8, 12, 5, 2),
                                  (1, 9, 8, 4, 72, 3, 2)])
def test_conv_dgrad_wgrad(be, case):
    fr, H, W, Cc, N, k, s = case
    x, w = rnd(fr, H, W, Cc), rnd(k, k, Cc, N, scale=0.3)
    Ho, Wo = calls.same_pad(H, k, s)[0], calls.same_pad(W, k, s)[0]
    dy = rnd(fr, Ho, Wo, N)
    gx, gw = _torch_conv_grads(x, w, dy, s)
    close(KH.conv2d_dgrad(be, dy, w, (H, W), s), gx, 5e-5)
    if s == 2:      # the parity-plane formulation the engine uses (no zero multiplications)
        close(KH.conv2d_dgrad_s2_parity(be, dy, w, (H, W)), gx, 5e-5)
    close(KH.conv2d_wgrad(be, x, dy, k, s, splits=1), gw, 1e-4)
    close(KH.conv2d_wgrad(be, x, dy, k, s, splits=3), gw, 1e-4)


def test_randomized_conv_shapes(be):
    """Seeded sweep over ragged shapes: kernel sizes 1/3/5/7, strides 1/2, 1..72 input channels, 1..136 output columns,
    K splits, all three weight layouts (plain fp32, fragment-packed fp32, bf16) for the forward convolution; input and
    weight gradients for the same layer.  (An offline 2000-case run of this generator found no mismatch.)"""
    rng = np.random.default_rng(2026)
    R = KH.bf16_round
    n_cases = 40 if be.name == 'emu' else 120
    for _ in range(n_cases):
        k, s_ = int(rng.choice([1, 3, 5, 7])), int(rng.choice([1, 2]))
        fr, H, W = int(rng.integers(1, 3)), int(rng.integers(3, 14)), int(rng.integers(3, 36))
        Cc = int(rng.choice([1, 3, 4, 8, 20, 33, 64, 72]))
        N = int(rng.choice([2, 3, 8, 24, 33, 70, 96, 136]))
        sp = int(rng.choice([1, 1, 2, 3]))
        prec = int(rng.choice([0, 0, 1, 1]))
        if Cc % 4 != 0:
            prec = 0
        x = f32(rng.standard_normal((fr, H, W, Cc)))
        w = f32(rng.standard_normal((k, k, Cc, N)) * 0.2)
        b = f32(rng.standard_normal(N))
        tag = (k, s_, fr, H, W, Cc, N, sp, prec)
        got = KH.conv2d(be, [x], [w], b, k, s_, splits=sp, precision=prec)
        ref = npo.conv2d_same(R(x), R(w), b, s_) if prec == 1 else npo.conv2d_same(x, w, b, s_)
        assert np.abs(got - ref).max() <= 1e-4, ('fwd', tag)
        if rng.random() < 0.4:
            Ho, Wo = calls.same_pad(H, k, s_)[0], calls.same_pad(W, k, s_)[0]
            dy = f32(rng.standard_normal((fr, Ho, Wo, N)))
            gx, gw = _torch_conv_grads(x, w, dy, s_)
            assert np.abs(KH.conv2d_wgrad(be, x, dy, k, s_, splits=sp) - gw).max() <= 3e-4 * max(1.0, np.abs(gw).max()), ('wgrad', tag)
            assert np.abs(KH.conv2d_dgrad(be, dy, w, (H, W), s_) - gx).max() <= 1e-4 * max(1.0, np.abs(gx).max()), ('dgrad', tag)


def test_wgrad_wide_channels_and_beta(be):
    x, dy = rnd(1, 6, 6, 132), rnd(1, 6, 6, 136)
    _, gw = _torch_conv_grads(x, rnd(3, 3, 132, 136), dy, 1)
    dw0 = rnd(3, 3, 132, 136)
    close(KH.conv2d_wgrad(be, x, dy, 3, 1, splits=2, dw0=dw0, beta=1.0), gw + dw0, 2e-4)
    x, dy = rnd(1, 6, 6, 40), rnd(1, 6, 6, 8)
    _, gw = _torch_conv_grads(x, rnd(3, 3, 40, 8), dy, 1)
    close(KH.conv2d_wgrad(be, x, dy, 3, 1), gw, 1e-4)
    for (fr, H, W, Cc, N, k) in [(2, 5, 16, 72, 136, 3), (1, 4, 32, 64, 128, 5), (1, 3, 16, 132, 72, 5)]:
        x, dy = rnd(fr, H, W, Cc), rnd(fr, H, W, N)         # kernel-row variant (W % 16 == 0, C >= 64)
        _, gw = _torch_conv_grads(x, rnd(k, k, Cc, N), dy, 1)
        close(KH.conv2d_wgrad(be, x, dy, k, 1, splits=1), gw, 2e-4)
        close(KH.conv2d_wgrad(be, x, dy, k, 1, splits=3), gw, 2e-4)
    x, dy = rnd(2, 5, 7, 72), rnd(2, 5, 7, 264)            # 128x256 tile variant (ConvLSTM-sized N), ragged edges
    _, gw = _torch_conv_grads(x, rnd(3, 3, 72, 264), dy, 1)
    close(KH.conv2d_wgrad(be, x, dy, 3, 1, splits=2), gw, 2e-4)


@pytest.mark.parametrize('precision', [0, 1])
def test_wgrad_fused_bias_gradient(be, precision):
    """dbias = column sums of dy, produced on the side by the kernel-row weight-gradient kernels (exact fp32 sums in
    both precisions, fixed order); other layer shapes must reject the request (callers use lu_colsum there)."""
    for (fr, H, W, Cc, N, k, sp) in [(2, 5, 32, 72, 136, 3, 3), (1, 4, 64, 128, 128, 5, 2)]:
        x, dy = rnd(fr, H, W, Cc), rnd(fr, H, W, N)
        db0 = rnd(N)
        dw, db = KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, precision=precision, dbias0=db0, dbias_beta=1.0)
        close(db, dy.reshape(-1, N).astype(np.float64).sum(0) + db0, 2e-4)
        dw2, db2 = KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, precision=precision, dbias0=db0, dbias_beta=0.0)
        close(db2, dy.reshape(-1, N).astype(np.float64).sum(0), 2e-4)
        assert np.array_equal(dw, KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, precision=precision))
    with pytest.raises(RuntimeError):
        KH.conv2d_wgrad(be, rnd(1, 6, 6, 8), rnd(1, 6, 6, 8), 3, 1, dbias0=rnd(8))


def test_wgrad_kernel_row_ragged_widths(be):
    """fp32 kernel-row weight gradient on widths that are NOT multiples of 16 (config-4's 248- / 124-pixel levels; real CTC
    crops): pixel slabs of whole rows, ceil(W / 16) runs per row with a masked tail.  Against the fp64 reference, against
    the one-tap-per-block kernel (LU_WGRAD_F_NO_RAGGED) and with the bias gradient on the side; slab counts that do / do not
    divide the row count, more slabs than rows."""
    nr = cabi.LU_WGRAD_F_NO_RAGGED
    for (fr, H, W, Cc, N, k, sp) in [(2, 5, 40, 72, 136, 3, 3), (1, 7, 44, 64, 128, 5, 2), (3, 3, 56, 128, 72, 5, 4),
                                     (1, 2, 124, 64, 40, 3, 5)]:
        x, dy = rnd(fr, H, W, Cc), rnd(fr, H, W, N)
        _, ref = _torch_conv_grads(x, rnd(k, k, Cc, N), dy, 1)
        db0 = rnd(N)
        dw, db = KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, dbias0=db0, dbias_beta=1.0)
        close(dw, ref, 2e-4)
        close(db, dy.reshape(-1, N).astype(np.float64).sum(0) + db0, 2e-4)
        close(KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, flags=nr), ref, 2e-4)
        dw0 = rnd(k, k, Cc, N)
        close(KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, dw0=dw0, beta=1.0), ref + dw0, 2e-4)
    with pytest.raises(RuntimeError):      # the one-tap-per-block kernel has no bias gradient on the side
        KH.conv2d_wgrad(be, rnd(1, 4, 40, 64), rnd(1, 4, 40, 64), 3, 1, dbias0=rnd(64), flags=nr)


def test_wgrad_kernel_row_32_pixel_stages(be):
    """fp32 kernel-row weight gradient with 32-pixel stages (W % 32 == 0: two loader passes per stage, half the block-wide barriers
    per MFMA -- the library's own choice since round 4) against the oracle: 5x5 and 3x3, pixel splits, a masked channel tail, the bias
    gradient with beta.  (Until round 6 the 16-pixel and the re-reading instances were kept for bit-identity A/Bs; the 16-pixel
    instance still runs every width with W % 32 != 0: test_wgrad_fused_bias_gradient, test_wgrad_kernel_row_ragged_widths.)"""
    cases = [(2, 5, 32, 72, 136, 3), (1, 4, 64, 64, 128, 2), (1, 3, 96, 132, 72, 1), (3, 2, 32, 64, 520, 4)]
    if be.name == 'emu':          # (the host emulator runs a subset: the CPU suite has to stay within minutes)
        cases = cases[:1] + cases[2:3]
    for (fr, H, W, Cc, N, sp) in cases:
        x, dy = rnd(fr, H, W, Cc), rnd(fr, H, W, N)
        _, ref = _torch_conv_grads(x, rnd(5, 5, Cc, N), dy, 1)
        db0 = rnd(N)
        dw, db = KH.conv2d_wgrad(be, x, dy, 5, 1, splits=sp, dbias0=db0, dbias_beta=1.0)
        close(dw, ref, 2e-4)
        close(db, dy.reshape(-1, N).astype(np.float64).sum(0) + db0, 2e-4)
        n3 = min(N, 128)
        x3, dy3 = np.ascontiguousarray(x[..., :64]), np.ascontiguousarray(dy[..., :n3])
        _, ref3 = _torch_conv_grads(x3, rnd(3, 3, 64, n3), dy3, 1)
        dw3, db3 = KH.conv2d_wgrad(be, x3, dy3, 3, 1, splits=sp, dbias0=db0[:n3], dbias_beta=1.0)
        close(dw3, ref3, 2e-4)
        close(db3, dy3.reshape(-1, n3).astype(np.float64).sum(0) + db0[:n3], 2e-4)


def test_wgrad_all_taps_narrow_layers(be):
    """wgrad_small3_kernel: stride-1 3x3 layers with C <= 64 and N <= 64 (W % 16 == 0) -- all nine taps per block, ragged
    channel counts, pixel splits, fused bias gradient."""
    for (fr, H, W, Cc, N, sp) in [(2, 5, 16, 32, 32, 1), (1, 4, 32, 64, 64, 3), (1, 3, 16, 20, 36, 2), (2, 4, 16, 64, 32, 2),
                                  (1, 5, 48, 8, 64, 4)]:
        x, dy = rnd(fr, H, W, Cc), rnd(fr, H, W, N)
        _, gw = _torch_conv_grads(x, rnd(3, 3, Cc, N), dy, 1)
        db0 = rnd(N)
        dw, db = KH.conv2d_wgrad(be, x, dy, 3, 1, splits=sp, dbias0=db0, dbias_beta=1.0)
        close(dw, gw, 2e-4)
        close(db, dy.reshape(-1, N).astype(np.float64).sum(0) + db0, 2e-4)
        dw0 = rnd(3, 3, Cc, N)
        close(KH.conv2d_wgrad(be, x, dy, 3, 1, splits=sp, dw0=dw0, beta=1.0), gw + dw0, 2e-4)


def test_wgrad_two_phase_call(be):
    """phase 1 (partial sums) + phase 2 (reduce) of lu_conv2d_wgrad == the one-call default, bit for bit."""
    x, dy = rnd(2, 5, 16, 72), rnd(2, 5, 16, 136)
    Cin, N, k = 72, 136, 3
    whole = KH.conv2d_wgrad(be, x, dy, k, 1, splits=3)
    dw = be.empty((k, k, Cin, N))
    xd, dyd = be.dev(x), be.dev(dy)
    d = calls.wgrad_desc(be.ptr(xd), 5 * 16 * Cin, Cin, Cin, be.ptr(dyd), 5 * 16 * N, N, N, 2, 5, 16, 5, 16, k, 1, 1, 1,
                         be.ptr(dw), Cin * N, N, 3, 0.0)
    ws = be.empty((be.lib.lu_conv2d_wgrad_workspace_bytes(C.byref(d)) // 4 + 4,))
    d.workspace = be.ptr(ws)
    for phase in (1, 2):
        d.phase = phase
        ck(be, be.lib.lu_conv2d_wgrad(C.byref(d), be.stream), 'wgrad phase %d' % phase)
    assert np.array_equal(be.host(dw), whole)


@pytest.mark.parametrize('ct', ['64', '128'])
def test_wgrad_bf16_mfma_variant(be, ct):
    _wgrad_bf16_cases(be, cabi.LU_WGRAD_F_CT64 if ct == '64' else cabi.LU_WGRAD_F_CT128)      # 64- / 128-channel block tiles


def _wgrad_bf16_cases(be, flags=0):
    """Kernel-row weight gradient on the bf16 MFMA (precision = 1, W % 32 == 0): x and dy rounded to bf16, fp32
    accumulation -- exact up to summation order against the fp32 reference on bf16-rounded operands.  Shapes the bf16
    kernel does not cover (W % 32 != 0) silently stay on the fp32 kernels."""
    R = KH.bf16_round
    for (fr, H, W, Cc, N, k, sp) in [(2, 5, 32, 72, 136, 3, 1), (1, 4, 64, 64, 128, 5, 3), (1, 3, 32, 132, 72, 5, 2)]:
        x, dy = rnd(fr, H, W, Cc), rnd(fr, H, W, N)
        _, gw = _torch_conv_grads(R(x), rnd(k, k, Cc, N), R(dy), 1)
        got = KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, precision=1, flags=flags)
        close(got, gw, 2e-4)
        _, gfull = _torch_conv_grads(x, rnd(k, k, Cc, N), dy, 1)
        assert np.abs(got - gfull).max() > 0 and np.abs(got - gfull).max() <= 2.0 ** -6 * np.abs(gfull).max()
    x, dy = rnd(1, 4, 16, 64), rnd(1, 4, 16, 128)             # W % 32 != 0 -> fp32 kernel, full precision
    _, gw = _torch_conv_grads(x, rnd(3, 3, 64, 128), dy, 1)
    close(KH.conv2d_wgrad(be, x, dy, 3, 1, precision=1), gw, 2e-4)
    # bf16 TENSORS as operands (h sequence / dz of the bf16 tape): identical arithmetic to rounding while staging; odd stage
    # counts and slab counts that are not multiples of 8 exercise the zero-filled tail stages and the XCD-aware numbering
    for (fr, H, W, Cc, N, k, sp, xb, yb) in [(2, 5, 32, 72, 136, 3, 3, True, True), (1, 4, 64, 64, 128, 5, 2, False, True),
                                             (1, 3, 32, 136, 72, 5, 1, True, False), (3, 3, 32, 64, 64, 5, 9, True, True)]:
        x, dy = rnd(fr, H, W, Cc), rnd(fr, H, W, N)
        base, db0 = KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, precision=1, flags=flags, dbias0=np.zeros(N, np.float32))
        got, db = KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, precision=1, flags=flags, x_bf16=xb, dy_bf16=yb,
                                  dbias0=np.zeros(N, np.float32))
        assert np.array_equal(got, base)
        close(db, (R(dy) if yb else dy).reshape(-1, N).sum(0), 2e-4)
        close(db0, dy.reshape(-1, N).sum(0), 2e-4)
    for (fr, H, W, Cc, N, sp, xb, yb, k2) in [(2, 6, 64, 72, 136, 2, True, False, 3), (1, 5, 64, 64, 64, 1, False, False, 3),
                                              (1, 8, 128, 128, 72, 3, True, True, 3), (2, 6, 64, 72, 136, 2, True, True, 5),
                                              (1, 7, 64, 128, 64, 1, False, True, 5)]:      # stride-2 3x3 / 5x5 (TF-SAME: odd H too)
        x, dy = rnd(fr, H, W, Cc), rnd(fr, (H + 1) // 2, W // 2, N)
        _, gw = _torch_conv_grads(R(x), rnd(k2, k2, Cc, N), R(dy), 2)
        got, db = KH.conv2d_wgrad(be, x, dy, k2, 2, splits=sp, precision=1, flags=flags, x_bf16=xb, dy_bf16=yb,
                                  dbias0=np.zeros(N, np.float32))
        close(got, gw, 2e-4)
        close(db, (R(dy) if yb else dy).reshape(-1, N).sum(0), 2e-4)
    x, dy = rnd(2, 4, 32, 32), rnd(2, 4, 32, 72)              # 1x1 problem over 32 rows: the im2col chunk of a thin input
    _, gw = _torch_conv_grads(R(x), rnd(1, 1, 32, 72), R(dy), 1)
    close(KH.conv2d_wgrad(be, x, dy, 1, 1, splits=2, precision=1, x_bf16=True, dy_bf16=True), gw, 2e-4)
    # the narrow decoder layers (C = 32 / 64 / 36, N = 32 / 64): masked channel / column tiles of the same kernel; with
    # LU_WGRAD_F_NO_NARROW_BF16 they stay on the fp32 all-taps kernel (full precision) -- the two differ by the operand rounding
    for (fr, H, W, Cc, N, k, sp) in [(2, 6, 32, 32, 32, 3, 2), (1, 5, 64, 64, 32, 3, 1), (1, 4, 32, 64, 64, 3, 3),
                                     (1, 6, 32, 36, 64, 5, 2)]:
        x, dy = rnd(fr, H, W, Cc), rnd(fr, H, W, N)
        _, gw = _torch_conv_grads(R(x), rnd(k, k, Cc, N), R(dy), 1)
        got, db = KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, precision=1, flags=flags, dbias0=np.zeros(N, np.float32))
        close(got, gw, 2e-4)
        close(db, dy.reshape(-1, N).sum(0), 2e-4)
        _, gfull = _torch_conv_grads(x, rnd(k, k, Cc, N), dy, 1)
        if k == 3:
            close(KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, precision=1, flags=flags | cabi.LU_WGRAD_F_NO_NARROW_BF16), gfull, 2e-4)
    with pytest.raises(RuntimeError):                         # bf16 operands need the bf16 kernel-row variant
        KH.conv2d_wgrad(be, rnd(1, 4, 16, 64), rnd(1, 4, 16, 128), 3, 1, precision=1, dy_bf16=True)


def test_wgrad_bf16_lds_dma_equals_register_staging(be):
    """The all-taps 3x3 form on bf16 operands reaches LDS by global_load_lds (dense rows, 64-byte column segments XOR-swizzled by the
    row; three stage buffers, counted vmcnt waits) instead of through staging registers and ds_write (LU_WGRAD_F_NO_DMA, the
    previous form): same tiles, same fragments, same MFMA order -- bit-identical weight AND bias gradients, and both against the
    oracle.  198- / 102-row x tiles (the swizzle term changes with the kernel row), 64- / 32-pixel stages, masked channel / column
    tails, image borders, odd slab counts.  (The 5x5 kernel-row form's DMA instances measured -3 % / -1.5 % and left in round 6.)"""
    cases = [(2, 5, 64, 136, 72, 3, 3), (1, 7, 32, 64, 128, 3, 1), (1, 4, 64, 72, 264, 3, 2)]
    if be.name == 'emu':          # (the host emulator runs a subset: the CPU suite has to stay within minutes)
        cases = [(2, 3, 64, 72, 72, 3, 3), (1, 4, 32, 64, 128, 3, 1)]
    for (fr, H, W, Cc, N, k, sp) in cases:
        x, dy = rnd(fr, H, W, Cc), rnd(fr, H, W, N)
        want, db0 = KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, precision=1, x_bf16=True, dy_bf16=True,
                                    dbias0=np.zeros(N, np.float32), flags=cabi.LU_WGRAD_F_NO_DMA)
        got, db = KH.conv2d_wgrad(be, x, dy, k, 1, splits=sp, precision=1, x_bf16=True, dy_bf16=True, dbias0=np.zeros(N, np.float32))
        assert np.array_equal(got, want), (Cc, N, k, sp)
        assert np.array_equal(db, db0), (Cc, N, k, sp)
        _, gw = _torch_conv_grads(KH.bf16_round(x), rnd(k, k, Cc, N), KH.bf16_round(dy), 1)
        close(got, gw, 2e-4)


@pytest.mark.parametrize('form', ['w8'])
def test_wgrad_bf16_all_taps_form_equals_the_kernel_row_form(be, form):
    """The all-taps form of the 3x3 layers (round 4): one block accumulates all nine taps of a 64-channel x 128-column tile (x tile = three
    input rows, dy tile shared by nine taps) on 8 waves.  Every (tap, c, n) sum still
    walks the pixels of its slab in the same 16-pixel MFMA steps: bit-identical to the kernel-row form, bias gradient included;
    operand types, ragged channel / column counts, top / bottom image rows, odd stage and slab counts."""
    fl = 0      # (the library's own choice = the 8-wave all-taps form; LU_WGRAD_F_CT64 below = the kernel-row form)
    R = KH.bf16_round
    for (fr, H, W, Cc, N, sp, xb, yb) in [(2, 5, 32, 72, 136, 3, True, True), (1, 4, 64, 64, 128, 1, True, True),
                                          (3, 3, 32, 128, 64, 9, True, True), (1, 6, 32, 136, 264, 2, True, False),
                                          (2, 5, 128, 72, 136, 3, True, True),
                                          (2, 3, 64, 64, 72, 2, False, True), (1, 5, 32, 96, 128, 1, False, False)]:
        x, dy = rnd(fr, H, W, Cc), rnd(fr, H, W, N)
        want, db0 = KH.conv2d_wgrad(be, x, dy, 3, 1, splits=sp, precision=1, x_bf16=xb, dy_bf16=yb, dbias0=np.zeros(N, np.float32),
                                    flags=cabi.LU_WGRAD_F_CT64)
        got, db = KH.conv2d_wgrad(be, x, dy, 3, 1, splits=sp, precision=1, x_bf16=xb, dy_bf16=yb, dbias0=np.zeros(N, np.float32),
                                  flags=fl)
        assert np.array_equal(got, want), (form, Cc, N, sp)
        if W % 64 == 0:      # (64-pixel stages where the width allows: the same pixels in the same 16-pixel steps)
            assert np.array_equal(KH.conv2d_wgrad(be, x, dy, 3, 1, splits=sp, precision=1, x_bf16=xb, dy_bf16=yb,
                                                  flags=fl | cabi.LU_WGRAD_F_PRB32), want)
        close(db, db0, 4e-5)      # (fp32 column sums in another order: one ulp of a sum of ~1e4 terms is 1.5e-5 of its maximum)
        _, gw = _torch_conv_grads(R(x), rnd(3, 3, Cc, N), R(dy), 1)
        close(got, gw, 2e-4)
        close(db, (R(dy) if yb else dy).reshape(-1, N).sum(0), 2e-4)


@pytest.mark.parametrize('k,cin', [(3, 8), (5, 1)])
def test_convlstm_fused_step(be, k, cin):
    """Fused two-source conv + gate epilogue == Keras ConvLSTM2D cell step."""
    F = 32
    x, h, c = rnd(2, 6, 7, cin), rnd(2, 6, 7, F, scale=0.5), rnd(2, 6, 7, F)
    ker, rec, b = rnd(k, k, cin, 4 * F, scale=0.3), rnd(k, k, F, 4 * F, scale=0.1), rnd(4 * F, scale=0.5)
    h1, c1 = npo.convlstm_step(x, h, c, ker, rec, b)
    hg, cg, gates = KH.convlstm_step_fused(be, x, h, c, ker, rec, b)
    close(hg, h1, 2e-5)
    close(cg, c1, 2e-5)
    z = npo.conv2d_same(x, ker, b) + npo.conv2d_same(h, rec)
    close(gates[..., :F], npo.hard_sigmoid(z[..., :F]), 2e-5)
    close(gates[..., 2 * F:3 * F], np.tanh(z[..., 2 * F:3 * F]), 2e-5)


def test_lstm_gates_pointwise_fwd_bwd(be):
    fr, H, W, F = 2, 4, 5, 6
    z, c0 = rnd(fr, H, W, 4 * F, scale=2.0), rnd(fr, H, W, F)
    zd, c0d = be.dev(z), be.dev(c0)
    c1 = be.empty(c0.shape)
    hseq = be.empty((fr, 3, H, W, F))   # h written into slot t=1 of a [B,T,...] buffer
    gates = be.empty(z.shape)
    ck(be, be.lib.lu_lstm_gates_fwd(be.ptr(zd), be.ptr(c0d), be.ptr(c1), be.ptr(hseq, H * W * F), be.ptr(gates), fr,
                                    H * W, F, 3 * H * W * F, be.stream), 'gates_fwd')
    zt = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    ct = torch.tensor(c0, dtype=torch.float64, requires_grad=True)
    i, f, g, o = [zt[..., j * F:(j + 1) * F] for j in range(4)]
    cn = tho.hard_sigmoid(f) * ct + tho.hard_sigmoid(i) * torch.tanh(g)
    hn = tho.hard_sigmoid(o) * torch.tanh(cn)
    close(be.host(c1), cn.detach().numpy(), 1e-5)
    close(be.host(hseq)[:, 1], hn.detach().numpy(), 1e-5)
    dh, dh2, dc = rnd(fr, H, W, F), rnd(fr, H, W, F), rnd(fr, H, W, F)
    gz, gc = torch.autograd.grad([hn, cn], [zt, ct], [torch.tensor(dh + dh2, dtype=torch.float64),
                                                      torch.tensor(dc, dtype=torch.float64)])
    dz, dcp = be.empty(z.shape), be.empty(c0.shape)
    dhd, dh2d, dcd = be.dev(dh), be.dev(dh2), be.dev(dc)
    ck(be, be.lib.lu_lstm_gates_bwd(be.ptr(gates), be.ptr(c0d), be.ptr(c1), be.ptr(dhd), H * W * F, be.ptr(dh2d),
                                    be.ptr(dcd), be.ptr(dz), be.ptr(dcp), fr, H * W, F, be.stream), 'gates_bwd')
    close(be.host(dz), gz.numpy(), 1e-5)
    close(be.host(dcp), gc.numpy(), 1e-5)


def test_bn_lrelu_fwd_bwd(be):
    rows, Cc = 300, 20
    x = rnd(rows, Cc, scale=2.0) + 0.5
    gamma, beta = f32(1 + 0.2 * RNG.random(Cc)), rnd(Cc, scale=0.3)
    mm, mv = rnd(Cc, scale=0.1), f32(1 + RNG.random(Cc))
    xd, gd, bd = be.dev(x), be.dev(gamma), be.dev(beta)
    ws = be.empty((be.lib.lu_colreduce_workspace_bytes(rows, Cc) // 8 + 1,), np.float64)
    sums = be.empty((2 * Cc,), np.float64)
    ck(be, be.lib.lu_bn_stats(be.ptr(xd), rows, Cc, be.ptr(sums), be.ptr(ws), be.stream), 'stats')
    close(be.host(sums)[:Cc], x.astype(np.float64).sum(0), 1e-3)
    scale, shift, smean, sinv = [be.empty((Cc,)) for _ in range(4)]
    mm2, mv2 = be.dev(mm), be.dev(mv)
    ck(be, be.lib.lu_bn_finalize_train(be.ptr(sums), float(rows), be.ptr(gd), be.ptr(bd), 1e-3, 0.99, be.ptr(mm2),
                                       be.ptr(mv2), be.ptr(scale), be.ptr(shift), be.ptr(smean), be.ptr(sinv), Cc,
                                       be.stream), 'fin')
    y = be.empty(x.shape)
    ck(be, be.lib.lu_bn_lrelu_apply(be.ptr(xd), be.ptr(y), be.ptr(scale), be.ptr(shift), 0.3, rows, Cc, be.stream),
       'apply')
    yr, mean, var = npo.batchnorm_train(x.reshape(1, 1, rows, Cc), gamma, beta)
    close(be.host(y), npo.leaky_relu(yr).reshape(rows, Cc), 2e-5)
    emm, emv = npo.batchnorm_moving_update(mm, mv, mean, var, rows)
    close(be.host(mm2), emm, 1e-6)
    close(be.host(mv2), emv, 1e-5)
    # backward vs torch autograd
    dy = rnd(rows, Cc)
    dyd = be.dev(dy)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    gt_, bt_ = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (gamma, beta)]
    z, _, _ = tho.bn_train(xt.reshape(1, 1, rows, Cc), gt_, bt_)
    out = torch.nn.functional.leaky_relu(z, 0.3).reshape(rows, Cc)
    gx, gg, gb = torch.autograd.grad(out, [xt, gt_, bt_], torch.tensor(dy, dtype=torch.float64))
    bs = be.empty((2 * Cc,), np.float64)
    ck(be, be.lib.lu_bn_lrelu_bwd_reduce(be.ptr(xd), be.ptr(dyd), be.ptr(scale), be.ptr(shift), be.ptr(smean),
                                         be.ptr(sinv), 0.3, rows, Cc, be.ptr(bs), be.ptr(ws), be.stream), 'bwd_reduce')
    dx, dg, db = be.empty(x.shape), be.empty((Cc,)), be.empty((Cc,))
    ck(be, be.lib.lu_bn_lrelu_bwd_apply(be.ptr(xd), be.ptr(dyd), be.ptr(scale), be.ptr(shift), be.ptr(smean),
                                        be.ptr(sinv), 0.3, be.ptr(bs), float(rows), be.ptr(dx), be.ptr(dg), be.ptr(db),
                                        rows, Cc, be.stream), 'bwd')
    close(be.host(dx), gx.numpy(), 2e-5)
    close(be.host(dg), gg.numpy(), 2e-4)
    close(be.host(db), gb.numpy(), 2e-4)
    # inference path
    mmd, mvd = be.dev(mm), be.dev(mv)
    ck(be, be.lib.lu_bn_finalize_infer(be.ptr(gd), be.ptr(bd), be.ptr(mmd), be.ptr(mvd), 1e-3, be.ptr(scale),
                                       be.ptr(shift), Cc, be.stream), 'fin_inf')
    ck(be, be.lib.lu_bn_lrelu_apply(be.ptr(xd), be.ptr(y), be.ptr(scale), be.ptr(shift), 0.3, rows, Cc, be.stream),
       'apply')
    close(be.host(y), npo.leaky_relu(npo.batchnorm_infer(x, gamma, beta, mm, mv)), 2e-5)


def test_colsum(be):
    x = rnd(777, 10)
    xd = be.dev(x)
    ws = be.empty((be.lib.lu_colreduce_workspace_bytes(777, 6) // 8 + 1,), np.float64)
    out = be.dev(np.ones(6))
    ck(be, be.lib.lu_colsum(be.ptr(xd, 2), 777, 6, 10, be.ptr(out), 1.0, be.ptr(ws), be.stream), 'colsum')
    close(be.host(out), 1 + x[:, 2:8].astype(np.float64).sum(0), 1e-4)


@pytest.mark.parametrize('resize', ['tf2.0', 'half_pixel'])
def test_upsample_fwd_bwd(be, resize):
    """Both bilinear source-coordinate conventions of k.backend.resize_images (Networks.py:143): the legacy v1 op of
    TF 2.0 / 2.1 (src = o / 2) and half-pixel centres (tf.image.resize v2)."""
    legacy = 1 if resize == 'tf2.0' else 0
    fr, H, W, Cc = 2, 5, 4, 3
    x = rnd(fr, H, W, Cc)
    xd = be.dev(x)
    y = be.empty((fr, 2 * H, 2 * W, Cc))
    ck(be, be.lib.lu_upsample2x_fwd(be.ptr(xd), be.ptr(y), fr, H, W, Cc, legacy, be.stream), 'up')
    close(be.host(y), npo.resize_bilinear(x, 2, resize), 1e-6)
    if legacy:          # out[2i] = in[i], out[2i+1] = mean of in[i], in[i+1] (edge clamp)
        assert np.array_equal(be.host(y)[:, ::2, ::2], x)
    dyw = rnd(fr, 2 * H, 2 * W, Cc + 2)    # gradient arrives as a channel slice of a wider tensor
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    up = tho.resize_bilinear(xt, 2, resize)
    assert np.abs(up.detach().numpy() - npo.resize_bilinear(x, 2, resize)).max() <= 1e-12      # the two oracles agree
    (gx,) = torch.autograd.grad(up, [xt], torch.tensor(dyw[..., :Cc], dtype=torch.float64))
    dx = be.empty(x.shape)
    dywd = be.dev(dyw)
    ck(be, be.lib.lu_upsample2x_bwd(be.ptr(dywd), Cc + 2, be.ptr(dx), fr, H, W, Cc, legacy, be.stream), 'upb')
    close(be.host(dx), gx.numpy(), 1e-5)


def test_bf16_result_variants_of_apply_and_resize(be):
    """lu_bn_lrelu_apply_bf16 / lu_upsample2x_fwd_bf16 (bf16 mode: activations whose every consumer rounds them to bf16): the
    fp32 result of the plain entry point, rounded to nearest even at the store -- bit for bit."""
    rows, Cc = 200, 24
    x, scale, shift = rnd(rows, Cc, scale=2.0), f32(0.5 + RNG.random(Cc)), rnd(Cc, scale=0.3)
    xd, sd, hd = be.dev(x), be.dev(scale), be.dev(shift)
    y32, y16 = be.empty(x.shape), be.empty(x.shape, np.int16)
    ck(be, be.lib.lu_bn_lrelu_apply(be.ptr(xd), be.ptr(y32), be.ptr(sd), be.ptr(hd), 0.3, rows, Cc, be.stream), 'apply')
    ck(be, be.lib.lu_bn_lrelu_apply_bf16(be.ptr(xd), be.ptr(y16), be.ptr(sd), be.ptr(hd), 0.3, rows, Cc, be.stream), 'apply16')
    assert np.array_equal(be.host(y16), KH.bf16_bits(be.host(y32)))
    for legacy in (1, 0):
        fr, H, W, Cc = 2, 5, 6, 8
        x = rnd(fr, H, W, Cc)
        xd = be.dev(x)
        u32, u16 = be.empty((fr, 2 * H, 2 * W, Cc)), be.empty((fr, 2 * H, 2 * W, Cc), np.int16)
        ck(be, be.lib.lu_upsample2x_fwd(be.ptr(xd), be.ptr(u32), fr, H, W, Cc, legacy, be.stream), 'up')
        ck(be, be.lib.lu_upsample2x_fwd_bf16(be.ptr(xd), be.ptr(u16), fr, H, W, Cc, legacy, be.stream), 'up16')
        assert np.array_equal(be.host(u16), KH.bf16_bits(be.host(u32)))
    with pytest.raises(RuntimeError):      # C % 4 != 0: the caller keeps such tensors in fp32
        ck(be, be.lib.lu_bn_lrelu_apply_bf16(be.ptr(xd), be.ptr(y16), be.ptr(sd), be.ptr(hd), 0.3, 10, 6, be.stream), 'apply16')


def test_window_copy_reflect_crop_embed(be):
    x = rnd(2, 5, 6, 2)
    xd = be.dev(x)
    y = be.empty((2, 10, 11, 2))
    ck(be, be.lib.lu_window_copy(be.ptr(xd), 2, be.ptr(y), 2, 5, 6, 10, 11, 2, 2, 1, 1, 0.0, be.stream), 'reflect')
    assert np.array_equal(be.host(y), npo.reflect_pad_hw(x, (2, 3), (1, 4)).astype(np.float32))
    crop = be.empty((2, 5, 6, 2))
    ck(be, be.lib.lu_window_copy(be.ptr(y), 2, be.ptr(crop), 2, 10, 11, 5, 6, 2, -2, -1, 0, 0.0, be.stream), 'crop')
    assert np.array_equal(be.host(crop), x)
    emb = be.dev(np.ones((2, 10, 11, 2)))
    ck(be, be.lib.lu_window_copy(be.ptr(xd), 2, be.ptr(emb), 2, 5, 6, 10, 11, 2, 2, 1, 0, 1.0, be.stream), 'embed')
    exp = np.ones((2, 10, 11, 2), np.float32)
    exp[:, 2:7, 1:7] += x
    assert np.array_equal(be.host(emb), exp)
    # the 16-byte form (C and the pixel stride in groups of four floats): reflect pad, crop out of a channel slice of a wider
    # tensor, and the zero-padded width extension precision 'bf16x3' feeds the kernel-row weight gradient (engine._x3_pad_w)
    x8 = rnd(2, 5, 6, 12)
    x8d = be.dev(x8)
    y8 = be.empty((2, 10, 11, 8))
    ck(be, be.lib.lu_window_copy(be.ptr(x8d, 4), 12, be.ptr(y8), 2, 5, 6, 10, 11, 8, 2, 1, 1, 0.0, be.stream), 'reflect vec4')
    assert np.array_equal(be.host(y8), npo.reflect_pad_hw(x8[..., 4:], (2, 3), (1, 4)).astype(np.float32))
    wide = be.dev(np.full((2, 5, 32, 12), 7.0))
    ck(be, be.lib.lu_window_copy(be.ptr(x8d), 12, be.ptr(wide), 2, 5, 6, 5, 32, 12, 0, 0, 0, 0.0, be.stream), 'pad to W % 32 == 0')
    exp8 = np.zeros((2, 5, 32, 12), np.float32)
    exp8[:, :, :6] = x8
    assert np.array_equal(be.host(wide), exp8)


def test_softmax_wce(be):
    rows = 1000
    lg = rnd(rows, 3, scale=2.0)
    gt = f32(RNG.integers(-1, 3, rows))
    cw = f32([0.15, 0.25, 0.6])
    lgd, gtd, cwd = be.dev(lg), be.dev(gt), be.dev(cw)
    ws = be.empty((be.lib.lu_wce_workspace_bytes(rows) // 8 + 1,), np.float64)
    sums, sm, loss = be.empty((2,), np.float64), be.empty(lg.shape), be.empty((1,))
    ck(be, be.lib.lu_softmax_wce_fwd(be.ptr(lgd), be.ptr(gtd), be.ptr(cwd), be.ptr(sm), be.ptr(sums), rows, be.ptr(ws),
                                     be.stream), 'f')
    ck(be, be.lib.lu_wce_finalize(be.ptr(sums), be.ptr(loss), be.stream), 'fin')
    assert abs(be.host(loss)[0] - npo.weighted_ce(gt, lg, cw)) < 1e-5
    close(be.host(sm), npo.softmax(lg.astype(np.float64)), 1e-6)
    sm2 = be.empty(lg.shape)
    ck(be, be.lib.lu_softmax3(be.ptr(lgd), be.ptr(sm2), rows, be.stream), 'softmax3')
    close(be.host(sm2), npo.softmax(lg.astype(np.float64)), 1e-6)
    for classes in (1, 2, 3, 5, 11):          # any head depth (Networks.py:205-206: last_depth = filters of the last up-block kernel)
        lgc = rnd(257, classes, scale=3.0)
        lgcd, smc = be.dev(lgc), be.empty(lgc.shape)
        ck(be, be.lib.lu_softmax_rows(be.ptr(lgcd), be.ptr(smc), 257, classes, be.stream), 'softmax_rows')
        close(be.host(smc), npo.softmax(lgc.astype(np.float64)), 1e-6)
    lt = torch.tensor(lg, dtype=torch.float64, requires_grad=True)
    (gl,) = torch.autograd.grad(tho.weighted_ce(torch.tensor(gt, dtype=torch.float64), lt, cw.tolist()), [lt])
    dl = be.empty(lg.shape)
    ck(be, be.lib.lu_softmax_wce_bwd(be.ptr(lgd), be.ptr(gtd), be.ptr(cwd), be.ptr(sums), 1.0, be.ptr(dl), rows,
                                     be.stream), 'b')
    close(be.host(dl), gl.numpy(), 1e-7)


def test_adam_scale_transpose_add(be):
    n = 1003
    p, g = rnd(n), rnd(n)
    m, v = rnd(n, scale=0.1), f32(RNG.random(n) * 0.01)
    step, lr = 3, 1e-3
    alpha = lr * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
    ep, em, ev = npo.adam_step(p.astype(np.float64), 0.5 * g.astype(np.float64), m.astype(np.float64),
                               v.astype(np.float64), step, lr=lr)
    pd, gd, md, vd = be.dev(p), be.dev(g), be.dev(m), be.dev(v)
    ck(be, be.lib.lu_adam_step(be.ptr(pd), be.ptr(gd), be.ptr(md), be.ptr(vd), n, alpha, 0.9, 0.999, 1e-7, 0.5,
                               be.stream), 'adam')
    close(be.host(pd), ep, 1e-6)
    close(be.host(md), em, 1e-6)
    close(be.host(vd), ev, 1e-6)
    st = rnd(3, 40)
    keep = f32([1, 0, 1])
    std, kd = be.dev(st), be.dev(keep)
    ck(be, be.lib.lu_scale_frames(be.ptr(std), be.ptr(kd), 3, 40, be.stream), 'mask')
    assert np.array_equal(be.host(std), st * keep[:, None])
    a = rnd(2, 35, 3)
    ad = be.dev(a)
    b = be.empty((2, 3, 35))
    ck(be, be.lib.lu_transpose_inner(be.ptr(ad), be.ptr(b), 2, 35, 3, be.stream), 'tr')
    assert np.array_equal(be.host(b), a.transpose(0, 2, 1))
    y0, x0 = rnd(100), rnd(100)
    yd, x0d = be.dev(y0), be.dev(x0)
    ck(be, be.lib.lu_add_inplace(be.ptr(yd), be.ptr(x0d), 100, be.stream), 'add')
    assert np.array_equal(be.host(yd), y0 + x0)


def test_bf16_tape_pointwise(be):
    """lu_lstm_gates_bwd_bf16 == the fp32 gate backward evaluated on the bf16-rounded saved gates, result rounded to bf16;
    lu_convert_* round trip; lu_im2col_bf16 against a direct gather."""
    R = KH.bf16_round
    F = 8
    sh = (2, 3, 5, F)
    gates = np.concatenate([RNG.random(sh + ()).astype(np.float32) for _ in range(4)], -1)
    gates[..., 2 * F:3 * F] = 2 * gates[..., 2 * F:3 * F] - 1
    gates[0, 0, 0, :4] = [0.0, 1.0, 0.5, 1.0]                      # saturated hard-sigmoid gates: zero slope
    c_prev, c_cur, dh_a, dh_b, dc_in = [rnd(*sh) for _ in range(5)]
    g = R(gates).astype(np.float64)
    gi, gf, gg, go = [g[..., i * F:(i + 1) * F] for i in range(4)]
    hs = lambda a: np.where((a > 0) & (a < 1), 0.2, 0.0)          # noqa: E731
    for (db, di) in [(dh_b, dc_in), (None, None)]:
        dh = dh_a.astype(np.float64) + (0 if db is None else db)
        tc = np.tanh(c_cur.astype(np.float64))
        dc = dh * go * (1 - tc * tc) + (0 if di is None else di)
        ref = np.concatenate([dc * gg * hs(gi), dc * c_prev * hs(gf), dc * gi * (1 - gg * gg), dh * tc * hs(go)], -1)
        dz, dcp = KH.lstm_gates_bwd_bf16(be, gates, c_prev, c_cur, dh_a, db, di)
        assert np.abs(dz - ref).max() <= 2.0 ** -8 * np.abs(ref).max() + 1e-6
        close(dcp, dc * gf, 1e-5)
    x = rnd(1000)
    yb, xd = be.empty((1000,), np.int16), be.dev(x)
    ck(be, be.lib.lu_convert_f32_bf16(be.ptr(xd), be.ptr(yb), 1000, be.stream), 'cvt')
    assert np.array_equal(KH.bf16_values(be.host(yb)), R(x))
    xf = be.empty((1000,))
    ck(be, be.lib.lu_convert_bf16_f32(be.ptr(yb), be.ptr(xf), 1000, be.stream), 'cvt back')
    assert np.array_equal(be.host(xf), R(x))
    for (k, cin) in [(5, 1), (3, 3), (1, 2)]:
        x = rnd(2, 6, 7, cin)
        y, xd = be.empty((2, 6, 7, 32), np.int16), be.dev(x)
        ck(be, be.lib.lu_im2col_bf16(be.ptr(xd), be.ptr(y), 2, 6, 7, cin, k, be.stream), 'im2col')
        p = (k - 1) // 2
        xp = np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)))
        ref = np.zeros((2, 6, 7, 32), np.float32)
        for kh in range(k):
            for kw in range(k):
                ref[..., (kh * k + kw) * cin:(kh * k + kw + 1) * cin] = xp[:, kh:kh + 6, kw:kw + 7, :]
        assert np.array_equal(KH.bf16_values(be.host(y)), R(ref))


def test_weight_prep_batch_equals_the_single_calls(be):
    """lu_weight_prep_batch: flips and bf16 packs of several kernels (ragged channel / column counts, a channel-slice flip, a
    pack of a channel-slice VIEW, a pack of a flip's output in the second launch) from one device table each -- bit for bit the
    images lu_weight_flip_transpose / lu_pack_weights_bf16 write."""
    ws = [rnd(3, 3, 40, 70), rnd(5, 5, 33, 64), rnd(1, 1, 32, 3), rnd(3, 3, 65, 32)]
    wd = [be.dev(w) for w in ws]
    flips = [(0, 0, 40), (1, 0, 33), (3, 1, 64), (2, 0, 32)]            # (kernel, c_off, c_sub)
    flip_ref = [KH.flip_transpose(be, ws[i], co, cs) for i, co, cs in flips]
    flip_out = [be.empty(r.shape) for r in flip_ref]

    def table(records):
        arr = (cabi.PrepOp * len(records))()
        blk = 0
        for o, rec in zip(arr, records):
            for name, val in rec.items():
                setattr(o, name, val)
            o.blk0 = blk
            blk += rec['nblk']
        return be.dev(np.frombuffer(bytes(arr), np.uint8), np.uint8), len(records), blk

    recs = []
    for (i, co, cs), out in zip(flips, flip_out):
        k, _, Ct, N = ws[i].shape
        recs.append(dict(kind=0, k=k, src=be.ptr(wd[i]), dst=be.ptr(out), C=cs, N=N, C_tot=Ct, c_off=co,
                         nblk=k * k * -(-cs // 32) * -(-N // 32)))
    t, n, blocks = table(recs)
    calls.check(be.lib, be.lib.lu_weight_prep_batch(be.ptr(t), n, blocks, be.stream), 'prep flips')
    for out, ref in zip(flip_out, flip_ref):
        assert np.array_equal(be.host(out), ref)

    # packs: whole kernels, a channel slice [8, 40) of kernel 0 (view strides), and the flip of kernel 1 (second level)
    packs = [(wd[0], 0, 3, 40, 70, 40 * 70, 70), (wd[1], 0, 5, 33, 64, 33 * 64, 64), (wd[0], 8 * 70, 3, 32, 70, 40 * 70, 70),
             (flip_out[1], 0, 5, 64, 33, 64 * 33, 33), (wd[2], 0, 1, 32, 3, 32 * 3, 3)]
    refs, outs, recs = [], [], []
    for nb, (src, off, k, Cc, N, ts, rs) in enumerate(packs):
        nbytes = be.lib.lu_pack_weights_bf16_bytes(k, Cc, N)
        ref, out = be.empty((nbytes // 4,)), be.empty((nbytes // 4,))
        calls.check(be.lib, be.lib.lu_pack_weights_bf16(be.ptr(src, off), ts, rs, k, Cc, N, be.ptr(ref), be.stream), 'pack')
        refs.append(ref)
        outs.append(out)
        recs.append(dict(kind=1, k=k, src=be.ptr(src, off), dst=be.ptr(out), tap_stride=ts, row_stride=rs, kk=k * k, C=Cc, N=N,
                         nblk=1 + nb % 3))
    t, n, blocks = table(recs)
    calls.check(be.lib, be.lib.lu_weight_prep_batch(be.ptr(t), n, blocks, be.stream), 'prep packs')
    for out, ref in zip(outs, refs):
        assert np.array_equal(be.host(out).view(np.uint32), be.host(ref).view(np.uint32))


def test_bn_lrelu_bwd_apply_bf16_result_and_state_begin(be):
    """Two helper entry points of the bf16 / lazy-state paths at kernel level.
    lu_bn_lrelu_bwd_apply_bf16: the fp32 backward rounded to nearest-even bf16 at the store -- exactly the bits a consumer that
    rounds the fp32 tensor itself would see.  lu_state_begin: dst = src * keep[frame] (+ bf16 copy), zeros without a source, a
    plain copy without a mask; 16-byte and scalar forms."""
    rows, Cc = 256, 24
    x, dy = rnd(rows, Cc, scale=2.0) + 0.5, rnd(rows, Cc)
    gamma, beta = f32(1 + 0.2 * RNG.random(Cc)), rnd(Cc, scale=0.3)
    xd, dyd, gd, bd = be.dev(x), be.dev(dy), be.dev(gamma), be.dev(beta)
    ws = be.empty((be.lib.lu_colreduce_workspace_bytes(rows, Cc) // 8 + 1,), np.float64)
    sums = be.empty((2 * Cc,), np.float64)
    ck(be, be.lib.lu_bn_stats(be.ptr(xd), rows, Cc, be.ptr(sums), be.ptr(ws), be.stream), 'stats')
    scale, shift, smean, sinv = [be.empty((Cc,)) for _ in range(4)]
    ck(be, be.lib.lu_bn_finalize_train(be.ptr(sums), float(rows), be.ptr(gd), be.ptr(bd), 1e-3, 0.99, None, None,
                                       be.ptr(scale), be.ptr(shift), be.ptr(smean), be.ptr(sinv), Cc, be.stream), 'fin')
    bs = be.empty((2 * Cc,), np.float64)
    ck(be, be.lib.lu_bn_lrelu_bwd_reduce(be.ptr(xd), be.ptr(dyd), be.ptr(scale), be.ptr(shift), be.ptr(smean),
                                         be.ptr(sinv), 0.3, rows, Cc, be.ptr(bs), be.ptr(ws), be.stream), 'bwd_reduce')
    dx, dg, db = be.empty(x.shape), be.empty((Cc,)), be.empty((Cc,))
    ck(be, be.lib.lu_bn_lrelu_bwd_apply(be.ptr(xd), be.ptr(dyd), be.ptr(scale), be.ptr(shift), be.ptr(smean), be.ptr(sinv),
                                        0.3, be.ptr(bs), float(rows), be.ptr(dx), be.ptr(dg), be.ptr(db), rows, Cc,
                                        be.stream), 'bwd')
    dx16, dg2, db2 = be.empty(x.shape, np.int16), be.empty((Cc,)), be.empty((Cc,))
    ck(be, be.lib.lu_bn_lrelu_bwd_apply_bf16(be.ptr(xd), be.ptr(dyd), be.ptr(scale), be.ptr(shift), be.ptr(smean), be.ptr(sinv),
                                             0.3, be.ptr(bs), float(rows), be.ptr(dx16), be.ptr(dg2), be.ptr(db2), rows, Cc,
                                             be.stream), 'bwd16')
    assert np.array_equal(be.host(dx16), KH.bf16_bits(be.host(dx)))
    assert np.array_equal(be.host(dg2), be.host(dg)) and np.array_equal(be.host(db2), be.host(db))

    for per_frame in (4 * 36, 7 * 5):          # float4 lanes / scalar tail form
        frames = 3
        src = rnd(frames, per_frame)
        keep = f32([1.0, 0.0, 0.5])
        sd, kd = be.dev(src), be.dev(keep)
        for use_src, use_keep, use16 in ((True, True, True), (True, False, False), (False, True, True), (True, True, False)):
            dst = be.empty((frames, per_frame))
            d16 = be.empty((frames, per_frame), np.int16) if use16 else None
            ck(be, be.lib.lu_state_begin(be.ptr(dst), be.ptr(d16), be.ptr(sd) if use_src else None,
                                         be.ptr(kd) if use_keep else None, frames, per_frame, be.stream), 'state_begin')
            want = (src * (keep[:, None] if use_keep else 1.0)) if use_src else np.zeros_like(src)
            assert np.array_equal(be.host(dst), f32(want)), (per_frame, use_src, use_keep)
            if use16:
                assert np.array_equal(be.host(d16), KH.bf16_bits(f32(want)))


def test_conv3_bf16_sources_tall_patch_and_the_8_row_kernel(be):
    """3x3 bf16 layers on bf16 sources run the third loop generation on 16-row patches (conv_halo_frag3_kernel<3, BIAS, 8, true>) when
    the launch has enough tiles, the first generation on 8-row patches otherwise; forced either way (LU_CONV_F_PATCH16 / PATCH8) both
    against the oracle on the same bf16-rounded operands, and against each other to fp32 summation order (kernel-column-major vs
    kernel-row-major taps); ragged extents and a partial column tile included."""
    for (H, W, Cc, N) in [(20, 40, 40, 136), (16, 32, 64, 128)]:
        x = KH.bf16_round(rnd(2, H, W, Cc))
        w, b = rnd(3, 3, Cc, N, scale=0.1), rnd(N)
        o8 = KH.conv2d(be, [x], [w], b, 3, precision=1, flags=cabi.LU_CONV_F_PATCH8, bf16_src=(0,))
        o16 = KH.conv2d(be, [x], [w], b, 3, precision=1, flags=cabi.LU_CONV_F_PATCH16, bf16_src=(0,))
        ref = npo.conv2d_same(x.astype(np.float64), KH.bf16_round(w).astype(np.float64), b.astype(np.float64), 1)
        tol = 5e-5 * max(1.0, float(np.abs(ref).max()))
        close(o16, ref, tol)
        close(o8, ref, tol)
        close(o8, o16, tol)


def np_split3(x):
    """numpy restatement of the three-way bf16 split (lu_split6): x == hi + mid + lo exactly."""
    x = f32(x)
    with np.errstate(over='ignore', invalid='ignore'):
        hi = KH.bf16_round(x)
        # a finite x within 2^-9 of FLT_MAX would round to inf: hi = the largest finite bf16, the residuals stay exact;  +-inf travel
        # in hi alone (lu_device.h lu_split3, round 6)
        hi = np.where(np.isinf(hi) & np.isfinite(x), np.copysign(np.float32(3.3895313892515355e38), x), hi).astype(np.float32)
        r1 = np.where(np.isfinite(x), x - hi, 0.0).astype(np.float32)
    mid = KH.bf16_round(r1)
    lo = KH.bf16_round((r1 - mid).astype(np.float32))
    return hi, mid, lo


SPLIT_ORDER = {0: (2, 1, 0, 1, 0, 0), 1: (0, 1, 2, 0, 1, 0)}      # piece (0 hi, 1 mid, 2 lo) in block j of order A / B


def split6_ref(x2d, lp, order):
    rows, L = x2d.shape
    pc = np_split3(x2d)
    out = np.zeros((rows, 6, lp), np.float32)
    for j, p in enumerate(SPLIT_ORDER[order]):
        out[:, j, :L] = pc[p]
    return out


def test_split6_pieces_are_exact_and_in_block_order(be):
    """lu_split6 (precision 'bf16x3'): the three bf16 pieces sum to the fp32 value EXACTLY, each block holds the piece its order
    names, pad columns are zero; 16-byte and scalar forms, bf16 and fp32 outputs, strided rows."""
    mags = np.exp(RNG.uniform(np.log(1e-12), np.log(1e12), size=(37, 16))) * RNG.choice([-1.0, 1.0], size=(37, 16))
    xs = f32(mags)
    xs[0, :4] = [0.0, -0.0, 1.0, np.float32(1.0) + np.float32(2.0 ** -23)]
    xs[1, :4] = [3.0e38, -3.0e38, 1.17549435e-38 * 4096, 65504.0]
    # values fp32 carries and a naive split turns into NaN: FLT_MAX, the first float that rounds to a bf16 inf, +-inf
    xs[2, :6] = [np.finfo(np.float32).max, -np.finfo(np.float32).max, np.float32(3.3961775292304601e38), np.float32(3.39e38), np.inf, -np.inf]
    for (L, lp, xstride, ypad) in [(16, 16, 16, 0), (12, 12, 16, 8), (5, 8, 16, 0), (1, 4, 16, 4)]:
        x2d = np.ascontiguousarray(xs[:, :L])
        hi, mid, lo = np_split3(x2d)
        assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x2d.astype(np.float64))
        xd = be.dev(xs)      # rows of 16 floats; the kernel reads the first L of each
        for order in (0, 1):
            ref = split6_ref(x2d, lp, order)
            ys = 6 * lp + ypad
            yb = be.empty((37, ys), np.int16)
            ck(be, be.lib.lu_split6(be.ptr(xd), 37, L, xstride, be.ptr(yb), ys, lp, order, cabi.LU_BF16, be.stream), 'split6 bf16')
            got = KH.bf16_values(be.host(yb))[:, :6 * lp].reshape(37, 6, lp)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (L, lp, order)
            yf = be.empty((37, ys))
            ck(be, be.lib.lu_split6(be.ptr(xd), 37, L, xstride, be.ptr(yf), ys, lp, order, cabi.LU_F32, be.stream), 'split6 f32')
            assert np.array_equal(be.host(yf)[:, :6 * lp].reshape(37, 6, lp).view(np.uint32), ref.view(np.uint32)), (L, lp, order)


@pytest.mark.parametrize('k', [3, 5])
def test_split6_convolution_on_the_bf16_kernels_is_fp32_arithmetic(be, k):
    """A bf16-MFMA convolution over the six channel blocks of a split6 activation (order A) and a split6 kernel (order B) IS the
    fp32 convolution: against the fp64 oracle it sits where the exact-fp32 MFMA kernel sits (<= 2x its error; 2^-26 per
    product, fp32 accumulation), a hundred times below the plain bf16 kernel's rounding."""
    fr, H, W, Cc, N = 1, 8, 32, 16, 128
    x, w, b = rnd(fr, H, W, Cc), rnd(k, k, Cc, N, scale=0.2), rnd(N)
    ref = npo.conv2d_same(x, w, b, 1)
    e32 = np.abs(KH.conv2d(be, [x], [w], b, k, 1) - ref).max()
    x6 = split6_ref(x.reshape(-1, Cc), Cc, 0).reshape(fr, H, W, 6 * Cc)
    w6 = split6_ref(w.reshape(k * k, Cc * N), Cc * N, 1).reshape(k, k, 6 * Cc, N)
    e6 = np.abs(KH.conv2d(be, [x6], [w6], b, k, 1, precision=1, bf16_src=(0,)) - ref).max()
    e16 = np.abs(KH.conv2d(be, [x], [w], b, k, 1, precision=1) - ref).max()
    print('split6 conv k=%d: |err| vs fp64  fp32 MFMA %.3e   bf16x3 %.3e   bf16 %.3e' % (k, e32, e6, e16))
    assert e6 <= 2.0 * e32 + 1e-6 and e16 >= 100.0 * e6


@pytest.mark.parametrize('case', [(5, 2, 4, 32, 64, 128, 3), (3, 3, 4, 64, 64, 128, 5), (5, 1, 8, 32, 128, 128, 1)])
def test_split6_weight_gradient_with_the_terms_as_frames_is_fp32_arithmetic(be, case):
    """lu_wgrad_desc.terms: ONE launch of the bf16 kernel-row weight gradient sums block t of a split6 x (order A) against block t
    of a split6 dy (order B) over the terms -- pixel slabs that begin inside any term (splits > 1), the two-launch form of the
    engine (terms 0-2 with the bias gradient = column sums of hi + mid + lo = dy, then terms 3-5 on top) -- and the result is the
    fp32 weight gradient: against the fp64 oracle it sits where the exact-fp32 kernel sits, far below the bf16 kernel."""
    k, fr, H, W, Cc, N, splits = case
    x, dy = rnd(fr, H, W, Cc), rnd(fr, H, W, N, scale=0.3)
    _, ref = _torch_conv_grads(x, np.zeros((k, k, Cc, N), np.float32), dy, 1)
    x6 = split6_ref(x.reshape(-1, Cc), Cc, 0).reshape(fr, H, W, 6 * Cc)
    dy6 = split6_ref(dy.reshape(-1, N), N, 1).reshape(fr, H, W, 6 * N)
    dw_a, db = KH.conv2d_wgrad(be, x6, dy6, k, 1, splits=splits, terms=(0, 3), dbias0=np.full(N, 7.0, np.float32))
    dw = KH.conv2d_wgrad(be, x6, dy6, k, 1, splits=splits, terms=(3, 3), dw0=dw_a, beta=1.0)
    one = KH.conv2d_wgrad(be, x6, dy6, k, 1, splits=max(1, splits - 1), terms=(0, 6))      # all six in one launch: the same sum
    e32 = np.abs(KH.conv2d_wgrad(be, x, dy, k, 1, splits=splits) - ref).max()
    e6 = np.abs(dw - ref).max()
    e16 = np.abs(KH.conv2d_wgrad(be, x, dy, k, 1, splits=splits, precision=1) - ref).max()
    print('split6 wgrad k=%d: |err| vs fp64  fp32 MFMA %.3e   bf16x3 %.3e   bf16 %.3e' % (k, e32, e6, e16))
    assert e6 <= 2.0 * e32 + 2e-6 * np.abs(ref).max() and e16 >= 30.0 * e6
    close(one, dw, 5e-5 * max(1.0, np.abs(ref).max()))
    close(db, dy.reshape(-1, N).astype(np.float64).sum(0), 2e-4)


@pytest.mark.parametrize('case', [(5, 2, 3, 64, 128, 128, 3), (3, 1, 4, 32, 128, 200, 2), (5, 1, 2, 32, 256, 128, 1)])
def test_split6_weight_gradient_piece_aware_kernel_is_fp32_arithmetic(be, case):
    """LU_WGRAD_F_PIECES3 (round 6, wgrad_row_x3_kernel): ONE pass over the frames stages the three pieces of a 32-pixel run of x and dy
    once and issues the six products of the split from registers.  Against the fp64 oracle the result sits where the exact-fp32 MFMA
    kernel sits (and where the terms-as-frames form of round 5 sits), far below the plain bf16 kernel; the bias gradient is the column sum
    of hi + mid + lo = dy; beta accumulates; ragged column counts (N = 200), two channel tiles (C = 256), slabs that start mid-frame."""
    k, fr, H, W, Cc, N, splits = case
    if be.name == 'emu':      # (the host emulator runs one frame per case: the CPU suite has to stay within minutes)
        fr, H, splits = 1, min(H, 3), min(splits, 2)
    x, dy = rnd(fr, H, W, Cc), rnd(fr, H, W, N, scale=0.3)
    _, ref = _torch_conv_grads(x, np.zeros((k, k, Cc, N), np.float32), dy, 1)
    x6 = split6_ref(x.reshape(-1, Cc), Cc, 0).reshape(fr, H, W, 6 * Cc)
    dy6 = split6_ref(dy.reshape(-1, N), N, 1).reshape(fr, H, W, 6 * N)
    dw, db = KH.conv2d_wgrad(be, x6, dy6, k, 1, splits=splits, terms=(0, 6), flags=cabi.LU_WGRAD_F_PIECES3, dbias0=np.full(N, 7.0, np.float32))
    frames_form = KH.conv2d_wgrad(be, x6, dy6, k, 1, splits=splits, terms=(0, 6))      # round 5: the six products as six times the frames
    e32 = np.abs(KH.conv2d_wgrad(be, x, dy, k, 1, splits=splits) - ref).max()
    e6, e6f = np.abs(dw - ref).max(), np.abs(frames_form - ref).max()
    e16 = np.abs(KH.conv2d_wgrad(be, x, dy, k, 1, splits=splits, precision=1) - ref).max()
    print('split6 wgrad, piece-aware k=%d C=%d N=%d: |err| vs fp64  fp32 MFMA %.3e   pieces %.3e   terms-as-frames %.3e   bf16 %.3e' % (k, Cc, N, e32, e6, e6f, e16))
    assert e6 <= 2.0 * e32 + 2e-6 * np.abs(ref).max() and e16 >= 30.0 * e6
    close(dw, frames_form, 5e-5 * max(1.0, np.abs(ref).max()))
    close(db, dy.reshape(-1, N).astype(np.float64).sum(0), 2e-4)
    acc = KH.conv2d_wgrad(be, x6, dy6, k, 1, splits=splits, terms=(0, 6), flags=cabi.LU_WGRAD_F_PIECES3, dw0=np.full_like(dw, 0.5), beta=1.0)
    close(acc, dw + 0.5, 1e-6 * max(1.0, np.abs(ref).max()))


def test_gate_backward_with_the_split_image_of_dz(be):
    """lu_lstm_gates_bwd_split (precision 'bf16x3'): dz in place of the fp32 gates and dc = what lu_lstm_gates_bwd writes (the same
    formulas; to fp32 rounding -- hipcc contracts the two kernels' multiply-adds differently), dz6 = the lu_split6 image (order B)
    of EXACTLY the dz this kernel wrote."""
    fr, H, W, F = 2, 3, 5, 8
    sh = (fr, H, W, F)
    gates = f32(RNG.uniform(-0.2, 1.2, size=(fr, H, W, 4 * F)).clip(0.0, 1.0))
    gates[..., 2 * F:3 * F] = np.tanh(rnd(*sh))
    c_prev, c_cur, dh_a, dh_b, dc_in = [rnd(*sh) for _ in range(5)]
    for (db, di) in [(dh_b, dc_in), (None, None)]:
        gd, cp, cc, da = be.dev(gates), be.dev(c_prev), be.dev(c_cur), be.dev(dh_a)
        dbd, did = (None if db is None else be.dev(db)), (None if di is None else be.dev(di))
        dz, dcp = be.empty((fr, H, W, 4 * F)), be.empty(sh)
        ck(be, be.lib.lu_lstm_gates_bwd(be.ptr(gd), be.ptr(cp), be.ptr(cc), be.ptr(da), H * W * F, be.ptr(dbd), be.ptr(did), be.ptr(dz),
                                        be.ptr(dcp), fr, H * W, F, be.stream), 'gates bwd')
        g2, dcp2, dz6 = be.dev(gates), be.empty(sh), be.empty((fr, H, W, 24 * F), np.int16)
        ck(be, be.lib.lu_lstm_gates_bwd_split(be.ptr(g2), be.ptr(cp), be.ptr(cc), be.ptr(da), H * W * F, be.ptr(dbd), be.ptr(did),
                                              be.ptr(dz6), be.ptr(dcp2), fr, H * W, F, be.stream), 'gates bwd split')
        dz_ref, dz_got = be.host(dz), be.host(g2)
        close(dz_got, dz_ref, 2e-6 * max(1.0, float(np.abs(dz_ref).max())))
        close(be.host(dcp2), be.host(dcp), 2e-6 * max(1.0, float(np.abs(be.host(dcp)).max())))
        ref6 = split6_ref(dz_got.reshape(-1, 4 * F), 4 * F, 1).reshape(fr, H, W, 24 * F)
        assert np.array_equal(KH.bf16_values(be.host(dz6)).view(np.uint32), ref6.view(np.uint32))


def test_split6_weight_image_in_one_pass(be):
    """lu_pack_weights_split6_bf16 == lu_pack_weights_bf16 of the lu_split6 (fp32) image of the kernel, bit for bit: both block
    orders, a thin kernel with zero-padded blocks (C = 1 -> cp = 4), ragged channel / column counts."""
    for (k, Cc, cp, N) in [(3, 16, 16, 40), (5, 1, 4, 64), (3, 20, 20, 33)]:
        w = rnd(k, k, Cc, N, scale=0.3)
        wd = be.dev(w)
        for order in (0, 1):
            w6 = split6_ref(w.reshape(k * k, Cc * N), Cc * N, order).reshape(k * k, 6, Cc, N)
            w6p = np.zeros((k * k, 6, cp, N), np.float32)
            w6p[:, :, :Cc] = w6
            ref = be.host(KH.pack_bf16(be, be.dev(w6p.reshape(k, k, 6 * cp, N)), k, 6 * cp, N))
            nbytes = be.lib.lu_pack_weights_bf16_bytes(k, 6 * cp, N)
            out = be.empty((nbytes // 4 + 4,))
            ck(be, be.lib.lu_pack_weights_split6_bf16(be.ptr(wd), Cc * N, N, k, Cc, cp, N, order, be.ptr(out), be.stream), 'pack split6')
            assert np.array_equal(be.host(out)[:nbytes // 4].view(np.uint32), ref[:nbytes // 4].view(np.uint32)), (k, Cc, cp, N, order)
