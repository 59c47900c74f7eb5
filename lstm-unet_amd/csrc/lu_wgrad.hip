// lu_wgrad.hip -- convolution weight gradient on the gfx950 fp32 matrix pipe.
//
//   dw[tap, c, n] = sum_p  x[shift_tap(p), c] * dy[p, n]          p = (frame, oy, ox)
//
// GEMM view per filter tap: M = input channels, N = output channels, K = pixels.  Both operands
// are "K-major" in channels-last memory (a pixel's channels are contiguous), which is exactly the
// LDS image v_mfma_f32_32x32x2_f32 wants: A[i = c][k = pixel], B[k = pixel][j = n] are read with
// conflict-free ds_read_b32 (32 consecutive channels per half-wave), no transposes anywhere.
// One block = one (tap, c-tile, n-tile, pixel-split); 16 pixels per pipeline stage, two LDS stages.
// The pixel axis is split into `splits` slabs (workspace) that a second kernel sums in a fixed
// order -- deterministic, no float atomics.
// Thin inputs (C % 4 != 0, e.g. the 1-channel image): M enumerates flattened (tap, c) instead.
#include <stdlib.h>
#include <string.h>
#include "lu_device.h"

namespace {

constexpr int KP = 16;

struct WgradArgs {
    const float* x;
    const float* dy;
    int64_t x_fs, dy_fs;
    int32_t x_ps, dy_ps, C, N;
    int64_t M;          // frames*Hout*Wout
    int64_t chunk;      // pixels per split (multiple of KP)
    int32_t HWo, Wout, Hout, Hin, Win;
    int32_t k, kk, stride, pad_t, pad_l;
    int32_t c_tiles;
    int32_t n_tiles, inner, splits;      // bf16 kernel-row variant: 128-column tiles, tiles per pixel slab, slabs
    float* ws;          // [splits][kk*C*N]
    int64_t slab;
    float* bias_ws;     // [splits][N] column sums of dy (bias gradient), or null; kernel-row variants only
    int32_t xfold;      // fp32 kernel-row variant: pixel slabs folded into grid.x (8 / column tiles; 0 / 1 = none)
    int32_t ragged, wst, rows_per;      // fp32 kernel-row variant, W % 16 != 0: slabs of whole rows, ceil(W / 16) runs per row
    const void* zero16; // device address of lu_zero16 (bf16 kernel-row variant: a kernel ARGUMENT lives in SGPRs; taken through the
                        // symbol it costs s_getpc + s_load + s_waitcnt lgkmcnt(0) -- which also drains the LDS reads -- per use)
    int32_t tf;         // bf16 kernel-row variant, lu_wgrad_desc.terms > 1: frames per term (0: one term); M counts terms * tf frames
    int64_t x_tj, y_tj; // ... what a term change adds to the element cursors on top of a frame step: term stride - tf * frame stride
};

template <int MF, int NF, int WM, int WN, bool THIN, bool YVEC>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs a) {
    constexpr int BMw = 32 * MF * WM, BNw = 32 * NF * WN;
    constexpr int QPA = BMw / 4, RPA = 256 / QPA, NPA = (KP + RPA - 1) / RPA;
    constexpr int QPB = BNw / 4, RPB = 256 / QPB, NPB = (KP + RPB - 1) / RPB;
    __shared__ __attribute__((aligned(16))) float As[2][KP * BMw];
    __shared__ __attribute__((aligned(16))) float Bs[2][KP * BNw];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.x * BNw;
    int tap = 0, c0 = 0, kh = 0, kw = 0;
    if (!THIN) {
        tap = blockIdx.y / a.c_tiles;
        c0 = (blockIdx.y % a.c_tiles) * BMw;
        kh = tap / a.k;
        kw = tap % a.k;
    }
    const int j0 = THIN ? blockIdx.y * BMw : 0;
    const int64_t p_begin = (int64_t)blockIdx.z * a.chunk;
    int64_t p_end = p_begin + a.chunk;
    if (p_end > a.M) p_end = a.M;
    const int n_it = p_end > p_begin ? (int)((p_end - p_begin + KP - 1) / KP) : 0;

    // thin: this thread's fixed flattened (tap,c) column
    const int tcol = tid & 31;
    int t_kh = 0, t_kw = 0, t_c = 0;
    bool t_ok = false;
    if (THIN) {
        int j = j0 + tcol;
        t_ok = j < a.kk * a.C;
        int tp = j / a.C;
        t_c = j - tp * a.C;
        t_kh = tp / a.k;
        t_kw = tp - t_kh * a.k;
    }
    const int aq = tid % QPA, arow0 = tid / QPA;
    const int bq = tid % QPB, brow0 = tid / QPB;

    // scalars, not arrays: hipcc promoted the small float4 staging arrays to LDS-backed allocas
    float4 ra0, ra1, rb0, rb1, rb2, rb3;
    ra0 = ra1 = rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    static_assert(NPA <= 2 && NPB <= 4, "staging registers are named scalars");
    float rat0 = 0.f, rat1 = 0.f;

    // All global loads are unconditional; masked lanes read lu_zero16 (see lu_conv.hip for why).
    const float* const zp = a.zero16 ? reinterpret_cast<const float*>(a.zero16) : lu_zero16;
    // Per-row pixel cursors (frame, oy, ox), advanced by KP pixels per stage with adds/compares only: the
    // per-stage integer divisions of a naive decode cost more VALU issue slots than the stage's MFMAs leave free.
    struct Pix {
        int64_t p;
        int f, oy, ox;
    };
    auto pix_init = [&](int row) {
        Pix c;
        c.p = p_begin + row;
        const int64_t pc = c.p < a.M ? c.p : 0;
        c.f = (int)(pc / a.HWo);
        const int r = (int)(pc - (int64_t)c.f * a.HWo);
        c.oy = r / a.Wout;
        c.ox = r - c.oy * a.Wout;
        return c;
    };
    auto pix_advance = [&](Pix& c) {
        c.p += KP;
        c.ox += KP;
        while (c.ox >= a.Wout) {
            c.ox -= a.Wout;
            if (++c.oy == a.Hout) {
                c.oy = 0;
                ++c.f;
            }
        }
    };
    Pix pa0 = pix_init(THIN ? (tid >> 5) : arow0), pa1 = pix_init(THIN ? (tid >> 5) + 8 : arow0 + RPA);
    Pix pb0 = pix_init(brow0), pb1 = pix_init(brow0 + RPB), pb2 = pix_init(brow0 + 2 * RPB), pb3 = pix_init(brow0 + 3 * RPB);

    auto load_a = [&](const Pix& c, int row) -> float4 {
        const int ch = c0 + 4 * aq;
        const int iy = c.oy * a.stride + kh - a.pad_t, ix = c.ox * a.stride + kw - a.pad_l;
        const bool ok = row < KP && c.p < p_end && ch < a.C && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        const float* src = a.x + (int64_t)c.f * a.x_fs + ((int64_t)iy * a.Win + ix) * a.x_ps + ch;
        return *reinterpret_cast<const float4*>(ok ? src : zp);
    };
    auto load_a_thin = [&](const Pix& c) -> float {
        const int iy = c.oy * a.stride + t_kh - a.pad_t, ix = c.ox * a.stride + t_kw - a.pad_l;
        const bool ok = t_ok && c.p < p_end && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        const float* src = a.x + (int64_t)c.f * a.x_fs + ((int64_t)iy * a.Win + ix) * a.x_ps + t_c;
        return *(ok ? src : zp);
    };
    auto load_b = [&](const Pix& c, int row) -> float4 {
        const int n = n0 + 4 * bq;
        const bool rok = row < KP && c.p < p_end;
        const float* yrow = a.dy + (int64_t)c.f * a.dy_fs + ((int64_t)c.oy * a.Wout + c.ox) * a.dy_ps + n;
        if (YVEC) {
            return *reinterpret_cast<const float4*>((rok && n < a.N) ? yrow : zp);
        } else {
            const float t0 = *((rok && n + 0 < a.N) ? yrow + 0 : zp), t1 = *((rok && n + 1 < a.N) ? yrow + 1 : zp),
                        t2 = *((rok && n + 2 < a.N) ? yrow + 2 : zp), t3 = *((rok && n + 3 < a.N) ? yrow + 3 : zp);
            return make_float4(t0, t1, t2, t3);
        }
    };
    auto load_stage = [&]() {
        if (!THIN) {
            ra0 = load_a(pa0, arow0);
            if (NPA > 1) ra1 = load_a(pa1, arow0 + RPA);
        } else {
            rat0 = load_a_thin(pa0);
            rat1 = load_a_thin(pa1);
        }
        rb0 = load_b(pb0, brow0);
        if (NPB > 1) rb1 = load_b(pb1, brow0 + RPB);
        if (NPB > 2) rb2 = load_b(pb2, brow0 + 2 * RPB);
        if (NPB > 3) rb3 = load_b(pb3, brow0 + 3 * RPB);
    };
    auto advance_stage = [&]() {
        pix_advance(pa0);
        if (THIN || NPA > 1) pix_advance(pa1);
        pix_advance(pb0);
        if (NPB > 1) pix_advance(pb1);
        if (NPB > 2) pix_advance(pb2);
        if (NPB > 3) pix_advance(pb3);
    };
    auto store_stage = [&](int buf) {
        if (!THIN) {
            if (RPA <= KP || arow0 < KP) *reinterpret_cast<float4*>(&As[buf][arow0 * BMw + 4 * aq]) = ra0;
            if (NPA > 1) *reinterpret_cast<float4*>(&As[buf][(arow0 + RPA) * BMw + 4 * aq]) = ra1;
        } else {
            As[buf][(tid >> 5) * BMw + tcol] = rat0;
            As[buf][((tid >> 5) + 8) * BMw + tcol] = rat1;
        }
        if (RPB <= KP || brow0 < KP) *reinterpret_cast<float4*>(&Bs[buf][brow0 * BNw + 4 * bq]) = rb0;
        if (NPB > 1) *reinterpret_cast<float4*>(&Bs[buf][(brow0 + RPB) * BNw + 4 * bq]) = rb1;
        if (NPB > 2) *reinterpret_cast<float4*>(&Bs[buf][(brow0 + 2 * RPB) * BNw + 4 * bq]) = rb2;
        if (NPB > 3) *reinterpret_cast<float4*>(&Bs[buf][(brow0 + 3 * RPB) * BNw + 4 * bq]) = rb3;
    };

    f32x16 acc[MF][NF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;

    if (n_it > 0) {
        load_stage();
        store_stage(0);
    }
    __syncthreads();
    const int khalf = lane >> 5, l31 = lane & 31;
    // one k-pair (2 pixels) = MF*NF MFMAs; its MF + NF operand values are requested one pair AHEAD (two register sets, round 4):
    // counted waits instead of read -> s_waitcnt lgkmcnt(0) -> MFMAs.  Same values, same MFMA order.
    float avr[2][MF], bvr[2][NF];
    auto rd_pair = [&](int buf, int kk2) {
        const int set = (kk2 >> 1) & 1;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) avr[set][mf] = As[buf][(kk2 + khalf) * BMw + wm * 32 * MF + mf * 32 + l31];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) bvr[set][nf] = Bs[buf][(kk2 + khalf) * BNw + wn * 32 * NF + nf * 32 + l31];
    };
    auto mma_pair = [&](int kk2) {
        const int set = (kk2 >> 1) & 1;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = lu_mfma(avr[set][mf], bvr[set][nf], acc[mf][nf]);
    };
    for (int it = 0; it < n_it; ++it) {
        const int buf = it & 1;
        // prefetch of the next stage is issued behind the first MFMA group, its LDS stores before the last one
        // (see lu_conv.hip); unguarded: the last iteration re-fetches its own stage
        rd_pair(buf, 0);
        rd_pair(buf, 2);
        mma_pair(0);
        LU_SCHED_FENCE();
        if (it + 1 < n_it) advance_stage();
        load_stage();
        LU_SCHED_FENCE();
#pragma unroll
        for (int kk2 = 2; kk2 < KP - 2; kk2 += 2) {
            rd_pair(buf, kk2 + 2);
            mma_pair(kk2);
            LU_SCHED_FENCE();
        }
        store_stage(buf ^ 1);
        LU_SCHED_FENCE();
        mma_pair(KP - 2);
        __syncthreads();
    }

    float* slab = a.ws + (int64_t)blockIdx.z * a.slab;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 * MF + mf * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            int64_t base;
            bool rok;
            if (!THIN) {
                const int c = c0 + row;
                rok = c < a.C;
                base = ((int64_t)tap * a.C + c) * a.N;
            } else {
                const int j = j0 + row;
                rok = j < a.kk * a.C;
                base = (int64_t)j * a.N;
            }
            if (!rok) continue;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int n = n0 + wn * 32 * NF + nf * 32 + l31;
                if (n < a.N) slab[base + n] = acc[mf][nf][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// Kernel-row variant for stride-1 K x K convolutions (K = 3, 5; W % 16 == 0; C >= 64): one block accumulates ALL K
// horizontal taps (kw = 0..K-1) of one kernel row kh for a 64 (c) x 128 (n) tile.  A pipeline stage is a run of
// 16 consecutive pixels of one image row: the dy tile [16][128] is shared by the K taps and the x tile is the
// same run widened by K-1 pixels ([16+K-1][64]); tap kw simply reads it shifted by kw rows.  Load bytes per FLOP
// drop 2.3x versus one-tap-per-block; 8 waves (2 x 4), K accumulators each -> 4 waves/SIMD.
// ---------------------------------------------------------------------------------------------------------
// KPT = pixels per stage: 32 wherever W % 32 == 0 (half as many block-wide barriers per MFMA, two loader passes per stage, LDS 59 KB;
// the library's own choice since round 4's lean stage loads), 16 on the other widths (W % 16 == 0, and the ragged rows of RG).
// The K x-rows a k-pair reads (rows kk2 + khalf + t) overlap the next pair's (kk2 + 2 + ...) in K - 2 rows: they slide through
// registers and a pair fetches only two new values -- 3 instead of 6 LDS reads per 5 MFMAs (the LDS pipe of a CU with 16 resident waves
// was ~60 % busy on those reads).  (The form that re-read all K rows per pair, bit-identical, was kept as an A/B instance until round 6.)
template <int K, bool RG, int KPT = 16>      // RG: ragged widths (W % 16 != 0), a compile-time property so that the aligned instance pays nothing
__global__ __launch_bounds__(512, 4) void wgrad_row_kernel(WgradArgs a) {
    constexpr int KP = KPT;      // (shadows the file-level stage length inside this kernel)
    static_assert(KPT == 16 || (KPT == 32 && !RG), "32-pixel stages: aligned widths only");
    constexpr int BMw = 64, BNw = 128, XP = KP + K - 1, NT = 512;
    constexpr int XPASS = (XP * 16 + NT - 1) / NT, YPASS = KP * 32 / NT;      // float4 items per thread and stage
    __shared__ __attribute__((aligned(16))) float Xs[2][XP * BMw];
    __shared__ __attribute__((aligned(16))) float Ys[2][KP * BNw];
    __shared__ __attribute__((aligned(16))) float Bsum[16 * BNw];     // bias-gradient partial sums: one float4 slot per loader thread
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    // Workgroup id % 8 picks the XCD and x is the fastest grid index: with 8 (16, ...) column tiles the K * c_tiles blocks
    // that stream the same dy tile sit 8 ids apart, i.e. on ONE XCD, and share it in that L2.  Layers with 1 / 2 / 4 column
    // tiles fold 8 / n_tiles pixel slabs into grid.x to keep that property (a.xfold; L0: 28.7 GB fetched per launch without).
    const int nxt = a.xfold > 1 ? (int)gridDim.x / a.xfold : (int)gridDim.x;
    const int bx = a.xfold > 1 ? (int)blockIdx.x % nxt : (int)blockIdx.x;
    const int bz = a.xfold > 1 ? (int)blockIdx.z * a.xfold + (int)blockIdx.x / nxt : (int)blockIdx.z;
    if (bz >= a.splits) return;
    const int n0 = bx * BNw;
    const int kh = blockIdx.y / a.c_tiles;
    const int c0 = (blockIdx.y - kh * a.c_tiles) * BMw;
    // pixel slab of this block: a.chunk consecutive pixels (W % 16 == 0: every 16-pixel run lies inside one image row) -- or,
    // a.ragged (any other width): a.rows_per whole image rows, each walked in a.wst = ceil(W / 16) runs whose last one is
    // masked beyond the row end (the dy loads return zeros there, so the x pixels under them do not matter)
    const int64_t p_begin = RG ? 0 : (int64_t)bz * a.chunk;
    int64_t p_end = p_begin + a.chunk;
    if (p_end > a.M) p_end = a.M;
    int n_it = p_end > p_begin ? (int)((p_end - p_begin + KP - 1) / KP) : 0;
    const float* const zp = a.zero16 ? reinterpret_cast<const float*>(a.zero16) : lu_zero16;

    // the pixel run of a stage is uniform over the block: (frame, row, first column) advance by 16 pixels per stage
    int64_t pf = 0;          // frame index
    int oy = 0, ox0 = 0;
    if (RG) {
        const int64_t rows = a.M / a.Wout, r0 = (int64_t)bz * a.rows_per;
        const int64_t r1 = r0 + a.rows_per < rows ? r0 + a.rows_per : rows;
        n_it = r1 > r0 ? (int)(r1 - r0) * a.wst : 0;
        pf = r0 / a.Hout;
        oy = (int)(r0 - pf * a.Hout);
    } else {
        const int64_t p = p_begin < a.M ? p_begin : 0;
        pf = p / a.HWo;
        const int r = (int)(p - pf * a.HWo);
        oy = r / a.Wout;
        ox0 = r - oy * a.Wout;
    }
    const int xrow = tid >> 4, xq = tid & 15;      // x tile: XP rows x 16 float4 (threads < XP*16)
    const int yrow = tid >> 5, yq = tid & 31;      // dy tile: 16 rows x 32 float4
    float4 rx[XPASS], ry[YPASS];
#pragma unroll
    for (int i = 0; i < XPASS; ++i) rx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < YPASS; ++i) ry[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    // bias gradient = column sums of dy: the blocks of kernel row 0 / channel tile 0 see every dy element of their
    // (column tile, pixel slab) exactly once in `ry`, so they add it up on the side (exact fp32, no extra HBM pass).
    // The running sums live in LDS, one float4 slot per loader thread: nothing is held in registers across the MFMAs.
    // ... shared out over the gridDim.y = K * c_tiles blocks that stream this dy tile: block y takes the stages s = y mod
    // gridDim.y (one block doing all of it lags its siblings, which then stop sharing dy / x in L2: see the bf16 variant)
    const bool want_bias = a.bias_ws != nullptr;
    const int brc = gridDim.y, bme = blockIdx.y;
    int bphase = brc > 1 ? 1 : 0;      // (stage it + 1) mod brc, the stage the loop body accumulates
    float4* const bslot = reinterpret_cast<float4*>(&Bsum[yrow * BNw + 4 * yq]);
    if (want_bias) *bslot = make_float4(0.f, 0.f, 0.f, 0.f);
    auto bias_acc = [&]() {
        float4 t = *bslot;
#pragma unroll
        for (int i = 0; i < YPASS; ++i) {      // (rows yrow, yrow + 16: same column group, same slot)
            t.x += ry[i].x;
            t.y += ry[i].y;
            t.z += ry[i].z;
            t.w += ry[i].w;
        }
        *bslot = t;
    };
    // Lean stage loads (round 4, aligned widths; the bf16 kernel's treatment): a run never straddles two image rows and Wout == Win,
    // so consecutive runs are consecutive in memory -- ONE running pointer per operand piece, advanced by a constant (a frame wrap adds
    // the frame gap); what is tested per stage is wave-uniform (the input row under the run, first / last run of an image row, pixels
    // left in the slab) against per-thread constants.  The (frame, row, column) -> address arithmetic redone every stage (64-bit
    // multiplies, four bounds per piece) was ~80 instructions per wave between the MFMAs.
    const float* xptr[XPASS];
    const float* yptr[YPASS];
    bool xleft[XPASS], xright[XPASS], xin[XPASS];
    const bool ycol = n0 + 4 * yq < a.N;
    int64_t p_cur = p_begin;      // first pixel of the run the NEXT load_stage fetches
    if (!RG) {
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            const int xr = xrow + 32 * i;
            xin[i] = xr < XP && c0 + 4 * xq < a.C;
            xleft[i] = xr < a.pad_l;                  // outside the image in the FIRST run of a row
            xright[i] = xr >= KP + a.pad_l;           // ... in the LAST run of a row
            xptr[i] = a.x + pf * a.x_fs + ((int64_t)(oy + kh - a.pad_t) * a.Win + ox0 - a.pad_l + xr) * a.x_ps + c0 + 4 * xq;
        }
#pragma unroll
        for (int i = 0; i < YPASS; ++i)
            yptr[i] = a.dy + pf * a.dy_fs + ((int64_t)oy * a.Wout + ox0 + yrow + 16 * i) * a.dy_ps + n0 + 4 * yq;
    }
    auto load_stage = [&](int it) {
        if (!RG) {
            const int iy = oy + kh - a.pad_t;
            const bool rowok = iy >= 0 && iy < a.Hin, first = ox0 == 0, last = ox0 + KP == a.Wout;
            const int left = (int)(p_end - p_cur < KP ? p_end - p_cur : KP);      // pixels of this run inside the slab
#pragma unroll
            for (int i = 0; i < XPASS; ++i) {
                const bool okx = xin[i] && rowok && !(first && xleft[i]) && !(last && xright[i]);
                rx[i] = *reinterpret_cast<const float4*>(okx ? xptr[i] : zp);
            }
#pragma unroll
            for (int i = 0; i < YPASS; ++i) {
                const bool oky = ycol && yrow + 16 * i < left;
                ry[i] = *reinterpret_cast<const float4*>(oky ? yptr[i] : zp);
            }
            return;
        }
        const int iy = oy + kh - a.pad_t;
        const int c = c0 + 4 * xq;
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            const int xr = xrow + 32 * i;
            const int ix = ox0 - a.pad_l + xr;
            const bool okx = xr < XP && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win && c < a.C;
            const float* px = a.x + pf * a.x_fs + ((int64_t)iy * a.Win + ix) * a.x_ps + c;
            rx[i] = *reinterpret_cast<const float4*>(okx ? px : zp);
        }
        const int n = n0 + 4 * yq;
#pragma unroll
        for (int i = 0; i < YPASS; ++i) {
            const int yr = yrow + 16 * i;
            const bool oky = (RG ? ox0 + yr < a.Wout : p_begin + (int64_t)it * KP + yr < p_end) && n < a.N;
            const float* py = a.dy + pf * a.dy_fs + ((int64_t)oy * a.Wout + ox0 + yr) * a.dy_ps + n;
            ry[i] = *reinterpret_cast<const float4*>(oky ? py : zp);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XPASS; ++i)
            if (xrow + 32 * i < XP) *reinterpret_cast<float4*>(&Xs[buf][(xrow + 32 * i) * BMw + 4 * xq]) = rx[i];
#pragma unroll
        for (int i = 0; i < YPASS; ++i) *reinterpret_cast<float4*>(&Ys[buf][(yrow + 16 * i) * BNw + 4 * yq]) = ry[i];
    };
    const int64_t xstep = (int64_t)KP * a.x_ps, ystep = (int64_t)KP * a.dy_ps;
    const int64_t xwrap = a.x_fs - (int64_t)a.HWo * a.x_ps, ywrap = a.dy_fs - (int64_t)a.HWo * a.dy_ps;      // frame gap (0: dense)
    auto advance = [&]() {
        ox0 += KP;
        bool wrapped = false;
        if (ox0 >= a.Wout) {          // (W % 16 == 0, or ragged rows with a masked last run): a run never straddles two rows
            ox0 = 0;
            if (++oy == a.Hout) {
                oy = 0;
                ++pf;
                wrapped = true;
            }
        }
        if (!RG) {
            p_cur += KP;
#pragma unroll
            for (int i = 0; i < XPASS; ++i) xptr[i] += xstep + (wrapped ? xwrap : 0);
#pragma unroll
            for (int i = 0; i < YPASS; ++i) yptr[i] += ystep + (wrapped ? ywrap : 0);
        }
    };

    f32x16 acc[K];
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (n_it > 0) {
        load_stage(0);
        store_stage(0);
        if (want_bias && bme == 0) bias_acc();
    }
    __syncthreads();
    const int khalf = lane >> 5, l31 = lane & 31;
    // The LDS reads run AHEAD of the MFMAs (round 4).  A lane reads x rows m = 0 .. KP + K - 3 (+ khalf) and
    // KP / 2 dy values per stage; k-pair p multiplies rows 2p .. 2p + K - 1 by dy value p, i.e. it adds two new rows and one dy value
    // to what pair p - 1 held.  Pair p + 1's three values are requested before pair p's MFMAs are issued, through ONE base register
    // per operand and immediate offsets, so the waits the compiler places are counted and a wave never sits on an LDS round trip
    // with an empty MFMA queue (rounds 1-3: read -> s_waitcnt lgkmcnt(0) -> 2-3 MFMAs, covered only by the other resident waves).
    constexpr int NP = KP / 2, NX = KP + K - 2;
    constexpr int PD = 2;      // pairs of look-ahead (5x5 with 32-pixel stages: two loop-invariant registers spill for it and it is still 0.9 % faster than one)
    const float* const xrd = &Xs[0][khalf * BMw + wm * 32 + l31];
    const float* const yrd = &Ys[0][khalf * BNw + wn * 32 + l31];
    float xv[NX], bvv[NP];
    auto rd_pair = [&](int buf, int p) {
        bvv[p] = yrd[buf * (KP * BNw) + 2 * p * BNw];
#pragma unroll
        for (int m = (p == 0 ? 0 : 2 * p + K - 2); m < 2 * p + K; ++m) xv[m] = xrd[buf * (XP * BMw) + m * BMw];
    };
    auto mma_p = [&](int p) {
#pragma unroll
        for (int t = 0; t < K; ++t) acc[t] = lu_mfma(xv[2 * p + t], bvv[p], acc[t]);
    };
    for (int it = 0; it < n_it; ++it) {
        const int buf = it & 1;
        rd_pair(buf, 0);
        rd_pair(buf, 1);
        if (PD > 1) rd_pair(buf, 2);
        mma_p(0);
        LU_SCHED_FENCE();
        if (it + 1 < n_it) advance();
        load_stage(it + 1 < n_it ? it + 1 : it);
        LU_SCHED_FENCE();
#pragma unroll
        for (int p = 1; p < NP - 1; ++p) {
            if (p + PD < NP) rd_pair(buf, p + PD);
            mma_p(p);
            LU_SCHED_FENCE();
        }
        LU_SCHED_FENCE();
        store_stage(buf ^ 1);
        if (want_bias && it + 1 < n_it && bphase == bme) bias_acc();      // (the last iteration re-fetched its own run: not counted twice)
        bphase = bphase + 1 == brc ? 0 : bphase + 1;
        LU_SCHED_FENCE();
        mma_p(NP - 1);
        __syncthreads();
    }

    float* slab = a.ws + (int64_t)bz * a.slab;
    if ((a.N & 3) == 0 && (a.slab & 3) == 0 && (reinterpret_cast<uintptr_t>(a.ws) & 15) == 0) {
        // 16-byte slab stores (see wgrad_row_bf16_kernel): each wave turns its fragments round, eight channel rows at a time, in
        // a private slice of the dy tile's LDS (dead: the loop ended on a barrier); a lane then owns (row, four columns).
        float* const Exw = &Ys[0][0] + wave * (8 * 36);      // [8 rows][32 columns + 4]
        const int lp = lane >> 3, cq = lane & 7;
        const int n = n0 + wn * 32 + 4 * cq;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const int tap = kh * K + t;
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)      // row 8 qt + rr + 4 (lane >> 5) of the fragment
                    Exw[(rr + 4 * (lane >> 5)) * 36 + l31] = acc[t][4 * qt + rr];
                LU_WAVE_SYNC();
                const float4 v = *reinterpret_cast<const float4*>(&Exw[lp * 36 + 4 * cq]);
                const int c = c0 + wm * 32 + 8 * qt + lp;
                if (c < a.C && n < a.N) *reinterpret_cast<float4*>(&slab[((int64_t)tap * a.C + c) * a.N + n]) = v;
                LU_WAVE_SYNC();
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const int tap = kh * K + t;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = n0 + wn * 32 + l31;
                if (c < a.C && n < a.N) slab[((int64_t)tap * a.C + c) * a.N + n] = acc[t][r];
            }
        }
    }
    if (want_bias) {          // 16 row-threads per column group -> one sum per column (fixed order: deterministic)
        const float* red = Bsum;
        __syncthreads();
        if (tid < BNw) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[r * BNw + tid];
            if (n0 + tid < a.N) a.bias_ws[((int64_t)bz * brc + bme) * a.N + n0 + tid] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16-MFMA kernel-row variant (precision = 1; W % 32 == 0).  Block tile = CT (c) x 128 (n) x K taps of one kernel row,
// 8 waves, 32-pixel runs per stage = two v_mfma_f32_32x32x16_bf16 k-steps per tap.  The operands stay pixel-major in
// LDS ([pixel][channel] bf16), i.e. k runs DOWN the rows, so the k-contiguous MFMA fragments are fetched with the
// transposing LDS read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group); a tap is a row shift of the x
// tile, which keeps every read 8-byte aligned.  Row pitches of 192 / 320 B put the 4 rows of a group on distinct
// 64-byte bank segments.
//   XB / YB: x / dy are ALREADY bf16 in HBM (the bf16 BPTT tape: h sequence, dz) -- half the bytes, no conversion; an
//            fp32 operand is rounded while it is staged.
//   At bf16 rates a stage is ~1300 cycles of MFMA per CU for 17-35 KB of operands, i.e. the loads of ONE stage in
//   flight cannot cover L2 / HBM latency: the register staging is two stages deep (sets A / B, loop unrolled by two),
//   so a load has two MFMA phases to land.  Stages past the end of the pixel slab read the zero block.
//   Blocks are numbered so that all (kernel row, c-tile, n-tile) blocks of a pixel slab sit on ONE XCD (block id % 8)
//   and run back to back there: x and dy come from HBM once per slab and from that XCD's L2 for the other tiles.
// ---------------------------------------------------------------------------------------------------------
constexpr int PRB = 32;      // pixels per stage of the bf16 kernel
#ifdef LU_WG_ABL      // tools-only builds (tools/gpu/r04_wg_ablate.sh): parts of the 64-pixel-stage loop compiled out, to see what bounds it
#define LU_WGA(bit) ((LU_WG_ABL) & (bit))      // 1 global loads, 2 LDS stores, 4 stage barrier, 8 bias sums, 16 LDS fragment reads
#else
#define LU_WGA(bit) 0
#endif

//   S = 2: the stride-2 3x3 layers (first convolution of a down block).  A stage is still 32 OUTPUT pixels of one output
//   row; the x tile holds the 2 * 31 + K input pixels under them and a tap-t fragment reads every second tile row
//   (the transposing read takes one address per lane, so a row stride costs nothing).
//   PRBT = 64 (W % 64 == 0): 64-pixel stages.  All eight waves meet at a barrier per stage, so whatever a stage spends on
//   loads, LDS stores and cursor arithmetic (about as many issue cycles as 20 MFMAs take) is exposed once per stage;
//   twice the MFMAs per stage halves that share.  One register set then suffices (a load has a whole 2600-cycle stage to land).
//   R = K ("all taps", round 4; 3x3 only): the block owns ALL K x K taps of its (c, n) tile -- the x tile holds the K input rows
//   under the 32-pixel run, the dy tile is shared by K * K taps instead of K: 1.5x the MFMAs per staged byte and per stage
//   barrier, x and dy fetched once instead of K times per 128 columns.  The K * K * NFW accumulator tiles do not fit beside a
//   second wave on the SIMD, so the block is 64 channels x 128 columns on 8 waves (2 x 4, 32c x 32n x 9 taps = 144 accumulator
//   registers each, two waves per SIMD).  (Four "fat" waves -- 288 accumulators in AGPRs, one wave per SIMD -- measured slower: removed.)
//   DMA (round 4; bf16 operands, stride 1; the all-taps 3x3 form's default, +3.5 %; -3 % / -1.5 % on the 5x5 kernel-row form, whose DMA
//   instances were removed in round 6): the tiles go from global memory STRAIGHT into LDS (global_load_lds_dwordx4: a wave
//   instruction copies 64 x 16 bytes to 1 KB of consecutive LDS) -- no staging registers, no ds_write_b128 (13 issue cycles
//   each, 5 per thread and 64-pixel stage), no load -> store dependency inside the stage.  Compile-time ablations of the
//   register-staged loop (tools/gpu/r04_wg_ablate.sh, L1 5x5: 5.34 ms): without the global loads 4.06, without the LDS stores
//   4.35, without both 3.76 ms -- a third of the kernel was moving operands through registers.  A DMA'd row cannot be padded
//   (the 1 KB of an instruction is contiguous), so bank conflicts of the transposing fragment reads are avoided by a swizzle
//   applied to the SOURCE address instead: the 64-byte column segment s of row r is stored at segment s ^ (r & 3) (256-byte
//   rows: four rows of a fragment read land on four different bank quarters) resp. s ^ ((r >> 1) & 1) (128-byte rows).
//   The 8 + K - 1 x rows of a k-step are fetched once and the K tap fragments cut out of the registers -- a 16-bit funnel shift per register for odd
//   taps, four v_mov per tap whose first register is odd (an MFMA operand is an EVEN-aligned register tuple).  Round 5 measured the alternative --
//   every tap reading its own re-aligned rows, no VALU work left in the loop body, twice the LDS reads -- at -0.5 %: removed in round 6.
template <int K, int CT, bool XB, bool YB, int S = 1, int PRBT = PRB, int R = 1, int NWV = 8, bool DMA = false>      // CT = channel tile: 64 (waves 2 x 4, 32c x 32n each) or 128 (waves 4 x 2, 32c x 64n each)
__global__ __launch_bounds__(64 * NWV, 2) void wgrad_row_bf16_kernel(WgradArgs a) {
    constexpr int PRB = PRBT;      // (shadows the file-level default inside this kernel)
    constexpr int BMw = CT, BNw = 128, XP = S * (PRB - 1) + K, NT = 64 * NWV;
    constexpr int WMC = CT / 32, WNN = NWV / WMC, NFW = BNw / (32 * WNN);
    static_assert(!DMA || (XB && YB && S == 1), "LDS-DMA staging: raw bf16 copies of stride-1 tiles");
    constexpr int XLD = DMA ? BMw : BMw + 32, YLD = DMA ? BNw : BNw + 32;      // bf16 per LDS row: pitch = 64 B (mod 256 B); DMA: dense, swizzled
    constexpr int XE = XB ? 8 : 4, YE = YB ? 8 : 4;     // elements per 16-byte piece
    constexpr int PX = BMw / XE, PY = BNw / YE;         // pieces per tile row
    constexpr int XROWS = R * XP;                       // rows of the x tile: the R input rows back to back
    constexpr int XPASS = (XROWS * PX + NT - 1) / NT, YPASS = PRB * PY / NT;
    static_assert(PRB * PY % NT == 0, "dy tile: whole passes");
    static_assert(R == 1 || (R == K && S == 1), "all-taps form: stride 1, every kernel row");
    static_assert(WMC * WNN == NWV && NFW * WNN * 32 == BNw, "wave grid");
    constexpr int XRPI = 64 / PX, YRPI = 64 / PY;       // DMA: tile rows per wave instruction (64 lanes x 16 bytes)
    constexpr int XI = (XROWS + XRPI - 1) / XRPI, YI = PRB / YRPI;      // ... instructions per x / dy tile
    constexpr int XPA = DMA ? XI * XRPI : XPASS * NT / PX;   // x-tile rows ALLOCATED: every (thread, pass) owns a slot, so the stores need no guard
    LU_DYN_LDS(unsigned short, smem);      // Xs[2][XPA * XLD] | Ys[2][PRB * YLD] | Bred[8 * 128] floats (wgrad_row_bf16_lds)
    constexpr int NBUF = DMA ? 3 : 2;
    unsigned short* const Xs = smem;
    unsigned short* const Ys = smem + NBUF * XPA * XLD;
    float* const Bred = reinterpret_cast<float*>(Ys + NBUF * PRB * YLD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WMC, wn = wave / WMC;
    // XCD-aware numbering: block b runs on XCD b % 8; the `inner` tiles of a pixel slab are consecutive on one XCD
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int z = (jb / a.inner) * 8 + xcd;
    if (z >= a.splits) return;
    // inside a slab: (kernel row, c-tile) fastest, n-tile slowest.  An XCD holds 32 blocks at a time; with more tiles than
    // that per slab (L1: 5 x 2 x 8 = 80) the concurrent set should share dy columns -- every "round" then reads all of x
    // (the small operand) but a DISJOINT part of dy, instead of all of dy every round (PMC: 3.3x -> see profiles/)
    int tl = jb % a.inner;
    const int rc = a.inner / a.n_tiles;          // K * c_tiles (all-taps form: c_tiles)
    const int n0 = (tl / rc) * BNw;
    tl -= (tl / rc) * rc;
    const int kh = R == 1 ? tl / a.c_tiles : 0;  // first kernel row of this block
    const int c0 = (tl - kh * a.c_tiles) * BMw;
    const int64_t p_begin = (int64_t)z * a.chunk;
    int64_t p_end = p_begin + a.chunk;
    if (p_end > a.M) p_end = a.M;
    const int n_it = p_end > p_begin ? (int)((p_end - p_begin + PRB - 1) / PRB) : 0;
    const lu_u4* const zp = reinterpret_cast<const lu_u4*>(a.zero16 ? a.zero16 : (const void*)lu_zero16);

    // cursor of the NEXT stage to fetch: (frame, row, first column), stage index
    int64_t pf = 0;
    int oy = 0, ox0 = 0, ls = 0;
    {
        const int64_t p = p_begin < a.M ? p_begin : 0;
        pf = p / a.HWo;
        const int r = (int)(p - pf * a.HWo);
        oy = r / a.Wout;
        ox0 = r - oy * a.Wout;
    }
    // terms (precision 'bf16x3'): frame pf of the launch is frame pf % tf of term pf / tf; a term change is a frame step plus a
    // constant (x_tj / y_tj), counted down at the frame wraps -- nothing per stage
    int tleft = 0x7fffffff;
    int64_t xterm0 = 0, yterm0 = 0;
    if (a.tf > 0) {
        const int term = (int)(pf / a.tf);
        tleft = a.tf - (int)(pf - (int64_t)term * a.tf);
        xterm0 = term * a.x_tj;
        yterm0 = term * a.y_tj;
    }
    // The bias gradient (column sums of the dy tile) is shared out over the K * c_tiles blocks that stream the SAME dy tile:
    // block `bslot` takes the stages s with s % brc == bslot.  (One block doing all of it runs a few percent behind its
    // siblings, the group stops sharing dy / x in L2 and the launch's fabric traffic doubles: 3.2 -> 5.3 GB measured.)
    const bool want_bias = a.bias_ws != nullptr;
    const int brc = a.inner / a.n_tiles, bslot = kh * a.c_tiles + c0 / BMw;
    int bphase = 0;
    float bsum[YE];
#pragma unroll
    for (int e = 0; e < YE; ++e) bsum[e] = 0.f;

    // ---- stride 1: lean stage loads (round 4).  The first version recomputed every piece's address and bounds from (frame, row,
    // column) each stage: ~110 scalar and ~60 vector instructions per 40 MFMAs, short-circuit branches around the loads and two
    // GOT look-ups of the zero block with s_waitcnt lgkmcnt(0) behind them (which also wait for the LDS fragment reads in flight).
    // Now: one 64-bit element cursor per operand advanced by a constant per stage (S == 1 and Wout == Win: consecutive runs are
    // consecutive in memory; a frame wrap adds the frame gap), per-thread piece offsets and validity BITS computed once, and the
    // per-stage test is bit arithmetic: valid & row-inside(r) & not(left edge & lo) & not(right edge & hi).
    int xvo[XPASS], yvo[YPASS];            // element offset of piece i from the stage cursor
    unsigned xbits[XPASS], ybits = 0;      // bit 0 valid, 1 "left of the run" (xr < pad_l), 2 "right of it", 4.. kernel row r
    int64_t xcur = 0, ycur = 0;
    if constexpr (S == 1) {
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            const int item = tid + NT * i;
            const int rr = item / PX, q = item - rr * PX;
            const int r = R == 1 ? 0 : rr / XP, xr = rr - r * XP;
            xvo[i] = (r * a.Win + xr) * a.x_ps + XE * q;
            xbits[i] = ((rr < XROWS && c0 + XE * q < a.C) ? 1u : 0u) | (xr < a.pad_l ? 2u : 0u) | (xr >= PRB + a.pad_l ? 4u : 0u) |
                       ((unsigned)r << 4);
        }
#pragma unroll
        for (int i = 0; i < YPASS; ++i) {
            const int item = tid + NT * i;
            const int yr = item / PY, q = item - yr * PY;
            yvo[i] = yr * a.dy_ps + YE * q;
            ybits |= (n0 + YE * q < a.N ? 1u : 0u) << i;
        }
        xcur = pf * a.x_fs + xterm0 + ((int64_t)(oy + kh - a.pad_t) * a.Win + (ox0 - a.pad_l)) * a.x_ps + c0;
        ycur = pf * a.dy_fs + yterm0 + ((int64_t)oy * a.Wout + ox0) * a.dy_ps + n0;
    }
    const int64_t x_gap = a.x_fs - (int64_t)a.HWo * a.x_ps, y_gap = a.dy_fs - (int64_t)a.HWo * a.dy_ps;      // (S == 1: Hin * Win == HWo)
    const lu_u4* const zpa = zp;
    auto load_stage = [&](lu_u4 (&rx)[XPASS], lu_u4 (&ry)[YPASS]) {
        if constexpr (S == 1) {
            const unsigned live = ls < n_it ? 1u : 0u;
            const int iy0 = oy + kh - a.pad_t;
            unsigned rowbits = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) rowbits |= ((unsigned)(iy0 + r) < (unsigned)a.Hin ? live : 0u) << r;
            const unsigned edge = (ox0 == 0 ? 2u : 0u) | (ox0 + PRB >= a.Wout ? 4u : 0u);
            const unsigned short* const xb = reinterpret_cast<const unsigned short*>(a.x);
            const float* const xf = a.x;
#pragma unroll
            for (int i = 0; i < XPASS; ++i) {
                const unsigned ok = xbits[i] & (rowbits >> (xbits[i] >> 4)) & ((xbits[i] & edge) == 0u ? 1u : 0u);
                const lu_u4* pp = XB ? reinterpret_cast<const lu_u4*>(xb + xcur + xvo[i]) : reinterpret_cast<const lu_u4*>(xf + xcur + xvo[i]);
                rx[i] = *((ok & 1u) ? pp : zpa);
            }
            const unsigned short* const yb = reinterpret_cast<const unsigned short*>(a.dy);
#pragma unroll
            for (int i = 0; i < YPASS; ++i) {
                const unsigned ok = (ybits >> i) & live;
                const lu_u4* pp = YB ? reinterpret_cast<const lu_u4*>(yb + ycur + yvo[i]) : reinterpret_cast<const lu_u4*>(a.dy + ycur + yvo[i]);
                ry[i] = *((ok & 1u) ? pp : zpa);
            }
            ++ls;
            ox0 += PRB;
            xcur += PRB * a.x_ps;
            ycur += PRB * a.dy_ps;
            if (ox0 >= a.Wout) {          // W % 32 == 0: a run never straddles two rows
                ox0 = 0;
                if (++oy == a.Hout) {
                    oy = 0;
                    ++pf;
                    xcur += x_gap;
                    ycur += y_gap;
                    if (--tleft == 0) {      // (terms: the next frame belongs to the next term)
                        tleft = a.tf;
                        xcur += a.x_tj;
                        ycur += a.y_tj;
                    }
                }
            }
            return;
        }
        const bool live = ls < n_it;
        const int iy0 = S * oy + kh - a.pad_t;
        const int64_t xrow = pf * a.x_fs + ((int64_t)iy0 * a.Win + (S * ox0 - a.pad_l)) * a.x_ps + c0;
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            const int item = tid + NT * i;
            const int rr = item / PX, q = item - rr * PX;      // (functions of the thread index only: hoisted out of the loop)
            const int r = R == 1 ? 0 : rr / XP, xr = rr - r * XP;
            const int iy = iy0 + r, ix = S * ox0 - a.pad_l + xr;
            const bool ok = live && iy >= 0 && iy < a.Hin && rr < XROWS && ix >= 0 && ix < a.Win && c0 + XE * q < a.C;
            const int64_t off = xrow + ((int64_t)r * a.Win + xr) * a.x_ps + XE * q;
            const lu_u4* pp = XB ? reinterpret_cast<const lu_u4*>(reinterpret_cast<const unsigned short*>(a.x) + off)
                                 : reinterpret_cast<const lu_u4*>(a.x + off);
            rx[i] = *(ok ? pp : zp);
        }
        const int64_t yrow = pf * a.dy_fs + ((int64_t)oy * a.Wout + ox0) * a.dy_ps + n0;
#pragma unroll
        for (int i = 0; i < YPASS; ++i) {
            const int item = tid + NT * i;
            const int yr = item / PY, q = item - yr * PY;
            const bool ok = live && n0 + YE * q < a.N;
            const int64_t off = yrow + (int64_t)yr * a.dy_ps + YE * q;
            const lu_u4* pp = YB ? reinterpret_cast<const lu_u4*>(reinterpret_cast<const unsigned short*>(a.dy) + off)
                                 : reinterpret_cast<const lu_u4*>(a.dy + off);
            ry[i] = *(ok ? pp : zp);
        }
        ++ls;
        ox0 += PRB;
        if (ox0 >= a.Wout) {          // W % 32 == 0: a run never straddles two rows
            ox0 = 0;
            if (++oy == a.Hout) {
                oy = 0;
                ++pf;
            }
        }
    };
    auto put = [&](unsigned short* dst, const lu_u4& v, bool is_bf16) {
        if (is_bf16) {
            *reinterpret_cast<lu_u4*>(dst) = v;
        } else {
            unsigned* d2 = reinterpret_cast<unsigned*>(dst);
            d2[0] = lu_pack2bf(lu_bits2f(v.x), lu_bits2f(v.y));
            d2[1] = lu_pack2bf(lu_bits2f(v.z), lu_bits2f(v.w));
        }
    };
    auto store_stage = [&](int buf, const lu_u4 (&rx)[XPASS], const lu_u4 (&ry)[YPASS]) {
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            const int item = tid + NT * i;
            const int xr = item / PX, q = item - xr * PX;
            put(&Xs[buf * (XPA * XLD) + xr * XLD + XE * q], rx[i], XB);      // (rows >= XP: zeros into slots nobody reads)
        }
#pragma unroll
        for (int i = 0; i < YPASS; ++i) {
            const int item = tid + NT * i;
            const int yr = item / PY, q = item - yr * PY;
            put(&Ys[buf * (PRB * YLD) + yr * YLD + YE * q], ry[i], YB);
        }
    };
    auto bias_stage = [&](const lu_u4 (&ry)[YPASS]) {
        const bool mine = want_bias && bphase == bslot;      // (uniform)
        bphase = bphase + 1 == brc ? 0 : bphase + 1;
        if (mine) {               // bias gradient = column sums of dy (of the values the MFMA sees when dy is bf16)
#pragma unroll
            for (int i = 0; i < YPASS; ++i) {
                if (YB) {
                    const unsigned w4[4] = {ry[i].x, ry[i].y, ry[i].z, ry[i].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bsum[(2 * e) % YE] += lu_bits2f(w4[e] << 16);
                        bsum[(2 * e + 1) % YE] += lu_bits2f(w4[e] & 0xffff0000u);
                    }
                } else {
                    bsum[0] += lu_bits2f(ry[i].x);
                    bsum[1] += lu_bits2f(ry[i].y);
                    bsum[2] += lu_bits2f(ry[i].z);
                    bsum[3] += lu_bits2f(ry[i].w);
                }
            }
        }
    };

    f32x16 acc[R * K][NFW];      // [kernel row r of this block][tap t][column fragment]
#pragma unroll
    for (int t = 0; t < R * K; ++t)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][nf][r] = 0.f;

    // this lane's address inside a 4-row x 32-column transposed fetch: row (lane & 15) >> 2 (+ 8 for the upper half-wave),
    // columns 16 * ((lane >> 4) & 1) + 4 * (lane & 3)
    const int frow = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int fcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    // DMA: 64-byte column segment g of tile row r lives at segment g ^ swz(r); every row a lane reads is r0 + frow + (multiple of
    // 4), so the swizzle term is a per-lane constant (per kernel row kr of the all-taps form: XP need not be a multiple of 4)
    auto swz = [](int row, int row_bytes) { return row_bytes == 256 ? (row & 3) : ((row >> 1) & 1); };
    int xseg[R], yseg[NFW];
#pragma unroll
    for (int kr = 0; kr < R; ++kr) xseg[kr] = DMA ? ((wm ^ swz(kr * XP + frow, 2 * BMw)) - wm) * 32 : 0;
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) yseg[nf] = DMA ? (((wn * NFW + nf) ^ swz(frow, 2 * BNw)) - (wn * NFW + nf)) * 32 : 0;
    const int xoff = S * frow * XLD + wm * 32 + fcol, yoff = frow * YLD + wn * 32 * NFW + fcol;
    auto frag = [&](const unsigned short* base, int ld) {      // 8 consecutive k (rows) of this lane's column
        if (LU_WGA(16)) {
            lu_bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (short)(lane + e);
            return v;
        }
        const lu_bf16x4 lo = lu_lds_tr16(base), hi = lu_lds_tr16(base + 4 * ld);
        lu_bf16x8 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };
    // The K taps of a kernel row read the SAME x columns shifted by one pixel (= one k) each: fetch the 8 + K - 1 rows once
    // (NR transposing reads instead of 2 K) and cut the per-tap fragments out of the registers -- a tap-t fragment is
    // elements t .. t + 7 of the 4 NR fetched (a 16-bit funnel shift for odd t, pure renaming for even t).  The LDS read
    // traffic of the loop halves, and it was the busiest unit: 14 reads per 10 MFMAs at K = 5.
    constexpr int NR = (8 + K - 1 + 3) / 4;
    auto mma_half = [&](int buf, int j) {
        lu_bf16x8 bv[NFW];
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) bv[nf] = frag(&Ys[buf * (PRB * YLD) + yoff + 16 * j * YLD + 32 * nf + yseg[nf]], YLD);
        if constexpr (S == 1) {
#pragma unroll
            for (int kr = 0; kr < R; ++kr) {      // the R input rows of the tile (all-taps form) -- one for the kernel-row form
                short xw[4 * NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    lu_bf16x4 q4;
                    if (LU_WGA(16)) { q4[0] = (short)lane; q4[1] = (short)(lane + r); q4[2] = (short)j; q4[3] = (short)kr; }
                    else q4 = lu_lds_tr16(&Xs[buf * (XPA * XLD) + xoff + xseg[kr] + (kr * XP + 16 * j + 4 * r) * XLD]);
                    xw[4 * r] = q4[0]; xw[4 * r + 1] = q4[1]; xw[4 * r + 2] = q4[2]; xw[4 * r + 3] = q4[3];
                }
#pragma unroll
                for (int t = 0; t < K; ++t) {
                    lu_bf16x8 av;
#pragma unroll
                    for (int e = 0; e < 8; ++e) av[e] = xw[t + e];
#pragma unroll
                    for (int nf = 0; nf < NFW; ++nf) acc[kr * K + t][nf] = lu_mfma_bf16(av, bv[nf], acc[kr * K + t][nf]);
                }
            }
        } else {      // strided rows: consecutive k are S tile rows apart -- one pair of reads per tap
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const unsigned short* base = &Xs[buf * (XPA * XLD) + xoff + (S * 16 * j + t) * XLD];
                const lu_bf16x4 lo = lu_lds_tr16(base), hi = lu_lds_tr16(base + S * 4 * XLD);
                lu_bf16x8 av;
                av[0] = lo[0]; av[1] = lo[1]; av[2] = lo[2]; av[3] = lo[3];
                av[4] = hi[0]; av[5] = hi[1]; av[6] = hi[2]; av[7] = hi[3];
#pragma unroll
                for (int nf = 0; nf < NFW; ++nf) acc[t][nf] = lu_mfma_bf16(av, bv[nf], acc[t][nf]);
            }
        }
    };

    // ---- DMA staging: wave w issues tile instructions w, w + NWV, ... (x rows first, then dy rows) of the stage under the cursor.
    // Branch-free: per slot the piece offset, validity bits and LDS destination are per-thread constants; x and dy slots differ
    // only in which cursor / base they select (wave-uniform). ----
    constexpr int DSLOTS = (XI + YI + NWV - 1) / NWV;
    int dvo[DSLOTS], dlds[DSLOTS];
    unsigned dbits[DSLOTS];      // bit 0 valid, 1 lo, 2 hi, 4..7 row-bit index (kernel row r; R for a dy slot), 8 x slot
    if constexpr (DMA) {
#pragma unroll
        for (int k2 = 0; k2 < DSLOTS; ++k2) {
            const int ii = wave + NWV * k2;
            if (ii < XI) {
                const int rr = XRPI * ii + lane / PX;
                const int pc = (lane % PX) ^ (4 * swz(rr, 2 * BMw));      // source piece: the swizzle is applied on the way in
                const int r = R == 1 ? 0 : rr / XP, xr = rr - r * XP;
                dvo[k2] = (r * a.Win + xr) * a.x_ps + 8 * pc;
                dbits[k2] = ((rr < XROWS && c0 + 8 * pc < a.C) ? 1u : 0u) | (xr < a.pad_l ? 2u : 0u) | (xr >= PRB + a.pad_l ? 4u : 0u) |
                            ((unsigned)(rr < XROWS ? r : 0) << 4) | 256u;
                dlds[k2] = XRPI * ii * XLD;
            } else {
                const int yi = ii - XI;
                const int yr = YRPI * yi + lane / PY;
                const int pc = (lane % PY) ^ (4 * swz(yr, 2 * BNw));
                dvo[k2] = yr * a.dy_ps + 8 * pc;
                dbits[k2] = ((yi < YI && n0 + 8 * pc < a.N) ? 1u : 0u) | ((unsigned)R << 4);
                dlds[k2] = (yi < YI ? YRPI * yi : 0) * YLD;
            }
        }
    }
    auto dma_stage = [&](unsigned short* Xd, unsigned short* Yd) {
        const unsigned live = ls < n_it ? 1u : 0u;
        const int iy0 = oy + kh - a.pad_t;
        unsigned rowbits = live << R;
#pragma unroll
        for (int r = 0; r < R; ++r) rowbits |= ((unsigned)(iy0 + r) < (unsigned)a.Hin ? live : 0u) << r;
        const unsigned edge = (ox0 == 0 ? 2u : 0u) | (ox0 + PRB >= a.Wout ? 4u : 0u);
        const unsigned short* const xb = reinterpret_cast<const unsigned short*>(a.x) + xcur;
        const unsigned short* const yb = reinterpret_cast<const unsigned short*>(a.dy) + ycur;
        const unsigned short* const zs = reinterpret_cast<const unsigned short*>(a.zero16 ? a.zero16 : (const void*)lu_zero16);
#pragma unroll
        for (int k2 = 0; k2 < DSLOTS; ++k2) {
            const bool isx = wave + NWV * k2 < XI;           // (wave-uniform)
            if (wave + NWV * k2 < XI + YI) {                  // (wave-uniform; only the last slot can be empty)
                const unsigned ok = dbits[k2] & (rowbits >> ((dbits[k2] >> 4) & 15u)) & ((dbits[k2] & edge) == 0u ? 1u : 0u);
                const unsigned short* sp = (isx ? xb : yb) + dvo[k2];
                lu_glds16(reinterpret_cast<const float*>((ok & 1u) ? sp : zs), reinterpret_cast<float*>((isx ? Xd : Yd) + dlds[k2]));
            }
        }
        ++ls;
        ox0 += PRB;
        xcur += PRB * a.x_ps;
        ycur += PRB * a.dy_ps;
        if (ox0 >= a.Wout) {
            ox0 = 0;
            if (++oy == a.Hout) {
                oy = 0;
                ++pf;
                xcur += x_gap;
                ycur += y_gap;
                if (--tleft == 0) {
                    tleft = a.tf;
                    xcur += a.x_tj;
                    ycur += a.y_tj;
                }
            }
        }
    };
    // bias gradient of a DMA'd stage: column sums of its dy tile read back from LDS, by the same (row, piece) -> thread map and in
    // the same stage order as the register-staged form sums its pieces (bit-identical partial sums)
    auto bias_stage_lds = [&](const unsigned short* Yr) {
        const bool mine = want_bias && bphase == bslot;      // (uniform)
        bphase = bphase + 1 == brc ? 0 : bphase + 1;
        if (mine) {
#pragma unroll
            for (int i = 0; i < YPASS; ++i) {
                const int item = tid + NT * i;
                const int yr = item / PY, q = item - yr * PY;
                const lu_u4 v = *reinterpret_cast<const lu_u4*>(&Yr[yr * YLD + 8 * (q ^ (4 * swz(yr, 2 * BNw)))]);
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[(2 * e) % YE] += lu_bits2f(w4[e] << 16);
                    bsum[(2 * e + 1) % YE] += lu_bits2f(w4[e] & 0xffff0000u);
                }
            }
        }
    };
    // One stage of the DMA loop.  hipcc drains the DMA counter (s_waitcnt vmcnt(0)) in front of every transposing LDS read that
    // MAY alias a transfer in flight -- and with one dynamic LDS array everything may.  As __restrict__ PARAMETERS of an inlined
    // function the buffer being filled (Xd / Yd) and the buffer being read (Xr / Yr) carry scoped-noalias metadata, the waits
    // disappear (checked in the ISA: the only vmcnt wait left is the one in front of the stage barrier), and the transfers of
    // stage it + 1 run under the MFMAs of stage it.  The sched_barrier keeps their issue in front of the stage.
    auto dma_iter = [&](unsigned short* __restrict__ Xd, unsigned short* __restrict__ Yd, const unsigned short* __restrict__ Xr,
                        const unsigned short* __restrict__ Yr) {
        dma_stage(Xd, Yd);
#ifndef LU_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
        bias_stage_lds(Yr);
#pragma unroll
        for (int j = 0; j < PRB / 16; ++j) {
            lu_bf16x8 bv[NFW];
#pragma unroll
            for (int nf = 0; nf < NFW; ++nf) bv[nf] = frag(&Yr[yoff + 16 * j * YLD + 32 * nf + yseg[nf]], YLD);
#pragma unroll
            for (int kr = 0; kr < R; ++kr) {
                short xw[4 * NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const lu_bf16x4 q4 = lu_lds_tr16(&Xr[xoff + xseg[kr] + (kr * XP + 16 * j + 4 * r) * XLD]);
                    xw[4 * r] = q4[0]; xw[4 * r + 1] = q4[1]; xw[4 * r + 2] = q4[2]; xw[4 * r + 3] = q4[3];
                }
#pragma unroll
                for (int t = 0; t < K; ++t) {
                    lu_bf16x8 av;
#pragma unroll
                    for (int e = 0; e < 8; ++e) av[e] = xw[t + e];
#pragma unroll
                    for (int nf = 0; nf < NFW; ++nf) acc[kr * K + t][nf] = lu_mfma_bf16(av, bv[nf], acc[kr * K + t][nf]);
                }
            }
        }
    };

    const int l31 = lane & 31;
    if constexpr (DMA) {
        // THREE stage buffers, the transfers of two stages in flight: at 60 % of the bf16 peak a 64-pixel stage lasts ~1.5 us, less
        // than a loaded memory round trip -- one stage of look-ahead (what the staging registers can afford) exposes part of every
        // fetch.  The wait in front of the stage barrier is COUNTED: this wave's transfers of stage it + 2 may still be in flight,
        // those of stage it + 1 must have landed (raw s_barrier: __syncthreads' fence would drain the counter).
        constexpr int XB3 = XPA * XLD, YB3 = PRB * YLD;
        auto stage_sync = [&]() {
#ifdef LU_EMU
            __syncthreads();
#else
            constexpr int REM = (XI + YI) % NWV;             // waves below REM issue DSLOTS transfers per stage, the others one fewer
            if (REM == 0 || wave < REM) __builtin_amdgcn_s_waitcnt(0x0070 | DSLOTS);      // vmcnt(DSLOTS) lgkmcnt(0)
            else __builtin_amdgcn_s_waitcnt(0x0070 | (DSLOTS - 1));
            __builtin_amdgcn_s_barrier();
#endif
        };
        dma_stage(Xs, Ys);                 // stage 0
        dma_stage(Xs + XB3, Ys + YB3);     // stage 1
        stage_sync();
        int b0 = 0, b2 = 2;
        for (int it = 0; it < n_it; ++it) {
            // stage it + 2 goes to buffer b2, which every wave left behind at the last barrier (it held stage it - 1)
            dma_iter(Xs + b2 * XB3, Ys + b2 * YB3, Xs + b0 * XB3, Ys + b0 * YB3);
            stage_sync();
            b0 = b0 == 2 ? 0 : b0 + 1;
            b2 = b2 == 2 ? 0 : b2 + 1;
        }
#ifndef LU_EMU
        __builtin_amdgcn_s_waitcnt(0x0070);      // nothing of the zero-filled tail stages may land in the epilogue's exchange area
        __builtin_amdgcn_s_barrier();
#endif
    } else if constexpr (PRB == 32) {
        lu_u4 rxA[XPASS], ryA[YPASS], rxB[XPASS], ryB[YPASS];
        load_stage(rxA, ryA);              // stage 0
        bias_stage(ryA);
        store_stage(0, rxA, ryA);
        load_stage(rxA, ryA);              // stage 1
        load_stage(rxB, ryB);              // stage 2
        __syncthreads();
        for (int it = 0; it < n_it; it += 2) {
            mma_half(0, 0);
            LU_SCHED_FENCE();
            bias_stage(ryA);
            store_stage(1, rxA, ryA);      // stage it + 1 (requested two MFMA phases ago)
            load_stage(rxA, ryA);          // stage it + 3
            LU_SCHED_FENCE();
            mma_half(0, 1);
            __syncthreads();
            mma_half(1, 0);
            LU_SCHED_FENCE();
            bias_stage(ryB);
            store_stage(0, rxB, ryB);      // stage it + 2
            load_stage(rxB, ryB);          // stage it + 4
            LU_SCHED_FENCE();
            mma_half(1, 1);
            __syncthreads();
        }
    } else {
        lu_u4 rxA[XPASS], ryA[YPASS];
        load_stage(rxA, ryA);              // stage 0
        bias_stage(ryA);
        store_stage(0, rxA, ryA);
        load_stage(rxA, ryA);              // stage 1
        __syncthreads();
        for (int it = 0; it < n_it; ++it) {
            const int buf = it & 1;
            mma_half(buf, 0);
            if (!LU_WGA(8)) bias_stage(ryA);
            LU_SCHED_FENCE();
            // One branch-free scheduling region: the LDS stores of stage it + 1, the loads of stage it + 2 and the remaining
            // MFMAs of this stage are independent of each other -- spread the bookkeeping BETWEEN the MFMAs (an in-order
            // wave cannot hide it behind its own MFMAs otherwise, and the barrier keeps all waves in the same phase)
            if (!LU_WGA(2)) store_stage(buf ^ 1, rxA, ryA);      // stage it + 1 (requested one whole stage ago)
            if (!LU_WGA(1)) load_stage(rxA, ryA);                // stage it + 2
#pragma unroll
            for (int j = 1; j < PRB / 16; ++j) mma_half(buf, j);
#pragma unroll
            for (int g = 0; g < (PRB / 16 - 1) * R * K * NFW; ++g) {
                LU_SCHED_GROUP(0x008, 1);        // one MFMA ...
                LU_SCHED_GROUP(0x366, 8);        // ... then up to 8 non-MFMA instructions (VALU / SALU / VMEM / DS)
            }
            LU_SCHED_FENCE();
            if (!LU_WGA(4)) __syncthreads();
        }
    }

    float* slab = a.ws + (int64_t)z * a.slab;
    if ((a.N & 3) == 0 && (a.slab & 3) == 0 && (reinterpret_cast<uintptr_t>(a.ws) & 15) == 0) {
        // 16-byte slab stores.  A lane holds ONE column of 16 channel rows per fragment: from the registers that is 16 K NFW
        // four-byte stores per lane with a 64-bit address each -- on the 3x3 layers (64 stages per block) more time than the
        // block's MFMAs.  Each wave turns its fragments round in a private slice of the operand LDS (dead: the loop ended on a
        // barrier), no block barrier; a lane then owns (channel row, four consecutive columns).
        float* const Exw = reinterpret_cast<float*>(smem) + wave * (16 * 36);      // [16 rows][32 columns + 4]
        const int cq = lane & 7;
#pragma unroll
        for (int t = 0; t < R * K; ++t) {
            const int tap = kh * K + t;
#pragma unroll
            for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr)      // row 16 half + (rr & 3) + 8 (rr >> 2) + 4 (lane >> 5) of the fragment
                        Exw[((rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5)) * 36 + l31] = acc[t][nf][8 * half + rr];
                    LU_WAVE_SYNC();
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int lp = (lane >> 3) + 8 * q;
                        const int c = c0 + wm * 32 + 16 * half + lp;
                        const int n = n0 + wn * 32 * NFW + 32 * nf + 4 * cq;
                        const float4 v = *reinterpret_cast<const float4*>(&Exw[lp * 36 + 4 * cq]);
                        if (c < a.C && n < a.N) *reinterpret_cast<float4*>(&slab[((int64_t)tap * a.C + c) * a.N + n]) = v;
                    }
                    LU_WAVE_SYNC();
                }
        }
    } else {
#pragma unroll
        for (int t = 0; t < R * K; ++t) {
            const int tap = kh * K + t;
#pragma unroll
            for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = c0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int n = n0 + wn * 32 * NFW + 32 * nf + l31;
                    if (c < a.C && n < a.N) slab[((int64_t)tap * a.C + c) * a.N + n] = acc[t][nf][r];
                }
        }
    }
    if (want_bias) {          // rows of a piece column live in different lanes / waves: shuffle, then 8 wave partials (fixed order)
        constexpr int RPW = 64 / PY;      // tile rows per wave and pass: 4 (bf16 dy) or 2 (fp32 dy)
#pragma unroll
        for (int e = 0; e < YE; ++e) {
            float v = bsum[e];
            if (RPW == 4) v += lu_shfl_xor(v, 32);
            v += lu_shfl_xor(v, RPW == 4 ? 16 : 32);
            bsum[e] = v;
        }
        if (lane < PY) {
#pragma unroll
            for (int e = 0; e < YE; ++e) Bred[wave * BNw + YE * lane + e] = bsum[e];
        }
        __syncthreads();
        if (tid < BNw) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) s += Bred[w * BNw + tid];
            if (n0 + tid < a.N) a.bias_ws[((int64_t)z * brc + bslot) * a.N + n0 + tid] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Piece-aware kernel-row weight gradient of precision 'bf16x3' (round 6; lu_wgrad_desc.terms == 6).
// x and dy are lu_split6 tensors; their first three channel blocks are the three bf16 pieces of the fp32 operand -- x (order A):
// lo, mid, hi; dy (order B): hi, mid, lo.  The six products of the mode (lo hi + mid mid + hi lo + mid hi + hi mid + hi hi) were, in
// round 5, six times the FRAMES of wgrad_row_bf16_kernel (terms as frames): every piece of x and dy was staged and its fragments
// were read from LDS once per product it takes part in -- 12 bytes per element, 0.7 transposing LDS reads per MFMA, on a kernel the
// round-6 ablations (profiles/r06_wgrad_bf16_ablation.txt) show paying for exactly those two streams: MFMAs alone 94 % busy at
// 2.30 GHz, + fragment reads 81 % at 1.88 GHz, + staging 67 % at 1.76 GHz.  Here a stage holds the THREE pieces of a 32-pixel run
// of both operands (6 bytes per element), a piece's fragments are read once per k-step and feed every product the piece occurs in
// from registers: 21 reads per 60 MFMAs (0.35 per MFMA), half the staged bytes per MFMA, 120 MFMAs per wave and stage barrier
// instead of 40.  Block = 128 channels x 128 columns x one kernel row, 8 waves (4 x 2) of 32c x 64n x K taps as in the bf16 kernel.
// Staging is LDS-DMA (global_load_lds_dwordx4, no staging registers -- the 42 fragment registers of a k-step need the room): dense
// 256-byte rows, the 64-byte segment s of row r stored at s ^ (r & 3) by swizzling the SOURCE address; two stage buffers, the
// transfers of stage it + 1 run under the 120 MFMAs of stage it (~4 us: longer than a loaded memory round trip).
// Order of the products inside a k-step: the five small ones first, hi x hi last; per fp32 accumulator that is one rounding per
// MFMA of 16 pixels, 6 per 16 products -- fewer than the fp32 MFMA kernel's one per 2.
// The bias gradient (column sums of dy = hi + mid + lo) is read back from the staged dy tiles by the block whose turn it is.
// ---------------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(512, 2) void wgrad_row_x3_kernel(WgradArgs a) {
    constexpr int PRB = 32, BM = 128, BN = 128, XP = PRB + K - 1, NWV = 8, NFW = 2, NT = 512;
    constexpr int XPA = (XP + 3) / 4 * 4;              // x rows allocated per piece: whole wave instructions of 4 rows
    constexpr int XI1 = XPA / 4, YI1 = PRB / 4;        // wave instructions per piece tile
    constexpr int XI = 3 * XI1, YI = 3 * YI1;
    constexpr int XB1 = 3 * XPA * BM, YB1 = 3 * PRB * BN;      // bf16 elements per stage buffer
    constexpr int NR = (8 + K - 1 + 3) / 4;
    LU_DYN_LDS(unsigned short, smem);      // Xs[2][3][XPA][128] | Ys[2][3][32][128] | Bred[8 * 128] floats (wgrad_row_x3_lds)
    unsigned short* const Xs = smem;
    unsigned short* const Ys = smem + 2 * XB1;
    float* const Bred = reinterpret_cast<float*>(Ys + 2 * YB1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % 4, wn = wave / 4;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;      // XCD-aware numbering, as wgrad_row_bf16_kernel
    const int z = (jb / a.inner) * 8 + xcd;
    if (z >= a.splits) return;
    int tl = jb % a.inner;
    const int rc = a.inner / a.n_tiles;          // K * c_tiles
    const int n0 = (tl / rc) * BN;
    tl -= (tl / rc) * rc;
    const int kh = tl / a.c_tiles;
    const int c0 = (tl - kh * a.c_tiles) * BM;
    const int64_t p_begin = (int64_t)z * a.chunk;
    int64_t p_end = p_begin + a.chunk;
    if (p_end > a.M) p_end = a.M;
    const int n_it = p_end > p_begin ? (int)((p_end - p_begin + PRB - 1) / PRB) : 0;

    int64_t pf = 0;
    int oy = 0, ox0 = 0, ls = 0;
    {
        const int64_t p = p_begin < a.M ? p_begin : 0;
        pf = p / a.HWo;
        const int r = (int)(p - pf * a.HWo);
        oy = r / a.Wout;
        ox0 = r - oy * a.Wout;
    }
    int64_t xcur = pf * a.x_fs + ((int64_t)(oy + kh - a.pad_t) * a.Win + (ox0 - a.pad_l)) * a.x_ps + c0;
    int64_t ycur = pf * a.dy_fs + ((int64_t)oy * a.Wout + ox0) * a.dy_ps + n0;
    const int64_t x_gap = a.x_fs - (int64_t)a.HWo * a.x_ps, y_gap = a.dy_fs - (int64_t)a.HWo * a.dy_ps;
    const bool want_bias = a.bias_ws != nullptr;
    const int brc = rc, bslot = kh * a.c_tiles + c0 / BM;
    int bphase = 0;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

    // DMA: wave instruction ii = wave + 8 k2 of a stage (x pieces first, then dy pieces) copies 4 tile rows x 256 bytes.  A lane's part
    // of the source address -- row (lane >> 4) of the four, swizzled 16-byte piece -- is the same for every instruction (all of them
    // start on a multiple of 4 rows): ONE offset register per operand; which instruction it is lives in scalar registers.
    const int uw = LU_UNIFORM(wave);
    const int l4 = lane >> 4;
    const int pc = (lane & 15) ^ (4 * l4);            // source piece: the swizzle is applied on the way in
    // (fields selected by a run-time condition below are copied to locals first: a select between two MEMBERS of the by-value argument
    // struct becomes a select between their addresses, and the whole struct moves to scratch memory -- 424 bytes per lane in the first build)
    const int x_ps = a.x_ps, dy_ps = a.dy_ps;
    const int64_t x_pj = a.x_tj, y_pj = a.y_tj;
    const int lx = l4 * x_ps + 8 * pc, ly = l4 * dy_ps + 8 * pc;
    // per-lane validity as 0 / 1 words and bit arithmetic (lane-varying bools with && / || become exec-masked branches around each transfer)
    const unsigned x_ok = c0 + 8 * pc < a.C ? 1u : 0u, y_ok = n0 + 8 * pc < a.N ? 1u : 0u;
    const unsigned x_lo = l4 < a.pad_l ? 1u : 0u;                                    // first row group: rows left of the run
    const unsigned x_hi = 4 * (XI1 - 1) + l4 >= PRB + a.pad_l ? 1u : 0u;             // last row group: rows right of it ...
    const unsigned x_out = 4 * (XI1 - 1) + l4 < XP ? 0u : 1u;                        // ... and rows past the tile (allocation only)
    auto dma_stage = [&](unsigned short* Xd, unsigned short* Yd) {
        const unsigned live = ls < n_it ? 1u : 0u;
        const unsigned rowok = (unsigned)(oy + kh - a.pad_t) < (unsigned)a.Hin ? live : 0u;
        const unsigned e_lo = ox0 == 0 ? 1u : 0u, e_hi = ox0 + PRB >= a.Wout ? 1u : 0u;
        const unsigned short* const xb = reinterpret_cast<const unsigned short*>(a.x) + xcur;
        const unsigned short* const yb = reinterpret_cast<const unsigned short*>(a.dy) + ycur;
        const unsigned short* const zs = reinterpret_cast<const unsigned short*>(a.zero16 ? a.zero16 : (const void*)lu_zero16);
        // x instructions uw, uw + 8, ... < XI, then dy instructions uw, uw + 8, ... < YI: which operand a slot belongs to is a compile-time
        // property (a run-time select between the two base pointers in front of the valid ? source : zeros select sent the argument struct
        // to scratch memory in the first build); only the last x slot is partial (wave-uniform test)
#pragma unroll
        for (int k2 = 0; k2 < (XI + NWV - 1) / NWV; ++k2) {
            const int ii = uw + NWV * k2;
            if (NWV * (k2 + 1) <= XI || ii < XI) {
                const int p = ii / XI1, g = ii - p * XI1;      // piece, row group
                const unsigned bad = (g == 0 ? (e_lo & x_lo) : 0u) | (g == XI1 - 1 ? ((e_hi & x_hi) | x_out) : 0u);      // (g: wave-uniform)
                const unsigned ok = rowok & x_ok & (bad ^ 1u);
                const unsigned short* sp = xb + ((int64_t)4 * g * x_ps + p * x_pj) + lx;
                lu_glds16(reinterpret_cast<const float*>((ok & 1u) ? sp : zs), reinterpret_cast<float*>(Xd + (p * XPA + 4 * g) * BM));
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < (YI + NWV - 1) / NWV; ++k2) {
            const int ii = uw + NWV * k2;
            if (NWV * (k2 + 1) <= YI || ii < YI) {
                const int q = ii / YI1, g = ii - q * YI1;
                const unsigned ok = live & y_ok;
                const unsigned short* sp = yb + ((int64_t)4 * g * dy_ps + q * y_pj) + ly;
                lu_glds16(reinterpret_cast<const float*>((ok & 1u) ? sp : zs), reinterpret_cast<float*>(Yd + (q * PRB + 4 * g) * BN));
            }
        }
        ++ls;
        ox0 += PRB;
        xcur += PRB * x_ps;
        ycur += PRB * dy_ps;
        if (ox0 >= a.Wout) {          // W % 32 == 0: a run never straddles two rows
            ox0 = 0;
            if (++oy == a.Hout) {
                oy = 0;
                xcur += x_gap;
                ycur += y_gap;
            }
        }
    };

    f32x16 acc[K][NFW];
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][nf][r] = 0.f;

    // this lane inside a 4-row x 32-column transposed fetch (wgrad_row_bf16_kernel); every row a lane reads is (a multiple of 4) +
    // frow, so the swizzle term of its 64-byte segment is a per-lane constant
    const int frow = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int fcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int xoff = frow * BM + ((wm ^ (frow & 3)) * 32) + fcol;
    int yoff[NFW];
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) yoff[nf] = frow * BN + (((wn * NFW + nf) ^ (frow & 3)) * 32) + fcol;

    auto bias_stage_lds = [&](const unsigned short* Yr) {
        const bool mine = want_bias && bphase == bslot;      // (uniform)
        bphase = bphase + 1 == brc ? 0 : bphase + 1;
        if (mine) {      // thread -> (row tid / 16, 16-byte piece tid % 16) of each of the three dy piece tiles: dy = hi + mid + lo
            const int yr = tid >> 4, q16 = tid & 15;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const lu_u4 v = *reinterpret_cast<const lu_u4*>(&Yr[(q * PRB + yr) * BN + 8 * (q16 ^ (4 * (yr & 3)))]);
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += lu_bits2f(w4[e] << 16);
                    bsum[2 * e + 1] += lu_bits2f(w4[e] & 0xffff0000u);
                }
            }
        }
    };
    // fragment helpers (kernel scope, not nested in the stage function: nested closures kept their captures -- and with them the whole
    // argument struct -- in scratch memory in the first build)
    auto yfr = [&](const unsigned short* __restrict__ Yr, int q, int j, lu_bf16x8 (&bv)[NFW]) {
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) {
            const unsigned short* base = &Yr[(q * PRB + 16 * j) * BN + yoff[nf]];
            const lu_bf16x4 lo = lu_lds_tr16(base), hi = lu_lds_tr16(base + 4 * BN);
            bv[nf][0] = lo[0]; bv[nf][1] = lo[1]; bv[nf][2] = lo[2]; bv[nf][3] = lo[3];
            bv[nf][4] = hi[0]; bv[nf][5] = hi[1]; bv[nf][6] = hi[2]; bv[nf][7] = hi[3];
        }
    };
    auto xrows = [&](const unsigned short* __restrict__ Xr, int p, int j, short (&xw)[4 * NR]) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const lu_bf16x4 q4 = lu_lds_tr16(&Xr[(p * XPA + 16 * j + 4 * r) * BM + xoff]);
            xw[4 * r] = q4[0]; xw[4 * r + 1] = q4[1]; xw[4 * r + 2] = q4[2]; xw[4 * r + 3] = q4[3];
        }
    };
    auto group = [&](const short (&xw)[4 * NR], const lu_bf16x8 (&bv)[NFW]) {      // the K taps of one product
#pragma unroll
        for (int t = 0; t < K; ++t) {
            lu_bf16x8 av;
#pragma unroll
            for (int e = 0; e < 8; ++e) av[e] = xw[t + e];
#pragma unroll
            for (int nf = 0; nf < NFW; ++nf) acc[t][nf] = lu_mfma_bf16(av, bv[nf], acc[t][nf]);
        }
    };
    // One stage.  As __restrict__ PARAMETERS of an inlined function the buffer being filled and the buffer being read carry
    // scoped-noalias metadata: hipcc would otherwise drain the DMA counter (vmcnt(0)) in front of every transposing LDS read that
    // MAY alias a transfer in flight (see wgrad_row_bf16_kernel's DMA loop).
    auto stage = [&](unsigned short* __restrict__ Xd, unsigned short* __restrict__ Yd, const unsigned short* __restrict__ Xr,
                     const unsigned short* __restrict__ Yr) {
        dma_stage(Xd, Yd);
#ifndef LU_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
        bias_stage_lds(Yr);
#pragma unroll
        for (int j = 0; j < PRB / 16; ++j) {
            // x pieces 0 / 1 / 2 = lo / mid / hi, dy pieces 0 / 1 / 2 = hi / mid / lo
            lu_bf16x8 y_hi[NFW], y_q[NFW];
            short xa[4 * NR], xb_[4 * NR];
            yfr(Yr, 0, j, y_hi);
            xrows(Xr, 0, j, xa);
            group(xa, y_hi);      // lo  x hi
            xrows(Xr, 1, j, xb_);
            yfr(Yr, 1, j, y_q);
            group(xb_, y_q);      // mid x mid
            xrows(Xr, 2, j, xa);
            group(xb_, y_hi);     // mid x hi
            group(xa, y_q);       // hi  x mid
            yfr(Yr, 2, j, y_q);
            group(xa, y_q);       // hi  x lo
            group(xa, y_hi);      // hi  x hi
        }
    };

    auto stage_sync = [&]() {
#ifdef LU_EMU
        __syncthreads();
#else
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0): this wave's transfers of the next stage have landed
        __builtin_amdgcn_s_barrier();
#endif
    };
    dma_stage(Xs, Ys);      // stage 0
    stage_sync();
    for (int it = 0; it < n_it; ++it) {
        const int b = it & 1;
        stage(Xs + (b ^ 1) * XB1, Ys + (b ^ 1) * YB1, Xs + b * XB1, Ys + b * YB1);
        stage_sync();
    }

    const int l31 = lane & 31;
    float* slab = a.ws + (int64_t)z * a.slab;
    {
        // 16-byte slab stores: each wave turns its fragments round in a private slice of the (dead) operand LDS (see wgrad_row_bf16_kernel)
        float* const Exw = reinterpret_cast<float*>(smem) + wave * (16 * 36);      // [16 rows][32 columns + 4]
        const int cq = lane & 7;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const int tap = kh * K + t;
#pragma unroll
            for (int nf = 0; nf < NFW; ++nf)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr)
                        Exw[((rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5)) * 36 + l31] = acc[t][nf][8 * half + rr];
                    LU_WAVE_SYNC();
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int lp = (lane >> 3) + 8 * q;
                        const int c = c0 + wm * 32 + 16 * half + lp;
                        const int n = n0 + wn * 32 * NFW + 32 * nf + 4 * cq;
                        const float4 v = *reinterpret_cast<const float4*>(&Exw[lp * 36 + 4 * cq]);
                        if (c < a.C && n < a.N) *reinterpret_cast<float4*>(&slab[((int64_t)tap * a.C + c) * a.N + n]) = v;
                    }
                    LU_WAVE_SYNC();
                }
        }
    }
    if (want_bias) {          // rows of a piece column live in different lanes / waves: shuffle, then 8 wave partials (fixed order)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = bsum[e];
            v += lu_shfl_xor(v, 32);
            v += lu_shfl_xor(v, 16);
            bsum[e] = v;
        }
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) Bred[wave * BN + 8 * lane + e] = bsum[e];
        }
        __syncthreads();
        if (tid < BN) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) s += Bred[w * BN + tid];
            if (n0 + tid < a.N) a.bias_ws[((int64_t)z * brc + bslot) * a.N + n0 + tid] = s;
        }
    }
}

size_t wgrad_row_x3_lds(int K) {
    const int XPA = (32 + K - 1 + 3) / 4 * 4;
    return (size_t)(2 * 3 * XPA * 128 + 2 * 3 * 32 * 128) * sizeof(unsigned short) + 8 * 128 * sizeof(float);
}

// ---------------------------------------------------------------------------------------------------------
// All-taps variant for the narrow decoder layers (stride-1 3x3, C <= 64, N <= 64, W % 16 == 0): with so few channels a
// 64 x 128 kernel-row tile is mostly padding and the layer is bound by reading x and dy once per kernel row (or, in
// wgrad_kernel, once per tap).  Here one block owns the WHOLE 9 x C x N gradient of a pixel slab: a stage is a 16-pixel run
// of one image row, the dy tile [16][64] is shared by all nine taps and the x tile holds the three input rows
// [3][16 + 2][64]; the (tap, 32-channel, 32-column) accumulator tiles (at most 36) are dealt round-robin to the 8 waves.
// x and dy are read exactly once.
// ---------------------------------------------------------------------------------------------------------
template <int MAXT>      // accumulator tiles per wave: ceil(9 * ceil(C/32) * ceil(N/32) / 8) = 2, 3 or 5
__global__ __launch_bounds__(512, 2) void wgrad_small3_kernel(WgradArgs a) {
    constexpr int K = 3, CT = 64, NTL = 64, XP = KP + K - 1, NT = 512;
    __shared__ __attribute__((aligned(16))) float Xs[2][K * XP * CT];
    __shared__ __attribute__((aligned(16))) float Ys[2][KP * NTL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cfr = (a.C + 31) >> 5, nfr = (a.N + 31) >> 5;            // 32-wide fragments in use (1 or 2 each)
    const int n_tiles = K * K * cfr * nfr;
    const int64_t p_begin = (int64_t)blockIdx.z * a.chunk;
    int64_t p_end = p_begin + a.chunk;
    if (p_end > a.M) p_end = a.M;
    const int n_it = p_end > p_begin ? (int)((p_end - p_begin + KP - 1) / KP) : 0;
    const float* const zp = a.zero16 ? reinterpret_cast<const float*>(a.zero16) : lu_zero16;

    int64_t pf = 0;
    int oy = 0, ox0 = 0;
    {
        const int64_t p = p_begin < a.M ? p_begin : 0;
        pf = p / a.HWo;
        const int r = (int)(p - pf * a.HWo);
        oy = r / a.Wout;
        ox0 = r - oy * a.Wout;
    }
    // x tile: K rows x XP pixels x 16 float4 = 864 items (two passes); dy tile: 16 pixels x 16 float4 (threads < 256)
    const int xq = tid & 15;
    const int xi0 = tid >> 4, xi1 = xi0 + NT / 16;               // item -> (kernel row, pixel) = (item / XP, item % XP)
    const int yq = tid & 15, yrow = tid >> 4;
    float4 rx0 = make_float4(0.f, 0.f, 0.f, 0.f), rx1 = rx0, ry = rx0;
    const bool want_bias = a.bias_ws != nullptr;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_x = [&](int item) -> float4 {
        const int kh = item / XP, xp = item - kh * XP;
        const int iy = oy + kh - a.pad_t, ix = ox0 - a.pad_l + xp;
        const int c = 4 * xq;
        const bool ok = item < K * XP && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win && c < a.C;
        const float* px = a.x + pf * a.x_fs + ((int64_t)iy * a.Win + ix) * a.x_ps + c;
        return *reinterpret_cast<const float4*>(ok ? px : zp);
    };
    auto load_stage = [&](int it) {
        rx0 = load_x(xi0);
        rx1 = load_x(xi1);
        const int n = 4 * yq;
        const bool oky = tid < 256 && p_begin + (int64_t)it * KP + yrow < p_end && n < a.N;
        const float* py = a.dy + pf * a.dy_fs + ((int64_t)oy * a.Wout + ox0 + yrow) * a.dy_ps + n;
        ry = *reinterpret_cast<const float4*>(oky ? py : zp);
    };
    auto store_stage = [&](int buf) {
        *reinterpret_cast<float4*>(&Xs[buf][xi0 * CT + 4 * xq]) = rx0;
        if (xi1 < K * XP) *reinterpret_cast<float4*>(&Xs[buf][xi1 * CT + 4 * xq]) = rx1;
        if (tid < 256) *reinterpret_cast<float4*>(&Ys[buf][yrow * NTL + 4 * yq]) = ry;
    };
    auto advance = [&]() {
        ox0 += KP;
        if (ox0 >= a.Wout) {
            ox0 = 0;
            if (++oy == a.Hout) {
                oy = 0;
                ++pf;
            }
        }
    };

    // this wave's accumulator tiles: id = wave + 8 j -> (tap, channel fragment, column fragment)
    int t_x[MAXT], t_y[MAXT];          // LDS offsets of the tile's A column (x) and B column (dy)
    bool t_ok[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
        const int id = wave + 8 * j;
        t_ok[j] = id < n_tiles;
        const int tap = id / (cfr * nfr), rem = id - tap * (cfr * nfr);
        const int cf = rem / nfr, nf = rem - cf * nfr;
        const int kh = tap / K, kw = tap - kh * K;
        t_x[j] = t_ok[j] ? (kh * XP + kw) * CT + 32 * cf + (lane & 31) : 0;
        t_y[j] = t_ok[j] ? 32 * nf + (lane & 31) : 0;
    }
    f32x16 acc[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    if (n_it > 0) {
        load_stage(0);
        store_stage(0);
        if (want_bias) {
            bsum.x += ry.x; bsum.y += ry.y; bsum.z += ry.z; bsum.w += ry.w;
        }
    }
    __syncthreads();
    const int khalf = lane >> 5;
    for (int it = 0; it < n_it; ++it) {
        const int buf = it & 1;
        if (it + 1 < n_it) advance();
        load_stage(it + 1 < n_it ? it + 1 : it);
        LU_SCHED_FENCE();
#pragma unroll
        for (int kk2 = 0; kk2 < KP; kk2 += 2) {
#pragma unroll
            for (int j = 0; j < MAXT; ++j) {      // (a slot beyond n_tiles recomputes tile 0 and is dropped: no branch here)
                const float av = Xs[buf][(kk2 + khalf) * CT + t_x[j]];
                const float bv = Ys[buf][(kk2 + khalf) * NTL + t_y[j]];
                acc[j] = lu_mfma(av, bv, acc[j]);
            }
        }
        LU_SCHED_FENCE();
        store_stage(buf ^ 1);
        if (want_bias && it + 1 < n_it) {
            bsum.x += ry.x; bsum.y += ry.y; bsum.z += ry.z; bsum.w += ry.w;
        }
        __syncthreads();
    }

    float* slab = a.ws + (int64_t)blockIdx.z * a.slab;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
        const int id = wave + 8 * j;
        if (id >= n_tiles) continue;
        const int tap = id / (cfr * nfr), rem = id - tap * (cfr * nfr);
        const int cf = rem / nfr, nf = rem - cf * nfr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * cf + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int n = 32 * nf + (lane & 31);
            if (c < a.C && n < a.N) slab[((int64_t)tap * a.C + c) * a.N + n] = acc[j][r];
        }
    }
    if (want_bias) {          // 16 pixel-row threads per column group -> one sum per column (fixed order)
        float* red = Ys[0];
        __syncthreads();
        if (tid < 256) *reinterpret_cast<float4*>(&red[yrow * NTL + 4 * yq]) = bsum;
        __syncthreads();
        if (tid < NTL) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < KP; ++r) sum += red[r * NTL + tid];
            if (tid < a.N) a.bias_ws[(int64_t)blockIdx.z * a.N + tid] = sum;
        }
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, int64_t slab, int splits, float* __restrict__ dw,
                                    int C, int N, int64_t tap_stride, int row_stride, float beta,
                                    const float* __restrict__ bias_ws, float* __restrict__ dbias, float dbias_beta,
                                    int bias_rows, int dw_blocks, int vec4) {
    // blocks [0, dw_blocks): the weight gradient.  Blocks beyond: the bias gradient riding on the same launch (its own
    // 40-microsecond launch per layer added up to 0.7 ms per step) -- 16 columns x 16 row lanes per block over the
    // [bias_rows][N] partial column sums (a few hundred rows: one thread per column would be one long dependent chain).
    if ((int)blockIdx.x >= dw_blocks) {
        __shared__ float red[16][17];
        const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
        const int n = ((int)blockIdx.x - dw_blocks) * 16 + cl;
        float s = 0.f;
        if (n < N)
            for (int z = rl; z < bias_rows; z += 16) s += bias_ws[(int64_t)z * N + n];
        red[rl][cl] = s;
        __syncthreads();
        if (rl == 0 && n < N) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 16; j += 4) t += (red[j][cl] + red[j + 1][cl]) + (red[j + 2][cl] + red[j + 3][cl]);
            dbias[n] = (dbias_beta != 0.f ? dbias_beta * dbias[n] : 0.f) + t;
        }
        return;
    }
    if (vec4 == 2) {      // tiny slab, many slabs (thin / 1x1 layers: a few hundred elements x 1024 slabs): 16 elements x 16 slab
                          // lanes per block, fixed order -- one thread per element would walk all the slabs alone (155 us measured)
        __shared__ float red[16][17];
        const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
        const int64_t i = (int64_t)blockIdx.x * 16 + cl;
        float s = 0.f;
        if (i < slab)
            for (int z = rl; z < splits; z += 16) s += ws[(int64_t)z * slab + i];
        red[rl][cl] = s;
        __syncthreads();
        if (rl == 0 && i < slab) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 16; j += 4) t += (red[j][cl] + red[j + 1][cl]) + (red[j + 2][cl] + red[j + 3][cl]);
            const int64_t row = i / N;
            const int n = (int)(i - row * N);
            const int64_t tap = row / C;
            const int c = (int)(row - tap * C);
            float* o = dw + tap * tap_stride + (int64_t)c * row_stride + n;
            *o = (beta != 0.f ? beta * *o : 0.f) + t;
        }
        return;
    }
    if (vec4) {      // four consecutive columns per thread (N % 4 == 0: same (tap, c) row), 16-byte loads, four slabs in flight;
                     // every element is still summed over the slabs in order 0, 1, 2, ...: bit-identical to the scalar form
        const int64_t slab4 = slab >> 2;
        for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < slab4; i4 += (int64_t)dw_blocks * blockDim.x) {
            const int64_t i = i4 << 2;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            int z = 0;
            for (; z + 4 <= splits; z += 4) {
                const float4 a0 = *reinterpret_cast<const float4*>(ws + (int64_t)z * slab + i);
                const float4 a1 = *reinterpret_cast<const float4*>(ws + (int64_t)(z + 1) * slab + i);
                const float4 a2 = *reinterpret_cast<const float4*>(ws + (int64_t)(z + 2) * slab + i);
                const float4 a3 = *reinterpret_cast<const float4*>(ws + (int64_t)(z + 3) * slab + i);
                s.x = (((s.x + a0.x) + a1.x) + a2.x) + a3.x;
                s.y = (((s.y + a0.y) + a1.y) + a2.y) + a3.y;
                s.z = (((s.z + a0.z) + a1.z) + a2.z) + a3.z;
                s.w = (((s.w + a0.w) + a1.w) + a2.w) + a3.w;
            }
            for (; z < splits; ++z) {
                const float4 a0 = *reinterpret_cast<const float4*>(ws + (int64_t)z * slab + i);
                s.x += a0.x;
                s.y += a0.y;
                s.z += a0.z;
                s.w += a0.w;
            }
            const int64_t row = i / N;
            const int n = (int)(i - row * N);
            const int64_t tap = row / C;
            const int c = (int)(row - tap * C);
            float4* o = reinterpret_cast<float4*>(dw + tap * tap_stride + (int64_t)c * row_stride + n);
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (beta != 0.f) t = *o;
            s.x = (beta != 0.f ? beta * t.x : 0.f) + s.x;      // (the scalar form's expression, element by element)
            s.y = (beta != 0.f ? beta * t.y : 0.f) + s.y;
            s.z = (beta != 0.f ? beta * t.z : 0.f) + s.z;
            s.w = (beta != 0.f ? beta * t.w : 0.f) + s.w;
            *o = s;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slab; i += (int64_t)dw_blocks * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += ws[(int64_t)z * slab + i];
        const int64_t row = i / N;
        const int n = (int)(i - row * N);
        const int64_t tap = row / C;
        const int c = (int)(row - tap * C);
        float* o = dw + tap * tap_stride + (int64_t)c * row_stride + n;
        *o = (beta != 0.f ? beta * *o : 0.f) + s;
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// device address of this translation unit's lu_zero16 on the current device (looked up once per device; null on failure: the kernel then takes the symbol itself)
const void* zero16_address() {
#ifdef LU_EMU
    return lu_zero16;
#else
    return LU_SYMBOL_ADDRESS(lu_zero16);      // per device (a process-wide cache handed device 0's address to device 1)
#endif
}

// dynamic LDS of wgrad_row_bf16_kernel<K, CT, *, *, S, PRBT>
size_t wgrad_row_bf16_lds(int K, int CT, int S, int prb, bool xb, int R = 1, int NT = 512, bool dma = false) {
    const int XP = S * (prb - 1) + K, XLD = dma ? CT : CT + 32, YLD = dma ? 128 : 128 + 32, PX = CT / (xb ? 8 : 4);
    const int XPA = dma ? (R * XP + 64 / PX - 1) / (64 / PX) * (64 / PX)      // whole wave instructions of 64 / PX rows
                        : (R * XP * PX + NT - 1) / NT * NT / PX;              // rows allocated = pass coverage (see the kernel)
    const int nbuf = dma ? 3 : 2;
    return (size_t)(nbuf * XPA * XLD + nbuf * prb * YLD) * sizeof(unsigned short) + 8 * 128 * sizeof(float);
}

}  // namespace

extern "C" size_t lu_conv2d_wgrad_workspace_bytes(const lu_wgrad_desc* d) {
    if (!d) return 0;
    int splits = d->splits > 0 ? d->splits : 1;
    // slabs + the bias rows: up to k * ceil(C / 64) partial column sums per split (the bf16 kernel-row variant shares the
    // bias work out over the blocks of a dy tile), one for the other variants
    return (size_t)splits * (d->k * d->k * (size_t)d->C + (d->dbias ? (size_t)d->k * ((d->C + 63) / 64) : 0)) * d->N * sizeof(float);
}

extern "C" int lu_conv2d_wgrad(const lu_wgrad_desc* d, lu_stream_t stream) {
    LU_REQUIRE(d && d->x && d->dy && d->dw && d->workspace, "lu_conv2d_wgrad: null pointer");
    LU_REQUIRE(d->k >= 1 && d->k <= 7 && (d->stride == 1 || d->stride == 2), "lu_conv2d_wgrad: bad k/stride");
    LU_REQUIRE(d->C > 0 && d->N > 0 && d->frames > 0, "lu_conv2d_wgrad: empty problem");
    const int splits = d->splits > 0 ? d->splits : 1;
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const float*)d->x;
    a.dy = (const float*)d->dy;
    a.x_fs = d->x_frame_stride;
    a.dy_fs = d->dy_frame_stride;
    a.x_ps = d->x_pix_stride;
    a.dy_ps = d->dy_pix_stride;
    a.C = d->C;
    a.N = d->N;
    // terms == 6 + LU_WGRAD_F_PIECES3 (ABI v12): the piece-aware kernel of precision 'bf16x3' -- all six products from the first three channel blocks of the
    // two split6 tensors in ONE pass over the frames (wgrad_row_x3_kernel); 2 .. 5: that many products with the terms as frames
    const bool pieces3 = d->terms == 6 && (d->flags & LU_WGRAD_F_PIECES3);
    const int terms = (d->terms > 1 && !pieces3) ? d->terms : 1;
    a.M = (int64_t)d->frames * terms * d->Hout * d->Wout;
    if (terms > 1) {
        a.tf = d->frames;
        a.x_tj = d->x_term_stride - (int64_t)d->frames * d->x_frame_stride;
        a.y_tj = d->dy_term_stride - (int64_t)d->frames * d->dy_frame_stride;
    }
    if (pieces3) {      // (the kernel reads the fields as the piece strides)
        a.x_tj = d->x_term_stride;
        a.y_tj = d->dy_term_stride;
    }
    a.chunk = ((a.M + splits - 1) / splits + 63) / 64 * 64;     // multiple of every kernel's pixel run (16 / 32 / 64)
    a.HWo = d->Hout * d->Wout;
    a.Wout = d->Wout;
    a.Hout = d->Hout;
    a.Hin = d->Hin;
    a.Win = d->Win;
    a.k = d->k;
    a.kk = d->k * d->k;
    a.stride = d->stride;
    a.pad_t = d->pad_t;
    a.pad_l = d->pad_l;
    a.ws = (float*)d->workspace;
    a.slab = (int64_t)a.kk * d->C * d->N;
    a.bias_ws = d->dbias ? a.ws + (int64_t)splits * a.slab : nullptr;
    a.zero16 = zero16_address();
    const bool xvec32 = d->x_dtype == LU_F32 && (d->C % 4 == 0) && (d->x_pix_stride % 4 == 0) && (d->x_frame_stride % 4 == 0) && aligned16(d->x);
    const bool yvec32 = d->dy_dtype == LU_F32 && (d->N % 4 == 0) && (d->dy_pix_stride % 4 == 0) && (d->dy_frame_stride % 4 == 0) && aligned16(d->dy);
    dim3 block(256);
#define LU_WG(MF_, NF_, WM_, WN_, THIN_, YV_, GY_)                                                          \
    do {                                                                                                    \
        const unsigned n_tiles = (unsigned)((d->N + 32 * NF_ * WN_ - 1) / (32 * NF_ * WN_));                 \
        dim3 grid(n_tiles, (unsigned)(GY_), (unsigned)splits);                                              \
        LU_LAUNCH((wgrad_kernel<MF_, NF_, WM_, WN_, THIN_, YV_>), grid, block, stream, a);                  \
    } while (0)
    LU_REQUIRE(d->phase >= 0 && d->phase <= 2, "lu_conv2d_wgrad: phase must be 0 (all), 1 (partial sums) or 2 (reduce)");
    const bool xb = d->x_dtype == LU_BF16, yb = d->dy_dtype == LU_BF16;
    const bool xvec = xvec32 || (xb && d->C % 8 == 0 && d->x_pix_stride % 8 == 0 && d->x_frame_stride % 8 == 0 && aligned16(d->x));
    const bool yvec = yvec32 || (yb && d->N % 8 == 0 && d->dy_pix_stride % 8 == 0 && d->dy_frame_stride % 8 == 0 && aligned16(d->dy));
    // (ragged: W % 16 != 0 -- fp32 kernel only, rows walked in ceil(W / 16) runs with a masked tail; from 40 pixels on, where the
    // masked share is <= 17 %: config-4's 248- and 124-pixel levels ran the one-tap-per-block kernel at 113 TFLOP/s without it)
    const bool ragged_w = d->Wout % 16 != 0 && d->Wout >= 40 && !xb && !yb && !(d->flags & LU_WGRAD_F_NO_RAGGED);
    // bf16 mode: the narrow decoder layers (C >= 32) take the bf16 kernel-row variant too (masked channel / column tiles: the
    // layers are HBM-bound, what counts is that x / dy are read once per slab and the MFMA is 16x the fp32 one)
    const bool narrow_bf16 = d->precision == 1 && d->Wout % PRB == 0 && !(d->flags & LU_WGRAD_F_NO_NARROW_BF16);
    const bool row_variant = xvec && yvec && d->stride == 1 && (d->k == 3 || d->k == 5) && (d->Wout % 16 == 0 || ragged_w) &&
                             d->C >= (narrow_bf16 ? 32 : 64) && d->Wout == d->Win && d->Hout == d->Hin &&
                             !(d->flags & LU_WGRAD_F_NO_ROW);
    const bool small3 = !xb && !yb && xvec && yvec && d->stride == 1 && d->k == 3 && d->C <= 64 && d->N <= 64 &&
                        d->Wout % 16 == 0 && d->Wout == d->Win && d->Hout == d->Hin && !(d->flags & LU_WGRAD_F_NO_SMALL3) &&
                        !(narrow_bf16 && row_variant);
    // (a 1x1 layer with >= 32 channels also fits the bf16 kernel-row scheme: one tap, no halo -- the im2col chunk of a thin input)
    const bool row_k1 = xvec && yvec && d->stride == 1 && d->k == 1 && d->precision == 1 && d->C >= 32 && d->Wout == d->Win &&
                        d->Hout == d->Hin && !d->dbias && !(d->flags & LU_WGRAD_F_NO_ROW);
    // ... and the stride-2 3x3 layers (x rows read with a stride; the general fp32 kernel ran them at 73 TFLOP/s in bf16 mode)
    // (round 4: 5x5 as well -- Networks.DEFAULT_NET_DOWN_PARAMS' down blocks; 64-channel tiles: five accumulator tiles per wave)
    const bool row_s2 = xvec && yvec && d->stride == 2 && (d->k == 3 || d->k == 5) && d->precision == 1 && d->C >= 64 &&
                        d->Hout == (d->Hin + 1) / 2 && d->Wout == (d->Win + 1) / 2 && !(d->flags & LU_WGRAD_F_NO_ROW);
    const bool row_bf16 = ((row_variant && !small3) || row_k1 || row_s2) && d->precision == 1 && d->Wout % PRB == 0;
    LU_REQUIRE((!xb && !yb) || row_bf16,
               "lu_conv2d_wgrad: bf16 operands need the bf16 kernel-row variant (precision 1, stride-1 3x3 / 5x5 with C >= 64 or 1x1 with C >= 32, "
               "C %% 8 == 0, N %% 8 == 0, W %% 32 == 0, 16-byte aligned)");
    LU_REQUIRE(terms == 1 || (row_bf16 && row_variant && xb && yb && d->x_term_stride % 8 == 0 && d->dy_term_stride % 8 == 0),
               "lu_conv2d_wgrad: terms > 1 belongs to the bf16 kernel-row variant on bf16 operands (stride-1 3x3 / 5x5, C >= 64, W %% 32 == 0)");
    LU_REQUIRE(!pieces3 || (row_bf16 && row_variant && xb && yb && d->C % 128 == 0 && d->x_term_stride % 8 == 0 && d->dy_term_stride % 8 == 0 &&
                            3 * (int64_t)d->x_term_stride <= d->x_pix_stride && 3 * (int64_t)d->dy_term_stride <= d->dy_pix_stride),
               "lu_conv2d_wgrad: LU_WGRAD_F_PIECES3 (piece-aware 'bf16x3' weight gradient) needs bf16 split6 operands, stride-1 3x3 / 5x5, C %% 128 == 0, W %% 32 == 0");
    LU_REQUIRE(!d->dbias || row_variant || small3 || (row_s2 && row_bf16),
               "lu_conv2d_wgrad: dbias is produced by the kernel-row / all-taps variants only (stride-1 3x3 / 5x5, W %% 16 == 0, "
               "aligned operands, C >= 64 or a narrow 3x3 layer); use lu_colsum for this layer");
    // bf16 kernel-row variant: channel tile, and the bias rows per split that go with it (also needed by a phase-2 call)
    const int ct_bf16 = (d->k == 1 || (d->stride == 2 && d->k == 5)) ? 64 : (d->flags & LU_WGRAD_F_CT64) ? 64 : (d->flags & LU_WGRAD_F_CT128) ? 128
                                                                                          : (d->C % 128 == 0 || d->C > 256 ? 128 : 64);
    // all-taps form of the 3x3 layers (one block = all nine taps of a 64-channel x 128-column tile, 8 waves)
    // Measured (round 4, same-box A/B, config-2 shapes): 0.196 -> 0.215 of peak on the Params-net 3x3 layers, 0.295 -> 0.345 on the 3x3
    // ConvLSTM kernels (64-pixel stages where W % 64 == 0: +1 %).  (A 4-fat-wave instance -- 288 accumulators in AGPRs, one wave per
    // SIMD -- measured 0.155 / 0.218 and was removed in round 6.)  LU_WGRAD_F_NO_TAPS9 keeps the kernel-row form (the previous form; tests).
    const bool taps9 = row_bf16 && row_variant && d->k == 3 && d->stride == 1 && d->C >= 64 && !(d->flags & LU_WGRAD_F_NO_TAPS9) &&
                       !(d->flags & (LU_WGRAD_F_CT64 | LU_WGRAD_F_CT128));
    const bool taps9_ = taps9 && !pieces3;
    const int bias_rows_per_split = small3 ? 1 : pieces3 ? d->k * (d->C / 128) : taps9_ ? (d->C + 63) / 64
                                                : row_bf16 ? d->k * ((d->C + ct_bf16 - 1) / ct_bf16)
                                                : row_variant ? d->k * ((d->C + 63) / 64) : 1;      // (blocks sharing a dy tile)
    if (d->phase == 2) {
        // reduce only: the slabs were produced by an earlier phase-1 call with the same descriptor
    } else if (small3) {
        const int tiles = 9 * ((d->C + 31) / 32) * ((d->N + 31) / 32);
        dim3 grid(1, 1, (unsigned)splits);
        if (tiles <= 16) LU_LAUNCH((wgrad_small3_kernel<2>), grid, dim3(512), stream, a);
        else if (tiles <= 24) LU_LAUNCH((wgrad_small3_kernel<3>), grid, dim3(512), stream, a);
        else LU_LAUNCH((wgrad_small3_kernel<5>), grid, dim3(512), stream, a);
    } else if (pieces3) {
        a.c_tiles = d->C / 128;
        a.n_tiles = (d->N + 127) / 128;
        a.inner = a.n_tiles * d->k * a.c_tiles;
        a.splits = splits;
        dim3 grid((unsigned)(8 * a.inner * ((splits + 7) / 8)));      // XCD-aware numbering: see the kernel
        if (d->k == 5) LU_LAUNCH_DYN((wgrad_row_x3_kernel<5>), grid, dim3(512), wgrad_row_x3_lds(5), stream, a);
        else LU_LAUNCH_DYN((wgrad_row_x3_kernel<3>), grid, dim3(512), wgrad_row_x3_lds(3), stream, a);
    } else if (taps9_) {
        a.c_tiles = (d->C + 63) / 64;
        a.n_tiles = (d->N + 127) / 128;
        a.inner = a.n_tiles * a.c_tiles;
        a.splits = splits;
        dim3 grid((unsigned)(8 * a.inner * ((splits + 7) / 8)));
#define LU_WG9(NWV_)                                                                                                                   \
    do {                                                                                                                               \
        if (xb && yb) LU_LAUNCH_DYN((wgrad_row_bf16_kernel<3, 64, true, true, 1, 32, 3, NWV_>), grid, dim3(64 * NWV_), wgrad_row_bf16_lds(3, 64, 1, 32, true, 3, 64 * NWV_), stream, a);     \
        else if (yb) LU_LAUNCH_DYN((wgrad_row_bf16_kernel<3, 64, false, true, 1, 32, 3, NWV_>), grid, dim3(64 * NWV_), wgrad_row_bf16_lds(3, 64, 1, 32, false, 3, 64 * NWV_), stream, a);    \
        else if (xb) LU_LAUNCH_DYN((wgrad_row_bf16_kernel<3, 64, true, false, 1, 32, 3, NWV_>), grid, dim3(64 * NWV_), wgrad_row_bf16_lds(3, 64, 1, 32, true, 3, 64 * NWV_), stream, a);    \
        else LU_LAUNCH_DYN((wgrad_row_bf16_kernel<3, 64, false, false, 1, 32, 3, NWV_>), grid, dim3(64 * NWV_), wgrad_row_bf16_lds(3, 64, 1, 32, false, 3, 64 * NWV_), stream, a);           \
    } while (0)
        const bool p64 = d->Wout % 64 == 0 && xb && yb && !(d->flags & LU_WGRAD_F_PRB32);      // 64-pixel stages (bf16 operands)
        // LDS-DMA staging (bf16 operands): measured +3.5 % on the all-taps 3x3 form (three stage buffers, counted waits); -3 % (round 4) / -1.5 %
        // (round 6) on the 5x5 kernel-row form, whose instances were removed in round 6 (LU_WGRAD_F_NO_DMA keeps the register-staged all-taps form)
        const bool dma9 = xb && yb && !(d->flags & LU_WGRAD_F_NO_DMA);
        if (p64 && dma9)
            LU_LAUNCH_DYN((wgrad_row_bf16_kernel<3, 64, true, true, 1, 64, 3, 8, true>), grid, dim3(512), wgrad_row_bf16_lds(3, 64, 1, 64, true, 3, 512, true), stream, a);
        else if (dma9)
            LU_LAUNCH_DYN((wgrad_row_bf16_kernel<3, 64, true, true, 1, 32, 3, 8, true>), grid, dim3(512), wgrad_row_bf16_lds(3, 64, 1, 32, true, 3, 512, true), stream, a);
        else if (p64)
            LU_LAUNCH_DYN((wgrad_row_bf16_kernel<3, 64, true, true, 1, 64, 3, 8>), grid, dim3(512), wgrad_row_bf16_lds(3, 64, 1, 64, true, 3, 512), stream, a);
        else LU_WG9(8);
#undef LU_WG9
    } else if (row_bf16) {
        const int ct = ct_bf16;
        a.c_tiles = (d->C + ct - 1) / ct;
        a.n_tiles = (d->N + 127) / 128;
        a.inner = a.n_tiles * d->k * a.c_tiles;
        a.splits = splits;
        dim3 grid((unsigned)(8 * a.inner * ((splits + 7) / 8)));      // XCD-aware numbering: see the kernel
#define LU_WGB(K_, CT_)                                                                                              \
    do {                                                                                                             \
        if (xb && yb) LU_LAUNCH_DYN((wgrad_row_bf16_kernel<K_, CT_, true, true>), grid, dim3(512), wgrad_row_bf16_lds(K_, CT_, 1, 32, true), stream, a);            \
        else if (yb) LU_LAUNCH_DYN((wgrad_row_bf16_kernel<K_, CT_, false, true>), grid, dim3(512), wgrad_row_bf16_lds(K_, CT_, 1, 32, false), stream, a);            \
        else if (xb) LU_LAUNCH_DYN((wgrad_row_bf16_kernel<K_, CT_, true, false>), grid, dim3(512), wgrad_row_bf16_lds(K_, CT_, 1, 32, true), stream, a);            \
        else LU_LAUNCH_DYN((wgrad_row_bf16_kernel<K_, CT_, false, false>), grid, dim3(512), wgrad_row_bf16_lds(K_, CT_, 1, 32, false), stream, a);                   \
    } while (0)
#define LU_WGB64(K_)      /* 64-pixel stages: 128-channel tiles of the 5x5 ConvLSTM kernels (the bulk of the step) */    \
    do {                                                                                                             \
        if (xb && yb) LU_LAUNCH_DYN((wgrad_row_bf16_kernel<K_, 128, true, true, 1, 64>), grid, dim3(512), wgrad_row_bf16_lds(K_, 128, 1, 64, true), stream, a);     \
        else if (yb) LU_LAUNCH_DYN((wgrad_row_bf16_kernel<K_, 128, false, true, 1, 64>), grid, dim3(512), wgrad_row_bf16_lds(K_, 128, 1, 64, false), stream, a);     \
        else if (xb) LU_LAUNCH_DYN((wgrad_row_bf16_kernel<K_, 128, true, false, 1, 64>), grid, dim3(512), wgrad_row_bf16_lds(K_, 128, 1, 64, true), stream, a);     \
        else LU_LAUNCH_DYN((wgrad_row_bf16_kernel<K_, 128, false, false, 1, 64>), grid, dim3(512), wgrad_row_bf16_lds(K_, 128, 1, 64, false), stream, a);            \
    } while (0)
#define LU_WGB2(K_, CT_)                                                                                             \
    do {                                                                                                             \
        if (xb && yb) LU_LAUNCH_DYN((wgrad_row_bf16_kernel<K_, CT_, true, true, 2>), grid, dim3(512), wgrad_row_bf16_lds(K_, CT_, 2, 32, true), stream, a);          \
        else if (yb) LU_LAUNCH_DYN((wgrad_row_bf16_kernel<K_, CT_, false, true, 2>), grid, dim3(512), wgrad_row_bf16_lds(K_, CT_, 2, 32, false), stream, a);          \
        else if (xb) LU_LAUNCH_DYN((wgrad_row_bf16_kernel<K_, CT_, true, false, 2>), grid, dim3(512), wgrad_row_bf16_lds(K_, CT_, 2, 32, true), stream, a);          \
        else LU_LAUNCH_DYN((wgrad_row_bf16_kernel<K_, CT_, false, false, 2>), grid, dim3(512), wgrad_row_bf16_lds(K_, CT_, 2, 32, false), stream, a);                 \
    } while (0)
        if (d->stride == 2 && d->k == 5) LU_WGB2(5, 64);
        else if (d->stride == 2 && ct == 128) LU_WGB2(3, 128);
        else if (d->stride == 2) LU_WGB2(3, 64);
        else if (d->k == 1) LU_WGB(1, 64);
        else if (d->k == 5 && ct == 128 && d->Wout % 64 == 0 && !(d->flags & LU_WGRAD_F_PRB32)) LU_WGB64(5);
        else if (d->k == 5 && ct == 128) LU_WGB(5, 128);
        else if (d->k == 5) LU_WGB(5, 64);
        else if (ct == 128) LU_WGB(3, 128);
        else LU_WGB(3, 64);
#undef LU_WGB
#undef LU_WGB64
#undef LU_WGB2
    } else if (row_variant) {
        a.c_tiles = (d->C + 63) / 64;
        const int nxt = (d->N + 127) / 128;
        a.splits = splits;
        if (d->Wout % 16 != 0) {
            a.ragged = 1;
            a.wst = (d->Wout + 15) / 16;
            const int64_t rows = (int64_t)d->frames * d->Hout;
            a.rows_per = (int32_t)((rows + splits - 1) / splits);
        }
        a.xfold = (nxt == 1 || nxt == 2 || nxt == 4) ? 8 / nxt : 1;
        dim3 grid((unsigned)(nxt * a.xfold), (unsigned)(d->k * a.c_tiles), (unsigned)((splits + a.xfold - 1) / a.xfold));
        // 32-pixel stages wherever a run of 32 stays inside an image row (half the block-wide barriers per MFMA; with the lean stage
        // loads of round 4 faster than 16: 5x5 0.899 -> 0.914, 3x3 0.716 -> 0.732 of peak; rounds 1-3 had measured them slower)
        const bool kp32 = !a.ragged && d->Wout % 32 == 0;
        if (d->k == 5 && a.ragged) LU_LAUNCH((wgrad_row_kernel<5, true>), grid, dim3(512), stream, a);
        else if (d->k == 5 && kp32) LU_LAUNCH((wgrad_row_kernel<5, false, 32>), grid, dim3(512), stream, a);
        else if (d->k == 5) LU_LAUNCH((wgrad_row_kernel<5, false>), grid, dim3(512), stream, a);
        else if (a.ragged) LU_LAUNCH((wgrad_row_kernel<3, true>), grid, dim3(512), stream, a);
        else if (kp32) LU_LAUNCH((wgrad_row_kernel<3, false, 32>), grid, dim3(512), stream, a);
        else LU_LAUNCH((wgrad_row_kernel<3, false>), grid, dim3(512), stream, a);
    } else if (!xvec) {
        const int gy = (a.kk * d->C + 31) / 32;
        if (yvec) LU_WG(1, 1, 1, 4, true, true, gy);
        else LU_WG(1, 1, 1, 4, true, false, gy);
    } else if (d->C > 64 && d->N >= 256 && yvec && !(d->flags & LU_WGRAD_F_SMALL_TILE)) {
        // 128 x 256 tile: 64 MFMAs per pipeline stage per wave, half the x re-reads (ConvLSTM kernels: N = 4F >= 512)
        a.c_tiles = (d->C + 127) / 128;
        LU_WG(2, 4, 2, 2, false, true, a.kk * a.c_tiles);
    } else if (d->C > 64) {
        a.c_tiles = (d->C + 127) / 128;
        if (yvec) LU_WG(2, 2, 2, 2, false, true, a.kk * a.c_tiles);
        else LU_WG(2, 2, 2, 2, false, false, a.kk * a.c_tiles);
    } else if (d->C > 32) {
        a.c_tiles = 1;
        if (yvec) LU_WG(2, 1, 1, 4, false, true, a.kk);
        else LU_WG(2, 1, 1, 4, false, false, a.kk);
    } else {
        a.c_tiles = 1;
        if (yvec) LU_WG(1, 1, 1, 4, false, true, a.kk);
        else LU_WG(1, 1, 1, 4, false, false, a.kk);
    }
#undef LU_WG
    int rc = LU_CHECK_LAUNCH();
    if (rc || d->phase == 1) return rc;
    const int vec4 = (d->N % 4 == 0 && d->dw_tap_stride % 4 == 0 && d->dw_row_stride % 4 == 0 && aligned16(d->dw) &&
                      aligned16(a.ws)) ? 1 : 0;      // (slab = k*k*C*N is then a multiple of 4 as well)
    const bool tiny = a.slab <= 16384 && splits >= 64;      // thin / 1x1 layers: parallel over the slabs as well
    const int64_t items = tiny ? a.slab * 16 : vec4 ? a.slab / 4 : a.slab;
    const unsigned rgrid = (unsigned)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096);
    const unsigned bgrid = d->dbias ? (unsigned)((d->N + 15) / 16) : 0;
    LU_LAUNCH(wgrad_reduce_kernel, dim3(rgrid + bgrid), dim3(256), stream, (const float*)a.ws, a.slab, splits, d->dw, d->C,
              d->N, d->dw_tap_stride, d->dw_row_stride, d->beta, (const float*)(d->dbias ? a.bias_ws : nullptr), d->dbias,
              d->dbias_beta, splits * bias_rows_per_split, (int)rgrid, tiny ? 2 : vec4);
    return LU_CHECK_LAUNCH();
}
