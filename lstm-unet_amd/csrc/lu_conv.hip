// lu_conv.hip -- implicit-GEMM convolution on the gfx950 matrix pipe.
//
// Kernels in this file (host dispatch: lu_conv2d_fwd at the bottom):
//   conv_fwd_kernel         general fp32 path: any k <= 7, stride 1/2, input dilation 2, thin sources, narrow outputs
//   conv_halo_kernel        fp32, stride-1 3x3 / 5x5 with > 64 output columns: 8 x 32 patch, halo staged once per chunk
//   conv_halo_frag_kernel / conv_halo_frag3_kernel   the halo kernel with fragment-order bf16 weights streamed from L2 (precision 1):
//                           first loop generation (3x3 on fp32 sources / 8-row patches, narrow N = 32 / 64 blocks) / third (everything else)
//   conv_gather_bf16_kernel bf16 MFMA for everything outside the halo kernel's domain (stride 2, parity planes, 1x1, 7x7)
//   ksplit_reduce / flip_transpose / s2_dgrad_weights / pack_weights_*   helpers around them
// The description below is the common scheme, written for conv_fwd_kernel.
//
// GEMM view: out[m, n] = sum_k A[m, k] * W[k, n],  m = (frame, oy, ox), n = output channel,
// k = (source, kh, kw, c).  A is never materialised: every pipeline stage gathers a [256 px x 16 ch]
// slab for ONE filter tap straight from the channels-last activation (64 B contiguous per pixel),
// so consecutive taps of a channel chunk re-hit the same lines in L1/L2.
//
// Tile: 256 threads = 4 waves stacked along M; wave tile 64 x (32*NF) built from
// v_mfma_f32_32x32x2_f32 (exact f32, 64 cycles each -- MFMA-bound as long as it is fed).
//   A in LDS: [256][16+4] floats, read with ONE ds_read_b128 per 4 MFMAs (lanes 0-31 take
//             k = 8s..8s+3, lanes 32-63 take k = 8s+4..8s+7; the k pairing only has to agree with B).
//             Row pitch 20 floats => the 16-lane ds_read_b128 groups hit 16 distinct 16-B slots.
//   B in LDS: [16][BN] floats, ds_read_b32, 32 consecutive columns per half-wave.
// Two LDS stages; global loads of stage i+1 are issued before the MFMAs of stage i.
//
// Up to two (activation, weight) sources accumulate into the same tile: ConvLSTM x_t and h_{t-1}
// (Networks.py:62-63) or UpBlock's [upsampled, skip] concat (Networks.py:145-147) without ever
// writing the concatenated tensor.  Sources whose channel count is not a multiple of 4 (the
// 1-channel microscopy image) use a "thin" path where k enumerates flattened (tap, c).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <type_traits>
#include <utility>
#include "lu_device.h"

namespace {

constexpr int CK = 16;
constexpr int BM = 256;
constexpr int A_LD = CK + 4;

struct SrcInfo {
    const float* x;
    const float* w;
    int64_t frame_stride;
    int64_t w_tap_stride;
    int32_t pix_stride, C, w_row_stride;
    int32_t thin;     // 1: scalar gather over flattened (tap, c)
    int32_t nchunk;   // vector: ceil(C/16); thin: ceil(k*k*C/16)
    int32_t bf16;     // fragment kernel only: x is a bf16 tensor (strides in elements), staged without conversion
};

struct ConvArgs {
    SrcInfo src[2];    // vector sources (C % 4 == 0, 16-byte aligned): the pipelined hot loop
    SrcInfo tsrc[2];   // thin sources (e.g. the 1-channel image): short synchronous prologue
    int32_t n_src, n_thin, n_it;
    int64_t M;
    int32_t HWo, Wout, Hin, Win;
    int32_t k, kk, stride, dsh /* dil-1 */, pad_t, pad_l;
    int32_t N, n_tiles, m_tiles;
    int32_t tiles_x, tiles_pf;   // halo kernel: 8x32-pixel tiles per row / per frame
    int32_t xcd_by_n;            // halo kernel: give each XCD its own n-tiles (weights stay L2-resident per XCD)
    int32_t balanced;            // > 0: few-tile launch, 1-D grid: the m_tiles x n_tiles x ksplit work items cut into 8 equal
                                 // runs of `balanced` items, one run per XCD (lu_block_tile) ...
    int32_t balanced_by_m;       // ... in (m, split, n) order (activations outweigh weights) instead of (n, split, m)
    int32_t src1_center;   // fragment kernel: source 1 contributes its centre tap only (im2col image of a thin input)
    int32_t gates_bf16;    // LSTM epilogue of the fragment kernel: gates_out is bf16
    unsigned short* h16_out;   // ... optional bf16 copy of h
    int64_t h16_fs;
    int32_t h16_split;     // ... (LU_CONV_F_H16_SPLIT) h16_out is the split6 image of h: [pixel][6][F], blocks lo, mid, hi, mid, hi, hi
    int32_t lstm_vec4;     // conv_halo_kernel, LSTM epilogue: every state / gate tensor 16-byte aligned -> float4 loads / stores through LDS
    int32_t out_vec4s;     // general kernel: as out_vec4, strided output rows (parity planes) allowed
    int32_t out_vec4;      // tile kernels, bias epilogue: dense 16-byte aligned output rows, N % 4 == 0 -> float4 stores through LDS
    const float* zero16;   // device address of lu_zero16: as a kernel ARGUMENT it lives in SGPRs -- through the symbol every use in a
                           // loop costs s_getpc + s_load + s_waitcnt lgkmcnt(0), and that wait also drains the LDS reads in flight
    int32_t dbg;           // ablation bits, honoured only in -DLU_ABLATION tool builds: 1 skip prefetch, 2 skip LDS stores, 4 skip barrier
    int32_t ksplit;        // > 1: K (tap x channel-chunk) range split over blockIdx.y, partial tiles -> ws
    int32_t split_chunks;  // fp32 halo kernel, ksplit > 1: slices are WHOLE 16-channel chunks (k*k stages each) -- the compile-time tap
                           // sequence then serves K-split launches too (round 5); 0: ceil(n_it / ksplit) stages from any tap on
    float* ws;             // [ksplit][M][N] partial sums (LU_EPI_BIAS only)
    int32_t out_pix_stride;
    const float* bias;
    float* out;
    int64_t out_frame_stride;
    int64_t out_row_stride;   // elements between output rows (0: dense = Wout * out_pix_stride)
    const float* post_scale;  // LU_EPI_BIAS, optional: out = lrelu(post_scale[n] * (acc + bias[n]) + post_shift[n]) -- inference
    const float* post_shift;  // BatchNorm + LeakyReLU of a conv unit: applied by the slab reduce of a K-split launch, by a short
    float post_alpha;         // in-place pass otherwise (the tile kernels' epilogues stay as they are: measured, see DESIGN)
    // lstm epilogue
    int32_t F;
    const float* c_prev;
    float* c_out;
    float* h_out;
    float* gates_out;
    int64_t c_prev_fs, c_out_fs, h_fs, gates_fs;
};

#ifdef LU_ABLATION      // tools-only builds (-DLU_ABLATION=<bits>): loop ablations as COMPILE-TIME constants -- a run-time
#define LU_DBG(a, bit) ((LU_ABLATION) & (bit))      // switch between accumulator updates costs 5x by itself (measured)
#else
#define LU_DBG(a, bit) 0
#endif

// blockIdx -> (m-tile, n-tile, K split).  Workgroup b runs on XCD b % 8 (observed dispatch rule, speed only).
//  default:  all n-tiles of one m-tile get consecutive slots of the SAME XCD, so the activation slab they share is fetched
//            into one L2; the grid is padded to a multiple of 8 m-tiles; blockIdx.y is the K split.
//  balanced: launches with few tiles (streaming inference, the coarse levels: 5 m-tiles would leave 3 XCDs idle and put 96
//            blocks on the 64 slots of the others).  1-D grid; the work items in (n-tile, split, m-tile) order are cut into
//            8 equal runs, one per XCD: every XCD streams its own 1/8 of the weights once, the (small) input is shared.
//            balanced_by_m: (m-tile, split, n-tile) order instead -- layers whose activations outweigh their weights.
// false: padding block (uniform per block, taken before any barrier).
__device__ __forceinline__ bool lu_block_tile(const ConvArgs& a, int& mt, int& nt, int& ks) {
    const int bid = blockIdx.x;
    if (a.balanced) {
        const int w = (bid & 7) * a.balanced + (bid >> 3);
        if ((bid >> 3) >= a.balanced || w >= a.m_tiles * a.n_tiles * a.ksplit) return false;
        if (a.balanced_by_m) {
            const int r = w / a.n_tiles;
            nt = w - r * a.n_tiles;
            mt = r / a.ksplit;
            ks = r - mt * a.ksplit;
        } else {
            const int r = w / a.m_tiles;
            mt = w - r * a.m_tiles;
            nt = r / a.ksplit;
            ks = r - nt * a.ksplit;
        }
        return true;
    }
    const int slot = bid >> 3;
    nt = slot % a.n_tiles;
    mt = (slot / a.n_tiles) * 8 + (bid & 7);
    ks = blockIdx.y;
    return mt < a.m_tiles;
}

struct IterState {
    int s, chunk, tap, kh, kw;
};

__device__ __forceinline__ void iter_advance(IterState& st, const ConvArgs& a) {
    ++st.tap;
    ++st.kw;
    if (st.kw == a.k) {
        st.kw = 0;
        ++st.kh;
    }
    if (st.tap < a.kk) return;
    st.tap = st.kh = st.kw = 0;
    ++st.chunk;
    if (st.chunk == a.src[st.s].nchunk) {
        st.chunk = 0;
        ++st.s;
    }
}

__device__ __forceinline__ float hard_sigmoid(float z) { return fminf(fmaxf(0.2f * z + 0.5f, 0.f), 1.f); }

// compile-time loop: fn(std::integral_constant<int, 0>{}), ..., fn(std::integral_constant<int, N - 1>{})
template <class Fn, int... T>
__device__ __forceinline__ void lu_static_for_impl(Fn&& fn, std::integer_sequence<int, T...>) {
    (fn(std::integral_constant<int, T>{}), ...);
}
template <int N, class Fn>
__device__ __forceinline__ void lu_static_for(Fn&& fn) {
    lu_static_for_impl(fn, std::make_integer_sequence<int, N>{});
}


// Epilogue for ONE accumulator row (pixel `pix` of frame `f`, linear row index m) of a lane: v[nf] are the lane's
// values in the NF column fragments (column = 32*nf + (lane & 31)).
template <int NF, int EPI>
__device__ __forceinline__ void conv_epilogue_row(const ConvArgs& a, const float (&v)[NF], int f, int64_t pix, int64_t m,
                                                  int nt, int n0, int ks, int ccol, int oy_known = -1) {
    if (EPI == LU_EPI_LSTM) {
        const int ch = nt * 32 + ccol;  // F % 32 == 0 is enforced by the host
        const int F = a.F;
        float zi = v[0], zf = v[NF > 1 ? 1 : 0], zg = v[NF > 2 ? 2 : 0], zo = v[NF > 3 ? 3 : 0];
        if (a.bias) {
            zi += a.bias[ch];
            zf += a.bias[F + ch];
            zg += a.bias[2 * F + ch];
            zo += a.bias[3 * F + ch];
        }
        const float gi = hard_sigmoid(zi), gf = hard_sigmoid(zf), gg = lu_tanh_fast(zg), go = hard_sigmoid(zo);
        const float cp = a.c_prev[(int64_t)f * a.c_prev_fs + pix * F + ch];
        const float cn = fmaf(gf, cp, gi * gg);      // explicit: every copy of the cell update must contract the same way
        const float hn = go * lu_tanh_fast(cn);
        a.c_out[(int64_t)f * a.c_out_fs + pix * F + ch] = cn;
        a.h_out[(int64_t)f * a.h_fs + pix * F + ch] = hn;
        if (a.gates_out) {
            float* gp = a.gates_out + (int64_t)f * a.gates_fs + pix * (4 * F) + ch;
            gp[0] = gi;
            gp[F] = gf;
            gp[2 * F] = gg;
            gp[3 * F] = go;
        }
    } else if (a.ksplit > 1) {
        float* op = a.ws + ((int64_t)ks * a.M + m) * a.N;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int col = n0 + 32 * nf + ccol;
            if (col < a.N) op[col] = v[nf];
        }
    } else {
        float* op = a.out + (int64_t)f * a.out_frame_stride + pix * a.out_pix_stride;
        if (a.out_row_stride) {      // strided output rows (parity planes of a stride-2 input gradient)
            const int64_t oy = oy_known >= 0 ? oy_known : pix / a.Wout;
            op = a.out + (int64_t)f * a.out_frame_stride + oy * a.out_row_stride + (pix - oy * a.Wout) * a.out_pix_stride;
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int col = n0 + 32 * nf + ccol;
            if (col < a.N) op[col] = v[nf] + (a.bias ? a.bias[col] : 0.f);
        }
    }
}

// (frame, pixel, row) of a linear output index, advanced by small positive steps without divisions: the general kernels'
// epilogues walk the 16 accumulator rows of a fragment (steps of 1 or 5 pixels) after ONE decode per fragment -- with 64
// decodes per thread the 64-bit divisions cost as much as a tenth of a short (stride-2 / parity-plane) block.
struct RowCursor {
    int f, pix, oy, ox;
    __device__ __forceinline__ void set(const ConvArgs& a, int64_t m) {
        f = (int)((uint32_t)m / (uint32_t)a.HWo);          // M < 2^31 (checked by the host)
        pix = (int)m - f * a.HWo;
        oy = pix / a.Wout;
        ox = pix - oy * a.Wout;
    }
    __device__ __forceinline__ void advance(const ConvArgs& a, int d) {
        pix += d;
        ox += d;
        while (ox >= a.Wout) {
            ox -= a.Wout;
            ++oy;
        }
        while (pix >= a.HWo) {
            pix -= a.HWo;
            ++f;
            oy -= a.HWo / a.Wout;
        }
    }
};

// GEN = general addressing (input dilation 2: the dgrad of a stride-2 conv); !GEN = the common dil == 1 case,
// where a tap is a constant element offset from a per-row base computed once per source.
// MF = 32-row MFMA fragments per wave along M: MF = 2 -> 4 waves of 64 x BN (128 accumulator VGPRs at NF = 4,
// 2 waves/SIMD); MF = 1 -> 8 waves of 32 x BN (64 accumulator VGPRs, 512-thread blocks, 4 waves/SIMD): same block tile
// and LDS image, twice the wave-level parallelism to cover the non-MFMA phases of each wave.
// DMA = stage tiles with LDS-DMA (global_load_lds): the A image is then unpadded [256][16] with the 16-byte slot
// index XOR-swizzled by (row >> 2) & 3 (the swizzle is applied on the SOURCE channel group; LDS stays lane-linear).
template <int NF, bool BVEC, int EPI, bool GEN, int MF, bool DMA>
__global__ __launch_bounds__(64 * (8 / MF), 2 * (2 / MF)) void conv_fwd_kernel(ConvArgs a) {
    const float* const lu_z16 = a.zero16 ? a.zero16 : lu_zero16;
    constexpr int NT = 64 * (8 / MF);    // threads per block
    constexpr int RA = 1024 / NT;        // A rows (16-byte column groups) gathered per thread per stage
    constexpr int BN = 32 * NF;
    constexpr int QPR = BN / 4;          // float4 per B row
    constexpr int RPP = NT / QPR;        // B rows per pass
    constexpr int NPB = (CK + RPP - 1) / RPP;
    constexpr int ALD = DMA ? CK : A_LD;
    __shared__ __attribute__((aligned(16))) float As[2][BM * ALD];
    __shared__ __attribute__((aligned(16))) float Bs[2][CK * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int mt, nt, ks;
    if (!lu_block_tile(a, mt, nt, ks)) return;     // XCD-aware tile order
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;

    // ---- A gather bookkeeping: RA pixel rows per thread, one 16-byte column group ----
    const int q = tid & 3;
    int fr[RA], vy0[RA], vx0[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int64_t m = m0 + (tid >> 2) + (NT / 4) * i;
        if (m < a.M) {
            int f = (int)(m / a.HWo);
            int r = (int)(m - (int64_t)f * a.HWo);
            int oy = r / a.Wout, ox = r - oy * a.Wout;
            fr[i] = f;
            vy0[i] = oy * a.stride - a.pad_t;
            vx0[i] = ox * a.stride - a.pad_l;
        } else {
            fr[i] = 0;
            vy0[i] = -(1 << 28);
            vx0[i] = -(1 << 28);
        }
    }
    // ---- B bookkeeping ----
    const int bq = tid % QPR, brow0 = tid / QPR;
    int bcol;  // first global column of this thread's float4
    if (EPI == LU_EPI_LSTM) {
        // tile columns = 4 gates x 32 channels: gate-major so a lane holds i,f,g,o of one channel
        bcol = (bq >> 3) * a.F + nt * 32 + 4 * (bq & 7);
    } else {
        bcol = n0 + 4 * bq;
    }

    float4 ra[RA];
    float4 rb0, rb1;   // (scalars, not an array: hipcc promoted `float4 rb[NPB]` to an LDS-backed alloca)
    rb0 = rb1 = make_float4(0.f, 0.f, 0.f, 0.f);

    // NOTE: every global load below is UNCONDITIONAL: masked lanes read lu_zero16 (16 B of device zeros).
    // Guarding the load (`if (ok) v = *p`) makes hipcc branch around each load with `s_waitcnt vmcnt(0)`
    // in between -- six serialized memory round trips per stage; zeroing AFTER the load drags the wait
    // up to the issue point.  Selecting the POINTER keeps all loads of a stage in flight across the MFMAs.
    const float* const zp = lu_z16;
    auto load_w = [&](const SrcInfo& si, bool thin, int tap_v, int chunk, int p) -> float4 {
        const int row = brow0 + RPP * p;
        int tap, c;
        bool rok;
        if (!thin) {
            tap = tap_v;
            c = chunk * CK + row;
            rok = row < CK && c < si.C;
        } else {
            int j = chunk * CK + row;
            rok = row < CK && j < a.kk * si.C;
            tap = j / si.C;
            c = j - tap * si.C;
        }
        const float* wp = si.w + (int64_t)tap * si.w_tap_stride + (int64_t)c * si.w_row_stride + bcol;
        if (BVEC) {
            const bool ok = rok && ((EPI == LU_EPI_LSTM) || bcol < a.N);
            return *reinterpret_cast<const float4*>(ok ? wp : zp);
        } else {
            const float t0 = *((rok && bcol + 0 < a.N) ? wp + 0 : zp), t1 = *((rok && bcol + 1 < a.N) ? wp + 1 : zp),
                        t2 = *((rok && bcol + 2 < a.N) ? wp + 2 : zp), t3 = *((rok && bcol + 3 < a.N) ? wp + 3 : zp);
            return make_float4(t0, t1, t2, t3);
        }
    };
    auto load_weights = [&](const SrcInfo& si, bool thin, int tap_v, int chunk) {
        rb0 = load_w(si, thin, tap_v, chunk, 0);
        if (NPB > 1) rb1 = load_w(si, thin, tap_v, chunk, 1);
    };
    int64_t rowoff[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) rowoff[i] = 0;
    int cached_s = -1;
    auto load_stage = [&](const IterState& st) {
        const SrcInfo& si = a.src[st.s];
        const int c = st.chunk * CK + 4 * q;
        const bool cok = c < si.C;
        const float* pv[RA];
        if (GEN) {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int vy = vy0[i] + st.kh, vx = vx0[i] + st.kw;
                const int iy = vy >> a.dsh, ix = vx >> a.dsh;
                const bool ok = cok && vy >= 0 && vx >= 0 && iy < a.Hin && ix < a.Win && ((vy | vx) & a.dsh) == 0;
                const float* p = si.x + (int64_t)fr[i] * si.frame_stride + ((int64_t)iy * a.Win + ix) * si.pix_stride + c;
                pv[i] = ok ? p : zp;
            }
        } else {
            if (st.s != cached_s) {      // once per source (uniform)
                cached_s = st.s;
#pragma unroll
                for (int i = 0; i < RA; ++i)
                    rowoff[i] = (int64_t)fr[i] * si.frame_stride + ((int64_t)vy0[i] * a.Win + vx0[i]) * si.pix_stride;
            }
            const float* tapbase = si.x + ((int64_t)st.kh * a.Win + st.kw) * si.pix_stride + c;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const bool ok = cok && (unsigned)(vy0[i] + st.kh) < (unsigned)a.Hin &&
                                (unsigned)(vx0[i] + st.kw) < (unsigned)a.Win;
                pv[i] = ok ? tapbase + rowoff[i] : zp;
            }
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) ra[i] = *reinterpret_cast<const float4*>(pv[i]);
        load_weights(si, false, st.tap, st.chunk);
    };
    // LDS-DMA version of load_stage + store_stage: lane-linear LDS image, swizzle folded into the source address
    auto dma_stage = [&](const IterState& st, int buf) {
        const SrcInfo& si = a.src[st.s];
        if (!GEN && st.s != cached_s) {
            cached_s = st.s;
#pragma unroll
            for (int i = 0; i < RA; ++i)
                rowoff[i] = (int64_t)fr[i] * si.frame_stride + ((int64_t)vy0[i] * a.Win + vx0[i]) * si.pix_stride;
        }
        const float* tapbase = si.x + ((int64_t)st.kh * a.Win + st.kw) * si.pix_stride;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int row = (tid >> 2) + (NT / 4) * i;
            const int c = st.chunk * CK + 4 * (q ^ ((row >> 2) & 3));     // this LDS slot holds channel group q ^ f(row)
            const float* p;
            bool ok;
            if (GEN) {
                const int vy = vy0[i] + st.kh, vx = vx0[i] + st.kw;
                const int iy = vy >> a.dsh, ix = vx >> a.dsh;
                ok = c < si.C && vy >= 0 && vx >= 0 && iy < a.Hin && ix < a.Win && ((vy | vx) & a.dsh) == 0;
                p = si.x + (int64_t)fr[i] * si.frame_stride + ((int64_t)iy * a.Win + ix) * si.pix_stride + c;
            } else {
                ok = c < si.C && (unsigned)(vy0[i] + st.kh) < (unsigned)a.Hin && (unsigned)(vx0[i] + st.kw) < (unsigned)a.Win;
                p = tapbase + rowoff[i] + c;
            }
            lu_glds16(ok ? p : zp, &As[buf][(wave * 16 + (NT / 4) * i) * ALD]);
        }
#pragma unroll
        for (int pp = 0; pp < NPB; ++pp) {
            const int row = brow0 + RPP * pp;
            const int c = st.chunk * CK + row;
            const bool ok = row < CK && c < si.C && ((EPI == LU_EPI_LSTM) || bcol < a.N);
            const float* wp = si.w + (int64_t)st.tap * si.w_tap_stride + (int64_t)c * si.w_row_stride + bcol;
            // rows of 64 lanes: lane -> (row = tid / QPR, 16-byte column tid % QPR) is linear in tid
            if (RPP * (pp + 1) <= CK || row < CK) lu_glds16(ok ? wp : zp, &Bs[buf][(wave * 64 + NT * pp) * 4]);
        }
    };

    auto load_thin = [&](const SrcInfo& si, int chunk) {
        const int kkC = a.kk * si.C;
        int dy[4], dx[4], cc[4];
        bool jok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int j = chunk * CK + 4 * q + e;
            jok[e] = j < kkC;
            int tap = j / si.C;
            cc[e] = j - tap * si.C;
            dy[e] = tap / a.k;
            dx[e] = tap - dy[e] * a.k;
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int vy = vy0[i] + dy[e], vx = vx0[i] + dx[e];
                const int iy = vy >> a.dsh, ix = vx >> a.dsh;
                const bool ok = jok[e] && vy >= 0 && vx >= 0 && iy < a.Hin && ix < a.Win && ((vy | vx) & a.dsh) == 0;
                const float* p = si.x + (int64_t)fr[i] * si.frame_stride + ((int64_t)iy * a.Win + ix) * si.pix_stride + cc[e];
                v[e] = *(ok ? p : zp);
            }
            ra[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
        load_weights(si, true, 0, chunk);
    };

    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i)
        {
            const int row = (tid >> 2) + (NT / 4) * i;
            const int slot = DMA ? (q ^ ((row >> 2) & 3)) : q;
            *reinterpret_cast<float4*>(&As[buf][row * ALD + 4 * slot]) = ra[i];
        }
        if (RPP <= CK || brow0 < CK) *reinterpret_cast<float4*>(&Bs[buf][brow0 * BN + 4 * bq]) = rb0;
        if (NPB > 1) *reinterpret_cast<float4*>(&Bs[buf][(brow0 + RPP) * BN + 4 * bq]) = rb1;
    };

    f32x16 acc[MF][NF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;

    const int arow = wave * (32 * MF) + (lane & 31);
    const int khalf = 4 * (lane >> 5);
    // one stage = 8 groups of MF*NF MFMAs; group g uses k = 8*(g/4) + {g%4, 4 + g%4}.  The LDS reads run ahead of the MFMAs
    // (round 4, as in conv_halo_kernel): both A fragments of the stage up front, the B values of group g + 1 requested before
    // group g is issued (two register sets) -- counted waits instead of read -> s_waitcnt lgkmcnt(0) -> NF MFMAs.
    float4 afr[MF][2];
    float bvr[2][NF];
    auto rd_af = [&](int buf, int s) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int row = arow + 32 * mf;
            const int grp = 2 * s + (khalf >> 2);     // 16-byte channel group wanted by this half-wave
            const int slot = DMA ? (grp ^ ((row >> 2) & 3)) : grp;
            afr[mf][s] = *reinterpret_cast<const float4*>(&As[buf][row * ALD + 4 * slot]);
        }
    };
    auto rd_bv = [&](int buf, int g) {
        const int s = g >> 2, j = g & 3;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) bvr[g & 1][nf] = Bs[buf][(8 * s + khalf + j) * BN + 32 * nf + (lane & 31)];
    };
    auto stage_begin = [&](int buf) {
        rd_af(buf, 0);
        rd_bv(buf, 0);
        rd_af(buf, 1);
    };
    auto mma_group = [&](int buf, int g) {
        const int s = g >> 2, j = g & 3;
        if (g + 1 < 8) rd_bv(buf, g + 1);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const float4& fa = afr[mf][s];
            const float av = j == 0 ? fa.x : j == 1 ? fa.y : j == 2 ? fa.z : fa.w;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = lu_mfma(av, bvr[g & 1][nf], acc[mf][nf]);
        }
    };
    auto mma_stage = [&](int buf) {
        stage_begin(buf);
#pragma unroll
        for (int g = 0; g < 8; ++g) mma_group(buf, g);
    };

    // ---- thin sources: a few synchronous stages (K = k*k*C is tiny), kept out of the hot loop ----
    for (int ts = 0; ts < (ks == 0 ? a.n_thin : 0); ++ts) {
        const SrcInfo& si = a.tsrc[ts];
        for (int ch = 0; ch < si.nchunk; ++ch) {
            load_thin(si, ch);
            store_stage(0);
            __syncthreads();
            mma_stage(0);
            __syncthreads();
        }
    }

    // ---- vector sources: two-stage software pipeline, one (tap, 16-channel chunk) per stage ----
    int it0 = 0, it1 = a.n_it;
    if (a.ksplit > 1) {
        const int per = (a.n_it + a.ksplit - 1) / a.ksplit;
        it0 = ks * per;
        it1 = it0 + per < a.n_it ? it0 + per : a.n_it;
    }
    if (it1 > it0) {
        IterState st{0, 0, 0, 0, 0};
        {   // decode the first stage of this K-range
            int r = it0;
            while (r >= a.src[st.s].nchunk * a.kk) {
                r -= a.src[st.s].nchunk * a.kk;
                ++st.s;
            }
            st.chunk = r / a.kk;
            st.tap = r - st.chunk * a.kk;
            st.kh = st.tap / a.k;
            st.kw = st.tap - st.kh * a.k;
        }
        if (DMA) {
            dma_stage(st, 0);
        } else {
            load_stage(st);
            store_stage(0);
        }
        __syncthreads();
        for (int it = it0; it < it1; ++it) {
            const int buf = (it - it0) & 1;
            // No `if (more)` around the prefetch / LDS store: hipcc merges the two equally-guarded blocks and
            // drags the weight-tile ds_writes (and their vmcnt wait) in front of the MFMAs.  The last
            // iteration simply re-fetches its own stage into the idle buffer (1/n_it extra traffic).
            // The wave issues in order, but a 64-cycle MFMA leaves ~15 issue slots free behind it: the address
            // arithmetic + global loads of the next stage go after the FIRST MFMA group and its LDS stores before
            // the LAST one, so their issue time hides under MFMA execution instead of idling the matrix pipe.
            stage_begin(buf);
            mma_group(buf, 0);
            LU_SCHED_FENCE();
            if (it + 1 < it1) iter_advance(st, a);
            if (DMA) {
                // straight into the other LDS buffer (last read before the previous barrier); __syncthreads() below
                // waits vmcnt(0) for the pending LDS writes
                if (!LU_DBG(a, 1)) dma_stage(st, buf ^ 1);
                LU_SCHED_FENCE();
#pragma unroll
                for (int g = 1; g < 8; ++g) {
                    mma_group(buf, g);
                    LU_SCHED_FENCE();
                }
            } else {
                if (!LU_DBG(a, 1)) load_stage(st);
                LU_SCHED_FENCE();
#pragma unroll
                for (int g = 1; g < 7; ++g) {
                    mma_group(buf, g);
                    LU_SCHED_FENCE();
                }
                if (!LU_DBG(a, 2)) store_stage(buf ^ 1);        // buf^1 was last read before the previous barrier
                LU_SCHED_FENCE();
                mma_group(buf, 7);
            }
            if (!LU_DBG(a, 4)) __syncthreads();
        }
    }

    // ---- epilogue ----
    if (EPI == LU_EPI_BIAS && a.out_vec4s && a.ksplit <= 1) {
        // 16-byte stores (see conv_halo_kernel): each wave turns its fragments round, 16 rows at a time, in a private slice of
        // the A tile's LDS (dead: the loop ended on a barrier); a lane then owns (row, four consecutive columns).
        float* const Exw = &As[0][0] + wave * (16 * 36);
        const int cq = lane & 7;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                int64_t obase[2];
                bool rok[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {      // the two rows this lane stores in every pass of this (mf, half)
                    const int64_t m = m0 + wave * (32 * MF) + mf * 32 + 16 * half + (lane >> 3) + 8 * q;
                    rok[q] = m < a.M;
                    RowCursor rc;
                    rc.set(a, rok[q] ? m : 0);
                    obase[q] = (int64_t)rc.f * a.out_frame_stride +
                               (a.out_row_stride ? (int64_t)rc.oy * a.out_row_stride + (int64_t)rc.ox * a.out_pix_stride
                                                 : (int64_t)rc.pix * a.out_pix_stride);
                }
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr)      // row 16 half + (rr & 3) + 8 (rr >> 2) + 4 (lane >> 5) of the fragment
                        Exw[((rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[mf][nf][8 * half + rr];
                    LU_WAVE_SYNC();
                    const int col = n0 + 32 * nf + 4 * cq;
                    const float4 bq = (a.bias && col < a.N) ? *reinterpret_cast<const float4*>(a.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float4 v = *reinterpret_cast<const float4*>(&Exw[((lane >> 3) + 8 * q) * 36 + 4 * cq]);
                        v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
                        if (rok[q] && col < a.N) *reinterpret_cast<float4*>(a.out + obase[q] + col) = v;
                    }
                    LU_WAVE_SYNC();
                }
            }
        return;
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int64_t mb = m0 + wave * (32 * MF) + mf * 32 + 4 * (lane >> 5);      // accumulator row r = 0 of this lane
        RowCursor rc;
        rc.set(a, mb < a.M ? mb : 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r) rc.advance(a, (r & 3) ? 1 : 5);          // rows 0,1,2,3, 8,9,10,11, 16,... (+ 4 for the upper half-wave)
            const int64_t m = mb + (r & 3) + 8 * (r >> 2);
            if (m >= a.M) continue;
            float v[NF];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) v[nf] = acc[mf][nf][r];
            conv_epilogue_row<NF, EPI>(a, v, rc.f, rc.pix, m, nt, n0, ks, lane & 31, rc.oy);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Halo-reuse variant for stride-1 K x K convolutions (K = 3, 5) with wide outputs (NF = 4): the block owns an
// 8 x 32-pixel patch of ONE frame (wave w = patch row w, lane & 31 = x) and stages the (8+K-1) x (32+K-1) input halo of a
// 16-channel chunk in LDS ONCE; the K*K taps then read shifted rows of that halo, so per tap only the 16 x 128
// weight tile is fetched.  Activation traffic per block drops ~15x (k = 5) versus the per-tap gather above, which
// an ablation showed costs ~10 % of the MFMA rate.  Same MFMA/LDS fragment scheme, same epilogues.
// ---------------------------------------------------------------------------------------------------------
template <int K, int EPI, bool ST = false>      // ST: the taps of a chunk as a compile-time sequence (launches without a K split)
__global__ __launch_bounds__(512, 4) void conv_halo_kernel(ConvArgs a) {
    const float* const lu_z16 = a.zero16 ? a.zero16 : lu_zero16;
    constexpr int NF = 4, BN = 128, NT = 512, TH = 8, TW = 32;
    constexpr int HWD = TW + K - 1, HHT = TH + K - 1, HP = HHT * HWD;    // halo width / height / pixels
    constexpr int HPASS = (HP * 4 + NT - 1) / NT;                        // 16-byte loads per thread per halo
    constexpr int PAD = (K - 1) / 2;
    static_assert(HP * A_LD >= BM * A_LD, "halo buffer doubles as the [256][20] tile of the thin-source prologue");
    __shared__ __attribute__((aligned(16))) float Ah[HP * A_LD];
    __shared__ __attribute__((aligned(16))) float Bs[3][CK * BN];      // weight tiles [16 k][128 columns]: three stages (LDS-DMA, see the loop)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = blockIdx.x;
    const int slot = bid >> 3;
    int tile, nt, ks;
    const bool live = lu_block_tile(a, tile, nt, ks);     // XCD-aware order
    if (a.xcd_by_n) {
        // n_tiles % 8 == 0: XCD x (= bid % 8) owns n-tiles [x*g, (x+1)*g): the 64 blocks resident on an XCD stream the
        // SAME 16x128 weight tiles (a few MB per XCD, L2-resident) instead of every n-tile's; the patch halo, fetched
        // once per 25 taps, is what crosses XCDs.
        const int g = a.n_tiles >> 3;
        nt = (bid & 7) * g + (slot % g);
        tile = slot / g;
        if (tile >= a.m_tiles) return;
    } else if (!live) {
        return;
    }
    const int f = tile / a.tiles_pf;
    const int t2 = tile - f * a.tiles_pf;
    const int y0 = (t2 / a.tiles_x) * TH, x0 = (t2 % a.tiles_x) * TW;
    const int n0 = nt * BN;
    const float* const zp = lu_z16;
    // Per-source fields live in registers and are picked with selects: indexing a.src[st.s] inside the tap loop costs a
    // scalar kernarg load + s_waitcnt lgkmcnt(0) per use, and that wait also drains the LDS fragment reads in flight.
    const float* const x_s0 = a.src[0].x + (int64_t)f * a.src[0].frame_stride;
    const float* const x_s1 = a.src[1].x + (int64_t)f * a.src[1].frame_stride;
    const float* const w_s0 = a.src[0].w;
    const float* const w_s1 = a.src[1].w;
    const int64_t wts_s0 = a.src[0].w_tap_stride, wts_s1 = a.src[1].w_tap_stride;
    const int wrs_s0 = a.src[0].w_row_stride, wrs_s1 = a.src[1].w_row_stride;
    const int ps_s0 = a.src[0].pix_stride, ps_s1 = a.src[1].pix_stride;
    const int C_s0 = a.src[0].C, C_s1 = a.src[1].C;
    const int nch_s0 = a.src[0].nchunk, nch_s1 = a.src[1].nchunk;
    // ---- halo gather: piece i of a thread is 4 channels of halo pixel (tid + 512 i) / 4.  The staging registers are a NATIVE
    // vector type and the pixel offsets are recomputed per chunk (a handful of integer ops every K*K stages): as `float4 rh[]` +
    // `int hoff[]` captured by the lambdas the arrays lived in SCRATCH memory (80 bytes per thread, rounds 1-2) -- every chunk the
    // kernel waited for its halo loads right where it issued them (s_waitcnt vmcnt + scratch_store) instead of under the MFMAs
    // of the stage, and the scratch lines showed up as 3-6x the algorithmic bytes in WRITE_SIZE / FETCH_SIZE.
    const int q = tid & 3;
    // ---- B bookkeeping: 16 rows x 32 float4, one per thread ----
    const int bq = tid & 31, brow = tid >> 5;
    const int bcol = (EPI == LU_EPI_LSTM) ? (bq >> 3) * a.F + nt * 32 + 4 * (bq & 7) : n0 + 4 * bq;

    lu_u4 rh[HPASS];
    float4 rb = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_halo = [&](int src, int chunk) {
        const int c = chunk * CK + 4 * q;
        const float* base = (src ? x_s1 : x_s0) + c;
        const int ps = src ? ps_s1 : ps_s0;
        const bool cok = c < (src ? C_s1 : C_s0);
#pragma unroll
        for (int i = 0; i < HPASS; ++i) {
            const int hp = (tid + NT * i) >> 2;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int iy = y0 + hy - PAD, ix = x0 + hx - PAD;
            const bool ok = cok && hp < HP && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            const float* p = base + (int64_t)(iy * a.Win + ix) * ps;
            rh[i] = *reinterpret_cast<const lu_u4*>(ok ? p : zp);
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int i = 0; i < HPASS; ++i) {
            const int hp = (tid + NT * i) >> 2;
            if (hp < HP) *reinterpret_cast<lu_u4*>(&Ah[hp * A_LD + 4 * q]) = rh[i];
        }
    };
    auto load_b = [&](const SrcInfo& si, bool thin, int tap_v, int chunk) {
        int tap, c;
        bool rok;
        if (!thin) {
            tap = tap_v;
            c = chunk * CK + brow;
            rok = c < si.C;
        } else {
            const int j = chunk * CK + brow;
            rok = j < a.kk * si.C;
            tap = j / si.C;
            c = j - tap * si.C;
        }
        const float* wp = si.w + (int64_t)tap * si.w_tap_stride + (int64_t)c * si.w_row_stride + bcol;
        const bool ok = rok && ((EPI == LU_EPI_LSTM) || bcol < a.N);
        rb = *reinterpret_cast<const float4*>(ok ? wp : zp);
    };
    // Weight tile of a stage by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write): thread (brow, bq) owns the
    // 16 bytes of row brow, columns 4 bq .. 4 bq + 3 = float tid * 4 of the [16][128] image, i.e. a wave's 64 lanes fill 1 KB.
    // The stream of weight tiles is walked by a RUNNING per-thread pointer (round 4): one 64-bit add per stage, a handful of scalar
    // instructions at a chunk change -- the (tap, chunk, source) -> address arithmetic redone every stage (64-bit multiplies,
    // selects between the two sources' fields: ~60 scalar + ~20 vector instructions per wave) was what "the loads" cost the loop:
    // 5x5 input gradients 0.853 -> 0.877 of peak with nothing else changed.
    int d_s = 0, d_chunk = 0, d_tap = 0;      // (source, chunk, tap) of the NEXT tile to request
    int64_t d_wts = 0;                          // ... elements between two taps of its source
    const float* wq = nullptr;                 // ... its 16 bytes for this thread
    bool wok = false;                          // ... inside the kernel (row c < C, column < N)
    const bool colok = (EPI == LU_EPI_LSTM) || bcol < a.N;
    auto dma_seek = [&](int src, int chunk, int tap) {
        d_s = src;
        d_chunk = chunk;
        d_tap = tap;
        const int c = chunk * CK + brow;
        wq = (src ? w_s1 : w_s0) + (int64_t)tap * (src ? wts_s1 : wts_s0) + (int64_t)c * (src ? wrs_s1 : wrs_s0) + bcol;
        wok = c < (src ? C_s1 : C_s0) && colok;
        d_wts = src ? wts_s1 : wts_s0;
    };
    auto dma_next = [&](float* Bd) {           // request the tile, step to the one after it
        lu_glds16(wok ? wq : zp, Bd + wave * 256);
        if (++d_tap < K * K) {
            wq += d_s ? wts_s1 : wts_s0;
        } else if (d_chunk + 1 < (d_s ? nch_s1 : nch_s0)) {
            dma_seek(d_s, d_chunk + 1, 0);
        } else {
            dma_seek(d_s + 1, 0, 0);           // (past the last source: never requested -- the callers count stages)
        }
    };
    // mode 0: dma_next (run-time tap counter); 1: the next tile is the next tap of the same chunk; 2: ... tap 0 of the next chunk
    auto dma_issue = [&](float* Bd, int mode) {
        if (mode == 0) {
            dma_next(Bd);
            return;
        }
        lu_glds16(wok ? wq : zp, Bd + wave * 256);
        if (mode == 1) wq += d_wts;
        else if (d_chunk + 1 < (d_s ? nch_s1 : nch_s0)) dma_seek(d_s, d_chunk + 1, 0);
        else dma_seek(d_s + 1, 0, 0);
    };
    auto store_b = [&](int buf) { *reinterpret_cast<float4*>(&Bs[buf][brow * BN + 4 * bq]) = rb; };      // (thin-source prologue)

    f32x16 acc[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nf][r] = 0.f;

    const int khalf = 4 * (lane >> 5);
    // One stage = 8 groups g = (s, j) of four MFMAs: k = 8 s + {j, 4 + j}; the A fragment row is the halo pixel (wave + kh,
    // x + kw).  The LDS reads run AHEAD of the MFMAs (round 4; rounds 1-3: ds_read2_b32 -> s_waitcnt lgkmcnt(0) -> two MFMAs):
    // the B values of group g + 1 are requested before group g is issued (two register sets), the s = 1 A fragment during group
    // 0, and -- the stage barrier sits in the MIDDLE of a stage, see the loop -- the next stage's first operands during groups 5
    // and 7.  Same values, same MFMA order as before: bit-identical results.
    const int boff = khalf * BN + (lane & 31);
    struct B4 {
        float x, y, z, w;
    };
    // Column fragments.  Gate epilogue: fragment nf = gate nf, lane l = channel l (tile column 32 nf + l: four dwords 128 bytes
    // apart).  Bias epilogue (round 4): INTERLEAVED fragments -- fragment nf covers tile columns 4 l + nf, so a lane's four B values
    // are 16 contiguous bytes of the [16][128] tile (ONE ds_read_b128 per k-pair instead of two ds_read2_b32) and its four
    // accumulators hold four CONSECUTIVE output columns of every pixel it owns: the epilogue stores 16 bytes per pixel straight
    // from the registers, coalesced across the lanes (512 bytes per pixel), without the exchange through LDS.  Which columns a
    // fragment covers does not touch any element's accumulation order: bit-identical results.
    auto rd_b = [&](const float* Br, int g) {
        if (EPI == LU_EPI_BIAS) {
            const float4 v = *reinterpret_cast<const float4*>(Br + khalf * BN + 4 * (lane & 31) + (8 * (g >> 2) + (g & 3)) * BN);
            return B4{v.x, v.y, v.z, v.w};
        }
        const float* p = Br + boff + (8 * (g >> 2) + (g & 3)) * BN;
        return B4{p[0], p[32], p[64], p[96]};
    };
    auto rd_a = [&](int arow, int s) { return *reinterpret_cast<const float4*>(&Ah[arow * A_LD + 8 * s + khalf]); };
    auto mma4 = [&](float av, const B4& b) {
        acc[0] = lu_mfma(av, b.x, acc[0]);
        acc[1] = lu_mfma(av, b.y, acc[1]);
        acc[2] = lu_mfma(av, b.z, acc[2]);
        acc[3] = lu_mfma(av, b.w, acc[3]);
    };
    float4 fa0, fa1;
    B4 fb[2];
    auto fa_of = [&](int g) {
        const float4& fa = (g >> 2) ? fa1 : fa0;
        const int j = g & 3;
        return j == 0 ? fa.x : j == 1 ? fa.y : j == 2 ? fa.z : fa.w;
    };
    auto simple_stage = [&](const float* Br, int arow) {      // (thin-source prologue: a few synchronous stages)
        fa0 = rd_a(arow, 0);
        fa1 = rd_a(arow, 1);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            fb[0] = rd_b(Br, g);
            mma4(fa_of(g), fb[0]);
        }
    };

    // ---- thin sources (e.g. the 1-channel image): per-tap gather into the buffer used as a [256][20] tile ----
    if (ks == 0) {
        const int oyr[2] = {y0 + ((tid >> 2) >> 5), y0 + (((tid >> 2) + 128) >> 5)};
        const int oxr = x0 + ((tid >> 2) & 31);
        for (int ts = 0; ts < a.n_thin; ++ts) {
            const SrcInfo& si = a.tsrc[ts];
            const int kkC = a.kk * si.C;
            for (int ch = 0; ch < si.nchunk; ++ch) {
                float4 rt[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int jj = ch * CK + 4 * q + e;
                        const int tap = jj / si.C, cc = jj - tap * si.C;
                        const int dy = tap / K, dx = tap - dy * K;
                        const int iy = oyr[i] + dy - PAD, ix = oxr + dx - PAD;
                        const bool ok = jj < kkC && oyr[i] < a.Hin && oxr < a.Win && iy >= 0 && iy < a.Hin && ix >= 0 &&
                                        ix < a.Win;
                        const float* p = si.x + (int64_t)f * si.frame_stride + ((int64_t)iy * a.Win + ix) * si.pix_stride + cc;
                        v[e] = *(ok ? p : zp);
                    }
                    rt[i] = make_float4(v[0], v[1], v[2], v[3]);
                }
                load_b(si, true, 0, ch);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    *reinterpret_cast<float4*>(&Ah[((tid >> 2) + 128 * i) * A_LD + 4 * q]) = rt[i];
                store_b(0);
                __syncthreads();
                simple_stage(&Bs[0][0], wave * 32 + (lane & 31));
                __syncthreads();
            }
        }
    }

    // ---- vector sources ----
    int it0 = 0, it1 = a.n_it;
    if (a.ksplit > 1) {
        // ST (whole chunks per slice, round 5): ceil(chunks / ksplit) chunks of K*K stages each; the counted loop: any stage range
        const int per = ST ? ((a.n_it / (K * K) + a.ksplit - 1) / a.ksplit) * (K * K) : (a.n_it + a.ksplit - 1) / a.ksplit;
        it0 = ks * per < a.n_it ? ks * per : a.n_it;
        it1 = it0 + per < a.n_it ? it0 + per : a.n_it;
    }
    if (it1 > it0) {
        IterState st{0, 0, 0, 0, 0};
        {
            int r = it0;
            while (r >= a.src[st.s].nchunk * a.kk) {
                r -= a.src[st.s].nchunk * a.kk;
                ++st.s;
            }
            st.chunk = r / a.kk;
            st.tap = r - st.chunk * a.kk;
            st.kh = st.tap / K;
            st.kw = st.tap - st.kh * K;
        }
        // Pipeline (round 4; DESIGN 3.1b has the measurements).  The weight tile of stage it + 2 is requested by LDS-DMA during stage
        // it into the third buffer: no staging registers, no ds_write, and -- what matters -- three buffers, so that the stage barrier
        // can sit in the MIDDLE of a stage (see `stage` below).  (The two stages of lead are not what it is for: the register-staged
        // form did not wait for memory either -- with L2-resident weights, tools/w_resident.py, it ran exactly as fast.)  The halo of
        // the next (source, chunk) is requested during the LAST BUT ONE tap of the current chunk and stored at the end of the last one --
        // its staging registers are allocated for the whole loop anyway.  The waits in front of the barriers are COUNTED (VMEM
        // operations retire in order: a halo requested in this stage may still be in flight, everything older has landed), the
        // barrier is the raw s_barrier (__syncthreads' fence would drain the counter).  hipcc puts s_waitcnt vmcnt(0) in front of
        // every LDS read that MAY alias a transfer in flight: the buffer being filled and the buffers being read are __restrict__
        // parameters of an inlined function (scoped-noalias metadata, as in wgrad_row_bf16_kernel's DMA loop).
        // current stage: (c_s, c_chunk) + tap (kh, kw); its A rows start at halo pixel aoff = kh * HWD + kw (+ wave row, + x)
        int c_s = st.s, c_chunk = st.chunk, tap = st.tap, kw = st.kw, aoff = st.kh * HWD + st.kw;
        auto next_chunk = [&](int& ns, int& nc) {      // the (source, chunk) after the current one
            ns = c_s;
            nc = c_chunk + 1;
            if (nc == (c_s ? nch_s1 : nch_s0)) {
                nc = 0;
                ++ns;
            }
        };
        load_halo(c_s, c_chunk);
        dma_seek(c_s, c_chunk, tap);
        dma_next(&Bs[0][0]);
        if (it0 + 1 < it1) dma_next(&Bs[1][0]);
        store_halo();
        if (it0 + 1 < it1 && tap == K * K - 1) {      // (a K-split range that begins on the last tap of a chunk)
            int ns, nc;
            next_chunk(ns, nc);
            load_halo(ns, nc);
        }
        __syncthreads();                               // (drains the transfers: vmcnt(0))
        // vm: VMEM operations of this wave that may stay in flight; lds: also wait for this wave's LDS operations (halo stores)
        auto stage_sync = [&](int vm, bool lds) {
#ifdef LU_EMU
            __syncthreads();
#else
            if (lds) {
                if (vm == 0) __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(n) lgkmcnt(0), expcnt untouched
                else __builtin_amdgcn_s_waitcnt(0x0071);
            } else {
                if (vm == 0) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(n) only: the operand reads in flight stay in flight
                else __builtin_amdgcn_s_waitcnt(0x0F70 | HPASS);
            }
            __builtin_amdgcn_s_barrier();
#endif
        };
        // One stage.  On entry fa0 (A fragment s = 0) and fb[0] (B values of group 0) of THIS stage are already requested.
        //   Bd: the buffer the transfer of stage it + 2 fills, Br: this stage's tile, Bn: the next stage's tile.
        // The stage barrier sits between groups 3 and 4 (all but the last tap of a chunk): what it orders -- the tile of stage
        // it + 1 landed for every wave before anyone reads it, every wave done with stage it - 1 before its buffer is refilled --
        // does not need the END of a stage, and in the middle a wave arrives with its next operands already requested and leaves
        // with nothing to wait for: no LDS round trip behind the barrier, none at the stage boundary (groups 5 and 7 request the
        // next stage's first operands).  Last tap of a chunk: two barriers at the end as before (halo exchange).
        auto stage = [&](float* __restrict__ Bd, const float* __restrict__ Br, const float* __restrict__ Bn, int arow, int arow_n,
                         bool dma, int dmode, bool halo_ld, bool halo_st, bool more) {
            fb[1] = rd_b(Br, 1);
            fa1 = rd_a(arow, 1);
            mma4(fa0.x, fb[0]);
            LU_SCHED_FENCE();
            if (halo_ld) {
                int ns, nc;
                next_chunk(ns, nc);
                load_halo(ns, nc);
            }
            LU_SCHED_FENCE();
#pragma unroll
            for (int g = 1; g < 4; ++g) {
                fb[(g + 1) & 1] = rd_b(Br, g + 1);
                mma4(fa_of(g), fb[g & 1]);
                LU_SCHED_FENCE();
            }
            if (!halo_st) {
                stage_sync(halo_ld ? 1 : 0, false);
                if (dma) dma_issue(Bd, dmode);
            }
            LU_SCHED_FENCE();
#pragma unroll
            for (int g = 4; g < 8; ++g) {
                if (g < 7) fb[(g + 1) & 1] = rd_b(Br, g + 1);
                else if (more && !halo_st) fb[0] = rd_b(Bn, 0);
                if (g == 5 && more && !halo_st) fa0 = rd_a(arow_n, 0);
                mma4(fa_of(g), fb[g & 1]);
                LU_SCHED_FENCE();
            }
            if (halo_st) {
#ifdef LU_EMU
                __syncthreads();
#else
                __builtin_amdgcn_s_barrier();      // every wave is done with the old halo (its reads were waited for by its MFMAs)
#endif
                store_halo();                      // (the compiler's wait for the staged halo is vmcnt(0): the transfer goes behind it)
                if (dma) dma_issue(Bd, dmode);
                stage_sync(dma ? 1 : 0, true);
                if (more) {
                    fa0 = rd_a(arow_n, 0);
                    fb[0] = rd_b(Bn, 0);
                }
            }
        };
        int b0 = 0, b1 = 1, b2 = 2;
        const int arow0 = wave * HWD + (lane & 31);
        fa0 = rd_a(arow0 + aoff, 0);
        fb[0] = rd_b(&Bs[0][0], 0);
        if constexpr (ST) {
            // Launches without a K split (the bulk of a training step) walk whole chunks: the kernel rows 0 .. K - 2 of a chunk in a
            // run-time loop whose body is the K taps of a row as a compile-time sequence -- no halo exchange can fall into them, the
            // next tile is always the next tap of the same chunk, the A rows are the row's first halo pixel plus a constant -- and
            // the last row apart, with the halo request (tap K*K - 2), the halo exchange (K*K - 1) and the step of the tile stream
            // into the next chunk (behind tap K*K - 3) at fixed places.  The common path has no conditional branch and a third of
            // the scalar instructions of the counted loop below (rocprofv3 SQ_INSTS_SALU / SQ_INSTS_BRANCH per MFMA: 2.2 / 0.45
            // there against 0.85 / 0.06 in wgrad_row_kernel, profiles/r04_pmc_sq.json).
            const int n_chunks = (it1 - it0) / (K * K);
            auto rotate = [&]() {
                const int t = b0;
                b0 = b1;
                b1 = b2;
                b2 = t;
            };
            for (int ci = 0; ci < n_chunks; ++ci) {
                const bool has_next = ci + 1 < n_chunks;
                int rowb = arow0;      // halo pixel under (kernel row, kw = 0) of this wave's patch row
                for (int kh = 0; kh < K - 1; ++kh) {
                    lu_static_for<K>([&](auto kc) {
                        constexpr int kw_ = decltype(kc)::value;
                        stage(&Bs[b2][0], &Bs[b0][0], &Bs[b1][0], rowb + kw_, kw_ + 1 < K ? rowb + kw_ + 1 : rowb + HWD, true, 1, false,
                              false, true);
                        rotate();
                    });
                    rowb += HWD;
                }
                lu_static_for<K>([&](auto kc) {
                    constexpr int kw_ = decltype(kc)::value;
                    stage(&Bs[b2][0], &Bs[b0][0], &Bs[b1][0], rowb + kw_, kw_ + 1 < K ? rowb + kw_ + 1 : arow0, has_next || kw_ + 2 < K,
                          kw_ == K - 3 ? 2 : 1, kw_ == K - 2 && has_next, kw_ == K - 1 && has_next, has_next || kw_ + 1 < K);
                    rotate();
                });
                next_chunk(c_s, c_chunk);
            }
        } else {
        for (int it = it0; it < it1; ++it) {
            const bool dma = it + 2 < it1, last_tap = tap == K * K - 1;
            int aoff_n = aoff + 1;
            if (kw + 1 == K) aoff_n += HWD - K;
            if (last_tap) aoff_n = 0;
            stage(&Bs[b2][0], &Bs[b0][0], &Bs[b1][0], arow0 + aoff, arow0 + aoff_n, dma, 0, dma && tap == K * K - 2,
                  it + 1 < it1 && last_tap, it + 1 < it1);
            aoff = aoff_n;
            if (++kw == K) kw = 0;
            if (++tap == K * K) {
                tap = 0;
                next_chunk(c_s, c_chunk);
            }
            const int t = b0;
            b0 = b1;
            b1 = b2;
            b2 = t;
        }
        }
        // The gate epilogue turns fragments round in the halo image.  A wave that gets there has passed the last mid-stage barrier,
        // i.e. every wave has ISSUED its last reads of the image -- in practice thousands of cycles before the first exchange store;
        // formally they have only completed once their owners have waited for them: one more barrier per tile.
        if (EPI == LU_EPI_LSTM) __syncthreads();
    }

    // ---- epilogue: wave = patch row, accumulator row = x ----
    const int oy = y0 + wave;
    // 16-byte stores.  A lane holds ONE column of 16 pixels per fragment: storing from the registers is 64 four-byte stores per
    // lane (96 + 16 scalar loads for the gate block), each with its own 64-bit address -- tools/tile_fit.py: ~90 us per round of
    // tiles, 10 % of a fused step at the 256^2 level, and the two resident blocks of a CU reach it at the same time.  Each wave
    // turns its fragments round in a PRIVATE slice of the (dead: the loop ended on a barrier) halo LDS, no block barrier: a
    // lane then owns (pixel, four consecutive columns / channels).
    if (EPI == LU_EPI_BIAS) {
        // interleaved fragments: acc[0..3][r] are columns n0 + 4 l .. + 3 of pixel (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the row
        const int col = n0 + 4 * (lane & 31);
        const bool slab = a.ksplit > 1;
        const bool vec = (a.N & 3) == 0 && (slab ? (reinterpret_cast<uintptr_t>(a.ws) & 15) == 0 : a.out_vec4 != 0);
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!slab && a.bias) {
            if (col + 3 < a.N) {
                b4 = make_float4(a.bias[col], a.bias[col + 1], a.bias[col + 2], a.bias[col + 3]);
            } else {
                if (col < a.N) b4.x = a.bias[col];
                if (col + 1 < a.N) b4.y = a.bias[col + 1];
                if (col + 2 < a.N) b4.z = a.bias[col + 2];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (oy >= a.Hin || ox >= a.Win || col >= a.N) continue;
            const int64_t pix = (int64_t)oy * a.Win + ox;
            float* op;
            if (slab) {
                op = a.ws + ((int64_t)ks * a.M + (int64_t)f * a.HWo + pix) * a.N + col;
            } else if (a.out_row_stride) {      // strided output rows
                op = a.out + (int64_t)f * a.out_frame_stride + (int64_t)oy * a.out_row_stride + (int64_t)ox * a.out_pix_stride + col;
            } else {
                op = a.out + (int64_t)f * a.out_frame_stride + pix * a.out_pix_stride + col;
            }
            const float4 v = make_float4(acc[0][r] + b4.x, acc[1][r] + b4.y, acc[2][r] + b4.z, acc[3][r] + b4.w);
            if (vec) {
                *reinterpret_cast<float4*>(op) = v;
            } else {
                op[0] = v.x;
                if (col + 1 < a.N) op[1] = v.y;
                if (col + 2 < a.N) op[2] = v.z;
                if (col + 3 < a.N) op[3] = v.w;
            }
        }
        return;
    }
    if (EPI == LU_EPI_LSTM && a.lstm_vec4 && NF == 4 && HP * A_LD >= 8 * (8 * 132)) {
        float* const Exw = Ah + wave * (8 * 132);          // [8 pixels][4 gates x 32 channels + 4]
        const int F = a.F;
        const int lp = lane >> 3, cq = lane & 7;
        const int ch = nt * 32 + 4 * cq;                    // F % 32 == 0 is enforced by the host
        const float* const bp = a.bias ? a.bias + ch : lu_z16;      // (lu_z16: 16 bytes of zeros)
        const int bst = a.bias ? F : 0;
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            // previous cell state and the gate biases of this pass: requested before the exchange, which covers part of their
            // latency (held across the passes -- as in the fragment kernels -- they spill here: 128 VGPRs, two blocks per CU)
            const bool okp = oy < a.Hin && x0 + 8 * qt + lp < a.Win;
            const float4 cp = *reinterpret_cast<const float4*>(
                okp ? a.c_prev + (int64_t)f * a.c_prev_fs + ((int64_t)oy * a.Win + x0 + 8 * qt + lp) * F + ch : lu_z16);
            const float4 bi = *reinterpret_cast<const float4*>(bp), bf = *reinterpret_cast<const float4*>(bp + bst),
                         bg = *reinterpret_cast<const float4*>(bp + 2 * bst), bo = *reinterpret_cast<const float4*>(bp + 3 * bst);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)              // pixel 8 qt + rr + 4 (lane >> 5) of the row, gate nf
                    Exw[(rr + 4 * (lane >> 5)) * 132 + 32 * nf + (lane & 31)] = acc[nf][4 * qt + rr];
            LU_WAVE_SYNC();
            // one gate at a time (read, activate, store): all four z vectors live at once cost two spilled registers
            const float* ex = &Exw[lp * 132 + 4 * cq];
            const int ox = x0 + 8 * qt + lp;
            const int64_t pix = (int64_t)oy * a.Win + ox;
            float* const gp = a.gates_out ? a.gates_out + (int64_t)f * a.gates_fs + pix * (4 * F) + ch : nullptr;
            float4 z, gi, gf, gg, go, cn, hn;
            z = *reinterpret_cast<const float4*>(ex);
            gi.x = hard_sigmoid(z.x + bi.x); gi.y = hard_sigmoid(z.y + bi.y); gi.z = hard_sigmoid(z.z + bi.z); gi.w = hard_sigmoid(z.w + bi.w);
            if (okp && gp) *reinterpret_cast<float4*>(gp) = gi;
            z = *reinterpret_cast<const float4*>(ex + 32);
            gf.x = hard_sigmoid(z.x + bf.x); gf.y = hard_sigmoid(z.y + bf.y); gf.z = hard_sigmoid(z.z + bf.z); gf.w = hard_sigmoid(z.w + bf.w);
            if (okp && gp) *reinterpret_cast<float4*>(gp + F) = gf;
            z = *reinterpret_cast<const float4*>(ex + 64);
            gg.x = lu_tanh_fast(z.x + bg.x); gg.y = lu_tanh_fast(z.y + bg.y); gg.z = lu_tanh_fast(z.z + bg.z); gg.w = lu_tanh_fast(z.w + bg.w);
            if (okp && gp) *reinterpret_cast<float4*>(gp + 2 * F) = gg;
            cn.x = fmaf(gf.x, cp.x, gi.x * gg.x); cn.y = fmaf(gf.y, cp.y, gi.y * gg.y);      // explicit fmaf: every copy of the cell update contracts the same way
            cn.z = fmaf(gf.z, cp.z, gi.z * gg.z); cn.w = fmaf(gf.w, cp.w, gi.w * gg.w);
            if (okp) *reinterpret_cast<float4*>(a.c_out + (int64_t)f * a.c_out_fs + pix * F + ch) = cn;
            z = *reinterpret_cast<const float4*>(ex + 96);
            go.x = hard_sigmoid(z.x + bo.x); go.y = hard_sigmoid(z.y + bo.y); go.z = hard_sigmoid(z.z + bo.z); go.w = hard_sigmoid(z.w + bo.w);
            if (okp && gp) *reinterpret_cast<float4*>(gp + 3 * F) = go;
            hn.x = go.x * lu_tanh_fast(cn.x); hn.y = go.y * lu_tanh_fast(cn.y); hn.z = go.z * lu_tanh_fast(cn.z); hn.w = go.w * lu_tanh_fast(cn.w);
            if (okp) *reinterpret_cast<float4*>(a.h_out + (int64_t)f * a.h_fs + pix * F + ch) = hn;
            LU_WAVE_SYNC();
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (oy >= a.Hin || ox >= a.Win) continue;
        const int64_t pix = (int64_t)oy * a.Win + ox;
        float v[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) v[nf] = acc[nf][r];
        conv_epilogue_row<NF, EPI>(a, v, f, pix, (int64_t)f * a.HWo + pix, nt, n0, ks, lane & 31);
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16-MFMA variant of the halo kernel (mixed precision: fp32 activations / weights / accumulators in HBM, bf16
// MFMA operands).  v_mfma_f32_32x32x16_bf16 runs at 16x the fp32-MFMA rate, so everything around it has to shrink:
//   * same 8 x 32-pixel patch x 128 columns per block, 32-channel chunks; the halo is converted to bf16 while it is
//     staged ([HP][32+8] bf16: one ds_read_b128 = one A fragment of 8 k) and stays put for the K*K taps;
//   * the weights never touch LDS: pack_weights_bf16_kernel lays them out in MFMA B-fragment order
//     ([tap][chunk][column fragment][k-step][lane][8 bf16]), so one coalesced 16-byte global load per lane IS the
//     fragment.  The 8 waves are a 2 (pixel-row groups of 4) x 4 (column fragments) grid: a wave re-uses its B
//     fragment for 4 patch rows and prefetches the fragments of the next two stages into registers;
//   * consequently the tap loop has no barrier at all -- waves only meet when the halo is replaced (every K*K stages);
//   * ConvLSTM epilogue: the four gates of a channel sit in four different waves, so the accumulators are exchanged
//     through LDS (the halo buffer is dead by then) before the usual gate epilogue.  Same K split as the fp32 kernel.
// ---------------------------------------------------------------------------------------------------------
constexpr int CKB = 32;      // channels per stage (two k = 16 MFMA steps)
constexpr int LDB = CKB + 8; // bf16 elements per halo pixel in LDS (80 B: conflict-free ds_read_b128, see A_LD)

// packed index of (tap, chunk, column n, channel c):  fragment-major, then k-step, lane, element
__device__ __forceinline__ void pack_weights_bf16_body(const float* __restrict__ w, int64_t tap_stride, int row_stride, int kk,
                                                       int C, int N, unsigned short* __restrict__ out, int64_t first, int64_t step) {
    // one thread = one lane's fragment (8 consecutive channels of one column): eight reads that are coalesced ACROSS the lanes
    // (consecutive columns), one 16-byte store (75 M weights are re-packed after every optimiser step)
    const int nchunk = (C + CKB - 1) / CKB, nfr = (N + 31) / 32;
    const int64_t total = (int64_t)kk * nchunk * nfr * 128;
    for (int64_t i = first; i < total; i += step) {
        const int ln = (int)(i & 63), j = (int)((i >> 6) & 1);
        int64_t t = i >> 7;
        const int fr = (int)(t % nfr);
        t /= nfr;
        const int chunk = (int)(t % nchunk);
        const int tap = (int)(t / nchunk);
        const int n = fr * 32 + (ln & 31);
        const int c0 = chunk * CKB + 16 * j + 8 * (ln >> 5);
        const float* src = w + (int64_t)tap * tap_stride + (int64_t)c0 * row_stride + n;
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = (c0 + e < C && n < N) ? lu_f2bf(src[(int64_t)e * row_stride]) : (unsigned short)0;
        lu_u4 pk;
        pk.x = (unsigned)v[0] | ((unsigned)v[1] << 16);
        pk.y = (unsigned)v[2] | ((unsigned)v[3] << 16);
        pk.z = (unsigned)v[4] | ((unsigned)v[5] << 16);
        pk.w = (unsigned)v[6] | ((unsigned)v[7] << 16);
        *reinterpret_cast<lu_u4*>(out + (i << 3)) = pk;
    }
}

__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, int64_t tap_stride, int row_stride, int kk, int C,
                                         int N, unsigned short* __restrict__ out) {
    pack_weights_bf16_body(w, tap_stride, row_stride, kk, C, N, out, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                           (int64_t)gridDim.x * blockDim.x);
}

// precision 'bf16x3': the fragment image of the SPLIT kernel [tap][6][cp][N] (block j = piece `order` names of w, rows [C, cp) of a block
// zero: lu_split6 of the weights + pack_weights_bf16_kernel in one pass -- 4 bytes read, 12 written per weight instead of 4 + 24 + 24 + 12)
__global__ void pack_weights_split6_bf16_kernel(const float* __restrict__ w, int64_t tap_stride, int row_stride, int kk, int C, int cp,
                                                int N, unsigned order, unsigned short* __restrict__ out) {
    const int C6 = 6 * cp;
    const int nchunk = (C6 + CKB - 1) / CKB, nfr = (N + 31) / 32;
    const int64_t total = (int64_t)kk * nchunk * nfr * 128;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ln = (int)(i & 63), j = (int)((i >> 6) & 1);
        int64_t t = i >> 7;
        const int fr = (int)(t % nfr);
        t /= nfr;
        const int chunk = (int)(t % nchunk);
        const int tap = (int)(t / nchunk);
        const int n = fr * 32 + (ln & 31);
        const int c0 = chunk * CKB + 16 * j + 8 * (ln >> 5);
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ca = c0 + e, blk = ca / cp, c = ca - blk * cp;      // augmented channel -> (block, channel)
            float hi = 0.f, mid = 0.f, lo = 0.f;
            if (ca < C6 && c < C && n < N) lu_split3(w[(int64_t)tap * tap_stride + (int64_t)c * row_stride + n], hi, mid, lo);
            const unsigned pc = (order >> (2 * (blk < 6 ? blk : 0))) & 3u;
            v[e] = lu_f2bf(pc == 0 ? hi : (pc == 1 ? mid : lo));
        }
        lu_u4 pk;
        pk.x = (unsigned)v[0] | ((unsigned)v[1] << 16);
        pk.y = (unsigned)v[2] | ((unsigned)v[3] << 16);
        pk.z = (unsigned)v[4] | ((unsigned)v[5] << 16);
        pk.w = (unsigned)v[6] | ((unsigned)v[7] << 16);
        *reinterpret_cast<lu_u4*>(out + (i << 3)) = pk;
    }
}

// Epilogue of the fragment kernels (both loop generations): bias / K-split stores, or the fused ConvLSTM gate block with
// its exchange of the four gate fragments through the (dead) halo LDS.
template <int EPI, int RW, int NFR = 4>
__device__ __forceinline__ void frag_epilogue(const ConvArgs& a, f32x16 (&acc)[RW], unsigned char* Ah, int f, int y0, int x0,
                                              int nt, int n0, int ks) {
    constexpr int BN = 32 * NFR, TW = 32, EX_LD = BN + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NFR, wn = wave % NFR;
    if (LU_DBG(a, 8)) {      // ablation: no epilogue (one store keeps the accumulators alive)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[i][r];
        if (t == 12345.678f) a.out[0] = t;
        return;
    }
    if (EPI == LU_EPI_LSTM) {
        // exchange: wave (wm, wn) holds gate wn of rows RW wm .. RW wm + RW - 1; the gate epilogue wants the four gates of
        // a (pixel, channel) in one lane.  Per pass one patch row of each row group goes through LDS.
        float* Ex = reinterpret_cast<float*>(Ah);      // [2 row groups][32 px][EX_LD]
        // One (pixel, channel quad) per thread and pass: 64 pixels x 8 quads = 512 threads; 16-byte loads / stores
        // (8-byte for the bf16 tape) instead of one scalar per gate plane.
        const int F = a.F;
        const int pr = tid >> 3, cq = tid & 7;
        const int g = pr >> 5, px = pr & 31;
        const int ch = nt * 32 + 4 * cq;                // F % 32 == 0 is enforced by the host
        float4 bi = make_float4(0.f, 0.f, 0.f, 0.f), bf = bi, bg = bi, bo = bi;
        if (a.bias) {
            bi = *reinterpret_cast<const float4*>(a.bias + ch);
            bf = *reinterpret_cast<const float4*>(a.bias + F + ch);
            bg = *reinterpret_cast<const float4*>(a.bias + 2 * F + ch);
            bo = *reinterpret_cast<const float4*>(a.bias + 3 * F + ch);
        }
        // the previous cell state of all RW passes up front: inside the pass loop (two barriers per pass) every pass would wait
        // for its own load -- RW global-memory latencies in a row, with one block per CU nothing else covers them
        float4 cpv[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int oy = y0 + RW * g + i, ox = x0 + px;
            const bool ok = oy < a.Hin && ox < a.Win;
            cpv[i] = *reinterpret_cast<const float4*>(ok ? a.c_prev + (int64_t)f * a.c_prev_fs + ((int64_t)oy * a.Win + ox) * F + ch
                                                         : (a.zero16 ? a.zero16 : lu_zero16));
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            __syncthreads();                            // pass 0: all halo reads finished; later: previous pass consumed
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pxr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Ex[(wm * TW + pxr) * EX_LD + 32 * wn + (lane & 31)] = acc[i][r];
            }
            __syncthreads();
            const int oy = y0 + RW * g + i, ox = x0 + px;
            if (oy >= a.Hin || ox >= a.Win) continue;
            const int64_t pix = (int64_t)oy * a.Win + ox;
            const float* ex = &Ex[(g * TW + px) * EX_LD + 4 * cq];
            const float4 zi = *reinterpret_cast<const float4*>(ex), zf = *reinterpret_cast<const float4*>(ex + 32),
                         zg = *reinterpret_cast<const float4*>(ex + 64), zo = *reinterpret_cast<const float4*>(ex + 96);
            const float4 cp = cpv[i];
            float4 gi, gf, gg, go, cn, hn;
#define LU_GATE(m)                                      \
    gi.m = hard_sigmoid(zi.m + bi.m);                   \
    gf.m = hard_sigmoid(zf.m + bf.m);                   \
    gg.m = lu_tanh_fast(zg.m + bg.m);                   \
    go.m = hard_sigmoid(zo.m + bo.m);                   \
    cn.m = fmaf(gf.m, cp.m, gi.m * gg.m);               \
    hn.m = go.m * lu_tanh_fast(cn.m);
            LU_GATE(x) LU_GATE(y) LU_GATE(z) LU_GATE(w)
#undef LU_GATE
            *reinterpret_cast<float4*>(a.c_out + (int64_t)f * a.c_out_fs + pix * F + ch) = cn;
            *reinterpret_cast<float4*>(a.h_out + (int64_t)f * a.h_fs + pix * F + ch) = hn;
            if (a.h16_out && a.h16_split) {
                // precision 'bf16x3': the next step's recurrent operand = the exact three-way bf16 split of h, six channel blocks
                // in order A (lu_split6) -- written here, where h is in registers, instead of by a pass over h per step
                float4 hh, hm, hl;
                lu_split3(hn.x, hh.x, hm.x, hl.x);
                lu_split3(hn.y, hh.y, hm.y, hl.y);
                lu_split3(hn.z, hh.z, hm.z, hl.z);
                lu_split3(hn.w, hh.w, hm.w, hl.w);
                lu_u2 vh, vm, vl;
                vh.x = lu_pack2bf(hh.x, hh.y); vh.y = lu_pack2bf(hh.z, hh.w);
                vm.x = lu_pack2bf(hm.x, hm.y); vm.y = lu_pack2bf(hm.z, hm.w);
                vl.x = lu_pack2bf(hl.x, hl.y); vl.y = lu_pack2bf(hl.z, hl.w);
                unsigned short* hp = a.h16_out + (int64_t)f * a.h16_fs + pix * (6 * F) + ch;
                *reinterpret_cast<lu_u2*>(hp) = vl;
                *reinterpret_cast<lu_u2*>(hp + F) = vm;
                *reinterpret_cast<lu_u2*>(hp + 2 * F) = vh;
                *reinterpret_cast<lu_u2*>(hp + 3 * F) = vm;
                *reinterpret_cast<lu_u2*>(hp + 4 * F) = vh;
                *reinterpret_cast<lu_u2*>(hp + 5 * F) = vh;
            } else if (a.h16_out) {
                lu_u2 hv;
                hv.x = lu_pack2bf(hn.x, hn.y);
                hv.y = lu_pack2bf(hn.z, hn.w);
                *reinterpret_cast<lu_u2*>(a.h16_out + (int64_t)f * a.h16_fs + pix * F + ch) = hv;
            }
            if (a.gates_out) {
                if (a.gates_bf16) {
                    unsigned short* gp = reinterpret_cast<unsigned short*>(a.gates_out) + (int64_t)f * a.gates_fs + pix * (4 * F) + ch;
                    lu_u2 v;
                    v.x = lu_pack2bf(gi.x, gi.y); v.y = lu_pack2bf(gi.z, gi.w);
                    *reinterpret_cast<lu_u2*>(gp) = v;
                    v.x = lu_pack2bf(gf.x, gf.y); v.y = lu_pack2bf(gf.z, gf.w);
                    *reinterpret_cast<lu_u2*>(gp + F) = v;
                    v.x = lu_pack2bf(gg.x, gg.y); v.y = lu_pack2bf(gg.z, gg.w);
                    *reinterpret_cast<lu_u2*>(gp + 2 * F) = v;
                    v.x = lu_pack2bf(go.x, go.y); v.y = lu_pack2bf(go.z, go.w);
                    *reinterpret_cast<lu_u2*>(gp + 3 * F) = v;
                } else {
                    float* gp = a.gates_out + (int64_t)f * a.gates_fs + pix * (4 * F) + ch;
                    *reinterpret_cast<float4*>(gp) = gi;
                    *reinterpret_cast<float4*>(gp + F) = gf;
                    *reinterpret_cast<float4*>(gp + 2 * F) = gg;
                    *reinterpret_cast<float4*>(gp + 3 * F) = go;
                }
            }
        }
        return;
    }
    if (EPI == LU_EPI_BIAS && a.out_vec4 && a.ksplit <= 1) {
        // Bias epilogue with 16-byte stores.  A lane of an accumulator fragment holds ONE column of 16 pixels, so storing from the
        // registers is 16 RW four-byte stores per lane, each with its own 64-bit address -- measured 36 us of a 16 x 32-pixel
        // tile (tools/tile_fit.py: the per-tile constant of the 5x5 input gradients, 51 -> 15 us without the epilogue), a quarter
        // of a 3x3 layer.  Each wave turns its fragments round, 16 pixels at a time, in a private slice of the (dead) halo LDS --
        // no block barrier after the first -- and a lane then owns (pixel, four consecutive columns).
        float* const Exw = reinterpret_cast<float*>(Ah) + wave * (16 * 36);      // this wave's slice: [16 pixels][32 columns + 4]
        const int cq = lane & 7;
        const int col = n0 + 32 * wn + 4 * cq;
        const float4 bq = (a.bias && col < a.N) ? *reinterpret_cast<const float4*>(a.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();                                    // every wave is done with the halo images
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int oy = y0 + RW * wm + i;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr)              // pixel 16 half + (rr & 3) + 8 (rr >> 2) + 4 (lane >> 5) of the row
                    Exw[((rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[i][8 * half + rr];
                LU_WAVE_SYNC();
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int lp = (lane >> 3) + 8 * q;
                    const int ox = x0 + 16 * half + lp;
                    float4 v = *reinterpret_cast<const float4*>(&Exw[lp * 36 + 4 * cq]);
                    v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
                    if (oy < a.Hin && ox < a.Win && col < a.N)
                        *reinterpret_cast<float4*>(a.out + (int64_t)f * a.out_frame_stride +
                                                   ((int64_t)oy * a.Win + ox) * a.out_pix_stride + col) = v;
                }
                LU_WAVE_SYNC();
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int oy = y0 + RW * wm + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (oy >= a.Hin || ox >= a.Win) continue;
            const int64_t pix = (int64_t)oy * a.Win + ox;
            float v[1] = {acc[i][r]};
            conv_epilogue_row<1, EPI>(a, v, f, pix, (int64_t)f * a.HWo + pix, nt, n0 + 32 * wn, ks, lane & 31);
        }
    }
}

// B16 = true (bf16 MFMA only): every source is ALREADY a bf16 tensor (the bf16 BPTT tape: h sequence, dz; the bf16 copy of a
// block input; the im2col image of the thin first input) -- a piece is then 8 channels = 16 raw bytes, half as many
// loads, no conversion.  The element type is a compile-time property of the launch: a run-time (even uniform) branch
// around the loads makes the compiler's vmcnt accounting conservative (measured: 5-28 % slower).
// NFR = column fragments (of 32) per block: 4 (128 columns, 2 row groups of waves) for the wide layers; 2 / 1 for the narrow
// decoder tail (N = 64 / 32: 4 / 8 row groups of RW = 2 / 1 rows, i.e. the same 8 x 32 patch) -- those layers are bound by
// HBM, not by the matrix pipe, and all they need is the halo staged once and every wave busy on its own rows.
template <int K, int EPI, int RW, bool B16, int NFR = 4>      // RW = patch rows per wave: 4 (8 x 32 patch) or 8 (16 x 32 patch)
__global__ __launch_bounds__(512, 2) void conv_halo_frag_kernel(ConvArgs a) {
    const float* const lu_z16 = a.zero16 ? a.zero16 : lu_zero16;
    static_assert(NFR == 4 || (EPI == LU_EPI_BIAS && (NFR == 1 || NFR == 2)), "narrow blocks: bias epilogue only");
    constexpr int BN = 32 * NFR, NT = 512, TH = (8 / NFR) * RW, TW = 32;
    constexpr int HWD = TW + K - 1, HHT = TH + K - 1, HP = HHT * HWD;
    constexpr int CKS = CKB;                           // channels per stage
    constexpr int PC = B16 ? 8 : 4;                    // channels per 16-byte piece
    constexpr int G = CKS / PC;                        // 16-byte global channel groups per halo pixel
    constexpr int ESZ = B16 ? 2 : 4;                   // bytes per source element
    constexpr int HPASS = (HP * G + NT - 1) / NT;      // halo pieces: 16 bytes of one pixel per thread each
    constexpr int PAD = (K - 1) / 2;
    constexpr int EX_LD = BN + 4;                      // floats per pixel of the gate-exchange buffer
    constexpr int PITCH = 80;                          // bytes per halo pixel: (16 + 4) floats or (32 + 8) bf16
    constexpr int AH_BYTES = HP * PITCH;               // one halo image; two of them live in dynamic LDS
    static_assert(HPASS + 2 <= K * K, "the next halo is fetched one piece per tap and stored two taps later");
    static_assert(EPI != LU_EPI_LSTM || 2 * TW * EX_LD * 4 <= 2 * AH_BYTES, "gate exchange aliases the two halo images");
    static_assert(8 * TW * EX_LD * 4 <= 2 * AH_BYTES * (BN / 32), "bias-epilogue exchange (8 / NFR row groups) aliases the two halo images");
    LU_DYN_LDS(unsigned char, Ah);                     // [2][AH_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NFR, wn = wave % NFR;
    int tile, nt, ks;
    if (!lu_block_tile(a, tile, nt, ks)) return;
    const int f = tile / a.tiles_pf;
    const int t2 = tile - f * a.tiles_pf;
    const int y0 = (t2 / a.tiles_x) * TH, x0 = (t2 % a.tiles_x) * TW;
    const int n0 = nt * BN;
    const lu_u4* const zp = reinterpret_cast<const lu_u4*>(lu_z16);
    const int q = tid % G;                 // channel group inside the chunk
    // this wave's column fragment (32 output columns): gate wn of channels [32 nt, 32 nt + 32) / plain columns
    const int nfr = (a.N + 31) >> 5;
    const int frag = (EPI == LU_EPI_LSTM) ? (wn * a.F + nt * 32) >> 5 : nt * NFR + wn;
    const bool frag_ok = frag < nfr;
    // Per-source fields live in registers and are picked with selects: indexing a.src[st.s] inside the tap loop costs a
    // scalar kernarg load + s_waitcnt lgkmcnt(0) per use, and that wait also drains the LDS fragment reads in flight.
    const unsigned char* const x_s0 = reinterpret_cast<const unsigned char*>(a.src[0].x) + (int64_t)f * a.src[0].frame_stride * ESZ;
    const unsigned char* const x_s1 = reinterpret_cast<const unsigned char*>(a.src[1].x) + (int64_t)f * a.src[1].frame_stride * ESZ;
    const unsigned char* const w_s0 = reinterpret_cast<const unsigned char*>(a.src[0].w) + (int64_t)frag * 2048 + lane * 16;
    const unsigned char* const w_s1 = reinterpret_cast<const unsigned char*>(a.src[1].w) + (int64_t)frag * 2048 + lane * 16;
    const int ps_s0 = a.src[0].pix_stride, ps_s1 = a.src[1].pix_stride;
    const int C_s0 = a.src[0].C, C_s1 = a.src[1].C;
    const int nch_s0 = a.src[0].nchunk, nch_s1 = a.src[1].nchunk;
    const bool ctr1 = a.src1_center != 0;      // source 1 = one chunk, centre tap only (it is the LAST stage)
    const int kk = K * K;
    auto tap_advance = [&](IterState& st) {          // iter_advance without the kernarg look-ups
        ++st.tap;
        if (++st.kw == K) {
            st.kw = 0;
            ++st.kh;
        }
        if (st.tap < kk && !(ctr1 && st.s)) return;
        st.tap = st.kh = st.kw = 0;
        if (++st.chunk == (st.s ? nch_s1 : nch_s0)) {
            st.chunk = 0;
            ++st.s;
        }
        if (ctr1 && st.s) st.kh = st.kw = PAD;
    };
    auto next_chunk = [&](IterState& st) {            // first tap of the next chunk
        st.tap = st.kh = st.kw = 0;
        if (++st.chunk == (st.s ? nch_s1 : nch_s0)) {
            st.chunk = 0;
            ++st.s;
        }
    };

    // piece p of the halo of chunk `st`: pixel hp = (tid + 512 p) / G, channels PC q .. PC q + PC - 1 (recomputed per use: the
    // addressing is a handful of integer ops once per K*K MFMA stages, the registers are worth more)
    auto piece_load = [&](int p, const IterState& st, lu_u4& r, bool want) {
        const int hp = (tid + NT * p) / G;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int iy = y0 + hy - PAD, ix = x0 + hx - PAD;
        const bool ok = want && hp < HP && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        const int c = st.chunk * CKS + PC * q;
        const int64_t off = ((int64_t)(iy * a.Win + ix) * (st.s ? ps_s1 : ps_s0) + c) * ESZ;
        r = *((ok && c < (st.s ? C_s1 : C_s0)) ? reinterpret_cast<const lu_u4*>((st.s ? x_s1 : x_s0) + off) : zp);   // (16-byte aligned)
    };
    auto piece_store = [&](int p, int hb, const lu_u4& r) {
        const int hp = (tid + NT * p) / G;
        if (hp < HP) {
            if (B16) {      // raw copy: 8 bf16
                *reinterpret_cast<lu_u4*>(&Ah[hb * AH_BYTES + hp * PITCH + 16 * q]) = r;
            } else {
                lu_u2 v;
                v.x = lu_pack2bf(lu_bits2f(r.x), lu_bits2f(r.y));
                v.y = lu_pack2bf(lu_bits2f(r.z), lu_bits2f(r.w));
                *reinterpret_cast<lu_u2*>(&Ah[hb * AH_BYTES + hp * PITCH + 8 * q]) = v;
            }
        }
    };
    // B fragments of one stage (two k-steps), straight from the packed weights
    auto load_b = [&](const IterState& st, float4& b0, float4& b1) {
        const unsigned char* wp = (st.s ? w_s1 : w_s0) +
                                  ((int64_t)st.tap * (st.s ? nch_s1 : nch_s0) + st.chunk) * nfr * 2048;
        const float* p0 = frag_ok ? reinterpret_cast<const float*>(wp) : lu_z16;
        const float* p1 = frag_ok ? reinterpret_cast<const float*>(wp + 1024) : lu_z16;
        b0 = *reinterpret_cast<const float4*>(p0);
        b1 = *reinterpret_cast<const float4*>(p1);
    };

    f32x16 acc[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int khalf16 = 16 * (lane >> 5);      // byte offset of this half-wave's k group inside a pixel: 8 bf16 / 4 floats
    auto mma_stage = [&](const IterState& st, int hb, const float4& b0, const float4& b1) {
        const int arow = (RW * wm + st.kh) * HWD + (lane & 31) + st.kw;
        const unsigned char* ab = &Ah[hb * AH_BYTES + arow * PITCH + khalf16];
#ifdef LU_ABL_MFMA_ONLY      // ablation build (tools only): the MFMA stream without LDS reads -- practical MFMA ceiling
    const lu_bf16x8 u0 = __builtin_bit_cast(lu_bf16x8, b0), u1 = __builtin_bit_cast(lu_bf16x8, b1);
#pragma unroll
        for (int i = 0; i < RW; ++i) acc[i] = lu_mfma_bf16(u1, u0, acc[i]);
#pragma unroll
        for (int i = 0; i < RW; ++i) acc[i] = lu_mfma_bf16(u0, u1, acc[i]);
        (void)ab;
        return;
#endif
        const lu_bf16x8 bv0 = __builtin_bit_cast(lu_bf16x8, b0), bv1 = __builtin_bit_cast(lu_bf16x8, b1);
        if (LU_DBG(a, 2)) {      // ablation: no A-fragment LDS reads
#pragma unroll
            for (int i = 0; i < RW; ++i) acc[i] = lu_mfma_bf16(bv1, bv0, acc[i]);
#pragma unroll
            for (int i = 0; i < RW; ++i) acc[i] = lu_mfma_bf16(bv0, bv1, acc[i]);
            return;
        }
        // k-step 0 for every row, then k-step 1 (back-to-back MFMAs never depend on each other); all k-step-0 fragments
        // are requested up front and every k-step-1 read hides behind a k-step-0 MFMA
        lu_bf16x8 a0[RW], a1[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) a0[i] = *reinterpret_cast<const lu_bf16x8*>(ab + i * HWD * PITCH);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            acc[i] = lu_mfma_bf16(a0[i], bv0, acc[i]);
            a1[i] = *reinterpret_cast<const lu_bf16x8*>(ab + i * HWD * PITCH + 32);
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) acc[i] = lu_mfma_bf16(a1[i], bv1, acc[i]);
        LU_SCHED_GROUP(0x100, RW);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            LU_SCHED_GROUP(0x008, 1);
            LU_SCHED_GROUP(0x100, 1);
        }
        LU_SCHED_GROUP(0x008, RW);
    };

    // K split: whole chunks per slice, so that every block starts at tap 0 of a chunk
    int it0 = 0, it1 = a.n_it;
    if (a.ksplit > 1) {
        const int chunks = a.n_it / kk;
        const int per = (chunks + a.ksplit - 1) / a.ksplit * kk;
        it0 = ks * per < a.n_it ? ks * per : a.n_it;
        it1 = it0 + per < a.n_it ? it0 + per : a.n_it;
    }
    if (it1 > it0) {
        IterState st{0, 0, 0, 0, 0};
        {
            int r = it0;
            if (r >= nch_s0 * kk) {
                r -= nch_s0 * kk;
                st.s = 1;
            }
            st.chunk = r / kk;       // r is a multiple of kk: tap = 0
        }
        // Register ring of B fragments, D stages deep: slot s holds stage it0 + s (mod D) and is refilled for the stage D
        // further on as soon as its MFMAs are issued -- the distance has to cover the L2 latency under load.
        constexpr int D = RW == 8 ? 4 : 2;
        float4 rb0[D], rb1[D];
        IterState sS[D];
        sS[0] = st;
#pragma unroll
        for (int j = 1; j < D; ++j) {
            sS[j] = sS[j - 1];
            if (it0 + j < it1) tap_advance(sS[j]);       // (clamped at the last stage)
        }
#pragma unroll
        for (int j = 0; j < D; ++j) load_b(sS[j], rb0[j], rb1[j]);
        {
            lu_u4 rh[HPASS];                     // first halo: all pieces at once
#pragma unroll
            for (int p = 0; p < HPASS; ++p) piece_load(p, st, rh[p], true);
#pragma unroll
            for (int p = 0; p < HPASS; ++p) piece_store(p, 0, rh[p]);
        }
        __syncthreads();
        int hb = 0;                              // halo image in use
        IterState nc = st;                       // chunk after the current one (valid while it exists)
        next_chunk(nc);
        lu_u4 rp[2] = {lu_u4{0u, 0u, 0u, 0u}, lu_u4{0u, 0u, 0u, 0u}};
        int pend[2] = {-1, -1};
        // One pipeline step = stage `it` in ring slot s.  While a chunk's taps run, the NEXT chunk's halo is fetched one
        // piece per tap into the other LDS image (two staging registers, alternating with the stage parity).
        auto step = [&](int it, auto slot) {
            constexpr int s = decltype(slot)::value;
            IterState& sc = sS[s];
            const bool fetch = sc.tap < HPASS && it - sc.tap + kk < it1;      // a next chunk exists in this slice
            const bool last_tap = sc.tap == kk - 1;
            LU_SCHED_FENCE();
            mma_stage(sc, hb, rb0[s], rb1[s]);
            LU_SCHED_FENCE();
            // The piece requested two stages ago is older than the B fragments the MFMAs above just waited for, so it has
            // landed: no extra vmcnt wait.  Then request this tap's piece and refill the B slot for stage it + D.
            // A piece load is issued every stage (the zero block when there is nothing to fetch) and the loop body has no
            // conditional step: the number of loads in flight is then the same on every path and the compiler's vmcnt
            // waits stay exact (a path-dependent count makes it wait for the youngest load, i.e. HBM latency per stage).
            if (pend[s & 1] >= 0) {
                if (!LU_DBG(a, 4)) piece_store(pend[s & 1], hb ^ 1, rp[s & 1]);
                pend[s & 1] = -1;
            }
#ifndef LU_ABL_MFMA_ONLY
            if (!LU_DBG(a, 4)) piece_load(fetch ? sc.tap : 0, fetch ? nc : sc, rp[s & 1], fetch);
#endif
            if (fetch) pend[s & 1] = sc.tap;
            sc = sS[(s + D - 1) % D];            // state of stage it + D - 1 ...
            if (it + D < it1) tap_advance(sc);      // ... + 1 (past the end: re-reads the last fragments, unused)
#ifndef LU_ABL_MFMA_ONLY
            if (!LU_DBG(a, 1)) load_b(sc, rb0[s], rb1[s]);
#endif
            if (last_tap && it + 1 < it1) {      // (HPASS + 2 <= K*K: the staged pieces have been retired by now)
                if (!LU_DBG(a, 16)) __syncthreads();          // the next halo is complete and every wave is done with the old one
                hb ^= 1;
                next_chunk(nc);
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2 % D>;
        using S3 = std::integral_constant<int, 3 % D>;
        int it = it0;
        for (; it + D <= it1; it += D) {
            step(it, S0());
            step(it + 1, S1());
            if (D == 4) {
                step(it + 2, S2());
                step(it + 3, S3());
            }
        }
        if (it < it1) step(it, S0());
        if (it + 1 < it1) step(it + 1, S1());
        if (D == 4 && it + 2 < it1) step(it + 2, S2());
    }

    frag_epilogue<EPI, RW, NFR>(a, acc, Ah, f, y0, x0, nt, n0, ks);
}

// ---------------------------------------------------------------------------------------------------------
// The bf16 fragment kernel the product launches for the wide layers is conv_halo_frag3_kernel below (third loop generation, round 5).
// Generation 1 (conv_halo_frag_kernel above) advances a run-time (source, chunk, tap) state machine per pipeline stage -- ~200 scalar /
// vector instructions between two groups of 16 MFMAs -- and is kept for the shapes it still serves: 3x3 layers on fp32 sources /
// 8-row patches and the narrow N = 32 / 64 blocks.  Generation 2 (round 2-4: the K*K taps of a chunk as a compile-time unrolled
// sequence, kernel row by kernel row, one LDS fragment read per MFMA) was the step between them; it left the library in round 6 with its
// measurements in DESIGN 3.3 / profiles/HISTORY.md.  What generation 3 inherits from it: only the chunk (source pointer, weight base) is
// run-time state, updated once per K*K stages.
// ---------------------------------------------------------------------------------------------------------
struct ChunkDesc {
    const unsigned char* x;      // source activations of this block's frame (byte pointer)
    const unsigned char* w;      // this lane's fragment bytes of (tap 0, chunk)
    int64_t wts;                 // bytes between taps
    int ps, C, c0;               // pixel stride (elements), channels of the source, first channel of the chunk
    int single;                  // 1: the centre-tap-only chunk (its packed image holds ONE tap)
};

// WM = row groups of waves per block.  2: 8 waves, a (2 RW) x 32 patch -- with RW = 8 (237 VGPRs) ONE block per CU, whose 8 waves
// meet at the chunk barriers, start together and reach the epilogue together: nothing covers a tile's prologue (first halo: a
// global-memory latency + LDS store + barrier), its epilogue (fragment exchange, gate math, stores) or the skew at a barrier.
// 1 ("half blocks", round 4): 4 waves = ONE row group x 4 column fragments, an RW x 32 patch, 256 threads -- the same RW MFMAs
// per weight fragment and the same registers per wave, but TWO INDEPENDENT blocks per CU (2 x 2 halo images of 8 + K - 1 rows:
// 138 KB at K = 5): one block's prologue / epilogue / barrier wait runs under the other block's MFMAs.
// ---------------------------------------------------------------------------------------------------------
// Third loop generation of the bf16 fragment kernel (round 5): same tile, same LDS images, same weight-fragment stream, same
// epilogue as the second generation -- another ORDER of the K*K taps of a chunk, chosen for the bytes it moves per MFMA.
// Generations 1-2 walk the taps kernel row by kernel row and read, per tap, one A fragment (ds_read_b128, 1 KB per wave) per
// patch row and channel half: ONE LDS read per MFMA -- at the bf16 rate that is half of the LDS pipe's 256 B/clk at full MFMA
// speed, and on these power-limited kernels (1.6-1.9 GHz under load) LDS bytes are watts.  But tap (kh, kw) on
// patch row i and tap (kh + 1, kw) on patch row i - 1 read the SAME halo pixels: walking a kernel COLUMN kw, halo row j of a
// wave's RW + K - 1 rows feeds the up-to-K products (row i = j - kh, tap (kh, kw)) from one read:
//     per (kernel column, channel half):  RW + K - 1 reads for RW * K MFMAs      (K = 5, RW = 8: 12 for 40, 0.3 per MFMA; K = 3: 10 for 24)
// The fragment of a halo row is live for its own step only (a ring of LA + 1 fragments runs LA steps ahead); the weight ring
// holds one kernel column (slot = kh, both channel halves; a slot is refilled, one half at a time, right behind its last use --
// RW + K - 1 + K steps of lead); the next chunk's halo pieces are requested / stored at sub-stage boundaries (a sub-stage = one
// kernel column x one channel half).  Everything is a compile-time sequence as in generation 2; only the chunk is run-time state.
// Each accumulator sums its taps column-major instead of row-major: another fp32 summation order -- results agree with
// generation 1 to rounding (tests compare with the oracle at 5e-5), not bit for bit.
// ---------------------------------------------------------------------------------------------------------
template <int K, int EPI, int RW, bool B16, int WM = 2>
__global__ __launch_bounds__(256 * WM, 2) void conv_halo_frag3_kernel(ConvArgs a) {
    const float* const lu_z16 = a.zero16 ? a.zero16 : lu_zero16;
    constexpr int BN = 128, NT = 256 * WM, TH = WM * RW, TW = 32, KK = K * K;
    constexpr int HWD = TW + K - 1, HHT = TH + K - 1, HP = HHT * HWD;
    constexpr int CKS = CKB;
    constexpr int PC = B16 ? 8 : 4;
    constexpr int G = CKS / PC;
    constexpr int ESZ = B16 ? 2 : 4;
    constexpr int HPASS = (HP * G + NT - 1) / NT;
    constexpr int PAD = (K - 1) / 2;
    constexpr int EX_LD = BN + 4;
    constexpr int PITCH = 80;
    constexpr int AH_BYTES = HP * PITCH;
    constexpr int NJ = RW + K - 1;                     // halo rows under a wave's RW patch rows = fragment steps per sub-stage
    constexpr int NSUB = 2 * K;                        // sub-stages per chunk: (kernel column, channel half)
    constexpr int NS = NSUB * NJ;                      // fragment steps per chunk
#ifndef LU_G3_LA
#define LU_G3_LA 3      // (tools/build_variant.py: A/B builds only)
#endif
    constexpr int LA = LU_G3_LA, NR = LA + 1;          // A-fragment look-ahead (steps) / ring size
    constexpr int PPS = (HPASS + NSUB - 3) / (NSUB - 2);      // halo pieces requested per sub-stage (stored two sub-stages later)
    static_assert(PPS * (NSUB - 2) >= HPASS, "every halo piece has a sub-stage to be requested in and one to be stored in");
    static_assert(WM == 1 || WM == 2, "one or two row groups of waves");
    static_assert(EPI != LU_EPI_LSTM || WM * TW * EX_LD * 4 <= 2 * AH_BYTES, "gate exchange aliases the two halo images");
    static_assert(4 * WM * 16 * 36 * 4 <= 2 * AH_BYTES, "bias-epilogue exchange (a 16 x 36 slice per wave) aliases the two halo images");
    LU_DYN_LDS(unsigned char, Ah);                      // [2][AH_BYTES]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = LU_UNIFORM(tid >> 6);              // wave-uniform: everything derived from it lives in scalar registers
    const int wm = wave >> 2, wn = wave & 3;
    int tile, nt, ks;
    if (!lu_block_tile(a, tile, nt, ks)) return;
    const int f = tile / a.tiles_pf;
    const int t2 = tile - f * a.tiles_pf;
    const int y0 = (t2 / a.tiles_x) * TH, x0 = (t2 % a.tiles_x) * TW;
    const int n0 = nt * BN;
    const lu_u4* const zp = reinterpret_cast<const lu_u4*>(lu_z16);
    const int q = tid % G;
    const int nfr = (a.N + 31) >> 5;
    // This wave's column fragment.  A fragment beyond N (a partial last column tile) is CLAMPED to the last real one instead of
    // being fed zeros: its accumulators are never stored (every epilogue path tests col < N), and the weight loads below stay
    // free of selects -- `frag_ok ? pointer : zeros` per load became an exec-masked branch per load in the instruction stream.
    const int frag_raw = (EPI == LU_EPI_LSTM) ? (wn * a.F + nt * 32) >> 5 : nt * 4 + wn;
    const int frag = frag_raw < nfr ? frag_raw : nfr - 1;
    const unsigned voff = (unsigned)lane * 16u;         // this lane's 16 bytes inside a 1 KB k-step of a fragment

    const int nch0 = a.src[0].nchunk, nch1 = a.n_src > 1 ? a.src[1].nchunk : 0;
    const bool ctr1 = a.src1_center != 0;
    const int n_full = nch0 + (ctr1 ? 0 : nch1);
    auto describe = [&](int ci) {
        ChunkDesc d;
        const int s_ = (ci >= nch0) ? 1 : 0;
        const int ch = ci - (s_ ? nch0 : 0);
        const int nch = s_ ? a.src[1].nchunk : nch0;
        d.x = reinterpret_cast<const unsigned char*>(a.src[s_].x) + (int64_t)f * a.src[s_].frame_stride * ESZ;
        // UNIFORM base of (tap 0, this chunk, this wave's fragment); the lane part travels as a 32-bit offset (saddr + voffset loads)
        d.w = reinterpret_cast<const unsigned char*>(a.src[s_].w) + ((int64_t)ch * nfr + frag) * 2048;
        d.wts = (s_ && ctr1) ? 0 : (int64_t)nch * nfr * 2048;      // (the centre-tap image holds ONE tap)
        d.ps = a.src[s_].pix_stride;
        d.C = a.src[s_].C;
        d.c0 = ch * CKS;
        d.single = (s_ && ctr1) ? 1 : 0;
        return d;
    };
    int cb = 0, ce = n_full;
    if (a.ksplit > 1) {
        const int per = (n_full + a.ksplit - 1) / a.ksplit;
        cb = ks * per < n_full ? ks * per : n_full;
        ce = cb + per < n_full ? cb + per : n_full;
    }

    auto piece_load = [&](int p, const ChunkDesc& d, lu_u4& r, bool want) {
        const int hp = (tid + NT * p) / G;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int iy = y0 + hy - PAD, ix = x0 + hx - PAD;
        const bool ok = want && hp < HP && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        const int c = d.c0 + PC * q;
        const int64_t off = ((int64_t)(iy * a.Win + ix) * d.ps + c) * ESZ;
        r = *((ok && c < d.C) ? reinterpret_cast<const lu_u4*>(d.x + off) : zp);
    };
    auto piece_store = [&](int p, int hb, const lu_u4& r) {
        const int hp = (tid + NT * p) / G;
        if (hp < HP) {
            if (B16) {
                *reinterpret_cast<lu_u4*>(&Ah[hb * AH_BYTES + hp * PITCH + 16 * q]) = r;
            } else {
                lu_u2 v;
                v.x = lu_pack2bf(lu_bits2f(r.x), lu_bits2f(r.y));
                v.y = lu_pack2bf(lu_bits2f(r.z), lu_bits2f(r.w));
                *reinterpret_cast<lu_u2*>(&Ah[hb * AH_BYTES + hp * PITCH + 8 * q]) = v;
            }
        }
    };
    // one channel half (k-step) of a weight fragment: 16 bytes per lane
    auto load_bh = [&](const ChunkDesc& d, int tap, int half, float4& b) {
        const unsigned char* wp = d.w + tap * d.wts + 1024 * half;      // scalar arithmetic
        b = *reinterpret_cast<const float4*>(wp + voff);
    };

    f32x16 acc[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // this lane's A-fragment address for (halo row 0 of its row group, kernel column 0, channel half 0)
    const int abase = (RW * wm * HWD + (lane & 31)) * PITCH + 16 * (lane >> 5);

    const bool have_center = ctr1 && ce == n_full;
    if (ce > cb || have_center) {
        ChunkDesc cur = describe(cb < ce ? cb : n_full);
        float4 rb0[K], rb1[K];                // the weight ring: slot kh = tap (kh, current kernel column), both channel halves
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            load_bh(cur, kh * K, 0, rb0[kh]);
            load_bh(cur, kh * K, 1, rb1[kh]);
        }
        {
            lu_u4 rh[HPASS];
#pragma unroll
            for (int p = 0; p < HPASS; ++p) piece_load(p, cur, rh[p], true);
#pragma unroll
            for (int p = 0; p < HPASS; ++p) piece_store(p, 0, rh[p]);
        }
        __syncthreads();
        int hb = 0;
        lu_u4 rp[2 * PPS];
#pragma unroll
        for (int i = 0; i < 2 * PPS; ++i) rp[i] = lu_u4{0u, 0u, 0u, 0u};
        lu_bf16x8 fr[NR];
        for (int ci = cb; ci < ce; ++ci) {
            const bool has_next = ci + 1 < ce || have_center;
            const ChunkDesc nxt = describe(has_next ? ci + 1 : ci);
            const unsigned char* const ab = &Ah[hb * AH_BYTES + abase];
            lu_static_for<NS>([&](auto sc) {
                constexpr int s_ = decltype(sc)::value;
                constexpr int u = s_ / NJ, j = s_ % NJ, kw = u / 2, half = u % 2;
                if (j == 0) {                 // sub-stage boundary: the next chunk's halo, PPS pieces at a time
#pragma unroll
                    for (int i = 0; i < PPS; ++i) {
                        const int ps_ = (u - 2) * PPS + i;      // requested two sub-stages ago: older than anything the MFMAs since waited for
                        if (u >= 2 && ps_ < HPASS) piece_store(ps_, hb ^ 1, rp[i + PPS * (u & 1)]);
                    }
#pragma unroll
                    for (int i = 0; i < PPS; ++i) {
                        const int pl_ = u * PPS + i;
                        if (u < NSUB - 2 && pl_ < HPASS) piece_load(pl_, nxt, rp[i + PPS * (u & 1)], has_next);
                    }
                }
                // A fragments run LA steps ahead of their MFMAs (cold only behind the chunk barrier: the image is complete there)
                if (s_ == 0) {
#pragma unroll
                    for (int t = 0; t < LA; ++t)
                        fr[t % NR] = *reinterpret_cast<const lu_bf16x8*>(ab + ((t % NJ) * HWD + 0) * PITCH);
                }
                if (s_ + LA < NS) {
                    constexpr int sn = s_ + LA < NS ? s_ + LA : 0;
                    constexpr int un = sn / NJ, jn = sn % NJ;
                    fr[sn % NR] = *reinterpret_cast<const lu_bf16x8*>(ab + (jn * HWD + un / 2) * PITCH + 32 * (un % 2));
                }
                // halo row j of this kernel column feeds patch row j - kh through tap (kh, kw)
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    if (j - kh >= 0 && j - kh < RW) {
                        const lu_bf16x8 bv = __builtin_bit_cast(lu_bf16x8, half ? rb1[kh] : rb0[kh]);
                        acc[j - kh] = lu_mfma_bf16(fr[s_ % NR], bv, acc[j - kh]);
                    }
                }
                // slot kh = j - (RW - 1) has seen its last product of this sub-stage: refill this half with the next kernel column's
                if (j >= RW - 1) {
                    constexpr int kh = j - (RW - 1) < K ? j - (RW - 1) : 0;
                    if (kw + 1 < K) load_bh(cur, kh * K + kw + 1, half, half ? rb1[kh] : rb0[kh]);
                    else load_bh(nxt, kh * K, half, half ? rb1[kh] : rb0[kh]);
                }
                LU_SCHED_FENCE();
            });
            __syncthreads();                     // the next halo is complete and every wave is done with the old one
            hb ^= 1;
            cur = nxt;
        }
        if (have_center) {                       // one more stage: the centre tap of the im2col chunk (ring slot 0 holds its fragments)
            const unsigned char* const ab = &Ah[hb * AH_BYTES + abase + (PAD * HWD + PAD) * PITCH];
            const lu_bf16x8 bv0 = __builtin_bit_cast(lu_bf16x8, rb0[0]), bv1 = __builtin_bit_cast(lu_bf16x8, rb1[0]);
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                const lu_bf16x8 a0 = *reinterpret_cast<const lu_bf16x8*>(ab + i * HWD * PITCH);
                const lu_bf16x8 a1 = *reinterpret_cast<const lu_bf16x8*>(ab + i * HWD * PITCH + 32);
                acc[i] = lu_mfma_bf16(a0, bv0, acc[i]);
                acc[i] = lu_mfma_bf16(a1, bv1, acc[i]);
            }
        }
    }
    frag_epilogue<EPI, RW>(a, acc, Ah, f, y0, x0, nt, n0, ks);
}

// ---------------------------------------------------------------------------------------------------------
// Input gradient of a stride-2 3x3 convolution (first layer of a down block; TF-SAME on even extents: pad 0 before, 1
// after), bf16 MFMA operands, ALL FOUR output parity classes in one pass over dy.
//   dx[2a+py, 2b+px, c] = sum over the taps (kh, kw) with kh = py (mod 2), kw = px (mod 2) of dy[a - (kh - py)/2, b - (kw - px)/2, n] w[kh, kw, c, n]
// i.e. 4 + 2 + 2 + 1 = 9 (tap, class) products over a 2 x 2 window of dy -- the same dy tile feeds all of them.  Round 2
// launched four gather convolutions (one per class): blocks of 4-16 k-steps, bound by their prologue / epilogue at
// ~100 TFLOP/s.  Here a block of 4 waves owns a 2 x 32-pixel tile of dy-space x 128 input channels and keeps the four
// classes' accumulators (4 x 2 rows x one 32-column fragment per wave = 128 VGPRs); per 32-channel chunk of dy the 3 x 33
// pixel halo is staged once in LDS (bf16, 80-byte pitch as in the halo kernels) and the nine products read shifted rows of
// it; weights come from L2 in MFMA-fragment order (lu_stride2_dgrad_weights -> lu_pack_weights_taps_bf16: the nine tap
// matrices in class order).  The next chunk's halo is fetched while the current one is multiplied.
// ---------------------------------------------------------------------------------------------------------
struct S2DgradArgs {
    const float* dy;
    const unsigned char* w;      // packed taps: [9][Nf / 32 chunks][ceil(C / 32) fragments][2 KB]
    float* dx;
    int64_t dy_fs;
    int32_t dy_ps, Hd, Wd, Nf, C;
    int32_t tiles_x, tiles_pf, m_tiles, n_tiles;
};

__global__ __launch_bounds__(256, 2) void conv_s2_dgrad_bf16_kernel(S2DgradArgs a) {
    constexpr int NT = 256, RW = 2, TW = 32, HWD = TW + 1, HHT = RW + 1, HP = HHT * HWD, PITCH = 80;
    constexpr int G = 8;                                    // 16-byte pieces (4 fp32 channels) per pixel and chunk
    constexpr int HPASS = (HP * G + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) unsigned char Ah[2][HP * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    // block -> (dy tile, column tile): the column tiles of one dy tile are neighbours on one XCD (block b runs on XCD b % 8)
    const int slot = blockIdx.x >> 3;
    const int nt = slot % a.n_tiles;
    const int tile = (slot / a.n_tiles) * 8 + (blockIdx.x & 7);
    if (tile >= a.m_tiles) return;
    const int f = tile / a.tiles_pf;
    const int t2 = tile - f * a.tiles_pf;
    const int y0 = (t2 / a.tiles_x) * RW, x0 = (t2 % a.tiles_x) * TW;
    const int nfr = (a.C + 31) >> 5, nch = a.Nf >> 5;
    const int frag = nt * 4 + wn;
    const bool frag_ok = frag < nfr;
    const lu_u4* const zp = reinterpret_cast<const lu_u4*>(lu_zero16);
    const float* const dyf = a.dy + (int64_t)f * a.dy_fs;
    const unsigned char* const wl = a.w + (int64_t)frag * 2048 + lane * 16;
    const int q = tid % G;

    auto piece_load = [&](int p, int chunk, lu_u4& r) {
        const int hp = (tid + NT * p) / G;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool ok = hp < HP && iy >= 0 && iy < a.Hd && ix >= 0 && ix < a.Wd;
        const int64_t off = (int64_t)(iy * a.Wd + ix) * a.dy_ps + chunk * 32 + 4 * q;
        r = *(ok ? reinterpret_cast<const lu_u4*>(dyf + off) : zp);
    };
    auto piece_store = [&](int p, int hb, const lu_u4& r) {
        const int hp = (tid + NT * p) / G;
        if (hp < HP) {
            lu_u2 v;
            v.x = lu_pack2bf(lu_bits2f(r.x), lu_bits2f(r.y));
            v.y = lu_pack2bf(lu_bits2f(r.z), lu_bits2f(r.w));
            *reinterpret_cast<lu_u2*>(&Ah[hb][hp * PITCH + 8 * q]) = v;
        }
    };
    auto load_b = [&](int tap, int chunk, float4& b0, float4& b1) {
        const unsigned char* wp = wl + ((int64_t)tap * nch + chunk) * nfr * 2048;
        b0 = *reinterpret_cast<const float4*>(frag_ok ? reinterpret_cast<const float*>(wp) : lu_zero16);
        b1 = *reinterpret_cast<const float4*>(frag_ok ? reinterpret_cast<const float*>(wp + 1024) : lu_zero16);
    };

    f32x16 acc[4][RW];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.f;

    {
        lu_u4 rh[HPASS];
#pragma unroll
        for (int p = 0; p < HPASS; ++p) piece_load(p, 0, rh[p]);
#pragma unroll
        for (int p = 0; p < HPASS; ++p) piece_store(p, 0, rh[p]);
    }
    __syncthreads();
    const int khalf16 = 16 * (lane >> 5);
    int hb = 0;
    float4 b0, b1, c0, c1;      // the weight fragments of the current tap and of the next one (two taps in flight)
    load_b(0, 0, b0, b1);
    load_b(1, 0, c0, c1);
    for (int chunk = 0; chunk < nch; ++chunk) {
        const bool more = chunk + 1 < nch;
        lu_u4 rn[HPASS];
#pragma unroll
        for (int p = 0; p < HPASS; ++p) piece_load(p, more ? chunk + 1 : chunk, rn[p]);      // (last chunk: re-read, unused)
        // tap t: class, dy offset (row, column) -- the class order of lu_stride2_dgrad_weights for k = 3, pads 0
        constexpr int T_CLS[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
        constexpr int T_DY[9] = {-1, -1, 0, 0, -1, 0, 0, 0, 0};
        constexpr int T_DX[9] = {-1, 0, -1, 0, 0, 0, -1, 0, 0};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float4 n0v, n1v;
            if (t + 2 < 9) load_b(t + 2, chunk, n0v, n1v);
            else load_b(t + 2 - 9, more ? chunk + 1 : chunk, n0v, n1v);
            const lu_bf16x8 bv0 = __builtin_bit_cast(lu_bf16x8, b0), bv1 = __builtin_bit_cast(lu_bf16x8, b1);
            // halo row hy = (output row i) + 1 + dy, halo column = pixel + 1 + dx
            const unsigned char* ab = &Ah[hb][((1 + T_DY[t]) * HWD + (lane & 31) + 1 + T_DX[t]) * PITCH + khalf16];
            lu_bf16x8 a0[RW], a1[RW];
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                a0[i] = *reinterpret_cast<const lu_bf16x8*>(ab + i * HWD * PITCH);
                a1[i] = *reinterpret_cast<const lu_bf16x8*>(ab + i * HWD * PITCH + 32);
            }
#pragma unroll
            for (int i = 0; i < RW; ++i) acc[T_CLS[t]][i] = lu_mfma_bf16(a0[i], bv0, acc[T_CLS[t]][i]);
#pragma unroll
            for (int i = 0; i < RW; ++i) acc[T_CLS[t]][i] = lu_mfma_bf16(a1[i], bv1, acc[T_CLS[t]][i]);
            b0 = c0;
            b1 = c1;
            c0 = n0v;
            c1 = n1v;
        }
        if (more) {
#pragma unroll
            for (int p = 0; p < HPASS; ++p) piece_store(p, hb ^ 1, rn[p]);
        }
        __syncthreads();
        hb ^= 1;
    }
    // epilogue: class (py, px) of dy-space pixel (a, b) is dx[2a + py, 2b + px]
    const int Hin = 2 * a.Hd, Win = 2 * a.Wd;
    const int col = frag * 32 + (lane & 31);
    if (!frag_ok || col >= a.C) return;
    float* const dxf = a.dx + (int64_t)f * Hin * Win * a.C + col;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int ay = y0 + i;
            if (ay >= a.Hd) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int bx = x0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (bx >= a.Wd) continue;
                dxf[((int64_t)(2 * ay + py) * Win + 2 * bx + px) * a.C] = acc[c][i][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stride-2 3x3 FORWARD convolution (first layer of a down block, TF-SAME on even extents: pad 0 before, 1 after) on bf16 MFMA
// operands, reading the bf16 copy of the ConvLSTM output that the bf16 tape keeps anyway.  The gather kernel ran these at
// twice their HBM time on the fp32 tensor (0.57 ms for 0.8 GB at level 0).  Block = 4 output rows x 32 output pixels x 128
// columns (8 waves: 2 row pairs x 4 column fragments); the 9 x 65 input pixels under the tile are staged once per 32-channel
// chunk, SPLIT BY COLUMN PARITY ([row][33 even | 32 odd] pixels, 80-byte pitch): tap kw then reads 32 consecutive entries --
// even[ox] / odd[ox] / even[ox + 1] -- i.e. the same conflict-free ds_read_b128 fragments as the stride-1 kernels.  Weights in
// MFMA-fragment order from L2 (lu_pack_weights_bf16); the next chunk's pixels are in flight while the current one is multiplied.
// ---------------------------------------------------------------------------------------------------------
struct S2FwdArgs {
    const unsigned short* x;      // bf16 [frames, Hin, Win, C]
    const unsigned char* w;       // packed [9][ceil(C / 32)][ceil(N / 32)][2 KB]
    const float* bias;
    float* out;                   // fp32 dense [frames, Hin / 2, Win / 2, N]
    int64_t x_fs;
    int32_t x_ps, Hin, Win, C, N;
    int32_t tiles_x, tiles_pf, m_tiles, n_tiles;
};

__global__ __launch_bounds__(512, 2) void conv_s2_fwd_bf16_kernel(S2FwdArgs a) {
    constexpr int NT = 512, R = 4, RW = 2, TW = 32, HR = 2 * R + 1, HC = 2 * TW + 1, HP = HR * HC, PITCH = 80;
    constexpr int G = 4;                                    // 16-byte pieces (8 bf16 channels) per pixel and chunk
    constexpr int HPASS = (HP * G + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) unsigned char Ah[HP * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int slot = blockIdx.x >> 3;
    const int nt = slot % a.n_tiles;
    const int tile = (slot / a.n_tiles) * 8 + (blockIdx.x & 7);
    if (tile >= a.m_tiles) return;
    const int Hout = a.Hin >> 1, Wout = a.Win >> 1;
    const int f = tile / a.tiles_pf;
    const int t2 = tile - f * a.tiles_pf;
    const int y0 = (t2 / a.tiles_x) * R, x0 = (t2 % a.tiles_x) * TW;
    const int nfr = (a.N + 31) >> 5, nch = (a.C + 31) >> 5;
    const int frag = nt * 4 + wn;
    const bool frag_ok = frag < nfr;
    const lu_u4* const zp = reinterpret_cast<const lu_u4*>(lu_zero16);
    const unsigned short* const xf = a.x + (int64_t)f * a.x_fs;
    const unsigned char* const wl = a.w + (int64_t)frag * 2048 + lane * 16;
    const int q = tid % G;

    auto piece_load = [&](int p, int chunk, lu_u4& r) {
        const int hp = (tid + NT * p) / G;
        const int hr = hp / HC, hc = hp - hr * HC;
        const int iy = 2 * y0 + hr, ix = 2 * x0 + hc;
        const int c = chunk * 32 + 8 * q;
        const bool ok = hp < HP && iy < a.Hin && ix < a.Win && c < a.C;
        r = *(ok ? reinterpret_cast<const lu_u4*>(xf + (int64_t)(iy * a.Win + ix) * a.x_ps + c) : zp);
    };
    auto piece_store = [&](int p, const lu_u4& r) {
        const int hp = (tid + NT * p) / G;
        if (hp < HP) {
            const int hr = hp / HC, hc = hp - hr * HC;
            const int slot_ = hr * HC + ((hc & 1) ? 33 + (hc >> 1) : (hc >> 1));      // [33 even | 32 odd] columns
            *reinterpret_cast<lu_u4*>(&Ah[slot_ * PITCH + 16 * q]) = r;
        }
    };
    auto load_b = [&](int tap, int chunk, float4& b0, float4& b1) {
        const unsigned char* wp = wl + ((int64_t)tap * nch + chunk) * nfr * 2048;
        b0 = *reinterpret_cast<const float4*>(frag_ok ? reinterpret_cast<const float*>(wp) : lu_zero16);
        b1 = *reinterpret_cast<const float4*>(frag_ok ? reinterpret_cast<const float*>(wp + 1024) : lu_zero16);
    };

    f32x16 acc[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    {
        lu_u4 rh[HPASS];
#pragma unroll
        for (int p = 0; p < HPASS; ++p) piece_load(p, 0, rh[p]);
#pragma unroll
        for (int p = 0; p < HPASS; ++p) piece_store(p, rh[p]);
    }
    __syncthreads();
    const int khalf16 = 16 * (lane >> 5);
    float4 b0, b1, c0, c1;
    load_b(0, 0, b0, b1);
    load_b(1, 0, c0, c1);
    for (int chunk = 0; chunk < nch; ++chunk) {
        const bool more = chunk + 1 < nch;
        lu_u4 rn[HPASS];
#pragma unroll
        for (int p = 0; p < HPASS; ++p) piece_load(p, more ? chunk + 1 : chunk, rn[p]);      // (last chunk: re-read, unused)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t - 3 * kh;
            float4 n0v, n1v;      // fragments of the tap after next (two taps of weights in flight)
            if (t + 2 < 9) load_b(t + 2, chunk, n0v, n1v);
            else load_b(t + 2 - 9, more ? chunk + 1 : chunk, n0v, n1v);
            const lu_bf16x8 bv0 = __builtin_bit_cast(lu_bf16x8, b0), bv1 = __builtin_bit_cast(lu_bf16x8, b1);
            const int col = (kw == 1 ? 33 : 0) + (kw == 2 ? 1 : 0) + (lane & 31);
            const unsigned char* ab = &Ah[((2 * RW * wm + kh) * HC + col) * PITCH + khalf16];
            lu_bf16x8 a0[RW], a1[RW];
#pragma unroll
            for (int i = 0; i < RW; ++i) {      // output row RW wm + i reads input rows 2 (RW wm + i) + kh
                a0[i] = *reinterpret_cast<const lu_bf16x8*>(ab + 2 * i * HC * PITCH);
                a1[i] = *reinterpret_cast<const lu_bf16x8*>(ab + 2 * i * HC * PITCH + 32);
            }
#pragma unroll
            for (int i = 0; i < RW; ++i) acc[i] = lu_mfma_bf16(a0[i], bv0, acc[i]);
#pragma unroll
            for (int i = 0; i < RW; ++i) acc[i] = lu_mfma_bf16(a1[i], bv1, acc[i]);
            b0 = c0;
            b1 = c1;
            c0 = n0v;
            c1 = n1v;
        }
        __syncthreads();                      // every wave is done with this chunk's pixels
        if (more) {
#pragma unroll
            for (int p = 0; p < HPASS; ++p) piece_store(p, rn[p]);
        }
        __syncthreads();
    }
    const int col = frag * 32 + (lane & 31);
    if (!frag_ok || col >= a.N) return;
    const float bv = a.bias ? a.bias[col] : 0.f;
    float* const of = a.out + (int64_t)f * Hout * Wout * a.N + col;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int oy = y0 + RW * wm + i;
        if (oy >= Hout) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (ox >= Wout) continue;
            of[((int64_t)oy * Wout + ox) * a.N] = acc[i][r] + bv;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// General bf16-MFMA convolution (precision = 1 where the halo kernel does not apply): stride 1 / 2, any k <= 7, any
// pads, narrow outputs, strided output rows (parity planes of a stride-2 input gradient).  Implicit GEMM over
// 256 linear pixels x 128 columns per block; a stage is one (tap, 32-channel chunk): the [256][32] activation slab is
// gathered from HBM/L2 (fp32 -> bf16 while staged, two LDS buffers, one barrier per stage), the weights come from L2 in
// MFMA-fragment order exactly as in conv_halo_frag_kernel.  Waves: 2 (groups of 128 pixels) x 4 (column fragments).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void conv_gather_bf16_kernel(ConvArgs a) {
    const float* const lu_z16 = a.zero16 ? a.zero16 : lu_zero16;
    constexpr int NT = 512, RA = 4, MFW = 4;           // RA pixel rows gathered per thread; MFW 32-pixel fragments per wave
    __shared__ __attribute__((aligned(16))) unsigned short As[2][BM * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    int mt, nt, ks;
    if (!lu_block_tile(a, mt, nt, ks)) return;
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * 128;
    const float* const zp = lu_z16;
    const int nfr = (a.N + 31) >> 5;
    const int frag = nt * 4 + wn;
    const bool frag_ok = frag < nfr;

    // gather bookkeeping: thread -> 16-byte channel group q of pixel rows (tid >> 3) + 64 i
    const int q = tid & 7;
    int fr[RA], vy0[RA], vx0[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int64_t m = m0 + (tid >> 3) + 64 * i;
        if (m < a.M) {
            const int f = (int)(m / a.HWo);
            const int r = (int)(m - (int64_t)f * a.HWo);
            const int oy = r / a.Wout, ox = r - oy * a.Wout;
            fr[i] = f;
            vy0[i] = oy * a.stride - a.pad_t;
            vx0[i] = ox * a.stride - a.pad_l;
        } else {
            fr[i] = 0;
            vy0[i] = vx0[i] = -(1 << 28);
        }
    }
    const float* const x_s0 = a.src[0].x;
    const float* const x_s1 = a.src[1].x;
    const int64_t fs_s0 = a.src[0].frame_stride, fs_s1 = a.src[1].frame_stride;
    const unsigned short* const w_s0 = reinterpret_cast<const unsigned short*>(a.src[0].w) + (int64_t)frag * 1024 + lane * 8;
    const unsigned short* const w_s1 = reinterpret_cast<const unsigned short*>(a.src[1].w) + (int64_t)frag * 1024 + lane * 8;
    const int ps_s0 = a.src[0].pix_stride, ps_s1 = a.src[1].pix_stride;
    const int C_s0 = a.src[0].C, C_s1 = a.src[1].C;
    const int nch_s0 = a.src[0].nchunk, nch_s1 = a.src[1].nchunk;
    const int K = a.k, kk = a.kk;
    auto tap_advance = [&](IterState& st) {
        ++st.tap;
        if (++st.kw == K) {
            st.kw = 0;
            ++st.kh;
        }
        if (st.tap < kk) return;
        st.tap = st.kh = st.kw = 0;
        if (++st.chunk == (st.s ? nch_s1 : nch_s0)) {
            st.chunk = 0;
            ++st.s;
        }
    };
    constexpr int D = 4;           // register rings: activation slabs and B fragments are requested D - 1 / D stages ahead
    float4 ra[D][RA];
    auto load_a = [&](const IterState& st, float4 (&r)[RA]) {
        const int c = st.chunk * CKB + 4 * q;
        const bool cok = c < (st.s ? C_s1 : C_s0);
        const float* xb = (st.s ? x_s1 : x_s0) + c;
        const int64_t fs = st.s ? fs_s1 : fs_s0;
        const int ps = st.s ? ps_s1 : ps_s0;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int iy = vy0[i] + st.kh, ix = vx0[i] + st.kw;
            const bool ok = cok && iy >= 0 && ix >= 0 && iy < a.Hin && ix < a.Win;
            const float* p = xb + (int64_t)fr[i] * fs + ((int64_t)iy * a.Win + ix) * ps;
            r[i] = *reinterpret_cast<const float4*>(ok ? p : zp);
        }
    };
    auto store_a = [&](int buf, const float4 (&r)[RA]) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            unsigned* dst = reinterpret_cast<unsigned*>(&As[buf][((tid >> 3) + 64 * i) * LDB + 4 * q]);
            dst[0] = lu_pack2bf(r[i].x, r[i].y);
            dst[1] = lu_pack2bf(r[i].z, r[i].w);
        }
    };
    auto load_b = [&](const IterState& st, float4& b0, float4& b1) {
        const unsigned short* wp = (st.s ? w_s1 : w_s0) + ((int64_t)st.tap * (st.s ? nch_s1 : nch_s0) + st.chunk) * nfr * 1024;
        b0 = *reinterpret_cast<const float4*>(frag_ok ? reinterpret_cast<const float*>(wp) : zp);
        b1 = *reinterpret_cast<const float4*>(frag_ok ? reinterpret_cast<const float*>(wp + 512) : zp);
    };

    f32x16 acc[MFW];
#pragma unroll
    for (int i = 0; i < MFW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    int it0 = 0, it1 = a.n_it;
    if (a.ksplit > 1) {
        const int per = (a.n_it + a.ksplit - 1) / a.ksplit;
        it0 = ks * per < a.n_it ? ks * per : a.n_it;
        it1 = it0 + per < a.n_it ? it0 + per : a.n_it;
    }
    if (it1 > it0) {
        IterState sS[D];               // sS[j]: state of the stage whose data lives in ring slot j
        {
            IterState st{0, 0, 0, 0, 0};
            int r = it0;
            if (r >= nch_s0 * kk) {
                r -= nch_s0 * kk;
                st.s = 1;
            }
            st.chunk = r / kk;
            st.tap = r - st.chunk * kk;
            st.kh = st.tap / K;
            st.kw = st.tap - st.kh * K;
            sS[0] = st;
        }
#pragma unroll
        for (int j = 1; j < D; ++j) {
            sS[j] = sS[j - 1];
            if (it0 + j < it1) tap_advance(sS[j]);       // (clamped at the last stage)
        }
        float4 rb0[D], rb1[D];
#pragma unroll
        for (int j = 0; j < D; ++j) load_b(sS[j], rb0[j], rb1[j]);
#pragma unroll
        for (int j = 0; j < D - 1; ++j) load_a(sS[j], ra[j]);
        store_a(0, ra[0]);
        __syncthreads();
        const int khalf8 = 8 * (lane >> 5);
        // Stage `it` in ring slot s: LDS buffer (it - it0) & 1 holds its slab, ra[s + 1] the next one (requested two stages ago);
        // request the slab of stage it + D - 1 into the slot this stage's slab came from, run the MFMAs, publish the next
        // slab, refill the B slot for stage it + D.  No conditional step in the loop body (exact vmcnt waits, see above).
        auto stage = [&](int it, auto slot) {
            constexpr int s = decltype(slot)::value;
            const IterState sa = sS[(s + D - 1) % D];     // state of stage it + D - 1 (clamped at the last stage)
            load_a(sa, ra[(s + D - 1) % D]);
            LU_SCHED_FENCE();
            const unsigned short* ab = &As[(it - it0) & 1][(128 * wm + (lane & 31)) * LDB + khalf8];
            const lu_bf16x8 bv0 = __builtin_bit_cast(lu_bf16x8, rb0[s]), bv1 = __builtin_bit_cast(lu_bf16x8, rb1[s]);
#pragma unroll
            for (int i = 0; i < MFW; ++i) {
                const lu_bf16x8 a0 = *reinterpret_cast<const lu_bf16x8*>(ab + 32 * i * LDB);
                acc[i] = lu_mfma_bf16(a0, bv0, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < MFW; ++i) {
                const lu_bf16x8 a1 = *reinterpret_cast<const lu_bf16x8*>(ab + 32 * i * LDB + 16);
                acc[i] = lu_mfma_bf16(a1, bv1, acc[i]);
            }
            LU_SCHED_FENCE();
            store_a((it - it0 + 1) & 1, ra[(s + 1) % D]);       // that buffer was last read before the previous barrier
            sS[s] = sa;                                   // state of stage it + D - 1 ...
            if (it + D < it1) tap_advance(sS[s]);         // ... + 1: the stage this slot serves next
            load_b(sS[s], rb0[s], rb1[s]);
            __syncthreads();
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>;
        using S3 = std::integral_constant<int, 3>;
        int it = it0;
        for (; it + D <= it1; it += D) {
            stage(it, S0());
            stage(it + 1, S1());
            stage(it + 2, S2());
            stage(it + 3, S3());
        }
        if (it < it1) stage(it, S0());
        if (it + 1 < it1) stage(it + 1, S1());
        if (it + 2 < it1) stage(it + 2, S2());
    }

#pragma unroll
    for (int i = 0; i < MFW; ++i) {
        const int64_t mb = m0 + 128 * wm + 32 * i + 4 * (lane >> 5);
        RowCursor rc;
        rc.set(a, mb < a.M ? mb : 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r) rc.advance(a, (r & 3) ? 1 : 5);
            const int64_t m = mb + (r & 3) + 8 * (r >> 2);
            if (m >= a.M) continue;
            float v[1] = {acc[i][r]};
            conv_epilogue_row<1, LU_EPI_BIAS>(a, v, rc.f, rc.pix, m, nt, n0 + 32 * wn, ks, lane & 31, rc.oy);
        }
    }
}

// out[m, n] = bias[n] + sum_s ws[s][m][n]   (fixed order: deterministic), optionally through the post affine + LeakyReLU
__global__ void ksplit_reduce_kernel(const float* __restrict__ ws, int ksplit, int64_t M, int N, int HWo,
                                     const float* __restrict__ bias, float* __restrict__ out, int64_t out_fs,
                                     int out_ps, int Wout, int64_t out_rs, const float* __restrict__ post_scale,
                                     const float* __restrict__ post_shift, float post_alpha) {
    const int64_t total = M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / N;
        const int n = (int)(i - m * N);
        float s = bias ? bias[n] : 0.f;
        for (int k = 0; k < ksplit; ++k) s += ws[(int64_t)k * total + i];
        if (post_scale) {
            s = fmaf(s, post_scale[n], post_shift[n]);
            s = s > 0.f ? s : post_alpha * s;
        }
        const int64_t f = m / HWo;
        const int64_t pix = m - f * HWo;
        if (out_rs) {
            const int64_t oy = pix / Wout;
            out[f * out_fs + oy * out_rs + (pix - oy * Wout) * out_ps + n] = s;
        } else {
            out[f * out_fs + pix * out_ps + n] = s;
        }
    }
}

// out[m, n] = lrelu(scale[n] * out[m, n] + shift[n]) in place on a dense [M, N] output (post affine of an unsplit launch)
__global__ void post_affine_kernel(float* __restrict__ out, int64_t total, int N, const float* __restrict__ scale,
                                   const float* __restrict__ shift, float alpha) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const float t = fmaf(out[i], scale[n], shift[n]);
        out[i] = t > 0.f ? t : alpha * t;
    }
}

// wt[kh'][kw'][co][ci] = w[k-1-kh'][k-1-kw'][c_off+ci][co]
// one block per (tap, 32x32 tile); LDS transpose for coalescing on both sides.  256 threads.
__device__ __forceinline__ void flip_transpose_body(const float* __restrict__ w, float* __restrict__ wt, int k, int C_tot, int N,
                                                    int c_off, int C_sub, int tap, int ci0, int co0, float (*tile)[33]) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
    const int src_tap = k * k - 1 - tap;
    for (int r = ty; r < 32; r += 8) {
        int ci = ci0 + r, co = co0 + tx;
        tile[r][tx] = (ci < C_sub && co < N) ? w[((int64_t)src_tap * C_tot + c_off + ci) * N + co] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int co = co0 + r, ci = ci0 + tx;
        if (co < N && ci < C_sub) wt[((int64_t)tap * N + co) * C_sub + ci] = tile[tx][r];
    }
}

__global__ void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int k, int C_tot, int N,
                                      int c_off, int C_sub) {
    __shared__ float tile[32][33];
    flip_transpose_body(w, wt, k, C_tot, N, c_off, C_sub, blockIdx.z, blockIdx.y * 32, blockIdx.x * 32, tile);
}

// Every derived weight image of a training step in TWO launches (lu_weight_prep_batch): the optimiser changes all
// parameters at once, and ~80 separate 8-10 us flips / packs per step were 0.7 ms of a 75 ms bf16 step.  A block finds its
// operation by bisection over the (block-offset sorted) table in device memory and runs the single-operation body on it.
struct PrepOp {              // = lu_prep_op (include/lstm_unet_hip.h)
    int32_t kind;            // 0: flip_transpose, 1: pack_weights_bf16
    int32_t blk0, nblk;      // this operation's blocks: [blk0, blk0 + nblk)
    int32_t k;               // flip: kernel size
    const float* src;
    void* dst;
    int64_t tap_stride;      // pack: elements between taps / channel rows of src
    int32_t row_stride, kk;  // pack: taps
    int32_t C, N;            // pack: channels, columns.  flip: C_sub, N
    int32_t C_tot, c_off;    // flip
};

__global__ __launch_bounds__(256) void weight_prep_batch_kernel(const PrepOp* __restrict__ ops, int n_ops) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n_ops - 1;
    const int b = blockIdx.x;
    while (lo < hi) {        // last operation with blk0 <= b (uniform per block)
        const int mid = (lo + hi + 1) >> 1;
        if (ops[mid].blk0 <= b) lo = mid;
        else hi = mid - 1;
    }
    const PrepOp op = ops[lo];
    const int lb = b - op.blk0;
    if (lb >= op.nblk) return;
    if (op.kind == 0) {
        const int nx = (op.N + 31) / 32, ny = (op.C + 31) / 32;
        const int tap = lb / (nx * ny), r = lb - tap * nx * ny;
        flip_transpose_body(op.src, (float*)op.dst, op.k, op.C_tot, op.N, op.c_off, op.C, tap, (r / nx) * 32, (r % nx) * 32, tile);
    } else {
        pack_weights_bf16_body(op.src, op.tap_stride, op.row_stride, op.kk, op.C, op.N, (unsigned short*)op.dst,
                               (int64_t)lb * 256 + threadIdx.x, (int64_t)op.nblk * 256);
    }
}

// Sub-kernels of the input gradient of a stride-2 convolution, one per output parity class (py, px):
//   dX[2a+py, 2b+px, c] = sum_{ty,tx,n} dY[a + ty - pad_y, b + tx - pad_x, n] * sub[cls][ty][tx][n][c]
// with sub[cls][ty][tx][n][c] = w[kh][kw][c][n], kh = py + pt - 2*(ty - pad_y) (zero when outside [0,k)).
__global__ void s2_dgrad_weights_kernel(const float* __restrict__ w, float* __restrict__ sub, int k, int C, int N, int pt,
                                        int pl, int pady0, int pady1, int padx0, int padx1, int ny0, int ny1, int nx0,
                                        int nx1) {
    // compact layout: plane cls = 2 py + px holds ny[py] x nx[px] taps, [ty][tx][n][c], planes back to back
    const int64_t nc = (int64_t)N * C;
    const int64_t sz[4] = {ny0 * nx0 * nc, ny0 * nx1 * nc, ny1 * nx0 * nc, ny1 * nx1 * nc};
    const int64_t total = sz[0] + sz[1] + sz[2] + sz[3];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cls = 0;
        int64_t j = i;
        while (j >= sz[cls]) j -= sz[cls++];
        const int py = cls >> 1, px = cls & 1;
        const int nx = px ? nx1 : nx0;
        const int c = (int)(j % C);
        int64_t t = j / C;
        const int n = (int)(t % N);
        t /= N;
        const int tx = (int)(t % nx);
        const int ty = (int)(t / nx);
        const int kh = py + pt - 2 * (ty - (py ? pady1 : pady0));
        const int kw = px + pl - 2 * (tx - (px ? padx1 : padx0));
        sub[i] = (kh >= 0 && kh < k && kw >= 0 && kw < k) ? w[(((int64_t)kh * k + kw) * C + c) * N + n] : 0.f;
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// non-empty slices when `chunks` chunks are dealt out ceil(chunks / splits) at a time
int lu_conv_chunk_splits(int chunks, int splits) {
    if (chunks <= 0 || splits <= 0) return 0;
    const int per = (chunks + splits - 1) / splits;
    return (chunks + per - 1) / per;
}

// device address of this translation unit's lu_zero16 on the current device (looked up once per device; null on failure: the kernels then take the symbol itself)
const void* conv_zero16_address() {
#ifdef LU_EMU
    return lu_zero16;
#else
    return LU_SYMBOL_ADDRESS(lu_zero16);      // per device (a process-wide cache handed device 0's address to device 1)
#endif
}

// dynamic LDS of conv_halo_frag_kernel<K, *, RW, *>: two halo images of 80 bytes per pixel
size_t halo_bf16_lds(int K, int RW) { return (size_t)2 * (2 * RW + K - 1) * (32 + K - 1) * LDB * sizeof(unsigned short); }
// (narrow blocks, NFR = 1 / 2: 8 / NFR row groups of RW = NFR rows -- the 8-row patch, same bytes as RW = 4)

}  // namespace

extern "C" int lu_conv2d_fwd(const lu_conv_desc* d, lu_stream_t stream) {
    LU_REQUIRE(d, "lu_conv2d_fwd: null descriptor");
    LU_REQUIRE(d->n_src == 1 || d->n_src == 2, "lu_conv2d_fwd: n_src must be 1 or 2 (got %d)", d->n_src);
    LU_REQUIRE(d->k >= 1 && d->k <= 7, "lu_conv2d_fwd: unsupported kernel size %d", d->k);
    LU_REQUIRE((d->stride == 1 || d->stride == 2) && (d->dil == 1 || d->dil == 2) && !(d->stride == 2 && d->dil == 2),
               "lu_conv2d_fwd: unsupported stride/dil %d/%d", d->stride, d->dil);
    LU_REQUIRE(d->frames > 0 && d->Hout > 0 && d->Wout > 0 && d->N > 0, "lu_conv2d_fwd: empty problem");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.zero16 = reinterpret_cast<const float*>(conv_zero16_address());
    const int k_h = d->k_h ? d->k_h : d->k;      // rectangular tap window k_h x k (parity planes of stride-2 gradients)
    LU_REQUIRE(k_h >= 1 && k_h <= 7, "lu_conv2d_fwd: unsupported kernel height %d", k_h);
    a.k = d->k;                                   // taps per kernel row
    a.kk = k_h * d->k;
    bool bvec = (d->N % 4 == 0);
    a.n_it = 0;
    a.n_src = 0;
    a.n_thin = 0;
    for (int s = 0; s < d->n_src; ++s) {
        const lu_conv_src& in = d->src[s];
        LU_REQUIRE(in.x && in.w && in.C > 0, "lu_conv2d_fwd: source %d incomplete", s);
        const bool vec = (in.C % 4 == 0) && (in.pix_stride % 4 == 0) && (in.frame_stride % 4 == 0) && aligned16(in.x);
        SrcInfo& si = (vec || d->precision != 0) ? a.src[a.n_src++] : a.tsrc[a.n_thin++];
        si.x = in.x;
        si.w = in.w;
        si.frame_stride = in.frame_stride;
        si.w_tap_stride = in.w_tap_stride;
        si.pix_stride = in.pix_stride;
        si.C = in.C;
        si.w_row_stride = in.w_row_stride;
        si.thin = vec ? 0 : 1;
        si.bf16 = in.dtype == LU_BF16 ? 1 : 0;
        LU_REQUIRE(in.dtype == LU_F32 || (in.dtype == LU_BF16 && d->precision == 1),
                   "lu_conv2d_fwd: source %d: bf16 activations need precision 1 (dtype %d)", s, in.dtype);
        si.nchunk = vec ? (in.C + CK - 1) / CK : (a.kk * in.C + CK - 1) / CK;
        if (d->precision != 0) {     // fragment-packed weights: bf16 (32-channel chunks) or fp32 (16-channel chunks)
            LU_REQUIRE(vec, "lu_conv2d_fwd: packed-weight modes need 16-byte aligned sources with C %% 4 == 0 (source %d); "
                            "pad thin inputs with zero channels", s);
            si.nchunk = d->precision == 1 ? (in.C + CKB - 1) / CKB : (in.C + CK - 1) / CK;
            if (s == 1 && (d->flags & LU_CONV_F_SRC1_CENTER)) {
                LU_REQUIRE(d->precision == 1 && si.nchunk == 1 && d->epilogue == LU_EPI_LSTM,
                           "lu_conv2d_fwd: LU_CONV_F_SRC1_CENTER needs precision 1, the ConvLSTM epilogue and C <= 32 on source 1");
                a.src1_center = 1;
                a.n_it += 1;
            } else {
                a.n_it += si.nchunk * a.kk;
            }
        } else if (vec) {
            a.n_it += si.nchunk * a.kk;
        }
        bvec = bvec && (in.w_row_stride % 4 == 0) && (in.w_tap_stride % 4 == 0) && aligned16(in.w);
    }
    a.M = (int64_t)d->frames * d->Hout * d->Wout;
    LU_REQUIRE(a.M < ((int64_t)1 << 31), "lu_conv2d_fwd: more than 2^31 output pixels in one call");
    a.HWo = d->Hout * d->Wout;
    a.Wout = d->Wout;
    a.Hin = d->Hin;
    a.Win = d->Win;
    a.stride = d->stride;
    a.dsh = d->dil - 1;
    a.pad_t = d->pad_t;
    a.pad_l = d->pad_l;
    a.N = d->N;
    a.out_pix_stride = d->out_pix_stride;
    a.bias = d->bias;
    a.out = d->out;
    a.out_frame_stride = d->out_frame_stride;
    a.out_row_stride = d->out_row_stride;
    a.out_vec4 = (d->out_row_stride == 0 && d->N % 4 == 0 && d->out_pix_stride % 4 == 0 && d->out_frame_stride % 4 == 0 &&
                  aligned16(d->out) && (!d->bias || aligned16(d->bias))) ? 1 : 0;
    a.out_vec4s = (d->N % 4 == 0 && d->out_pix_stride % 4 == 0 && d->out_frame_stride % 4 == 0 && d->out_row_stride % 4 == 0 &&
                   aligned16(d->out) && (!d->bias || aligned16(d->bias))) ? 1 : 0;
    a.post_scale = d->post_scale;
    a.post_shift = d->post_shift;
    a.post_alpha = d->post_alpha;
    LU_REQUIRE((d->post_scale == nullptr) == (d->post_shift == nullptr) && (!d->post_scale || d->epilogue == LU_EPI_BIAS),
               "lu_conv2d_fwd: post_scale / post_shift come as a pair and belong to LU_EPI_BIAS");
    LU_REQUIRE(!d->post_scale || (d->out_row_stride == 0 && d->out_pix_stride == d->N &&
                                  d->out_frame_stride == (int64_t)d->Hout * d->Wout * d->N),
               "lu_conv2d_fwd: the post affine needs a dense [frames, Hout, Wout, N] output");
    auto post_pass = [&]() -> int {      // unsplit launch with a post affine: short in-place pass over the dense output
        if (!a.post_scale) return 0;
        const int64_t tot = a.M * a.N;
        const unsigned g = (unsigned)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192);
        LU_LAUNCH(post_affine_kernel, dim3(g), dim3(256), stream, a.out, tot, a.N, a.post_scale, a.post_shift, a.post_alpha);
        return LU_CHECK_LAUNCH();
    };
    const bool slabs_only = (d->flags & LU_CONV_F_SLABS_ONLY) != 0;
    LU_REQUIRE(!slabs_only || (d->epilogue == LU_EPI_BIAS && d->splits > 1 && !d->post_scale),
               "lu_conv2d_fwd: LU_CONV_F_SLABS_ONLY needs LU_EPI_BIAS with splits > 1 (and no post affine)");
    int64_t m_tiles = (a.M + BM - 1) / BM;
    // halo-reuse kernel: stride-1 SAME 3x3 / 5x5, wide 16-byte-aligned outputs, <= 25 % of the 8x32 patches wasted
    const int64_t tiles_x = (d->Wout + 31) / 32;
    int th = 8;      // patch height; the bf16 kernel takes 16-row patches when that still leaves >= 1 block per CU
    // bf16, N = 32 / 64 (the decoder tail): narrow blocks of the fragment kernel, always 8-row patches
    const bool narrow_n = d->precision == 1 && d->epilogue == LU_EPI_BIAS && (d->N == 32 || d->N == 64) &&
                          !(d->flags & LU_CONV_F_NO_NARROW);
    if (d->precision != 0 && d->k == 5 && !narrow_n) {      // (3x3: below)
        const int64_t nt_est = d->epilogue == LU_EPI_LSTM ? d->N / 128 : (d->N + 127) / 128;
        const int force = (d->flags & LU_CONV_F_PATCH16) ? 16 : (d->flags & LU_CONV_F_PATCH8) ? 8 : 0;      // tests and A/B runs
        const int64_t sp = d->epilogue == LU_EPI_LSTM || d->splits < 1 ? 1 : d->splits;
        if (force ? force == 16
                  : (d->precision == 1 && (int64_t)d->frames * ((d->Hout + 15) / 16) * tiles_x * nt_est * sp >= 256))
            th = 16;      // (fp32 fragment mode: 8-row patches, two blocks per CU, unless forced)
    }
    {   // 3x3 bf16 layers whose sources are all bf16 tensors (activations / BatchNorm-backward gradients stored as bf16): the
        // second loop generation on 16-row patches as well -- 16 MFMAs per tap and wave and a three-deep weight-fragment ring
        // instead of 8 and two.  (Round 2 measured this slower; the scalar epilogue, twice as long on the tall tile, was why.)
        // fp32 sources stay on 8-row patches: their halo is twice the 16-byte pieces, more than one piece per tap.
        bool all16 = d->n_src > 0;
        for (int i = 0; i < d->n_src; ++i) all16 = all16 && d->src[i].dtype == LU_BF16;
        const int force = (d->flags & LU_CONV_F_PATCH16) ? 16 : (d->flags & LU_CONV_F_PATCH8) ? 8 : 0;
        if (d->precision == 1 && d->k == 3 && !narrow_n && d->epilogue == LU_EPI_BIAS && all16 &&
            d->splits <= 1 &&
            (force ? force == 16 : (int64_t)d->frames * ((d->Hout + 15) / 16) * tiles_x * ((d->N + 127) / 128) >= 256))
            th = 16;
    }
    // half blocks (conv_halo_frag2_kernel<..., WM = 1>): 8 x 32 patches, 4 waves of 8 rows each, two independent blocks per CU
    bool half_blk = false;
    {
        bool all16 = d->n_src > 0;
        for (int i = 0; i < d->n_src; ++i) all16 = all16 && d->src[i].dtype == LU_BF16;
        // Measured (round 4, same-box A/B on the config-2 shapes): the 3x3 fused step 0.33 -> 0.42 of peak (it leaves the first loop
        // generation), the 5x5 fused step +2 %, 3x3 bias layers +3 %, 5x5 bias layers (the recurrent / input gradients) -0.5 %:
        // the library's own choice is half blocks for the ConvLSTM epilogue and for 3x3 layers, given two blocks per CU of work.
        const int64_t blocks8 = (int64_t)d->frames * ((d->Hout + 7) / 8) * tiles_x *
                                (d->epilogue == LU_EPI_LSTM ? d->N / 128 : (d->N + 127) / 128) * (d->epilogue == LU_EPI_LSTM || d->splits < 1 ? 1 : d->splits);
        const bool forced = (d->flags & (LU_CONV_F_PATCH8 | LU_CONV_F_PATCH16)) != 0;
        const bool own_choice = !forced && blocks8 >= 512 && (d->epilogue == LU_EPI_LSTM || d->k == 3);
        half_blk = d->precision == 1 && !narrow_n && (d->k == 5 || (d->k == 3 && all16)) &&
                   ((d->flags & LU_CONV_F_HALF_BLOCK) || own_choice);
        if (half_blk) th = 8;
    }
    const int64_t tiles_y = (d->Hout + th - 1) / th;
    const bool halo = d->stride == 1 && d->dil == 1 && k_h == d->k && (d->k == 3 || d->k == 5) && d->pad_t == (d->k - 1) / 2 &&
                      d->pad_l == (d->k - 1) / 2 && d->Hout == d->Hin && d->Wout == d->Win && bvec &&
                      (d->N > 64 || narrow_n) &&
                      a.n_src > 0 && d->out_row_stride == 0 &&
                      (d->precision != 0 || (tiles_y * tiles_x * 256 * 4 <= (int64_t)d->Hout * d->Wout * 5 &&
                                             !(d->flags & LU_CONV_F_NO_HALO)));
    if (halo) {
        a.tiles_x = (int32_t)tiles_x;
        a.tiles_pf = (int32_t)(tiles_y * tiles_x);
        m_tiles = (int64_t)d->frames * tiles_y * tiles_x;
    }
    const bool want_xcd_n = (d->flags & LU_CONV_F_XCD_BY_N) != 0;
#define LU_ARGS(...) __VA_ARGS__
#define LU_LAUNCH_FRAG23(targs, grid_, blk_, lds_) LU_LAUNCH_DYN((conv_halo_frag3_kernel<targs>), grid_, blk_, lds_, stream, a)
    bool src16 = false;      // all sources bf16 tensors (a property of the launch: mixed element types are rejected)
    for (int s2 = 0; s2 < a.n_src; ++s2) {
        LU_REQUIRE(!a.src[s2].bf16 || (halo && d->precision == 1),
                   "lu_conv2d_fwd: bf16 activations are read by the bf16 halo kernel only (stride-1 3x3 / 5x5, N > 64 or N = 32 / 64)");
        LU_REQUIRE(a.src[s2].bf16 == a.src[0].bf16, "lu_conv2d_fwd: the sources of one launch must share an element type");
        LU_REQUIRE(!a.src[s2].bf16 || (a.src[s2].C % 8 == 0 && a.src[s2].pix_stride % 8 == 0 && a.src[s2].frame_stride % 8 == 0),
                   "lu_conv2d_fwd: a bf16 source needs C, pixel and frame strides that are multiples of 8 (source %d)", s2);
        src16 = a.src[s2].bf16 != 0;
    }
    LU_REQUIRE(!a.src1_center || (halo && a.n_src == 2), "lu_conv2d_fwd: LU_CONV_F_SRC1_CENTER needs the halo kernel and two sources");
    LU_REQUIRE(d->precision == 0 || d->precision == 1, "lu_conv2d_fwd: unknown precision %d", d->precision);
    if (d->precision == 1)
        LU_REQUIRE(d->dil == 1 && (halo || d->epilogue == LU_EPI_BIAS),
                   "lu_conv2d_fwd: bf16 mode has no input dilation, and the ConvLSTM epilogue needs a stride-1 3x3 / 5x5 layer");
    a.m_tiles = (int32_t)m_tiles;
    const int64_t m_tiles8 = (m_tiles + 7) / 8 * 8;     // XCD-aware order pads the m-tile count to 8
    a.ksplit = 1;
    a.balanced = a.balanced_by_m = 0;
    a.dbg = (d->flags >> 16) & 0xff;      // (only -DLU_ABLATION tool builds look at it)
    // grid of a tile kernel: few-tile launches take the balanced numbering of lu_block_tile (a.n_tiles / a.ksplit set before)
    const bool no_balance = (d->flags & LU_CONV_F_NO_BALANCE) != 0;
    auto tile_grid = [&]() {
        const int64_t work = m_tiles * a.n_tiles * a.ksplit;
        if (!no_balance && !a.xcd_by_n && work <= 2048 && (a.ksplit > 1 || m_tiles < 64)) {
            a.balanced = (int32_t)((work + 7) / 8);
            a.balanced_by_m = (int64_t)a.kk * a.N < a.M;      // weight bytes / input bytes = k*k*N / M
            return dim3((unsigned)(a.balanced * 8));
        }
        return dim3((unsigned)((a.xcd_by_n ? m_tiles : m_tiles8) * a.n_tiles), (unsigned)a.ksplit);
    };
    dim3 block(256);
    // (LDS-DMA tile staging of the general fp32 kernel measured 4-5 % slower than VGPR staging, its 4-wave form of the wide tiles neutral:
    // both opt-in instances left with ABI v12)
    if (d->epilogue == LU_EPI_LSTM) {
        LU_REQUIRE(d->N % 4 == 0 && (d->N / 4) % 32 == 0, "lu_conv2d_fwd: LSTM epilogue needs F %% 32 == 0 (N=%d)", d->N);
        LU_REQUIRE(bvec, "lu_conv2d_fwd: LSTM epilogue needs 16-byte aligned weights");
        LU_REQUIRE(d->c_prev && d->c_out && d->h_out, "lu_conv2d_fwd: LSTM epilogue pointers missing");
        a.F = d->N / 4;
        a.c_prev = d->c_prev;
        a.c_out = d->c_out;
        a.h_out = d->h_out;
        a.gates_out = d->gates_out;
        a.c_prev_fs = d->c_prev_frame_stride;
        a.c_out_fs = d->c_out_frame_stride;
        a.h_fs = d->h_frame_stride;
        a.gates_fs = d->gates_frame_stride;
        a.gates_bf16 = (d->flags & LU_CONV_F_GATES_BF16) ? 1 : 0;
        a.lstm_vec4 = (aligned16(d->c_prev) && aligned16(d->c_out) && aligned16(d->h_out) && (!d->bias || aligned16(d->bias)) &&
                       (!d->gates_out || aligned16(d->gates_out)) && d->c_prev_frame_stride % 4 == 0 && d->c_out_frame_stride % 4 == 0 &&
                       d->h_frame_stride % 4 == 0 && d->gates_frame_stride % 4 == 0) ? 1 : 0;
        a.h16_out = (unsigned short*)d->h16_out;
        a.h16_fs = d->h16_frame_stride;
        a.h16_split = (d->flags & LU_CONV_F_H16_SPLIT) ? 1 : 0;
        LU_REQUIRE(!a.h16_split || a.h16_out, "lu_conv2d_fwd: LU_CONV_F_H16_SPLIT needs h16_out");
        LU_REQUIRE((!a.gates_bf16 && !a.h16_out) || (d->precision == 1 && halo),
                   "lu_conv2d_fwd: the bf16 tape outputs (h16_out, LU_CONV_F_GATES_BF16) belong to the bf16 halo kernel");
        if (d->precision != 0 && halo)
            LU_REQUIRE(aligned16(d->c_prev) && aligned16(d->c_out) && aligned16(d->h_out) && aligned16(d->bias) &&
                           (!d->gates_out || (reinterpret_cast<uintptr_t>(d->gates_out) & 7) == 0) &&
                           (!d->h16_out || (reinterpret_cast<uintptr_t>(d->h16_out) & 7) == 0),
                       "lu_conv2d_fwd: the fragment kernel's ConvLSTM epilogue needs 16-byte aligned state / bias pointers");
        a.n_tiles = a.F / 32;
        a.xcd_by_n = (halo && want_xcd_n && a.n_tiles % 8 == 0) ? 1 : 0;
        const dim3 grid = tile_grid();
        LU_REQUIRE(d->dil == 1 && d->stride == 1 && k_h == d->k, "lu_conv2d_fwd: LSTM epilogue needs stride 1, dil 1, a square kernel");
        // (3x3: the first loop generation -- its 8-row-patch instance fits 4 waves per SIMD, the unrolled one does not: measured)
        if (half_blk && src16 && d->k == 5) LU_LAUNCH_FRAG23(LU_ARGS(5, LU_EPI_LSTM, 8, true, 1), grid, dim3(256), halo_bf16_lds(5, 4));
        else if (half_blk && d->k == 5) LU_LAUNCH_FRAG23(LU_ARGS(5, LU_EPI_LSTM, 8, false, 1), grid, dim3(256), halo_bf16_lds(5, 4));
        else if (half_blk) LU_LAUNCH_FRAG23(LU_ARGS(3, LU_EPI_LSTM, 8, true, 1), grid, dim3(256), halo_bf16_lds(3, 4));
        else if (d->precision == 1 && src16 && d->k == 5 && th == 16) LU_LAUNCH_FRAG23(LU_ARGS(5, LU_EPI_LSTM, 8, true), grid, dim3(512), halo_bf16_lds(5, 8));
        else if (d->precision == 1 && src16 && d->k == 5) LU_LAUNCH_FRAG23(LU_ARGS(5, LU_EPI_LSTM, 4, true), grid, dim3(512), halo_bf16_lds(5, 4));
        else if (d->precision == 1 && d->k == 5 && th == 16) LU_LAUNCH_FRAG23(LU_ARGS(5, LU_EPI_LSTM, 8, false), grid, dim3(512), halo_bf16_lds(5, 8));
        else if (d->precision == 1 && d->k == 5) LU_LAUNCH_FRAG23(LU_ARGS(5, LU_EPI_LSTM, 4, false), grid, dim3(512), halo_bf16_lds(5, 4));
        else if (d->precision == 1 && src16) LU_LAUNCH_DYN((conv_halo_frag_kernel<3, LU_EPI_LSTM, 4, true>), grid, dim3(512), halo_bf16_lds(3, 4), stream, a);
        else if (d->precision == 1) LU_LAUNCH_DYN((conv_halo_frag_kernel<3, LU_EPI_LSTM, 4, false>), grid, dim3(512), halo_bf16_lds(3, 4), stream, a);
        else if (halo && d->k == 5) LU_LAUNCH((conv_halo_kernel<5, LU_EPI_LSTM, true>), grid, dim3(512), stream, a);      // (the gate epilogue takes no K split)
        else if (halo) LU_LAUNCH((conv_halo_kernel<3, LU_EPI_LSTM, true>), grid, dim3(512), stream, a);
        else LU_LAUNCH((conv_fwd_kernel<4, true, LU_EPI_LSTM, false, 1, false>), grid, dim3(512), stream, a);
        return LU_CHECK_LAUNCH();
    }
    LU_REQUIRE(d->epilogue == LU_EPI_BIAS, "lu_conv2d_fwd: unknown epilogue %d", d->epilogue);
    LU_REQUIRE(d->out || slabs_only, "lu_conv2d_fwd: out is null");
    const int nf = d->N > 64 ? 4 : (d->N > 32 ? 2 : 1);
    a.n_tiles = (d->N + 32 * nf - 1) / (32 * nf);
    if (d->splits > 1 && a.n_it >= 2 * d->splits) {
        LU_REQUIRE(d->workspace, "lu_conv2d_fwd: splits > 1 needs a workspace");
        a.ksplit = d->splits;
        a.ws = (float*)d->workspace;
    }
    LU_REQUIRE(!slabs_only || a.ksplit > 1, "lu_conv2d_fwd: LU_CONV_F_SLABS_ONLY with more splits than half the k-steps (%d)", a.n_it);
    if (d->precision == 1) {     // bf16 MFMA operands: halo kernel where it applies, the gather kernel everywhere else
        a.n_tiles = (d->N + 127) / 128;
        const bool narrow = halo && d->N <= 64;      // N = 32 / 64: one block covers every column (NFR = 1 / 2)
        const dim3 gridb = tile_grid();
        if (narrow && d->N == 32 && d->k == 5 && src16)
            LU_LAUNCH_DYN((conv_halo_frag_kernel<5, LU_EPI_BIAS, 1, true, 1>), gridb, dim3(512), halo_bf16_lds(5, 4), stream, a);
        else if (narrow && d->N == 32 && d->k == 5)
            LU_LAUNCH_DYN((conv_halo_frag_kernel<5, LU_EPI_BIAS, 1, false, 1>), gridb, dim3(512), halo_bf16_lds(5, 4), stream, a);
        else if (narrow && d->N == 32 && src16)
            LU_LAUNCH_DYN((conv_halo_frag_kernel<3, LU_EPI_BIAS, 1, true, 1>), gridb, dim3(512), halo_bf16_lds(3, 4), stream, a);
        else if (narrow && d->N == 32)
            LU_LAUNCH_DYN((conv_halo_frag_kernel<3, LU_EPI_BIAS, 1, false, 1>), gridb, dim3(512), halo_bf16_lds(3, 4), stream, a);
        else if (narrow && d->k == 5 && src16)
            LU_LAUNCH_DYN((conv_halo_frag_kernel<5, LU_EPI_BIAS, 2, true, 2>), gridb, dim3(512), halo_bf16_lds(5, 4), stream, a);
        else if (narrow && d->k == 5)
            LU_LAUNCH_DYN((conv_halo_frag_kernel<5, LU_EPI_BIAS, 2, false, 2>), gridb, dim3(512), halo_bf16_lds(5, 4), stream, a);
        else if (narrow && src16)
            LU_LAUNCH_DYN((conv_halo_frag_kernel<3, LU_EPI_BIAS, 2, true, 2>), gridb, dim3(512), halo_bf16_lds(3, 4), stream, a);
        else if (narrow)
            LU_LAUNCH_DYN((conv_halo_frag_kernel<3, LU_EPI_BIAS, 2, false, 2>), gridb, dim3(512), halo_bf16_lds(3, 4), stream, a);
        else if (halo && half_blk && src16 && d->k == 5)
            LU_LAUNCH_FRAG23(LU_ARGS(5, LU_EPI_BIAS, 8, true, 1), gridb, dim3(256), halo_bf16_lds(5, 4));
        else if (halo && half_blk && d->k == 5)
            LU_LAUNCH_FRAG23(LU_ARGS(5, LU_EPI_BIAS, 8, false, 1), gridb, dim3(256), halo_bf16_lds(5, 4));
        else if (halo && half_blk)
            LU_LAUNCH_FRAG23(LU_ARGS(3, LU_EPI_BIAS, 8, true, 1), gridb, dim3(256), halo_bf16_lds(3, 4));
        else if (halo && src16 && d->k == 5 && th == 16)
            LU_LAUNCH_FRAG23(LU_ARGS(5, LU_EPI_BIAS, 8, true), gridb, dim3(512), halo_bf16_lds(5, 8));
        else if (halo && src16 && d->k == 5)
            LU_LAUNCH_FRAG23(LU_ARGS(5, LU_EPI_BIAS, 4, true), gridb, dim3(512), halo_bf16_lds(5, 4));
        else if (halo && d->k == 5 && th == 16)
            LU_LAUNCH_FRAG23(LU_ARGS(5, LU_EPI_BIAS, 8, false), gridb, dim3(512), halo_bf16_lds(5, 8));
        else if (halo && d->k == 5)
            LU_LAUNCH_FRAG23(LU_ARGS(5, LU_EPI_BIAS, 4, false), gridb, dim3(512), halo_bf16_lds(5, 4));
        else if (halo && src16 && d->k == 3 && th == 16)
            LU_LAUNCH_FRAG23(LU_ARGS(3, LU_EPI_BIAS, 8, true), gridb, dim3(512), halo_bf16_lds(3, 8));
        else if (halo && src16)
            LU_LAUNCH_DYN((conv_halo_frag_kernel<3, LU_EPI_BIAS, 4, true>), gridb, dim3(512), halo_bf16_lds(3, 4), stream, a);
        else if (halo)
            LU_LAUNCH_DYN((conv_halo_frag_kernel<3, LU_EPI_BIAS, 4, false>), gridb, dim3(512), halo_bf16_lds(3, 4), stream, a);
        else
            LU_LAUNCH(conv_gather_bf16_kernel, gridb, dim3(512), stream, a);
        int rcb = LU_CHECK_LAUNCH();
        if (rcb || slabs_only) return rcb;
        if (a.ksplit == 1) return post_pass();
        const int64_t totb = a.M * a.N;
        const unsigned rgb = (unsigned)((totb + 255) / 256 < 8192 ? (totb + 255) / 256 : 8192);
        LU_LAUNCH(ksplit_reduce_kernel, dim3(rgb), dim3(256), stream, (const float*)a.ws, a.ksplit, a.M, a.N, a.HWo, a.bias,
                  a.out, a.out_frame_stride, a.out_pix_stride, a.Wout, a.out_row_stride, a.post_scale, a.post_shift, a.post_alpha);
        return LU_CHECK_LAUNCH();
    }
    // fp32 halo kernel with a K split: whole chunks per slice (the compile-time tap sequence, round 5) whenever every slice gets
    // at least one chunk; otherwise the counted loop (slices of ceil(n_it / splits) stages from any tap on)
    a.split_chunks = (halo && nf == 4 && a.ksplit > 1 &&
                      lu_conv_chunk_splits(a.n_it / a.kk, a.ksplit) == a.ksplit) ? 1 : 0;
    const dim3 grid = tile_grid();
    const bool gen = d->dil != 1 || (d->flags & LU_CONV_F_GENERAL) != 0;   // A/B knob for tools/kbench.py
#define LU_CONV_CASE(NF_, BV_)                                                                  \
    if (nf == NF_ && bvec == BV_) {                                                             \
        if (halo && NF_ == 4 && d->k == 5 && (a.ksplit <= 1 || a.split_chunks)) LU_LAUNCH((conv_halo_kernel<5, LU_EPI_BIAS, true>), grid, dim3(512), stream, a); \
        else if (halo && NF_ == 4 && (a.ksplit <= 1 || a.split_chunks)) LU_LAUNCH((conv_halo_kernel<3, LU_EPI_BIAS, true>), grid, dim3(512), stream, a);   \
        else if (halo && NF_ == 4 && d->k == 5) LU_LAUNCH((conv_halo_kernel<5, LU_EPI_BIAS>), grid, dim3(512), stream, a); \
        else if (halo && NF_ == 4) LU_LAUNCH((conv_halo_kernel<3, LU_EPI_BIAS>), grid, dim3(512), stream, a);   \
        else if (gen) LU_LAUNCH((conv_fwd_kernel<NF_, BV_, LU_EPI_BIAS, true, 2, false>), grid, block, stream, a); \
        else if (NF_ == 4 && BV_)                                                                               \
            LU_LAUNCH((conv_fwd_kernel<4, true, LU_EPI_BIAS, false, 1, false>), grid, dim3(512), stream, a);    \
        else LU_LAUNCH((conv_fwd_kernel<NF_, BV_, LU_EPI_BIAS, false, 2, false>), grid, block, stream, a);      \
        int rc_ = LU_CHECK_LAUNCH();                                                            \
        if (rc_ || slabs_only) return rc_;                                                      \
        if (a.ksplit == 1) return post_pass();                                                  \
        const int64_t tot_ = a.M * a.N;                                                         \
        const unsigned rg_ = (unsigned)((tot_ + 255) / 256 < 8192 ? (tot_ + 255) / 256 : 8192); \
        LU_LAUNCH(ksplit_reduce_kernel, dim3(rg_), dim3(256), stream, (const float*)a.ws, a.ksplit, a.M, a.N,   \
                  a.HWo, a.bias, a.out, a.out_frame_stride, a.out_pix_stride, a.Wout, a.out_row_stride, a.post_scale,  \
                  a.post_shift, a.post_alpha);                                                                   \
        return LU_CHECK_LAUNCH();                                                               \
    }
    LU_CONV_CASE(4, true)
    LU_CONV_CASE(2, true)
    LU_CONV_CASE(1, true)
    LU_CONV_CASE(4, false)
    LU_CONV_CASE(2, false)
    LU_CONV_CASE(1, false)
#undef LU_CONV_CASE
    lu_set_error("lu_conv2d_fwd: no kernel variant");
    return 1;
}

extern "C" int lu_stride2_dgrad_weights(const float* w, float* sub, int k, int ks, int C, int N, int pt, int pl,
                                        int pady0, int pady1, int padx0, int padx1, lu_stream_t stream) {
    LU_REQUIRE(w && sub && k > 0 && ks > 0 && C > 0 && N > 0, "lu_stride2_dgrad_weights: bad arguments");
    auto count = [&](int par, int pad) {      // taps kh of a parity class: (par + pad - kh) even, 0 <= kh < k
        int n = 0;
        for (int kh = 0; kh < k; ++kh) n += ((par + pad - kh) % 2 == 0);
        return n;
    };
    const int ny0 = count(0, pt), ny1 = count(1, pt), nx0 = count(0, pl), nx1 = count(1, pl);
    const int64_t total = (int64_t)(ny0 + ny1) * (nx0 + nx1) * N * C;
    const unsigned g = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    LU_LAUNCH(s2_dgrad_weights_kernel, dim3(g), dim3(256), stream, w, sub, k, C, N, pt, pl, pady0, pady1, padx0, padx1, ny0,
              ny1, nx0, nx1);
    return LU_CHECK_LAUNCH();
}

/* Input gradient of a stride-2 3x3 convolution on even input extents (TF-SAME pads 0 / 1), all four parity classes in ONE
 * launch on bf16 MFMA operands (reference layers: Networks.py:52-56, the first Conv2D of a down block).
 *   dy [frames, Hd, Wd, Nf] fp32 (frame / pixel strides given), packed = lu_pack_weights_taps_bf16 of the nine tap matrices
 *   [Nf][C] that lu_stride2_dgrad_weights(w, ..., k = 3, pad_t = pad_l = 0) writes (class order), dx dense [frames, 2 Hd, 2 Wd, C]. */
extern "C" int lu_conv2d_s2_dgrad_bf16(const float* dy, int64_t dy_frame_stride, int32_t dy_pix_stride, const void* packed,
                                       int32_t frames, int32_t Hd, int32_t Wd, int32_t Nf, int32_t C, float* dx,
                                       lu_stream_t stream) {
    LU_REQUIRE(dy && packed && dx && frames > 0 && Hd > 0 && Wd > 0 && Nf > 0 && C > 0, "lu_conv2d_s2_dgrad_bf16: bad arguments");
    LU_REQUIRE(Nf % 32 == 0 && dy_pix_stride % 4 == 0 && dy_frame_stride % 4 == 0 && aligned16(dy),
               "lu_conv2d_s2_dgrad_bf16: dy needs Nf %% 32 == 0 and 16-byte aligned pixels");
    S2DgradArgs a;
    memset(&a, 0, sizeof(a));
    a.dy = dy;
    a.w = (const unsigned char*)packed;
    a.dx = dx;
    a.dy_fs = dy_frame_stride;
    a.dy_ps = dy_pix_stride;
    a.Hd = Hd;
    a.Wd = Wd;
    a.Nf = Nf;
    a.C = C;
    a.tiles_x = (Wd + 31) / 32;
    a.tiles_pf = ((Hd + 1) / 2) * a.tiles_x;
    const int64_t m_tiles = (int64_t)frames * a.tiles_pf;
    LU_REQUIRE(m_tiles < ((int64_t)1 << 27), "lu_conv2d_s2_dgrad_bf16: too many tiles");
    a.m_tiles = (int32_t)m_tiles;
    a.n_tiles = (C + 127) / 128;
    const int64_t m8 = (m_tiles + 7) / 8 * 8;
    LU_LAUNCH(conv_s2_dgrad_bf16_kernel, dim3((unsigned)(m8 * a.n_tiles)), dim3(256), stream, a);
    return LU_CHECK_LAUNCH();
}

/* Stride-2 3x3 forward convolution on even input extents (TF-SAME pads 0 / 1; Networks.py:52-56: the first Conv2D of a down
 * block) reading a BF16 tensor x [frames, Hin, Win, C] (C % 8 == 0; the bf16 copy of the ConvLSTM output), weights packed by
 * lu_pack_weights_bf16(w [3,3,C,N]), out fp32 dense [frames, Hin / 2, Win / 2, N] = conv + bias. */
extern "C" int lu_conv2d_s2_fwd_bf16(const void* x_bf16, int64_t x_frame_stride, int32_t x_pix_stride, const void* packed,
                                     const float* bias, int32_t frames, int32_t Hin, int32_t Win, int32_t C, int32_t N,
                                     float* out, lu_stream_t stream) {
    LU_REQUIRE(x_bf16 && packed && out && frames > 0 && Hin > 0 && Win > 0 && C > 0 && N > 0, "lu_conv2d_s2_fwd_bf16: bad arguments");
    LU_REQUIRE(Hin % 2 == 0 && Win % 2 == 0 && C % 8 == 0 && x_pix_stride % 8 == 0 && x_frame_stride % 8 == 0 && aligned16(x_bf16),
               "lu_conv2d_s2_fwd_bf16: needs even extents, C %% 8 == 0 and 16-byte aligned pixels");
    S2FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const unsigned short*)x_bf16;
    a.w = (const unsigned char*)packed;
    a.bias = bias;
    a.out = out;
    a.x_fs = x_frame_stride;
    a.x_ps = x_pix_stride;
    a.Hin = Hin;
    a.Win = Win;
    a.C = C;
    a.N = N;
    const int Hout = Hin / 2, Wout = Win / 2;
    a.tiles_x = (Wout + 31) / 32;
    a.tiles_pf = ((Hout + 3) / 4) * a.tiles_x;
    const int64_t m_tiles = (int64_t)frames * a.tiles_pf;
    LU_REQUIRE(m_tiles < ((int64_t)1 << 27), "lu_conv2d_s2_fwd_bf16: too many tiles");
    a.m_tiles = (int32_t)m_tiles;
    a.n_tiles = (N + 127) / 128;
    const int64_t m8 = (m_tiles + 7) / 8 * 8;
    LU_LAUNCH(conv_s2_fwd_bf16_kernel, dim3((unsigned)(m8 * a.n_tiles)), dim3(512), stream, a);
    return LU_CHECK_LAUNCH();
}

extern "C" size_t lu_pack_weights_bf16_bytes(int k, int C, int N) {
    return (size_t)k * k * ((C + CKB - 1) / CKB) * (size_t)((N + 31) / 32) * 1024 * sizeof(unsigned short);
}

extern "C" int lu_pack_weights_bf16(const float* w, int64_t w_tap_stride, int w_row_stride, int k, int C, int N,
                                    void* out, lu_stream_t stream) {
    LU_REQUIRE(w && out && k > 0 && C > 0 && N > 0, "lu_pack_weights_bf16: bad arguments");
    const int64_t total = (int64_t)k * k * ((C + CKB - 1) / CKB) * ((N + 31) / 32) * 128;      // one thread per 8 elements
    const unsigned g = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    LU_LAUNCH(pack_weights_bf16_kernel, dim3(g), dim3(256), stream, w, w_tap_stride, w_row_stride, k * k, C, N,
              (unsigned short*)out);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_pack_weights_split6_bf16(const float* w, int64_t w_tap_stride, int w_row_stride, int k, int C, int cp, int N,
                                           int order, void* out, lu_stream_t stream) {
    LU_REQUIRE(w && out && k > 0 && C > 0 && cp >= C && N > 0 && (order == 0 || order == 1), "lu_pack_weights_split6_bf16: bad arguments");
    // piece (0 hi, 1 mid, 2 lo) of block j, two bits each: order 0 = A (lo, mid, hi, mid, hi, hi), 1 = B (hi, mid, lo, hi, mid, hi) -- lu_split6
    const unsigned ord = order == 0 ? (2u | (1u << 2) | (0u << 4) | (1u << 6) | (0u << 8) | (0u << 10))
                                    : (0u | (1u << 2) | (2u << 4) | (0u << 6) | (1u << 8) | (0u << 10));
    const int64_t total = (int64_t)k * k * ((6 * cp + CKB - 1) / CKB) * ((N + 31) / 32) * 128;
    const unsigned g = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    LU_LAUNCH(pack_weights_split6_bf16_kernel, dim3(g), dim3(256), stream, w, w_tap_stride, w_row_stride, k * k, C, cp, N, ord,
              (unsigned short*)out);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_pack_weights_taps_bf16(const float* w, int64_t w_tap_stride, int w_row_stride, int taps, int C, int N,
                                         void* out, lu_stream_t stream) {
    LU_REQUIRE(w && out && taps > 0 && C > 0 && N > 0, "lu_pack_weights_taps_bf16: bad arguments");
    const int64_t total = (int64_t)taps * ((C + CKB - 1) / CKB) * ((N + 31) / 32) * 128;      // one thread per 8 elements
    const unsigned g = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    LU_LAUNCH(pack_weights_bf16_kernel, dim3(g), dim3(256), stream, w, w_tap_stride, w_row_stride, taps, C, N,
              (unsigned short*)out);
    return LU_CHECK_LAUNCH();
}

extern "C" size_t lu_conv2d_workspace_bytes(const lu_conv_desc* d) {
    if (!d || d->splits <= 1) return 0;
    return (size_t)d->splits * d->frames * d->Hout * d->Wout * (size_t)d->N * sizeof(float);
}

extern "C" int lu_weight_flip_transpose(const float* w, float* wt, int k, int C_tot, int N, int c_off, int C_sub,
                                        lu_stream_t stream) {
    LU_REQUIRE(w && wt && k > 0 && C_sub > 0 && N > 0 && c_off >= 0 && c_off + C_sub <= C_tot,
               "lu_weight_flip_transpose: bad arguments");
    dim3 grid((N + 31) / 32, (C_sub + 31) / 32, k * k);
    LU_LAUNCH(flip_transpose_kernel, grid, dim3(256), stream, w, wt, k, C_tot, N, c_off, C_sub);
    return LU_CHECK_LAUNCH();
}

/* dev_ops: n_ops lu_prep_op records in DEVICE memory, sorted by blk0, covering blocks [0, total_blocks). */
extern "C" int lu_weight_prep_batch(const void* dev_ops, int n_ops, int total_blocks, lu_stream_t stream) {
    static_assert(sizeof(PrepOp) == 64, "lu_prep_op layout");
    LU_REQUIRE(dev_ops && n_ops > 0 && total_blocks > 0, "lu_weight_prep_batch: bad arguments");
    LU_LAUNCH(weight_prep_batch_kernel, dim3((unsigned)total_blocks), dim3(256), stream, (const PrepOp*)dev_ops, n_ops);
    return LU_CHECK_LAUNCH();
}
