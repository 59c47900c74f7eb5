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
    int32_t HWo, Wout, Hin, Win;
    int32_t k, kk, stride, pad_t, pad_l;
    int32_t c_tiles;
    float* ws;          // [splits][kk*C*N]
    int64_t slab;
};

template <int MF, int NF, int WM, int WN, bool THIN, bool YVEC>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
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

    float4 ra[THIN ? 1 : NPA];
    float rat[2];
    float4 rb[NPB];

    auto load_stage = [&](int it) {
        const int64_t pb = p_begin + (int64_t)it * KP;
        if (!THIN) {
#pragma unroll
            for (int ps = 0; ps < NPA; ++ps) {
                const int row = arow0 + RPA * ps;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int64_t p = pb + row;
                const int c = c0 + 4 * aq;
                if (row < KP && p < p_end && c < a.C) {
                    int f = (int)(p / a.HWo);
                    int r = (int)(p - (int64_t)f * a.HWo);
                    int oy = r / a.Wout, ox = r - oy * a.Wout;
                    int iy = oy * a.stride + kh - a.pad_t, ix = ox * a.stride + kw - a.pad_l;
                    if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
                        v = *reinterpret_cast<const float4*>(a.x + (int64_t)f * a.x_fs +
                                                             ((int64_t)iy * a.Win + ix) * a.x_ps + c);
                }
                ra[ps] = v;
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int row = (tid >> 5) + 8 * ps;
                const int64_t p = pb + row;
                float v = 0.f;
                if (t_ok && p < p_end) {
                    int f = (int)(p / a.HWo);
                    int r = (int)(p - (int64_t)f * a.HWo);
                    int oy = r / a.Wout, ox = r - oy * a.Wout;
                    int iy = oy * a.stride + t_kh - a.pad_t, ix = ox * a.stride + t_kw - a.pad_l;
                    if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
                        v = a.x[(int64_t)f * a.x_fs + ((int64_t)iy * a.Win + ix) * a.x_ps + t_c];
                }
                rat[ps] = v;
            }
        }
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) {
            const int row = brow0 + RPB * ps;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int64_t p = pb + row;
            const int n = n0 + 4 * bq;
            if (row < KP && p < p_end) {
                int f = (int)(p / a.HWo);
                int64_t r = p - (int64_t)f * a.HWo;
                const float* yp = a.dy + (int64_t)f * a.dy_fs + r * a.dy_ps + n;
                if (YVEC) {
                    if (n < a.N) v = *reinterpret_cast<const float4*>(yp);
                } else {
                    if (n + 0 < a.N) v.x = yp[0];
                    if (n + 1 < a.N) v.y = yp[1];
                    if (n + 2 < a.N) v.z = yp[2];
                    if (n + 3 < a.N) v.w = yp[3];
                }
            }
            rb[ps] = v;
        }
    };
    auto store_stage = [&](int buf) {
        if (!THIN) {
#pragma unroll
            for (int ps = 0; ps < NPA; ++ps) {
                const int row = arow0 + RPA * ps;
                if (row < KP) *reinterpret_cast<float4*>(&As[buf][row * BMw + 4 * aq]) = ra[ps];
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) As[buf][((tid >> 5) + 8 * ps) * BMw + tcol] = rat[ps];
        }
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) {
            const int row = brow0 + RPB * ps;
            if (row < KP) *reinterpret_cast<float4*>(&Bs[buf][row * BNw + 4 * bq]) = rb[ps];
        }
    };

    f32x16 acc[MF][NF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;

    if (n_it > 0) {
        load_stage(0);
        store_stage(0);
    }
    __syncthreads();
    const int khalf = lane >> 5, l31 = lane & 31;
    for (int it = 0; it < n_it; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < n_it;
        if (more) load_stage(it + 1);
#pragma unroll
        for (int kk2 = 0; kk2 < KP; kk2 += 2) {
            float av[MF], bv[NF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) av[mf] = As[buf][(kk2 + khalf) * BMw + wm * 32 * MF + mf * 32 + l31];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) bv[nf] = Bs[buf][(kk2 + khalf) * BNw + wn * 32 * NF + nf * 32 + l31];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = lu_mfma(av[mf], bv[nf], acc[mf][nf]);
        }
        if (more) store_stage(buf ^ 1);
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

__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, int64_t slab, int splits, float* __restrict__ dw,
                                    int C, int N, int64_t tap_stride, int row_stride, float beta) {
    const int64_t total = slab;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
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

}  // namespace

extern "C" size_t lu_conv2d_wgrad_workspace_bytes(const lu_wgrad_desc* d) {
    if (!d) return 0;
    int splits = d->splits > 0 ? d->splits : 1;
    return (size_t)splits * d->k * d->k * (size_t)d->C * d->N * sizeof(float);
}

extern "C" int lu_conv2d_wgrad(const lu_wgrad_desc* d, lu_stream_t stream) {
    LU_REQUIRE(d && d->x && d->dy && d->dw && d->workspace, "lu_conv2d_wgrad: null pointer");
    LU_REQUIRE(d->k >= 1 && d->k <= 7 && (d->stride == 1 || d->stride == 2), "lu_conv2d_wgrad: bad k/stride");
    LU_REQUIRE(d->C > 0 && d->N > 0 && d->frames > 0, "lu_conv2d_wgrad: empty problem");
    const int splits = d->splits > 0 ? d->splits : 1;
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = d->x;
    a.dy = d->dy;
    a.x_fs = d->x_frame_stride;
    a.dy_fs = d->dy_frame_stride;
    a.x_ps = d->x_pix_stride;
    a.dy_ps = d->dy_pix_stride;
    a.C = d->C;
    a.N = d->N;
    a.M = (int64_t)d->frames * d->Hout * d->Wout;
    a.chunk = ((a.M + splits - 1) / splits + KP - 1) / KP * KP;
    a.HWo = d->Hout * d->Wout;
    a.Wout = d->Wout;
    a.Hin = d->Hin;
    a.Win = d->Win;
    a.k = d->k;
    a.kk = d->k * d->k;
    a.stride = d->stride;
    a.pad_t = d->pad_t;
    a.pad_l = d->pad_l;
    a.ws = (float*)d->workspace;
    a.slab = (int64_t)a.kk * d->C * d->N;
    const bool xvec = (d->C % 4 == 0) && (d->x_pix_stride % 4 == 0) && (d->x_frame_stride % 4 == 0) && aligned16(d->x);
    const bool yvec = (d->N % 4 == 0) && (d->dy_pix_stride % 4 == 0) && (d->dy_frame_stride % 4 == 0) && aligned16(d->dy);
    dim3 block(256);
    const unsigned n_tiles = (unsigned)((d->N + 127) / 128);
#define LU_WG(MF_, NF_, WM_, WN_, THIN_, YV_, GY_)                                                          \
    do {                                                                                                    \
        dim3 grid(n_tiles, (unsigned)(GY_), (unsigned)splits);                                              \
        LU_LAUNCH((wgrad_kernel<MF_, NF_, WM_, WN_, THIN_, YV_>), grid, block, stream, a);                  \
    } while (0)
    if (!xvec) {
        const int gy = (a.kk * d->C + 31) / 32;
        if (yvec) LU_WG(1, 1, 1, 4, true, true, gy);
        else LU_WG(1, 1, 1, 4, true, false, gy);
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
    if (rc) return rc;
    int64_t total = a.slab;
    unsigned rgrid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    LU_LAUNCH(wgrad_reduce_kernel, dim3(rgrid), dim3(256), stream, (const float*)a.ws, a.slab, splits, d->dw, d->C,
              d->N, d->dw_tap_stride, d->dw_row_stride, d->beta);
    return LU_CHECK_LAUNCH();
}
