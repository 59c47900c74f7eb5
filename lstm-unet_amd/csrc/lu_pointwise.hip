// lu_pointwise.hip -- the HBM-bound kernels of the ConvLSTM-UNet step: ConvLSTM gate block,
// BatchNorm(+LeakyReLU) statistics / apply / backward, bilinear x2 up-sampling, reflect-pad / crop
// window copies, 3-class softmax + weighted cross-entropy, Adam, state mask, layout transposes.
// All tensors are channels-last so consecutive lanes touch consecutive channels (coalesced);
// reductions are two-stage and deterministic (per-block partials in a caller workspace, summed in a
// fixed order in double) -- no float atomics anywhere.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "lu_device.h"

// ---------------------------------------------------------------------------------------------
// error plumbing (shared by all translation units)
// ---------------------------------------------------------------------------------------------
static thread_local char g_lu_err[512] = "";

void lu_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_lu_err, sizeof(g_lu_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* lu_last_error(void) { return g_lu_err; }
extern "C" int lu_abi_version(void) { return 12; }

// ---------------------------------------------------------------------------------------------
// CRC-32C (Castagnoli), slicing-by-8, HOST code: the checksum of TensorFlow tensor bundles (tf_bundle.py reads / writes
// the reference's `model.ckpt` files, train2D.py:235 / Inference2D.py:34) -- 300 MB of weights in a Python byte loop
// would take minutes.
// ---------------------------------------------------------------------------------------------
extern "C" uint32_t lu_crc32c(const void* data, size_t n, uint32_t crc) {
    static uint32_t T[8][256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
        ready = true;
    }
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = crc ^ 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = T[7][lo & 0xFF] ^ T[6][(lo >> 8) & 0xFF] ^ T[5][(lo >> 16) & 0xFF] ^ T[4][lo >> 24] ^ T[3][hi & 0xFF] ^
            T[2][(hi >> 8) & 0xFF] ^ T[1][(hi >> 16) & 0xFF] ^ T[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = T[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

#ifndef LU_EMU
int lu_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        lu_set_error("HIP launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
#endif

namespace {

constexpr int NT = 256;

inline unsigned grid_for(int64_t n, int per_thread = 1) {
    int64_t b = (n + (int64_t)NT * per_thread - 1) / ((int64_t)NT * per_thread);
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;
    return (unsigned)b;
}

__device__ __forceinline__ float hsig(float z) { return fminf(fmaxf(0.2f * z + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float hsig_grad_from_out(float o) { return (o > 0.f && o < 1.f) ? 0.2f : 0.f; }

// ---------------------------------------------------------------------------------------------
// ConvLSTM gate block
// ---------------------------------------------------------------------------------------------
__global__ void lstm_gates_fwd_kernel(const float* __restrict__ z, const float* __restrict__ c_prev,
                                      float* __restrict__ c_out, float* __restrict__ h_out,
                                      float* __restrict__ gates_out, int64_t total, int64_t ppf, int F, int64_t h_fs) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t row = i / F;
        const int ch = (int)(i - row * F);
        const float* zp = z + row * 4 * F + ch;
        const float gi = hsig(zp[0]), gf = hsig(zp[F]), gg = lu_tanh_fast(zp[2 * F]), go = hsig(zp[3 * F]);
        const float cn = fmaf(gf, c_prev[i], gi * gg);      // (same contraction as the fused epilogues in lu_conv.hip)
        c_out[i] = cn;
        const int64_t f = row / ppf;
        h_out[f * h_fs + (row - f * ppf) * F + ch] = go * lu_tanh_fast(cn);
        if (gates_out) {
            float* gp = gates_out + row * 4 * F + ch;
            gp[0] = gi;
            gp[F] = gf;
            gp[2 * F] = gg;
            gp[3 * F] = go;
        }
    }
}

// The same gate block on the partial slabs of a K-split convolution: z = bias + slab 0 + slab 1 + ... (the order of
// ksplit_reduce_kernel, so both routes produce the same bits).
__global__ void lstm_gates_fwd_slabs_kernel(const float* __restrict__ slabs, int splits, int64_t slab_stride,
                                            const float* __restrict__ bias, const float* __restrict__ c_prev,
                                            float* __restrict__ c_out, float* __restrict__ h_out,
                                            float* __restrict__ gates_out, int64_t total, int64_t ppf, int F, int64_t h_fs) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t row = i / F;
        const int ch = (int)(i - row * F);
        const float* zp = slabs + row * 4 * F + ch;
        float zi = 0.f, zf = 0.f, zg = 0.f, zo = 0.f;
        if (bias) {
            zi = bias[ch];
            zf = bias[F + ch];
            zg = bias[2 * F + ch];
            zo = bias[3 * F + ch];
        }
        for (int k = 0; k < splits; ++k, zp += slab_stride) {
            zi += zp[0];
            zf += zp[F];
            zg += zp[2 * F];
            zo += zp[3 * F];
        }
        const float gi = hsig(zi), gf = hsig(zf), gg = lu_tanh_fast(zg), go = hsig(zo);
        const float cn = fmaf(gf, c_prev[i], gi * gg);
        c_out[i] = cn;
        const int64_t f = row / ppf;
        h_out[f * h_fs + (row - f * ppf) * F + ch] = go * lu_tanh_fast(cn);
        if (gates_out) {
            float* gp = gates_out + row * 4 * F + ch;
            gp[0] = gi;
            gp[F] = gf;
            gp[2 * F] = gg;
            gp[3 * F] = go;
        }
    }
}

// `dz` may alias `gates` (in-place: each thread reads its four gate values before it writes the four
// pre-activation gradients to the same addresses), hence no __restrict__ on those two.
__global__ void lstm_gates_bwd_kernel(const float* gates, const float* __restrict__ c_prev,
                                      const float* __restrict__ c_cur, const float* __restrict__ dh_a, int64_t dha_fs,
                                      const float* __restrict__ dh_b, const float* __restrict__ dc_in,
                                      float* dz, float* __restrict__ dc_prev_out, int64_t total,
                                      int64_t ppf, int F) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t row = i / F;
        const int ch = (int)(i - row * F);
        const int64_t f = row / ppf;
        float dh = dh_a[f * dha_fs + (row - f * ppf) * F + ch];
        if (dh_b) dh += dh_b[i];
        const float* gp = gates + row * 4 * F + ch;
        const float gi = gp[0], gf = gp[F], gg = gp[2 * F], go = gp[3 * F];
        const float tc = lu_tanh_fast(c_cur[i]);
        float dc = dh * go * (1.f - tc * tc);
        if (dc_in) dc += dc_in[i];
        float* dp = dz + row * 4 * F + ch;
        dp[0] = dc * gg * hsig_grad_from_out(gi);
        dp[F] = dc * c_prev[i] * hsig_grad_from_out(gf);
        dp[2 * F] = dc * gi * (1.f - gg * gg);
        dp[3 * F] = dh * tc * hsig_grad_from_out(go);
        dc_prev_out[i] = dc * gf;
    }
}

// The same step on the bf16 BPTT tape: gates in / dz out (in place) as bf16, four channels per thread (8-byte tape
// accesses, 16-byte fp32 accesses); F % 4 == 0.
__device__ __forceinline__ float bf_lo(unsigned w) { return lu_bits2f(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return lu_bits2f(w & 0xffff0000u); }
__global__ void lstm_gates_bwd_bf16_kernel(unsigned short* gates_dz, const float* __restrict__ c_prev,
                                           const float* __restrict__ c_cur, const float* __restrict__ dh_a, int64_t dha_fs,
                                           const float* __restrict__ dh_b, const float* __restrict__ dc_in,
                                           float* __restrict__ dc_prev_out, int64_t total4, int64_t ppf, int F) {
    const int F4 = F >> 2;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total4; i += (int64_t)gridDim.x * NT) {
        const int64_t row = i / F4;
        const int ch = 4 * (int)(i - row * F4);
        const int64_t f = row / ppf;
        const int64_t e = row * F + ch;
        float4 dh = *reinterpret_cast<const float4*>(dh_a + f * dha_fs + (row - f * ppf) * F + ch);
        if (dh_b) {
            const float4 b = *reinterpret_cast<const float4*>(dh_b + e);
            dh.x += b.x; dh.y += b.y; dh.z += b.z; dh.w += b.w;
        }
        unsigned short* gp = gates_dz + row * 4 * F + ch;
        const lu_u2 ri = *reinterpret_cast<const lu_u2*>(gp), rf = *reinterpret_cast<const lu_u2*>(gp + F),
                    rg = *reinterpret_cast<const lu_u2*>(gp + 2 * F), ro = *reinterpret_cast<const lu_u2*>(gp + 3 * F);
        const float4 cc = *reinterpret_cast<const float4*>(c_cur + e), cp = *reinterpret_cast<const float4*>(c_prev + e);
        float4 dcin = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dc_in) dcin = *reinterpret_cast<const float4*>(dc_in + e);
        float zi[4], zf[4], zg[4], zo[4], dcp[4];
        const float gi[4] = {bf_lo(ri.x), bf_hi(ri.x), bf_lo(ri.y), bf_hi(ri.y)};
        const float gf[4] = {bf_lo(rf.x), bf_hi(rf.x), bf_lo(rf.y), bf_hi(rf.y)};
        const float gg[4] = {bf_lo(rg.x), bf_hi(rg.x), bf_lo(rg.y), bf_hi(rg.y)};
        const float go[4] = {bf_lo(ro.x), bf_hi(ro.x), bf_lo(ro.y), bf_hi(ro.y)};
        const float dhv[4] = {dh.x, dh.y, dh.z, dh.w}, ccv[4] = {cc.x, cc.y, cc.z, cc.w}, cpv[4] = {cp.x, cp.y, cp.z, cp.w},
                    dci[4] = {dcin.x, dcin.y, dcin.z, dcin.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float tc = lu_tanh_fast(ccv[j]);
            const float dc = dhv[j] * go[j] * (1.f - tc * tc) + dci[j];
            zi[j] = dc * gg[j] * hsig_grad_from_out(gi[j]);
            zf[j] = dc * cpv[j] * hsig_grad_from_out(gf[j]);
            zg[j] = dc * gi[j] * (1.f - gg[j] * gg[j]);
            zo[j] = dhv[j] * tc * hsig_grad_from_out(go[j]);
            dcp[j] = dc * gf[j];
        }
        lu_u2 v;
        v.x = lu_pack2bf(zi[0], zi[1]); v.y = lu_pack2bf(zi[2], zi[3]);
        *reinterpret_cast<lu_u2*>(gp) = v;
        v.x = lu_pack2bf(zf[0], zf[1]); v.y = lu_pack2bf(zf[2], zf[3]);
        *reinterpret_cast<lu_u2*>(gp + F) = v;
        v.x = lu_pack2bf(zg[0], zg[1]); v.y = lu_pack2bf(zg[2], zg[3]);
        *reinterpret_cast<lu_u2*>(gp + 2 * F) = v;
        v.x = lu_pack2bf(zo[0], zo[1]); v.y = lu_pack2bf(zo[2], zo[3]);
        *reinterpret_cast<lu_u2*>(gp + 3 * F) = v;
        *reinterpret_cast<float4*>(dc_prev_out + e) = make_float4(dcp[0], dcp[1], dcp[2], dcp[3]);
    }
}

// The fp32 step with the split image of dz written in the same pass (precision 'bf16x3'): dz6 [row][6][4F] bf16, channel blocks in
// order B (hi, mid, lo, hi, mid, hi of the exact three-way bf16 split) -- the operand of the recurrent / input gradient convolutions
// and of the weight gradients -- next to the fp32 dz (in place of the gates: the bias gradient and thin-input layers read it).
// Four channels per thread, 16-byte fp32 accesses, 8-byte bf16 stores; F % 4 == 0.
__global__ void lstm_gates_bwd_split_kernel(float* gates_dz, const float* __restrict__ c_prev, const float* __restrict__ c_cur,
                                            const float* __restrict__ dh_a, int64_t dha_fs, const float* __restrict__ dh_b,
                                            const float* __restrict__ dc_in, unsigned short* __restrict__ dz6,
                                            float* __restrict__ dc_prev_out, int64_t total4, int64_t ppf, int F) {
    const int F4 = F >> 2;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total4; i += (int64_t)gridDim.x * NT) {
        const int64_t row = i / F4;
        const int ch = 4 * (int)(i - row * F4);
        const int64_t f = row / ppf;
        const int64_t e = row * F + ch;
        float4 dh = *reinterpret_cast<const float4*>(dh_a + f * dha_fs + (row - f * ppf) * F + ch);
        if (dh_b) {
            const float4 b = *reinterpret_cast<const float4*>(dh_b + e);
            dh.x += b.x; dh.y += b.y; dh.z += b.z; dh.w += b.w;
        }
        float* gp = gates_dz + row * 4 * F + ch;
        const float4 ri = *reinterpret_cast<const float4*>(gp), rf = *reinterpret_cast<const float4*>(gp + F),
                     rg = *reinterpret_cast<const float4*>(gp + 2 * F), ro = *reinterpret_cast<const float4*>(gp + 3 * F);
        const float4 cc = *reinterpret_cast<const float4*>(c_cur + e), cp = *reinterpret_cast<const float4*>(c_prev + e);
        float4 dcin = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dc_in) dcin = *reinterpret_cast<const float4*>(dc_in + e);
        float z[4][4], dcp[4];      // [gate][channel]
        const float gi[4] = {ri.x, ri.y, ri.z, ri.w}, gf[4] = {rf.x, rf.y, rf.z, rf.w}, gg[4] = {rg.x, rg.y, rg.z, rg.w},
                    go[4] = {ro.x, ro.y, ro.z, ro.w};
        const float dhv[4] = {dh.x, dh.y, dh.z, dh.w}, ccv[4] = {cc.x, cc.y, cc.z, cc.w}, cpv[4] = {cp.x, cp.y, cp.z, cp.w},
                    dci[4] = {dcin.x, dcin.y, dcin.z, dcin.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {      // (the formulas of lstm_gates_bwd_kernel, in its order of operations)
            const float tc = lu_tanh_fast(ccv[j]);
            float dc = dhv[j] * go[j] * (1.f - tc * tc);
            if (dc_in) dc += dci[j];
            z[0][j] = dc * gg[j] * hsig_grad_from_out(gi[j]);
            z[1][j] = dc * cpv[j] * hsig_grad_from_out(gf[j]);
            z[2][j] = dc * gi[j] * (1.f - gg[j] * gg[j]);
            z[3][j] = dhv[j] * tc * hsig_grad_from_out(go[j]);
            dcp[j] = dc * gf[j];
        }
        unsigned short* sp = dz6 + row * (24 * (int64_t)F) + ch;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4*>(gp + g * F) = make_float4(z[g][0], z[g][1], z[g][2], z[g][3]);
            float h[4], m[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) lu_split3(z[g][j], h[j], m[j], l[j]);
            lu_u2 vh, vm, vl;
            vh.x = lu_pack2bf(h[0], h[1]); vh.y = lu_pack2bf(h[2], h[3]);
            vm.x = lu_pack2bf(m[0], m[1]); vm.y = lu_pack2bf(m[2], m[3]);
            vl.x = lu_pack2bf(l[0], l[1]); vl.y = lu_pack2bf(l[2], l[3]);
            unsigned short* q = sp + g * F;      // block j of the pixel at + j * 4F: hi, mid, lo, hi, mid, hi
            *reinterpret_cast<lu_u2*>(q) = vh;
            *reinterpret_cast<lu_u2*>(q + 4 * F) = vm;
            *reinterpret_cast<lu_u2*>(q + 8 * F) = vl;
            *reinterpret_cast<lu_u2*>(q + 12 * F) = vh;
            *reinterpret_cast<lu_u2*>(q + 16 * F) = vm;
            *reinterpret_cast<lu_u2*>(q + 20 * F) = vh;
        }
        *reinterpret_cast<float4*>(dc_prev_out + e) = make_float4(dcp[0], dcp[1], dcp[2], dcp[3]);
    }
}

__global__ void convert_f32_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) y[i] = lu_f2bf(x[i]);
}
__global__ void convert_bf16_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        y[i] = lu_bits2f((unsigned)x[i] << 16);
}

// ---------------------------------------------------------------------------------------------
// Three-way bf16 split of fp32 values (precision = 'bf16x3'): x = hi + mid + lo EXACTLY, hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid) (both residuals are exact in fp32; the last one has at most 8 significant bits left).  A product
// of two fp32 numbers is then the sum of nine bf16 x bf16 products (each exact in fp32); the six of relative size >= 2^-18
// -- hi hi, hi mid, mid hi, hi lo, mid mid, lo hi -- carry it to 2^-26, below fp32's own rounding unit.  The bf16 MFMA
// kernels form those six products by themselves when the REDUCTION axis of the GEMM is laid out six times: an activation
// row [L] becomes [6][Lp] in block order A = (lo, mid, hi, mid, hi, hi), a weight row in block order B = (hi, mid, lo, hi,
// mid, hi) -- block j of one times block j of the other runs through the six products, small terms first (they are summed
// before the accumulator is large).  16x the fp32 MFMA rate / 6 products = 2.7x the fp32 MFMA roofline at fp32 accuracy.
//   x [rows][L] (row stride xs) -> y [rows][6][Lp] (row stride ys), zero for l in [L, Lp); OUT32: the pieces as fp32 values
//   (weights on their way into the fragment packers, which round exactly-representable values without changing them).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void split3(float x, float& hi, float& mid, float& lo) { lu_split3(x, hi, mid, lo); }
__device__ __forceinline__ float split_pick(int piece, float hi, float mid, float lo) { return piece == 0 ? hi : (piece == 1 ? mid : lo); }
// piece (0 hi, 1 mid, 2 lo) held by block j: order A = 2 1 0 1 0 0, order B = 0 1 2 0 1 0 (two bits per block, block 0 lowest)
constexpr unsigned SPLIT_ORDER_A = 2u | (1u << 2) | (0u << 4) | (1u << 6) | (0u << 8) | (0u << 10);
constexpr unsigned SPLIT_ORDER_B = 0u | (1u << 2) | (2u << 4) | (0u << 6) | (1u << 8) | (0u << 10);

template <bool OUT32>
__global__ void split6_vec4_kernel(const float* __restrict__ x, int64_t xs, void* __restrict__ yv, int64_t ys, int64_t rows, int L4,
                                   int Lp, unsigned order) {
    // one thread per (row, four consecutive l): a 16-byte read, six 8-byte (bf16) or 16-byte (fp32) writes
    const int64_t total = rows * L4;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t r = i / L4;
        const int q = (int)(i - r * L4);
        const float4 v = *reinterpret_cast<const float4*>(x + r * xs + 4 * q);
        float h[4], m[4], l[4];
        split3(v.x, h[0], m[0], l[0]);
        split3(v.y, h[1], m[1], l[1]);
        split3(v.z, h[2], m[2], l[2]);
        split3(v.w, h[3], m[3], l[3]);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int pc = (int)((order >> (2 * j)) & 3u);
            const float a = split_pick(pc, h[0], m[0], l[0]), b = split_pick(pc, h[1], m[1], l[1]);
            const float c = split_pick(pc, h[2], m[2], l[2]), d = split_pick(pc, h[3], m[3], l[3]);
            if (OUT32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(yv) + r * ys + (int64_t)j * Lp + 4 * q) = make_float4(a, b, c, d);
            } else {
                lu_u2 o;
                o.x = lu_pack2bf(a, b);
                o.y = lu_pack2bf(c, d);
                *reinterpret_cast<lu_u2*>(reinterpret_cast<unsigned short*>(yv) + r * ys + (int64_t)j * Lp + 4 * q) = o;
            }
        }
    }
}

template <bool OUT32>
__global__ void split6_kernel(const float* __restrict__ x, int64_t xs, void* __restrict__ yv, int64_t ys, int64_t rows, int L, int Lp,
                              unsigned order) {
    // any L / Lp (thin inputs: the one-channel image padded to 4 channels per block): one thread per (row, l < Lp)
    const int64_t total = rows * Lp;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int64_t r = i / Lp;
        const int l = (int)(i - r * Lp);
        float h = 0.f, m = 0.f, lo = 0.f;
        if (l < L) split3(x[r * xs + l], h, m, lo);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float v = split_pick((int)((order >> (2 * j)) & 3u), h, m, lo);
            if (OUT32) reinterpret_cast<float*>(yv)[r * ys + (int64_t)j * Lp + l] = v;
            else reinterpret_cast<unsigned short*>(yv)[r * ys + (int64_t)j * Lp + l] = lu_f2bf(v);
        }
    }
}

// y[f, oy, ox, (kh*k + kw)*C + c] = bf16(x[f, oy+kh-p, ox+kw-p, c]); 32 bf16 per pixel (zero beyond k*k*C); one thread per
// (pixel, pair of output channels)
__global__ void im2col_bf16_kernel(const float* __restrict__ x, unsigned* __restrict__ y, int64_t total, int H, int W, int C,
                                   int k) {
    const int p = (k - 1) / 2, kkc = k * k * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int j2 = (int)(i & 15);
        const int64_t pix = i >> 4;
        const int ox = (int)(pix % W);
        const int64_t t = pix / W;
        const int oy = (int)(t % H);
        const int64_t f = t / H;
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = 2 * j2 + e;
            const int tap = j / C, c = j - tap * C;
            const int iy = oy + tap / k - p, ix = ox + tap % k - p;
            v[e] = (j < kkc && iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((f * H + iy) * W + ix) * C + c] : 0.f;
        }
        y[i] = lu_pack2bf(v[0], v[1]);
    }
}

// ---------------------------------------------------------------------------------------------
// two-stage column reduction:  partial[blk][q][c] (double), q < 2
// block = 64 channels x 4 row lanes
// ---------------------------------------------------------------------------------------------
struct ColPlan {
    int64_t rows_per_blk;
    int nblk, ctiles;
};
inline ColPlan col_plan(int64_t rows, int C) {
    ColPlan p;
    p.ctiles = (C + 63) / 64;
    int64_t target = 768 / p.ctiles;
    if (target < 1) target = 1;
    p.rows_per_blk = (rows + target - 1) / target;
    if (p.rows_per_blk < 64) p.rows_per_blk = 64;
    p.rows_per_blk = (p.rows_per_blk + 3) / 4 * 4;
    p.nblk = (int)((rows + p.rows_per_blk - 1) / p.rows_per_blk);
    if (p.nblk < 1) p.nblk = 1;
    return p;
}

struct FnSum {
    const float* x;
    int ld;
    __device__ void operator()(int64_t r, int c, float& v0, float& v1) const {
        v0 = x[r * ld + c];
        v1 = 0.f;
    }
};
struct FnStats {
    const float* x;
    int ld;
    __device__ void operator()(int64_t r, int c, float& v0, float& v1) const {
        float v = x[r * ld + c];
        v0 = v;
        v1 = v * v;
    }
};
struct FnBnBwd {
    const float *x, *dy, *scale, *shift, *mean, *invstd;
    float alpha;
    int ld;
    __device__ void operator()(int64_t r, int c, float& v0, float& v1) const {
        float xv = x[r * ld + c];
        float zz = xv * scale[c] + shift[c];
        float dz = dy[r * ld + c] * (zz > 0.f ? 1.f : alpha);
        v0 = dz;
        v1 = dz * (xv - mean[c]) * invstd[c];
    }
};

template <class Fn>
__global__ void colreduce_kernel(Fn fn, int64_t rows, int C, int64_t rows_per_blk, double* __restrict__ partial) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > rows) r1 = rows;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            float v0, v1;
            fn(r, c, v0, v1);
            s0 += v0;
            s1 += v1;
        }
    }
    red[0][rl][cl] = s0;
    red[1][rl][cl] = s1;
    __syncthreads();
    if (rl == 0 && c < C) {
        double t0 = (double)red[0][0][cl] + (double)red[0][1][cl] + (double)red[0][2][cl] + (double)red[0][3][cl];
        double t1 = (double)red[1][0][cl] + (double)red[1][1][cl] + (double)red[1][2][cl] + (double)red[1][3][cl];
        partial[((int64_t)blockIdx.x * 2 + 0) * C + c] = t0;
        partial[((int64_t)blockIdx.x * 2 + 1) * C + c] = t1;
    }
}

// mode 0: out_d[q*C + c] = sum ; mode 1: out_f[c] = beta*out_f[c] + sum(q=0)
// block = 16 columns x 16 partial-block lanes (fixed summation order): a column's few hundred partials are 16 short
// dependent chains instead of 4 long ones -- the kernel is pure latency (26 us -> ~8 us per BatchNorm pass).
__global__ void colreduce_final_kernel(const double* __restrict__ partial, int nblk, int C, double* out_d, float* out_f,
                                       float beta, int mode) {
    // block = 8 flattened (q, c) entries x 32 partial-block lanes, fixed summation order: a column's up to ~1000 partials are
    // 32 short dependent chains (16 lanes x 16 columns took 41 us per call at 2048 partial blocks)
    __shared__ double red[32][9];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int nq = mode == 0 ? 2 : 1;
    const int i = blockIdx.x * 8 + cl;          // flattened (q, c)
    const bool ok = i < nq * C;
    const int q = ok ? i / C : 0, c = ok ? i - q * C : 0;
    double s = 0.0;
    if (ok)
        for (int b = rl; b < nblk; b += 32) s += partial[((int64_t)b * 2 + q) * C + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && ok) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 32; j += 4) t += (red[j][cl] + red[j + 1][cl]) + (red[j + 2][cl] + red[j + 3][cl]);
        if (mode == 0) out_d[i] = t;
        else out_f[c] = (beta != 0.f ? beta * out_f[c] : 0.f) + (float)t;
    }
}

// ---- 16-byte variant of the column reduction (C % 4 == 0, 16-byte aligned rows): a thread owns FOUR consecutive columns of its
// row lane, so a narrow tensor (C = 32: the 256^2 decoder tail) keeps every lane busy -- the scalar form above gives a lane one
// column, i.e. half the block idle at C = 32 and 128-byte wave loads (0.9 TB/s measured on those layers); per-column constants
// (BatchNorm scale / shift / mean / invstd) are loaded once per thread instead of once per element.
// block = LPR column quads x (256 / LPR) row lanes; partial[blk][q][c] as above; fixed summation order.
struct ColPlan4 {
    int64_t rows_per_blk;
    int nblk, ctiles, lpr;
};
inline ColPlan4 col_plan4(int64_t rows, int C) {
    ColPlan4 p;
    const int g = C / 4;
    p.lpr = 1;
    while (p.lpr < g && p.lpr < 64) p.lpr *= 2;
    p.ctiles = (g + 63) / 64;
    const int rl = NT / p.lpr;
    int64_t target = 1024 / p.ctiles;      // (the final pass walks the partial blocks: 2048 of them cost it 41 us per call)
    if (target < 1) target = 1;
    p.rows_per_blk = (rows + target - 1) / target;
    if (p.rows_per_blk < 4 * rl) p.rows_per_blk = 4 * rl;
    p.rows_per_blk = (p.rows_per_blk + rl - 1) / rl * rl;
    p.nblk = (int)((rows + p.rows_per_blk - 1) / p.rows_per_blk);
    if (p.nblk < 1) p.nblk = 1;
    return p;
}

struct FnStats4 {
    const float* x;
    int ld;
    __device__ void init(int) {}
    __device__ void operator()(int64_t r, int c, float4& v0, float4& v1) const {
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
        v0 = v;
        v1 = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
    }
};
struct FnBnBwd4 {
    const float *x, *dy, *scale, *shift, *mean, *invstd;
    float alpha;
    int ld;
    float4 sc, sh, mu, is;
    __device__ void init(int c) {
        sc = *reinterpret_cast<const float4*>(scale + c);
        sh = *reinterpret_cast<const float4*>(shift + c);
        mu = *reinterpret_cast<const float4*>(mean + c);
        is = *reinterpret_cast<const float4*>(invstd + c);
    }
    __device__ void operator()(int64_t r, int c, float4& v0, float4& v1) const {
        const float4 xv = *reinterpret_cast<const float4*>(x + r * ld + c);
        const float4 dv = *reinterpret_cast<const float4*>(dy + r * ld + c);
#define LU_BNB(m)                                                  \
    {                                                              \
        const float zz = xv.m * sc.m + sh.m;                       \
        const float dz = dv.m * (zz > 0.f ? 1.f : alpha);          \
        v0.m = dz;                                                 \
        v1.m = dz * (xv.m - mu.m) * is.m;                          \
    }
        LU_BNB(x) LU_BNB(y) LU_BNB(z) LU_BNB(w)
#undef LU_BNB
    }
};

template <class Fn4>
__global__ void colreduce4_kernel(Fn4 fn, int64_t rows, int C, int64_t rows_per_blk, int lpr, double* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float red[2][NT * 4];
    const int cl = threadIdx.x % lpr, rl = threadIdx.x / lpr, nrl = NT / lpr;
    const int c = (blockIdx.y * 64 + cl) * 4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > rows) r1 = rows;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (c < C) {
        fn.init(c);
        for (int64_t r = r0 + rl; r < r1; r += nrl) {
            float4 v0, v1;
            fn(r, c, v0, v1);
            s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
            s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
        }
    }
    *reinterpret_cast<float4*>(&red[0][(rl * lpr + cl) * 4]) = s0;
    *reinterpret_cast<float4*>(&red[1][(rl * lpr + cl) * 4]) = s1;
    __syncthreads();
    // lpr * 4 columns x 2 sums, each over the nrl row lanes in order
    for (int j = threadIdx.x; j < 2 * lpr * 4; j += NT) {
        const int q = j / (lpr * 4), cc = j - q * (lpr * 4);
        const int col = blockIdx.y * 256 + cc;
        if (col >= C) continue;
        double t = 0.0;
        for (int k = 0; k < nrl; ++k) t += (double)red[q][k * lpr * 4 + cc];
        partial[((int64_t)blockIdx.x * 2 + q) * C + col] = t;
    }
}

template <class Fn4>
int run_colreduce4(Fn4 fn, int64_t rows, int C, void* ws, double* out_d, lu_stream_t stream) {
    const ColPlan4 p = col_plan4(rows, C);
    LU_LAUNCH((colreduce4_kernel<Fn4>), dim3(p.nblk, p.ctiles), dim3(NT), stream, fn, rows, C, p.rows_per_blk, p.lpr,
              (double*)ws);
    int rc = LU_CHECK_LAUNCH();
    if (rc) return rc;
    LU_LAUNCH(colreduce_final_kernel, dim3((2 * C + 7) / 8), dim3(NT), stream, (const double*)ws, p.nblk, C, out_d,
              (float*)nullptr, 0.f, 0);
    return LU_CHECK_LAUNCH();
}

inline bool vec4_ok(const void* a, const void* b, int C) {
    return C % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

template <class Fn>
int run_colreduce(Fn fn, int64_t rows, int C, void* ws, double* out_d, float* out_f, float beta, int mode,
                  lu_stream_t stream) {
    ColPlan p = col_plan(rows, C);
    LU_LAUNCH((colreduce_kernel<Fn>), dim3(p.nblk, p.ctiles), dim3(NT), stream, fn, rows, C, p.rows_per_blk,
              (double*)ws);
    int rc = LU_CHECK_LAUNCH();
    if (rc) return rc;
    LU_LAUNCH(colreduce_final_kernel, dim3((2 * C + 7) / 8), dim3(NT), stream, (const double*)ws, p.nblk, C,
              out_d, out_f, beta, mode);
    return LU_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------
// BatchNorm finalize / apply / backward-apply
// ---------------------------------------------------------------------------------------------
__global__ void bn_finalize_train_kernel(const double* __restrict__ sums, double count, const float* gamma,
                                         const float* beta, float eps, float momentum, float* moving_mean,
                                         float* moving_var, float* scale, float* shift, float* save_mean,
                                         float* save_invstd, int C) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    const double mean = sums[c] / count;
    double var = sums[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    if (moving_mean) {
        const double unb = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
        moving_mean[c] = momentum * moving_mean[c] + (1.f - momentum) * (float)mean;
        moving_var[c] = momentum * moving_var[c] + (1.f - momentum) * (float)unb;
    }
}

__global__ void bn_finalize_infer_kernel(const float* gamma, const float* beta, const float* mm, const float* mv,
                                         float eps, float* scale, float* shift, int C) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(mv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - mm[c] * sc;
}

template <bool VEC>
__global__ void bn_lrelu_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ scale,
                                      const float* __restrict__ shift, float alpha, int64_t total, int C) {
    if (VEC) {
        const int64_t n4 = total >> 2;
        const int c4n = C >> 2;
        const bool pow2 = (c4n & (c4n - 1)) == 0;      // channel quad of element i without a 64-bit division
        for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
            const int c = (pow2 ? (int)(i & (c4n - 1)) : (int)(i % c4n)) * 4;
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
            float4 o;
            float t;
            t = fmaf(v.x, sc.x, sh.x);
            o.x = t > 0.f ? t : alpha * t;
            t = fmaf(v.y, sc.y, sh.y);
            o.y = t > 0.f ? t : alpha * t;
            t = fmaf(v.z, sc.z, sh.z);
            o.z = t > 0.f ? t : alpha * t;
            t = fmaf(v.w, sc.w, sh.w);
            o.w = t > 0.f ? t : alpha * t;
            reinterpret_cast<float4*>(y)[i] = o;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
            const int c = (int)(i % C);
            const float t = fmaf(x[i], scale[c], shift[c]);
            y[i] = t > 0.f ? t : alpha * t;
        }
    }
}

// bf16 result (bf16 mode, training: an activation whose every consumer rounds it to a bf16 MFMA operand): C % 4 == 0
__global__ void bn_lrelu_apply_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y,
                                           const float* __restrict__ scale, const float* __restrict__ shift, float alpha,
                                           int64_t total, int C) {
    const int64_t n4 = total >> 2;
    const int c4n = C >> 2;
    const bool pow2 = (c4n & (c4n - 1)) == 0;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
        const int c = (pow2 ? (int)(i & (c4n - 1)) : (int)(i % c4n)) * 4;
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        float4 o;
        float t;
        t = fmaf(v.x, sc.x, sh.x);
        o.x = t > 0.f ? t : alpha * t;
        t = fmaf(v.y, sc.y, sh.y);
        o.y = t > 0.f ? t : alpha * t;
        t = fmaf(v.z, sc.z, sh.z);
        o.z = t > 0.f ? t : alpha * t;
        t = fmaf(v.w, sc.w, sh.w);
        o.w = t > 0.f ? t : alpha * t;
        lu_u2 pk;
        pk.x = lu_pack2bf(o.x, o.y);
        pk.y = lu_pack2bf(o.z, o.w);
        *reinterpret_cast<lu_u2*>(y + i * 4) = pk;
    }
}

__global__ void bn_lrelu_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                          float alpha, const double* __restrict__ sums, double count,
                                          float* __restrict__ dx, float* dgamma, float* dbeta, int64_t total, int C,
                                          int vec4, unsigned short* __restrict__ dx16 = nullptr) {
    if (blockIdx.x == 0 && dgamma) {
        for (int c = threadIdx.x; c < C; c += NT) {
            dbeta[c] = (float)sums[c];
            dgamma[c] = (float)sums[C + c];
        }
    }
    const float inv_n = (float)(1.0 / count);
    if (vec4) {      // 16-byte loads / stores; C % 4 == 0: the four elements are four consecutive channels
        const int64_t n4 = total >> 2;
        const int c4n = C >> 2;
        const bool pow2 = (c4n & (c4n - 1)) == 0;
        for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
            const int c = (pow2 ? (int)(i & (c4n - 1)) : (int)(i % c4n)) * 4;
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            const float4 dv = reinterpret_cast<const float4*>(dy)[i];
            const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
            const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
            const double2 a01 = *reinterpret_cast<const double2*>(sums + c), a23 = *reinterpret_cast<const double2*>(sums + c + 2);
            const double2 b01 = *reinterpret_cast<const double2*>(sums + C + c), b23 = *reinterpret_cast<const double2*>(sums + C + c + 2);
            float4 o;
#define LU_BWD(m, s0_, s1_)                                                                  \
    {                                                                                        \
        const float zz = xv.m * sc.m + sh.m;                                                 \
        const float dz = dv.m * (zz > 0.f ? 1.f : alpha);                                    \
        const float xhat = (xv.m - mu.m) * is.m;                                             \
        o.m = sc.m * (dz - (float)(s0_) * inv_n - xhat * (float)(s1_) * inv_n);              \
    }
            LU_BWD(x, a01.x, b01.x) LU_BWD(y, a01.y, b01.y) LU_BWD(z, a23.x, b23.x) LU_BWD(w, a23.y, b23.y)
#undef LU_BWD
            if (dx16) {      // bf16 result (vec4 only): every consumer rounds dx to bf16 MFMA operands anyway
                lu_u2 b;
                b.x = lu_pack2bf(o.x, o.y);
                b.y = lu_pack2bf(o.z, o.w);
                reinterpret_cast<lu_u2*>(dx16)[i] = b;
            } else {
                reinterpret_cast<float4*>(dx)[i] = o;
            }
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        const float xv = x[i];
        const float zz = xv * scale[c] + shift[c];
        const float dz = dy[i] * (zz > 0.f ? 1.f : alpha);
        const float xhat = (xv - mean[c]) * invstd[c];
        dx[i] = scale[c] * (dz - (float)sums[c] * inv_n - xhat * (float)sums[C + c] * inv_n);
    }
}

// ---------------------------------------------------------------------------------------------
// bilinear x2, edge clamp.  Two source-coordinate conventions (k.backend.resize_images(..., 'bilinear'), Networks.py:143,
// depends on the TensorFlow release):
//   legacy = 1  src = o / 2          (out[2i] = in[i], out[2i+1] = (in[i] + in[i+1]) / 2): the v1 resize_bilinear op with
//               align_corners=False, half_pixel_centers=False -- what keras.backend.resize_images calls in TF 2.0 / 2.1,
//               i.e. in the release the reference pins (README: tensorflow 2.0.0a0)
//   legacy = 0  src = (o + 0.5) / 2 - 0.5   half-pixel centres: tf.image.resize (v2), used by later Keras releases
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void up2_taps(int o, int n_in, int legacy, int& lo, int& hi, float& frac) {
    const float src = legacy ? o * 0.5f : (o + 0.5f) * 0.5f - 0.5f;
    const float fl = floorf(src);
    frac = src - fl;
    const int i0 = (int)fl;
    lo = i0 < 0 ? 0 : (i0 > n_in - 1 ? n_in - 1 : i0);
    hi = i0 + 1 < 0 ? 0 : (i0 + 1 > n_in - 1 ? n_in - 1 : i0 + 1);
}

// VW = 4: four channels per thread (C % 4 == 0, 16-byte aligned): 16-byte loads / stores and a quarter of the index
// arithmetic -- the same per-element expressions as VW = 1, so both produce the same bits.
// OB16: the result is stored as bf16 (round to nearest even after the fp32 interpolation) -- bf16 mode, where the only consumer is
// a convolution that rounds its operands to bf16 anyway: same values, half the bytes written and read.
template <int VW, bool OB16 = false>
__global__ void upsample2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int frames, int H, int W,
                                      int C, int legacy) {
    const int CV = C / VW;
    const int64_t total = (int64_t)frames * 4 * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % CV) * VW;
        int64_t t = i / CV;
        const int ox = (int)(t % (2 * W));
        t /= 2 * W;
        const int oy = (int)(t % (2 * H));
        const int64_t f = t / (2 * H);
        int ylo, yhi, xlo, xhi;
        float fy, fx;
        up2_taps(oy, H, legacy, ylo, yhi, fy);
        up2_taps(ox, W, legacy, xlo, xhi, fx);
        const float* xf = x + f * (int64_t)H * W * C + c;
        const float* p00 = xf + ((int64_t)ylo * W + xlo) * C;
        const float* p01 = xf + ((int64_t)ylo * W + xhi) * C;
        const float* p10 = xf + ((int64_t)yhi * W + xlo) * C;
        const float* p11 = xf + ((int64_t)yhi * W + xhi) * C;
        float v00[VW], v01[VW], v10[VW], v11[VW], o[VW];
        if (VW == 4) {
            *reinterpret_cast<float4*>(v00) = *reinterpret_cast<const float4*>(p00);
            *reinterpret_cast<float4*>(v01) = *reinterpret_cast<const float4*>(p01);
            *reinterpret_cast<float4*>(v10) = *reinterpret_cast<const float4*>(p10);
            *reinterpret_cast<float4*>(v11) = *reinterpret_cast<const float4*>(p11);
        } else {
            v00[0] = *p00;
            v01[0] = *p01;
            v10[0] = *p10;
            v11[0] = *p11;
        }
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const float top = v00[e] * (1.f - fx) + v01[e] * fx, bot = v10[e] * (1.f - fx) + v11[e] * fx;
            o[e] = top * (1.f - fy) + bot * fy;
        }
        if (OB16) {
            lu_u2 v;
            v.x = lu_pack2bf(o[0], o[VW > 1 ? 1 : 0]);
            v.y = lu_pack2bf(o[VW > 2 ? 2 : 0], o[VW > 3 ? 3 : 0]);
            *reinterpret_cast<lu_u2*>(reinterpret_cast<unsigned short*>(y) + i * 4) = v;      // (VW = 4 only)
        } else if (VW == 4)
            *reinterpret_cast<float4*>(y + i * 4) = *reinterpret_cast<const float4*>(o);
        else
            y[i] = o[0];
    }
}

template <int VW>
__global__ void upsample2x_bwd_kernel(const float* __restrict__ dy, int dy_ps, float* __restrict__ dx, int frames,
                                      int H, int W, int C, int legacy) {
    const int CV = C / VW;
    const int64_t total = (int64_t)frames * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % CV) * VW;
        int64_t t = i / CV;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const int64_t f = t / H;
        const float* df = dy + f * (int64_t)4 * H * W * dy_ps + c;
        float acc[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) acc[e] = 0.f;
        for (int oy = 2 * iy - 1; oy <= 2 * iy + 2; ++oy) {
            if (oy < 0 || oy >= 2 * H) continue;
            int ylo, yhi;
            float fy;
            up2_taps(oy, H, legacy, ylo, yhi, fy);
            const float wy = (ylo == iy ? 1.f - fy : 0.f) + (yhi == iy ? fy : 0.f);
            if (wy == 0.f) continue;
            for (int ox = 2 * ix - 1; ox <= 2 * ix + 2; ++ox) {
                if (ox < 0 || ox >= 2 * W) continue;
                int xlo, xhi;
                float fx;
                up2_taps(ox, W, legacy, xlo, xhi, fx);
                const float wx = (xlo == ix ? 1.f - fx : 0.f) + (xhi == ix ? fx : 0.f);
                if (wx == 0.f) continue;
                const float* p = df + ((int64_t)oy * 2 * W + ox) * dy_ps;
                float v[VW];
                if (VW == 4)
                    *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(p);
                else
                    v[0] = *p;
#pragma unroll
                for (int e = 0; e < VW; ++e) acc[e] += wy * wx * v[e];
            }
        }
        if (VW == 4)
            *reinterpret_cast<float4*>(dx + i * 4) = *reinterpret_cast<const float4*>(acc);
        else
            dx[i] = acc[0];
    }
}

// ---------------------------------------------------------------------------------------------
// window copy (reflect pad / zero embed / crop)
// ---------------------------------------------------------------------------------------------
__global__ void window_copy_kernel(const float* __restrict__ x, int x_ps, float* __restrict__ y, int frames, int Hx,
                                   int Wx, int Hy, int Wy, int C, int off_y, int off_x, int mode, float beta) {
    const int64_t total = (int64_t)frames * Hy * Wy * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ox = (int)(t % Wy);
        t /= Wy;
        const int oy = (int)(t % Hy);
        const int64_t f = t / Hy;
        int sy = oy - off_y, sx = ox - off_x;
        bool ok = true;
        if (mode == 1) {
            if (sy < 0) sy = -sy;
            if (sy >= Hx) sy = 2 * (Hx - 1) - sy;
            if (sx < 0) sx = -sx;
            if (sx >= Wx) sx = 2 * (Wx - 1) - sx;
        } else {
            ok = sy >= 0 && sy < Hx && sx >= 0 && sx < Wx;
        }
        const float v = ok ? x[((f * Hx + sy) * Wx + sx) * x_ps + c] : 0.f;
        y[i] = (beta != 0.f ? beta * y[i] : 0.f) + v;
    }
}

// 16-byte form (C, the pixel stride and both pointers in 16-byte groups, beta == 0): one thread per (pixel, four channels) -- a quarter of
// the index arithmetic and full-width memory transactions.  Round 6: precision 'bf16x3' pads the split tensors of ragged-width levels to
// W % 32 == 0 for the kernel-row weight gradient with this kernel (a bf16 [.., 6 C] row seen as 3 C floats); as a torch strided copy of
// 2-byte elements that was 74 ms of a config-4 step (round-5 verdict, weak #7).
__global__ void window_copy_vec4_kernel(const float4* __restrict__ x, int x_ps4, float4* __restrict__ y, int frames, int Hx, int Wx, int Hy,
                                        int Wy, int C4, int off_y, int off_x, int mode) {
    const int64_t total = (int64_t)frames * Hy * Wy * C4;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C4);
        int64_t t = i / C4;
        const int ox = (int)(t % Wy);
        t /= Wy;
        const int oy = (int)(t % Hy);
        const int64_t f = t / Hy;
        int sy = oy - off_y, sx = ox - off_x;
        bool ok = true;
        if (mode == 1) {
            if (sy < 0) sy = -sy;
            if (sy >= Hx) sy = 2 * (Hx - 1) - sy;
            if (sx < 0) sx = -sx;
            if (sx >= Wx) sx = 2 * (Wx - 1) - sx;
        } else {
            ok = sy >= 0 && sy < Hx && sx >= 0 && sx < Wx;
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = x[((f * Hx + sy) * Wx + sx) * x_ps4 + c];
        y[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// softmax + weighted CE
// ---------------------------------------------------------------------------------------------
__global__ void wce_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ gt, const float* cw,
                               float* __restrict__ sm_out, int64_t rows, double* __restrict__ partial) {
    __shared__ float red[2][NT];
    float s_loss = 0.f, s_valid = 0.f;
    const float w0 = cw[0], w1 = cw[1], w2 = cw[2];
    for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < rows; r += (int64_t)gridDim.x * NT) {
        const float l0 = logits[r * 3], l1 = logits[r * 3 + 1], l2 = logits[r * 3 + 2];
        const float m = fmaxf(l0, fmaxf(l1, l2));
        const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
        const float se = e0 + e1 + e2;
        if (sm_out) {
            const float inv = 1.f / se;
            sm_out[r * 3] = e0 * inv;
            sm_out[r * 3 + 1] = e1 * inv;
            sm_out[r * 3 + 2] = e2 * inv;
        }
        const float g = gt[r];
        if (g > -1.f) {
            const int gi = (int)g;
            const float w = gi == 0 ? w0 : (gi == 1 ? w1 : (gi == 2 ? w2 : 0.f));
            const float picked = gi <= 0 ? l0 : (gi == 1 ? l1 : l2);
            s_loss += (logf(se) + m - picked) * w;
            s_valid += 1.f;
        }
    }
    red[0][threadIdx.x] = s_loss;
    red[1][threadIdx.x] = s_valid;
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0.0;
        for (int i = 0; i < NT; ++i) t += (double)red[threadIdx.x][i];
        partial[(int64_t)blockIdx.x * 2 + threadIdx.x] = t;
    }
}

// sums[q] = sum over the blocks' partials, q = 0 (loss) / 1 (weight): 32 lanes per sum, fixed order (two threads walking
// 2048 partials one by one took 177 us)
__global__ void wce_final_kernel(const double* __restrict__ partial, int nblk, double* sums) {
    __shared__ double red[64];
    const int q = threadIdx.x & 1, l = threadIdx.x >> 1;      // 64 threads: 32 lanes x 2 sums
    double t = 0.0;
    for (int b = l; b < nblk; b += 32) t += partial[(int64_t)b * 2 + q];
    red[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x < 2) {
        double s = 0.0;
        for (int j = 0; j < 32; ++j) s += red[2 * j + threadIdx.x];
        sums[threadIdx.x] = s;
    }
}

__global__ void wce_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ gt, const float* cw,
                               const double* __restrict__ sums, float grad_scale, float* __restrict__ dl,
                               int64_t rows) {
    const float inv = grad_scale / (float)(sums[1] + 0.00001);
    const float w0 = cw[0], w1 = cw[1], w2 = cw[2];
    for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < rows; r += (int64_t)gridDim.x * NT) {
        const float g = gt[r];
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
        if (g > -1.f) {
            const int gi = (int)g;
            const float w = (gi == 0 ? w0 : (gi == 1 ? w1 : (gi == 2 ? w2 : 0.f))) * inv;
            const float l0 = logits[r * 3], l1 = logits[r * 3 + 1], l2 = logits[r * 3 + 2];
            const float m = fmaxf(l0, fmaxf(l1, l2));
            const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
            const float is = 1.f / (e0 + e1 + e2);
            d0 = w * (e0 * is - (gi <= 0 ? 1.f : 0.f));
            d1 = w * (e1 * is - (gi == 1 ? 1.f : 0.f));
            d2 = w * (e2 * is - (gi >= 2 ? 1.f : 0.f));
        }
        dl[r * 3] = d0;
        dl[r * 3 + 1] = d1;
        dl[r * 3 + 2] = d2;
    }
}

__global__ void softmax3_kernel(const float* __restrict__ logits, float* __restrict__ out, int64_t rows) {
    for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < rows; r += (int64_t)gridDim.x * NT) {
        const float l0 = logits[r * 3], l1 = logits[r * 3 + 1], l2 = logits[r * 3 + 2];
        const float m = fmaxf(l0, fmaxf(l1, l2));
        const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
        const float inv = 1.f / (e0 + e1 + e2);
        out[r * 3] = e0 * inv;
        out[r * 3 + 1] = e1 * inv;
        out[r * 3 + 2] = e2 * inv;
    }
}

__global__ void wce_loss_kernel(const double* sums, float* loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) loss[0] = (float)(sums[0] / (sums[1] + 0.00001));
}

// ---------------------------------------------------------------------------------------------
// Adam, state mask, transposes, add
// ---------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float alpha, float b1, float b2, float eps, float gs) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float gv = g[i] * gs;
        const float mn = b1 * m[i] + (1.f - b1) * gv;
        const float vn = b2 * v[i] + (1.f - b2) * gv * gv;
        m[i] = mn;
        v[i] = vn;
        p[i] = p[i] - alpha * mn / (sqrtf(vn) + eps);
    }
}

__global__ void scale_frames_kernel(float* __restrict__ x, const float* __restrict__ keep, int64_t per_frame,
                                    int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT)
        x[i] *= keep[i / per_frame];
}

// Start of a training window for one ConvLSTM state tensor, one pass instead of three (keep mask in place, copy into slot 0 of
// the tape, bf16 copy):  dst[f, :] = src ? src[f, :] * keep[f] : 0, dst16 = bf16(dst).  blockIdx.y = frame; VEC: float4 lanes.
template <bool VEC>
__global__ void state_begin_kernel(float* __restrict__ dst, unsigned short* __restrict__ dst16, const float* __restrict__ src,
                                   const float* __restrict__ keep, int64_t per_frame) {
    const int f = blockIdx.y;
    const float kf = (src && keep) ? keep[f] : 1.f;
    const int64_t base = (int64_t)f * per_frame;
    if (VEC) {
        const int64_t n4 = per_frame >> 2;
        for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src) {
                v = *reinterpret_cast<const float4*>(src + base + 4 * i);
                if (keep) { v.x *= kf; v.y *= kf; v.z *= kf; v.w *= kf; }
            }
            *reinterpret_cast<float4*>(dst + base + 4 * i) = v;
            if (dst16) {
                lu_u2 b;
                b.x = lu_pack2bf(v.x, v.y);
                b.y = lu_pack2bf(v.z, v.w);
                *reinterpret_cast<lu_u2*>(dst16 + base + 4 * i) = b;
            }
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < per_frame; i += (int64_t)gridDim.x * NT) {
            float v = 0.f;
            if (src) v = keep ? src[base + i] * kf : src[base + i];
            dst[base + i] = v;
            if (dst16) dst16[base + i] = lu_f2bf(v);
        }
    }
}

__global__ void transpose_inner_kernel(const float* __restrict__ x, float* __restrict__ y, int a, int b) {
    __shared__ float tile[32][33];
    const int64_t n = blockIdx.z;
    const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* xn = x + n * (int64_t)a * b;
    float* yn = y + n * (int64_t)a * b;
    for (int r = ty; r < 32; r += 8)
        if (a0 + r < a && b0 + tx < b) tile[r][tx] = xn[(int64_t)(a0 + r) * b + b0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (b0 + r < b && a0 + tx < a) yn[(int64_t)(b0 + r) * a + a0 + tx] = tile[tx][r];
}

__global__ void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) y[i] += x[i];
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" int lu_lstm_gates_fwd(const float* z, const float* c_prev, float* c_out, float* h_out, float* gates_out,
                                 int32_t frames, int64_t ppf, int32_t F, int64_t h_fs, lu_stream_t stream) {
    LU_REQUIRE(z && c_prev && c_out && h_out && frames > 0 && ppf > 0 && F > 0, "lu_lstm_gates_fwd: bad arguments");
    const int64_t total = (int64_t)frames * ppf * F;
    LU_LAUNCH(lstm_gates_fwd_kernel, dim3(grid_for(total)), dim3(NT), stream, z, c_prev, c_out, h_out, gates_out,
              total, ppf, (int)F, h_fs);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_lstm_gates_fwd_slabs(const float* slabs, int32_t splits, const float* bias, const float* c_prev,
                                       float* c_out, float* h_out, float* gates_out, int32_t frames, int64_t ppf, int32_t F,
                                       int64_t h_fs, lu_stream_t stream) {
    LU_REQUIRE(slabs && splits > 0 && c_prev && c_out && h_out && frames > 0 && ppf > 0 && F > 0,
               "lu_lstm_gates_fwd_slabs: bad arguments");
    const int64_t total = (int64_t)frames * ppf * F;
    LU_LAUNCH(lstm_gates_fwd_slabs_kernel, dim3(grid_for(total)), dim3(NT), stream, slabs, (int)splits, total * 4, bias, c_prev,
              c_out, h_out, gates_out, total, ppf, (int)F, h_fs);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_lstm_gates_bwd(const float* gates, const float* c_prev, const float* c_cur, const float* dh_a,
                                 int64_t dh_a_fs, const float* dh_b, const float* dc_in, float* dz,
                                 float* dc_prev_out, int32_t frames, int64_t ppf, int32_t F, lu_stream_t stream) {
    LU_REQUIRE(gates && c_prev && c_cur && dh_a && dz && dc_prev_out && frames > 0 && ppf > 0 && F > 0,
               "lu_lstm_gates_bwd: bad arguments");
    const int64_t total = (int64_t)frames * ppf * F;
    LU_LAUNCH(lstm_gates_bwd_kernel, dim3(grid_for(total)), dim3(NT), stream, gates, c_prev, c_cur, dh_a, dh_a_fs,
              dh_b, dc_in, dz, dc_prev_out, total, ppf, (int)F);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_lstm_gates_bwd_split(float* gates_dz, const float* c_prev, const float* c_cur, const float* dh_a,
                                       int64_t dh_a_fs, const float* dh_b, const float* dc_in, void* dz6, float* dc_prev_out,
                                       int32_t frames, int64_t ppf, int32_t F, lu_stream_t stream) {
    LU_REQUIRE(gates_dz && c_prev && c_cur && dh_a && dz6 && dc_prev_out && frames > 0 && ppf > 0 && F > 0 && F % 4 == 0 &&
                   dh_a_fs % 4 == 0,
               "lu_lstm_gates_bwd_split: bad arguments (F %% 4 == 0 required)");
    LU_REQUIRE(((reinterpret_cast<uintptr_t>(gates_dz) | reinterpret_cast<uintptr_t>(c_prev) | reinterpret_cast<uintptr_t>(c_cur) |
                 reinterpret_cast<uintptr_t>(dh_a) | reinterpret_cast<uintptr_t>(dh_b) | reinterpret_cast<uintptr_t>(dc_in) |
                 reinterpret_cast<uintptr_t>(dc_prev_out)) & 15) == 0 && (reinterpret_cast<uintptr_t>(dz6) & 7) == 0,
               "lu_lstm_gates_bwd_split: 16-byte aligned fp32 tensors, 8-byte aligned dz6");
    const int64_t total4 = (int64_t)frames * ppf * (F / 4);
    LU_LAUNCH(lstm_gates_bwd_split_kernel, dim3(grid_for(total4)), dim3(NT), stream, gates_dz, c_prev, c_cur, dh_a, dh_a_fs, dh_b,
              dc_in, (unsigned short*)dz6, dc_prev_out, total4, ppf, (int)F);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_lstm_gates_bwd_bf16(void* gates_dz, const float* c_prev, const float* c_cur, const float* dh_a,
                                      int64_t dh_a_fs, const float* dh_b, const float* dc_in, float* dc_prev_out,
                                      int32_t frames, int64_t ppf, int32_t F, lu_stream_t stream) {
    LU_REQUIRE(gates_dz && c_prev && c_cur && dh_a && dc_prev_out && frames > 0 && ppf > 0 && F > 0 && F % 4 == 0 &&
                   dh_a_fs % 4 == 0,
               "lu_lstm_gates_bwd_bf16: bad arguments (F %% 4 == 0 required)");
    const uintptr_t al = reinterpret_cast<uintptr_t>(c_prev) | reinterpret_cast<uintptr_t>(c_cur) |
                         reinterpret_cast<uintptr_t>(dh_a) | reinterpret_cast<uintptr_t>(dh_b) |
                         reinterpret_cast<uintptr_t>(dc_in) | reinterpret_cast<uintptr_t>(dc_prev_out);
    LU_REQUIRE((al & 15) == 0 && (reinterpret_cast<uintptr_t>(gates_dz) & 7) == 0, "lu_lstm_gates_bwd_bf16: unaligned pointers");
    const int64_t total4 = (int64_t)frames * ppf * (F / 4);
    LU_LAUNCH(lstm_gates_bwd_bf16_kernel, dim3(grid_for(total4)), dim3(NT), stream, (unsigned short*)gates_dz, c_prev, c_cur,
              dh_a, dh_a_fs, dh_b, dc_in, dc_prev_out, total4, ppf, (int)F);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_convert_f32_bf16(const float* x, void* y, int64_t n, lu_stream_t stream) {
    LU_REQUIRE(x && y && n > 0, "lu_convert_f32_bf16: bad arguments");
    LU_LAUNCH(convert_f32_bf16_kernel, dim3(grid_for(n, 4)), dim3(NT), stream, x, (unsigned short*)y, n);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_convert_bf16_f32(const void* x, float* y, int64_t n, lu_stream_t stream) {
    LU_REQUIRE(x && y && n > 0, "lu_convert_bf16_f32: bad arguments");
    LU_LAUNCH(convert_bf16_f32_kernel, dim3(grid_for(n, 4)), dim3(NT), stream, (const unsigned short*)x, y, n);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_split6(const float* x, int64_t rows, int32_t L, int64_t x_row_stride, void* y, int64_t y_row_stride,
                        int32_t Lp, int32_t order, int32_t out_dtype, lu_stream_t stream) {
    LU_REQUIRE(x && y && rows > 0 && L > 0 && Lp >= L && x_row_stride >= L && y_row_stride >= 6 * (int64_t)Lp,
               "lu_split6: bad arguments");
    LU_REQUIRE((order == 0 || order == 1) && (out_dtype == LU_F32 || out_dtype == LU_BF16), "lu_split6: order is 0 (A) or 1 (B), out_dtype LU_F32 / LU_BF16");
    const unsigned ord = order == 0 ? SPLIT_ORDER_A : SPLIT_ORDER_B;
    const bool o32 = out_dtype == LU_F32;
    const bool vec = L == Lp && (L & 3) == 0 && (x_row_stride & 3) == 0 && (y_row_stride & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    if (vec) {
        const int64_t total = rows * (L / 4);
        if (o32) {
            LU_LAUNCH(split6_vec4_kernel<true>, dim3(grid_for(total, 2)), dim3(NT), stream, x, x_row_stride, y, y_row_stride, rows, (int)(L / 4), (int)Lp, ord);
        } else {
            LU_LAUNCH(split6_vec4_kernel<false>, dim3(grid_for(total, 2)), dim3(NT), stream, x, x_row_stride, y, y_row_stride, rows, (int)(L / 4), (int)Lp, ord);
        }
    } else {
        const int64_t total = rows * Lp;
        if (o32) {
            LU_LAUNCH(split6_kernel<true>, dim3(grid_for(total, 2)), dim3(NT), stream, x, x_row_stride, y, y_row_stride, rows, (int)L, (int)Lp, ord);
        } else {
            LU_LAUNCH(split6_kernel<false>, dim3(grid_for(total, 2)), dim3(NT), stream, x, x_row_stride, y, y_row_stride, rows, (int)L, (int)Lp, ord);
        }
    }
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_im2col_bf16(const float* x, void* y, int32_t frames, int32_t H, int32_t W, int32_t C, int32_t k,
                              lu_stream_t stream) {
    LU_REQUIRE(x && y && frames > 0 && H > 0 && W > 0 && C > 0 && k >= 1 && (k & 1) && k * k * C <= 32,
               "lu_im2col_bf16: needs an odd k with k*k*C <= 32");
    const int64_t total = (int64_t)frames * H * W * 16;
    LU_LAUNCH(im2col_bf16_kernel, dim3(grid_for(total, 2)), dim3(NT), stream, x, (unsigned*)y, total, (int)H, (int)W, (int)C,
              (int)k);
    return LU_CHECK_LAUNCH();
}

extern "C" size_t lu_colreduce_workspace_bytes(int64_t rows, int32_t C) {
    const ColPlan p = col_plan(rows, C);
    int nblk = p.nblk;
    if (C % 4 == 0) {
        const ColPlan4 p4 = col_plan4(rows, C);
        if (p4.nblk > nblk) nblk = p4.nblk;
    }
    return (size_t)nblk * 2 * C * sizeof(double);
}

extern "C" int lu_colsum(const float* x, int64_t rows, int32_t C, int32_t ld, float* out, float beta, void* ws,
                         lu_stream_t stream) {
    LU_REQUIRE(x && out && ws && rows > 0 && C > 0 && ld >= C, "lu_colsum: bad arguments");
    FnSum fn{x, ld};
    return run_colreduce(fn, rows, C, ws, nullptr, out, beta, 1, stream);
}

extern "C" int lu_bn_stats(const float* x, int64_t rows, int32_t C, double* sums, void* ws, lu_stream_t stream) {
    LU_REQUIRE(x && sums && ws && rows > 0 && C > 0, "lu_bn_stats: bad arguments");
    if (vec4_ok(x, x, C)) {
        FnStats4 fn4{x, C};
        return run_colreduce4(fn4, rows, C, ws, sums, stream);
    }
    FnStats fn{x, C};
    return run_colreduce(fn, rows, C, ws, sums, nullptr, 0.f, 0, stream);
}

extern "C" int lu_bn_finalize_train(const double* sums, double count, const float* gamma, const float* beta,
                                    float eps, float momentum, float* moving_mean, float* moving_var, float* scale,
                                    float* shift, float* save_mean, float* save_invstd, int32_t C,
                                    lu_stream_t stream) {
    LU_REQUIRE(sums && gamma && beta && scale && shift && save_mean && save_invstd && C > 0 && count > 0,
               "lu_bn_finalize_train: bad arguments");
    LU_LAUNCH(bn_finalize_train_kernel, dim3((C + NT - 1) / NT), dim3(NT), stream, sums, count, gamma, beta, eps,
              momentum, moving_mean, moving_var, scale, shift, save_mean, save_invstd, (int)C);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_bn_finalize_infer(const float* gamma, const float* beta, const float* mm, const float* mv, float eps,
                                    float* scale, float* shift, int32_t C, lu_stream_t stream) {
    LU_REQUIRE(gamma && beta && mm && mv && scale && shift && C > 0, "lu_bn_finalize_infer: bad arguments");
    LU_LAUNCH(bn_finalize_infer_kernel, dim3((C + NT - 1) / NT), dim3(NT), stream, gamma, beta, mm, mv, eps, scale,
              shift, (int)C);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_bn_lrelu_apply(const float* x, float* y, const float* scale, const float* shift, float alpha,
                                 int64_t rows, int32_t C, lu_stream_t stream) {
    LU_REQUIRE(x && y && scale && shift && rows > 0 && C > 0, "lu_bn_lrelu_apply: bad arguments");
    const int64_t total = rows * C;
    const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(scale) |
                                       reinterpret_cast<uintptr_t>(shift)) & 15) == 0;
    if (vec)
        LU_LAUNCH((bn_lrelu_apply_kernel<true>), dim3(grid_for(total / 4)), dim3(NT), stream, x, y, scale, shift,
                  alpha, total, (int)C);
    else
        LU_LAUNCH((bn_lrelu_apply_kernel<false>), dim3(grid_for(total)), dim3(NT), stream, x, y, scale, shift, alpha,
                  total, (int)C);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_bn_lrelu_bwd_reduce(const float* x, const float* dy, const float* scale, const float* shift,
                                      const float* save_mean, const float* save_invstd, float alpha, int64_t rows,
                                      int32_t C, double* sums, void* ws, lu_stream_t stream) {
    LU_REQUIRE(x && dy && scale && shift && save_mean && save_invstd && sums && ws && rows > 0 && C > 0,
               "lu_bn_lrelu_bwd_reduce: bad arguments");
    if (vec4_ok(x, dy, C) && vec4_ok(scale, shift, C) && vec4_ok(save_mean, save_invstd, C)) {
        FnBnBwd4 fn4{x, dy, scale, shift, save_mean, save_invstd, alpha, C};
        return run_colreduce4(fn4, rows, C, ws, sums, stream);
    }
    FnBnBwd fn{x, dy, scale, shift, save_mean, save_invstd, alpha, C};
    return run_colreduce(fn, rows, C, ws, sums, nullptr, 0.f, 0, stream);
}

extern "C" int lu_bn_lrelu_bwd_apply(const float* x, const float* dy, const float* scale, const float* shift,
                                     const float* save_mean, const float* save_invstd, float alpha,
                                     const double* sums, double count, float* dx, float* dgamma, float* dbeta,
                                     int64_t rows, int32_t C, lu_stream_t stream) {
    LU_REQUIRE(x && dy && scale && shift && save_mean && save_invstd && sums && dx && rows > 0 && C > 0 && count > 0,
               "lu_bn_lrelu_bwd_apply: bad arguments");
    const int64_t total = rows * C;
    const int vec4 = (vec4_ok(x, dy, C) && vec4_ok(dx, scale, C) && vec4_ok(shift, save_mean, C) && vec4_ok(save_invstd, sums, C)) ? 1 : 0;
    LU_LAUNCH(bn_lrelu_bwd_apply_kernel, dim3(grid_for(vec4 ? total / 4 : total)), dim3(NT), stream, x, dy, scale, shift,
              save_mean, save_invstd, alpha, sums, count, dx, dgamma, dbeta, total, (int)C, vec4);
    return LU_CHECK_LAUNCH();
}

/* lu_bn_lrelu_bwd_apply with a bf16 result (C % 4 == 0, 16-byte aligned operands, 8-byte aligned dx) */
extern "C" int lu_bn_lrelu_bwd_apply_bf16(const float* x, const float* dy, const float* scale, const float* shift,
                                          const float* save_mean, const float* save_invstd, float alpha,
                                          const double* sums, double count, void* dx_bf16, float* dgamma, float* dbeta,
                                          int64_t rows, int32_t C, lu_stream_t stream) {
    LU_REQUIRE(x && dy && scale && shift && save_mean && save_invstd && sums && dx_bf16 && rows > 0 && C > 0 && count > 0,
               "lu_bn_lrelu_bwd_apply_bf16: bad arguments");
    LU_REQUIRE(vec4_ok(x, dy, C) && vec4_ok(scale, shift, C) && vec4_ok(save_mean, save_invstd, C) && vec4_ok(sums, sums, C) &&
                   (reinterpret_cast<uintptr_t>(dx_bf16) & 7) == 0,
               "lu_bn_lrelu_bwd_apply_bf16: needs C %% 4 == 0 and aligned tensors");
    const int64_t total = rows * C;
    LU_LAUNCH(bn_lrelu_bwd_apply_kernel, dim3(grid_for(total / 4)), dim3(NT), stream, x, dy, scale, shift, save_mean,
              save_invstd, alpha, sums, count, (float*)nullptr, dgamma, dbeta, total, (int)C, 1, (unsigned short*)dx_bf16);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_upsample2x_fwd(const float* x, float* y, int32_t frames, int32_t H, int32_t W, int32_t C,
                                 int32_t legacy, lu_stream_t stream) {
    LU_REQUIRE(x && y && frames > 0 && H > 0 && W > 0 && C > 0 && (legacy == 0 || legacy == 1), "lu_upsample2x_fwd: bad arguments");
    const int64_t total = (int64_t)frames * 4 * H * W * C;
    if (C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0)
        LU_LAUNCH((upsample2x_fwd_kernel<4>), dim3(grid_for(total / 4)), dim3(NT), stream, x, y, (int)frames, (int)H, (int)W,
                  (int)C, (int)legacy);
    else
        LU_LAUNCH((upsample2x_fwd_kernel<1>), dim3(grid_for(total)), dim3(NT), stream, x, y, (int)frames, (int)H, (int)W,
                  (int)C, (int)legacy);
    return LU_CHECK_LAUNCH();
}

/* lu_upsample2x_fwd with a bf16 result (C % 4 == 0, 16-byte aligned x, 8-byte aligned y) */
extern "C" int lu_upsample2x_fwd_bf16(const float* x, void* y_bf16, int32_t frames, int32_t H, int32_t W, int32_t C,
                                      int32_t legacy, lu_stream_t stream) {
    LU_REQUIRE(x && y_bf16 && frames > 0 && H > 0 && W > 0 && C > 0 && (legacy == 0 || legacy == 1), "lu_upsample2x_fwd_bf16: bad arguments");
    LU_REQUIRE(C % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y_bf16) & 7) == 0,
               "lu_upsample2x_fwd_bf16: needs C %% 4 == 0 and aligned tensors");
    const int64_t total = (int64_t)frames * 4 * H * W * C;
    LU_LAUNCH((upsample2x_fwd_kernel<4, true>), dim3(grid_for(total / 4)), dim3(NT), stream, x, (float*)y_bf16, (int)frames, (int)H,
              (int)W, (int)C, (int)legacy);
    return LU_CHECK_LAUNCH();
}

/* lu_bn_lrelu_apply with a bf16 result (C % 4 == 0, 16-byte aligned x / scale / shift, 8-byte aligned y) */
extern "C" int lu_bn_lrelu_apply_bf16(const float* x, void* y_bf16, const float* scale, const float* shift, float alpha,
                                      int64_t rows, int32_t C, lu_stream_t stream) {
    LU_REQUIRE(x && y_bf16 && scale && shift && rows > 0 && C > 0, "lu_bn_lrelu_apply_bf16: bad arguments");
    LU_REQUIRE(C % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(y_bf16) & 7) == 0,
               "lu_bn_lrelu_apply_bf16: needs C %% 4 == 0 and aligned tensors");
    const int64_t total = rows * C;
    LU_LAUNCH(bn_lrelu_apply_bf16_kernel, dim3(grid_for(total / 4)), dim3(NT), stream, x, (unsigned short*)y_bf16, scale, shift,
              alpha, total, (int)C);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_upsample2x_bwd(const float* dy, int32_t dy_ps, float* dx, int32_t frames, int32_t H, int32_t W,
                                 int32_t C, int32_t legacy, lu_stream_t stream) {
    LU_REQUIRE(dy && dx && frames > 0 && H > 0 && W > 0 && C > 0 && dy_ps >= C && (legacy == 0 || legacy == 1),
               "lu_upsample2x_bwd: bad arguments");
    const int64_t total = (int64_t)frames * H * W * C;
    if (C % 4 == 0 && dy_ps % 4 == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0)
        LU_LAUNCH((upsample2x_bwd_kernel<4>), dim3(grid_for(total / 4)), dim3(NT), stream, dy, (int)dy_ps, dx, (int)frames,
                  (int)H, (int)W, (int)C, (int)legacy);
    else
        LU_LAUNCH((upsample2x_bwd_kernel<1>), dim3(grid_for(total)), dim3(NT), stream, dy, (int)dy_ps, dx, (int)frames,
                  (int)H, (int)W, (int)C, (int)legacy);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_window_copy(const float* x, int32_t x_ps, float* y, int32_t frames, int32_t Hx, int32_t Wx,
                              int32_t Hy, int32_t Wy, int32_t C, int32_t off_y, int32_t off_x, int32_t mode,
                              float beta, lu_stream_t stream) {
    LU_REQUIRE(x && y && frames > 0 && Hx > 0 && Wx > 0 && Hy > 0 && Wy > 0 && C > 0 && x_ps >= C,
               "lu_window_copy: bad arguments");
    if (mode == 1)
        LU_REQUIRE(off_y < Hx && off_x < Wx && Hy - off_y - Hx < Hx && Wy - off_x - Wx < Wx,
                   "lu_window_copy: reflect pad must be smaller than the image");
    if (beta == 0.f && C % 4 == 0 && x_ps % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
        LU_LAUNCH(window_copy_vec4_kernel, dim3(grid_for((int64_t)frames * Hy * Wy * (C / 4))), dim3(NT), stream,
                  reinterpret_cast<const float4*>(x), (int)(x_ps / 4), reinterpret_cast<float4*>(y), (int)frames, (int)Hx, (int)Wx, (int)Hy,
                  (int)Wy, (int)(C / 4), (int)off_y, (int)off_x, (int)mode);
        return LU_CHECK_LAUNCH();
    }
    LU_LAUNCH(window_copy_kernel, dim3(grid_for((int64_t)frames * Hy * Wy * C)), dim3(NT), stream, x, (int)x_ps, y,
              (int)frames, (int)Hx, (int)Wx, (int)Hy, (int)Wy, (int)C, (int)off_y, (int)off_x, (int)mode, beta);
    return LU_CHECK_LAUNCH();
}

static int wce_blocks(int64_t rows) {
    int64_t b = (rows + NT - 1) / NT;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}
extern "C" size_t lu_wce_workspace_bytes(int64_t rows) { return (size_t)wce_blocks(rows) * 2 * sizeof(double); }

extern "C" int lu_softmax_wce_fwd(const float* logits, const float* gt, const float* class_w, float* softmax_out,
                                  double* sums, int64_t rows, void* ws, lu_stream_t stream) {
    LU_REQUIRE(logits && gt && class_w && sums && ws && rows > 0, "lu_softmax_wce_fwd: bad arguments");
    const int nb = wce_blocks(rows);
    LU_LAUNCH(wce_fwd_kernel, dim3(nb), dim3(NT), stream, logits, gt, class_w, softmax_out, rows, (double*)ws);
    int rc = LU_CHECK_LAUNCH();
    if (rc) return rc;
    LU_LAUNCH(wce_final_kernel, dim3(1), dim3(64), stream, (const double*)ws, nb, sums);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_softmax_wce_bwd(const float* logits, const float* gt, const float* class_w, const double* sums,
                                  float grad_scale, float* dlogits, int64_t rows, lu_stream_t stream) {
    LU_REQUIRE(logits && gt && class_w && sums && dlogits && rows > 0, "lu_softmax_wce_bwd: bad arguments");
    LU_LAUNCH(wce_bwd_kernel, dim3(grid_for(rows)), dim3(NT), stream, logits, gt, class_w, sums, grad_scale, dlogits,
              rows);
    return LU_CHECK_LAUNCH();
}

// softmax over the last axis for any class count (Networks.py:205-206: the head's depth is whatever the last up-block kernel says;
// the 3-class kernel above is the reference's own configuration and stays the fast path)
__global__ void softmax_rows_kernel(const float* __restrict__ logits, float* __restrict__ out, int64_t rows, int C) {
    for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < rows; r += (int64_t)gridDim.x * NT) {
        const float* l = logits + r * C;
        float m = l[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, l[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(l[c] - m);
        const float inv = 1.f / sum;
        for (int c = 0; c < C; ++c) out[r * C + c] = expf(l[c] - m) * inv;
    }
}

extern "C" int lu_softmax_rows(const float* logits, float* out, int64_t rows, int32_t classes, lu_stream_t stream) {
    LU_REQUIRE(logits && out && rows > 0 && classes >= 1, "lu_softmax_rows: bad arguments");
    LU_LAUNCH(softmax_rows_kernel, dim3(grid_for(rows)), dim3(NT), stream, logits, out, rows, (int)classes);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_softmax3(const float* logits, float* out, int64_t rows, lu_stream_t stream) {
    LU_REQUIRE(logits && out && rows > 0, "lu_softmax3: bad arguments");
    LU_LAUNCH(softmax3_kernel, dim3(grid_for(rows)), dim3(NT), stream, logits, out, rows);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_wce_finalize(const double* sums, float* loss, lu_stream_t stream) {
    LU_REQUIRE(sums && loss, "lu_wce_finalize: bad arguments");
    LU_LAUNCH(wce_loss_kernel, dim3(1), dim3(64), stream, sums, loss);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float alpha, float b1, float b2,
                            float eps, float grad_scale, lu_stream_t stream) {
    LU_REQUIRE(p && g && m && v && n > 0, "lu_adam_step: bad arguments");
    LU_LAUNCH(adam_kernel, dim3(grid_for(n, 4)), dim3(NT), stream, p, g, m, v, n, alpha, b1, b2, eps, grad_scale);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_scale_frames(float* x, const float* keep, int32_t frames, int64_t per_frame, lu_stream_t stream) {
    LU_REQUIRE(x && keep && frames > 0 && per_frame > 0, "lu_scale_frames: bad arguments");
    const int64_t total = (int64_t)frames * per_frame;
    LU_LAUNCH(scale_frames_kernel, dim3(grid_for(total)), dim3(NT), stream, x, keep, per_frame, total);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_state_begin(float* dst, void* dst_bf16, const float* src, const float* keep, int32_t frames, int64_t per_frame,
                              lu_stream_t stream) {
    LU_REQUIRE(dst && frames > 0 && frames < 65536 && per_frame > 0, "lu_state_begin: bad arguments");
    const bool vec = per_frame % 4 == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(dst_bf16) & 7) == 0;
    unsigned gx = grid_for(vec ? per_frame / 4 : per_frame);
    if (gx > 4096) gx = 4096;
    if (vec)
        LU_LAUNCH(state_begin_kernel<true>, dim3(gx, (unsigned)frames), dim3(NT), stream, dst, (unsigned short*)dst_bf16, src, keep,
                  per_frame);
    else
        LU_LAUNCH(state_begin_kernel<false>, dim3(gx, (unsigned)frames), dim3(NT), stream, dst, (unsigned short*)dst_bf16, src, keep,
                  per_frame);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_transpose_inner(const float* x, float* y, int64_t n, int32_t a, int32_t b, lu_stream_t stream) {
    LU_REQUIRE(x && y && n > 0 && a > 0 && b > 0 && n < 65536, "lu_transpose_inner: bad arguments");
    LU_LAUNCH(transpose_inner_kernel, dim3((b + 31) / 32, (a + 31) / 32, (unsigned)n), dim3(NT), stream, x, y, (int)a,
              (int)b);
    return LU_CHECK_LAUNCH();
}

extern "C" int lu_add_inplace(float* y, const float* x, int64_t n, lu_stream_t stream) {
    LU_REQUIRE(x && y && n > 0, "lu_add_inplace: bad arguments");
    LU_LAUNCH(add_inplace_kernel, dim3(grid_for(n, 4)), dim3(NT), stream, y, x, n);
    return LU_CHECK_LAUNCH();
}
