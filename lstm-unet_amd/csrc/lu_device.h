// lu_device.h -- device-side vocabulary of the gfx950 kernels.
// The product build is HIP for gfx950 only.  The single abstraction point below (LU_EMU) exists so
// that tests/emu can compile the SAME kernel sources for the host SIMT emulator in the GPU-less
// build container; it is never defined in the product build.
#pragma once
#include <stdint.h>
#include <string.h>
#include "../../include/lstm_unet_hip.h"

#ifdef LU_EMU
#include "emu_runtime.h"
#define LU_LAUNCH(kernel, grid, block, stream, ...) \
    lu_emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
#define LU_LAUNCH_DYN(kernel, grid, block, lds_bytes, stream, ...) LU_LAUNCH(kernel, grid, block, stream, __VA_ARGS__)
#define LU_DYN_LDS(type, name) type* name = reinterpret_cast<type*>(lu_emu::g_dyn_lds)
static inline f32x16 lu_mfma(float a, float b, f32x16 c) { return lu_emu::mfma_32x32x2(a, b, c); }
static inline f32x16 lu_mfma_bf16(lu_bf16x8 a, lu_bf16x8 b, f32x16 c) { return lu_emu::mfma_32x32x16_bf16(a, b, c); }
static inline lu_bf16x4 lu_lds_tr16(const unsigned short* p) { return lu_emu::lds_read_tr16_b64(p); }
static inline float lu_shfl_xor(float v, int m) { return lu_emu::shfl_xor(v, m); }
static inline float lu_shfl_down(float v, int d) { return lu_emu::shfl_down(v, d); }
// global_load_lds_dwordx4: every lane copies 16 bytes from ITS global pointer to (wave-uniform LDS base + lane*16)
static inline void lu_glds16(const float* gptr, float* lds_wave_base) {
    const int lane = lu_emu::g_rt.cur->lin & 63;
    memcpy(reinterpret_cast<char*>(lds_wave_base) + lane * 16, gptr, 16);
}
#define LU_CHECK_LAUNCH() 0
#define LU_SCHED_FENCE() ((void)0)
#define LU_SCHED_GROUP(mask, n) ((void)0)
#define LU_WAVE_SYNC() lu_emu::wave_barrier()      // lanes of a wave exchange data through LDS without a block barrier
#define LU_UNIFORM(x) (x)                          // (device: v_readfirstlane -- tells the compiler a value is wave-uniform)
#else
#include <hip/hip_runtime.h>
#include <atomic>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// Per-DEVICE host-side caches (function attributes, device-symbol addresses): one process may drive several devices (a test
// that selects a second GPU, a multi-device host), and both a kernel's attributes and the address of a __device__ symbol belong
// to the device they were looked up on.  Slots are indexed by hipGetDevice(); racing first calls compute the same value.
#define LU_MAX_DEVICES 64
static inline int lu_current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= LU_MAX_DEVICES) return -1;
    return d;
}
// address of a __device__ symbol on the current device, cached per device; null on failure (callers fall back to the symbol)
#define LU_SYMBOL_ADDRESS(sym)                                                                                       \
    ([]() -> const void* {                                                                                           \
        static std::atomic<const void*> lu_addr_[LU_MAX_DEVICES];                                                    \
        const int lu_d_ = lu_current_device();                                                                       \
        if (lu_d_ < 0) return nullptr;                                                                               \
        const void* lu_p_ = lu_addr_[lu_d_].load(std::memory_order_acquire);                                         \
        if (!lu_p_) {                                                                                                \
            void* lu_q_ = nullptr;                                                                                   \
            if (hipGetSymbolAddress(&lu_q_, HIP_SYMBOL(sym)) == hipSuccess && lu_q_) {                               \
                lu_addr_[lu_d_].store(lu_q_, std::memory_order_release);                                             \
                lu_p_ = lu_q_;                                                                                       \
            }                                                                                                        \
        }                                                                                                            \
        return lu_p_;                                                                                                \
    }())
#define LU_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), 0, (hipStream_t)(stream), __VA_ARGS__)
// launch with `lds_bytes` of dynamic LDS (extern __shared__); beyond the 64 KB static limit the kernel has to opt in once
#define LU_LAUNCH_DYN(kernel, grid, block, lds_bytes, stream, ...)                                                  \
    do {                                                                                                            \
        static std::atomic<unsigned long long> lu_dyn_ready_{0ull};      /* one bit per device */                    \
        const int lu_dev_ = lu_current_device();                                                                    \
        const unsigned long long lu_bit_ = lu_dev_ < 0 ? 0ull : (1ull << lu_dev_);                                  \
        if (!(lu_dyn_ready_.load(std::memory_order_acquire) & lu_bit_) || !lu_bit_) {                               \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel),                                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                      \
            lu_dyn_ready_.fetch_or(lu_bit_, std::memory_order_release);                                             \
        }                                                                                                           \
        hipLaunchKernelGGL(kernel, (grid), (block), (lds_bytes), (hipStream_t)(stream), __VA_ARGS__);               \
    } while (0)
#define LU_DYN_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
// v_mfma_f32_32x32x2_f32: exact f32, k-ordered fmaf chain; 64 cycles / SIMD
__device__ __forceinline__ f32x16 lu_mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
typedef short lu_bf16x8 __attribute__((ext_vector_type(8)));   // raw bf16 bit patterns
// v_mfma_f32_32x32x16_bf16: bf16 operands (8 per lane), fp32 accumulate; 16x the fp32-MFMA rate
__device__ __forceinline__ f32x16 lu_mfma_bf16(lu_bf16x8 a, lu_bf16x8 b, f32x16 c) {
    typedef __bf16 hw_bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hw_bf16x8, a), __builtin_bit_cast(hw_bf16x8, b), c, 0,
                                                   0, 0);
}
typedef short lu_bf16x4 __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16: transposing LDS read.  Each lane passes the address of 4 consecutive bf16 (8-byte aligned); per
// 16-lane group, lanes 4j .. 4j+3 address row j of a 4 x 16 block and lane t receives column t (rows 0..3).  It turns a
// pixel-major [k][channel] LDS image into the k-contiguous fragments the bf16 MFMA wants (tools/probe/tr_probe.hip).
__device__ __forceinline__ lu_bf16x4 lu_lds_tr16(const unsigned short* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lu_bf16x4*)p);
}
__device__ __forceinline__ float lu_shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ float lu_shfl_down(float v, int d) { return __shfl_down(v, d, 64); }
// LDS-DMA (global_load_lds_dwordx4): each lane copies 16 bytes from ITS global pointer straight into LDS at
// (wave-uniform base + lane*16) -- no VGPR staging, no ds_write; completion is tracked by vmcnt.
__device__ __forceinline__ void lu_glds16(const float* gptr, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#define LU_CHECK_LAUNCH() lu_check_launch()
// pins instruction order across this point: sched_barrier stops the machine scheduler, the empty asm with a
// memory clobber stops the IR optimiser (which otherwise hoists the next stage's LDS stores -- and the vmcnt
// wait they need -- above the MFMA block, destroying the load/compute overlap).  Emits no instruction.
// scheduling recipe for the instructions between two fences: "next come n instructions of class `mask`"
// (0x008 MFMA, 0x020 VMEM read, 0x100 LDS read, 0x200 LDS write)
#define LU_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#define LU_SCHED_FENCE()                      \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
// The 64 lanes of a wave run in lockstep and a wave's LDS operations execute in order: an LDS exchange INSIDE a wave needs no
// s_barrier, only the compiler kept from moving the reads above the writes.
#define LU_WAVE_SYNC()                                          \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)
int lu_check_launch();
// a value that is the same in every lane of the wave (derived from the wave index, say), moved to a scalar register: address
// arithmetic and selects on it become SALU instructions, and a branch / select on it needs no exec masking
#define LU_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

// 16 bytes of zeros in device memory: masked-out lanes of the tile loaders read THIS instead of branching
// around the load or zeroing afterwards (either makes hipcc wait for the load right where it was issued).
// (deliberately NOT const: a const would live in the constant address space and turn the selected pointer
// into a flat pointer -> flat_load instead of global_load.)
__device__ __attribute__((aligned(16))) static float lu_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// 16 raw bytes (one global_load_dwordx4 / ds_write_b128): staging registers of tiles whose element type is a template parameter
typedef unsigned lu_u4 __attribute__((ext_vector_type(4)));
typedef unsigned lu_u2 __attribute__((ext_vector_type(2)));
__device__ __host__ static inline float lu_bits2f(unsigned u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__device__ __host__ static inline unsigned lu_f2bits(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return u;
}

void lu_set_error(const char* fmt, ...);

#define LU_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            lu_set_error(__VA_ARGS__); \
            return 1;                  \
        }                              \
    } while (0)

static inline int64_t lu_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// two fp32 -> packed bf16 pair (lo in bits 0..15), round to nearest even: one v_cvt_pk_bf16_f32 on the device
#ifdef LU_EMU
static inline unsigned lu_pack2bf(float lo, float hi);
#else
__device__ __forceinline__ unsigned lu_pack2bf(float lo, float hi) {
    typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
    hw_bf16x2 v;
    v.x = (__bf16)lo;
    v.y = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}
#endif

// tanh of every gate kernel (fused ConvLSTM epilogues, the pointwise gate forward / backward), ~15 VALU instructions instead of
// the ~50 of ocml's tanhf: the gate block of the fused bf16 step evaluates 64 of them per thread and tile, and with one
// 16 x 32 tile per CU nothing covers that time (20 us of a 96 us tile at the 256^2 level).  |x| >= 0.1: 1 - 2 / (exp(2|x|) + 1)
// on v_exp_f32 / v_rcp_f32 (absolute error ~1.5e-7 = 1-2 ulp of the result's range; exp overflow -> inf -> exactly 1); below:
// x - x^3/3 + 2x^5/15 (relative error < 6e-8).  ONE definition for all paths and both precisions: the fused and the unfused
// (tile-starved / K-split) route of a layer must give the same bits, or the route choice -- which depends on the batch size --
// shows up as bf16 re-rounding differences between a data-parallel run and the single-process run of the same batch.
#ifdef LU_EMU
static inline float lu_tanh_fast(float x) {
    const float ax = fabsf(x);
    const float e = exp2f(ax * 2.8853900817779268f);
    const float t = copysignf(1.f - 2.f / (e + 1.f), x);
    const float x2 = x * x;
    const float s = fmaf(x * x2, fmaf(x2, 0.13333333f, -0.33333334f), x);
    return ax < 0.1f ? s : t;
}
#else
__device__ __forceinline__ float lu_tanh_fast(float x) {
    const float ax = fabsf(x);
    const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);
    const float t = copysignf(fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f), x);
    const float x2 = x * x;
    const float s = fmaf(x * x2, fmaf(x2, 0.13333333f, -0.33333334f), x);
    return ax < 0.1f ? s : t;
}
#endif

// fp32 -> bf16 bits, round to nearest even (finite inputs)
__device__ __host__ static inline unsigned short lu_f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
#ifdef LU_EMU
static inline unsigned lu_pack2bf(float lo, float hi) { return (unsigned)lu_f2bf(lo) | ((unsigned)lu_f2bf(hi) << 16); }
#endif

// Exact three-way bf16 split of an fp32 value (precision 'bf16x3', lu_split6): x == hi + mid + lo; both residuals are exact in fp32.
// (contract(off): when x is the product of a caller's multiply, hipcc would otherwise fuse it into these subtractions -- fma(a, b, -hi) --
// and the pieces would sum to the UNROUNDED product instead of the fp32 value that was stored; found on the MI355X, not on the emulator)
__device__ __host__ static inline void lu_split3(float x, float& hi, float& mid, float& lo) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    unsigned u = (unsigned)lu_f2bf(x) << 16;
    if ((u & 0x7F800000u) == 0x7F800000u) {      // the rounded hi is not finite (round 6)
        unsigned xb;
        memcpy(&xb, &x, 4);
        if ((xb & 0x7F800000u) != 0x7F800000u) {
            u = (xb & 0x80000000u) | 0x7F7F0000u;      // finite x within 2^-9 of FLT_MAX: hi = the largest finite bf16, the residuals stay exact
        } else {                                       // inf / NaN travel in hi alone (inf - inf would make the other pieces NaN)
            u = xb & 0xFFFF0000u;
            if ((xb & 0x007FFFFFu) && !(u & 0x007F0000u)) u |= 0x00400000u;      // a NaN whose payload sits in the low half stays a NaN
            memcpy(&hi, &u, 4);
            mid = 0.f;
            lo = 0.f;
            return;
        }
    }
    memcpy(&hi, &u, 4);
    const float r1 = x - hi;
    u = (unsigned)lu_f2bf(r1) << 16;
    memcpy(&mid, &u, 4);
    u = (unsigned)lu_f2bf(r1 - mid) << 16;
    memcpy(&lo, &u, 4);
}
