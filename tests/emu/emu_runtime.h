// emu_runtime.h -- TEST INFRASTRUCTURE ONLY.
// A tiny single-OS-thread SIMT emulator so that the *unmodified* kernel sources under
// lstm-unet_amd/csrc can be compiled for the host (clang++ -DLU_EMU) and checked against the
// oracle in the GPU-less build container.  Each GPU thread is a ucontext fiber; a block's 256
// fibers run round-robin between barriers; one block runs at a time.  MFMA and wave shuffles are
// emulated through a per-wave exchange buffer with the exact gfx950 lane->element maps.
// It is never built into, loaded by, or reachable from the product library.
#pragma once
#include <setjmp.h>
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct float2 { float x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short lu_bf16x8 __attribute__((ext_vector_type(8)));   // raw bf16 bit patterns
typedef short lu_bf16x4 __attribute__((ext_vector_type(4)));

namespace lu_emu {

struct Fiber {
    ucontext_t ctx;      // first entry only (makecontext); later switches use _setjmp/_longjmp: no sigprocmask syscalls
    jmp_buf jb;
    bool started = false;
    char* stack = nullptr;
    dim3 tid;
    int lin = 0;
    int state = 0;  // 0 runnable, 1 blocked, 2 done
};

struct Runtime {
    ucontext_t main_ctx;
    jmp_buf main_jb;
    std::vector<Fiber> fibers;
    Fiber* cur = nullptr;
    dim3 blockIdx_, blockDim_, gridDim_;
    int nthreads = 0;
    int block_arrived = 0;
    int wave_arrived[16] = {0};
    float xa[16][64];
    float xb[16][64];
    short xa8[16][64][8];
    short xb8[16][64][8];
    std::function<void()> body;
};
inline Runtime g_rt;
// dynamic LDS (extern __shared__): one block runs at a time, so one 160 KB arena serves every launch
alignas(16) inline unsigned char g_dyn_lds[160 * 1024];

inline void yield_to_main() {
    if (!_setjmp(g_rt.cur->jb)) _longjmp(g_rt.main_jb, 1);
}

inline void block_barrier() {
    Runtime& r = g_rt;
    int alive = 0;
    for (int i = 0; i < r.nthreads; ++i) alive += (r.fibers[i].state != 2);
    if (++r.block_arrived == alive) {
        r.block_arrived = 0;
        for (int i = 0; i < r.nthreads; ++i)
            if (r.fibers[i].state == 1) r.fibers[i].state = 0;
        return;
    }
    r.cur->state = 1;
    yield_to_main();
}

inline void wave_barrier() {
    Runtime& r = g_rt;
    int w = r.cur->lin >> 6;
    int lo = w * 64, hi = lo + 64 < r.nthreads ? lo + 64 : r.nthreads;
    int alive = 0;
    for (int i = lo; i < hi; ++i) alive += (r.fibers[i].state != 2);
    if (++r.wave_arrived[w] == alive) {
        r.wave_arrived[w] = 0;
        for (int i = lo; i < hi; ++i)
            if (r.fibers[i].state == 3) r.fibers[i].state = 0;
        return;
    }
    r.cur->state = 3;  // blocked on wave barrier (distinct from block barrier)
    yield_to_main();
}

inline void fiber_entry() {
    g_rt.body();
    g_rt.cur->state = 2;
    _longjmp(g_rt.main_jb, 1);      // never resumed
}

inline void run_block() {
    Runtime& r = g_rt;
    const size_t STK = 256 * 1024;
    r.block_arrived = 0;
    memset(r.wave_arrived, 0, sizeof(r.wave_arrived));
    for (int i = 0; i < r.nthreads; ++i) {
        Fiber& f = r.fibers[i];
        if (!f.stack) f.stack = (char*)malloc(STK);
        f.state = 0;
        f.started = false;
        f.lin = i;
        f.tid = dim3(i % r.blockDim_.x, (i / r.blockDim_.x) % r.blockDim_.y, i / (r.blockDim_.x * r.blockDim_.y));
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STK;
        f.ctx.uc_link = &r.main_ctx;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    int done = 0;
    long spins = 0;
    while (done < r.nthreads) {
        bool progressed = false;
        for (int i = 0; i < r.nthreads; ++i) {
            Fiber& f = r.fibers[i];
            if (f.state != 0) continue;
            r.cur = &f;
            if (!_setjmp(r.main_jb)) {
                if (f.started) {
                    _longjmp(f.jb, 1);
                } else {
                    f.started = true;
                    setcontext(&f.ctx);
                }
            }
            progressed = true;
            if (f.state == 2) ++done;
        }
        if (!progressed) {
            fprintf(stderr, "lu_emu: deadlock (divergent barrier?) block (%u,%u,%u)\n", r.blockIdx_.x, r.blockIdx_.y,
                    r.blockIdx_.z);
            abort();
        }
        if (++spins > 100000000L) abort();
    }
}

template <class F>
inline void launch(dim3 grid, dim3 block, F body) {
    Runtime& r = g_rt;
    r.gridDim_ = grid;
    r.blockDim_ = block;
    r.nthreads = block.x * block.y * block.z;
    if ((int)r.fibers.size() < r.nthreads) r.fibers.resize(r.nthreads);
    r.body = body;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                r.blockIdx_ = dim3(bx, by, bz);
                run_block();
            }
}

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5).
// Result is a k-ordered fmaf chain (guide: bitwise equal to the hardware).
inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    Runtime& r = g_rt;
    int lane = r.cur->lin & 63, w = r.cur->lin >> 6;
    r.xa[w][lane] = a;
    r.xb[w][lane] = b;
    wave_barrier();
    int col = lane & 31;
    for (int reg = 0; reg < 16; ++reg) {
        int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        float acc = c[reg];
        for (int k = 0; k < 2; ++k) acc = fmaf(r.xa[w][row + 32 * k], r.xb[w][col + 32 * k], acc);
        c[reg] = acc;
    }
    wave_barrier();
    return c;
}

inline float bf16_bits_to_float(short b) {
    unsigned u = ((unsigned)(unsigned short)b) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// v_mfma_f32_32x32x16_bf16: lane l holds A[i = l&31][k = 8*(l>>5) + j], B[k = 8*(l>>5) + j][n = l&31], j < 8; D as above.
inline f32x16 mfma_32x32x16_bf16(lu_bf16x8 a, lu_bf16x8 b, f32x16 c) {
    Runtime& r = g_rt;
    int lane = r.cur->lin & 63, w = r.cur->lin >> 6;
    for (int j = 0; j < 8; ++j) {
        r.xa8[w][lane][j] = a[j];
        r.xb8[w][lane][j] = b[j];
    }
    wave_barrier();
    int col = lane & 31;
    for (int reg = 0; reg < 16; ++reg) {
        int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        float acc = c[reg];
        for (int half = 0; half < 2; ++half)
            for (int j = 0; j < 8; ++j)
                acc = fmaf(bf16_bits_to_float(r.xa8[w][row + 32 * half][j]), bf16_bits_to_float(r.xb8[w][col + 32 * half][j]), acc);
        c[reg] = acc;
    }
    wave_barrier();
    return c;
}

// ds_read_b64_tr_b16 (probed on gfx950, tools/probe/tr_probe.hip): every lane supplies the address of 4 consecutive
// 16-bit elements; within each 16-lane group the 4 x 16 block formed by lanes (4j .. 4j+3) = row j is transposed, so
// lane t of the group receives column t: element j comes from lane 4j + (t >> 2), position t & 3.
inline lu_bf16x4 lds_read_tr16_b64(const unsigned short* p) {
    Runtime& r = g_rt;
    int lane = r.cur->lin & 63, w = r.cur->lin >> 6;
    for (int j = 0; j < 4; ++j) r.xa8[w][lane][j] = (short)p[j];
    wave_barrier();
    const int g = lane & ~15, t = lane & 15;
    lu_bf16x4 o;
    for (int j = 0; j < 4; ++j) o[j] = r.xa8[w][g + 4 * j + (t >> 2)][t & 3];
    wave_barrier();
    return o;
}

inline float shfl_xor(float v, int mask) {
    Runtime& r = g_rt;
    int lane = r.cur->lin & 63, w = r.cur->lin >> 6;
    r.xa[w][lane] = v;
    wave_barrier();
    float o = r.xa[w][lane ^ mask];
    wave_barrier();
    return o;
}
inline float shfl_down(float v, int d) {
    Runtime& r = g_rt;
    int lane = r.cur->lin & 63, w = r.cur->lin >> 6;
    r.xa[w][lane] = v;
    wave_barrier();
    float o = lane + d < 64 ? r.xa[w][lane + d] : v;
    wave_barrier();
    return o;
}

}  // namespace lu_emu

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define threadIdx (lu_emu::g_rt.cur->tid)
#define blockIdx (lu_emu::g_rt.blockIdx_)
#define blockDim (lu_emu::g_rt.blockDim_)
#define gridDim (lu_emu::g_rt.gridDim_)
#define __syncthreads() lu_emu::block_barrier()

// integer atomics / bit ops of the post-processing kernels: fibers of one block run interleaved on ONE OS thread and only
// switch at barriers, so plain read-modify-write is atomic here
static inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }

typedef void* hipStream_t;
