// Probe of ds_read_b64_tr_b16 lane/element mapping on gfx950 (tools only; prints what each lane receives).
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.hip -o tools/probe/tr_probe   (run on the GPU box; prints the mapping)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int pitch) {
    __shared__ __attribute__((aligned(16))) unsigned short L[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) L[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // lane supplies the address of (row = 8*(l>>5) + ((l&15)>>2), col = 16*((l>>4)&1) + 4*(l&3)) in a [rows][pitch] image
    const int row = 8 * (l >> 5) + ((l & 15) >> 2), col = 16 * ((l >> 4) & 1) + 4 * (l & 3);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(L + row * pitch + col));
    out[4 * l + 0] = v.x; out[4 * l + 1] = v.y; out[4 * l + 2] = v.z; out[4 * l + 3] = v.w;
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    const int pitch = 96;
    k<<<1, 64>>>(d, pitch);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int v = h[4 * l + j];
            printf(" (r%d,c%d)", v / pitch, v % pitch);
            // expectation: lane gets column (l&31), rows 8*(l>>5) + j
            if (v / pitch != 8 * (l >> 5) + j || v % pitch != (l & 31)) ++bad;
        }
        printf("\n");
    }
    printf("mismatches vs expectation: %d\n", bad);
    return 0;
}
