// Probe: what does ds_read_b64_tr_b16 deliver? (gfx950; guide T10.) LDS holds u16 element e = e; lane L supplies byte address
// pattern(L); the dump shows, per lane, which (source lane, element) pairs arrived -- i.e. the transpose the hardware performs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(uint16_t *out, int pattern) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    int addr = 0;
    if (pattern == 0) addr = lane * 8;                                  // lane L <-> elements 4L .. 4L+3
    if (pattern == 1) addr = (lane & 15) * 64 + (lane >> 4) * 8;        // 16 rows of 64 B, lane group g reads column block g
    if (pattern == 2) addr = (lane & 3) * 256 + ((lane >> 2) & 3) * 8 + (lane >> 4) * 32;  // [4 rows][4 x 4 elements] per 16-lane group
    const uint32_t a = (uint32_t)(uintptr_t)lds + addr;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[lane * 4 + 0] = v[0] & 0xffff;
    out[lane * 4 + 1] = v[0] >> 16;
    out[lane * 4 + 2] = v[1] & 0xffff;
    out[lane * 4 + 3] = v[1] >> 16;
}

int main() {
    uint16_t *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int p = 0; p < 3; ++p) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d: lane -> 4 values (as source element index; /4 = source lane for pattern 0)\n", p);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d: %5d %5d %5d %5d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
            if (p == 0) printf("   = (lane,elem) (%d,%d) (%d,%d) (%d,%d) (%d,%d)", h[l*4]/4, h[l*4]%4, h[l*4+1]/4, h[l*4+1]%4, h[l*4+2]/4, h[l*4+2]%4, h[l*4+3]/4, h[l*4+3]%4);
            printf("\n");
        }
    }
    return 0;
}
