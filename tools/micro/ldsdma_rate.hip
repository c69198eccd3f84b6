// Probe: how many bytes per clock one CU moves L2 -> LDS with LDS-DMA (global_load_lds_dwordx4 / buffer_load_dwordx4 ... lds), against
// plain global_load_dwordx4 into VGPRs, with every CU of the chip doing the same. One 512-thread workgroup per CU (160 KB of LDS
// requested), 8 waves, each wave-instruction moves 1 KiB (8 rows x 128 B, 16 B per lane) -- the request shape of the igemm kernels.
//   mode 0: LDS-DMA, global form, every workgroup its own region      mode 1: the same through a buffer descriptor
//   mode 2: plain loads into VGPRs (own region)                        mode 3: LDS-DMA, all workgroups read the SAME region (weights)
//   mode 4: LDS-DMA, own region for 5/8 of the requests + the shared region for 3/8 (the mix of a 256 x 160 conv tile)
//   mode 5: plain loads, same mix as 4                                 mode 6: half the requests LDS-DMA (own), half plain (shared)
// region bytes per workgroup and row pitch are arguments; `depth` = requests a wave keeps in flight (vmcnt).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) * gsrc_t;
typedef __attribute__((address_space(3))) void *ldst_t;

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int DEPTH>
__global__ void __launch_bounds__(512, 2) rate_kernel(const char *own, const char *shared, int64_t own_stride, int rows, int pitch, int iters, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char *mine = own + (int64_t)blockIdx.x * own_stride;
    // a "tile" = `rows` rows of 128 bytes; wave w takes rows w*8 .. w*8+7 of every 64-row pass
    const int passes = rows / 64;
    const int r_in_pass = wave * 8 + (lane >> 3);
    const int coff = (lane & 7) * 16;
    __amdgpu_buffer_rsrc_t rs_own = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(mine), 0, (unsigned)own_stride, 0x00020000);
    u32x4 accv = {0, 0, 0, 0};
    u32x4 prev[6] = {}, cur[6] = {};  // plain loads: consumed one iteration later (6 - 12 requests in flight per wave, compiler-counted)
    int n = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            if (p >= passes) break;
            const int row = p * 64 + r_in_pass;
            // tiles alternate between two halves of the region so consecutive iterations do not hit the same lines back to back in L1
            const int64_t off = (int64_t)((it & 1) * rows + row) * pitch + coff;
            ldst_t dst = (ldst_t)(smem + ((it % 3) * rows + p * 64) * 128 + wave * 1024);
            const bool use_shared = (MODE == 3) || ((MODE == 4 || MODE == 5) && (p % 8) >= 5);
            const char *src = (use_shared ? shared : mine) + off;
            if (MODE == 0 || MODE == 3 || MODE == 4 || MODE >= 7) {
                __builtin_amdgcn_global_load_lds((gsrc_t)(const void *)src, dst, 16, 0, 0);
            } else if (MODE == 1) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_own, dst, 16, (int)off, 0, 0, 0);
            } else if (MODE == 2 || MODE == 5) {
                cur[p] = *(gsrc_t)(const void *)src;
            } else if (MODE == 6) {
                if (p & 1) {
                    cur[p] = *(gsrc_t)(const void *)(shared + off);
                } else {
                    __builtin_amdgcn_global_load_lds((gsrc_t)(const void *)src, dst, 16, 0, 0);
                }
            }
            if (MODE != 2 && MODE != 5 && MODE != 6 && MODE < 7)
                if (++n > DEPTH) wait_vm<DEPTH>();
        }
        if (MODE >= 7) {  // the K-loop's synchronisation: counted wait for the previous tile(s), then MODE - 6 barriers per tile
            wait_vm<DEPTH>();
            __builtin_amdgcn_s_barrier();
            if (MODE >= 8) __builtin_amdgcn_s_barrier();
        }
        if (MODE == 2 || MODE == 5 || MODE == 6) {
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                accv ^= prev[p];
                prev[p] = cur[p];
            }
        }
    }
    wait_vm<0>();
    __syncthreads();
    if (sink && tid == 0 && blockIdx.x == 0xffffff) sink[0] = accv[0] + *(uint32_t *)smem;
}

template <int MODE, int DEPTH> float run(const char *own, const char *shared, int64_t own_stride, int rows, int pitch, int iters, uint32_t *sink, int blocks) {
    auto k = rate_kernel<MODE, DEPTH>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 160 * 1024, 0, own, shared, own_stride, rows, pitch, iters, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const int blocks = 256;
    const int rows = 384;  // 48 KB per tile: 6 passes of 64 rows
    const int iters = 400;
    uint32_t *sink;
    CHECK(hipMalloc(&sink, 64));
    for (int pitch : {128, 640, 2560}) {
        const int64_t own_stride = (int64_t)2 * rows * pitch + 4096;
        char *own, *shared;
        CHECK(hipMalloc(&own, own_stride * blocks));
        CHECK(hipMalloc(&shared, own_stride));
        CHECK(hipMemset(own, 1, own_stride * blocks));
        CHECK(hipMemset(shared, 1, own_stride));
        const double bytes = (double)rows * 128 * iters;  // per workgroup
        printf("pitch %d B, region per workgroup %.0f KB (x%d = %.1f MB), tile %d KB, %d tiles\n", pitch, own_stride / 1024.0, blocks, own_stride * blocks / 1048576.0,
               rows * 128 / 1024, iters);
        const char *names[7] = {"LDS-DMA global form, own region", "LDS-DMA buffer form, own region", "plain loads -> VGPR, own region", "LDS-DMA, shared region",
                                "LDS-DMA, 5/8 own + 3/8 shared", "plain loads, 5/8 own + 3/8 shared", "half LDS-DMA (own) + half plain (shared)"};
        float t[7][2];
        t[0][0] = run<0, 12>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[0][1] = run<0, 24>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[1][0] = run<1, 12>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[1][1] = run<1, 24>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[2][0] = run<2, 12>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[2][1] = run<2, 24>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[3][0] = run<3, 12>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[3][1] = run<3, 24>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[4][0] = run<4, 12>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[4][1] = run<4, 24>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[5][0] = run<5, 12>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[5][1] = run<5, 24>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[6][0] = run<6, 12>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        t[6][1] = run<6, 24>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
        for (int m = 0; m < 7; ++m)
            for (int d = 0; d < 2; ++d)
                printf("  %-44s depth %2d: %8.3f ms  %6.1f GB/s per CU  %5.1f B/clk at 2.4 GHz  (chip %5.2f TB/s)\n", names[m], d ? 24 : 12, t[m][d],
                       bytes / t[m][d] / 1e6, bytes / t[m][d] / 1e6 / 2.4, bytes * blocks / t[m][d] / 1e9);
        {
            float a6 = run<7, 6>(own, shared, own_stride, rows, pitch, iters, sink, blocks), b6 = run<8, 6>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
            float a12 = run<7, 12>(own, shared, own_stride, rows, pitch, iters, sink, blocks), b12 = run<8, 12>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
            float a18 = run<7, 18>(own, shared, own_stride, rows, pitch, iters, sink, blocks), b18 = run<8, 18>(own, shared, own_stride, rows, pitch, iters, sink, blocks);
            printf("  LDS-DMA, per tile {6 requests, vmcnt(N), 1 barrier}: N=6 %6.1f GB/s  N=12 %6.1f  N=18 %6.1f   2 barriers: N=6 %6.1f  N=12 %6.1f  N=18 %6.1f\n",
                   bytes / a6 / 1e6, bytes / a12 / 1e6, bytes / a18 / 1e6, bytes / b6 / 1e6, bytes / b12 / 1e6, bytes / b18 / 1e6);
        }
        // one workgroup alone (no competition for L2 / fabric)
        float s0 = run<0, 24>(own, shared, own_stride, rows, pitch, iters, sink, 1);
        float s2 = run<2, 24>(own, shared, own_stride, rows, pitch, iters, sink, 1);
        printf("  ONE workgroup on the chip: LDS-DMA %6.1f GB/s (%4.1f B/clk), plain loads %6.1f GB/s (%4.1f B/clk)\n", bytes / s0 / 1e6, bytes / s0 / 1e6 / 2.4,
               bytes / s2 / 1e6, bytes / s2 / 1e6 / 2.4);
        CHECK(hipFree(own));
        CHECK(hipFree(shared));
    }
    return 0;
}
