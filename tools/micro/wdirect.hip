// Probe (round 4): the K loop of a "patch-resident activations + weights straight from global memory" GEMM / conv workgroup.
//   * the activation operand (MFMA B) sits in LDS for the whole workgroup lifetime (no ring, no barrier in the loop);
//   * the weight operand (MFMA A) goes global -> VGPR in MFMA layout: lane (r = l%32, g = l/32) reads 64 contiguous bytes of
//     weight row r (four dwordx4 = the lane's share of FOUR 16-wide k-steps), so two lanes cover one 128-byte line of a row --
//     K is permuted inside a 64-wide chunk (k-step s of lane group g = elements g*32 + s*8 .. +8), the LDS side reads the
//     same permutation;
//   * per k-step a wave issues MB ds_read_b128 (activation fragments) + NB fragments already in registers -> MB*NB MFMAs.
// Round 3 measured MFMA time + fragment-read time + LDS-DMA time ADDING in the ring kernels;
// this probe asks whether the structure above (half the LDS reads per MFMA, no LDS writes, no barrier) overlaps them.
// MODE bits: 1 = weight loads, 2 = LDS fragment reads, 4 = MFMAs, 8 = every wave streams its OWN weight rows (intra-workgroup
// K split; otherwise the four waves read the same rows, as four M-slices of one N tile would).
// Build: hipcc --offload-arch=gfx950 -O3 -o wdirect wdirect.hip ; run: ./wdirect
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int ROW_BYTES = 272;  // 128 channels * 2 B + 16 B pad: rows r, r+1 land 4 banks apart -> conflict-free ds_read_b128
constexpr int LDS_ROWS = 576;

template <int MB, int NB, int D, int MODE, int AUX>
__global__ __launch_bounds__(256, 1) void probe(const f16 *W, float *out, int iters, long wg_stride, long wave_stride, int ldw) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, g = lane >> 5;
    for (int i = tid; i < LDS_ROWS * ROW_BYTES / 4; i += 256) ((uint32_t *)lds)[i] = 0x2c002c00u + (i & 0x3ff);  // small f16 values
    __syncthreads();
    const f16 *wb = W + (long)blockIdx.x * wg_stride + ((MODE & 8) ? (long)wave * wave_stride : 0);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)wb);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)wb >> 32));
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void *)(((uintptr_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
    int voff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) voff[nb] = (MODE & 16) ? lane * 16 + nb * 4096 : ((nb * 32 + r) * ldw + g * 32) * 2;
    // MODE bit 16: PRE-SWIZZLED weights -- every (row block, k-step) fragment is one contiguous 1 KB block [lane][8 elements], so a
    // buffer_load_dwordx4 touches 8 full 128-byte lines instead of 32 lines at 32 bytes each (the tag rate of the vector L1 is one line
    // per clock: 32 clocks per strided fragment load, which is what the "shared rows" lines above measure)
    constexpr int JS = (MODE & 16) ? 1024 : 16;          // byte step between the four k-steps of a chunk
    constexpr int CS_ = (MODE & 16) ? NB * 4096 : 128;   // byte step between chunks
    const int lbase = (wave * (MB * 32) % 512 + r) * ROW_BYTES + g * 64;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
    u32x4 wreg[D][NB][4];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE & 1)
                    wreg[d][nb][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, voff[nb] + j * JS, d * CS_, AUX));
                else
                    wreg[d][nb][j] = u32x4{0x2c002c00u + lane, 0x2c002c00u, 0x2c012c00u, 0x2c002c01u};
                __builtin_amdgcn_sched_barrier(0);  // issue order = consumption order (else the loop-header wait collapses to vmcnt(0..3))
            }
    // the compiler sinks loads to their first use when left alone (all refills at the loop tail behind vmcnt(0), every
    // ds_read directly in front of its MFMA): sched_barrier fences pin the intended placement -- fragments of k-step s+1 are
    // requested before the MFMAs of k-step s, the refill of a weight buffer right behind the chunk that consumed it.
    f16x8 bf[2][MB];
    auto read_b = [&](int c, int s, f16x8 (&dst)[MB]) __attribute__((always_inline)) {
        const int loff = lbase + (c & 31) * ROW_BYTES + (c & 1) * 128;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            if (MODE & 2)
                dst[mb] = *(const f16x8 *)(lds + loff + mb * 32 * ROW_BYTES + s * 16);
            else
                dst[mb] = __builtin_bit_cast(f16x8, u32x4{0x2c002c00u, 0x2c012c00u + (uint32_t)mb, 0x2c002c00u, 0x2c002c01u});
        }
    };
    read_b(0, 0, bf[0]);
    for (int it = 0; it < iters; it += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int c = it + d;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 3)
                    read_b(c, s + 1, bf[(s + 1) & 1]);
                else
                    read_b(c + 1, 0, bf[0]);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE & 4) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wreg[d][nb][s]), bf[s & 1][mb], acc[mb][nb], 0, 0, 0);
                } else {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[mb][nb][s] += (float)bf[s & 1][mb][0] + __builtin_bit_cast(float, wreg[d][nb][s][0]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE & 1) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        wreg[d][nb][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, voff[nb] + j * JS, (c + D) * CS_, AUX));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[mb][nb][i];
    out[blockIdx.x * 256 + tid] = s;
}

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);   \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

template <int MB, int NB, int D, int MODE, int AUX> void run(const char *name, const f16 *W, float *out, size_t wbytes, int iters, bool hbm) {
    const int ldw = (iters + D + 1) * 64;                    // one row = the whole K stream of a wave
    long wave_stride = (long)NB * 32 * ldw;                  // elements
    long wg_stride = hbm ? ((MODE & 8) ? 4 * wave_stride : wave_stride) : 0;  // L2 mode: every workgroup reads the same rows
    const int wgs = 256;
    size_t need = ((size_t)(wgs - 1) * wg_stride + 4 * wave_stride) * 2;
    if (need > wbytes) {
        printf("%-44s skipped (needs %.1f MB)\n", name, need / 1e6);
        return;
    }
    auto k = probe<MB, NB, D, MODE, AUX>;
    const int lds_bytes = LDS_ROWS * ROW_BYTES;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), lds_bytes, 0, W, out, iters, wg_stride, wave_stride, ldw);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double ns_per_iter = best * 1e6 / iters;
    const double mfma_per_iter = 4.0 * MB * NB;  // per wave and 64-wide chunk
    const double tflops = (MODE & 4) ? wgs * 4 * (double)iters * mfma_per_iter * 32768.0 / (best * 1e-3) / 1e12 : 0.0;
    const double wbytes_moved = (MODE & 1) ? (double)wgs * ((MODE & 8) ? 4 : 1) * NB * 32.0 * iters * 128.0 : 0.0;
    printf("%-44s %8.3f ms  %7.1f ns/chunk  %6.1f TFLOP/s (%4.1f %% of 2500)  unique W %6.2f TB/s\n", name, best, ns_per_iter, tflops,
           tflops / 25.0, wbytes_moved / (best * 1e-3) / 1e12);
}

int main() {
    size_t wbytes = (size_t)3 << 30;
    f16 *W;
    float *out;
    CK(hipMalloc(&W, wbytes));
    CK(hipMemset(W, 0x2c, wbytes));  // 0x2c2c = small positive f16
    CK(hipMalloc(&out, 256 * 256 * 4));
    const int IT = 288, ITS = 72;
    printf("per chunk = 64 K elements; a wave issues 4*MB*NB MFMAs (32 cycles each on its SIMD), 4*MB ds_read_b128, 4*NB buffer_load_dwordx4\n");
    run<4, 2, 4, 4, 0>("MB4 NB2      MFMA only", W, out, wbytes, IT, false);
    run<4, 2, 4, 6, 0>("MB4 NB2      MFMA + LDS", W, out, wbytes, IT, false);
    run<4, 2, 4, 5, 0>("MB4 NB2 D4   MFMA + W(L2, shared rows)", W, out, wbytes, IT, false);
    run<4, 2, 4, 7, 0>("MB4 NB2 D4   MFMA + LDS + W(L2, shared)", W, out, wbytes, IT, false);
    run<4, 2, 4, 7, 0>("MB4 NB2 D4   all, W from HBM, shared rows", W, out, wbytes, IT, true);
    run<4, 2, 4, 7, 2>("MB4 NB2 D4   all, W from HBM, shared, nt", W, out, wbytes, IT, true);
    run<4, 2, 2, 7, 0>("MB4 NB2 D2   all, W from HBM, shared rows", W, out, wbytes, IT, true);
    run<4, 2, 6, 7, 0>("MB4 NB2 D6   all, W from HBM, shared rows", W, out, wbytes, IT, true);
    run<4, 2, 4, 15, 0>("MB4 NB2 D4   all, W from HBM, own rows/wave", W, out, wbytes, ITS, true);
    run<4, 2, 4, 15, 2>("MB4 NB2 D4   all, HBM, own rows/wave, nt", W, out, wbytes, ITS, true);
    run<4, 2, 4, 3, 2>("MB4 NB2 D4   LDS + W(HBM own rows, nt) no MFMA", W, out, wbytes, ITS, true);
    run<4, 1, 4, 7, 0>("MB4 NB1 D4   all, W from HBM, shared rows", W, out, wbytes, IT, true);
    run<4, 1, 6, 15, 2>("MB4 NB1 D6   all, HBM, own rows/wave, nt", W, out, wbytes, ITS, true);
    run<2, 2, 4, 7, 0>("MB2 NB2 D4   all, W from HBM, shared rows", W, out, wbytes, IT, true);
    run<2, 4, 3, 7, 0>("MB2 NB4 D3   all, W from HBM, shared rows", W, out, wbytes, IT, true);
    run<4, 4, 2, 7, 0>("MB4 NB4 D2   all, W from HBM, shared rows", W, out, wbytes, IT, true);
    run<4, 4, 2, 6, 0>("MB4 NB4      MFMA + LDS", W, out, wbytes, IT, false);
    run<4, 4, 2, 4, 0>("MB4 NB4      MFMA only", W, out, wbytes, IT, false);
    // round 4, second pass: every wave owns its weight rows (4 x 1 waves over the weight rows of a 256 x 128 tile: no two waves of a
    // workgroup load the same bytes) with the rows L2-resident (all workgroups share them, as the M-tiles of one N-tile do), strided
    // against pre-swizzled fragments
    run<4, 2, 4, 15, 0>("MB4 NB2 D4   all, W in L2, own rows/wave", W, out, wbytes, IT, false);
    run<4, 2, 4, 31, 0>("MB4 NB2 D4   all, W in L2, own rows, SWIZZLED", W, out, wbytes, IT, false);
    run<4, 2, 6, 31, 0>("MB4 NB2 D6   all, W in L2, own rows, SWIZZLED", W, out, wbytes, IT, false);
    run<4, 2, 4, 29, 0>("MB4 NB2 D4   MFMA + W(L2 own, SWIZZLED), no LDS", W, out, wbytes, IT, false);
    run<4, 2, 4, 23, 0>("MB4 NB2 D4   all, W in L2, shared rows, SWIZZLED", W, out, wbytes, IT, false);
    run<4, 2, 4, 31, 0>("MB4 NB2 D4   all, W from HBM, own rows, SWIZZLED", W, out, wbytes, ITS, true);
    run<4, 4, 2, 31, 0>("MB4 NB4 D2   all, W in L2, own rows, SWIZZLED", W, out, wbytes, IT, false);
    run<2, 4, 3, 31, 0>("MB2 NB4 D3   all, W in L2, own rows, SWIZZLED", W, out, wbytes, IT, false);
    run<2, 2, 4, 31, 0>("MB2 NB2 D4   all, W in L2, own rows, SWIZZLED", W, out, wbytes, IT, false);
    return 0;
}
