// Un-fused LoRA on the compiled UNet: every LoRA'd linear's effective weight  W_eff = W + s * up @ down  is rebuilt from the LIVE
// parameters by ONE launch per step (a device table of all the UNet's LoRA'd linears), and the plan's GEMMs read W_eff.
//
// Reference behaviour this stands for: a UNet with LoRA layers loaded and NOT fused is traced as it is -- per LoRA'd linear the traced
// graph holds linear(x, W) + scale * up(down(x)) (diffusers LoRACompatibleLinear / peft lora.Linear forward; three GEMM launches, a
// multiply and an add through sfast::cublas_lowp_linear, /root/reference/src/sfast/csrc/operators/cublas/cublas_gemm.cpp:798-948) --
// and "Dynamically Switch LoRA" (/root/reference/README.md:228-265, tests/compilers/test_stable_diffusion_pipeline_compiler.py:438-465)
// copies another adapter's tensors into the same storage in place; the captured graph sees them at the next replay. Here the live
// down / up / base tensors are read at every launch too, so the same in-place switch works without re-capture; the low-rank product
// is folded into the weight (N*K*r MACs once per step) instead of being applied to the activations (M*(N+K)*r MACs and two more
// launches per linear per step): 128 LoRA'd attention projections of an SD1.5 UNet cost one ~100 us launch on the side lane.
//
// Rounding: W_eff is rounded to the parameter dtype once per element (the reference rounds down(x), up(.), the scaled product and the
// sum, each to f16): a relative perturbation of <= 2^-11 (f16) of each weight, the size of the weight's own storage rounding.
#include "common.h"

namespace sfast {

constexpr int LORA_TN = SFAST_LORA_TILE_N, LORA_TK = SFAST_LORA_TILE_K, LORA_JC = 16, LORA_RMAX = SFAST_LORA_MAX_RANK;

template <typename T>
__global__ void __launch_bounds__(256) lora_merge_kernel(const sfast_lora_entry *__restrict__ tab, int n, const float *__restrict__ scales) {
    __shared__ float us[LORA_TN][LORA_RMAX + 1];                           // up tile as fp32 (+1: rows land in different banks)
    __shared__ __attribute__((aligned(16))) T ds[LORA_JC][LORA_TK];         // a chunk of down rows
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    // the entry this tile belongs to: the last one whose tile_begin <= blockIdx.x (uniform binary search over the table)
    int lo = 0, hi = n - 1;
    const int b = (int)blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].tile_begin <= b) lo = mid;
        else hi = mid - 1;
    }
    const sfast_lora_entry e = tab[lo];
    const int tiles_k = (e.K + LORA_TK - 1) / LORA_TK;
    const int t = b - e.tile_begin;
    const int n0 = (t / tiles_k) * LORA_TN, k0 = (t % tiles_k) * LORA_TK;
    const float s = scales ? scales[e.scale_index] : 1.0f;
    const T *up = (const T *)e.up, *down = (const T *)e.down, *w = (const T *)e.w;
    for (int i = tid; i < LORA_TN * e.r; i += 256) {
        const int row = i / e.r, j = i - row * e.r;
        us[row][j] = (n0 + row < e.N) ? Elem<T>::to_f32(up[(int64_t)(n0 + row) * e.ldu + j]) : 0.f;
    }
    const int kc = k0 + tx * 8;
    const bool k_ok = kc < e.K;  // K % 8 == 0: a chunk is inside or outside as a whole
    float acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[i][c] = 0.f;
    for (int j0 = 0; j0 < e.r; j0 += LORA_JC) {
        __syncthreads();  // previous chunk consumed (first trip: the up tile is complete after the next barrier)
        for (int q = tid; q < LORA_JC * (LORA_TK / 8); q += 256) {
            const int jj = q / (LORA_TK / 8), cc = (q - jj * (LORA_TK / 8)) * 8;
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (j0 + jj < e.r && k0 + cc < e.K) v = *reinterpret_cast<const u32x4 *>(down + (int64_t)(j0 + jj) * e.ldd + k0 + cc);
            *reinterpret_cast<u32x4 *>(&ds[jj][cc]) = v;
        }
        __syncthreads();
        const int jn = min(LORA_JC, e.r - j0);
        for (int jj = 0; jj < jn; ++jj) {
            float d[8];
            unpack8<T>(*reinterpret_cast<const u32x4 *>(&ds[jj][tx * 8]), d);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float u = us[ty + 8 * i][j0 + jj];
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[i][c] = fmaf(u, d[c], acc[i][c]);
            }
        }
    }
    if (!k_ok) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = n0 + ty + 8 * i;
        if (row < e.N) {
            float f[8];
            unpack8<T>(*reinterpret_cast<const u32x4 *>(w + (int64_t)row * e.ldw + kc), f);
#pragma unroll
            for (int c = 0; c < 8; ++c) f[c] = fmaf(s, acc[i][c], f[c]);
            *reinterpret_cast<u32x4 *>((T *)e.out + (int64_t)row * e.K + kc) = pack8<T>(f);
        }
    }
}

}  // namespace sfast

using namespace sfast;

static int lora_entry_ok(const sfast_lora_entry &e) {
    return e.w && e.down && e.up && e.out && e.N > 0 && e.K > 0 && e.K % 8 == 0 && e.r > 0 && e.r <= LORA_RMAX && e.ldw % 8 == 0 && e.ldw >= e.K &&
           e.ldd % 8 == 0 && e.ldd >= e.K && e.ldu >= e.r && e.scale_index >= 0 && aligned16(e.w) && aligned16(e.down) && aligned16(e.out);
}

extern "C" int sfast_hip_lora_merge_plan(sfast_lora_entry *entries, int32_t n, int32_t *total_tiles) {
    SFAST_REQUIRE(entries && total_tiles && n > 0, SFAST_ERR_INVALID, "lora_merge_plan: null argument");
    int64_t tiles = 0;
    for (int i = 0; i < n; ++i) {
        SFAST_REQUIRE(lora_entry_ok(entries[i]), SFAST_ERR_UNSUPPORTED,
                      "lora_merge_plan: entry %d outside the kernel's coverage (K %% 8, rank <= %d, 16-byte aligned rows)", i, LORA_RMAX);
        entries[i].tile_begin = (int32_t)tiles;
        tiles += (int64_t)((entries[i].N + LORA_TN - 1) / LORA_TN) * ((entries[i].K + LORA_TK - 1) / LORA_TK);
        SFAST_REQUIRE(tiles < (1ll << 30), SFAST_ERR_UNSUPPORTED, "lora_merge_plan: too many tiles");
    }
    *total_tiles = (int32_t)tiles;
    return SFAST_OK;
}

extern "C" int sfast_hip_lora_merge(const sfast_lora_entry *entries_device, int32_t n, int32_t total_tiles, const float *scales_device,
                                    int32_t dtype, sfast_stream_t stream) {
    SFAST_REQUIRE(entries_device && n > 0 && total_tiles > 0, SFAST_ERR_INVALID, "lora_merge: null / empty table");
    hipStream_t st = (hipStream_t)stream;
    set_kernel_name("lora_merge[%d linears,%d tiles]", n, total_tiles);
    if (dtype == SFAST_F16)
        hipLaunchKernelGGL(lora_merge_kernel<f16>, dim3((unsigned)total_tiles), dim3(256), 0, st, entries_device, n, scales_device);
    else if (dtype == SFAST_BF16)
        hipLaunchKernelGGL(lora_merge_kernel<bf16>, dim3((unsigned)total_tiles), dim3(256), 0, st, entries_device, n, scales_device);
    else {
        set_error("lora_merge: dtype %d", dtype);
        return SFAST_ERR_UNSUPPORTED;
    }
    return check_launch("lora_merge");
}
