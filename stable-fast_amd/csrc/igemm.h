// Argument block + host entry of the MFMA implicit-GEMM kernels (igemm.hip, igemm_glds.hip).
#pragma once
#include "common.h"

namespace sfast {

struct IgemmArgs {
    const void *x, *x2;
    const void *w[SFAST_MAX_WSEG];
    const void *wpk[SFAST_MAX_WSEG];  // pipe 4: the segments' packed copies (sfast_hip_pack_weight), or nullptr
    int pk_ksteps;                    //         16-wide k-steps per 32-row block of a packed segment: ceil(K / 64) * 4
    const void *bias, *rowbias, *res;
    void *out;
    float *partial;
    unsigned *tickets;  // split-K ticket counters, one per output tile (zero between launches), or nullptr: separate reduce kernel
    int M, N, K;
    int64_t ldx, ldw, ldo, ldr, ld_rowbias;
    int rows_per_seg, rows_per_batch;
    int act, res_before_act;
    float alpha;
    // conv geometry (MODE 1)
    int H, W, C1, C2, Ho, Wo, KH, KW, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, ups;
    int tiles_m, tiles_n, ktiles, ktiles_per_split, splits;
    unsigned long long *trace;  // profiling: per-workgroup phase timestamps (8 slots each); nullptr in production
    // ---- epilogue extensions (sfast_epilogue_ext) -----------------------------------------------------------
    float out_scale;   // v = out_scale * acc + bias + ...   (1 = plain)
    int stage_out;     // 1: the finished tile is staged in LDS and leaves as whole 16-byte row segments
    float *gn_stats;   // GroupNorm partial statistics of the output (float2 {mean, M2} per tile slot), or nullptr
    int gn_unit;       // channels per statistics unit (divides every consumer's channels-per-group)
    int gn_slots;      // unit slots per tile_n (host: stats_slots(BNO, unit))
    int gn_rows_per_sample;  // host-side validation only
    int w_int8;              // weights are int8 [N][K] (ldw in bytes), dequantised to T while they are staged (igemm_w8_kernel)
    // GroupNorm(+SiLU) of the OUTPUT, computed by the split-K reduce launch (splitk_reduce_gn_kernel; sfast_epilogue_ext.gn_out):
    void *gn_out;            // dense [M][N] normalised tensor, or nullptr
    const void *gn_gamma, *gn_beta;
    int gn_groups, gn_act;   // rows_per_batch = pixels per sample
    float gn_eps;
    // ---- block -> (tile, K-split) map over the 8 XCDs (decode_block, igemm_device.h) ----------------------------
    // xmap = 1: 1-D grid of tiles*splits blocks; XCD b%8 owns a box of x_sp K-splits x x_tm tile rows x x_tn tile columns
    // (2^x_lxn boxes along n, 2^x_lxm along m, the rest of the 8 along the K-splits). xmap = 0: grid (tiles, splits), every XCD a
    // contiguous run of row-major tiles.
    // xmap = 2 (round 4): the (split, tile_n, tile_m) triples in ONE linear order, XCD b%8 owns the contiguous run [xcd * x_per,
    // (xcd + 1) * x_per) of it -- no divisibility needed; x_order 0: tile_m fastest (a run stays inside few (split, tile_n) pairs: the
    // weight panel is fetched by ~one XCD), 1: tile_n fastest (the activation panel is). Grid = 8 * x_per blocks, the surplus exit at once.
    int xmap, x_lxn, x_lxm, x_tn, x_tm, x_sp, x_per, x_order;
};

// launch grid of the MFMA kernels for the block map carried by `a`
static inline dim3 igemm_grid(const IgemmArgs &a) {
    if (a.xmap == 2) return dim3((unsigned)(8 * a.x_per), 1, 1);
    return a.xmap ? dim3((unsigned)(a.tiles_m * a.tiles_n * a.splits), 1, 1) : dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)a.splits, 1);
}

// statistics slots a tile of `bno` output columns can overlap: units are `unit` channels wide, tile origins multiples of bno
static inline int stats_slots(int bno, int unit) { return (bno - 1) / unit + 2; }

// `caps` of a problem (the `glds_ok` argument of the planning functions below, an int): bit 0 = the LDS-DMA pipes may be chosen;
// bit 1 = packed copies of the weights are at hand (pipe 4, igemm_pk.hip);
// for convs the patch pipe (conv_patch.hip) may be chosen can: 3x3, stride 1, padding 1, dense NHWC, C1 / C2 multiples of 64 -- then
// bits 8..19 = image width W, bits 20..31 = image height H (the tile has to cover whole image rows).
// bit 2 (round 6) = a conv with a fused nearest-2x upsample that the 12-wave forms of pipe 5 (variants 55 ..) can take although no other LDS-DMA pipe can
static inline int igemm_caps(bool glds_ok, int patch_h, int patch_w, bool packed = false, bool pp_ups = false) {
    return (glds_ok ? 1 : 0) | (packed ? 2 : 0) | (pp_ups ? 4 : 0) | ((patch_h > 0 && patch_w > 0 && patch_h < 4096 && patch_w < 4096) ? ((patch_w << 8) | (patch_h << 20)) : 0);
}
bool conv_patch_fits(int H, int W, int M, int BM, int BN);                                 // conv_patch.hip
int conv_patch_launch(const IgemmArgs &a, int dtype, int BM, int BN, hipStream_t st);       // conv_patch.hip

// mode: 0 = linear (x row m at x + m*ldx), 1 = conv (implicit im2col over dense NHWC x / x2).
// Fills the plan fields of `a` (tiles, split) and launches on `st`.
int igemm_run(IgemmArgs &a, int dtype, int mode, bool geglu, int variant, int split, void *ws, size_t ws_bytes,
              hipStream_t st);
// GroupNorm-statistics layout the epilogue of this problem would write (tile rows, tile columns, slots per tile_n, tiles_n,
// float2 records in total); returns false when the chosen kernel cannot emit them
struct StatsLayout {
    int rb_rows, bno, slots, tiles_n, n_rb;
};
bool igemm_stats_layout(int M, int N, int K, bool geglu, int variant, int split, int glds_ok, int unit, int rows_per_sample,
                        bool tickets, StatsLayout &out);
// glds_ok: whether the LDS-DMA pipe may be chosen for this problem (see igemm_glds_eligible)
void igemm_plan_query(int M, int N, int K, bool geglu, int variant, int split, int glds_ok, int out[5]);
size_t igemm_workspace_bytes(int M, int N, int K, bool geglu, int variant, int split, int glds_ok);
bool igemm_glds_eligible(const IgemmArgs &a, int mode);
bool igemm_reduce_gn_ok(int M, int N, int rows_per_batch, int groups);  // igemm.hip: can the reduce launch normalise its output?
// split-K reduce + epilogue over fp32 slabs [splits][M][N] another kernel wrote (gnconv.hip)
int igemm_reduce_only(const IgemmArgs &a, int dtype, hipStream_t st);
// GroupNorm(+SiLU) -> 3x3 conv as one weight-streaming launch for B*H*W <= 128 (gnconv.hip; api: sfast_hip_gn_conv2d)
struct GnConvPlan {
    int MB, NB, CS, S;  // 32-pixel blocks, 32-channel output blocks per wave, channel slice, slices
    size_t lds_bytes, slab_bytes;
};
bool gnconv_plan(int B, int H, int W, int C1, int C2, int Cout, int groups, GnConvPlan &pl);
int gnconv_run(IgemmArgs &a, int dtype, int B, const void *gamma, const void *beta, int groups, float eps, int silu, void *ws, size_t ws_bytes,
               hipStream_t st);
// weight-only int8 linear (sfast::cutlass_qlinear_dynamic): register-staged pipe, no split-K
int igemm_run_w8(IgemmArgs &a, int dtype, hipStream_t st);
// grouped launch (register-staged pipe): n_groups problems of identical [M, N, K] sharing x (api: sfast_hip_gemm_grouped)
int igemm_run_grouped(IgemmArgs &a, int dtype, int n_groups, const void *const *xs, const void *const *w_segs, int n_wseg,
                      const void *const *bias, void *const *out, hipStream_t st);

}  // namespace sfast
