// extern "C" entry points of the GEMM and conv families: validation + kernel selection.
//   sfast_hip_gemm   -> gemv_small_m | MFMA igemm (linear) | naive
//   sfast_hip_conv2d -> MFMA igemm (implicit im2col) | conv_small_n | conv_small_c | naive
// No path leaves the GPU and nothing is allocated: unsupported fast-path preconditions select the
// generic HIP kernel, never an ATen / host fallback.
#include "igemm.h"
#include "small.h"

using namespace sfast;

namespace {

bool is_half(int dtype) { return dtype == SFAST_F16 || dtype == SFAST_BF16; }

struct GemmRoute {
    enum Kind { GEMV, IGEMM, NAIVE } kind;
};

GemmRoute gemm_route(const void *x, const void *const *w_segs, const void *bias, const void *rowbias,
                     const void *residual, const void *out, const sfast_gemm_params *p) {
    GemmRoute r{GemmRoute::NAIVE};
    if (!is_half(p->dtype)) return r;
    bool vec_in = p->K % 8 == 0 && p->ldx % 8 == 0 && p->ldw % 8 == 0 && aligned16(x);
    for (int i = 0; i < p->n_wseg; ++i) vec_in = vec_in && aligned16(w_segs[i]);
    if (!vec_in) return r;
    if (p->variant == 100) return r;  // forced naive
    if ((p->M <= 16 && p->variant == 0) || p->variant == 101) {
        r.kind = GemmRoute::GEMV;
        return r;
    }
    bool vec_out = p->N % 4 == 0 && p->ldo % 4 == 0 && aligned8(out) && p->in_act == SFAST_ACT_NONE;
    if (bias) vec_out = vec_out && aligned8(bias);
    if (rowbias) vec_out = vec_out && aligned8(rowbias) && p->ld_rowbias % 4 == 0 && p->rows_per_batch > 0;
    if (residual) vec_out = vec_out && aligned8(residual) && p->ldr % 4 == 0;
    if (p->geglu) vec_out = vec_out && p->n_wseg <= 2;  // one [2N, K] weight or (hidden, gate) segments
    if (vec_out) r.kind = GemmRoute::IGEMM;
    return r;
}

int validate_gemm(const void *x, const void *const *w_segs, const void *out, const sfast_gemm_params *p) {
    SFAST_REQUIRE(p && x && w_segs && out, SFAST_ERR_INVALID, "gemm: null argument");
    SFAST_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, SFAST_ERR_INVALID, "gemm: bad shape %dx%dx%d", p->M, p->N, p->K);
    SFAST_REQUIRE(p->n_wseg >= 1 && p->n_wseg <= SFAST_MAX_WSEG, SFAST_ERR_INVALID, "gemm: n_wseg=%d", p->n_wseg);
    const int wrows = p->geglu ? 2 * p->N : p->N;
    SFAST_REQUIRE(p->rows_per_seg > 0 && (int64_t)p->rows_per_seg * p->n_wseg >= wrows, SFAST_ERR_INVALID,
                  "gemm: %d segments of %d rows do not cover %d weight rows", p->n_wseg, p->rows_per_seg, wrows);
    // GEGLU: one [2N, K] weight (hidden rows, then gate rows) or TWO [N, K] segments (hidden, gate) -- the reference's two-weight
    // cutlass_linear_geglu(input, weight0, bias0, weight1, bias1) without a concatenated copy
    SFAST_REQUIRE(!p->geglu || p->n_wseg == 1 || (p->n_wseg == 2 && p->rows_per_seg == p->N), SFAST_ERR_INVALID,
                  "gemm: geglu takes one [2N, K] weight or two [N, K] segments (got %d segments of %d rows, N = %d)", p->n_wseg, p->rows_per_seg, p->N);
    for (int i = 0; i < p->n_wseg; ++i) SFAST_REQUIRE(w_segs[i], SFAST_ERR_INVALID, "gemm: null weight segment %d", i);
    SFAST_REQUIRE(p->ldx >= p->K && p->ldw >= p->K && p->ldo >= p->N, SFAST_ERR_INVALID, "gemm: bad leading dims");
    return SFAST_OK;
}

struct ConvGeom {
    int Ho, Wo;
    bool x_dense, x2_dense, w_kcontig, out_dense, z_dense, z_bcast;
    int64_t ldo, ldr;
};

ConvGeom conv_geom(const sfast_conv_params *p, const void *z) {
    ConvGeom g{};
    const int Hin = p->upsample2x ? 2 * p->H : p->H, Win = p->upsample2x ? 2 * p->W : p->W;
    g.Ho = (Hin + 2 * p->pad_h + p->pad_h_extra - p->dil_h * (p->KH - 1) - 1) / p->stride_h + 1;
    g.Wo = (Win + 2 * p->pad_w + p->pad_w_extra - p->dil_w * (p->KW - 1) - 1) / p->stride_w + 1;
    const int C1 = p->C1, C2 = p->Cin - p->C1;
    g.x_dense = p->xs[3] == 1 && p->xs[2] == C1 && p->xs[1] == (int64_t)p->W * C1 && p->xs[0] == (int64_t)p->H * p->W * C1;
    g.x2_dense = C2 == 0 || (p->x2s[3] == 1 && p->x2s[2] == C2 && p->x2s[1] == (int64_t)p->W * C2 &&
                             p->x2s[0] == (int64_t)p->H * p->W * C2);
    // size-1 kernel dims have ambiguous strides (NCHW-contiguous 1x1 weights are K-contiguous too)
    g.w_kcontig = (p->Cin == 1 || p->ws[1] == 1) && (p->KW == 1 || p->ws[3] == p->Cin) &&
                  (p->KH == 1 || p->ws[2] == (int64_t)p->KW * p->Cin) &&
                  p->ws[0] == (int64_t)p->KH * p->KW * p->Cin;
    g.ldo = p->os[2];
    g.out_dense = p->os[3] == 1 && g.ldo >= p->Cout && p->os[1] == (int64_t)g.Wo * g.ldo &&
                  p->os[0] == (int64_t)g.Ho * g.Wo * g.ldo;
    g.ldr = 0;
    g.z_dense = false;
    g.z_bcast = false;
    if (z) {
        g.ldr = p->zs[2];
        g.z_dense = p->zs[3] == 1 && g.ldr >= p->Cout && p->zs[1] == (int64_t)g.Wo * g.ldr &&
                    p->zs[0] == (int64_t)g.Ho * g.Wo * g.ldr;
        g.z_bcast = p->zs[3] == 1 && p->zs[1] == 0 && p->zs[2] == 0;
    }
    return g;
}

enum ConvKind { CONV_IGEMM, CONV_SMALL_N, CONV_SMALL_C, CONV_NAIVE };

ConvKind conv_route(const void *x, const void *x2, const void *w, const void *bias, const void *rowbias,
                    const void *z, const void *out, const sfast_conv_params *p, const ConvGeom &g,
                    bool &fold_z_to_rowbias) {
    fold_z_to_rowbias = false;
    if (p->variant == 100) return CONV_NAIVE;
    const int C2 = p->Cin - p->C1;
    if (is_half(p->dtype)) {
        const bool in_ok = g.x_dense && g.x2_dense && g.w_kcontig && p->C1 % 8 == 0 && C2 % 8 == 0 && aligned16(x) &&
                           (C2 == 0 || aligned16(x2)) && aligned16(w);
        if (in_ok && p->Cout >= 16 && p->Cout % 4 == 0 && g.out_dense && g.ldo % 4 == 0 && aligned8(out) &&
            (!bias || aligned8(bias)) && (!rowbias || (aligned8(rowbias) && p->ld_rowbias % 4 == 0))) {
            bool z_ok = true;
            if (z) {
                if (g.z_dense && g.ldr % 4 == 0 && aligned8(z)) {
                    z_ok = true;
                } else if (g.z_bcast && !rowbias && p->alpha == 1.0f && aligned8(z) && p->zs[0] % 4 == 0 &&
                           (p->res_before_act || p->act == SFAST_ACT_NONE)) {
                    fold_z_to_rowbias = true;
                } else {
                    z_ok = false;
                }
            }
            if (z_ok) return CONV_IGEMM;
        }
        if (in_ok && C2 == 0 && p->Cout <= 8 && p->variant != 102) return CONV_SMALL_N;
        if ((int64_t)p->KH * p->KW * p->Cin * p->Cout * 4 <= 64 * 1024 && p->Cout % 8 == 0 && p->variant != 101)
            return CONV_SMALL_C;
    }
    return CONV_NAIVE;
}

int validate_conv(const void *x, const void *x2, const void *w, const void *out, const sfast_conv_params *p) {
    SFAST_REQUIRE(p && x && w && out, SFAST_ERR_INVALID, "conv2d: null argument");
    SFAST_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0 && p->Cin > 0 && p->Cout > 0 && p->KH > 0 && p->KW > 0,
                  SFAST_ERR_INVALID, "conv2d: bad shape");
    SFAST_REQUIRE(p->stride_h > 0 && p->stride_w > 0 && p->dil_h > 0 && p->dil_w > 0 && p->pad_h >= 0 && p->pad_w >= 0 &&
                      p->pad_h_extra >= 0 && p->pad_w_extra >= 0,
                  SFAST_ERR_INVALID, "conv2d: bad stride/dilation/padding");
    SFAST_REQUIRE(p->C1 > 0 && p->C1 <= p->Cin, SFAST_ERR_INVALID, "conv2d: bad C1=%d", p->C1);
    SFAST_REQUIRE(p->C1 == p->Cin || x2, SFAST_ERR_INVALID, "conv2d: concat needs x2");
    return SFAST_OK;
}

void fill_small_conv(SmallConvArgs &a, const void *x, const void *x2, const void *w, const void *bias,
                     const void *rowbias, const void *z, void *out, const sfast_conv_params *p, const ConvGeom &g) {
    a.x = x;
    a.x2 = x2;
    a.w = w;
    a.bias = bias;
    a.rowbias = rowbias;
    a.z = z;
    a.out = out;
    a.B = p->B;
    a.H = p->H;
    a.W = p->W;
    a.Cin = p->Cin;
    a.C1 = p->C1;
    a.Cout = p->Cout;
    a.KH = p->KH;
    a.KW = p->KW;
    a.Ho = g.Ho;
    a.Wo = g.Wo;
    a.stride_h = p->stride_h;
    a.stride_w = p->stride_w;
    a.pad_h = p->pad_h;
    a.pad_w = p->pad_w;
    a.dil_h = p->dil_h;
    a.dil_w = p->dil_w;
    a.ups = p->upsample2x;
    for (int i = 0; i < 4; ++i) {
        a.xs[i] = p->xs[i];
        a.x2s[i] = p->x2s[i];
        a.ws[i] = p->ws[i];
        a.os[i] = p->os[i];
        a.zs[i] = p->zs[i];
    }
    a.ld_rowbias = p->ld_rowbias;
    a.act = p->act;
    a.res_before_act = p->res_before_act;
    a.alpha = p->alpha;
}

// planning capabilities of a conv problem (igemm.h igemm_caps): LDS-DMA pipes need 64-channel slices per source and uniform taps per
// K-tile; the patch pipe (conv_patch.hip) additionally 3x3 / stride 1 / padding 1 / no dilation / no fused upsample
static int conv_caps(const sfast_conv_params *p) {
    const int C2 = p->Cin - p->C1;
    const bool glds_ok = !p->upsample2x && p->C1 % 64 == 0 && C2 % 64 == 0 && p->KH * p->KW <= 32;
    const bool patch = glds_ok && p->KH == 3 && p->KW == 3 && p->stride_h == 1 && p->stride_w == 1 && p->pad_h == 1 && p->pad_w == 1 &&
                       p->pad_h_extra == 0 && p->pad_w_extra == 0 && p->dil_h == 1 && p->dil_w == 1;
    // forced pipe-4 variants (41..): the query assumes the packed copy the launch will be handed (sfast_epilogue_ext.w_packed)
    const bool pp_ups = p->upsample2x && C2 == 0 && p->C1 % 64 == 0 && p->KH * p->KW <= 30 && p->dil_h == 1 && p->dil_w == 1 &&
                        (int64_t)p->B * p->H * p->W * p->C1 < (1ll << 31);
    return igemm_caps(glds_ok, patch ? p->H : 0, patch ? p->W : 0, p->variant >= 40 && p->variant < 100, pp_ups);
}

}  // namespace

// what igemm_run would choose for this conv under a forced (variant, split): out = {BM, BN, splits, K-tiles per split, variant id}
extern "C" int sfast_hip_conv2d_plan(const sfast_conv_params *p, int32_t variant, int32_t split_k, int32_t out[5]) {
    if (!p || !out) return SFAST_ERR_INVALID;
    const ConvGeom g = conv_geom(p, nullptr);
    const int64_t M = (int64_t)p->B * g.Ho * g.Wo;
    if (M <= 0 || M > INT32_MAX) return SFAST_ERR_INVALID;
    int o[5];
    igemm_plan_query((int)M, p->Cout, p->KH * p->KW * p->Cin, false, variant < 100 ? variant : 0, split_k, conv_caps(p), o);
    for (int i = 0; i < 5; ++i) out[i] = o[i];
    return SFAST_OK;
}

extern "C" int sfast_hip_igemm_plan(int32_t M, int32_t N, int32_t K, int32_t geglu, int32_t variant, int32_t split_k,
                                    int32_t out[5]) {
    if (!out || M <= 0 || N <= 0 || K <= 0) return SFAST_ERR_INVALID;
    int o[5];
    igemm_plan_query(M, N, K, geglu != 0, variant < 100 ? variant : 0, split_k, igemm_caps(true, 0, 0, variant >= 40 && variant < 100), o);
    for (int i = 0; i < 5; ++i) out[i] = o[i];
    return SFAST_OK;
}

extern "C" size_t sfast_hip_gemm_workspace_bytes(const sfast_gemm_params *p) {
    if (!p || !is_half(p->dtype) || (p->M <= 16 && p->variant == 0) || p->variant >= 100 || p->K % 8 != 0) return 0;
    return igemm_workspace_bytes(p->M, p->N, p->K, p->geglu != 0, p->variant < 100 ? p->variant : 0, p->split_k,
                                 igemm_caps(true, 0, 0, p->variant >= 40 && p->variant < 100));
}

static float ext_scale(const sfast_epilogue_ext *ext) { return (ext && ext->out_scale != 0.0f) ? ext->out_scale : 1.0f; }
// sfast_epilogue_ext.gn_out: the consumer GroupNorm rides in the split-K reduce launch (igemm.hip splitk_reduce_gn_kernel)
static int ext_gn(const sfast_epilogue_ext *ext, const void *gn_stats, IgemmArgs &a) {
    a.gn_out = nullptr;
    if (!ext || !ext->gn_out) return SFAST_OK;
    SFAST_REQUIRE(!gn_stats && ext->gn_unit == 0, SFAST_ERR_INVALID, "epilogue ext: gn_out and statistics emission are exclusive");
    SFAST_REQUIRE(ext->gn_act == SFAST_ACT_NONE || ext->gn_act == SFAST_ACT_SILU, SFAST_ERR_UNSUPPORTED, "epilogue ext: gn_act %d", ext->gn_act);
    SFAST_REQUIRE(ext->gn_rows_per_sample > 0 && igemm_reduce_gn_ok(a.M, a.N, ext->gn_rows_per_sample, ext->gn_groups), SFAST_ERR_UNSUPPORTED,
                  "epilogue ext: fused GroupNorm outside coverage (M=%d N=%d rows/sample=%d groups=%d)", a.M, a.N, ext->gn_rows_per_sample, ext->gn_groups);
    SFAST_REQUIRE(aligned8(ext->gn_out) && (!ext->gn_gamma || aligned8(ext->gn_gamma)) && (!ext->gn_beta || aligned8(ext->gn_beta)), SFAST_ERR_INVALID,
                  "epilogue ext: gn_out / gamma / beta must be 8-byte aligned");
    a.gn_out = ext->gn_out;
    a.gn_gamma = ext->gn_gamma;
    a.gn_beta = ext->gn_beta;
    a.gn_groups = ext->gn_groups;
    a.gn_eps = ext->gn_eps;
    a.gn_act = ext->gn_act;
    a.rows_per_batch = ext->gn_rows_per_sample;
    return SFAST_OK;
}
// sfast_epilogue_ext.w_packed: packed copies of the weight segments (sfast_hip_pack_weight) -> pipe 4 may be chosen
static void ext_packed(const sfast_epilogue_ext *ext, int nseg, IgemmArgs &a) {
    for (int i = 0; i < SFAST_MAX_WSEG; ++i) a.wpk[i] = (ext && ext->w_packed && i < nseg) ? ext->w_packed[i] : nullptr;
}
static bool ext_tickets(const sfast_epilogue_ext *ext) { return ext && (ext->flags & SFAST_EXT_WS_TICKETS) != 0; }

// SFAST_EXT_WS_TICKETS: the last SFAST_WS_TICKET_BYTES of the workspace are the split-K ticket counters; the rest is scratch
static void split_workspace(const sfast_epilogue_ext *ext, void *workspace, size_t &workspace_bytes, IgemmArgs &a) {
    a.tickets = nullptr;
    if (ext_tickets(ext) && workspace && workspace_bytes >= SFAST_WS_TICKET_BYTES && (workspace_bytes % 4) == 0) {
        workspace_bytes -= SFAST_WS_TICKET_BYTES;
        a.tickets = (unsigned *)((char *)workspace + workspace_bytes);
    }
}

extern "C" int sfast_hip_workspace_init(void *workspace, size_t workspace_bytes, sfast_stream_t stream) {
    SFAST_REQUIRE(workspace && workspace_bytes >= SFAST_WS_TICKET_BYTES && workspace_bytes % 4 == 0, SFAST_ERR_WORKSPACE,
                  "workspace_init: needs a workspace of at least %d bytes (a multiple of 4)", SFAST_WS_TICKET_BYTES);
    hipError_t e = hipMemsetAsync((char *)workspace + workspace_bytes - SFAST_WS_TICKET_BYTES, 0, SFAST_WS_TICKET_BYTES, (hipStream_t)stream);
    SFAST_REQUIRE(e == hipSuccess, SFAST_ERR_LAUNCH, "workspace_init: %s", hipGetErrorString(e));
    return SFAST_OK;
}

extern "C" int sfast_hip_gemm(const void *x, const void *const *w_segs, const void *bias, const void *rowbias,
                              const void *residual, void *out, const sfast_gemm_params *p, void *workspace,
                              size_t workspace_bytes, sfast_stream_t stream) {
    return sfast_hip_gemm_ex(x, w_segs, bias, rowbias, residual, out, p, nullptr, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int sfast_hip_gemm_stats_layout(const sfast_gemm_params *p, const sfast_epilogue_ext *ext, sfast_gn_stats_layout *out) {
    SFAST_REQUIRE(p && ext && out, SFAST_ERR_INVALID, "gemm_stats_layout: null argument");
    StatsLayout l{};
    const bool igemm = is_half(p->dtype) && p->K % 8 == 0 && !(p->M <= 16 && p->variant == 0) && p->variant < 100 && p->N % 8 == 0 && p->ldo % 8 == 0;
    if (!igemm || !igemm_stats_layout(p->M, p->N, p->K, p->geglu != 0, p->variant, p->split_k, igemm_caps(true, 0, 0, p->variant >= 40 && p->variant < 100), ext->gn_unit, ext->gn_rows_per_sample, ext_tickets(ext), l)) {
        set_error("gemm_stats_layout: this problem / kernel choice cannot emit GroupNorm statistics");
        return SFAST_ERR_UNSUPPORTED;
    }
    out->rb_rows = l.rb_rows;
    out->n_rb = l.n_rb;
    out->bno = l.bno;
    out->tiles_n = l.tiles_n;
    out->slots = l.slots;
    out->unit = ext->gn_unit;
    return SFAST_OK;
}

extern "C" int sfast_hip_gemm_ex(const void *x, const void *const *w_segs, const void *bias, const void *rowbias,
                                 const void *residual, void *out, const sfast_gemm_params *p, const sfast_epilogue_ext *ext,
                                 void *gn_stats, void *workspace, size_t workspace_bytes, sfast_stream_t stream) {
    int rc = validate_gemm(x, w_segs, out, p);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const GemmRoute route = gemm_route(x, w_segs, bias, rowbias, residual, out, p);
    SFAST_REQUIRE(!gn_stats || (ext && route.kind == GemmRoute::IGEMM), SFAST_ERR_UNSUPPORTED,
                  "gemm: GroupNorm statistics are emitted by the MFMA path only (and need an sfast_epilogue_ext)");
    SFAST_REQUIRE(!(ext && ext->gn_out) || route.kind == GemmRoute::IGEMM, SFAST_ERR_UNSUPPORTED,
                  "gemm: the fused GroupNorm epilogue (sfast_epilogue_ext.gn_out) exists on the split-K MFMA path only");
    if (route.kind == GemmRoute::IGEMM) {
        IgemmArgs a{};
        a.x = x;
        a.x2 = nullptr;
        for (int i = 0; i < SFAST_MAX_WSEG; ++i) a.w[i] = i < p->n_wseg ? w_segs[i] : w_segs[0];
        if (p->geglu && p->n_wseg == 1)  // the MFMA GEGLU kernels read hidden rows from w[0] and gate rows from w[1]
            a.w[1] = (const char *)w_segs[0] + (int64_t)p->N * p->ldw * (int64_t)dtype_bytes(p->dtype);
        a.bias = bias;
        a.rowbias = rowbias;
        a.res = residual;
        a.out = out;
        a.M = p->M;
        a.N = p->N;
        a.K = p->K;
        a.ldx = p->ldx;
        a.ldw = p->ldw;
        a.ldo = p->ldo;
        a.ldr = p->ldr;
        a.ld_rowbias = p->ld_rowbias;
        a.rows_per_seg = p->rows_per_seg;
        a.rows_per_batch = p->rows_per_batch > 0 ? p->rows_per_batch : 1;
        a.act = p->act;
        a.res_before_act = p->res_before_act;
        a.alpha = p->alpha;
        a.out_scale = ext_scale(ext);
        a.gn_stats = (float *)gn_stats;
        a.gn_unit = ext ? ext->gn_unit : 0;
        a.gn_rows_per_sample = ext ? ext->gn_rows_per_sample : 0;
        if (ext && ext->gn_out) {
            SFAST_REQUIRE(!rowbias || p->rows_per_batch == ext->gn_rows_per_sample, SFAST_ERR_INVALID, "gemm: row-bias batches and GroupNorm samples differ");
            rc = ext_gn(ext, gn_stats, a);
            if (rc) return rc;
        }
        ext_packed(ext, p->n_wseg, a);
        split_workspace(ext, workspace, workspace_bytes, a);
        return igemm_run(a, p->dtype, 0, p->geglu != 0, p->variant < 100 ? p->variant : 0, p->split_k, workspace, workspace_bytes, st);
    }
    SmallGemmArgs a{};
    a.x = x;
    for (int i = 0; i < SFAST_MAX_WSEG; ++i) a.w[i] = i < p->n_wseg ? w_segs[i] : w_segs[0];
    a.bias = bias;
    a.rowbias = rowbias;
    a.res = residual;
    a.out = out;
    a.M = p->M;
    a.N = p->N;
    a.K = p->K;
    a.ldx = p->ldx;
    a.ldw = p->ldw;
    a.ldo = p->ldo;
    a.ldr = p->ldr;
    a.ld_rowbias = p->ld_rowbias;
    a.rows_per_seg = p->rows_per_seg;
    a.rows_per_batch = p->rows_per_batch > 0 ? p->rows_per_batch : 1;
    a.geglu = p->geglu;
    a.act = p->act;
    a.res_before_act = p->res_before_act;
    a.in_act = p->in_act;
    a.alpha = p->alpha;
    a.out_scale = ext_scale(ext);
    if (route.kind == GemmRoute::GEMV) return small_gemv(a, p->dtype, st);
    return small_gemm_naive(a, p->dtype, st);
}

extern "C" int sfast_hip_gemm_grouped(const void *const *x, const void *const *w_segs, const void *const *bias, void *const *out,
                                      const sfast_gemm_params *p, int32_t n_groups, sfast_stream_t stream) {
    SFAST_REQUIRE(p && x && w_segs && out, SFAST_ERR_INVALID, "gemm_grouped: null argument");
    SFAST_REQUIRE(n_groups >= 1 && n_groups <= SFAST_MAX_GEMM_GROUPS, SFAST_ERR_INVALID, "gemm_grouped: n_groups=%d (1..%d)", n_groups,
                  SFAST_MAX_GEMM_GROUPS);
    SFAST_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, SFAST_ERR_INVALID, "gemm_grouped: bad shape %dx%dx%d", p->M, p->N, p->K);
    SFAST_REQUIRE(p->n_wseg >= 1 && p->n_wseg <= 2 && (int64_t)p->rows_per_seg * p->n_wseg >= p->N, SFAST_ERR_INVALID,
                  "gemm_grouped: %d segments of %d rows do not cover N=%d (at most 2 segments)", p->n_wseg, p->rows_per_seg, p->N);
    SFAST_REQUIRE(!p->geglu && p->split_k <= 1 && p->rows_per_batch == 0 && p->in_act == SFAST_ACT_NONE, SFAST_ERR_UNSUPPORTED,
                  "gemm_grouped: geglu / split-K / rowbias / in_act are not available in grouped launches");
    SFAST_REQUIRE(is_half(p->dtype), SFAST_ERR_UNSUPPORTED, "gemm_grouped: dtype %d", p->dtype);
    SFAST_REQUIRE(p->ldx >= p->K && p->ldw >= p->K && p->ldo >= p->N, SFAST_ERR_INVALID, "gemm_grouped: bad leading dims");
    bool ok = p->K % 8 == 0 && p->ldx % 8 == 0 && p->ldw % 8 == 0 && p->N % 4 == 0 && p->ldo % 4 == 0;
    for (int i = 0; i < n_groups; ++i) {
        SFAST_REQUIRE(out[i] && x[i], SFAST_ERR_INVALID, "gemm_grouped: null input / output %d", i);
        ok = ok && aligned16(x[i]) && aligned8(out[i]) && (!bias || !bias[i] || aligned8(bias[i]));
        for (int j = 0; j < p->n_wseg; ++j) {
            SFAST_REQUIRE(w_segs[i * p->n_wseg + j], SFAST_ERR_INVALID, "gemm_grouped: null weight %d/%d", i, j);
            ok = ok && aligned16(w_segs[i * p->n_wseg + j]);
        }
    }
    SFAST_REQUIRE(ok, SFAST_ERR_UNSUPPORTED, "gemm_grouped: operands must be 16-byte aligned rows (K, ldx, ldw % 8, N, ldo % 4)");
    IgemmArgs a{};
    a.x = x[0];
    a.x2 = nullptr;
    a.rowbias = nullptr;
    a.res = nullptr;
    a.M = p->M;
    a.N = p->N;
    a.K = p->K;
    a.ldx = p->ldx;
    a.ldw = p->ldw;
    a.ldo = p->ldo;
    a.ldr = 0;
    a.ld_rowbias = 0;
    a.rows_per_seg = p->rows_per_seg;
    a.rows_per_batch = 1;
    a.act = p->act;
    a.res_before_act = 0;
    a.alpha = 1.0f;
    return igemm_run_grouped(a, p->dtype, n_groups, x, w_segs, p->n_wseg, bias, out, (hipStream_t)stream);
}

extern "C" int sfast_hip_qlinear_w8(const void *x, const void *w_int8, const void *bias, void *out, const sfast_gemm_params *p,
                                    float dq_scale, sfast_stream_t stream) {
    SFAST_REQUIRE(p && x && w_int8 && out, SFAST_ERR_INVALID, "qlinear_w8: null argument");
    SFAST_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, SFAST_ERR_INVALID, "qlinear_w8: bad shape %dx%dx%d", p->M, p->N, p->K);
    SFAST_REQUIRE(is_half(p->dtype), SFAST_ERR_UNSUPPORTED, "qlinear_w8: activations must be f16 / bf16 (dtype %d)", p->dtype);
    SFAST_REQUIRE(!p->geglu && p->n_wseg <= 1 && p->rows_per_batch == 0 && p->in_act == SFAST_ACT_NONE, SFAST_ERR_UNSUPPORTED,
                  "qlinear_w8: plain linear (+bias, +activation) only");
    SFAST_REQUIRE(p->K % 8 == 0 && p->ldx % 8 == 0 && p->ldw % 8 == 0 && p->ldw >= p->K && p->ldx >= p->K && p->N % 4 == 0 && p->ldo % 4 == 0 &&
                      p->ldo >= p->N && aligned16(x) && aligned8(w_int8) && aligned8(out) && (!bias || aligned8(bias)),
                  SFAST_ERR_UNSUPPORTED, "qlinear_w8: K, ldx, ldw (bytes) must be multiples of 8, N and ldo of 4, operands aligned");
    IgemmArgs a{};
    a.x = x;
    for (int i = 0; i < SFAST_MAX_WSEG; ++i) a.w[i] = w_int8;
    a.bias = bias;
    a.out = out;
    a.M = p->M;
    a.N = p->N;
    a.K = p->K;
    a.ldx = p->ldx;
    a.ldw = p->ldw;
    a.ldo = p->ldo;
    a.rows_per_seg = p->N;
    a.rows_per_batch = 1;
    a.act = p->act;
    a.res_before_act = 0;
    a.alpha = 1.0f;
    a.out_scale = dq_scale;
    return igemm_run_w8(a, p->dtype, (hipStream_t)stream);
}

extern "C" size_t sfast_hip_conv2d_workspace_bytes(const sfast_conv_params *p) {
    if (!p || !is_half(p->dtype) || p->Cout < 16) return 0;
    ConvGeom g = conv_geom(p, nullptr);
    const int64_t M = (int64_t)p->B * g.Ho * g.Wo;
    const int K = p->KH * p->KW * p->Cin;
    if (M <= 0 || M > INT32_MAX || K % 8 != 0) return 0;
    return igemm_workspace_bytes((int)M, p->Cout, K, false, p->variant < 100 ? p->variant : 0, p->split_k, conv_caps(p));
}

extern "C" int sfast_hip_conv2d(const void *x, const void *x2, const void *w, const void *bias, const void *rowbias,
                                const void *z, void *out, const sfast_conv_params *p, void *workspace,
                                size_t workspace_bytes, sfast_stream_t stream) {
    return sfast_hip_conv2d_ex(x, x2, w, bias, rowbias, z, out, p, nullptr, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int sfast_hip_conv2d_stats_layout(const sfast_conv_params *p, const sfast_epilogue_ext *ext, sfast_gn_stats_layout *out) {
    SFAST_REQUIRE(p && ext && out, SFAST_ERR_INVALID, "conv2d_stats_layout: null argument");
    const ConvGeom g = conv_geom(p, nullptr);
    const int64_t M = (int64_t)p->B * g.Ho * g.Wo;
    const int K = p->KH * p->KW * p->Cin;
    const int C2 = p->Cin - p->C1;
    // the conditions of conv_route()'s MFMA branch that do not depend on pointers (dense NHWC activations, K-contiguous weights)
    const bool igemm = is_half(p->dtype) && p->Cout >= 16 && p->Cout % 8 == 0 && M > 0 && M <= INT32_MAX && K % 8 == 0 && p->C1 % 8 == 0 &&
                       C2 % 8 == 0 && g.x_dense && g.x2_dense && g.w_kcontig && g.out_dense && g.ldo % 8 == 0 && p->variant < 100;
    StatsLayout l{};
    if (!igemm || !igemm_stats_layout((int)M, p->Cout, K, false, p->variant, p->split_k, conv_caps(p), ext->gn_unit, ext->gn_rows_per_sample, ext_tickets(ext), l)) {
        set_error("conv2d_stats_layout: this problem / kernel choice cannot emit GroupNorm statistics");
        return SFAST_ERR_UNSUPPORTED;
    }
    out->rb_rows = l.rb_rows;
    out->n_rb = l.n_rb;
    out->bno = l.bno;
    out->tiles_n = l.tiles_n;
    out->slots = l.slots;
    out->unit = ext->gn_unit;
    return SFAST_OK;
}

// ---- GroupNorm(+SiLU) -> 3x3 conv as one weight-streaming launch (gnconv.hip) ----------------------------------------------------
static bool gn_conv_shape_ok(const sfast_gn_conv_params *q, sfast::GnConvPlan &pl) {
    if (!q) return false;
    const sfast_conv_params *p = &q->conv;
    if (!is_half(p->dtype) || p->KH != 3 || p->KW != 3 || p->stride_h != 1 || p->stride_w != 1 || p->pad_h != 1 || p->pad_w != 1 ||
        p->dil_h != 1 || p->dil_w != 1 || p->upsample2x || p->pad_h_extra || p->pad_w_extra || p->variant >= 100)
        return false;
    if (!(q->gn_act == SFAST_ACT_NONE || q->gn_act == SFAST_ACT_SILU)) return false;
    const ConvGeom g = conv_geom(p, nullptr);
    if (!(g.x_dense && g.x2_dense && g.w_kcontig && g.out_dense && g.ldo % 4 == 0 && p->Cout % 32 == 0)) return false;
    return sfast::gnconv_plan(p->B, p->H, p->W, p->C1, p->Cin - p->C1, p->Cout, q->groups, pl);
}

extern "C" int sfast_hip_gn_conv2d_supported(const sfast_gn_conv_params *q) {
    sfast::GnConvPlan pl{};
    return gn_conv_shape_ok(q, pl) ? 1 : 0;
}

extern "C" size_t sfast_hip_gn_conv2d_workspace_bytes(const sfast_gn_conv_params *q) {
    sfast::GnConvPlan pl{};
    return gn_conv_shape_ok(q, pl) ? pl.slab_bytes : 0;
}

extern "C" int sfast_hip_gn_conv2d(const void *x, const void *x2, const void *gamma, const void *beta, const void *w, const void *bias,
                                   const void *rowbias, const void *z, void *out, const sfast_gn_conv_params *q, void *workspace,
                                   size_t workspace_bytes, sfast_stream_t stream) {
    SFAST_REQUIRE(q, SFAST_ERR_INVALID, "gn_conv2d: null parameters");
    const sfast_conv_params *p = &q->conv;
    int rc = validate_conv(x, x2, w, out, p);
    if (rc) return rc;
    sfast::GnConvPlan pl{};
    SFAST_REQUIRE(gn_conv_shape_ok(q, pl), SFAST_ERR_UNSUPPORTED,
                  "gn_conv2d: outside the fused kernel's coverage (ask sfast_hip_gn_conv2d_supported; run sfast_hip_group_norm + sfast_hip_conv2d instead)");
    const ConvGeom g = conv_geom(p, z);
    bool fold = false;
    const ConvKind kind = conv_route(x, x2, w, bias, rowbias, z, out, p, g, fold);
    SFAST_REQUIRE(kind == CONV_IGEMM && aligned16(out) && (!gamma || aligned16(gamma)) && (!beta || aligned16(beta)), SFAST_ERR_UNSUPPORTED,
                  "gn_conv2d: operand alignment / strides outside the fused kernel's coverage");
    IgemmArgs a{};
    a.x = x;
    a.x2 = x2;
    for (int i = 0; i < SFAST_MAX_WSEG; ++i) a.w[i] = w;
    a.bias = bias;
    a.rowbias = fold ? z : rowbias;
    a.res = fold ? nullptr : z;
    a.out = out;
    a.M = p->B * p->H * p->W;
    a.N = p->Cout;
    a.K = 9 * p->Cin;
    a.ldw = a.K;
    a.ldo = g.ldo;
    a.ldr = g.ldr;
    a.ld_rowbias = fold ? p->zs[0] : p->ld_rowbias;
    a.rows_per_seg = p->Cout;
    a.rows_per_batch = p->H * p->W;
    a.act = p->act;
    a.res_before_act = p->res_before_act;
    a.alpha = p->alpha;
    a.H = p->H;
    a.W = p->W;
    a.C1 = p->C1;
    a.C2 = p->Cin - p->C1;
    a.Ho = p->H;
    a.Wo = p->W;
    a.KH = a.KW = 3;
    a.stride_h = a.stride_w = a.pad_h = a.pad_w = a.dil_h = a.dil_w = 1;
    a.out_scale = 1.0f;
    return sfast::gnconv_run(a, p->dtype, p->B, gamma, beta, q->groups, q->eps, q->gn_act == SFAST_ACT_SILU ? 1 : 0, workspace, workspace_bytes,
                             (hipStream_t)stream);
}

extern "C" int sfast_hip_conv2d_ex(const void *x, const void *x2, const void *w, const void *bias, const void *rowbias,
                                   const void *z, void *out, const sfast_conv_params *p, const sfast_epilogue_ext *ext, void *gn_stats,
                                   void *workspace, size_t workspace_bytes, sfast_stream_t stream) {
    int rc = validate_conv(x, x2, w, out, p);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const ConvGeom g = conv_geom(p, z);
    SFAST_REQUIRE(g.Ho > 0 && g.Wo > 0, SFAST_ERR_INVALID, "conv2d: empty output %dx%d", g.Ho, g.Wo);
    const int64_t M64 = (int64_t)p->B * g.Ho * g.Wo;
    SFAST_REQUIRE(M64 <= INT32_MAX, SFAST_ERR_UNSUPPORTED, "conv2d: too many output pixels");
    bool fold = false;
    const ConvKind kind = conv_route(x, x2, w, bias, rowbias, z, out, p, g, fold);
    SFAST_REQUIRE(!gn_stats || (ext && kind == CONV_IGEMM), SFAST_ERR_UNSUPPORTED,
                  "conv2d: GroupNorm statistics are emitted by the MFMA path only (and need an sfast_epilogue_ext)");
    SFAST_REQUIRE(!(ext && ext->gn_out) || kind == CONV_IGEMM, SFAST_ERR_UNSUPPORTED,
                  "conv2d: the fused GroupNorm epilogue (sfast_epilogue_ext.gn_out) exists on the split-K MFMA path only");
    if (kind == CONV_IGEMM) {
        IgemmArgs a{};
        a.x = x;
        a.x2 = x2;
        for (int i = 0; i < SFAST_MAX_WSEG; ++i) a.w[i] = w;
        a.bias = bias;
        a.rowbias = fold ? z : rowbias;
        a.res = fold ? nullptr : z;
        a.out = out;
        a.M = (int)M64;
        a.N = p->Cout;
        a.K = p->KH * p->KW * p->Cin;
        a.ldx = 0;
        a.ldw = a.K;
        a.ldo = g.ldo;
        a.ldr = g.ldr;
        a.ld_rowbias = fold ? p->zs[0] : p->ld_rowbias;
        a.rows_per_seg = p->Cout;
        a.rows_per_batch = g.Ho * g.Wo;
        a.act = p->act;
        a.res_before_act = p->res_before_act;
        a.alpha = p->alpha;
        a.H = p->H;
        a.W = p->W;
        a.C1 = p->C1;
        a.C2 = p->Cin - p->C1;
        a.Ho = g.Ho;
        a.Wo = g.Wo;
        a.KH = p->KH;
        a.KW = p->KW;
        a.stride_h = p->stride_h;
        a.stride_w = p->stride_w;
        a.pad_h = p->pad_h;
        a.pad_w = p->pad_w;
        a.dil_h = p->dil_h;
        a.dil_w = p->dil_w;
        a.ups = p->upsample2x;
        a.out_scale = ext_scale(ext);
        a.gn_stats = (float *)gn_stats;
        a.gn_unit = ext ? ext->gn_unit : 0;
        a.gn_rows_per_sample = ext ? ext->gn_rows_per_sample : 0;
        if (ext && ext->gn_out) {
            SFAST_REQUIRE(ext->gn_rows_per_sample == g.Ho * g.Wo, SFAST_ERR_INVALID, "conv2d: gn_rows_per_sample must be Ho * Wo");
            rc = ext_gn(ext, gn_stats, a);
            if (rc) return rc;
        }
        ext_packed(ext, 1, a);
        split_workspace(ext, workspace, workspace_bytes, a);
        return igemm_run(a, p->dtype, 1, false, p->variant < 100 ? p->variant : 0, p->split_k, workspace, workspace_bytes, st);
    }
    SmallConvArgs a{};
    fill_small_conv(a, x, x2, w, bias, rowbias, z, out, p, g);
    a.out_scale = ext_scale(ext);
    if (kind == CONV_SMALL_N) return small_conv_n(a, p->dtype, st);
    if (kind == CONV_SMALL_C) return small_conv_c(a, p->dtype, st);
    return small_conv_naive(a, p->dtype, st);
}
