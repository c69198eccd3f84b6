/*
 * sfast_hip.h -- C ABI of libsfast_hip.so, the MI355X (gfx950 / CDNA4) kernel
 * library behind the stable-fast diffusion-UNet hot path.
 *
 * This is the drop-in boundary: every entry point below replaces one operator
 * family the reference binds through TORCH_LIBRARY / its Python op registry
 * (paths relative to the reference tree, src/sfast/):
 *
 *   sfast_hip_group_norm   <- sfast_triton::group_norm / group_norm_silu
 *                             (triton/torch_ops.py:108-238, triton/ops/group_norm.py:352-479)
 *   sfast_hip_layer_norm   <- sfast_triton::layer_norm
 *                             (triton/torch_ops.py:241-255, triton/ops/layer_norm.py:276-322)
 *   sfast_hip_gemm         <- sfast::cublas_lowp_linear / _linear_add / _linear_relu /
 *                             _linear_gelu / _addmm* / _mm / _matmul
 *                             (csrc/operators/cublas/cublas_gemm.h:12-49),
 *                             sfast::linear_relu / linear_gelu (csrc/operators/fused_linear.h:11-15),
 *                             sfast::cutlass_linear_geglu[_unified]
 *                             (csrc/operators/cutlass/cutlass_dual_linear_kernel.h:6-15)
 *   sfast_hip_conv2d       <- sfast::cudnn_convolution_bias[_add][_sigmoid|_relu|_tanh]
 *                             (csrc/operators/cudnn/cudnn_convolution.h:12-78)
 *   sfast_hip_attention    <- sfast_xformers::memory_efficient_attention
 *                             (libs/xformers/xformers_attention.py:26-48), q/k/v as [B,S,H,D]
 *                             strided views (libs/diffusers/xformers_attention.py:37-69)
 *   sfast_hip_strided_copy <- sfast_triton::contiguous / clone / reshape
 *                             (triton/torch_ops.py:24-106, triton/ops/copy.py:184-270)
 *   sfast_hip_softmax_rows  <- the VAE decoder's single-head attention (compile_vae path), between two sfast_hip_gemm calls
 *   sfast_hip_add_strided   <- ControlNet residual adds of UNet2DConditionModel.forward
 *   sfast_hip_image_postprocess <- patched VaeImageProcessor (libs/diffusers/image_processor.py:13-108)
 *   sfast_hip_timestep_embedding, sfast_hip_cfg_ddim_step, sfast_hip_linear_step, sfast_hip_schedule_advance
 *                          <- host-side glue of the denoise loop that the reference leaves to
 *                             diffusers / trace_scheduler (compilers/diffusion_pipeline_compiler.py:103-107)
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer except the params struct and `w_segs` is a DEVICE pointer
 *     owned by the caller; nothing is allocated, freed or synchronised inside -> safe to record
 *     into a hipGraph; work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - return SFAST_OK (0) or a negative error; sfast_hip_last_error() gives a thread-local message.
 *   - scratch memory is caller-provided: ask sfast_hip_<op>_workspace_bytes(params) first
 *     (0 means the pointer may be NULL).
 *   - inputs are never written; weights are read from the caller's (live) storage on every call.
 *   - arithmetic: f16/bf16 (and f32 on the generic kernels) I/O, fp32 accumulation and fp32
 *     epilogue math everywhere (CDNA4 MFMA accumulates in fp32 only).
 */
#ifndef SFAST_HIP_H
#define SFAST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFAST_HIP_ABI_VERSION 9

typedef void *sfast_stream_t; /* hipStream_t */

enum sfast_status {
    SFAST_OK = 0,
    SFAST_ERR_UNSUPPORTED = -1, /* shape / dtype / alignment not handled by any kernel */
    SFAST_ERR_INVALID = -2,     /* inconsistent parameters */
    SFAST_ERR_WORKSPACE = -3,   /* workspace too small / NULL */
    SFAST_ERR_LAUNCH = -4       /* hipLaunchKernel reported an error */
};

enum sfast_dtype { SFAST_F16 = 0, SFAST_BF16 = 1, SFAST_F32 = 2 };

enum sfast_act {
    SFAST_ACT_NONE = 0,
    SFAST_ACT_RELU = 1,
    SFAST_ACT_GELU = 2,      /* erf form, aten::gelu(approximate='none') */
    SFAST_ACT_GELU_TANH = 3, /* tanh form */
    SFAST_ACT_SILU = 4,
    SFAST_ACT_SIGMOID = 5,
    SFAST_ACT_TANH = 6
};

/* ---- library ---------------------------------------------------------------------------- */
int sfast_hip_abi_version(void);
/* one-time per-process setup (kernel attributes); idempotent, cheap, call before capturing. */
int sfast_hip_init(void);
const char *sfast_hip_last_error(void);
/* name of the kernel variant chosen by the most recent gemm / conv2d / attention call on this
 * thread (diagnostics, tests and bench roofline bookkeeping). */
const char *sfast_hip_last_kernel(void);
/* profiling: while `buf` is non-NULL every MFMA GEMM workgroup writes 16 uint64 (100 MHz wall-clock stamps of
 * its phases, HW_ID, shader-clock counter at entry/exit) at buf[(blockIdx.y*gridDim.x+blockIdx.x)*16]; the caller sizes buf for the launch
 * (sfast_hip_igemm_plan gives the grid). NULL (default) = production behaviour. */
int sfast_hip_set_trace(void *buf);
/* 1 when the library was built with -DSFAST_PROBES (stable-fast_amd/build.py --probes -> libsfast_hip_probes.so): timing-only
 * experiment / ablation instantiations whose RESULTS ARE GARBAGE (SFAST_IGEMM_EXP, attention variant >= 1000), the never-selected
 * LDS-patch conv pipe and the in-kernel split-K join (SFAST_EXT_WS_TICKETS) exist only there. The product library returns 0:
 * no environment variable or parameter of this ABI can make it run a kernel whose output is not the operator's result
 * (attention variant >= 1000 -> SFAST_ERR_UNSUPPORTED; SFAST_EXT_WS_TICKETS is accepted and ignored: the reduce launch runs).
 * ABI 6. Replaces nothing in the reference -- build hygiene of this library. */
int sfast_hip_has_probes(void);

/* ---- GroupNorm (+SiLU) ------------------------------------------------------------------ */
enum sfast_layout { SFAST_NHWC = 0, SFAST_NCHW = 1 };

typedef struct {
    int32_t dtype;  /* sfast_dtype (f16/bf16/f32) */
    int32_t layout; /* sfast_layout: NHWC = dense channels_last, NCHW = contiguous */
    int32_t N, C, HW, G;
    int32_t C1;  /* NHWC only: channels [0,C1) are read from x (pitch C1), [C1,C) from x2
                    (pitch C-C1): a virtual channel concat. C1 == C -> x only. */
    int32_t act; /* SFAST_ACT_NONE or SFAST_ACT_SILU */
    float eps;
} sfast_gn_params;

size_t sfast_hip_group_norm_workspace_bytes(const sfast_gn_params *p);
int sfast_hip_group_norm(const void *x, const void *x2, const void *gamma, const void *beta,
                         void *y, const sfast_gn_params *p, void *workspace,
                         size_t workspace_bytes, sfast_stream_t stream);

/* ---- LayerNorm --------------------------------------------------------------------------- */
typedef struct {
    int32_t dtype;
    int32_t M, N; /* x[M,N] contiguous rows, normalised over N */
    float eps;
} sfast_ln_params;

int sfast_hip_layer_norm(const void *x, const void *gamma, const void *beta, void *y,
                         const sfast_ln_params *p, sfast_stream_t stream);

/* ---- row softmax: y[m][:] = softmax(scale * x[m][:]) over N, fp32 math ------------------------
 * The VAE decoder's single-head attention (head dim 512, /root/reference compile_vae path,
 * compilers/diffusion_pipeline_compiler.py:154-190 -> diffusers AttnProcessor) runs as
 * GEMM(Q.K^T) -> this kernel -> GEMM(P.V): its head dim is outside the flash kernel's register budget. */
typedef struct {
    int32_t dtype; /* f16 / bf16 */
    int32_t M, N;  /* N % 8 == 0 */
    int64_t ldx, ldy; /* row strides in elements (multiples of 8); y may alias x */
    float scale;
} sfast_softmax_params;

int sfast_hip_softmax_rows(const void *x, void *y, const sfast_softmax_params *p, sfast_stream_t stream);

/* ---- GEMM: out[M,N] = epilogue(x[M,K] . W[N,K]^T) ----------------------------------------
 * epilogue (fp32):  v = acc + bias[n] + rowbias[m / rows_per_batch][n]
 *                   res_before_act:  out = act(v + alpha*res[m][n])     (cuDNN-style z add)
 *                   otherwise     :  out = act(v) + alpha*res[m][n]     (cublas_lowp_linear_add)
 * geglu: W is [2N,K]; rows [0,N) produce h, rows [N,2N) produce g; bias likewise [2N];
 *        out = h * gelu(g)   (rewritten pattern jit/passes/__init__.py:643-649, halves order
 *        cutlass_dual_linear_kernel.cu:531-538)
 * W may be given as n_wseg stacked segments of rows_per_seg rows each (e.g. the live to_q /
 * to_k / to_v weights of one attention block) -> one launch, no concatenated copy.           */
#define SFAST_MAX_WSEG 4
typedef struct {
    int32_t dtype;
    int32_t M, N, K;
    int64_t ldx, ldw, ldo, ldr; /* row strides in elements (x, W, out, residual) */
    int32_t n_wseg, rows_per_seg;
    int32_t geglu;
    int32_t act;
    int32_t res_before_act;
    float alpha;
    int32_t rows_per_batch; /* 0 = no rowbias */
    int64_t ld_rowbias;
    int32_t in_act;  /* activation applied to x on load; small-M (M <= 16) path only */
    int32_t variant; /* 0 = auto; otherwise force a kernel variant (tuning / tests) */
    int32_t split_k; /* 0 = auto */
} sfast_gemm_params;

size_t sfast_hip_gemm_workspace_bytes(const sfast_gemm_params *p);
int sfast_hip_gemm(const void *x, const void *const *w_segs, const void *bias,
                   const void *rowbias, const void *residual, void *out,
                   const sfast_gemm_params *p, void *workspace, size_t workspace_bytes,
                   sfast_stream_t stream);

/* ---- weight-only int8 linear: out = act(dq_scale * (x . Wq^T) + bias), Wq int8 [N][K] -------------------------------
 * sfast::cutlass_qlinear_dynamic / quantized::linear_dynamic of the reference (csrc/operators/cutlass/cutlass_qlinear.cc:73-89,
 * cutlass_qlinear_dynamic_kernel.cu:259-294: per-tensor scale = weight.q_scale(), zero point ignored, f16 / bf16 activations,
 * fp32 accumulate). p: M, N, K, ldx, ldo in elements, ldw in BYTES (= elements of the int8 matrix), act; the rest must be 0. */
int sfast_hip_qlinear_w8(const void *x, const void *w_int8, const void *bias, void *out, const sfast_gemm_params *p,
                         float dq_scale, sfast_stream_t stream);

/* ---- grouped GEMM: n_groups problems of identical shape in one launch ----------------------------------------
 * out_g[M,N] = act(x_g[M,K] . W_g[N,K]^T + bias_g),  g = 0 .. n_groups-1  (<= SFAST_MAX_GEMM_GROUPS).
 * Two callers: (1) the cross-attention K/V projections of all transformer blocks of one UNet level, which read the SAME text
 * context (every x_g the same pointer; one sfast::cublas_lowp_linear per to_k / to_v in the reference,
 * csrc/operators/cublas/cublas_gemm.cpp:798-860); (2) sfast::cublas_lowp_bmm / _baddbmm / batched _matmul (cublas_gemm.h:30-38):
 * one group per batch element. p describes ONE problem (M, N, K, ldx, ldw, ldo, n_wseg <= 2 stacked segments of rows_per_seg
 * rows, act; no residual / rowbias / geglu / split-K). `x`, `bias` (or NULL) and `out` are HOST arrays of n_groups DEVICE
 * pointers, `w_segs` a host array of n_groups * n_wseg device pointers (group-major). K % 8 == 0, N % 4 == 0. */
#define SFAST_MAX_GEMM_GROUPS 64
int sfast_hip_gemm_grouped(const void *const *x, const void *const *w_segs, const void *const *bias, void *const *out,
                           const sfast_gemm_params *p, int32_t n_groups, sfast_stream_t stream);

/* ---- grouped GEMV: n_groups independent weight matrices W_g[n_rows[g], K] applied to ONE small input ----
 * out[m][off_g + n] = act( sum_k in_act(x[m][k]) * W_g[n][k] + bias_g[n] ),  off_g = n_rows[0] + ... + n_rows[g-1].
 * One launch for the 22 `time_emb_proj` Linear layers of the UNet's resnets: each is a
 * sfast::cublas_lowp_linear call in the reference (csrc/operators/cublas/cublas_gemm.cpp:798-860) on the same
 * silu(emb) input, i.e. pure weight streaming that depends only on the timestep. `w` / `bias` are HOST arrays of
 * n_groups DEVICE pointers (bias may be NULL, or hold NULL entries); the weights are read in place. M <= 64. */
#define SFAST_MAX_GROUPS 32
typedef struct {
    int32_t dtype; /* f16 / bf16 */
    int32_t M, K;
    int32_t n_groups;
    int32_t n_rows[SFAST_MAX_GROUPS];
    int64_t ldx, ldw, ldo; /* row strides in elements: x, every W_g, out */
    int32_t act, in_act;
} sfast_gemv_grouped_params;

int sfast_hip_gemv_grouped(const void *x, const void *const *w, const void *const *bias, void *out,
                           const sfast_gemv_grouped_params *p, sfast_stream_t stream);

/* diagnostic, host-only: the tile / split-K choice the MFMA path would make for an [M,N,K] problem.
 * out = {BM, BN (weight rows per tile), splits, k_tiles_per_split, variant id}; variant ids 1..5 are the
 * register-staged pipe, 11..18 the LDS-DMA ring (same tile shapes, ring depths 2..5), 21..23 the wave-specialised
 * LDS-DMA pipe (producer waves + consumer waves), 31..34 the LDS-resident-patch conv pipe (probe build), 41..46 the packed-weight
 * pipe (needs sfast_epilogue_ext.w_packed), 51..58 the 256-row tiles of pipe 5 (csrc/igemm_pp.h; K % 64 == 0 and M >= 256 only):
 * 51 / 52 / 53 = 256 x 128 / 160 / 256 eight-wave ping-pong, 55 / 56 = 256 x 128 / 160 with four producer waves, 57 / 58 = the same with
 * lockstep consumers (one barrier per K-tile; GEGLU: 53 and 57). Ids >= 16 are never chosen by the analytic model: they are measured
 * candidates of the caller's autotuner (sfast/engine/autotune.py). */
int sfast_hip_igemm_plan(int32_t M, int32_t N, int32_t K, int32_t geglu, int32_t variant, int32_t split_k,
                         int32_t out[5]);

/* ---- conv2d (cross-correlation, groups = 1) ------------------------------------------------
 * y = act(conv(x, w) + bias + rowbias[b] + alpha*z)  (res_before_act = 1, cuDNN fused form,
 *      cudnn_convolution_impl.cc:995-998) or act(...) + alpha*z (res_before_act = 0).
 * Tensors are described by strides, order (n, h, w, c), so NHWC (channels_last) and NCHW are
 * both expressible; the MFMA implicit-GEMM path needs dense NHWC activations and K-contiguous
 * (channels_last) weights, everything else runs on the generic kernel.
 * upsample2x: the conv reads nearest-neighbour 2x upsampled x without materialising it.
 * C1 < Cin: input channels [C1,Cin) come from x2 (virtual concat, as in GroupNorm).
 * Output size: Ho = (Hin + 2*pad_h + pad_h_extra - dil_h*(KH-1) - 1) / stride_h + 1 (Hin = 2H when upsample2x). */
typedef struct {
    int32_t dtype;
    int32_t B, H, W, Cin, Cout, KH, KW;
    int32_t stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
    int32_t upsample2x;
    int32_t C1;
    int64_t xs[4], x2s[4]; /* element strides (n,h,w,c) */
    int64_t ws[4];         /* weight strides (o,i,kh,kw) */
    int64_t os[4];         /* output strides (n,h,w,c) */
    int64_t zs[4];         /* residual strides (n,h,w,c); 0 = broadcast */
    int32_t act, res_before_act;
    float alpha;
    int64_t ld_rowbias; /* rowbias[b][cout] row stride; ignored when rowbias == NULL */
    int32_t variant, split_k;
    /* ABI 2: extra zero padding at the bottom / right on top of the symmetric pad_h / pad_w (0 = symmetric).
     * diffusers' Downsample2D in the VAE encoder pads (0,1,0,1) before a stride-2 conv: pad 0, extra 1. */
    int32_t pad_h_extra, pad_w_extra;
} sfast_conv_params;

size_t sfast_hip_conv2d_workspace_bytes(const sfast_conv_params *p);
int sfast_hip_conv2d(const void *x, const void *x2, const void *w, const void *bias,
                     const void *rowbias, const void *z, void *out, const sfast_conv_params *p,
                     void *workspace, size_t workspace_bytes, sfast_stream_t stream);

/* ---- epilogue extensions of the GEMM / conv families (sfast_hip_gemm_ex, sfast_hip_conv2d_ex) --------------------------
 * out_scale : v = out_scale * acc + bias + ...  (0 is read as 1). The VAE's single-head attention scales q.k^T BEFORE the f16
 *             store this way (diffusers applies the scale inside the fp32 product; unscaled f16 logits can overflow).
 * gn_unit / gn_rows_per_sample + the `gn_stats` pointer: the epilogue additionally reduces its output tile to GroupNorm partial
 *             statistics -- one float2 {mean, M2} per (row block, tile_n, unit slot), sfast_gn_stats_layout -- so that the
 *             GroupNorm that consumes this tensor is ONE normalisation pass (sfast_hip_group_norm_apply) instead of a statistics
 *             kernel plus a normalisation kernel (reference: triton/ops/group_norm.py:111-165 + :272-349, two launches).
 *             gn_unit = channels per statistics unit: >= 8, a divisor of every consumer's C/G (and of the channel offset at
 *             which the tensor sits inside a virtual concat); gn_rows_per_sample = H*W of the output. The records of one tensor
 *             can serve several consumers (e.g. a skip connection normalised alone and again inside a concat).
 * flags & SFAST_EXT_WS_TICKETS: split-K problems are finished INSIDE the GEMM kernel instead of by a second (reduce + epilogue)
 *             launch: the workgroups of a tile leave fp32 partials in the workspace, draw a ticket from the tile's counter, and the
 *             one that draws the last ticket sums the partials in split order (bitwise reproducible, same order as the reduce
 *             kernel) and runs the epilogue. The caller promises that the LAST SFAST_WS_TICKET_BYTES of
 *             [workspace, workspace + workspace_bytes) are ticket counters: zeroed once (sfast_hip_workspace_init), never written
 *             by anybody else, and that every call sharing the workspace passes the same workspace_bytes. The library leaves
 *             them zero after every launch. sfast_hip_*_workspace_bytes already includes the block for split-K problems; a
 *             workspace shared with other operators should be sized max(needs) + SFAST_WS_TICKET_BYTES so that no operator's
 *             scratch reaches into it. Without the flag nothing changes (row-major slabs + reduce kernel).                    */
#define SFAST_WS_TICKET_BYTES 65536
#define SFAST_EXT_WS_TICKETS 1
/* gn_out (ABI 8)   : the GroupNorm(+SiLU) that CONSUMES the output rides in the split-K reduce launch: besides `out`, the dense
 *             [M][N] tensor gn_act(GroupNorm(out; gn_groups, gn_eps) * gn_gamma + gn_beta) is written to gn_out -- statistics over
 *             the STORED (rounded) output per (sample, group), sample = gn_rows_per_sample consecutive rows; the arithmetic contract
 *             of sfast_hip_group_norm on `out`. Replaces the separate launch of the reference's fused GroupNorm
 *             (triton/torch_ops.py:179-189) behind a split-K conv / GEMM at the low-resolution levels. Only for problems that run
 *             split-K (ask sfast_hip_conv2d_plan / sfast_hip_igemm_plan: out[2] > 1) with (N / gn_groups) % 4 == 0 and
 *             gn_rows_per_sample * N / gn_groups <= 16384; otherwise SFAST_ERR_UNSUPPORTED. Not combinable with gn_unit / gn_stats.
 *             ROUND 5: measured slower in the SD1.5 step than reduce + separate GroupNorm (profiles/r04_reduce_gn_ab_run{7,8,9}.log), so
 *             the launch exists in the PROBE build only (libsfast_hip_probes.so); the product library answers SFAST_ERR_UNSUPPORTED for
 *             any non-NULL gn_out BEFORE it launches anything (`out` is left untouched).                                            */
/* w_packed (ABI 9): packed copies of the weight segments, made by sfast_hip_pack_weight from the live parameters. With them the
 *             planner may pick a pipe-4 kernel (variant ids 41 ..: weight fragments global -> VGPR in MFMA order, activations through
 *             the LDS ring; csrc/igemm_pk.h) -- the reference's vendor GEMMs / convs re-lay out filters inside the call as well.
 *             The caller keeps a packed copy as fresh as it needs (re-pack after the parameter changed); `w` is still passed and
 *             still read by every other pipe. Every segment needs a copy, and segments of a multi-segment weight must hold a
 *             multiple of 32 rows; otherwise the pipe is not eligible and the launch uses another one.                              */
typedef struct {
    float out_scale;
    int32_t gn_unit;
    int32_t gn_rows_per_sample;
    int32_t flags;
    void *gn_out;           /* NULL: no fused GroupNorm */
    const void *gn_gamma;   /* [N] or NULL */
    const void *gn_beta;    /* [N] or NULL */
    int32_t gn_groups;
    float gn_eps;
    int32_t gn_act;         /* SFAST_ACT_NONE or SFAST_ACT_SILU */
    int32_t reserved;
    const void *const *w_packed; /* ABI 9: n_wseg (conv: 1) packed copies of the weight segments (sfast_hip_pack_weight), or NULL */
} sfast_epilogue_ext;

/* zero the ticket block of a fresh workspace (asynchronous, on `stream`) */
int sfast_hip_workspace_init(void *workspace, size_t workspace_bytes, sfast_stream_t stream);

typedef struct {
    int32_t rb_rows; /* rows per row block (a tile's BM, or the rows a split-K reduce workgroup owns) */
    int32_t n_rb;    /* row blocks over all samples: rb_rows * n_rb == M */
    int32_t bno;     /* output columns per tile_n */
    int32_t tiles_n;
    int32_t slots;   /* records per (row block, tile_n) */
    int32_t unit;
} sfast_gn_stats_layout; /* buffer: n_rb * tiles_n * slots float2 */

/* host-only: the layout the (variant, split_k) choice carried by p would write; SFAST_ERR_UNSUPPORTED when that kernel cannot */
int sfast_hip_gemm_stats_layout(const sfast_gemm_params *p, const sfast_epilogue_ext *ext, sfast_gn_stats_layout *out);
int sfast_hip_conv2d_stats_layout(const sfast_conv_params *p, const sfast_epilogue_ext *ext, sfast_gn_stats_layout *out);

int sfast_hip_gemm_ex(const void *x, const void *const *w_segs, const void *bias, const void *rowbias, const void *residual,
                      void *out, const sfast_gemm_params *p, const sfast_epilogue_ext *ext /* or NULL */,
                      void *gn_stats /* or NULL */, void *workspace, size_t workspace_bytes, sfast_stream_t stream);
int sfast_hip_conv2d_ex(const void *x, const void *x2, const void *w, const void *bias, const void *rowbias, const void *z,
                        void *out, const sfast_conv_params *p, const sfast_epilogue_ext *ext /* or NULL */,
                        void *gn_stats /* or NULL */, void *workspace, size_t workspace_bytes, sfast_stream_t stream);

/* ---- GroupNorm(+SiLU) -> 3x3 conv as ONE weight-streaming launch (ABI 7) ----------------------------------------------------
 * out = act(conv3x3(gn_act(GroupNorm(cat(x, x2)))) + bias + rowbias[b] + alpha * z): the reference's PAIR
 * sfast_triton::group_norm[_silu] (triton/torch_ops.py:179-189) -> sfast::cudnn_convolution_bias[_add][_act]
 * (csrc/operators/cudnn/cudnn_convolution_impl.cc:995-998) in one call, for the UNet's low-resolution levels where the conv is
 * weight-streaming bound (B*H*W <= 128 pixels: a 1280 -> 1280 conv moves 29.5 MB of weights for 3.8 GFLOP) and a separate
 * normalisation launch costs as much as a third of that stream. Every workgroup keeps a channel slice of ALL pixels in LDS, computes
 * the exact two-pass statistics of the slice's groups itself and streams its weight rows global -> registers exactly once
 * (stable-fast_amd/csrc/gnconv.hip). Same arithmetic contract as sfast_hip_group_norm followed by sfast_hip_conv2d (fp32 statistics
 * and affine, f16/bf16 rounding of the normalised tensor, fp32 accumulate, the conv epilogue of sfast_conv_params).
 * Coverage (sfast_hip_gn_conv2d_supported == 1): f16 / bf16; 3x3, stride 1, padding 1, no dilation / upsample / extra padding;
 * dense NHWC sources, [Cout][3][3][Cin] weights, dense NHWC output; B*H*W <= 128; (Cin / groups) % 8 == 0; Cout % 32 == 0; a channel
 * slice of >= 80 channels that is a whole number of groups and of 16-channel steps and divides both concat sources. Everything
 * else: SFAST_ERR_UNSUPPORTED -- callers run the two operators. workspace: sfast_hip_gn_conv2d_workspace_bytes (fp32 slabs).
 * ROUND 5: measured 1 % slower in the SD1.5 step than the two operators (profiles/r04_gnconv_step_ab_run6.log): the kernel lives in the
 * PROBE build only. In the product library sfast_hip_gn_conv2d_supported() == 0 for every problem, the workspace query returns 0 and
 * sfast_hip_gn_conv2d() returns SFAST_ERR_UNSUPPORTED -- the entry points stay so that one binding serves both builds. */
typedef struct {
    sfast_conv_params conv; /* geometry, strides and epilogue of the convolution (of the NORMALISED input) */
    int32_t groups;         /* GroupNorm groups over conv.Cin channels; 0: no normalisation (the weight-streaming conv alone, a measured
                             * candidate that no plan selects: gamma / beta / eps / gn_act are ignored, Cin % 16 == 0) */
    float eps;
    int32_t gn_act;         /* SFAST_ACT_NONE or SFAST_ACT_SILU, applied after the affine */
} sfast_gn_conv_params;
int sfast_hip_gn_conv2d_supported(const sfast_gn_conv_params *p);
size_t sfast_hip_gn_conv2d_workspace_bytes(const sfast_gn_conv_params *p);
int sfast_hip_gn_conv2d(const void *x, const void *x2, const void *gamma, const void *beta, const void *w, const void *bias,
                        const void *rowbias, const void *z, void *out, const sfast_gn_conv_params *p, void *workspace,
                        size_t workspace_bytes, sfast_stream_t stream);

/* GroupNorm(+SiLU) of an NHWC tensor (optionally a virtual concat x | x2) whose statistics were left behind by the kernels that
 * produced x (stats1 / l1) and x2 (stats2 / l2): one pass, x read once. Records of different row blocks are merged with the
 * pairwise variance update in a fixed order (bitwise reproducible). Same arithmetic contract as sfast_hip_group_norm. */
int sfast_hip_group_norm_apply(const void *x, const void *x2, const void *gamma, const void *beta, void *y,
                               const sfast_gn_params *p, const void *stats1, const sfast_gn_stats_layout *l1,
                               const void *stats2, const sfast_gn_stats_layout *l2, sfast_stream_t stream);

/* ---- scaled-dot-product attention: out = softmax(q k^T * scale) v -------------------------
 * q [B,Sq,H,D], k/v [B,Skv,H,D], out [B,Sq,H,D]; element strides (b, s, h), d stride 1.       */
typedef struct {
    int32_t dtype;
    int32_t B, H, Sq, Skv, D;
    int64_t qs[3], ks[3], vs[3], os[3];
    float scale;
    int32_t variant;
} sfast_attn_params;

int sfast_hip_attention(const void *q, const void *k, const void *v, void *out,
                        const sfast_attn_params *p, sfast_stream_t stream);
/* out = softmax(q k^T * scale + bias) v: the `attn_bias` argument of sfast_xformers::memory_efficient_attention
 * (libs/xformers/xformers_attention.py:30-47) / diffusers' attention_mask. bias[b][h][q][key] in the dtype of q, key stride 1,
 * element strides bias_strides = (b, h, q), 0 = broadcast (a key-padding mask is (ld, 0, 0)); -inf entries mask a key.
 * The MFMA kernel reads the bias in groups of 4 keys with dword loads: it is taken when the strides are even, the base is
 * 4-byte aligned and every row is READABLE up to the next multiple of 4 keys (rows padded to 8 elements -- xformers' own
 * requirement on attn_bias -- satisfy all of it; the padding values are never used); other layouts run the generic kernel.
 * bias == NULL is sfast_hip_attention. */
int sfast_hip_attention_bias(const void *q, const void *k, const void *v, const void *bias, const int64_t *bias_strides,
                             void *out, const sfast_attn_params *p, sfast_stream_t stream);

/* ---- strided copy (rank <= 4) ------------------------------------------------------------- */
typedef struct {
    int32_t elem_bytes; /* 1, 2, 4, 8 */
    int32_t ndim;       /* <= 4 */
    int64_t shape[4];
    int64_t src_strides[4], dst_strides[4]; /* in elements */
} sfast_copy_params;

int sfast_hip_strided_copy(const void *src, void *dst, const sfast_copy_params *p,
                           sfast_stream_t stream);

/* ---- strided accumulate: dst[i0..i3] += src[i0..i3] (fp32 add) ------------------------------------
 * ControlNet residuals (diffusers UNet2DConditionModel.forward: `down_block_additional_residuals`,
 * `mid_block_additional_residual`; reference compile() keeps ControlNet pipelines working,
 * compilers/diffusion_pipeline_compiler.py:89-90) arrive NCHW and are added onto the engine's NHWC skip tensors. */
typedef struct {
    int32_t dtype;      /* f16 / bf16 / f32 (src and dst) */
    int32_t ndim;       /* <= 4 */
    int64_t shape[4];
    int64_t src_strides[4], dst_strides[4]; /* in elements */
} sfast_add_params;

int sfast_hip_add_strided(const void *src, void *dst, const sfast_add_params *p, sfast_stream_t stream);

/* ---- sinusoidal timestep embedding -> out[B, dim] ------------------------------------------ */
typedef struct {
    int32_t dtype; /* output dtype */
    int32_t B, dim;
    int32_t flip_sin_to_cos;
    float downscale_freq_shift;
    float max_period;
} sfast_temb_params;

int sfast_hip_timestep_embedding(const float *timesteps /* [B] device */, void *out,
                                 const sfast_temb_params *p, sfast_stream_t stream);

/* ---- classifier-free-guidance combine + DDIM update (eta = 0) -----------------------------
 * eps = eps_u + g (eps_c - eps_u); x0 = (x - sqrt(1-a_t) eps) / sqrt(a_t);
 * x' = sqrt(a_prev) x0 + sqrt(1-a_prev) eps.  eps_uc holds [eps_u ; eps_c] (2*numel).
 * x' is written to latents_out and, when unet_in != NULL, duplicated into unet_in[0:2*numel]
 * (the next step's CFG batch). coef = device float[4] {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev),
 * sqrt(1-a_prev)} so the same graph node serves every step.                                   */
int sfast_hip_cfg_ddim_step(const void *eps_uc, const void *latents, void *latents_out,
                            void *unet_in, const float *coef, float guidance, int64_t numel,
                            int32_t dtype, sfast_stream_t stream);

/* ---- what sfast_hip_conv2d would run for this problem under a forced (variant, split_k) ---------------------
 * out = {tile rows, tile columns, K-splits, K-tiles per split, variant id}: the measured kernel selection (the reference's cuDNN
 * benchmark cache, cudnn_convolution_impl.cc:344-370) asks this before timing a candidate -- a variant the problem cannot take
 * (the LDS-patch pipe needs 3x3 / stride 1 / pad 1 / 64-channel slices / tiles of whole image rows) comes back as another id.   */
int sfast_hip_conv2d_plan(const sfast_conv_params *p, int32_t variant, int32_t split_k, int32_t out[5]);

/* ---- device-side schedule cursor ---------------------------------------------------------------------------
 * idx = *cursor mod n_steps; ts_out[0:ts_cols] = ts_table[idx][:]; coef_out[0:coef_cols] = coef_table[idx][:]; *cursor = idx + 1
 * (mod n_steps). Recorded as the first node of the step's hipGraph it replaces the per-step host-issued copies of the timestep
 * and the scheduler coefficients (the reference's pipeline loop feeds `t` from the host every iteration,
 * examples/optimize_stable_diffusion_pipeline.py:127-151): one denoise iteration = one graph launch. All float tables.          */
int sfast_hip_schedule_advance(int32_t *cursor, const float *ts_table, int32_t ts_cols, float *ts_out,
                               const float *coef_table, int32_t coef_cols, float *coef_out, int32_t n_steps,
                               sfast_stream_t stream);

/* ---- scheduler update in linear form: out = A*sample + B*model_output, fp32 math ---------------------------
 * (A, B) = coef[2*idx], coef[2*idx+1] with idx = *index (device int32 / int64 scalar, clamped to [0, index_limit)) or 0 when
 * index == NULL. Every deterministic one-step sampler update is this form -- DDIM with eta = 0 for epsilon or v prediction
 * being the case `compile(..., trace_scheduler=True)` routes here (reference: compilers/diffusion_pipeline_compiler.py:103-107
 * traces scheduler.step into one TorchScript graph); the coefficient table over all training timesteps is built on the host
 * once per set_timesteps(), the timestep itself stays on the device. */
int sfast_hip_linear_step(const void *model_output, const void *sample, void *out, const float *coef, const void *index,
                          int32_t index_is_i64, int64_t index_limit, int64_t numel, int32_t dtype, sfast_stream_t stream);

/* ---- row-wise mix: out[r][:] = wx*x[r][:] + wy*y[r][:] + vec[(r / vec_rows) % vec_mod][:]  (fp32 math) -------------------
 * The elementwise glue of the spatio-temporal (Stable Video Diffusion) UNet the reference compiles in
 * examples/optimize_stable_video_diffusion_pipeline.py: diffusers' AlphaBlender (x = a*x_spatial + (1-a)*x_temporal with
 * a = sigmoid(mix_factor), `mix_factor` != NULL: read from the live parameter on the device, 1-a when switch_spatial_to_temporal),
 * the frame-position embedding added to every token of a frame (vec_rows = tokens per frame, vec_mod = frames) and the
 * single-key cross-attention result added to every token of a sample (vec_rows = tokens per sample). y, vec may be NULL;
 * out may alias x or y. C % 8 == 0, f16 / bf16. */
typedef struct {
    int32_t dtype;
    int64_t M;          /* rows */
    int32_t C;          /* row length, dense */
    int32_t vec_rows, vec_mod;
    int64_t ld_vec;
    float wx, wy;       /* used when mix_factor == NULL */
    int32_t switch_spatial_to_temporal;
} sfast_mix_params;

int sfast_hip_mix_rows(const void *x, const void *y, const void *vec, const void *mix_factor, void *out,
                       const sfast_mix_params *p, sfast_stream_t stream);

/* ---- packed weights for the pipe-4 GEMM / conv kernels (ABI 9) ------------------------------------------------------------
 * packed[(nb * KS + s) * 64 + lane] (16 bytes each) = the 8 elements w[nb * 32 + lane % 32][s * 16 + (lane / 32) * 8 .. + 8], zero
 * outside [N) x [K); KS = ceil(K / 64) * 4 -- i.e. one contiguous 1 KB block per (32-row block, 16-wide k-step) in the A-operand
 * layout of v_mfma_f32_32x32x16. w: [N][K] with rows ldw elements apart (a conv weight [Cout][KH][KW][Cin]: N = Cout,
 * K = KH * KW * Cin). sfast_hip_packed_weight_bytes gives the size of `packed` (16-byte aligned). Measured motive:
 * profiles/r04_wdirect_probe_run15.log (row-major fragment loads 44-47 % of the MFMA peak, packed 78 %). */
size_t sfast_hip_packed_weight_bytes(int32_t N, int32_t K);
int sfast_hip_pack_weight(const void *w, void *packed, int32_t N, int32_t K, int64_t ldw, int32_t dtype, sfast_stream_t stream);

/* ---- un-fused LoRA: effective weights rebuilt from the live parameters, one launch for all LoRA'd linears (ABI 9) ------------
 * out_i[n][k] = w_i[n][k] + scales[scale_index_i] * sum_j up_i[n][j] * down_i[j][k]      (fp32 math, one rounding to the dtype)
 * for every entry i of a DEVICE-resident table. Stands for what the reference does with a UNet whose LoRA layers are loaded but not
 * fused: it traces linear(x, W) + scale * up(down(x)) as it is (diffusers LoRACompatibleLinear / peft lora.Linear forward, through
 * sfast::cublas_lowp_linear, src/sfast/csrc/operators/cublas/cublas_gemm.cpp:798-948) and lets the user switch adapters by copying
 * into the same tensors in place (README.md:228-265 "Dynamically Switch LoRA", tests/compilers/
 * test_stable_diffusion_pipeline_compiler.py:327-328,438-465). The plan's GEMMs read `out`; base / down / up are read at EVERY
 * launch, so the in-place switch needs no re-capture. `scales` is a device float array (diffusers' cross_attention_kwargs["scale"]
 * times network_alpha / rank, or peft's scaling[adapter]; the caller rewrites it when a value changes), NULL = 1.0 everywhere.
 * Geometry: w rows ldw apart, down [r][K] rows ldd apart, up [N][r] rows ldu apart, out dense [N][K]; K % 8 == 0, ldw % 8 == 0,
 * ldd % 8 == 0, r <= SFAST_LORA_MAX_RANK, w / down / out 16-byte aligned; f16 / bf16.
 * sfast_hip_lora_merge_plan (host, no device work) validates a HOST copy of the table and fills tile_begin / *total_tiles; the
 * caller uploads the filled table and passes both to sfast_hip_lora_merge. */
#define SFAST_LORA_TILE_N 32
#define SFAST_LORA_TILE_K 256
#define SFAST_LORA_MAX_RANK 128
typedef struct {
    const void *w, *down, *up;
    void *out;
    int32_t N, K, r;
    int32_t tile_begin;  /* filled by sfast_hip_lora_merge_plan */
    int64_t ldw, ldd, ldu;
    int32_t scale_index;
    int32_t reserved;
} sfast_lora_entry;

int sfast_hip_lora_merge_plan(sfast_lora_entry *entries_host, int32_t n, int32_t *total_tiles);
int sfast_hip_lora_merge(const sfast_lora_entry *entries_device, int32_t n, int32_t total_tiles, const float *scales_device,
                         int32_t dtype, sfast_stream_t stream);

/* ---- image post-process: NCHW f16/bf16/f32 image -> NHWC uint8 or float32 ----------------------
 * replaces the reference's patched VaeImageProcessor.postprocess / pt_to_pil / pt_to_numpy
 * (libs/diffusers/image_processor.py:23-108: denormalize (x/2+0.5).clamp(0,1), permute(0,2,3,1), and for PIL
 * output mul(255).round().to(uint8) -- all on the GPU so that only the final bytes cross PCIe). */
typedef struct {
    int32_t dtype;       /* input dtype */
    int32_t B, C, H, W;  /* input is dense NCHW */
    int32_t denormalize; /* 1: (x/2 + 0.5).clamp(0, 1) first */
    int32_t to_uint8;    /* 1: out = uint8 round(255*x) ; 0: out = float32 */
} sfast_image_params;

int sfast_hip_image_postprocess(const void *image, void *out, const sfast_image_params *p, sfast_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SFAST_HIP_H */
