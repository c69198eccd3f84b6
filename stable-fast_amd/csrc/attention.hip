// Flash-style scaled-dot-product attention forward for gfx950 (no mask, no dropout).
//
// Replaces the external xformers.ops.memory_efficient_attention the reference binds as
// sfast_xformers::memory_efficient_attention (src/sfast/libs/xformers/xformers_attention.py:26-48)
// with q/k/v kept as [B, S, H, D] strided views (src/sfast/libs/diffusers/xformers_attention.py:37-69),
// so the fused-QKV projection output is consumed in place (no head permute copies).
//
// CDNA4 design:
//   * one workgroup = NW waves = NW*32 query rows of one (batch, head); K/V tiles of 64 keys are
//     staged once in LDS and shared by the waves.
//   * "swapped" products so every reduction stays inside a lane:
//       S^T = K . Q^T   (MFMA A = K rows from LDS, B = Q rows held in registers)
//     leaves lane (q = lane&31) with the scores of its query row for 16 keys per 32-key block
//     (the other half-wave holds the other 16) -> row max / row sum are register loops plus ONE
//     cross-half exchange; the probabilities convert to f16 in place and ARE the B operand of
//       O^T = V^T . P^T (MFMA A = V^T rows from LDS)
//     with the k-slot <-> key permutation chosen to match the C/D register layout, so P never
//     moves between lanes and never touches LDS.
//   * V is transposed while it is written to LDS (pairs of keys packed into one dword, conflict-free
//     ds_write_b32); V^T fragments are two 8-byte reads. K rows are padded to an odd number of
//     16-byte slots -> conflict-free ds_read_b128.
//   * head dims 40 / 80 (SD1.5) are zero-padded to the MFMA K-step inside LDS/registers only.
//   * softmax in fp32 with exp2 and a folded scale*log2(e) (`exp2(fma(s, c, -m*c))`, the running max tracked on raw
//     scores); the O / l rescale is skipped while no lane's running max moves (exact); accumulate fp32.
//   * K / V^T tiles are double-buffered in dynamic LDS and fetched with raw buffer loads (one descriptor per operand,
//     rows past Skv / head-dim padding / idle lanes out of range = 0, ONE v_add per load and tile); the loads of the tiles
//     two (V) and four (K) ahead are issued at the end of an iteration and written to LDS by the next one.
//   * the tile loop is software-pipelined and hand-interleaved (sched_barrier fences): iteration t issues the QK^T MFMAs of
//     tile t+1 between the exp2 chunks of tile t, and the PV MFMAs of tile t between the row-max of tile t+1, the LDS
//     staging and the fragment reads of the tiles after it. Measured on gfx950 (tools/micro/overlap.hip): VALU work only
//     hides behind an MFMA when it follows it in the SAME wave; from a second wave of the SIMD, 3-operand VALU
//     (fma, max3, pk_fma) serialises with MFMAs, and the previous tile loop ran at exactly MFMA + softmax + staging time.
//   * head dims below the 32-row MFMA block (40 -> 64, 80 -> 96) carry a ONES row in the V^T padding, so the softmax
//     denominator is accumulated by the PV MFMA itself (O^T[D][q] = sum_k P[q][k]) instead of 32 VALU adds per tile;
//     f16 probabilities then use packed round-toward-zero converts (the bias cancels in O / l).
//   * built with the VGPR form of the MFMAs (build.py EXTRA_FLAGS): the softmax reads every S accumulator on the
//     VALU each tile, AGPR-allocated results cost ~150 v_accvgpr moves per tile.
#include "attention.h"

namespace sfast {

extern unsigned long long *g_igemm_trace;  // igemm_glds.hip
static int g_attn_xmap = 1;                // SFAST_XCD_MAP=0: (q-block, head, batch) grid as in round 1 (A/B)
static int g_attn_q64 = -1;                // SFAST_ATTN_Q64: 0 = never, 1 = whenever the shape allows, unset = by work-unit count

// TRACE = 1 (profiling instantiation, chosen while a trace buffer is set): wave 0 sums s_memtime deltas of the tile
// phases -- record [top, phase 1, phase 2, barrier wait, whole kernel, tiles] per workgroup (tools/attn_ab.py --trace).
template <typename T, int D, int NW, int TRACE = 0, bool BIAS = false>
__global__ void __launch_bounds__(NW * 64) attn_fwd_kernel(const AttnArgs a) {
    unsigned long long tr_acc[4] = {0, 0, 0, 0}, tr_t0 = 0, tr_last = 0;
    if constexpr (TRACE) tr_t0 = __builtin_amdgcn_s_memtime();
    auto tr_mark = [&](int slot) __attribute__((always_inline)) {
        if constexpr (TRACE) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (slot >= 0) tr_acc[slot] += now - tr_last;
            tr_last = now;
        }
    };
    using vec8 = typename Elem<T>::vec8;
    using G = AttnGeom<D>;
    constexpr int NT = NW * 64;
    constexpr int DP = G::DP, DO = G::DO, KSTR = G::KSTR, VSTR = G::VSTR, STAGE = G::STAGE;
    constexpr int KD = DP / 16;   // MFMA k-steps over d
    constexpr int DB = DO / 32;   // 32-row blocks of O^T
    constexpr int KCH = DP / 8;   // 16-B chunks per K row (incl. zero padding)
    constexpr int VCH = D / 8;    // 16-B chunks per V row
    constexpr int KTASK = (64 * KCH + NT - 1) / NT;
    constexpr int VTASK = (32 * VCH + NT - 1) / NT;
    static_assert(D % 8 == 0, "head dim must be a multiple of 8");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    int b, h, qb;
    if (a.xmap) {
        const int bid = blockIdx.x, xcd = bid & 7, k = bid >> 3;
        int pl = (int)((float)k * __builtin_amdgcn_rcpf((float)a.nqb));  // k / nqb, exact after one correction each way (k < 2^22)
        int r = k - pl * a.nqb;
        pl += (r >= a.nqb) ? 1 : 0;
        pl -= (r < 0) ? 1 : 0;
        r = k - pl * a.nqb;
        // (batch, head) pairs: an XCD owns CONSECUTIVE heads -- in the fused-QKV layout a head's row is D*2 bytes (80 B at D = 40)
        // inside a 128-B line shared with its neighbour head; neighbours on different XCDs would each fetch the whole line
        const int pair = __builtin_amdgcn_readfirstlane(xcd * a.ppx + pl);
        qb = __builtin_amdgcn_readfirstlane(r);
        b = pair / a.H;
        h = pair - b * a.H;
    } else {
        b = blockIdx.z, h = blockIdx.y, qb = blockIdx.x;
    }
    const int q0 = qb * (NW * 32) + wave * 32;
    const int qrow = q0 + l31;

    const T *Qp = (const T *)a.q + (int64_t)b * a.qs[0] + (int64_t)h * a.qs[2];
    const T *Kp = (const T *)a.k + (int64_t)b * a.ks[0] + (int64_t)h * a.ks[2];
    const T *Vp = (const T *)a.v + (int64_t)b * a.vs[0] + (int64_t)h * a.vs[2];

    // V^T rows D..DO-1 pad the head dim up to the MFMA block in both stages (never written by the tile loads).
    // Row D is set to ONES instead of zero when there is padding: O^T[D][q] then accumulates sum_k P[q][k] -- the
    // softmax denominator -- inside the PV MFMA, rescaled with the rest of O for free. That removes 32 VALU adds per
    // 64-key tile from a loop that is VALU-bound for the small head dims (exp2 + max + convert outweigh 14 MFMAs).
    constexpr bool MFMA_ROWSUM = DO > D;
    {
        constexpr int PADW = (DO - D) * (VSTR / 2);  // dwords of padding rows per stage
        const uint32_t one2 = std::is_same<T, f16>::value ? 0x3C003C00u : 0x3F803F80u;
        for (int i = tid; i < 2 * PADW; i += NT) {
            const int st = i / (PADW > 0 ? PADW : 1);
            const int j = i - st * PADW;
            reinterpret_cast<uint32_t *>(smem + st * STAGE + 64 * KSTR * 2 + D * VSTR * 2)[j] = (j < VSTR / 2) ? one2 : 0u;
        }
    }

    // ---- Q fragments (B operand), kept in registers for the whole kernel -------------------------
    typedef const u32x4 __attribute__((address_space(1))) * gvec_ptr;
    auto ldg16 = [&](const T *p, bool ok) -> u32x4 {
        // unconditional load; out-of-range chunks read the device zero block (see igemm.hip)
        const gvec_ptr q = ok ? (gvec_ptr)(const void *)p : (gvec_ptr)(const void *)g_zero16;
        return *q;
    };
    vec8 qf[KD];
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) {
        const int d0 = kd * 16 + hi * 8;
        qf[kd] = __builtin_bit_cast(vec8, ldg16(Qp + (int64_t)qrow * a.qs[1] + d0, d0 < D && qrow < a.Sq));
    }

    u32x4 kreg[KTASK];
    u32x4 vreg[VTASK][2];

    // LDS destinations of this thread's staging tasks, per stage. Idle lanes (the ragged last task) write to a DUMP area
    // behind the stages instead of being masked off: exec-masked stores would split the tile body into basic blocks,
    // and the body must stay ONE block for the MFMA / VALU interleave below.
    constexpr int DUMP = G::LDS;
    int kdst[2][KTASK], vdst[2][VTASK];
#pragma unroll
    for (int i = 0; i < KTASK; ++i) {
        const int id = tid + i * NT;
        const int key = id / KCH, ch = id - key * KCH;
#pragma unroll
        for (int st = 0; st < 2; ++st) kdst[st][i] = id < 64 * KCH ? st * STAGE + key * (KSTR * 2) + ch * 16 : DUMP + tid * 16;
    }
#pragma unroll
    for (int i = 0; i < VTASK; ++i) {
        const int id = tid + i * NT;
        const int kp = id & 31, ch = id >> 5;
#pragma unroll
        for (int st = 0; st < 2; ++st)
            vdst[st][i] = ch < VCH ? st * STAGE + 64 * KSTR * 2 + (ch * 8) * (VSTR * 2) + kp * 4 : DUMP + tid * 4;
    }

    // K / V tiles arrive through raw buffer loads: one descriptor per operand covering the rows of this (batch, head), a
    // per-lane byte offset advanced by ONE v_add per tile (64-bit address arithmetic per load cost ~250 cycles a tile,
    // profiles/r01_attn_phase_trace.log). Rows past Skv, the zero padding of the head dim and idle lanes are out of range
    // of the descriptor and read as 0; idle lanes sit at 2^31, which the advance cannot wrap (spans are < 2^31 bytes).
    auto make_srd = [&](const T *base, uint32_t bytes) __attribute__((always_inline)) {
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base);
        const uint32_t hi32 = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)base >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void *)(((uintptr_t)hi32 << 32) | lo), 0, (int)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t ksrd = make_srd(Kp, a.kspan), vsrd = make_srd(Vp, a.vspan);
    // additive bias: the lane that owns query row qrow reads, per 32-key block, the four 4-key groups its S^T registers hold
    // (keys 8g + 4hi .. +3 <-> registers 4g .. 4g+3). Rows / keys outside the tensor are outside the descriptor and read 0.
    const T *Bp = BIAS ? (const T *)a.bias + (int64_t)b * a.bs[0] + (int64_t)h * a.bs[1] : nullptr;
    const __amdgpu_buffer_rsrc_t bsrd = make_srd(BIAS ? Bp : Kp, BIAS ? a.bspan : 0u);
    const uint32_t brow = BIAS ? (uint32_t)(((int64_t)qrow * a.bs[2] + 4 * hi) * 2) : 0u;
    u32x2 breg[BIAS ? 8 : 1];
    auto load_bias = [&](int kt_) __attribute__((always_inline)) {
        if constexpr (BIAS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // j = kb * 4 + g
                const uint32_t off = brow + (uint32_t)(kt_ * 64 + (j >> 2) * 32 + (j & 3) * 8) * 2u;
                breg[j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(bsrd, qrow < a.Sq ? off : 0x80000000u, 0, 0));
            }
        }
    };
    auto add_bias = [&](f32x16 (&t)[2]) __attribute__((always_inline)) {
        if constexpr (BIAS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f[4];
                unpack4<T>(breg[j], f);
#pragma unroll
                for (int i = 0; i < 4; ++i) t[j >> 2][4 * (j & 3) + i] = fmaf(f[i], a.inv_scale, t[j >> 2][4 * (j & 3) + i]);
            }
        }
    };
    const uint32_t ktile_bytes = (uint32_t)a.ks[1] * 128u, vtile_bytes = (uint32_t)a.vs[1] * 128u;
    uint32_t koff[KTASK], voff[VTASK][2];
#pragma unroll
    for (int i = 0; i < KTASK; ++i) {
        const int id = tid + i * NT;
        const int key = id / KCH, ch = id - key * KCH;
        koff[i] = (id < 64 * KCH && ch * 8 < D) ? ((uint32_t)key * (uint32_t)a.ks[1] + ch * 8) * 2u : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < VTASK; ++i) {
        const int id = tid + i * NT;
        const int kp = id & 31, ch = id >> 5;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            voff[i][j] = ch < VCH ? ((uint32_t)(2 * kp + j) * (uint32_t)a.vs[1] + ch * 8) * 2u : 0x80000000u;
    }
    // each call fetches the NEXT tile (tiles are requested strictly in order)
    auto prefetch_k = [&](u32x4 (&dst)[KTASK]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KTASK; ++i) {
            dst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ksrd, koff[i], 0, 0));
            koff[i] += ktile_bytes;
        }
    };
    auto prefetch_v = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < VTASK; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                vreg[i][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vsrd, voff[i][j], 0, 0));
                voff[i][j] += vtile_bytes;
            }
    };
    auto stage_k_one = [&](int st, int i, const u32x4 (&src)[KTASK]) __attribute__((always_inline)) {
        *reinterpret_cast<u32x4 *>(smem + kdst[st][i]) = src[i];
    };
    // V^T element e of task i: keys (2kp, 2kp+1) of channel ch*8+e packed into one dword -- one v_perm_b32
    auto stage_v_one = [&](int st, int i, int e) __attribute__((always_inline)) {
        const uint32_t w0 = vreg[i][0][e >> 1], w1 = vreg[i][1][e >> 1];
        const uint32_t packed = __builtin_amdgcn_perm(w1, w0, (e & 1) ? 0x07060302u : 0x05040100u);
        *reinterpret_cast<uint32_t *>(smem + vdst[st][i] + e * (VSTR * 2)) = packed;
    };

    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    // running max is tracked on the RAW scores; the softmax scale (> 0) is folded into the exp2 argument
    float m_run = -INFINITY, l_run = 0.f;
    const float c = a.scale_log2e;
    const int ntiles = (a.Skv + 63) / 64;

    // S^T tiles: s[t & 1] holds tile t. The loop is software-pipelined over tiles: iteration t issues the QK^T MFMAs of
    // tile t+1 BETWEEN the exp2 chunks of tile t, and the PV MFMAs of tile t between the row-max of tile t+1 and the LDS
    // staging of the tiles after it. On gfx950 an MFMA only overlaps with VALU work placed behind it in the SAME wave
    // (tools/micro/overlap.hip: 5 v_fma/v_max3/v_cvt per MFMA are free in-wave, while the same instructions issued by a
    // second wave of the SIMD serialise with the MFMA), and the old tile loop measured as the plain SUM of its MFMA,
    // softmax and staging time (profiles/r01_attn_loop_experiments.log) -- so the interleave is spelled out with
    // sched_barriers instead of being left to occupancy.
    f32x16 s[2][2];
    float mloc;  // row max of the tile about to be exponentiated (own half-wave merged with the other)

    auto qk_frag = [&](const char *Ksm, int kb, int kd) __attribute__((always_inline)) -> vec8 {
        return *reinterpret_cast<const vec8 *>(Ksm + (kb * 32 + l31) * (KSTR * 2) + (kd * 16 + hi * 8) * 2);
    };
    auto row_max = [&](const f32x16 (&t)[2]) __attribute__((always_inline)) -> float {
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, t[kb][r]);
        return fmaxf(m, __shfl_xor(m, 32, 64));
    };

    // K runs one tile further ahead than V: iteration t multiplies with the K(t+1) fragments it READ during iteration t-1
    // (kf, registers), reads the K(t+2) fragments from LDS under its PV MFMAs and stages K(t+3) -- so no LDS latency sits
    // between the barrier and the first MFMA. Tile j of K / V lives in stage j & 1.
    vec8 kf[2][KD];
    auto read_kf = [&](const char *Ksm) __attribute__((always_inline)) {
#pragma unroll
        for (int kd = 0; kd < KD; ++kd)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) kf[kb][kd] = qk_frag(Ksm, kb, kd);
    };

    // prologue: all four tile loads in flight at once; K(0), V(0) -> stage 0, K(1) -> stage 1; S(0) = K(0) . Q^T and
    // kf = K(1) fragments; then K(2) -> stage 0 and the loads of K(3), V(1) that iteration 0 stages.
    {
        u32x4 k1[KTASK], k2[KTASK];
        prefetch_k(kreg);
        prefetch_v();
        prefetch_k(k1);
        prefetch_k(k2);
#pragma unroll
        for (int i = 0; i < KTASK; ++i) stage_k_one(0, i, kreg);
#pragma unroll
        for (int i = 0; i < VTASK; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) stage_v_one(0, i, e);
#pragma unroll
        for (int i = 0; i < KTASK; ++i) stage_k_one(1, i, k1);
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[0][kb][r] = 0.f;
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) s[0][kb] = amfma32(qk_frag(smem, kb, kd), qf[kd], s[0][kb]);
        }
        read_kf(smem + STAGE);
        load_bias(0);
        add_bias(s[0]);
        mloc = row_max(s[0]);
        prefetch_k(kreg);  // K(3), V(1): staged by iteration 0
        prefetch_v();
        __syncthreads();  // every wave has read K(0) and K(1)
#pragma unroll
        for (int i = 0; i < KTASK; ++i) stage_k_one(0, i, k2);
        __syncthreads();  // K(2) visible
    }

    constexpr int NQK = 2 * KD;  // QK^T MFMAs per tile
    constexpr int NPV = 4 * DB;  // PV MFMAs per tile
    constexpr int NVE = VTASK * 8;
    using Buf0 = std::integral_constant<int, 0>;
    using Buf1 = std::integral_constant<int, 1>;

    auto tile = [&](int kt, auto CUR) __attribute__((always_inline)) {
        constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
        const char *Kn2 = smem + cur * STAGE;                    // K(kt+2)
        const char *Vcu = smem + cur * STAGE + 64 * KSTR * 2;    // V^T(kt)

        tr_mark(-1);
        if ((kt + 1) * 64 > a.Skv) {  // tail tile (once per kernel): mask keys >= Skv, redo the row max
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= a.Skv) s[cur][kb][r] = -INFINITY;
                }
            mloc = row_max(s[cur]);
        }
        // exact lazy rescale: O and l only need rescaling when some row's running max actually grows
        if (__any(mloc > m_run)) {
            const float m_new = fmaxf(m_run, mloc);
            // (rows whose keys so far are all masked keep m = -inf: -inf - -inf would poison O with NaN)
            const float alpha = (BIAS && m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            if constexpr (!MFMA_ROWSUM) l_run *= alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        // (a row whose keys so far are all masked with -inf has m_run = -inf: exp2(-inf * c - 0) = 0, not NaN)
        const float mc = (BIAS && m_run == -INFINITY) ? 0.f : m_run * c;
        float rowsum = 0.f;
        load_bias(kt + 1);  // bias of the tile whose S^T this iteration computes; added before its row max (phase 2)

        // V^T(kt) fragments are read during phase 1: their LDS latency runs under it instead of stalling the PV MFMAs (with
        // one or two waves per SIMD nothing else hides it -- phase 2 measured 870 cycles for 8 MFMAs before this)
        auto v_frag = [&](int grp, int db) __attribute__((always_inline)) -> u32x4 {
            const int base = (grp >> 1) * 32 + 16 * (grp & 1) + 4 * hi;
            const char *vrow = Vcu + (db * 32 + l31) * (VSTR * 2);
            const u32x2 v0 = *reinterpret_cast<const u32x2 *>(vrow + base * 2);
            const u32x2 v1 = *reinterpret_cast<const u32x2 *>(vrow + (base + 8) * 2);
            return u32x4{v0[0], v0[1], v1[0], v1[1]};
        };
        u32x4 vf[4][DB];
        // ---- phase 1: S^T(kt+1) = K . Q^T  ||  P(kt) = exp2(S(kt) * c - m * c) ---------------------------------------
        __builtin_amdgcn_sched_barrier(0);
        tr_mark(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NQK; ++g) {
            const int kd = g >> 1, kb = g & 1;  // alternate the two accumulators
            if (kd == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[nxt][kb][r] = 0.f;
            }
            s[nxt][kb] = amfma32(kf[kb][kd], qf[kd], s[nxt][kb]);
            __builtin_amdgcn_sched_barrier(0);
            // a slice of the V^T(kt) fragment reads per gap: spread, the four waves of the workgroup do not hit the LDS
            // with 8 KB each right behind the barrier
#pragma unroll
            for (int j = g * NPV / NQK; j < (g + 1) * NPV / NQK; ++j) vf[j / DB][j % DB] = v_frag(j / DB, j % DB);
#pragma unroll
            for (int e = g * 32 / NQK; e < (g + 1) * 32 / NQK; ++e) {
                const float pr = __builtin_amdgcn_exp2f(fmaf(s[cur][e >> 4][e & 15], c, -mc));
                s[cur][e >> 4][e & 15] = pr;
                if constexpr (!MFMA_ROWSUM) rowsum += pr;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!MFMA_ROWSUM) l_run += rowsum;
        tr_mark(1);
        __builtin_amdgcn_sched_barrier(0);

        // ---- phase 2: O^T += V^T . P^T  ||  row max of S(kt+1), staging of K(kt+3) / V(kt+1) ---------------------------
        add_bias(s[nxt]);
        float mpart[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int kb = grp >> 1, s2 = grp & 1;
            // P^T fragment: registers [8*s2, 8*s2+8) of block kb, converted in place
            vec8 pf;
            if constexpr (std::is_same<T, f16>::value && MFMA_ROWSUM) {
                // f16: packed round-toward-zero converts (one instruction per pair instead of three). The denominator comes
                // from the SAME rounded probabilities through the ones-row, so the rounding bias cancels in O / l.
                u32x4 w;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    w[jj] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(s[cur][kb][8 * s2 + 2 * jj], s[cur][kb][8 * s2 + 2 * jj + 1]));
                pf = __builtin_bit_cast(vec8, w);
            } else {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pf[jj] = Elem<T>::from_f32(s[cur][kb][8 * s2 + jj]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const int gap = grp * DB + db;
                o[db] = amfma32(__builtin_bit_cast(vec8, vf[grp][db]), pf, o[db]);
                __builtin_amdgcn_sched_barrier(0);
                // fillers: first half of the gaps take the 16 max3 of S(kt+1), second half the V^T staging, the last the K rows
                if (gap < NPV / 2) {
                    // K(kt+2) fragments for the next iteration's phase 1, a slice per gap
#pragma unroll
                    for (int j = gap * NQK / (NPV / 2); j < (gap + 1) * NQK / (NPV / 2); ++j) kf[j & 1][j >> 1] = qk_frag(Kn2, j & 1, j >> 1);
#pragma unroll
                    for (int j = gap * 16 / (NPV / 2); j < (gap + 1) * 16 / (NPV / 2); ++j)
                        // volatile asm: a plain fmaxf is not ordered against sched_barrier and sinks behind the last MFMA
                        asm volatile("v_max3_f32 %0, %0, %1, %2"
                                     : "+v"(mpart[j & 3])
                                     : "v"(s[nxt][j >> 3][2 * (j & 7)]), "v"(s[nxt][j >> 3][2 * (j & 7) + 1]));
                } else {
                    const int hgap = gap - NPV / 2, nh = NPV - NPV / 2;
#pragma unroll
                    for (int j = hgap * NVE / nh; j < (hgap + 1) * NVE / nh; ++j) stage_v_one(nxt, j >> 3, j & 7);
                    if (gap == NPV - 1) {
#pragma unroll
                        for (int i = 0; i < KTASK; ++i) stage_k_one(nxt, i, kreg);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the staging registers are free again: global loads of K(kt+4), V(kt+2), staged by the NEXT iteration's phase 2 --
        // a full iteration of flight time. Tiles past the end are out of range and read 0 (unconditional: no branch here).
        prefetch_k(kreg);
        prefetch_v();
        const float m4 = fmaxf(fmaxf(mpart[0], mpart[1]), fmaxf(mpart[2], mpart[3]));
        mloc = fmaxf(m4, __shfl_xor(m4, 32, 64));
        tr_mark(2);
        __syncthreads();
        tr_mark(3);
    };

    // whole pairs in the loop, an odd last tile after it (the S buffers and LDS stages are compile-time per half)
    int kt = 0;
    for (; kt + 1 < ntiles; kt += 2) {
        tile(kt, Buf0{});
        tile(kt + 1, Buf1{});
    }
    if (kt < ntiles) tile(kt, Buf0{});

    if constexpr (TRACE) {
        if (a.trace != nullptr && tid == 0) {
            unsigned long long *rec = a.trace + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16;
            for (int i = 0; i < 4; ++i) rec[i] = tr_acc[i];
            rec[4] = __builtin_amdgcn_s_memtime() - tr_t0;
            rec[5] = ntiles;
        }
    }
    // ---- epilogue: normalise and store 4 consecutive d per lane ----------------------------------------
    float l_tot;
    if constexpr (MFMA_ROWSUM) {
        // O^T row D of query l31: block D/32, row D%32 = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        constexpr int RR = D % 32;
        constexpr int LREG = (RR & 3) + 4 * (RR >> 3);
        constexpr int LHI = (RR >> 2) & 1;
        l_tot = __shfl(o[D / 32][LREG], l31 + 32 * LHI, 64);
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv = 1.0f / l_tot;
    if (qrow < a.Sq) {
        T *Op = (T *)a.out + (int64_t)b * a.os[0] + (int64_t)qrow * a.os[1] + (int64_t)h * a.os[2];
#pragma unroll
        for (int db = 0; db < DB; ++db) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * hi;
                if (d < D) {
                    *reinterpret_cast<u32x2 *>(Op + d) =
                        pack4<T>(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
                }
            }
        }
    }
}

// Generic fallback: any head dim / dtype. One wave per (b, h, q); scores staged in LDS (Skv <= 12288).
template <typename T>
__global__ void __launch_bounds__(64) attn_naive_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sc[];
    const int lane = threadIdx.x;
    const int q = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const T *Qp = (const T *)a.q + (int64_t)b * a.qs[0] + (int64_t)q * a.qs[1] + (int64_t)h * a.qs[2];
    const T *Kp = (const T *)a.k + (int64_t)b * a.ks[0] + (int64_t)h * a.ks[2];
    const T *Vp = (const T *)a.v + (int64_t)b * a.vs[0] + (int64_t)h * a.vs[2];
    float mx = -INFINITY;
    for (int k = lane; k < a.Skv; k += 64) {
        const T *kr = Kp + (int64_t)k * a.ks[1];
        float acc = 0.f;
        for (int d = 0; d < a.D; ++d) acc = fmaf(Elem<T>::to_f32(Qp[d]), Elem<T>::to_f32(kr[d]), acc);
        acc *= a.scale;
        if (a.bias) acc += Elem<T>::to_f32(((const T *)a.bias)[(int64_t)b * a.bs[0] + (int64_t)h * a.bs[1] + (int64_t)q * a.bs[2] + k]);
        sc[k] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    if (mx == -INFINITY) mx = 0.f;  // fully masked row: all probabilities 0 (0/0 below -> NaN, as the reference's kernels give)
    for (int k = lane; k < a.Skv; k += 64) {
        const float p = __expf(sc[k] - mx);
        sc[k] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.0f / sum;
    T *Op = (T *)a.out + (int64_t)b * a.os[0] + (int64_t)q * a.os[1] + (int64_t)h * a.os[2];
    for (int d = lane; d < a.D; d += 64) {
        float acc = 0.f;
        for (int k = 0; k < a.Skv; ++k) acc = fmaf(sc[k], Elem<T>::to_f32(Vp[(int64_t)k * a.vs[1] + d]), acc);
        Op[d] = Elem<T>::from_f32(acc * inv);
    }
}

template <typename T, int D, int NW> static int attn_set_attr() {
    auto kern = attn_fwd_kernel<T, D, NW>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       AttnGeom<D>::LDS_TOTAL);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(attn D=%d): %s", D, hipGetErrorString(e));
        return SFAST_ERR_LAUNCH;
    }
    return 0;
}

template <typename T, int D> static int attn_set_attr_bias() {
    auto kern = attn_fwd_kernel<T, D, 4, 0, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       AttnGeom<D>::LDS_TOTAL);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(attn bias D=%d): %s", D, hipGetErrorString(e));
        return SFAST_ERR_LAUNCH;
    }
    return 0;
}

template <typename T> static int attn_init_t() {
    int rc = 0;
#define A_INIT(D)                                             \
    if constexpr (D < 128) {                                  \
        if (!rc) rc = attn_set_attr<T, D, 2>();               \
    }                                                         \
    if (!rc) rc = attn_set_attr<T, D, 4>();                   \
    if (!rc) rc = attn_set_attr_bias<T, D>();
    A_INIT(40) A_INIT(64) A_INIT(80) A_INIT(128) A_INIT(160)
#undef A_INIT
    return rc;
}

int attention_init() {
    const char *xm = getenv("SFAST_XCD_MAP");
    g_attn_xmap = (xm && xm[0] == '0') ? 0 : 1;
    const char *q6 = getenv("SFAST_ATTN_Q64");
    g_attn_q64 = q6 ? (q6[0] == '0' ? 0 : 1) : -1;
    int rc = attn_init_t<f16>();
    if (!rc) rc = attn_init_t<bf16>();
    if (!rc) rc = attention_q64_init();
    return rc;
}

static dim3 attn_grid(AttnArgs &a, int rows_per_block) {
    a.nqb = ceil_div(a.Sq, rows_per_block);
    a.xmap = (g_attn_xmap && (a.B * a.H) % 8 == 0 && (int64_t)a.nqb * a.B * a.H < (1 << 22)) ? 1 : 0;
    a.ppx = a.B * a.H / 8;
    return a.xmap ? dim3((unsigned)(a.nqb * a.B * a.H), 1, 1) : dim3((unsigned)a.nqb, (unsigned)a.H, (unsigned)a.B);
}

template <typename T, int D>
static int attn_launch_d(const AttnArgs &a_in, int nw, hipStream_t st) {
    AttnArgs a = a_in;
    if (a.bias) {  // biased instantiation: four waves per workgroup
        const dim3 gridb = attn_grid(a, 128);
        hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 0, true>), gridb, dim3(256), AttnGeom<D>::LDS_TOTAL, st, a);
        return check_launch("attention(bias)");
    }
    const dim3 grid = attn_grid(a, nw * 32);
    if constexpr (std::is_same<T, f16>::value && (D == 40 || D == 64)) {
        if (a.trace != nullptr && nw == 4) {
            hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4, 1>), grid, dim3(256), AttnGeom<D>::LDS_TOTAL, st, a);
            return check_launch("attention(trace)");
        }
    }
    if constexpr (D < 128) {  // the 2-wave form of the wide heads would sit at the 512-register limit (and spill): never launched
        if (nw == 2) {
            hipLaunchKernelGGL((attn_fwd_kernel<T, D, 2>), grid, dim3(128), AttnGeom<D>::LDS_TOTAL, st, a);
            return check_launch("attention");
        }
    }
    hipLaunchKernelGGL((attn_fwd_kernel<T, D, 4>), grid, dim3(256), AttnGeom<D>::LDS_TOTAL, st, a);
    return check_launch("attention");
}

template <typename T>
static int attn_launch(const AttnArgs &a, int nw, hipStream_t st) {
    switch (a.D) {
    case 40: return attn_launch_d<T, 40>(a, nw, st);
    case 64: return attn_launch_d<T, 64>(a, nw, st);
    case 80: return attn_launch_d<T, 80>(a, nw, st);
    case 128: return attn_launch_d<T, 128>(a, 4, st);
    case 160: return attn_launch_d<T, 160>(a, 4, st);
    }
    set_error("attention: head dim %d has no MFMA instantiation", a.D);
    return SFAST_ERR_UNSUPPORTED;
}

}  // namespace sfast

using namespace sfast;

extern "C" int sfast_hip_attention(const void *q, const void *k, const void *v, void *out, const sfast_attn_params *p,
                                   sfast_stream_t stream) {
    return sfast_hip_attention_bias(q, k, v, nullptr, nullptr, out, p, stream);
}

extern "C" int sfast_hip_attention_bias(const void *q, const void *k, const void *v, const void *bias, const int64_t *bias_strides,
                                        void *out, const sfast_attn_params *p, sfast_stream_t stream) {
    SFAST_REQUIRE(p && q && k && v && out, SFAST_ERR_INVALID, "attention: null argument");
    SFAST_REQUIRE(!bias || bias_strides, SFAST_ERR_INVALID, "attention: a bias needs its (b, h, q) strides");
    SFAST_REQUIRE(!bias || (bias_strides[0] >= 0 && bias_strides[1] >= 0 && bias_strides[2] >= 0), SFAST_ERR_INVALID,
                  "attention: negative bias strides");
    SFAST_REQUIRE(p->B > 0 && p->H > 0 && p->Sq > 0 && p->Skv > 0 && p->D > 0, SFAST_ERR_INVALID, "attention: bad shape");
#ifndef SFAST_PROBES
    SFAST_REQUIRE(p->variant < 1000, SFAST_ERR_UNSUPPORTED,
                  "attention: variant %d selects a timing-only ablation (garbage results); this library was built without -DSFAST_PROBES", p->variant);
#endif
    hipStream_t st = (hipStream_t)stream;
    AttnArgs a{};
    a.trace = g_igemm_trace;
    a.q = q;
    a.k = k;
    a.v = v;
    a.out = out;
    a.B = p->B;
    a.H = p->H;
    a.Sq = p->Sq;
    a.Skv = p->Skv;
    a.D = p->D;
    for (int i = 0; i < 3; ++i) {
        a.qs[i] = p->qs[i];
        a.ks[i] = p->ks[i];
        a.vs[i] = p->vs[i];
        a.os[i] = p->os[i];
    }
    a.scale = p->scale;
    a.scale_log2e = p->scale * 1.44269504088896340736f;
    a.bias = bias;
    a.inv_scale = p->scale != 0.f ? 1.0f / p->scale : 0.f;
    int64_t bspan = 0;
    if (bias) {
        for (int i = 0; i < 3; ++i) a.bs[i] = bias_strides[i];
        // the kernel reads whole 4-key groups: the last group of a row may reach up to 3 elements past key Skv-1 (caller's contract:
        // every bias row is readable up to the next multiple of 4 keys -- rows padded to 8 elements, as xformers requires, satisfy it)
        bspan = ((int64_t)(p->Sq - 1) * a.bs[2] + (p->Skv + 3) / 4 * 4) * 2;
    }
    const bool half = p->dtype == SFAST_F16 || p->dtype == SFAST_BF16;
    const bool d_ok = p->D == 40 || p->D == 64 || p->D == 80 || p->D == 128 || p->D == 160;
    bool vec = half && d_ok && aligned16(q) && aligned16(k) && aligned16(v) && aligned8(out);
    for (int i = 0; i < 3; ++i)
        vec = vec && p->qs[i] % 8 == 0 && p->ks[i] % 8 == 0 && p->vs[i] % 8 == 0 && p->os[i] % 4 == 0;
    // buffer-descriptor ranges of the MFMA kernel: the K / V rows of one (batch, head), < 2^31 bytes each
    const int64_t kspan = p->ks[1] > 0 ? ((int64_t)(p->Skv - 1) * p->ks[1] + p->D) * 2 : -1;
    const int64_t vspan = p->vs[1] > 0 ? ((int64_t)(p->Skv - 1) * p->vs[1] + p->D) * 2 : -1;
    vec = vec && kspan > 0 && vspan > 0 && kspan < (1ll << 31) && vspan < (1ll << 31);
    vec = vec && p->ks[1] < (1 << 24) && p->vs[1] < (1 << 24);  // 64-row tile advance and row offsets stay in 32 bits
    a.kspan = (uint32_t)kspan;
    a.vspan = (uint32_t)vspan;
    if (bias) {  // MFMA path: dword-aligned 4-key groups, descriptor range below 2^31 bytes
        vec = vec && half && bspan < (1ll << 31) && a.bs[2] % 2 == 0 && a.bs[0] % 2 == 0 && a.bs[1] % 2 == 0 && (((uintptr_t)bias) & 3) == 0;
        a.bspan = (uint32_t)bspan;
    }
    if (vec && p->variant != 100 && p->scale > 0.f && !bias && (p->D == 40 || p->D == 64 || p->D == 80) && p->Skv >= 128) {
        // second-generation kernel (attention_q64.hip): 64 query rows per wave, one wave per SIMD. A work unit is one wave's 64 rows and the
        // chip runs 1024 of them at a time (one per SIMD), every unit taking the same time T whatever the grid:
        //   * up to 1024 units the launch takes T; the 32-row kernel does half the work per wave, twice the waves -- it wins below ~half a
        //     chip of units (measured, profiles/r03_attn_q64_ab_*.log: D=40 at 256 units 15.9 vs 19.6 us, D=64 at 640 units 25.6 vs 21.8 us);
        //   * beyond 1024 units the launch takes ceil(units / 1024) rounds: a last round that is less than half full costs a whole T
        //     (SDXL's 10-head 64x64 level: 1280 units = 2 rounds, 137 vs 132 us) unless there are many rounds to spread it over.
        // variant 64 / 62: forced, 4 / 2 waves per workgroup; variant 32 (and 2 / 4): the 32-row kernel.
        const int64_t units = (int64_t)ceil_div(p->Sq, 64) * p->H * (g_batch_ref > 0 ? g_batch_ref : p->B);  // (SFAST_BATCH_INVARIANT: common.h)
        const int64_t tail = units % 1024;
        const bool fills = units <= 1024 ? units >= (p->D == 80 ? 1024 : 512) : (units >= 3072 || tail == 0 || tail >= 512);
        int use = (p->variant == 64 || p->variant == 62) ? 1 : (p->variant != 0 ? 0 : (g_attn_q64 >= 0 ? g_attn_q64 : (fills ? 1 : 0)));
        if (p->variant >= 1000 && p->variant < 2024) {  // timing-only ablations of the 64-row kernel (tools/attn_ablate.py; probe build only)
            set_kernel_name("attn_q64_ablation[%d]", p->variant - 1000);
            const int rc = attention_q64_launch(a, p->dtype, 4 | (g_attn_xmap << 8) | ((p->variant - 1000) << 16), st);
            if (rc != -1) return rc;
            SFAST_REQUIRE(false, SFAST_ERR_UNSUPPORTED, "attention: no ablation instantiation %d", p->variant - 1000);
        }
        if (use) {
            const int nw = p->variant == 62 ? 2 : (p->variant == 64 ? 4 : (p->Sq % 256 == 0 || p->Sq > 1024 ? 4 : 2));
            set_kernel_name("attn_q64[D=%d,BQ=%d]", p->D, nw * 64);
            const int rc = attention_q64_launch(a, p->dtype, nw | (g_attn_xmap << 8), st);
            if (rc != -1) return rc;
        }
    }
    if (vec && p->variant != 100 && p->scale > 0.f) {
        int nw = 4;
        const int64_t blocks4 = (int64_t)ceil_div(p->Sq, 128) * p->H * (g_batch_ref > 0 ? g_batch_ref : p->B);
        if (blocks4 < 256) nw = 2;
        if (p->variant == 2 || p->variant == 4) nw = p->variant;  // (variant 32: the automatic choice of this kernel)
        if (bias || p->D >= 128) nw = 4;  // wide heads and the biased instantiation: four waves per workgroup only
        set_kernel_name("attn_fwd[D=%d,BQ=%d]%s", p->D, nw * 32, bias ? "+bias" : "");
        if (p->dtype == SFAST_F16) return attn_launch<f16>(a, nw, st);
        return attn_launch<bf16>(a, nw, st);
    }
    SFAST_REQUIRE(p->Skv <= 12288, SFAST_ERR_UNSUPPORTED, "attention: generic path supports Skv <= 12288 (got %d)", p->Skv);
    set_kernel_name("attn_naive");
    const dim3 grid(p->Sq, p->H, p->B);
    const size_t smem = (size_t)p->Skv * sizeof(float);
    switch (p->dtype) {
    case SFAST_F16: hipLaunchKernelGGL(attn_naive_kernel<f16>, grid, dim3(64), smem, st, a); break;
    case SFAST_BF16: hipLaunchKernelGGL(attn_naive_kernel<bf16>, grid, dim3(64), smem, st, a); break;
    case SFAST_F32: hipLaunchKernelGGL(attn_naive_kernel<float>, grid, dim3(64), smem, st, a); break;
    default: set_error("attention: bad dtype %d", p->dtype); return SFAST_ERR_UNSUPPORTED;
    }
    return check_launch("attn_naive");
}
