"""CPU emulator of the libsfast_hip.so C ABI -- TEST INFRASTRUCTURE ONLY.

Implements every compute entry point of include/sfast_hip.h on raw HOST pointers with the fp32
oracle (oracle/ops_ref.py). It lets the CPU test-suite execute a `UNet2DEngine` plan end to end --
pointer arithmetic, strides, buffer reuse, time-embedding offsets, virtual concat / upsample
bookkeeping -- and compare it with the oracle UNet, without a GPU. The product never imports this
module: on a GPU box the engine always calls the real library and raises if it is missing.

The workspace-size queries are forwarded to the REAL library (pure host code), so the kernel
planner (tile / split-K selection) is exercised too.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-fast_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import ops_ref as R  # noqa: E402
from oracle.unet_ref import timestep_embedding as temb_ref  # noqa: E402
from sfast.hip import lib as L  # noqa: E402

_NP = {L.F16: np.float16, L.F32: np.float32, "i32": np.int32}
_ACT = {0: None, 1: "relu", 2: "gelu", 3: "gelu_tanh", 4: "silu", 5: "sigmoid", 6: "tanh"}


def _flat(ptr, numel, dtype):
    npd = _NP[dtype]
    nbytes = int(numel) * np.dtype(npd).itemsize
    buf = (C.c_char * nbytes).from_address(int(ptr))
    return torch.from_numpy(np.frombuffer(buf, dtype=npd))


def _strided(ptr, shape, strides, dtype):
    shape = [int(s) for s in shape]
    strides = [int(s) for s in strides]
    extent = 1 + sum((s - 1) * st for s, st in zip(shape, strides))
    return torch.as_strided(_flat(ptr, extent, dtype), shape, strides)


def _p(ref):
    return ref._obj


class EmuLib:
    """Duck-typed stand-in for the ctypes library handle."""

    def __init__(self, real=None):
        self.real = real if real is not None else L.load()   # host-only queries (plans, layouts, coverage) go to a real library
        self.calls = []
        self.err = b""

    def sfast_hip_workspace_init(self, ws, nbytes, stream):
        self.calls.append(("workspace_init", int(nbytes)))
        return 0 if ws and nbytes >= L.WS_TICKET_BYTES else L.ERR_WORKSPACE if hasattr(L, "ERR_WORKSPACE") else -3

    # ---- forwarded host-only queries ---------------------------------------------------------
    def sfast_hip_group_norm_workspace_bytes(self, ref):
        return self.real.sfast_hip_group_norm_workspace_bytes(ref)

    def sfast_hip_gemm_workspace_bytes(self, ref):
        return self.real.sfast_hip_gemm_workspace_bytes(ref)

    def sfast_hip_conv2d_workspace_bytes(self, ref):
        return self.real.sfast_hip_conv2d_workspace_bytes(ref)

    def sfast_hip_gemm_stats_layout(self, p, ext, out):
        return self.real.sfast_hip_gemm_stats_layout(p, ext, out)

    def sfast_hip_conv2d_stats_layout(self, p, ext, out):
        return self.real.sfast_hip_conv2d_stats_layout(p, ext, out)

    def sfast_hip_conv2d_plan(self, ref, variant, split_k, out):
        return self.real.sfast_hip_conv2d_plan(ref, variant, split_k, out)

    def sfast_hip_igemm_plan(self, M, N, K, geglu, variant, split_k, out):
        return self.real.sfast_hip_igemm_plan(M, N, K, geglu, variant, split_k, out)

    def sfast_hip_gn_conv2d_supported(self, ref):
        return self.real.sfast_hip_gn_conv2d_supported(ref)

    def sfast_hip_gn_conv2d_workspace_bytes(self, ref):
        return self.real.sfast_hip_gn_conv2d_workspace_bytes(ref)

    def sfast_hip_last_error(self):
        return self.err

    def sfast_hip_init(self):
        return 0

    # ---- compute entry points ------------------------------------------------------------------
    def sfast_hip_group_norm(self, x, x2, gamma, beta, y, ref, ws, ws_bytes, stream):
        p = _p(ref)
        self.calls.append("group_norm")
        assert ws_bytes >= self.real.sfast_hip_group_norm_workspace_bytes(ref)
        C1, C2 = p.C1, p.C - p.C1
        g = _flat(gamma, p.C, p.dtype).float() if gamma else None
        b = _flat(beta, p.C, p.dtype).float() if beta else None
        if p.layout == L.NHWC:
            xa = _flat(x, p.N * p.HW * C1, p.dtype).reshape(p.N, p.HW, C1)
            if C2:
                xb = _flat(x2, p.N * p.HW * C2, p.dtype).reshape(p.N, p.HW, C2)
                xa = torch.cat([xa, xb], dim=2)
            xin = xa.permute(0, 2, 1)  # [N, C, HW]
            out = R.group_norm_ref(xin, p.G, g, b, p.eps, p.act == L.ACT_SILU)
            _flat(y, p.N * p.HW * p.C, p.dtype).reshape(p.N, p.HW, p.C).copy_(out.permute(0, 2, 1))
        else:
            xin = _flat(x, p.N * p.C * p.HW, p.dtype).reshape(p.N, p.C, p.HW)
            out = R.group_norm_ref(xin, p.G, g, b, p.eps, p.act == L.ACT_SILU)
            _flat(y, p.N * p.C * p.HW, p.dtype).reshape(p.N, p.C, p.HW).copy_(out)
        return 0

    def sfast_hip_layer_norm(self, x, gamma, beta, y, ref, stream):
        p = _p(ref)
        self.calls.append("layer_norm")
        xin = _flat(x, p.M * p.N, p.dtype).reshape(p.M, p.N)
        g = _flat(gamma, p.N, p.dtype) if gamma else None
        b = _flat(beta, p.N, p.dtype) if beta else None
        _flat(y, p.M * p.N, p.dtype).reshape(p.M, p.N).copy_(R.layer_norm_ref(xin, (p.N,), g, b, p.eps))
        return 0

    def sfast_hip_add_strided(self, src, dst, ref, stream):
        p = _p(ref)
        self.calls.append("add_strided")
        shape = [p.shape[i] for i in range(p.ndim)]
        a = _strided(src, shape, [p.src_strides[i] for i in range(p.ndim)], p.dtype)
        d = _strided(dst, shape, [p.dst_strides[i] for i in range(p.ndim)], p.dtype)
        d.copy_((d.float() + a.float()).to(d.dtype))
        return 0

    def sfast_hip_softmax_rows(self, x, y, ref, stream):
        p = _p(ref)
        self.calls.append("softmax_rows")
        xin = _flat(x, (p.M - 1) * p.ldx + p.N, p.dtype).as_strided((p.M, p.N), (p.ldx, 1))
        out = torch.softmax(xin.float() * p.scale, dim=-1).to(xin.dtype)
        _flat(y, (p.M - 1) * p.ldy + p.N, p.dtype).as_strided((p.M, p.N), (p.ldy, 1)).copy_(out)
        return 0

    def sfast_hip_gemm(self, x, segs, bias, rowbias, res, out, ref, ws, ws_bytes, stream):
        p = _p(ref)
        self.calls.append("gemm")
        assert ws_bytes >= self.real.sfast_hip_gemm_workspace_bytes(ref)
        wrows = 2 * p.N if p.geglu else p.N
        assert p.rows_per_seg * p.n_wseg >= wrows
        xin = _strided(x, (p.M, p.K), (p.ldx, 1), p.dtype)
        ws_ = [_strided(segs[i], (p.rows_per_seg, p.K), (p.ldw, 1), p.dtype) for i in range(p.n_wseg)]
        w = torch.cat(ws_, dim=0)[:wrows]
        b = _flat(bias, wrows, p.dtype) if bias else None
        r = _strided(res, (p.M, p.N), (p.ldr, 1), p.dtype).clone() if res else None
        rb = None
        if rowbias:
            nb = (p.M + p.rows_per_batch - 1) // p.rows_per_batch
            rb = _strided(rowbias, (nb, p.N), (p.ld_rowbias, 1), p.dtype)
        o = R.linear_ref(xin, w, b, _ACT[p.act], r, p.alpha, bool(p.res_before_act), bool(p.geglu), rb, p.rows_per_batch,
                         _ACT[p.in_act])
        _strided(out, (p.M, p.N), (p.ldo, 1), p.dtype).copy_(o)
        return 0

    # ---- epilogue extensions: out_scale + GroupNorm partial statistics in the layout the real library reports ------------
    def _emit_stats(self, out2d, lay, stats_ptr):
        """out2d: [M, N] view of the op's (rounded) output. Writes one {mean, M2} float2 per (row block, tile_n, slot)."""
        M, N = out2d.shape
        R, bno, S, tn_n, u = lay.rb_rows, lay.bno, lay.slots, lay.tiles_n, lay.unit
        assert R * lay.n_rb == M
        rec = _flat(stats_ptr, lay.n_rb * tn_n * S * 2, L.F32).reshape(lay.n_rb, tn_n, S, 2)
        rec.fill_(float("nan"))  # slots a tile does not overlap are never read by a correct consumer
        o = out2d.float()
        for rb in range(lay.n_rb):
            blk = o[rb * R:(rb + 1) * R]
            for tn in range(tn_n):
                for j in range(S):
                    U = (tn * bno) // u + j
                    lo, hi = max(tn * bno, U * u), min(min((tn + 1) * bno, N), (U + 1) * u)
                    if hi > lo:
                        v = blk[:, lo:hi]
                        m = v.mean()
                        rec[rb, tn, j, 0] = m
                        rec[rb, tn, j, 1] = ((v - m) ** 2).sum()

    @staticmethod
    def _splits(query, M, N, K, geglu, variant, split_k):
        o5 = (C.c_int32 * 5)()
        assert query(M, N, K, int(geglu), variant if variant < 100 else 0, split_k, o5) == 0
        return int(o5[2])

    def _emit_fused_gn(self, out2d, ext, M, N, dtype, splits):
        """sfast_epilogue_ext.gn_out: GroupNorm(+SiLU) of the STORED output, written dense [M][N]; only split-K plans have the reduce
        launch it rides in (the real library refuses the others -- so does the emulator)."""
        assert splits > 1, "fused GroupNorm epilogue on a plan without a split-K reduce launch"
        assert ext.gn_unit == 0 and ext.gn_rows_per_sample > 0 and M % ext.gn_rows_per_sample == 0
        HW = ext.gn_rows_per_sample
        Bn = M // HW
        assert (N // ext.gn_groups) % 4 == 0 and HW * (N // ext.gn_groups) <= 16384
        g = _flat(ext.gn_gamma, N, dtype).float() if ext.gn_gamma else None
        b = _flat(ext.gn_beta, N, dtype).float() if ext.gn_beta else None
        xin = out2d.reshape(Bn, HW, N).permute(0, 2, 1)   # [B, C, HW] of the rounded output
        y = R.group_norm_ref(xin, ext.gn_groups, g, b, ext.gn_eps, ext.gn_act == L.ACT_SILU)
        _flat(ext.gn_out, M * N, dtype).reshape(Bn, HW, N).copy_(y.permute(0, 2, 1))
        self.calls.append("fused_gn")

    def sfast_hip_gemm_ex(self, x, segs, bias, rowbias, res, out, ref, ext_ref, stats, ws, ws_bytes, stream):
        ext = _p(ext_ref) if ext_ref is not None else None
        p0 = _p(ref)
        if ext is not None and ext.out_scale not in (0.0, 1.0):
            # scale folded into the activation operand in fp32 (only the bias-free, residual-free form is used that way)
            assert not bias and not res and not rowbias and not p0.geglu
            xin = _strided(x, (p0.M, p0.K), (p0.ldx, 1), p0.dtype)
            w = torch.cat([_strided(segs[i], (p0.rows_per_seg, p0.K), (p0.ldw, 1), p0.dtype) for i in range(p0.n_wseg)], dim=0)[:p0.N]
            o = R.act_ref((xin.float() @ w.float().t()) * ext.out_scale, _ACT[p0.act])
            _strided(out, (p0.M, p0.N), (p0.ldo, 1), p0.dtype).copy_(o)
            self.calls.append("gemm")
            return 0
        rc = self.sfast_hip_gemm(x, segs, bias, rowbias, res, out, ref, ws, ws_bytes, stream)
        if ext is not None and ext.gn_out:
            self._emit_fused_gn(_strided(out, (p0.M, p0.N), (p0.ldo, 1), p0.dtype), ext, p0.M, p0.N, p0.dtype,
                                self._splits(self.real.sfast_hip_igemm_plan, p0.M, p0.N, p0.K, p0.geglu, p0.variant, p0.split_k))
        if stats:
            p = _p(ref)
            lay = L.GnStatsLayout()
            assert self.real.sfast_hip_gemm_stats_layout(ref, ext_ref, C.byref(lay)) == 0
            self._emit_stats(_strided(out, (p.M, p.N), (p.ldo, 1), p.dtype), lay, stats)
            self.calls.append("gn_stats")
        return rc

    def sfast_hip_conv2d_ex(self, x, x2, w, bias, rowbias, z, out, ref, ext_ref, stats, ws, ws_bytes, stream):
        ext = _p(ext_ref) if ext_ref is not None else None
        assert ext is None or ext.out_scale in (0.0, 1.0)
        rc = self.sfast_hip_conv2d(x, x2, w, bias, rowbias, z, out, ref, ws, ws_bytes, stream)
        if ext is not None and ext.gn_out:
            p = _p(ref)
            M = p.B * ext.gn_rows_per_sample
            o5 = (C.c_int32 * 5)()
            assert self.real.sfast_hip_conv2d_plan(ref, p.variant, p.split_k, o5) == 0
            self._emit_fused_gn(_strided(out, (M, p.Cout), (p.os[2], 1), p.dtype), ext, M, p.Cout, p.dtype, int(o5[2]))
        if stats:
            p = _p(ref)
            lay = L.GnStatsLayout()
            assert self.real.sfast_hip_conv2d_stats_layout(ref, ext_ref, C.byref(lay)) == 0
            M = lay.rb_rows * lay.n_rb
            self._emit_stats(_strided(out, (M, p.Cout), (p.os[2], 1), p.dtype), lay, stats)
            self.calls.append("gn_stats")
        return rc

    def sfast_hip_group_norm_apply(self, x, x2, gamma, beta, y, ref, s1, l1_ref, s2, l2_ref, stream):
        """Uses ONLY the handed-over records for the statistics (a planner that wires a stale or foreign buffer shows up as a
        wrong result), merged with the pairwise update like the kernel."""
        p = _p(ref)
        self.calls.append("group_norm_apply")
        C1, C2 = p.C1, p.C - p.C1
        cpg = p.C // p.G
        xa = _flat(x, p.N * p.HW * C1, p.dtype).reshape(p.N, p.HW, C1).float()
        if C2:
            xa = torch.cat([xa, _flat(x2, p.N * p.HW * C2, p.dtype).reshape(p.N, p.HW, C2).float()], dim=2)
        srcs = [(s1, _p(l1_ref), 0, C1)]
        if C2:
            srcs.append((s2, _p(l2_ref), C1, C2))
        cnt = torch.zeros(p.N, p.G, dtype=torch.float64)
        acc1 = torch.zeros(p.N, p.G, dtype=torch.float64)
        recs = []
        for sp, lay, coff, nch in srcs:
            assert cpg % lay.unit == 0 and coff % lay.unit == 0 and lay.rb_rows * lay.n_rb == p.N * p.HW
            per = lay.n_rb // p.N
            rec = _flat(sp, lay.n_rb * lay.tiles_n * lay.slots * 2, L.F32).reshape(p.N, per, lay.tiles_n, lay.slots, 2).double()
            for tn in range(lay.tiles_n):
                for j in range(lay.slots):
                    U = (tn * lay.bno) // lay.unit + j
                    lo, hi = max(tn * lay.bno, U * lay.unit), min(min((tn + 1) * lay.bno, nch), (U + 1) * lay.unit)
                    if hi <= lo:
                        continue
                    g = (coff + lo) // cpg
                    n_i = float((hi - lo) * lay.rb_rows)
                    recs.append((g, n_i, rec[:, :, tn, j, 0], rec[:, :, tn, j, 1]))
                    cnt[:, g] += n_i * per
                    acc1[:, g] += n_i * rec[:, :, tn, j, 0].sum(dim=1)
        assert torch.all(cnt == p.HW * cpg), "statistics records do not cover every group exactly once"
        mean = acc1 / cnt
        m2 = torch.zeros_like(mean)
        for g, n_i, mi, m2i in recs:
            m2[:, g] += m2i.sum(dim=1) + (n_i * (mi - mean[:, g:g + 1]) ** 2).sum(dim=1)
        rstd = 1.0 / torch.sqrt(m2 / cnt + p.eps)
        xg = xa.reshape(p.N, p.HW, p.G, cpg).double()
        yv = (xg - mean[:, None, :, None]) * rstd[:, None, :, None]
        yv = yv.reshape(p.N, p.HW, p.C).float()
        if gamma:
            yv = yv * _flat(gamma, p.C, p.dtype).float()
        if beta:
            yv = yv + _flat(beta, p.C, p.dtype).float()
        if p.act == L.ACT_SILU:
            yv = torch.nn.functional.silu(yv)
        _flat(y, p.N * p.HW * p.C, p.dtype).reshape(p.N, p.HW, p.C).copy_(yv)
        return 0

    def sfast_hip_gemm_grouped(self, x, segs, bias, out, ref, n_groups, stream):
        p = _p(ref)
        self.calls.append("gemm_grouped")
        for g in range(n_groups):
            xin = _strided(x[g], (p.M, p.K), (p.ldx, 1), p.dtype)
            w = torch.cat([_strided(segs[g * p.n_wseg + j], (p.rows_per_seg, p.K), (p.ldw, 1), p.dtype) for j in range(p.n_wseg)], dim=0)[:p.N]
            b = _flat(bias[g], p.N, p.dtype) if bias and bias[g] else None
            _strided(out[g], (p.M, p.N), (p.ldo, 1), p.dtype).copy_(R.linear_ref(xin, w, b, _ACT[p.act]))
        return 0

    def sfast_hip_gemv_grouped(self, x, w, bias, out, ref, stream):
        p = _p(ref)
        self.calls.append("gemv_grouped")
        xin = _strided(x, (p.M, p.K), (p.ldx, 1), p.dtype)
        off = 0
        for g in range(p.n_groups):
            n = p.n_rows[g]
            wg = _strided(w[g], (n, p.K), (p.ldw, 1), p.dtype)
            bg = _flat(bias[g], n, p.dtype) if bias and bias[g] else None
            o = R.linear_ref(xin, wg, bg, _ACT[p.act], None, 1.0, False, False, None, 0, _ACT[p.in_act])
            _strided(out + off * 2, (p.M, n), (p.ldo, 1), p.dtype).copy_(o)
            off += n
        return 0

    def sfast_hip_gn_conv2d(self, x, x2, gamma, beta, w, bias, rowbias, z, out, ref, ws, ws_bytes, stream):
        """GroupNorm(+SiLU) of the (virtually concatenated) NHWC input, rounded to the I/O dtype as the two-operator path stores it,
        then the 3x3 conv with its epilogue -- the contract of sfast_hip_gn_conv2d."""
        q = _p(ref)
        p = q.conv
        self.calls.append("gn_conv2d")
        assert self.real.sfast_hip_gn_conv2d_supported(ref) == 1
        assert ws_bytes >= self.real.sfast_hip_gn_conv2d_workspace_bytes(ref)
        C1, C2 = p.C1, p.Cin - p.C1
        HW = p.H * p.W
        xa = _flat(x, p.B * HW * C1, p.dtype).reshape(p.B, HW, C1)
        if C2:
            xa = torch.cat([xa, _flat(x2, p.B * HW * C2, p.dtype).reshape(p.B, HW, C2)], dim=2)
        g = _flat(gamma, p.Cin, p.dtype).float() if gamma else None
        b = _flat(beta, p.Cin, p.dtype).float() if beta else None
        n = R.group_norm_ref(xa.permute(0, 2, 1), q.groups, g, b, q.eps, q.gn_act == L.ACT_SILU)      # [B, Cin, HW] fp32
        n = n.to(xa.dtype).reshape(p.B, p.Cin, p.H, p.W)

        def nchw(ptr, c, s):
            return _strided(ptr, (p.B, c, p.H, p.W), (s[0], s[3], s[1], s[2]), p.dtype)

        wt = _strided(w, (p.Cout, p.Cin, 3, 3), tuple(p.ws), p.dtype)
        bb = _flat(bias, p.Cout, p.dtype) if bias else None
        rb = _strided(rowbias, (p.B, p.Cout), (p.ld_rowbias, 1), p.dtype) if rowbias else None
        zz = nchw(z, p.Cout, p.zs).clone() if z else None
        o = R.conv2d_ref(n, wt, bb, zz, p.alpha, 1, 1, 1, _ACT[p.act], bool(p.res_before_act), None, False, rb)
        nchw(out, p.Cout, p.os).copy_(o)
        return 0

    def sfast_hip_conv2d(self, x, x2, w, bias, rowbias, z, out, ref, ws, ws_bytes, stream):
        p = _p(ref)
        self.calls.append("conv2d")
        assert ws_bytes >= self.real.sfast_hip_conv2d_workspace_bytes(ref)
        C1, C2 = p.C1, p.Cin - p.C1

        def nchw(ptr, c, h, w_, s):  # strides given as (n,h,w,c) -> logical NCHW view
            return _strided(ptr, (p.B, c, h, w_), (s[0], s[3], s[1], s[2]), p.dtype)

        xin = nchw(x, C1, p.H, p.W, p.xs)
        x2in = nchw(x2, C2, p.H, p.W, p.x2s) if C2 else None
        wt = _strided(w, (p.Cout, p.Cin, p.KH, p.KW), tuple(p.ws), p.dtype)
        Hin, Win = (2 * p.H, 2 * p.W) if p.upsample2x else (p.H, p.W)
        Ho = (Hin + 2 * p.pad_h + p.pad_h_extra - p.dil_h * (p.KH - 1) - 1) // p.stride_h + 1
        Wo = (Win + 2 * p.pad_w + p.pad_w_extra - p.dil_w * (p.KW - 1) - 1) // p.stride_w + 1
        if p.pad_h_extra or p.pad_w_extra:  # extra zero rows / columns at the bottom / right (no concat / upsample with it)
            assert x2in is None and not p.upsample2x
            xin = torch.nn.functional.pad(xin.float(), (0, p.pad_w_extra, 0, p.pad_h_extra)).to(xin.dtype)
        b = _flat(bias, p.Cout, p.dtype) if bias else None
        rb = _strided(rowbias, (p.B, p.Cout), (p.ld_rowbias, 1), p.dtype) if rowbias else None
        zz = nchw(z, p.Cout, Ho, Wo, p.zs).clone() if z else None
        o = R.conv2d_ref(xin, wt, b, zz, p.alpha, (p.stride_h, p.stride_w), (p.pad_h, p.pad_w), (p.dil_h, p.dil_w),
                         _ACT[p.act], bool(p.res_before_act), x2in, bool(p.upsample2x), rb)
        nchw(out, p.Cout, Ho, Wo, p.os).copy_(o)
        return 0

    def sfast_hip_attention(self, q, k, v, out, ref, stream):
        return self.sfast_hip_attention_bias(q, k, v, None, None, out, ref, stream)

    def sfast_hip_attention_bias(self, q, k, v, bias, bstr, out, ref, stream):
        p = _p(ref)
        self.calls.append("attention")
        qv = _strided(q, (p.B, p.Sq, p.H, p.D), tuple(p.qs) + (1,), p.dtype)
        kv = _strided(k, (p.B, p.Skv, p.H, p.D), tuple(p.ks) + (1,), p.dtype)
        vv = _strided(v, (p.B, p.Skv, p.H, p.D), tuple(p.vs) + (1,), p.dtype)
        bv = None
        if bias:
            st = tuple(bstr) if not hasattr(bstr, "_obj") else tuple(bstr._obj)
            bv = _strided(bias, (p.B, p.H, p.Sq, p.Skv), st + (1,), p.dtype)
        o = R.attention_ref(qv, kv, vv, p.scale, bv)
        _strided(out, (p.B, p.Sq, p.H, p.D), tuple(p.os) + (1,), p.dtype).copy_(o)
        return 0

    def sfast_hip_packed_weight_bytes(self, N, K):
        return self.real.sfast_hip_packed_weight_bytes(N, K)  # host-only

    def sfast_hip_pack_weight(self, w, packed, N, K, ldw, dtype, stream):
        """packed[(nb * KS + s) * 64 + lane] = the 8 elements w[nb * 32 + lane % 32][s * 16 + (lane // 32) * 8 : + 8], zeros outside."""
        self.calls.append("pack_weight")
        KS, NB = (K + 63) // 64 * 4, (N + 31) // 32
        src = _strided(w, (N, K), (ldw, 1), dtype)
        full = torch.zeros(NB * 32, KS * 16, dtype=src.dtype)
        full[:N, :K] = src
        out = full.reshape(NB, 32, KS, 2, 8).permute(0, 2, 3, 1, 4).contiguous()   # [nb][s][g][r][8]
        _flat(packed, NB * KS * 64 * 8, dtype).copy_(out.reshape(-1))
        return 0

    def sfast_hip_lora_merge_plan(self, entries, n, total):
        return self.real.sfast_hip_lora_merge_plan(entries, n, total)  # host-only: validates the table, fills tile_begin

    def sfast_hip_lora_merge(self, table, n, total_tiles, scales, dtype, stream):
        self.calls.append("lora_merge")
        ents = (L.LoraEntry * int(n)).from_address(int(table))
        sc = _flat(scales, n, L.F32) if scales else None
        for e in ents:
            w = _strided(e.w, (e.N, e.K), (e.ldw, 1), dtype).float()
            d = _strided(e.down, (e.r, e.K), (e.ldd, 1), dtype).float()
            u = _strided(e.up, (e.N, e.r), (e.ldu, 1), dtype).float()
            s = float(sc[e.scale_index]) if sc is not None else 1.0
            _flat(e.out, e.N * e.K, dtype).reshape(e.N, e.K).copy_(w + s * (u @ d))
        return 0

    def sfast_hip_mix_rows(self, x, y, vec, mix, out, ref, stream):
        p = _p(ref)
        self.calls.append("mix_rows")
        xv = _flat(x, p.M * p.C, p.dtype).reshape(p.M, p.C).float()
        wx, wy = p.wx, p.wy
        if mix:
            a = torch.sigmoid(_flat(mix, 1, p.dtype).float())[0]
            if p.switch_spatial_to_temporal:
                a = 1.0 - a
            wx, wy = a, 1.0 - a
        o = wx * xv
        if y:
            o = o + wy * _flat(y, p.M * p.C, p.dtype).reshape(p.M, p.C).float()
        if vec:
            rows = (torch.arange(p.M) // p.vec_rows) % p.vec_mod
            vv = _strided(vec, (p.vec_mod, p.C), (p.ld_vec, 1), p.dtype).float()
            o = o + vv[rows]
        _flat(out, p.M * p.C, p.dtype).reshape(p.M, p.C).copy_(o)
        return 0

    def sfast_hip_strided_copy(self, src, dst, ref, stream):
        p = _p(ref)
        self.calls.append("strided_copy")
        dt = {2: L.F16, 4: L.F32}[p.elem_bytes]
        shape = tuple(p.shape)[:p.ndim]
        s = _strided(src, shape, tuple(p.src_strides)[:p.ndim], dt)
        _strided(dst, shape, tuple(p.dst_strides)[:p.ndim], dt).copy_(s)
        return 0

    def sfast_hip_timestep_embedding(self, t, out, ref, stream):
        p = _p(ref)
        self.calls.append("timestep_embedding")
        tv = _flat(t, p.B, L.F32)
        e = temb_ref(tv, p.dim, bool(p.flip_sin_to_cos), p.downscale_freq_shift, p.max_period)
        _flat(out, p.B * p.dim, p.dtype).reshape(p.B, p.dim).copy_(e)
        return 0

    def sfast_hip_schedule_advance(self, cursor, ts_table, ts_cols, ts_out, coef_table, coef_cols, coef_out, n_steps, stream):
        self.calls.append("schedule_advance")
        cur = _flat(cursor, 1, "i32")
        idx = max(0, int(cur[0])) % n_steps
        if ts_cols:
            _flat(ts_out, ts_cols, L.F32).copy_(_flat(ts_table, n_steps * ts_cols, L.F32)[idx * ts_cols:(idx + 1) * ts_cols])
        if coef_cols:
            _flat(coef_out, coef_cols, L.F32).copy_(_flat(coef_table, n_steps * coef_cols, L.F32)[idx * coef_cols:(idx + 1) * coef_cols])
        cur[0] = (idx + 1) % n_steps
        return 0

    def sfast_hip_cfg_ddim_step(self, eps_uc, lat, lat_out, unet_in, coef, g, numel, dtype, stream):
        self.calls.append("cfg_ddim_step")
        g = float(getattr(g, "value", g))
        e = _flat(eps_uc, 2 * numel, dtype)
        x = _flat(lat, numel, dtype)
        c = _flat(coef, 4, L.F32)
        r = R.cfg_ddim_ref(e, x, c.tolist(), g).to(x.dtype)
        _flat(lat_out, numel, dtype).copy_(r)
        if unet_in:
            u = _flat(unet_in, 2 * numel, dtype)
            u[:numel].copy_(r)
            u[numel:].copy_(r)
        return 0


class EmuHost:
    """Stands where `sfast.engine.unet2d.DeviceHost` stands in product code: hands an engine the emulator as its library, accepts CPU
    parameters, and has no device, no streams and no measured tuning. The engines themselves carry no emulation branch."""

    def __init__(self, lib=None):
        self.lib = lib if lib is not None else EmuLib()

    def library(self):
        return self.lib

    def require_device(self, device, who):
        pass

    def init_device(self, device):
        pass

    def stream_ptr(self, device):
        return None

    def new_stream(self, device):
        return None

    def tuning(self):
        return False


def emulated_denoise_loop(engine, **kw):
    """`sfast.engine.denoise.DenoiseLoop` driven through the host emulator: the product class knows nothing about emulation (it always
    takes the real library and the device's current stream); the two seams it exposes are overridden HERE, in test code."""
    from sfast.engine.denoise import DenoiseLoop

    class EmulatedDenoiseLoop(DenoiseLoop):
        @staticmethod
        def _library(eng):
            return eng.lib           # the EmuLib instance injected into the engine

        def _stream_ptr(self):
            return None              # the emulator has no streams

    kw.setdefault("use_graph", False)
    return EmulatedDenoiseLoop(engine, **kw)
