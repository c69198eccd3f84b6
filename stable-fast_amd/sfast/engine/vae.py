"""Static-plan executor for the diffusers `AutoencoderKL` decoder (`vae.decoder`) on libsfast_hip.so.

The reference optimises the VAE through the same trace + fusion pipeline as the UNet
(`compile_vae`, /root/reference/src/sfast/compilers/diffusion_pipeline_compiler.py:154-190: memory format, xformers
attention patch, TorchScript fusion of conv+bias, GroupNorm+SiLU). Here the decoder is one more plan of the same
C-ABI launches the UNet engine uses: NHWC activations, fused GroupNorm+SiLU, implicit-GEMM
convs with the residual add and the nearest-2x upsample folded in, conv_in / conv_out through strides of the NCHW
latent / image.

The one op the UNet does not have is the mid block's single-head attention with head dim 512 (S = H*W tokens). That is
outside the flash kernel's register budget (O^T alone would be 256 accumulator registers per lane), so it runs as
    [q|k|v] = GN(x) . [Wq|Wk|Wv]^T      one GEMM, stacked weight segments
    S = q . k^T                          MFMA GEMM  [S, S] (f16 storage, fp32 accumulate)
    P = softmax(S / sqrt(512))           sfast_hip_softmax_rows (fp32 math, in place)
    O = P . v                            MFMA GEMM against v^T (strided-copy transpose)
    out = O . Wo^T + bo + x              GEMM with fused residual
per sample.
"""
import ctypes as C
import threading

import torch

from ..hip import lib as L
from .unet2d import EXT_FLAGS, DeviceHost, UNet2DEngine, UNetPlan, UnsupportedUNet, _Pool, _as2d, _cfg_get, live_norm_eps


class UnsupportedVae(UnsupportedUNet):
    pass


class VaeDecoderEngine(UNet2DEngine):
    """Executor for `AutoencoderKL.decoder` parameter sets (SD1.x / SD2.x / SDXL VAE family)."""

    def __init__(self, config, params, device=None, dtype=None, _host=None):
        self.host = _host if _host is not None else DeviceHost()
        self.lib = self.host.library()
        self.cfg = config
        self.params = params
        first = params["conv_in.weight"]
        self.device = device or first.device
        self.dtype = dtype or first.dtype
        if self.dtype not in (torch.float16, torch.bfloat16):
            raise UnsupportedVae(f"VaeDecoderEngine runs f16/bf16 parameters, got {self.dtype}")
        self.host.require_device(self.device, type(self).__name__)
        self.dt = L.F16 if self.dtype == torch.float16 else L.BF16
        self.esize = 2
        self.norm_eps = {}
        self._parse_config()
        self._plans = {}
        self._lock = threading.Lock()

    @classmethod
    def from_module(cls, m, config=None, _host=None):
        """Build from a diffusers-style `Decoder` module (live parameter storage). `config` may be the owning
        AutoencoderKL's config (only `norm_num_groups` is read; the layout comes from the parameter shapes)."""
        cfg = config if config is not None else getattr(m, "config", None)
        if cfg is None:
            gn = getattr(m, "conv_norm_out", None)
            cfg = {"norm_num_groups": getattr(gn, "num_groups", 32)}
        params = {}
        with torch.no_grad():
            for name, p in m.named_parameters():
                if p.ndim == 4 and not p.data.is_contiguous(memory_format=torch.channels_last):
                    p.data = p.data.contiguous(memory_format=torch.channels_last)
                params[name] = p.data
        eng = cls(cfg, params, _host=_host)
        eng.norm_eps = live_norm_eps(m)
        eng._param_objs = dict(m.named_parameters())
        return eng

    def _parse_config(self):
        P = self.params
        g = lambda k, d=None: _cfg_get(self.cfg, k, d)
        # everything structural is read off the parameter shapes (a bare `Decoder` module carries no config)
        self.in_ch = P["conv_in.weight"].shape[1]
        self.out_ch = P["conv_out.weight"].shape[0]
        self.groups = g("norm_num_groups", 32)
        self.eps = 1e-6
        n_up = 0
        while f"up_blocks.{n_up}.resnets.0.conv1.weight" in P:
            n_up += 1
        if n_up == 0:
            raise UnsupportedVae("no up_blocks.*.resnets found")
        self.n_up = n_up
        self.n_res = []
        self.up_out = []
        for i in range(n_up):
            j = 0
            while f"up_blocks.{i}.resnets.{j}.conv1.weight" in P:
                j += 1
            self.n_res.append(j)
            self.up_out.append(P[f"up_blocks.{i}.resnets.0.conv1.weight"].shape[0])
        self.mid_ch = P["conv_in.weight"].shape[0]
        self.has_attn = "mid_block.attentions.0.to_q.weight" in P
        if not self.has_attn and any(k.startswith("mid_block.attentions.0.") for k in P):
            raise UnsupportedVae("mid-block attention uses the deprecated query/key/value parameter names")
        if g("act_fn", "silu") not in ("silu", "swish"):
            raise UnsupportedVae("act_fn")
        for c in [self.mid_ch] + self.up_out:
            if c % self.groups or c % 8:
                raise UnsupportedVae(f"channel count {c} (needs a multiple of 8 and of the group count)")

    # ------------------------------------------------------------------------------------------
    def _vae_resnet(self, plan, pre, x, Cin, Cout, B, H, W):
        """ResnetBlock2D without a time embedding (temb_channels=None in the VAE)."""
        pool, P = plan.pool, self.params
        M = B * H * W
        n1 = pool.get(M * Cin)
        self._op_gn(plan, pre + ".norm1", x, None, Cin, Cin, B, H * W, n1, self.eps, True, pre + ".norm1")
        h1 = pool.get(M * Cout)
        self._op_conv(plan, pre + ".conv1", n1, None, P[pre + ".conv1.weight"], P[pre + ".conv1.bias"], h1, B, H, W, Cin, 0, Cout, 3, 1, 1)
        pool.put(n1)
        n2 = pool.get(M * Cout)
        self._op_gn(plan, pre + ".norm2", h1, None, Cout, Cout, B, H * W, n2, self.eps, True, pre + ".norm2")
        pool.put(h1)
        if (pre + ".conv_shortcut.weight") in P:
            sc = pool.get(M * Cout)
            self._op_gemm(plan, pre + ".conv_shortcut", x, [_as2d(P[pre + ".conv_shortcut.weight"], Cout, Cin)],
                          P[pre + ".conv_shortcut.bias"], sc, M, Cout, Cin, Cin, Cout, kind="conv1x1")
            res, own = sc, True
        else:
            if Cin != Cout:
                raise UnsupportedVae(f"{pre}: no conv_shortcut for {Cin}->{Cout}")
            res, own = x, False
        out = pool.get(M * Cout)
        self._op_conv(plan, pre + ".conv2", n2, None, P[pre + ".conv2.weight"], P[pre + ".conv2.bias"], out, B, H, W, Cout, 0, Cout, 3, 1, 1,
                      z=res)
        pool.put(n2)
        if own:
            pool.put(res)
        return out

    def _op_softmax(self, plan, name, x, M, N, ld, scale):
        lib = self.lib
        p = L.SoftmaxParams(self.dt, M, N, ld, ld, float(scale))
        xp = x.data_ptr()
        plan.keep.append(p)
        self._add(plan, "softmax", name, 0.0, 2.0 * M * N * self.esize,
                  lambda s, p=p: L.check(lib.sfast_hip_softmax_rows(xp, xp, C.byref(p), s), name))

    def _op_transpose(self, plan, name, src, src_off, rows, cols, src_ld, dst):
        """dst[c][r] = src[r][c] (dst contiguous [cols, rows])."""
        lib = self.lib
        cp = L.CopyParams()
        cp.elem_bytes, cp.ndim = 2, 2
        cp.shape = (C.c_int64 * 4)(cols, rows, 1, 1)
        cp.src_strides = (C.c_int64 * 4)(1, src_ld, 0, 0)
        cp.dst_strides = (C.c_int64 * 4)(rows, 1, 0, 0)
        plan.keep.append(cp)
        sp, dp = src.data_ptr() + src_off * self.esize, dst.data_ptr()
        self._add(plan, "misc", name, 0.0, 2.0 * rows * cols * self.esize,
                  lambda s, cp=cp: L.check(lib.sfast_hip_strided_copy(sp, dp, C.byref(cp), s), name))

    def _vae_attention(self, plan, pre, x, Cc, B, H, W):
        pool, P = plan.pool, self.params
        S = H * W
        M = B * S
        if S % 8:
            raise UnsupportedVae(f"attention over {S} tokens (needs a multiple of 8)")
        hn = pool.get(M * Cc)
        self._op_gn(plan, pre + ".group_norm", x, None, Cc, Cc, B, S, hn, self.eps, False, pre + ".group_norm")
        qkv = pool.get(M * 3 * Cc)
        for i, nm in enumerate(("to_q", "to_k", "to_v")):  # biases are separate live parameters: three launches into column slices
            self._op_gemm(plan, f"{pre}.{nm}", hn, [P[f"{pre}.{nm}.weight"]], P[f"{pre}.{nm}.bias"], qkv, M, Cc, Cc, Cc, 3 * Cc,
                          out_offset=i * Cc)
        pool.put(hn)
        o = pool.get(M * Cc)
        scores = pool.get(S * S)
        vt = pool.get(Cc * S)
        for b in range(B):
            base = b * S * 3 * Cc
            # S = q . k^T : x = q rows (ld 3C), "weight" rows = k rows (ld 3C)
            # logits leave the GEMM already scaled (fp32 accumulator x 1/sqrt(C) before the f16 store, sfast_epilogue_ext.out_scale):
            # unscaled q.k^T over 512 channels can exceed f16's range on real VAE activations
            self._op_gemm_raw(plan, f"{pre}.qk.{b}", qkv, base, qkv, base + Cc, 3 * Cc, scores, 0, S, S, Cc, 3 * Cc, S, kind="attn_vae",
                              out_scale=float(Cc) ** -0.5)
            self._op_softmax(plan, f"{pre}.softmax.{b}", scores, S, S, S, 1.0)
            self._op_transpose(plan, f"{pre}.vT.{b}", qkv, base + 2 * Cc, S, Cc, 3 * Cc, vt)
            # O = P . v : "weight" rows = v^T rows [C][S]
            self._op_gemm_raw(plan, f"{pre}.pv.{b}", scores, 0, vt, 0, S, o, b * S * Cc, S, Cc, S, S, Cc, kind="attn_vae")
        pool.put(scores)
        pool.put(vt)
        pool.put(qkv)
        out = pool.get(M * Cc)
        self._op_gemm(plan, pre + ".to_out.0", o, [P[pre + ".to_out.0.weight"]], P[pre + ".to_out.0.bias"], out, M, Cc, Cc, Cc, Cc,
                      residual=x, ldr=Cc)
        pool.put(o)
        return out

    def _op_gemm_raw(self, plan, name, x, x_off, w, w_off, ldw, out, out_off, M, N, K, ldx, ldo, kind="linear", out_scale=1.0):
        """out[M,N] = x[M,K] . w[N,K]^T on raw (buffer, element offset, leading dimension) operands."""
        lib = self.lib
        p = L.GemmParams()
        p.dtype, p.M, p.N, p.K = self.dt, M, N, K
        p.ldx, p.ldw, p.ldo, p.ldr = ldx, ldw, ldo, 0
        p.n_wseg, p.rows_per_seg = 1, N
        p.geglu, p.act, p.res_before_act, p.alpha = 0, L.ACT_NONE, 0, 1.0
        p.rows_per_batch, p.ld_rowbias, p.in_act, p.variant, p.split_k = 0, 0, 0, 0, 0
        self._need_ws(plan, lib.sfast_hip_gemm_workspace_bytes(C.byref(p)))
        segs = (C.c_void_p * 1)(w.data_ptr() + w_off * self.esize)
        xp = x.data_ptr() + x_off * self.esize
        op = out.data_ptr() + out_off * self.esize
        ws = plan.ws
        ext = L.EpilogueExt(float(out_scale), 0, 0, EXT_FLAGS)
        plan.keep += [p, segs, ext]
        plan.writer.pop(id(out), None)

        def launch(stream, p=p, segs=segs, ext=ext):
            L.check(lib.sfast_hip_gemm_ex(xp, segs, None, None, None, op, C.byref(p), C.byref(ext), None,
                                          ws[0].data_ptr() if ws[0] is not None else None, ws[1], stream), name)

        def launch_with(stream, ws_ptr, ws_bytes, p=p, segs=segs, ext=ext):
            return lib.sfast_hip_gemm_ex(xp, segs, None, None, None, op, C.byref(p), C.byref(ext), None, ws_ptr, ws_bytes, stream)

        self._add(plan, kind, name, 2.0 * M * N * K, (M * K + N * K + M * N) * self.esize, launch, tune=(p, launch_with))

    # ------------------------------------------------------------------------------------------
    def build_plan(self, B, H, W, S_ctx=0):
        self.host.init_device(self.device)
        P = self.params
        dev, dt = self.device, self.dtype
        plan = UNetPlan(self, B, H, W, 0)
        pool = plan.pool = _Pool(dev, dt)
        pool.writer = plan.writer
        scale = 1 << (self.n_up - 1)
        z = torch.zeros((B, self.in_ch, H, W), dtype=dt, device=dev)
        img = torch.zeros((B, self.out_ch, H * scale, W * scale), dtype=dt, device=dev)
        plan.static_in = {"sample": z}
        plan.static_out = img
        c = self.mid_ch
        h = pool.get(B * H * W * c)
        self._op_conv(plan, "conv_in", z, None, P["conv_in.weight"], P["conv_in.bias"], h, B, H, W, self.in_ch, 0, c, 3, 1, 1,
                      xs=(self.in_ch * H * W, W, 1, H * W), kind="conv_in")
        # ---- mid block -----------------------------------------------------------------------------
        hn = self._vae_resnet(plan, "mid_block.resnets.0", h, c, c, B, H, W)
        pool.put(h)
        h = hn
        if self.has_attn:
            hn = self._vae_attention(plan, "mid_block.attentions.0", h, c, B, H, W)
            pool.put(h)
            h = hn
        hn = self._vae_resnet(plan, "mid_block.resnets.1", h, c, c, B, H, W)
        pool.put(h)
        h = hn
        # ---- up blocks -----------------------------------------------------------------------------
        cH, cW, ch = H, W, c
        for i in range(self.n_up):
            co = self.up_out[i]
            for j in range(self.n_res[i]):
                hn = self._vae_resnet(plan, f"up_blocks.{i}.resnets.{j}", h, ch, co, B, cH, cW)
                pool.put(h)
                h, ch = hn, co
            un = f"up_blocks.{i}.upsamplers.0.conv"
            if (un + ".weight") in P:
                hu = pool.get(B * (2 * cH) * (2 * cW) * ch)
                self._op_conv(plan, un, h, None, P[un + ".weight"], P[un + ".bias"], hu, B, cH, cW, ch, 0, ch, 3, 1, 1, ups=True)
                pool.put(h)
                h = hu
                cH, cW = 2 * cH, 2 * cW
        if (cH, cW) != (H * scale, W * scale):
            raise UnsupportedVae("upsampler layout does not match the block count")
        # ---- out -----------------------------------------------------------------------------------
        nout = pool.get(B * cH * cW * ch)
        self._op_gn(plan, "conv_norm_out", h, None, ch, ch, B, cH * cW, nout, self.eps, True, "conv_norm_out")
        pool.put(h)
        self._op_conv(plan, "conv_out", nout, None, P["conv_out.weight"], P["conv_out.bias"], img, B, cH, cW, ch, 0, self.out_ch, 3, 1, 1,
                      os_=(self.out_ch * cH * cW, cW, 1, cH * cW), kind="conv_out")
        pool.put(nout)
        self._finish_plan(plan)
        return plan

    def get_plan(self, B, H, W, S_ctx=0):
        return super().get_plan(B, H, W, 0)

    def load_inputs(self, plan, z, *_, **__):
        plan.static_in["sample"].copy_(z)

    def forward(self, z):
        """Eager (no graph) execution on the current stream; returns a fresh NCHW image tensor."""
        B, _, H, W = z.shape
        plan = self.get_plan(B, H, W)
        self.sync_packed()  # pipe-4 launches read packed copies: follow the live parameters' version counters
        self.load_inputs(plan, z)
        plan.run(self.host.stream_ptr(self.device))
        return plan.static_out.clone()


class VaeEncoderEngine(VaeDecoderEngine):
    """Executor for `AutoencoderKL.encoder` parameter sets: conv_in -> DownEncoderBlock2D x N (ResnetBlock2D pairs, a
    stride-2 3x3 conv over F.pad(x, (0,1,0,1)) = conv with `pad_extra`) -> mid block -> GroupNorm+SiLU -> conv_out.
    Output = the moments tensor [B, 2*latent, H/2^(N-1), W/2^(N-1)] (quant_conv and the Gaussian sampling stay in
    the caller, as in diffusers' AutoencoderKL.encode)."""

    def _parse_config(self):
        P = self.params
        g = lambda k, d=None: _cfg_get(self.cfg, k, d)
        self.in_ch = P["conv_in.weight"].shape[1]
        self.out_ch = P["conv_out.weight"].shape[0]
        self.groups = g("norm_num_groups", 32)
        self.eps = 1e-6
        n = 0
        while f"down_blocks.{n}.resnets.0.conv1.weight" in P:
            n += 1
        if n == 0:
            raise UnsupportedVae("no down_blocks.*.resnets found")
        self.n_down = n
        self.n_res, self.down_out = [], []
        for i in range(n):
            j = 0
            while f"down_blocks.{i}.resnets.{j}.conv1.weight" in P:
                j += 1
            self.n_res.append(j)
            self.down_out.append(P[f"down_blocks.{i}.resnets.0.conv1.weight"].shape[0])
        self.mid_ch = P["conv_out.weight"].shape[1]
        self.has_attn = "mid_block.attentions.0.to_q.weight" in P
        if not self.has_attn and any(k.startswith("mid_block.attentions.0.") for k in P):
            raise UnsupportedVae("mid-block attention uses the deprecated query/key/value parameter names")
        for c in self.down_out:
            if c % self.groups or c % 8:
                raise UnsupportedVae(f"channel count {c} (needs a multiple of 8 and of the group count)")

    def build_plan(self, B, H, W, S_ctx=0):
        self.host.init_device(self.device)
        P = self.params
        dev, dt = self.device, self.dtype
        n_ds = sum(1 for i in range(self.n_down) if f"down_blocks.{i}.downsamplers.0.conv.weight" in P)
        if H % (1 << n_ds) or W % (1 << n_ds):
            raise UnsupportedVae(f"image {H}x{W} not divisible by {1 << n_ds}")
        plan = UNetPlan(self, B, H, W, 0)
        pool = plan.pool = _Pool(dev, dt)
        pool.writer = plan.writer
        img = torch.zeros((B, self.in_ch, H, W), dtype=dt, device=dev)
        out = torch.zeros((B, self.out_ch, H >> n_ds, W >> n_ds), dtype=dt, device=dev)
        plan.static_in = {"sample": img}
        plan.static_out = out
        c = P["conv_in.weight"].shape[0]
        h = pool.get(B * H * W * c)
        self._op_conv(plan, "conv_in", img, None, P["conv_in.weight"], P["conv_in.bias"], h, B, H, W, self.in_ch, 0, c, 3, 1, 1,
                      xs=(self.in_ch * H * W, W, 1, H * W), kind="conv_in")
        cH, cW, ch = H, W, c
        for i in range(self.n_down):
            co = self.down_out[i]
            for j in range(self.n_res[i]):
                hn = self._vae_resnet(plan, f"down_blocks.{i}.resnets.{j}", h, ch, co, B, cH, cW)
                pool.put(h)
                h, ch = hn, co
            dn = f"down_blocks.{i}.downsamplers.0.conv"
            if (dn + ".weight") in P:
                hd = pool.get(B * (cH // 2) * (cW // 2) * ch)
                self._op_conv(plan, dn, h, None, P[dn + ".weight"], P[dn + ".bias"], hd, B, cH, cW, ch, 0, ch, 3, 2, 0, pad_extra=1)
                pool.put(h)
                h = hd
                cH, cW = cH // 2, cW // 2
        hn = self._vae_resnet(plan, "mid_block.resnets.0", h, ch, ch, B, cH, cW)
        pool.put(h)
        h = hn
        if self.has_attn:
            hn = self._vae_attention(plan, "mid_block.attentions.0", h, ch, B, cH, cW)
            pool.put(h)
            h = hn
        hn = self._vae_resnet(plan, "mid_block.resnets.1", h, ch, ch, B, cH, cW)
        pool.put(h)
        h = hn
        nout = pool.get(B * cH * cW * ch)
        self._op_gn(plan, "conv_norm_out", h, None, ch, ch, B, cH * cW, nout, self.eps, True, "conv_norm_out")
        pool.put(h)
        self._op_conv(plan, "conv_out", nout, None, P["conv_out.weight"], P["conv_out.bias"], out, B, cH, cW, ch, 0, self.out_ch, 3, 1, 1,
                      os_=(self.out_ch * cH * cW, cW, 1, cH * cW), kind="conv_out")
        pool.put(nout)
        self._finish_plan(plan)
        return plan
