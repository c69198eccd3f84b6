"""MI355X-native executor of the spatio-temporal UNet of Stable Video Diffusion (diffusers `UNetSpatioTemporalConditionModel`).

BASELINE.json configs[4] / SURVEY.md section 8f rank 4: the model the reference accelerates in
/root/reference/examples/optimize_stable_video_diffusion_pipeline.py (it hands `pipe.unet` to the same compile(); the architecture
itself is diffusers'). Same construction as `UNet2DEngine`: a static plan of C-ABI launches on NHWC buffers, frames folded into the
batch ([B*F, H*W, C] = [B, F, H*W, C] row-major), captured into one hipGraph. What the temporal path adds, in kernel terms:

  * TemporalResnetBlock: GroupNorm over (C/G, F, H, W) per VIDEO = the NHWC GroupNorm kernels with N = B, "pixels" = F*H*W (free:
    frames are contiguous); Conv3d (3,1,1) = the implicit-GEMM conv with a 3 x 1 kernel over the [F, H*W] "image" of each video
    (weights converted once to channels_last_3d = K-contiguous); time-embedding row bias per video.
  * temporal attention: sequences of F frames at every spatial site = the flash kernel on STRIDED views of the fused QKV buffer
    (batch = site, stride 3C; sequence = frame, stride H*W*3C): no [B,F,S,C] <-> [B,S,F,C] transposes are materialised.
  * every cross-attention of this model attends to ONE context token (the CLIP image embedding), so softmax == 1 and the layer is
    exactly to_out(to_v(context)): two GEMVs per layer on the text/image side lane + one row-broadcast add; to_q / to_k never run.
  * AlphaBlender, frame-position embedding: `sfast_hip_mix_rows` (sigmoid of the live mix_factor evaluated on the device).
"""
import ctypes as C

import torch

from ..hip import lib as L
from .unet2d import LANE_KV, LANE_MAIN, LANE_TEMB, UNet2DEngine, UNetPlan, UnsupportedUNet, _Pool, _cfg_get, live_norm_eps


class SVDUNetEngine(UNet2DEngine):
    """Executor for `UNetSpatioTemporalConditionModel` parameter sets (SVD / SVD-XT)."""

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_module(cls, m, _host=None):
        cfg = getattr(m, "config", None)
        if cfg is None:
            raise UnsupportedUNet("module has no .config")
        params = {}
        with torch.no_grad():
            for name, p in m.named_parameters():
                d = p.data
                if d.ndim == 4 and not d.is_contiguous(memory_format=torch.channels_last):
                    p.data = d = d.contiguous(memory_format=torch.channels_last)
                elif d.ndim == 5 and not d.is_contiguous(memory_format=torch.channels_last_3d):
                    # Conv3d (3,1,1): [Cout][kt][1][1][Cin] physical order = the K-contiguous image of a 3 x 1 conv
                    p.data = d = d.contiguous(memory_format=torch.channels_last_3d)
                params[name] = d
        eng = cls(cfg, params, _host=_host)
        eng.norm_eps = live_norm_eps(m)
        eng._param_objs = dict(m.named_parameters())
        return eng

    def _parse_config(self):
        g = lambda k, d=None: _cfg_get(self.cfg, k, d)
        self.boc = tuple(g("block_out_channels"))
        n = len(self.boc)
        self.layers = g("layers_per_block", 2)
        if isinstance(self.layers, (tuple, list)):
            if len(set(self.layers)) != 1:
                raise UnsupportedUNet("per-block layers_per_block")
            self.layers = self.layers[0]
        self.down_types = tuple(g("down_block_types"))
        self.up_types = tuple(g("up_block_types"))
        for t in self.down_types:
            if t not in ("CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal"):
                raise UnsupportedUNet(f"down block {t}")
        for t in self.up_types:
            if t not in ("CrossAttnUpBlockSpatioTemporal", "UpBlockSpatioTemporal"):
                raise UnsupportedUNet(f"up block {t}")
        heads = g("num_attention_heads")
        self.heads = tuple(heads) if isinstance(heads, (tuple, list)) else (heads,) * n
        d = g("transformer_layers_per_block", 1)
        self.depth = tuple(d) if isinstance(d, (tuple, list)) else (d,) * n
        self.groups = g("norm_num_groups", 32) or 32
        self.ctx_dim = g("cross_attention_dim")
        self.in_ch, self.out_ch = g("in_channels", 8), g("out_channels", 4)
        self.add_time_dim = g("addition_time_embed_dim")
        self.is_controlnet = False
        self.eps = 1e-6  # spatial ResnetBlock2D of SpatioTemporalResBlock; the temporal blocks and conv_norm_out use 1e-5
        self.add_type = "time_ids"
        self.linear_proj = True
        P = self.params
        self.temb_dim = P["time_embedding.linear_1.weight"].shape[0]
        need = ["conv_in.weight", "add_embedding.linear_1.weight", "conv_norm_out.weight", "conv_out.weight",
                "down_blocks.0.resnets.0.spatial_res_block.conv1.weight", "down_blocks.0.resnets.0.temporal_res_block.conv1.weight",
                "down_blocks.0.resnets.0.time_mixer.mix_factor", "mid_block.attentions.0.temporal_transformer_blocks.0.ff_in.net.0.proj.weight"]
        missing = [k for k in need if k not in P]
        if missing:
            raise UnsupportedUNet(f"parameters missing for the spatio-temporal plan: {missing[:3]}")
        if P["add_embedding.linear_1.weight"].shape[1] % self.add_time_dim:
            raise UnsupportedUNet("projection_class_embeddings_input_dim is not a multiple of addition_time_embed_dim")
        self.n_time_ids = P["add_embedding.linear_1.weight"].shape[1] // self.add_time_dim

    # ------------------------------------------------------------------------------------------
    def _single_key_cross_attention(self, plan, pre, ctx, B, Cc):
        """Cross-attention against ONE context token: softmax over a single key is exactly 1, so the layer's output is
        to_out(to_v(context)) for every query -- a [B, C] table computed on the context side lane. Returns that table."""
        P = self.params
        v = torch.empty(B * Cc, dtype=self.dtype, device=self.device)
        o = torch.empty(B * Cc, dtype=self.dtype, device=self.device)
        plan.keep += [v, o]
        self._op_gemm(plan, pre + ".to_v", ctx, [P[pre + ".to_v.weight"]], None, v, B, Cc, self.ctx_dim, self.ctx_dim, Cc, kind="linear", lane=LANE_KV)
        self._op_gemm(plan, pre + ".to_out", v, [P[pre + ".to_out.0.weight"]], P[pre + ".to_out.0.bias"], o, B, Cc, Cc, Cc, Cc, kind="linear",
                      lane=LANE_KV)
        return o

    def _st_resnet(self, plan, pre, x, x2, C1, C2, Cout, B, Fr, H, W, temb_bf, temb_ld, offs):
        """SpatioTemporalResBlock: spatial ResnetBlock2D per frame -> TemporalResnetBlock over frames -> AlphaBlender."""
        pool, P = plan.pool, self.params
        BF, HW = B * Fr, H * W
        sp, tp = pre + ".spatial_res_block", pre + ".temporal_res_block"
        eps_t = 1e-5
        hs = self._resnet(plan, sp, x, x2, C1, C2, Cout, BF, H, W, temb_bf, temb_ld, offs[sp])  # GroupNorm eps = self.eps (1e-6)
        M = BF * HW
        # temporal block: "image" of a video = [F rows, H*W columns]; GroupNorm statistics per video
        n1 = pool.get(M * Cout)
        self._op_gn(plan, tp + ".norm1", hs, None, Cout, Cout, B, Fr * HW, n1, eps_t, True, tp + ".norm1")
        h1 = pool.get(M * Cout)
        w1 = P[tp + ".conv1.weight"]
        self._op_conv(plan, tp + ".conv1", n1, None, w1.reshape(w1.shape[0], w1.shape[1], 3, 1) if w1.ndim == 5 else w1, P[tp + ".conv1.bias"], h1,
                      B, Fr, HW, Cout, 0, Cout, 3, 1, 1, kw=1, pad_w=0, rowbias=temb_bf, ld_rowbias=temb_ld * Fr, rowbias_offset=offs[tp],
                      kind="conv_temporal")
        pool.put(n1)
        n2 = pool.get(M * Cout)
        self._op_gn(plan, tp + ".norm2", h1, None, Cout, Cout, B, Fr * HW, n2, eps_t, True, tp + ".norm2")
        pool.put(h1)
        ht = pool.get(M * Cout)
        w2 = P[tp + ".conv2.weight"]
        self._op_conv(plan, tp + ".conv2", n2, None, w2.reshape(w2.shape[0], w2.shape[1], 3, 1) if w2.ndim == 5 else w2, P[tp + ".conv2.bias"], ht,
                      B, Fr, HW, Cout, 0, Cout, 3, 1, 1, kw=1, pad_w=0, z=hs, kind="conv_temporal")
        pool.put(n2)
        # out = (1 - s) * spatial + s * temporal, s = sigmoid(mix_factor)   (switch_spatial_to_temporal_mix)
        self._op_mix(plan, pre + ".time_mixer", hs, ht, None, hs, M, Cout, mix=P[pre + ".time_mixer.mix_factor"], switch=True)
        pool.put(ht)
        return hs

    def _st_transformer(self, plan, pre, x, Cc, B, Fr, H, W, heads, depth, ctx):
        pool, P = plan.pool, self.params
        BF, S = B * Fr, H * W
        M = BF * S
        D = Cc // heads
        g = pool.get(M * Cc)
        self._op_gn(plan, pre + ".norm", x, None, Cc, Cc, BF, S, g, 1e-6, False, pre + ".norm")
        t = pool.get(M * Cc)
        self._op_gemm(plan, pre + ".proj_in", g, [P[pre + ".proj_in.weight"]], P[pre + ".proj_in.bias"], t, M, Cc, Cc, Cc, Cc)
        pool.put(g)
        # frame-position embedding [F, C]: sinusoid(arange(F)) -> MLP; depends on the weights only (side lane)
        fe0 = torch.empty(Fr * Cc, dtype=self.dtype, device=self.device)
        fe1 = torch.empty(Fr * 4 * Cc, dtype=self.dtype, device=self.device)
        femb = torch.empty(Fr * Cc, dtype=self.dtype, device=self.device)
        plan.keep += [fe0, fe1, femb]
        tp_ = L.TembParams(self.dt, Fr, Cc, 1, 0.0, 10000.0)
        plan.keep.append(tp_)
        fi_ptr, fe0_ptr = plan.frame_index.data_ptr(), fe0.data_ptr()
        lib = self.lib
        self._add(plan, "misc", pre + ".time_proj", 0.0, Fr * Cc * 2.0,
                  lambda s, tp_=tp_: L.check(lib.sfast_hip_timestep_embedding(fi_ptr, fe0_ptr, C.byref(tp_), s), "time_proj"), lane=LANE_KV)
        self._op_gemm(plan, pre + ".time_pos_embed.linear_1", fe0, [P[pre + ".time_pos_embed.linear_1.weight"]], P[pre + ".time_pos_embed.linear_1.bias"],
                      fe1, Fr, 4 * Cc, Cc, Cc, 4 * Cc, act=L.ACT_SILU, lane=LANE_KV)
        self._op_gemm(plan, pre + ".time_pos_embed.linear_2", fe1, [P[pre + ".time_pos_embed.linear_2.weight"]], P[pre + ".time_pos_embed.linear_2.bias"],
                      femb, Fr, Cc, 4 * Cc, 4 * Cc, Cc, lane=LANE_KV)
        for d in range(depth):
            # ---- spatial BasicTransformerBlock (tokens of one frame) ----------------------------------------------------
            bp = f"{pre}.transformer_blocks.{d}"
            n = pool.get(M * Cc)
            self._op_ln(plan, bp + ".norm1", t, n, M, Cc, bp + ".norm1")
            qkv = pool.get(M * 3 * Cc)
            self._op_gemm(plan, bp + ".attn1.to_qkv", n, [P[bp + ".attn1.to_q.weight"], P[bp + ".attn1.to_k.weight"], P[bp + ".attn1.to_v.weight"]],
                          None, qkv, M, 3 * Cc, Cc, Cc, 3 * Cc)
            a = pool.get(M * Cc)
            st = (S * 3 * Cc, 3 * Cc, D)
            self._op_attn(plan, bp + ".attn1", qkv, qkv, qkv, a, BF, heads, S, S, D, st, st, st, (S * Cc, Cc, D), q_off=0, k_off=Cc, v_off=2 * Cc)
            self._op_gemm(plan, bp + ".attn1.to_out", a, [P[bp + ".attn1.to_out.0.weight"]], P[bp + ".attn1.to_out.0.bias"], t, M, Cc, Cc, Cc, Cc,
                          residual=t, ldr=Cc)
            c2 = self._single_key_cross_attention(plan, bp + ".attn2", ctx, B, Cc)
            self._op_mix(plan, bp + ".attn2.add", t, None, c2, t, M, Cc, vec_rows=Fr * S, vec_mod=B, ld_vec=Cc, needs=LANE_KV)
            self._op_ln(plan, bp + ".norm3", t, n, M, Cc, bp + ".norm3")
            gg = pool.get(M * 4 * Cc)
            self._op_gemm(plan, bp + ".ff.geglu", n, [P[bp + ".ff.net.0.proj.weight"]], P[bp + ".ff.net.0.proj.bias"], gg, M, 4 * Cc, Cc, Cc, 4 * Cc,
                          geglu=True)
            self._op_gemm(plan, bp + ".ff.out", gg, [P[bp + ".ff.net.2.weight"]], P[bp + ".ff.net.2.bias"], t, M, Cc, 4 * Cc, 4 * Cc, Cc, residual=t, ldr=Cc)
            # ---- temporal block on hm = t + frame embedding (rows stay in [B, F, S] order; only attention sees the permutation) ----
            tb = f"{pre}.temporal_transformer_blocks.{d}"
            hm = pool.get(M * Cc)
            self._op_mix(plan, tb + ".add_frame_embedding", t, None, femb, hm, M, Cc, vec_rows=S, vec_mod=Fr, ld_vec=Cc, needs=LANE_KV)
            self._op_ln(plan, tb + ".norm_in", hm, n, M, Cc, tb + ".norm_in")
            self._op_gemm(plan, tb + ".ff_in.geglu", n, [P[tb + ".ff_in.net.0.proj.weight"]], P[tb + ".ff_in.net.0.proj.bias"], gg, M, 4 * Cc, Cc, Cc,
                          4 * Cc, geglu=True)
            self._op_gemm(plan, tb + ".ff_in.out", gg, [P[tb + ".ff_in.net.2.weight"]], P[tb + ".ff_in.net.2.bias"], hm, M, Cc, 4 * Cc, 4 * Cc, Cc,
                          residual=hm, ldr=Cc)
            self._op_ln(plan, tb + ".norm1", hm, n, M, Cc, tb + ".norm1")
            self._op_gemm(plan, tb + ".attn1.to_qkv", n, [P[tb + ".attn1.to_q.weight"], P[tb + ".attn1.to_k.weight"], P[tb + ".attn1.to_v.weight"]],
                          None, qkv, M, 3 * Cc, Cc, Cc, 3 * Cc)
            # sequences of F frames at every spatial site: batch = site (stride 3C), sequence = frame (stride S*3C); one launch per video
            tst = (3 * Cc, S * 3 * Cc, D)
            for b in range(B):
                base_q = b * Fr * S * 3 * Cc
                base_o = b * Fr * S * Cc
                self._op_attn(plan, f"{tb}.attn1.{b}", qkv, qkv, qkv, a, S, heads, Fr, Fr, D, tst, tst, tst, (Cc, S * Cc, D), q_off=base_q,
                              k_off=base_q + Cc, v_off=base_q + 2 * Cc, out_off=base_o, kind="attn_temporal", variant=2)
            self._op_gemm(plan, tb + ".attn1.to_out", a, [P[tb + ".attn1.to_out.0.weight"]], P[tb + ".attn1.to_out.0.bias"], hm, M, Cc, Cc, Cc, Cc,
                          residual=hm, ldr=Cc)
            c2t = self._single_key_cross_attention(plan, tb + ".attn2", ctx, B, Cc)   # time_context = the first frame's (= every frame's) context
            self._op_mix(plan, tb + ".attn2.add", hm, None, c2t, hm, M, Cc, vec_rows=Fr * S, vec_mod=B, ld_vec=Cc, needs=LANE_KV)
            self._op_ln(plan, tb + ".norm3", hm, n, M, Cc, tb + ".norm3")
            self._op_gemm(plan, tb + ".ff.geglu", n, [P[tb + ".ff.net.0.proj.weight"]], P[tb + ".ff.net.0.proj.bias"], gg, M, 4 * Cc, Cc, Cc, 4 * Cc,
                          geglu=True)
            self._op_gemm(plan, tb + ".ff.out", gg, [P[tb + ".ff.net.2.weight"]], P[tb + ".ff.net.2.bias"], hm, M, Cc, 4 * Cc, 4 * Cc, Cc, residual=hm,
                          ldr=Cc)
            pool.put(gg)
            pool.put(qkv)
            pool.put(a)
            pool.put(n)
            # t = s * spatial + (1 - s) * temporal
            self._op_mix(plan, pre + f".time_mixer.{d}", t, hm, None, t, M, Cc, mix=P[pre + ".time_mixer.mix_factor"], switch=False)
            pool.put(hm)
        out = pool.get(M * Cc)
        self._op_gemm(plan, pre + ".proj_out", t, [P[pre + ".proj_out.weight"]], P[pre + ".proj_out.bias"], out, M, Cc, Cc, Cc, Cc, residual=x, ldr=Cc)
        pool.put(t)
        return out

    def _st_resnet_names(self):
        names = []
        for i in range(len(self.down_types)):
            names += [f"down_blocks.{i}.resnets.{j}" for j in range(self.layers)]
        names += ["mid_block.resnets.0", "mid_block.resnets.1"]
        for i in range(len(self.up_types)):
            names += [f"up_blocks.{i}.resnets.{j}" for j in range(self.layers + 1)]
        out = []
        for n in names:
            out += [n + ".spatial_res_block", n + ".temporal_res_block"]
        return out

    # ------------------------------------------------------------------------------------------
    def build_plan(self, B, Fr, H, W):
        self.host.init_device(self.device)
        nlev = len(self.boc)
        if H % (1 << (nlev - 1)) or W % (1 << (nlev - 1)):
            raise UnsupportedUNet(f"latent {H}x{W} not divisible by {1 << (nlev - 1)}")
        BF = B * Fr
        if BF > 64:
            raise UnsupportedUNet(f"{B} videos x {Fr} frames: the time-embedding projections take at most 64 rows per launch")
        P = self.params
        dev, dt = self.device, self.dtype
        plan = UNetPlan(self, BF, H, W, 1)
        plan.frames = Fr
        pool = plan.pool = _Pool(dev, dt)
        pool.writer = plan.writer
        sample = torch.zeros((B, Fr, self.in_ch, H, W), dtype=dt, device=dev)
        tbuf = torch.zeros((B,), dtype=torch.float32, device=dev)
        ctx = torch.zeros((B, 1, self.ctx_dim), dtype=dt, device=dev)
        tids = torch.zeros((B * self.n_time_ids,), dtype=torch.float32, device=dev)
        out = torch.zeros((B, Fr, self.out_ch, H, W), dtype=dt, device=dev)
        plan.static_in = {"sample": sample, "timestep": tbuf, "encoder_hidden_states": ctx, "added_time_ids": tids}
        plan.static_out = out
        plan.frame_index = torch.arange(Fr, dtype=torch.float32, device=dev)
        lib = self.lib
        c0, T, td = self.boc[0], self.temb_dim, self.add_time_dim
        # ---- time embedding: emb = MLP(sinusoid(t)) + add_embedding(sinusoid(added_time_ids).flatten) ; act_emb = silu(emb) ----
        t_emb = pool.get(B * c0)
        tp = L.TembParams(self.dt, B, c0, 1, 0.0, 10000.0)
        tp2 = L.TembParams(self.dt, B * self.n_time_ids, td, 1, 0.0, 10000.0)
        plan.keep += [tp, tp2]
        tb_ptr, te_ptr = tbuf.data_ptr(), t_emb.data_ptr()
        self._add(plan, "misc", "timestep_embedding", 0.0, B * c0 * 2.0,
                  lambda s: L.check(lib.sfast_hip_timestep_embedding(tb_ptr, te_ptr, C.byref(tp), s), "timestep_embedding"), lane=LANE_TEMB)
        e1 = pool.get(B * T)
        self._op_gemm(plan, "time_embedding.linear_1", t_emb, [P["time_embedding.linear_1.weight"]], P["time_embedding.linear_1.bias"], e1, B, T, c0,
                      c0, T, act=L.ACT_SILU, kind="temb", lane=LANE_TEMB)
        Din = self.n_time_ids * td
        tide = pool.get(B * Din)
        ti_ptr, tide_ptr = tids.data_ptr(), tide.data_ptr()
        self._add(plan, "misc", "add_time_ids_embedding", 0.0, B * Din * 2.0,
                  lambda s: L.check(lib.sfast_hip_timestep_embedding(ti_ptr, tide_ptr, C.byref(tp2), s), "time_ids"), lane=LANE_TEMB)
        a1 = pool.get(B * T)
        self._op_gemm(plan, "add_embedding.linear_1", tide, [P["add_embedding.linear_1.weight"]], P["add_embedding.linear_1.bias"], a1, B, T, Din, Din,
                      T, act=L.ACT_SILU, kind="temb", lane=LANE_TEMB)
        aug = pool.get(B * T)
        self._op_gemm(plan, "add_embedding.linear_2", a1, [P["add_embedding.linear_2.weight"]], P["add_embedding.linear_2.bias"], aug, B, T, T, T, T,
                      kind="temb", lane=LANE_TEMB)
        act_emb = pool.get(B * T)
        self._op_gemm(plan, "time_embedding.linear_2", e1, [P["time_embedding.linear_2.weight"]], P["time_embedding.linear_2.bias"], act_emb, B, T, T,
                      T, T, act=L.ACT_SILU, residual=aug, ldr=T, res_before_act=True, kind="temb", lane=LANE_TEMB)
        # silu(emb) repeated per frame -> every time_emb_proj (spatial AND temporal resnets) in grouped GEMV launches: [B*F, tot]
        act_bf = pool.get(BF * T)
        cp = L.CopyParams()
        cp.elem_bytes, cp.ndim = 2, 3
        cp.shape = (C.c_int64 * 4)(B, Fr, T, 1)
        cp.src_strides = (C.c_int64 * 4)(T, 0, 1, 0)
        cp.dst_strides = (C.c_int64 * 4)(Fr * T, T, 1, 0)
        plan.keep.append(cp)
        ae_ptr, ab_ptr = act_emb.data_ptr(), act_bf.data_ptr()
        self._add(plan, "misc", "emb.repeat_per_frame", 0.0, BF * T * 4.0,
                  lambda s, cp=cp: L.check(lib.sfast_hip_strided_copy(ae_ptr, ab_ptr, C.byref(cp), s), "emb.repeat_per_frame"), lane=LANE_TEMB)
        rnames = self._st_resnet_names()
        offs, tot = {}, 0
        for rn in rnames:
            offs[rn] = tot
            tot += P[rn + ".time_emb_proj.weight"].shape[0]
        temb_bf = pool.get(BF * tot)
        for g0 in range(0, len(rnames), L.MAX_GROUPS):
            grp = rnames[g0:g0 + L.MAX_GROUPS]
            self._op_gemv_grouped(plan, f"time_emb_proj[{g0}:{g0 + len(grp)}]", act_bf, [P[rn + ".time_emb_proj.weight"] for rn in grp],
                                  [P[rn + ".time_emb_proj.bias"] for rn in grp], temb_bf, BF, T, T, tot, out_offset=offs[grp[0]], lane=LANE_TEMB)
        # ---- conv_in on the frames (NCHW per frame, read through strides) -----------------------------------------------------
        h = pool.get(BF * H * W * c0)
        if self.in_ch % 8 == 0:
            # 8 latent + conditioning channels: one strided copy to NHWC makes conv_in an MFMA implicit GEMM (K = 72) -- read through
            # NCHW strides it ran on the generic small-channel kernel: 18 ms of a 292 ms step at 25 x 72 x 128 (profiles/r02_bench_svd_run5)
            x_nhwc = pool.get(BF * H * W * self.in_ch)
            cpi = L.CopyParams()
            cpi.elem_bytes, cpi.ndim = 2, 3
            cpi.shape = (C.c_int64 * 4)(BF, H * W, self.in_ch, 1)
            cpi.src_strides = (C.c_int64 * 4)(self.in_ch * H * W, 1, H * W, 0)
            cpi.dst_strides = (C.c_int64 * 4)(H * W * self.in_ch, self.in_ch, 1, 0)
            plan.keep.append(cpi)
            sp_, dp_ = sample.data_ptr(), x_nhwc.data_ptr()
            self._add(plan, "misc", "sample.to_nhwc", 0.0, 2.0 * BF * H * W * self.in_ch * self.esize,
                      lambda s, cpi=cpi: L.check(lib.sfast_hip_strided_copy(sp_, dp_, C.byref(cpi), s), "sample.to_nhwc"))
            self._op_conv(plan, "conv_in", x_nhwc, None, P["conv_in.weight"], P["conv_in.bias"], h, BF, H, W, self.in_ch, 0, c0, 3, 1, 1,
                          kind="conv_in")
            pool.put(x_nhwc)
        else:
            self._op_conv(plan, "conv_in", sample, None, P["conv_in.weight"], P["conv_in.bias"], h, BF, H, W, self.in_ch, 0, c0, 3, 1, 1,
                          xs=(self.in_ch * H * W, W, 1, H * W), kind="conv_in")
        skips = [(h, c0)]
        ch, cH, cW = c0, H, W
        for i, t in enumerate(self.down_types):
            co = self.boc[i]
            for j in range(self.layers):
                rn = f"down_blocks.{i}.resnets.{j}"
                hn = self._st_resnet(plan, rn, h, None, ch, 0, co, B, Fr, cH, cW, temb_bf, tot, offs)
                ch = co
                if t == "CrossAttnDownBlockSpatioTemporal":
                    ha = self._st_transformer(plan, f"down_blocks.{i}.attentions.{j}", hn, co, B, Fr, cH, cW, self.heads[i], self.depth[i], ctx)
                    pool.put(hn)
                    hn = ha
                h = hn
                skips.append((h, ch))
            if i < nlev - 1:
                dn = f"down_blocks.{i}.downsamplers.0.conv"
                hd = pool.get(BF * (cH // 2) * (cW // 2) * ch)
                self._op_conv(plan, dn, h, None, P[dn + ".weight"], P[dn + ".bias"], hd, BF, cH, cW, ch, 0, ch, 3, 2, 1)
                cH, cW = cH // 2, cW // 2
                h = hd
                skips.append((h, ch))
        hm = self._st_resnet(plan, "mid_block.resnets.0", h, None, ch, 0, ch, B, Fr, cH, cW, temb_bf, tot, offs)
        ha = self._st_transformer(plan, "mid_block.attentions.0", hm, ch, B, Fr, cH, cW, self.heads[-1], self.depth[-1], ctx)
        pool.put(hm)
        h = self._st_resnet(plan, "mid_block.resnets.1", ha, None, ch, 0, ch, B, Fr, cH, cW, temb_bf, tot, offs)
        pool.put(ha)
        rboc, rheads, rdepth = self.boc[::-1], self.heads[::-1], self.depth[::-1]
        for i, t in enumerate(self.up_types):
            co = rboc[i]
            for j in range(self.layers + 1):
                rn = f"up_blocks.{i}.resnets.{j}"
                sk, sc_ = skips.pop()
                hn = self._st_resnet(plan, rn, h, sk, ch, sc_, co, B, Fr, cH, cW, temb_bf, tot, offs)
                pool.put(h)
                pool.put(sk)
                ch = co
                if t == "CrossAttnUpBlockSpatioTemporal":
                    ha = self._st_transformer(plan, f"up_blocks.{i}.attentions.{j}", hn, co, B, Fr, cH, cW, rheads[i], rdepth[i], ctx)
                    pool.put(hn)
                    hn = ha
                h = hn
            if i < nlev - 1:
                un = f"up_blocks.{i}.upsamplers.0.conv"
                hu = pool.get(BF * (2 * cH) * (2 * cW) * ch)
                self._op_conv(plan, un, h, None, P[un + ".weight"], P[un + ".bias"], hu, BF, cH, cW, ch, 0, ch, 3, 1, 1, ups=True)
                pool.put(h)
                h = hu
                cH, cW = 2 * cH, 2 * cW
        assert not skips and (cH, cW) == (H, W)
        nout = pool.get(BF * H * W * ch)
        self._op_gn(plan, "conv_norm_out", h, None, ch, ch, BF, H * W, nout, 1e-5, True, "conv_norm_out")
        pool.put(h)
        self._op_conv(plan, "conv_out", nout, None, P["conv_out.weight"], P["conv_out.bias"], out, BF, H, W, ch, 0, self.out_ch, 3, 1, 1,
                      os_=(self.out_ch * H * W, W, 1, H * W), kind="conv_out")
        pool.put(nout)
        self._finish_plan(plan)
        return plan

    # ------------------------------------------------------------------------------------------
    def get_plan(self, B, Fr, H, W):
        key = (B, Fr, H, W)
        plan = self._plans.get(key)
        if plan is None:
            with self._lock:
                plan = self._plans.get(key)
                if plan is None:
                    plan = self.build_plan(B, Fr, H, W)
                    self._plans[key] = plan
        return plan

    def load_inputs(self, plan, sample, timestep, encoder_hidden_states, added_time_ids):
        si = plan.static_in
        si["sample"].copy_(sample)
        B = si["sample"].shape[0]
        if torch.is_tensor(timestep):
            si["timestep"].copy_(timestep.reshape(-1).to(torch.float32).expand(B), non_blocking=True)
        else:
            si["timestep"].fill_(float(timestep))
        si["encoder_hidden_states"].copy_(encoder_hidden_states)
        si["added_time_ids"].copy_(added_time_ids.reshape(-1).to(torch.float32))

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids):
        """Eager (no graph) execution on the current stream. sample [B, F, C, H, W] -> fresh [B, F, C_out, H, W]."""
        B, Fr, _, H, W = sample.shape
        if encoder_hidden_states.shape[1] != 1:
            raise UnsupportedUNet("the spatio-temporal plan takes one context token per video (CLIP image embedding)")
        plan = self.get_plan(B, Fr, H, W)
        self.sync_packed()  # pipe-4 launches read packed copies: follow the live parameters' version counters
        self.load_inputs(plan, sample, timestep, encoder_hidden_states, added_time_ids)
        plan.run(self.host.stream_ptr(self.device))
        return plan.static_out.clone()
