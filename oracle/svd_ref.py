"""ORACLE -- test infrastructure only (imported by tests/ and the cpu_baseline leg of bench.py; never by the product package).

Plain-PyTorch restatement of diffusers' `UNetSpatioTemporalConditionModel` (Stable Video Diffusion / SVD-XT), with diffusers' module
names and state-dict keys -- BASELINE.json configs[4] ("StableVideoDiffusion-XT 576x1024, 25 frames, temporal-attn path"), the model
the reference accelerates in /root/reference/examples/optimize_stable_video_diffusion_pipeline.py (it traces whatever module it is
given; the architecture itself lives in diffusers, which is neither vendored nor installable here).

PARITY UNPINNED for the topology (public diffusers semantics restated from knowledge of the library, as for oracle/unet_ref.py).
Pins: (a) the parameter count of the restated SVD UNet, 1,524,623,082, equals the published size of the SVD / SVD-XT UNet
(`tests/test_oracle.py`); (b) every leaf op is an ATen op the reference's operator tests compare against.

Semantics restated (diffusers >= 0.24 line):
  sample [B, F, 8, H, W] -> flatten to [B*F, ...]; emb = time_embedding(sinusoid(t)) + add_embedding(sinusoid(added_time_ids).flatten);
  emb and encoder_hidden_states [B, 1, 1024] are repeated per frame.
  SpatioTemporalResBlock   = ResnetBlock2D (per frame, eps 1e-6) -> TemporalResnetBlock (GroupNorm over (C/G, F, H, W) per video,
                             Conv3d (3,1,1) over frames, eps 1e-5) -> AlphaBlender(learned_with_images, switch_spatial_to_temporal_mix):
                             out = (1 - s) * x_spatial + s * x_temporal,  s = sigmoid(mix_factor)   (image_only_indicator = 0)
  TransformerSpatioTemporalModel = GroupNorm(eps 1e-6) -> proj_in (Linear) -> per layer [BasicTransformerBlock (spatial tokens of a frame),
                             + frame-position embedding, TemporalBasicTransformerBlock (sequences of F frames at every spatial site,
                             cross-attention to the FIRST frame's context), AlphaBlender: s * spatial + (1 - s) * temporal] -> proj_out + residual
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet_ref import Attention, BasicTransformerBlock, Downsample2D, FeedForward, ResnetBlock2D, TimestepEmbedding, Upsample2D, timestep_embedding

SVD_CONFIG = dict(
    sample_size=96, in_channels=8, out_channels=4,
    down_block_types=("CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal",
                      "DownBlockSpatioTemporal"),
    up_block_types=("UpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal",
                    "CrossAttnUpBlockSpatioTemporal"),
    block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256, projection_class_embeddings_input_dim=768,
    layers_per_block=2, cross_attention_dim=1024, transformer_layers_per_block=1, num_attention_heads=(5, 10, 20, 20), num_frames=25,
)


def tiny_svd_config(**over):
    cfg = dict(SVD_CONFIG)
    cfg.update(sample_size=16, block_out_channels=(64, 128, 128), layers_per_block=1, cross_attention_dim=48,
               down_block_types=("CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal"),
               up_block_types=("UpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal"),
               num_attention_heads=(2, 4, 4), addition_time_embed_dim=32, projection_class_embeddings_input_dim=96, num_frames=5,
               norm_num_groups=8)
    cfg.update(over)
    return cfg


class AlphaBlender(nn.Module):
    """merge_strategy 'learned_with_images' with image_only_indicator == 0 everywhere: alpha = sigmoid(mix_factor)."""

    def __init__(self, alpha, switch_spatial_to_temporal_mix=False):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([float(alpha)]))
        self.switch = switch_spatial_to_temporal_mix

    def forward(self, x_spatial, x_temporal):
        a = torch.sigmoid(self.mix_factor).to(x_spatial.dtype)
        if self.switch:
            a = 1.0 - a
        return a * x_spatial + (1.0 - a) * x_temporal


class TemporalResnetBlock(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv3d(cin, cout, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv3d(cout, cout, (3, 1, 1), padding=(1, 0, 0))
        self.conv_shortcut = nn.Conv3d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):  # x [B, C, F, H, W], temb [B, F, T]
        h = self.conv1(F.silu(self.norm1(x)))
        t = self.time_emb_proj(F.silu(temb))[:, :, :, None, None].permute(0, 2, 1, 3, 4)
        h = h + t
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps=1e-6, temporal_eps=1e-5, merge_factor=0.5):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(cin, cout, temb_dim, groups, eps)
        self.temporal_res_block = TemporalResnetBlock(cout, cout, temb_dim, groups, temporal_eps)
        self.time_mixer = AlphaBlender(merge_factor, switch_spatial_to_temporal_mix=True)

    def forward(self, x, temb, num_frames):
        h = self.spatial_res_block(x, temb)
        BF, C, H, W = h.shape
        B = BF // num_frames
        h5 = h.reshape(B, num_frames, C, H, W).permute(0, 2, 1, 3, 4)
        ht = self.temporal_res_block(h5, temb.reshape(B, num_frames, -1))
        h5 = self.time_mixer(h5, ht)
        return h5.permute(0, 2, 1, 3, 4).reshape(BF, C, H, W)


class TemporalBasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, ctx_dim):
        super().__init__()
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = FeedForward(dim)
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, ctx_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, num_frames, time_context):  # x [B*F, S, C]; time_context [B*S, 1, ctx]
        BF, S, C = x.shape
        B = BF // num_frames
        h = x.reshape(B, num_frames, S, C).permute(0, 2, 1, 3).reshape(B * S, num_frames, C)
        res = h
        h = self.ff_in(self.norm_in(h)) + res      # is_res: dim == time_mix_inner_dim
        h = self.attn1(self.norm1(h)) + h
        h = self.attn2(self.norm2(h), time_context) + h
        h = self.ff(self.norm3(h)) + h
        return h.reshape(B, S, num_frames, C).permute(0, 2, 1, 3).reshape(BF, S, C)


class TransformerSpatioTemporalModel(nn.Module):
    def __init__(self, dim, heads, ctx_dim, depth, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, ctx_dim) for _ in range(depth)])
        self.temporal_transformer_blocks = nn.ModuleList([TemporalBasicTransformerBlock(dim, heads, ctx_dim) for _ in range(depth)])
        self.time_pos_embed = TimestepEmbedding(dim, dim * 4)
        self.time_pos_embed.linear_2 = nn.Linear(dim * 4, dim)  # TimestepEmbedding(in_channels, 4 * in_channels, out_dim=in_channels)
        self.time_mixer = AlphaBlender(0.5)
        self.proj_out = nn.Linear(dim, dim)
        self.dim = dim

    def forward(self, x, ctx, num_frames):  # x [B*F, C, H, W], ctx [B*F, 1, ctx_dim]
        BF, C, H, W = x.shape
        B = BF // num_frames
        S = H * W
        time_context = ctx.reshape(B, num_frames, -1, ctx.shape[-1])[:, 0]                     # first frame's context [B, 1, ctx]
        time_context = time_context[:, None].expand(B, S, *time_context.shape[1:]).reshape(B * S, -1, ctx.shape[-1])
        res = x
        h = self.norm(x).permute(0, 2, 3, 1).reshape(BF, S, C)
        h = self.proj_in(h)
        frames = torch.arange(num_frames, device=x.device).repeat(B)
        emb = self.time_pos_embed(timestep_embedding(frames, self.dim, True, 0.0).to(h.dtype))[:, None, :]
        for blk, tblk in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            h = blk(h, ctx)
            hm = tblk(h + emb, num_frames, time_context)
            h = self.time_mixer(h, hm)
        h = self.proj_out(h).reshape(BF, H, W, C).permute(0, 3, 1, 2)
        return h + res


class UNetSpatioTemporalConditionModel(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        full = dict(SVD_CONFIG)
        full.setdefault("norm_num_groups", 32)
        full.update(cfg)
        full.setdefault("norm_num_groups", 32)
        self.config = SimpleNamespace(**full)
        c = self.config
        boc = tuple(c.block_out_channels)
        n = len(boc)
        heads = tuple(c.num_attention_heads) if isinstance(c.num_attention_heads, (tuple, list)) else (c.num_attention_heads,) * n
        depth = (c.transformer_layers_per_block,) * n if isinstance(c.transformer_layers_per_block, int) else tuple(c.transformer_layers_per_block)
        T = boc[0] * 4
        g, L = c.norm_num_groups, c.layers_per_block
        self.conv_in = nn.Conv2d(c.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], T)
        self.add_embedding = TimestepEmbedding(c.projection_class_embeddings_input_dim, T)
        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i, t in enumerate(c.down_block_types):
            blk = nn.Module()
            blk.resnets = nn.ModuleList([SpatioTemporalResBlock(ch if j == 0 else boc[i], boc[i], T, g) for j in range(L)])
            blk.has_attn = t == "CrossAttnDownBlockSpatioTemporal"
            if blk.has_attn:
                blk.attentions = nn.ModuleList([TransformerSpatioTemporalModel(boc[i], heads[i], c.cross_attention_dim, depth[i], g) for _ in range(L)])
            blk.has_down = i < n - 1
            if blk.has_down:
                blk.downsamplers = nn.ModuleList([Downsample2D(boc[i])])
            self.down_blocks.append(blk)
            ch = boc[i]
        mid = nn.Module()
        mid.resnets = nn.ModuleList([SpatioTemporalResBlock(boc[-1], boc[-1], T, g) for _ in range(2)])
        mid.attentions = nn.ModuleList([TransformerSpatioTemporalModel(boc[-1], heads[-1], c.cross_attention_dim, depth[-1], g)])
        self.mid_block = mid
        self.up_blocks = nn.ModuleList()
        rev, rheads, rdepth = boc[::-1], heads[::-1], depth[::-1]
        prev = rev[0]
        for i, t in enumerate(c.up_block_types):
            out_c, in_c = rev[i], rev[min(i + 1, n - 1)]
            blk = nn.Module()
            cins = [(prev if j == 0 else out_c) + (in_c if j == L else out_c) for j in range(L + 1)]
            blk.resnets = nn.ModuleList([SpatioTemporalResBlock(ci, out_c, T, g) for ci in cins])
            blk.has_attn = t == "CrossAttnUpBlockSpatioTemporal"
            if blk.has_attn:
                blk.attentions = nn.ModuleList([TransformerSpatioTemporalModel(out_c, rheads[i], c.cross_attention_dim, rdepth[i], g) for _ in cins])
            blk.has_up = i < n - 1
            if blk.has_up:
                blk.upsamplers = nn.ModuleList([Upsample2D(out_c)])
            self.up_blocks.append(blk)
            prev = out_c
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], c.out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, return_dict=True):
        c = self.config
        B, Fr = sample.shape[:2]
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float32, device=sample.device)
        t = t.to(sample.device).reshape(-1).expand(B)
        emb = self.time_embedding(timestep_embedding(t, c.block_out_channels[0], True, 0.0).to(sample.dtype))
        te = timestep_embedding(added_time_ids.flatten(), c.addition_time_embed_dim, True, 0.0).reshape(B, -1).to(emb.dtype)
        emb = emb + self.add_embedding(te)
        x = sample.flatten(0, 1)
        emb = emb.repeat_interleave(Fr, dim=0)
        ctx = encoder_hidden_states.repeat_interleave(Fr, dim=0)
        h = self.conv_in(x)
        skips = [h]
        for blk in self.down_blocks:
            for j, r in enumerate(blk.resnets):
                h = r(h, emb, Fr)
                if blk.has_attn:
                    h = blk.attentions[j](h, ctx, Fr)
                skips.append(h)
            if blk.has_down:
                h = blk.downsamplers[0](h)
                skips.append(h)
        h = self.mid_block.resnets[0](h, emb, Fr)
        h = self.mid_block.attentions[0](h, ctx, Fr)
        h = self.mid_block.resnets[1](h, emb, Fr)
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                h = r(torch.cat([h, skips.pop()], dim=1), emb, Fr)
                if blk.has_attn:
                    h = blk.attentions[j](h, ctx, Fr)
            if blk.has_up:
                h = blk.upsamplers[0](h)
        out = self.conv_out(F.silu(self.conv_norm_out(h)))
        out = out.reshape(B, Fr, *out.shape[1:])
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)


def build(config="svd", seed=0, dtype=torch.float32, device="cpu", **over):
    cfg = dict({"svd": SVD_CONFIG, "tiny": tiny_svd_config()}[config]) if isinstance(config, str) else dict(config)
    cfg.update(over)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        m = UNetSpatioTemporalConditionModel(**cfg)
        for name, p in m.named_parameters():
            if name.endswith("mix_factor"):
                p.copy_(torch.randn(p.shape, generator=g))  # sigmoid -> blends well away from 0 / 1: both branches matter
            elif p.ndim == 1:
                if "norm" in name and name.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / math.sqrt(fan_in)))
    return m.to(device=device, dtype=dtype).eval()


def param_count(m):
    return sum(p.numel() for p in m.parameters())
