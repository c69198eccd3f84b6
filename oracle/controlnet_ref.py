"""CPU fp32 restatement of diffusers `ControlNetModel` (the reference hands
`pipe.controlnet` to `compile_unet`, /root/reference/src/sfast/compilers/diffusion_pipeline_compiler.py:89-90).

TEST INFRASTRUCTURE ONLY -- imported by tests/, never by the product path.

diffusers is not installable here; the architecture is restated from public knowledge of
`diffusers.models.controlnet.ControlNetModel` (state-dict names kept): the UNet's conv_in / time embedding / down blocks
/ mid block (shared classes, oracle/unet_ref.py), plus
    controlnet_cond_embedding  conv_in(3->16) SiLU, [conv(c_i->c_i) SiLU, conv(c_i->c_{i+1}, stride 2) SiLU] over
                               (16, 32, 96, 256), conv_out(256->320): added to conv_in(sample)
    controlnet_down_blocks     one 1x1 "zero conv" per skip tensor;  controlnet_mid_block  1x1 on the mid-block output
forward(...) -> (down_block_res_samples, mid_block_res_sample), each scaled by conditioning_scale.
Pin: the SD1.5 ControlNet has 361,279,120 parameters; tests/test_oracle.py checks the restatement reproduces it.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet_ref import SD15_CONFIG, DownBlock, MidBlock, TimestepEmbedding, _per_block, timestep_embedding, tiny_config as _unet_tiny

SD15_CONTROLNET_PARAMS = 361_279_120


def tiny_config(**over):
    cfg = _unet_tiny()
    cfg.update(conditioning_embedding_out_channels=(8, 16, 32), conditioning_channels=3)
    cfg.update(over)
    return cfg


class ControlNetConditioningEmbedding(nn.Module):
    def __init__(self, out_channels, cond_channels=3, block_out_channels=(16, 32, 96, 256)):
        super().__init__()
        b = tuple(block_out_channels)
        self.conv_in = nn.Conv2d(cond_channels, b[0], 3, padding=1)
        blocks = []
        for i in range(len(b) - 1):
            blocks.append(nn.Conv2d(b[i], b[i], 3, padding=1))
            blocks.append(nn.Conv2d(b[i], b[i + 1], 3, padding=1, stride=2))
        self.blocks = nn.ModuleList(blocks)
        self.conv_out = nn.Conv2d(b[-1], out_channels, 3, padding=1)

    def forward(self, x):
        h = F.silu(self.conv_in(x))
        for blk in self.blocks:
            h = F.silu(blk(h))
        return self.conv_out(h)


class ControlNetModel(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        full = dict(SD15_CONFIG)
        full.update(conditioning_embedding_out_channels=(16, 32, 96, 256), conditioning_channels=3)
        full.update(cfg)
        full.pop("up_block_types", None)
        self.config = SimpleNamespace(**full)
        c = self.config
        boc = tuple(c.block_out_channels)
        n = len(boc)
        heads = _per_block(c.attention_head_dim, n)
        depth = _per_block(c.transformer_layers_per_block, n)
        temb = boc[0] * 4
        g, eps, lp, L = c.norm_num_groups, c.norm_eps, c.use_linear_projection, c.layers_per_block
        self.conv_in = nn.Conv2d(c.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        if getattr(c, "addition_embed_type", None) == "text_time":
            # SDXL ControlNets (diffusers ControlNetModel.__init__): the same pooled-text + micro-conditioning embedding as the SDXL UNet
            self.add_embedding = TimestepEmbedding(c.projection_class_embeddings_input_dim, temb)
        self.controlnet_cond_embedding = ControlNetConditioningEmbedding(boc[0], c.conditioning_channels,
                                                                         c.conditioning_embedding_out_channels)
        self.down_blocks = nn.ModuleList()
        zero = [nn.Conv2d(boc[0], boc[0], 1)]
        ch = boc[0]
        for i, t in enumerate(c.down_block_types):
            self.down_blocks.append(DownBlock(ch, boc[i], temb, L, g, eps, t == "CrossAttnDownBlock2D", heads[i],
                                              c.cross_attention_dim, depth[i], lp, i < n - 1))
            ch = boc[i]
            zero += [nn.Conv2d(ch, ch, 1) for _ in range(L)]
            if i < n - 1:
                zero.append(nn.Conv2d(ch, ch, 1))
        self.controlnet_down_blocks = nn.ModuleList(zero)
        self.mid_block = MidBlock(boc[-1], temb, g, eps, heads[-1], c.cross_attention_dim, depth[-1], lp)
        self.controlnet_mid_block = nn.Conv2d(boc[-1], boc[-1], 1)

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0, guess_mode=False,
                added_cond_kwargs=None, return_dict=True, **_):
        c = self.config
        B = sample.shape[0]
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float32, device=sample.device)
        t = t.to(sample.device).reshape(-1).expand(B)
        emb = self.time_embedding(timestep_embedding(t, c.block_out_channels[0], c.flip_sin_to_cos, c.freq_shift).to(sample.dtype))
        if getattr(c, "addition_embed_type", None) == "text_time":
            if not added_cond_kwargs or "text_embeds" not in added_cond_kwargs or "time_ids" not in added_cond_kwargs:
                raise ValueError("addition_embed_type 'text_time' requires text_embeds and time_ids in added_cond_kwargs")
            te, tid = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
            tide = timestep_embedding(tid.flatten(), c.addition_time_embed_dim, c.flip_sin_to_cos, c.freq_shift)
            emb = emb + self.add_embedding(torch.cat([te, tide.reshape(B, -1).to(te.dtype)], dim=-1).to(emb.dtype))
        h = self.conv_in(sample) + self.controlnet_cond_embedding(controlnet_cond)
        skips = [h]
        for blk in self.down_blocks:
            h = blk(h, emb, encoder_hidden_states, skips)
        h = self.mid_block(h, emb, encoder_hidden_states)
        down = [z(s) for s, z in zip(skips, self.controlnet_down_blocks)]
        mid = self.controlnet_mid_block(h)
        if guess_mode and not getattr(c, "global_pool_conditions", False):
            # diffusers ControlNetModel.forward ("6. scaling"): guess mode weights the residuals 0.1 ... 1.0 from the shallowest skip
            # to the mid block -- torch.logspace(-1, 0, len(down) + 1) * conditioning_scale
            scales = torch.logspace(-1, 0, len(down) + 1, device=sample.device) * conditioning_scale
            down = [d * s for d, s in zip(down, scales)]
            mid = mid * scales[-1]
        else:
            down = [d * conditioning_scale for d in down]
            mid = mid * conditioning_scale
        if not return_dict:
            return down, mid
        return SimpleNamespace(down_block_res_samples=down, mid_block_res_sample=mid)


def param_count(m):
    return sum(p.numel() for p in m.parameters())


def build(config="sd15", seed=0, dtype=torch.float32, device="cpu", **over):
    """Seeded default-initialised model. The real ControlNet zero-initialises its 1x1 output convs and the conditioning
    embedding's conv_out; random values are kept here so that parity tests exercise them."""
    cfg = dict(config) if isinstance(config, dict) else (tiny_config() if config == "tiny" else {})
    cfg.update(over)
    torch.manual_seed(seed)
    m = ControlNetModel(**cfg)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith(".bias") or "norm" in name:
                p.add_(0.05 * torch.randn_like(p))
    m = m.to(device=device, dtype=dtype).eval()
    for p in m.parameters():
        p.requires_grad_(False)
    return m
