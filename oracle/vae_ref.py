"""CPU fp32 restatement of the diffusers `AutoencoderKL` DECODER (`vae.decoder`), the module the reference's
`compile_vae` optimises next to the UNet (/root/reference/src/sfast/compilers/diffusion_pipeline_compiler.py:154-190;
SURVEY.md section 8f rank 1).

TEST INFRASTRUCTURE ONLY -- imported by tests/, never by the product path.

diffusers is not installable here, so the architecture is restated from public knowledge of
`diffusers.models.autoencoders.vae.Decoder` (state-dict names kept so a real checkpoint loads unchanged):

    conv_in(4 -> 512) -> mid_block[ResnetBlock2D, Attention(1 head, dim 512, GroupNorm eps 1e-6, residual), ResnetBlock2D]
    -> 4 x UpDecoderBlock2D over reversed channels (512, 512, 256, 128): 3 ResnetBlock2D (eps 1e-6, no time embedding,
       1x1 conv_shortcut when channels change) + nearest-2x Upsample2D conv (all but the last block)
    -> GroupNorm(32, eps 1e-6) -> SiLU -> conv_out(128 -> 3)

Pin: the SD1.x / SD2.x / SDXL VAE decoder has 49,490,179 parameters (AutoencoderKL total 83,653,863 = encoder 34,163,592 +
decoder 49,490,179 + quant_conv 72 + post_quant_conv 20); `tests/test_oracle.py` checks the restatement reproduces it.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

SD_VAE_DECODER_CONFIG = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                             norm_num_groups=32)
SD_VAE_DECODER_PARAMS = 49_490_179


def tiny_config(**over):
    cfg = dict(latent_channels=4, out_channels=3, block_out_channels=(32, 64), layers_per_block=1, norm_num_groups=8)
    cfg.update(over)
    return cfg


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, groups, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class Attention(nn.Module):
    """Single-head spatial self-attention of the VAE mid block (diffusers Attention with
    `_from_deprecated_attn_block`: GroupNorm, biased q/k/v/out projections, residual connection)."""

    def __init__(self, dim, groups, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, dim, eps=eps)
        self.to_q = nn.Linear(dim, dim)
        self.to_k = nn.Linear(dim, dim)
        self.to_v = nn.Linear(dim, dim)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Identity()])
        self.scale = dim ** -0.5

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x.reshape(B, C, H * W)).transpose(1, 2)  # [B, HW, C]
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        p = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * self.scale, dim=-1)
        o = self.to_out[0](torch.bmm(p, v))
        return o.transpose(1, 2).reshape(B, C, H, W) + x


class MidBlock(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, groups), ResnetBlock2D(c, c, groups)])
        self.attentions = nn.ModuleList([Attention(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, layers, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, groups) for j in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Decoder(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        c = SimpleNamespace(**cfg)
        self.config = c
        boc = tuple(c.block_out_channels)
        g = c.norm_num_groups
        self.conv_in = nn.Conv2d(c.latent_channels, boc[-1], 3, padding=1)
        self.mid_block = MidBlock(boc[-1], g)
        rev = boc[::-1]
        blocks, prev = [], rev[0]
        for i, ch in enumerate(rev):
            blocks.append(UpDecoderBlock2D(prev, ch, c.layers_per_block + 1, g, up=i != len(rev) - 1))
            prev = ch
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], c.out_channels, 3, padding=1)

    def forward(self, z):
        h = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            h = b(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class Downsample2D(nn.Module):
    """diffusers Downsample2D with padding=0: F.pad(x, (0, 1, 0, 1)) then a stride-2 3x3 conv."""

    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, layers, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, groups) for j in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class Encoder(nn.Module):
    """`AutoencoderKL.encoder`: conv_in(3 -> 128) -> 4 x DownEncoderBlock2D (2 resnets, stride-2 conv on all but the last)
    -> mid block -> GroupNorm -> SiLU -> conv_out(512 -> 2 * latent_channels). 34,163,592 parameters for the SD VAE."""

    def __init__(self, **cfg):
        super().__init__()
        c = SimpleNamespace(**cfg)
        self.config = c
        boc = tuple(c.block_out_channels)
        g = c.norm_num_groups
        self.conv_in = nn.Conv2d(c.out_channels, boc[0], 3, padding=1)  # image channels = the decoder's out_channels
        blocks, prev = [], boc[0]
        for i, ch in enumerate(boc):
            blocks.append(DownEncoderBlock2D(prev, ch, c.layers_per_block, g, down=i != len(boc) - 1))
            prev = ch
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = MidBlock(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * c.latent_channels, 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for b in self.down_blocks:
            h = b(h)
        h = self.mid_block(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


SD_VAE_ENCODER_PARAMS = 34_163_592


def param_count(m):
    return sum(p.numel() for p in m.parameters())


def build_encoder(config="sd", seed=0, dtype=torch.float32, device="cpu", **over):
    cfg = dict(SD_VAE_DECODER_CONFIG if config == "sd" else tiny_config())
    cfg.update(over)
    torch.manual_seed(seed)
    m = Encoder(**cfg).to(device=device, dtype=dtype).eval()
    for p in m.parameters():
        p.requires_grad_(False)
    return m


def build(config="sd", seed=0, dtype=torch.float32, device="cpu", **over):
    cfg = dict(SD_VAE_DECODER_CONFIG if config == "sd" else tiny_config())
    cfg.update(over)
    torch.manual_seed(seed)
    m = Decoder(**cfg).to(device=device, dtype=dtype).eval()
    for p in m.parameters():
        p.requires_grad_(False)
    return m


__all__ = ["Decoder", "Encoder", "build", "build_encoder", "SD_VAE_ENCODER_PARAMS", "tiny_config", "param_count", "SD_VAE_DECODER_CONFIG", "SD_VAE_DECODER_PARAMS", "math"]
