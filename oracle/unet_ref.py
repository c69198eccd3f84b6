"""ORACLE -- test infrastructure only (imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package).

Plain-PyTorch restatement of diffusers' `UNet2DConditionModel` for the SD1.5 / SD2.1 / SDXL
family, with diffusers-compatible module names and state-dict keys so real checkpoints drop in.

PARITY UNPINNED for the whole-UNet topology: diffusers is neither vendored in the reference
(`/root/reference/setup.py:246-249` only pins `diffusers>=0.19.0`) nor installable here, and the
reference's own integration test prints the image instead of comparing numbers
(`/root/reference/tests/compilers/test_stable_diffusion_pipeline_compiler.py:435-436`). The
restatement follows the published diffusers semantics recorded in SURVEY.md Appendix A and is
pinned by (a) the public parameter counts -- 859,520,964 for SD1.5 and 2,567,463,684 for
SDXL-base, asserted in `tests/test_oracle.py` -- and (b) per-op equivalence with the ATen ops the
reference's operator tests compare against (F.group_norm, F.layer_norm, F.conv2d,
F.scaled_dot_product_attention, chunk/gelu GEGLU; `/root/reference/tests/operators/*.py`).

The per-module arithmetic follows what the reference's fused ops must reproduce:
  GroupNorm(+SiLU)  /root/reference/src/sfast/triton/torch_ops.py:179-189 (= native_group_norm + silu)
  LayerNorm         /root/reference/src/sfast/triton/ops/layer_norm.py:52-133
  GEGLU             /root/reference/src/sfast/jit/passes/__init__.py:643-649 (linear -> chunk -> h*gelu(g))
  conv+bias(+add)   /root/reference/src/sfast/csrc/operators/cudnn/cudnn_convolution_impl.cc:995-998
  attention         /root/reference/src/sfast/libs/xformers/xformers_attention.py:26-43, [B,S,H,D] layout
                    /root/reference/src/sfast/libs/diffusers/xformers_attention.py:66-69
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

SD15_CONFIG = dict(
    sample_size=64, in_channels=4, out_channels=4,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, cross_attention_dim=768,
    attention_head_dim=8, transformer_layers_per_block=1, norm_num_groups=32, norm_eps=1e-5,
    use_linear_projection=False, flip_sin_to_cos=True, freq_shift=0,
    addition_embed_type=None, addition_time_embed_dim=None, projection_class_embeddings_input_dim=None,
    time_cond_proj_dim=None, class_embed_type=None,
)

SDXL_CONFIG = dict(
    sample_size=128, in_channels=4, out_channels=4,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    block_out_channels=(320, 640, 1280), layers_per_block=2, cross_attention_dim=2048,
    attention_head_dim=(5, 10, 20), transformer_layers_per_block=(1, 2, 10), norm_num_groups=32,
    norm_eps=1e-5, use_linear_projection=True, flip_sin_to_cos=True, freq_shift=0,
    addition_embed_type="text_time", addition_time_embed_dim=256,
    projection_class_embeddings_input_dim=2816,
)


def tiny_config(**over):
    """A small config with the SD1.5 topology (cross-attn down/up blocks, odd skip widths)."""
    cfg = dict(SD15_CONFIG)
    cfg.update(sample_size=16, block_out_channels=(64, 128, 128), layers_per_block=1,
               down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
               up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
               cross_attention_dim=64, attention_head_dim=2, norm_num_groups=8)
    cfg.update(over)
    return cfg


def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000.0):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device)
    exponent = exponent / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    """diffusers TimestepEmbedding; `cond_proj_dim` (UNet config `time_cond_proj_dim`, the LCM guidance-scale embedding w):
    a bias-free Linear whose output is added to the sinusoid BEFORE the MLP (`sample = sample + cond_proj(condition)`)."""

    def __init__(self, cin, dim, cond_proj_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)
        self.cond_proj = nn.Linear(cond_proj_dim, cin, bias=False) if cond_proj_dim is not None else None

    def forward(self, x, condition=None):
        if condition is not None:
            x = x + self.cond_proj(condition)
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class LoRALinearLayer(nn.Module):
    """diffusers models/lora.py `LoRALinearLayer`: up(down(x)), times network_alpha / rank when network_alpha is set."""

    def __init__(self, in_features, out_features, rank=4, network_alpha=None):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False)
        self.up = nn.Linear(rank, out_features, bias=False)
        self.network_alpha, self.rank = network_alpha, rank

    def forward(self, x):
        y = self.up(self.down(x))
        return y * (self.network_alpha / self.rank) if self.network_alpha is not None else y


class LoRACompatibleLinear(nn.Linear):
    """diffusers models/lora.py `LoRACompatibleLinear` (0.21 - 0.24: what `unet.load_attn_procs(lora)` leaves on to_q / to_k / to_v /
    to_out.0): linear(x) + scale * lora_layer(x); `scale` is the call's cross_attention_kwargs["scale"] (read from a box shared by
    the UNet's LoRA layers). State-dict keys: `<linear>.lora_layer.down.weight`, `<linear>.lora_layer.up.weight`."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.lora_layer = None
        self._scale_box = [1.0]

    def forward(self, x):
        y = super().forward(x)
        if self.lora_layer is not None:
            y = y + self._scale_box[0] * self.lora_layer(x)
        return y


class PeftLoraLinear(nn.Module):
    """The attribute layout of `peft.tuners.lora.Linear` (diffusers >= 0.25 with the peft backend): base_layer, lora_A / lora_B
    ModuleDicts keyed by adapter name, scaling[adapter] = lora_alpha / r, active_adapters, merged, disable_adapters. State-dict keys:
    `<linear>.base_layer.weight`, `<linear>.lora_A.<adapter>.weight` [r, in], `<linear>.lora_B.<adapter>.weight` [out, r]. diffusers
    multiplies `scaling` by the call's cross_attention_kwargs["scale"] for the duration of the forward (scale_lora_layers)."""

    def __init__(self, base, rank, lora_alpha, adapter="default"):
        super().__init__()
        self.base_layer = base
        self.lora_A = nn.ModuleDict({adapter: nn.Linear(base.in_features, rank, bias=False)})
        self.lora_B = nn.ModuleDict({adapter: nn.Linear(rank, base.out_features, bias=False)})
        self.scaling = {adapter: lora_alpha / rank}
        self.r = {adapter: rank}
        self.active_adapters = [adapter]
        self.merged, self.disable_adapters = False, False
        self._scale_box = [1.0]

    def forward(self, x):
        y = self.base_layer(x)
        if not self.disable_adapters and not self.merged:
            for a in self.active_adapters:
                y = y + self.lora_B[a](self.lora_A[a](x)) * (self.scaling[a] * self._scale_box[0])
        return y


class IPAdapterAttnProcessor(nn.Module):
    """diffusers `IPAdapterAttnProcessor2_0` (models/attention_processor.py), the decoupled image cross-attention of IP-Adapter: per
    adapter i a key / value projection of the image tokens, `to_k_ip[i]` / `to_v_ip[i]` (Linear(cross_attention_dim -> hidden, bias =
    False); state-dict keys `...attn2.processor.to_k_ip.{i}.weight`), and a python float `scale[i]`. The processor computes
    `attn(q, k_text, v_text) + sum_i scale[i] * attn(q, to_k_ip[i](ip_i), to_v_ip[i](ip_i))` -- two SEPARATE softmaxes -- before `to_out`."""

    def __init__(self, hidden_size, cross_attention_dim, num_tokens=(4,), scale=1.0):
        super().__init__()
        self.num_tokens = tuple(num_tokens)
        self.scale = [float(scale)] * len(self.num_tokens)
        self.to_k_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in self.num_tokens])
        self.to_v_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in self.num_tokens])


class Attention(nn.Module):
    def __init__(self, dim, heads, ctx_dim=None):
        super().__init__()
        ctx_dim = ctx_dim or dim
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Identity()])
        self.processor = None  # an IPAdapterAttnProcessor on attn2 after load_ip_adapter()

    def forward(self, x, ctx=None, bias=None, ip=None):
        ctx = x if ctx is None else ctx
        B, S, C = x.shape
        H = self.heads
        q = self.to_q(x).view(B, S, H, C // H).transpose(1, 2)
        k = self.to_k(ctx).view(B, -1, H, C // H).transpose(1, 2)
        v = self.to_v(ctx).view(B, -1, H, C // H).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None if bias is None else bias[:, None].to(q.dtype))  # scale = head_dim ** -0.5
        if ip is not None and self.processor is not None:
            for i, ip_i in enumerate(ip):
                ik = self.processor.to_k_ip[i](ip_i).view(B, -1, H, C // H).transpose(1, 2)
                iv = self.processor.to_v_ip[i](ip_i).view(B, -1, H, C // H).transpose(1, 2)
                o = o + self.processor.scale[i] * F.scaled_dot_product_attention(q, ik, iv)
        return self.to_out[0](o.transpose(1, 2).reshape(B, S, C))


class ImageProjection(nn.Module):
    """diffusers `ImageProjection` (models/embeddings.py): image embedding [B, D_img] -> num_image_text_embeds tokens of width
    cross_attention_dim: Linear, reshape, LayerNorm. `load_ip_adapter` installs it as `unet.encoder_hid_proj`
    (`encoder_hid_dim_type = "ip_image_proj"`)."""

    def __init__(self, image_embed_dim, cross_attention_dim, num_image_text_embeds=4):
        super().__init__()
        self.num_image_text_embeds = num_image_text_embeds
        self.image_embeds = nn.Linear(image_embed_dim, num_image_text_embeds * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)

    def forward(self, image_embeds):
        B = image_embeds.shape[0]
        x = self.image_embeds(image_embeds.to(self.image_embeds.weight.dtype))
        return self.norm(x.reshape(B, self.num_image_text_embeds, -1))


class GEGLU(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.proj = nn.Linear(din, dout * 2)

    def forward(self, x):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class ImageResampler(nn.Module):
    """Stand-in for diffusers `IPAdapterPlusImageProjection` (the perceiver resampler of IP-Adapter Plus): learned latents
    cross-attend to the patch embeddings of the image encoder, then a projection and a LayerNorm -- [N, patches, D] -> [N, T, ctx].
    One attention layer instead of four: what matters here is that it is NOT Linear + LayerNorm, so the engine cannot run it."""

    def __init__(self, embed_dim, cross_attention_dim, num_queries=6, heads=2):
        super().__init__()
        self.latents = nn.Parameter(torch.randn(1, num_queries, embed_dim) / embed_dim ** 0.5)
        self.norm_in = nn.LayerNorm(embed_dim)
        self.attn = nn.MultiheadAttention(embed_dim, heads, batch_first=True)
        self.proj_out = nn.Linear(embed_dim, cross_attention_dim)
        self.norm_out = nn.LayerNorm(cross_attention_dim)

    def forward(self, x):
        lat = self.latents.expand(x.shape[0], -1, -1).to(x.dtype)
        xn = self.norm_in(x)
        lat = lat + self.attn(lat, xn, xn, need_weights=False)[0]
        return self.norm_out(self.proj_out(lat))


class MultiIPAdapterImageProjection(nn.Module):
    """diffusers `MultiIPAdapterImageProjection` (models/embeddings.py): one projection layer per adapter; forward takes the LIST of
    image embeddings ([B, images, ...] each), projects every image and returns one [B, images * T, ctx] tensor per adapter."""

    def __init__(self, layers):
        super().__init__()
        self.image_projection_layers = nn.ModuleList(layers)

    def forward(self, image_embeds):
        out = []
        for emb, layer in zip(image_embeds, self.image_projection_layers):
            b, n = emb.shape[:2]
            y = layer(emb.reshape((b * n,) + emb.shape[2:]))
            out.append(y.reshape((b, n * y.shape[1]) + y.shape[2:]))
        return out


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, ctx_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, ctx_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx):
        cross_bias, ip, self_bias = None, None, None
        if isinstance(ctx, tuple):  # (encoder_hidden_states, additive cross-attention bias [B, 1, S_ctx] or None[, IP-Adapter image tokens[, self-attention bias]])
            ctx, cross_bias, ip, self_bias = (tuple(ctx) + (None, None))[:4]
        if self_bias is not None and self_bias.shape[-1] != x.shape[1]:
            # diffusers Attention.prepare_attention_mask: a mask whose length differs from the layer's key count is padded by
            # F.pad(mask, (0, target_length)) -- to current + target keys, not to target -- and scaled_dot_product_attention then refuses
            # the shape. A UNet-level `attention_mask` is therefore usable only when every self-attention layer has exactly that many tokens.
            raise RuntimeError(f"attention_mask of {self_bias.shape[-1]} keys on a self-attention layer with {x.shape[1]} tokens "
                               "(diffusers pads it to the sum of both and scaled_dot_product_attention rejects the shape)")
        x = self.attn1(self.norm1(x), None, self_bias) + x
        x = self.attn2(self.norm2(x), ctx, cross_bias, ip) + x
        return self.ff(self.norm3(x)) + x


class Transformer2DModel(nn.Module):
    def __init__(self, dim, heads, ctx_dim, depth, groups, linear_proj):
        super().__init__()
        self.linear_proj = linear_proj
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        if linear_proj:
            self.proj_in = nn.Linear(dim, dim)
            self.proj_out = nn.Linear(dim, dim)
        else:
            self.proj_in = nn.Conv2d(dim, dim, 1)
            self.proj_out = nn.Conv2d(dim, dim, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(dim, heads, ctx_dim) for _ in range(depth)])

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        res = x
        h = self.norm(x)
        if self.linear_proj:
            h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
            h = self.proj_in(h)
        else:
            h = self.proj_in(h).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        if self.linear_proj:
            h = self.proj_out(h).reshape(B, H, W, C).permute(0, 3, 1, 2)
        else:
            h = self.proj_out(h.reshape(B, H, W, C).permute(0, 3, 1, 2))
        return h + res


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, layers, groups, eps, attn, heads, ctx_dim, depth, linear_proj, down):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups, eps) for i in range(layers)])
        if attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, heads, ctx_dim, depth, groups, linear_proj) for _ in range(layers)])
        self.has_attn = attn
        if down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        self.has_down = down

    def forward(self, h, temb, ctx, skips):
        for i, r in enumerate(self.resnets):
            h = r(h, temb)
            if self.has_attn:
                h = self.attentions[i](h, ctx)
            skips.append(h)
        if self.has_down:
            h = self.downsamplers[0](h)
            skips.append(h)
        return h


class MidBlock(nn.Module):
    def __init__(self, c, temb, groups, eps, heads, ctx_dim, depth, linear_proj):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, ctx_dim, depth, groups, linear_proj)])

    def forward(self, h, temb, ctx):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, ctx)
        return self.resnets[1](h, temb)


class UpBlock(nn.Module):
    def __init__(self, cin_list, cout, temb, groups, eps, attn, heads, ctx_dim, depth, linear_proj, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ci, cout, temb, groups, eps) for ci in cin_list])
        if attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, heads, ctx_dim, depth, groups, linear_proj) for _ in cin_list])
        self.has_attn = attn
        if up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.has_up = up

    def forward(self, h, temb, ctx, skips):
        for i, r in enumerate(self.resnets):
            h = r(torch.cat([h, skips.pop()], dim=1), temb)
            if self.has_attn:
                h = self.attentions[i](h, ctx)
        if self.has_up:
            h = self.upsamplers[0](h)
        return h


def _per_block(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


class UNet2DConditionModel(nn.Module):
    """fp32 CPU oracle of the denoising UNet (and, on the GPU box, the eager-fp16 stand-in for
    'the reference diffusers fp16 UNet')."""

    def __init__(self, **cfg):
        super().__init__()
        full = dict(SD15_CONFIG)
        full.update(cfg)
        self.config = SimpleNamespace(**full)
        c = self.config
        boc = tuple(c.block_out_channels)
        n = len(boc)
        heads = _per_block(c.attention_head_dim, n)  # diffusers misnomer: number of heads
        depth = _per_block(c.transformer_layers_per_block, n)
        temb = boc[0] * 4
        g, eps, lp, L = c.norm_num_groups, c.norm_eps, c.use_linear_projection, c.layers_per_block
        self.conv_in = nn.Conv2d(c.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb, getattr(c, "time_cond_proj_dim", None))
        # class conditioning (diffusers UNet2DConditionModel._set_class_embedding): "timestep" embeds class_labels through the
        # sinusoid + an MLP, "projection" feeds a float vector of width projection_class_embeddings_input_dim to the MLP
        cet = getattr(c, "class_embed_type", None)
        if cet == "timestep":
            self.class_embedding = TimestepEmbedding(boc[0], temb)
        elif cet == "projection":
            self.class_embedding = TimestepEmbedding(c.projection_class_embeddings_input_dim, temb)
        elif cet is not None:
            raise NotImplementedError(f"class_embed_type {cet}")
        if c.addition_embed_type == "text_time":
            self.add_embedding = TimestepEmbedding(c.projection_class_embeddings_input_dim, temb)
        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i, t in enumerate(c.down_block_types):
            self.down_blocks.append(DownBlock(ch, boc[i], temb, L, g, eps, t == "CrossAttnDownBlock2D", heads[i],
                                              c.cross_attention_dim, depth[i], lp, i < n - 1))
            ch = boc[i]
        self.mid_block = MidBlock(boc[-1], temb, g, eps, heads[-1], c.cross_attention_dim, depth[-1], lp)
        self.up_blocks = nn.ModuleList()
        rev = boc[::-1]
        rheads, rdepth = heads[::-1], depth[::-1]
        prev = rev[0]
        for i, t in enumerate(c.up_block_types):
            out_c = rev[i]
            in_c = rev[min(i + 1, n - 1)]
            cins = []
            for j in range(L + 1):
                skip_c = in_c if j == L else out_c
                res_in = prev if j == 0 else out_c
                cins.append(res_in + skip_c)
            self.up_blocks.append(UpBlock(cins, out_c, temb, g, eps, t == "CrossAttnUpBlock2D", rheads[i],
                                          c.cross_attention_dim, rdepth[i], lp, i < n - 1))
            prev = out_c
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], c.out_channels, 3, padding=1)

    def load_ip_adapter_plus(self, image_embed_dim=32, num_tokens=6, scale=1.0, seed=0):
        """IP-Adapter Plus: as load_ip_adapter(), with a resampler over PATCH embeddings ([B, images, patches, D]) as the projection."""
        self.load_ip_adapter(image_embed_dim, num_tokens, scale, seed)
        torch.manual_seed(seed + 100)
        res = ImageResampler(image_embed_dim, self.config.cross_attention_dim, num_tokens).to(self.device, self.dtype)
        self.encoder_hid_proj = MultiIPAdapterImageProjection([res])
        for q in self.parameters():
            q.requires_grad_(False)
        return self

    def load_ip_adapter(self, image_embed_dim=32, num_tokens=4, scale=1.0, seed=0):
        """What diffusers' `pipe.load_ip_adapter(...)` does to the UNet (loaders/unet.py `_load_ip_adapter_weights`), with seeded random
        weights: `encoder_hid_proj` = ImageProjection, config.encoder_hid_dim_type = "ip_image_proj", and an IPAdapterAttnProcessor
        (to_k_ip / to_v_ip) on every cross-attention (`attn2`)."""
        g = torch.Generator().manual_seed(seed)
        ctx_dim = self.config.cross_attention_dim
        self.encoder_hid_proj = ImageProjection(image_embed_dim, ctx_dim, num_tokens).to(self.device, self.dtype)
        self.config.encoder_hid_dim_type = "ip_image_proj"
        for name, m in self.named_modules():
            if name.endswith(".attn2"):
                p = IPAdapterAttnProcessor(m.to_q.weight.shape[0], ctx_dim, (num_tokens,), scale)
                with torch.no_grad():
                    for q in p.parameters():
                        q.copy_(torch.randn(q.shape, generator=g) * q.shape[1] ** -0.5)
                m.processor = p.to(self.device, self.dtype)
        with torch.no_grad():
            for q in self.encoder_hid_proj.parameters():
                q.copy_((torch.randn(q.shape, generator=g) * (q.shape[-1] ** -0.5 if q.ndim > 1 else 0.1) + (1.0 if q.ndim == 1 and q is self.encoder_hid_proj.norm.weight else 0.0)).to(q.dtype))
        for q in self.parameters():
            q.requires_grad_(False)
        return self

    def load_lora(self, rank=4, network_alpha=None, seed=0, targets=("to_q", "to_k", "to_v", "to_out.0"), style="diffusers", up_scale=0.3):
        """What `unet.load_attn_procs(lora)` / `pipe.load_lora_weights(lora)` leave behind, UN-fused, with seeded random factors: every
        attention projection named in `targets` becomes a LoRACompatibleLinear with a lora_layer (style "diffusers") or a peft-style
        wrapper (style "peft"); the base weight / bias Parameters are the same objects as before."""
        g = torch.Generator().manual_seed(seed)
        box = self.__dict__.setdefault("_lora_scale_box", [1.0])
        for name, attn in [(n, m) for n, m in self.named_modules() if isinstance(m, Attention)]:
            for t in targets:
                holder, key = (attn.to_out, 0) if t == "to_out.0" else (attn, t)
                lin = holder[key] if isinstance(key, int) else getattr(holder, key)
                if style == "peft":
                    new = PeftLoraLinear(lin, rank, float(network_alpha if network_alpha is not None else rank))
                    fac = [new.lora_A["default"].weight, new.lora_B["default"].weight]
                else:
                    new = LoRACompatibleLinear(lin.in_features, lin.out_features, bias=lin.bias is not None)
                    new.weight, new.bias = lin.weight, lin.bias
                    new.lora_layer = LoRALinearLayer(lin.in_features, lin.out_features, rank, network_alpha)
                    fac = [new.lora_layer.down.weight, new.lora_layer.up.weight]
                new._scale_box = box
                new.to(lin.weight.device, lin.weight.dtype)
                with torch.no_grad():
                    fac[0].copy_(torch.randn(fac[0].shape, generator=g) * fac[0].shape[1] ** -0.5)
                    fac[1].copy_(torch.randn(fac[1].shape, generator=g) * up_scale)
                if isinstance(key, int):
                    holder[key] = new
                else:
                    setattr(holder, key, new)
        for q in self.parameters():
            q.requires_grad_(False)
        return self

    def set_ip_adapter_scale(self, scale):
        for m in self.modules():
            if isinstance(m, IPAdapterAttnProcessor):
                m.scale = [float(scale)] * len(m.scale)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, return_dict=True,
                down_block_additional_residuals=None, mid_block_additional_residual=None, encoder_attention_mask=None,
                timestep_cond=None, class_labels=None, cross_attention_kwargs=None, attention_mask=None, **_):
        # cross_attention_kwargs: only {"scale": s} is meaningful here: diffusers scales the LoRA layers by it for this call
        # (load_lora(); a no-op without them)
        assert not cross_attention_kwargs or set(cross_attention_kwargs) <= {"scale"}
        if "_lora_scale_box" in self.__dict__:
            self.__dict__["_lora_scale_box"][0] = float((cross_attention_kwargs or {}).get("scale", 1.0))
        c = self.config
        B = sample.shape[0]
        if encoder_attention_mask is not None:
            # diffusers UNet2DConditionModel.forward: keep-mask [B, S] -> additive bias (1 - mask) * -10000, unsqueezed to [B, 1, S];
            # applied to the cross-attention (attn2) of every transformer block
            m_ = encoder_attention_mask
            if m_.ndim == 2:
                m_ = ((1 - m_.to(sample.dtype)) * -10000.0).unsqueeze(1)
            encoder_hidden_states = (encoder_hidden_states, m_)
        if getattr(c, "encoder_hid_dim_type", None) == "ip_image_proj":
            # diffusers UNet2DConditionModel.process_encoder_hidden_states: the image embeddings of `added_cond_kwargs` pass through
            # encoder_hid_proj and travel beside the text context: encoder_hidden_states = (text, [image tokens per adapter])
            if not added_cond_kwargs or "image_embeds" not in added_cond_kwargs:
                raise ValueError("encoder_hid_dim_type 'ip_image_proj' requires `image_embeds` in added_cond_kwargs")
            ie = added_cond_kwargs["image_embeds"]
            ie = list(ie) if isinstance(ie, (list, tuple)) else [ie]
            ip = []
            if isinstance(self.encoder_hid_proj, MultiIPAdapterImageProjection):
                ip = [t_.to(sample.dtype) for t_ in self.encoder_hid_proj([t_.to(self.dtype) for t_ in ie])]
                ie = []
            for t_ in ie:  # MultiIPAdapterImageProjection.forward: [B, images, D] -> project every image -> [B, images * T, ctx]
                t_ = t_[:, None] if t_.ndim == 2 else t_
                b_, n_ = t_.shape[:2]
                ip.append(self.encoder_hid_proj(t_.reshape(b_ * n_, -1)).reshape(b_, -1, c.cross_attention_dim).to(sample.dtype))
            base = encoder_hidden_states if isinstance(encoder_hidden_states, tuple) else (encoder_hidden_states, None)
            encoder_hidden_states = (base[0], base[1], ip)
        if attention_mask is not None:
            # diffusers UNet2DConditionModel.forward: a keep-mask [B, L] -> additive bias (1 - mask) * -10000, unsqueezed to [B, 1, L], handed to
            # the SELF-attention (attn1) of every transformer block
            sb = ((1 - attention_mask.to(sample.dtype)) * -10000.0).unsqueeze(1)
            base = encoder_hidden_states if isinstance(encoder_hidden_states, tuple) else (encoder_hidden_states, None)
            encoder_hidden_states = (tuple(base) + (None, None))[:3] + (sb,)
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float32, device=sample.device)
        t = t.to(sample.device).reshape(-1).expand(B)
        t_emb = timestep_embedding(t, c.block_out_channels[0], c.flip_sin_to_cos, c.freq_shift).to(sample.dtype)
        emb = self.time_embedding(t_emb, timestep_cond)
        cet = getattr(c, "class_embed_type", None)
        if cet is not None:
            if class_labels is None:
                raise ValueError("class_labels should be provided when num_class_embeds > 0")
            if cet == "timestep":
                class_labels = timestep_embedding(class_labels.reshape(-1), c.block_out_channels[0], c.flip_sin_to_cos, c.freq_shift).to(sample.dtype)
            emb = emb + self.class_embedding(class_labels.to(sample.dtype))
        if c.addition_embed_type == "text_time":
            te = added_cond_kwargs["text_embeds"]
            tid = added_cond_kwargs["time_ids"]
            tide = timestep_embedding(tid.flatten(), c.addition_time_embed_dim, c.flip_sin_to_cos, c.freq_shift)
            add = torch.cat([te, tide.reshape(B, -1).to(te.dtype)], dim=-1).to(emb.dtype)
            emb = emb + self.add_embedding(add)
        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            h = blk(h, emb, encoder_hidden_states, skips)
        if down_block_additional_residuals is not None:
            # ControlNet (diffusers UNet2DConditionModel.forward): residuals are added to COPIES of the skip tensors,
            # the mid block still sees the un-augmented activation
            assert len(down_block_additional_residuals) == len(skips)
            skips = [s_ + r_ for s_, r_ in zip(skips, down_block_additional_residuals)]
        h = self.mid_block(h, emb, encoder_hidden_states)
        if mid_block_additional_residual is not None:
            h = h + mid_block_additional_residual
        for blk in self.up_blocks:
            h = blk(h, emb, encoder_hidden_states, skips)
        out = self.conv_out(F.silu(self.conv_norm_out(h)))
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)


def param_count(model):
    return sum(p.numel() for p in model.parameters())


def build(config="sd15", seed=0, dtype=torch.float32, device="cpu", **over):
    """Seeded default-initialised oracle model. Biases / norm affine params are perturbed so that a
    kernel which drops a bias or a gamma cannot pass parity (default init zeroes / ones them)."""
    cfg = dict({"sd15": SD15_CONFIG, "sdxl": SDXL_CONFIG, "tiny": tiny_config()}[config]) if isinstance(config, str) else dict(config)
    cfg.update(over)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        m = UNet2DConditionModel(**cfg)
        for name, p in m.named_parameters():
            if p.ndim == 1:
                if "norm" in name and name.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / math.sqrt(fan_in)))
    return m.to(device=device, dtype=dtype).eval()
