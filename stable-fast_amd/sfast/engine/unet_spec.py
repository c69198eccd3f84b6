"""Parameter inventory of `UNet2DConditionModel` (diffusers state-dict names and shapes) derived from
a config -- lets the engine be driven from a bare state dict (safetensors header) or from seeded
random weights (bench / smoke runs; there are no checkpoints or network on the GPU box) without
instantiating a PyTorch module.

Names follow the public diffusers layout recorded in SURVEY.md Appendix A (the reference itself
never spells them out: it traces whatever module it is given).
"""
import math
import os
from typing import Dict, Tuple

import torch

SD15_CONFIG = dict(
    sample_size=64, in_channels=4, out_channels=4,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, cross_attention_dim=768,
    attention_head_dim=8, transformer_layers_per_block=1, norm_num_groups=32, norm_eps=1e-5,
    use_linear_projection=False, flip_sin_to_cos=True, freq_shift=0, addition_embed_type=None,
    addition_time_embed_dim=None, projection_class_embeddings_input_dim=None,
    time_cond_proj_dim=None, class_embed_type=None,
)

SDXL_CONFIG = dict(
    sample_size=128, in_channels=4, out_channels=4,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    block_out_channels=(320, 640, 1280), layers_per_block=2, cross_attention_dim=2048,
    attention_head_dim=(5, 10, 20), transformer_layers_per_block=(1, 2, 10), norm_num_groups=32, norm_eps=1e-5,
    use_linear_projection=True, flip_sin_to_cos=True, freq_shift=0, addition_embed_type="text_time",
    addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
)


def _per_block(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def unet2d_param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    boc = tuple(cfg["block_out_channels"])
    n = len(boc)
    L = cfg.get("layers_per_block", 2)
    depth = _per_block(cfg.get("transformer_layers_per_block", 1), n)
    ctx = cfg["cross_attention_dim"]
    lin = bool(cfg.get("use_linear_projection", False))
    T = boc[0] * 4
    out: Dict[str, Tuple[int, ...]] = {}

    def conv(name, cout, cin, k):
        out[name + ".weight"] = (cout, cin, k, k)
        out[name + ".bias"] = (cout,)

    def linear(name, cout, cin, bias=True):
        out[name + ".weight"] = (cout, cin)
        if bias:
            out[name + ".bias"] = (cout,)

    def norm(name, c):
        out[name + ".weight"] = (c,)
        out[name + ".bias"] = (c,)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cout, cin, 3)
        linear(name + ".time_emb_proj", cout, T)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cout, cin, 1)

    def transformer(name, c, d):
        norm(name + ".norm", c)
        for pn in ("proj_in", "proj_out"):
            if lin:
                linear(f"{name}.{pn}", c, c)
            else:
                conv(f"{name}.{pn}", c, c, 1)
        for j in range(d):
            b = f"{name}.transformer_blocks.{j}"
            for k_, kv in (("attn1", c), ("attn2", ctx)):
                norm(f"{b}.norm{1 if k_ == 'attn1' else 2}", c)
                linear(f"{b}.{k_}.to_q", c, c, bias=False)
                linear(f"{b}.{k_}.to_k", c, kv, bias=False)
                linear(f"{b}.{k_}.to_v", c, kv, bias=False)
                linear(f"{b}.{k_}.to_out.0", c, c)
            norm(f"{b}.norm3", c)
            linear(f"{b}.ff.net.0.proj", 8 * c, c)
            linear(f"{b}.ff.net.2", c, 4 * c)

    conv("conv_in", boc[0], cfg.get("in_channels", 4), 3)
    linear("time_embedding.linear_1", T, boc[0])
    linear("time_embedding.linear_2", T, T)
    if cfg.get("time_cond_proj_dim") is not None:       # LCM: w-embedding added to the sinusoid before the MLP
        linear("time_embedding.cond_proj", boc[0], cfg["time_cond_proj_dim"], bias=False)
    cet = cfg.get("class_embed_type")
    if cet == "timestep":
        linear("class_embedding.linear_1", T, boc[0])
        linear("class_embedding.linear_2", T, T)
    elif cet == "projection":
        linear("class_embedding.linear_1", T, cfg["projection_class_embeddings_input_dim"])
        linear("class_embedding.linear_2", T, T)
    if cfg.get("addition_embed_type") == "text_time":
        linear("add_embedding.linear_1", T, cfg["projection_class_embeddings_input_dim"])
        linear("add_embedding.linear_2", T, T)
    ch = boc[0]
    for i, t in enumerate(cfg["down_block_types"]):
        for j in range(L):
            resnet(f"down_blocks.{i}.resnets.{j}", ch if j == 0 else boc[i], boc[i])
            if t == "CrossAttnDownBlock2D":
                transformer(f"down_blocks.{i}.attentions.{j}", boc[i], depth[i])
        ch = boc[i]
        if i < n - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", ch, ch, 3)
    resnet("mid_block.resnets.0", boc[-1], boc[-1])
    transformer("mid_block.attentions.0", boc[-1], depth[-1])
    resnet("mid_block.resnets.1", boc[-1], boc[-1])
    rev, rdepth = boc[::-1], depth[::-1]
    prev = rev[0]
    for i, t in enumerate(cfg["up_block_types"]):
        co, ci = rev[i], rev[min(i + 1, n - 1)]
        for j in range(L + 1):
            skip = ci if j == L else co
            rin = prev if j == 0 else co
            resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, co)
            if t == "CrossAttnUpBlock2D":
                transformer(f"up_blocks.{i}.attentions.{j}", co, rdepth[i])
        if i < n - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", co, co, 3)
        prev = co
    norm("conv_norm_out", boc[0])
    conv("conv_out", cfg.get("out_channels", 4), boc[0], 3)
    return out


def random_params(cfg: dict, seed: int = 0, dtype=torch.float16, device="cuda") -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights with a variance-preserving init (no checkpoint is reachable offline).
    4-D weights are produced in channels_last (the K-contiguous layout the conv kernels read)."""
    g = torch.Generator(device=device).manual_seed(seed)
    params = {}
    for name, shape in unet2d_param_shapes(cfg).items():
        if len(shape) == 1:
            if "norm" in name and name.endswith("weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
            else:
                t = 0.05 * torch.randn(shape, generator=g, device=device)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g, device=device) * (1.0 / math.sqrt(fan_in))
        t = t.to(dtype)
        if t.ndim == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        params[name] = t
    return params


def find_unet_weights(path: str) -> str:
    """`path`: a diffusers model directory (…/unet/diffusion_pytorch_model[.fp16].safetensors inside it or in its `unet/`), or the
    weight file itself (.safetensors, or a torch-pickled .bin / .pt state dict)."""
    if os.path.isfile(path):
        return path
    for sub in ("unet", ""):
        for name in ("diffusion_pytorch_model.fp16.safetensors", "diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"):
            cand = os.path.join(path, sub, name)
            if os.path.isfile(cand):
                return cand
    raise FileNotFoundError(f"no diffusers UNet weights under {path!r} (looked for unet/diffusion_pytorch_model[.fp16].safetensors / .bin)")


def load_params(path: str, cfg: dict, dtype=torch.float16, device="cuda") -> Dict[str, torch.Tensor]:
    """Real weights behind the same interface as `random_params` (round 6): a diffusers-layout UNet checkpoint is read
    into the {state-dict name: tensor} form the engines and `oracle/unet_ref.py` take -- the parameter names of `unet2d_param_shapes`
    ARE diffusers' state-dict keys, so the first real checkpoint that becomes reachable also checks that appendix:
    a missing / unexpected key or a shape mismatch raises with the offending names. 4-D weights come back channels_last."""
    f = find_unet_weights(path)
    if f.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(f, device="cpu")
    else:
        sd = torch.load(f, map_location="cpu", weights_only=True)
    want = unet2d_param_shapes(cfg)
    missing = sorted(k for k in want if k not in sd)
    extra = sorted(k for k in sd if k not in want)
    wrong = sorted(k for k in want if k in sd and tuple(sd[k].shape) != tuple(want[k]))
    if missing or extra or wrong:
        raise ValueError(f"{f} does not match this UNet config: {len(missing)} missing (e.g. {missing[:3]}), {len(extra)} unexpected "
                         f"(e.g. {extra[:3]}), {len(wrong)} shape mismatches (e.g. {[(k, tuple(sd[k].shape), want[k]) for k in wrong[:2]]})")
    params = {}
    for k in want:
        t = sd[k].to(device=device, dtype=dtype)
        params[k] = t.contiguous(memory_format=torch.channels_last) if t.ndim == 4 else t.contiguous()
    return params


SD_VAE_DECODER_CONFIG = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                             norm_num_groups=32)


def vae_decoder_param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """Parameter inventory of diffusers `AutoencoderKL.decoder` (names as in its state dict); 49,490,179 parameters
    for SD_VAE_DECODER_CONFIG."""
    boc = tuple(cfg["block_out_channels"])
    shapes: Dict[str, Tuple[int, ...]] = {}

    def conv(name, cout, cin, k):
        shapes[name + ".weight"] = (cout, cin, k, k)
        shapes[name + ".bias"] = (cout,)

    def norm(name, c):
        shapes[name + ".weight"] = (c,)
        shapes[name + ".bias"] = (c,)

    def lin(name, cout, cin):
        shapes[name + ".weight"] = (cout, cin)
        shapes[name + ".bias"] = (cout,)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cout, cin, 3)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cout, cin, 1)

    c = boc[-1]
    conv("conv_in", c, cfg["latent_channels"], 3)
    resnet("mid_block.resnets.0", c, c)
    norm("mid_block.attentions.0.group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        lin("mid_block.attentions.0." + n, c, c)
    resnet("mid_block.resnets.1", c, c)
    rev = boc[::-1]
    prev = rev[0]
    for i, ch in enumerate(rev):
        for j in range(cfg["layers_per_block"] + 1):
            resnet(f"up_blocks.{i}.resnets.{j}", prev if j == 0 else ch, ch)
        if i != len(rev) - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", ch, ch, 3)
        prev = ch
    norm("conv_norm_out", boc[0])
    conv("conv_out", cfg["out_channels"], boc[0], 3)
    return shapes


def random_vae_decoder_params(cfg: dict, seed: int = 0, dtype=torch.float16, device="cuda") -> Dict[str, torch.Tensor]:
    """Seeded synthetic decoder weights (variance-preserving), 4-D weights in channels_last."""
    g = torch.Generator(device=device).manual_seed(seed)
    params = {}
    for name, shape in vae_decoder_param_shapes(cfg).items():
        if len(shape) == 1:
            t = (1.0 if ("norm" in name and name.endswith("weight")) else 0.0) + 0.05 * torch.randn(shape, generator=g, device=device)
        else:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            t = torch.randn(shape, generator=g, device=device) * (1.0 / math.sqrt(fan_in))
        t = t.to(dtype)
        if t.ndim == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        params[name] = t
    return params


class _Node(torch.nn.Module):
    """Attribute container of `module_from_params`; it has no forward of its own."""

    def forward(self, *args, **kwargs):
        raise RuntimeError("this module is a bare parameter container (sfast.engine.unet_spec.module_from_params): it has no eager "
                           "forward -- hand it to sfast.compilers.compile_unet() / UNet2DEngine.from_module()")


def module_from_params(cfg: dict, params: Dict[str, torch.Tensor]) -> torch.nn.Module:
    """A `torch.nn.Module` tree with diffusers' attribute / state-dict layout whose parameters ARE the tensors of `params` (no copy)
    and whose `.config` is `cfg` -- what `compile_unet()` needs from a UNet (named parameters + config) without diffusers and
    without an eager forward. Used to drive bare state dicts (safetensors) and the benchmark's synthetic weights through the
    drop-in `compile()` surface."""
    root = _Node()
    for name, t in params.items():
        parts = name.split(".")
        node = root
        for part in parts[:-1]:
            nxt = node._modules.get(part)
            if nxt is None:
                nxt = _Node()
                node.add_module(part, nxt)
            node = nxt
        node.register_parameter(parts[-1], torch.nn.Parameter(t, requires_grad=False))

    class _Cfg(dict):
        __getattr__ = dict.get

    root.config = _Cfg(cfg)
    first = next(iter(params.values()))
    root.dtype = first.dtype
    return root


SVD_CONFIG = dict(
    sample_size=96, in_channels=8, out_channels=4,
    down_block_types=("CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal",
                      "DownBlockSpatioTemporal"),
    up_block_types=("UpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal",
                    "CrossAttnUpBlockSpatioTemporal"),
    block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256, projection_class_embeddings_input_dim=768,
    layers_per_block=2, cross_attention_dim=1024, transformer_layers_per_block=1, num_attention_heads=(5, 10, 20, 20), num_frames=25,
)


def svd_param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """Parameter inventory of diffusers `UNetSpatioTemporalConditionModel` (state-dict names); 1,524,623,082 parameters for
    SVD_CONFIG (the published size of the SVD / SVD-XT UNet)."""
    boc = tuple(cfg["block_out_channels"])
    n = len(boc)
    Lp = cfg.get("layers_per_block", 2)
    d_ = cfg.get("transformer_layers_per_block", 1)
    depth = tuple(d_) if isinstance(d_, (tuple, list)) else (d_,) * n
    ctx = cfg["cross_attention_dim"]
    T = boc[0] * 4
    out: Dict[str, Tuple[int, ...]] = {}

    def conv(name, cout, cin, k):
        out[name + ".weight"] = (cout, cin, k, k)
        out[name + ".bias"] = (cout,)

    def conv3(name, cout, cin):
        out[name + ".weight"] = (cout, cin, 3, 1, 1)
        out[name + ".bias"] = (cout,)

    def linear(name, cout, cin, bias=True):
        out[name + ".weight"] = (cout, cin)
        if bias:
            out[name + ".bias"] = (cout,)

    def norm(name, c):
        out[name + ".weight"] = (c,)
        out[name + ".bias"] = (c,)

    def resnet(name, cin, cout):
        s, t = name + ".spatial_res_block", name + ".temporal_res_block"
        norm(s + ".norm1", cin)
        conv(s + ".conv1", cout, cin, 3)
        linear(s + ".time_emb_proj", cout, T)
        norm(s + ".norm2", cout)
        conv(s + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(s + ".conv_shortcut", cout, cin, 1)
        norm(t + ".norm1", cout)
        conv3(t + ".conv1", cout, cout)
        linear(t + ".time_emb_proj", cout, T)
        norm(t + ".norm2", cout)
        conv3(t + ".conv2", cout, cout)
        out[name + ".time_mixer.mix_factor"] = (1,)

    def attn_block(b, c, kv, pre_ff=False):
        if pre_ff:
            norm(b + ".norm_in", c)
            linear(b + ".ff_in.net.0.proj", 8 * c, c)
            linear(b + ".ff_in.net.2", c, 4 * c)
        for k_, kvd in (("attn1", c), ("attn2", kv)):
            norm(f"{b}.norm{1 if k_ == 'attn1' else 2}", c)
            linear(f"{b}.{k_}.to_q", c, c, bias=False)
            linear(f"{b}.{k_}.to_k", c, kvd, bias=False)
            linear(f"{b}.{k_}.to_v", c, kvd, bias=False)
            linear(f"{b}.{k_}.to_out.0", c, c)
        norm(b + ".norm3", c)
        linear(b + ".ff.net.0.proj", 8 * c, c)
        linear(b + ".ff.net.2", c, 4 * c)

    def transformer(name, c, d):
        norm(name + ".norm", c)
        linear(name + ".proj_in", c, c)
        for j in range(d):
            attn_block(f"{name}.transformer_blocks.{j}", c, ctx)
        for j in range(d):
            attn_block(f"{name}.temporal_transformer_blocks.{j}", c, ctx, pre_ff=True)
        linear(name + ".time_pos_embed.linear_1", 4 * c, c)
        linear(name + ".time_pos_embed.linear_2", c, 4 * c)
        out[name + ".time_mixer.mix_factor"] = (1,)
        linear(name + ".proj_out", c, c)

    conv("conv_in", boc[0], cfg.get("in_channels", 8), 3)
    linear("time_embedding.linear_1", T, boc[0])
    linear("time_embedding.linear_2", T, T)
    linear("add_embedding.linear_1", T, cfg["projection_class_embeddings_input_dim"])
    linear("add_embedding.linear_2", T, T)
    ch = boc[0]
    for i, t in enumerate(cfg["down_block_types"]):
        for j in range(Lp):
            resnet(f"down_blocks.{i}.resnets.{j}", ch if j == 0 else boc[i], boc[i])
        if t == "CrossAttnDownBlockSpatioTemporal":
            for j in range(Lp):
                transformer(f"down_blocks.{i}.attentions.{j}", boc[i], depth[i])
        ch = boc[i]
        if i < n - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", ch, ch, 3)
    resnet("mid_block.resnets.0", boc[-1], boc[-1])
    resnet("mid_block.resnets.1", boc[-1], boc[-1])
    transformer("mid_block.attentions.0", boc[-1], depth[-1])
    rev, rdepth = boc[::-1], depth[::-1]
    prev = rev[0]
    for i, t in enumerate(cfg["up_block_types"]):
        co, ci = rev[i], rev[min(i + 1, n - 1)]
        for j in range(Lp + 1):
            resnet(f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else co) + (ci if j == Lp else co), co)
        if t == "CrossAttnUpBlockSpatioTemporal":
            for j in range(Lp + 1):
                transformer(f"up_blocks.{i}.attentions.{j}", co, rdepth[i])
        if i < n - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", co, co, 3)
        prev = co
    norm("conv_norm_out", boc[0])
    conv("conv_out", cfg.get("out_channels", 4), boc[0], 3)
    return out


def random_svd_params(cfg: dict, seed: int = 0, dtype=torch.float16, device="cuda") -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights of the spatio-temporal UNet; 4-D / 5-D weights in channels_last / channels_last_3d."""
    g = torch.Generator(device=device).manual_seed(seed)
    params = {}
    for name, shape in svd_param_shapes(cfg).items():
        if name.endswith("mix_factor"):
            t = torch.randn(shape, generator=g, device=device)
        elif len(shape) == 1:
            t = (1.0 if ("norm" in name and name.endswith("weight")) else 0.0) + 0.05 * torch.randn(shape, generator=g, device=device)
        else:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            t = torch.randn(shape, generator=g, device=device) * (1.0 / math.sqrt(fan_in))
        t = t.to(dtype)
        if t.ndim == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        elif t.ndim == 5:
            t = t.contiguous(memory_format=torch.channels_last_3d)
        params[name] = t
    return params
