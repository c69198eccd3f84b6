"""ORACLE -- test infrastructure only (tests/, smoke(), bench.py cpu_baseline).

fp32 PyTorch restatements of each operator family of the hot path, written against the arithmetic
the reference's fused ops are tested to reproduce:

  group_norm(+silu)  = aten.native_group_norm (+ aten.silu): biased variance, rstd = 1/sqrt(var+eps),
                       affine then activation  (/root/reference/src/sfast/triton/torch_ops.py:179-189,
                       /root/reference/src/sfast/triton/ops/group_norm.py:48-49,75-82)
  layer_norm         = aten.layer_norm        (/root/reference/src/sfast/triton/torch_ops.py:241-246)
  linear family      = y = act(x W^T + b) + alpha*other  (/root/reference/src/sfast/csrc/operators/cublas/
                       cublas_gemm.cpp:798-948); GEGLU = linear -> chunk(2,-1) -> h * gelu(g)
                       (/root/reference/src/sfast/jit/passes/__init__.py:643-649,
                       /root/reference/tests/operators/test_cutlass_dual_linear.py:37-40)
  conv family        = act(conv(x, w) + alpha*z + b)  (/root/reference/src/sfast/csrc/operators/cudnn/
                       cudnn_convolution_impl.cc:995-998; test model tests/operators/test_cudnn_convolution.py:14-27)
  attention          = softmax(q k^T * scale) v on [B, S, H, D]  (xformers.ops.memory_efficient_attention as
                       called at /root/reference/src/sfast/libs/xformers/xformers_attention.py:36-42)

Pinned against the ATen ops the reference's own tests use as ground truth in tests/test_oracle.py.
All functions take and return fp32 unless stated; callers round to the kernel's I/O dtype.
"""
import math

import torch
import torch.nn.functional as F


def act_ref(v, act):
    if act in (None, "none", 0):
        return v
    if act in ("relu", 1):
        return torch.relu(v)
    if act in ("gelu", 2):
        return F.gelu(v)
    if act in ("gelu_tanh", 3):
        return F.gelu(v, approximate="tanh")
    if act in ("silu", 4):
        return F.silu(v)
    if act in ("sigmoid", 5):
        return torch.sigmoid(v)
    if act in ("tanh", 6):
        return torch.tanh(v)
    raise ValueError(act)


def group_norm_ref(x, num_groups, weight=None, bias=None, eps=1e-5, silu=False):
    """x: [N, C, *] fp32."""
    y = F.group_norm(x.float(), num_groups, None if weight is None else weight.float(),
                     None if bias is None else bias.float(), eps)
    return F.silu(y) if silu else y


def group_norm_manual(x, num_groups, weight=None, bias=None, eps=1e-5):
    """Independent restatement (no aten.group_norm) used to pin group_norm_ref: biased variance over
    (C/G, *spatial), rstd = 1/sqrt(var + eps), then per-channel affine."""
    N, C = x.shape[:2]
    xs = x.double().reshape(N, num_groups, -1)
    mean = xs.mean(dim=2, keepdim=True)
    var = ((xs - mean) ** 2).mean(dim=2, keepdim=True)
    y = ((xs - mean) / torch.sqrt(var + eps)).reshape(x.shape)
    shape = (1, C) + (1,) * (x.ndim - 2)
    if weight is not None:
        y = y * weight.double().reshape(shape)
    if bias is not None:
        y = y + bias.double().reshape(shape)
    return y.float()


def layer_norm_ref(x, normalized_shape, weight=None, bias=None, eps=1e-5):
    return F.layer_norm(x.float(), tuple(normalized_shape), None if weight is None else weight.float(),
                        None if bias is None else bias.float(), eps)


def linear_ref(x, weight, bias=None, act=None, residual=None, alpha=1.0, res_before_act=False, geglu=False,
               rowbias=None, rows_per_batch=0, in_act=None):
    """x [..., K] ; weight [N(or 2N), K]. Returns fp32."""
    xf = act_ref(x.float(), in_act)
    v = xf @ weight.float().t()
    if bias is not None:
        v = v + bias.float()
    if geglu:
        h, g = v.chunk(2, dim=-1)
        return h * F.gelu(g)
    if rowbias is not None:
        M = v.reshape(-1, v.shape[-1]).shape[0]
        idx = torch.arange(M) // rows_per_batch
        v = (v.reshape(M, -1) + rowbias.float()[idx]).reshape(v.shape)
    r = 0.0 if residual is None else alpha * residual.float()
    if res_before_act:
        return act_ref(v + r, act)
    return act_ref(v, act) + r


def conv2d_ref(x, weight, bias=None, z=None, alpha=1.0, stride=1, padding=0, dilation=1, act=None, res_before_act=True,
               x2=None, upsample2x=False, rowbias=None):
    """Logical NCHW tensors. y = act(conv + bias + rowbias[b] + alpha*z)."""
    xf = x.float()
    if x2 is not None:
        xf = torch.cat([xf, x2.float()], dim=1)
    if upsample2x:
        xf = F.interpolate(xf, scale_factor=2.0, mode="nearest")
    v = F.conv2d(xf, weight.float(), None if bias is None else bias.float(), stride, padding, dilation)
    if rowbias is not None:
        v = v + rowbias.float()[:, :, None, None]
    r = 0.0 if z is None else alpha * z.float()
    if res_before_act:
        return act_ref(v + r, act)
    return act_ref(v, act) + r


def attention_ref(q, k, v, scale=None, attn_bias=None):
    """q [B, Sq, H, D], k/v [B, Skv, H, D] -> [B, Sq, H, D]; explicit softmax form in fp32. attn_bias: additive, broadcastable
    to [B, H, Sq, Skv] (xformers.ops.memory_efficient_attention's tensor bias as the reference passes it through,
    /root/reference/src/sfast/libs/xformers/xformers_attention.py:30-47)."""
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    s = scale if scale is not None else 1.0 / math.sqrt(q.shape[-1])
    logits = qf @ kf.transpose(-1, -2) * s
    if attn_bias is not None:
        logits = logits + attn_bias.float()
    p = torch.softmax(logits, dim=-1)
    return (p @ vf).transpose(1, 2).contiguous()


def cfg_ddim_ref(eps_uc, latents, coef, guidance):
    eu, ec = eps_uc.float().reshape(2, -1)
    e = eu + guidance * (ec - eu)
    x = latents.float().reshape(-1)
    sa, s1a, sp, s1p = [float(c) for c in coef]
    x0 = (x - s1a * e) / sa
    return (sp * x0 + s1p * e).reshape(latents.shape)


def ddim_schedule(num_steps=50, num_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
    """SD1.5 DDIM constants: scaled-linear betas, leading spacing, offset 1,
    set_alpha_to_one=False. Returns (timesteps, [(sqrt a_t, sqrt(1-a_t), sqrt a_prev, sqrt(1-a_prev))])."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float64) ** 2
    acp = torch.cumprod(1.0 - betas, dim=0)
    ratio = num_train // num_steps
    ts = (torch.arange(0, num_steps) * ratio).flip(0) + steps_offset
    coefs = []
    for t in ts.tolist():
        a_t = acp[t]
        prev = t - ratio
        a_p = acp[prev] if prev >= 0 else acp[0]
        coefs.append((float(a_t.sqrt()), float((1 - a_t).sqrt()), float(a_p.sqrt()), float((1 - a_p).sqrt())))
    return ts.tolist(), coefs
