"""`torch.ops.sfast_xformers.memory_efficient_attention` served by the gfx950 flash-attention kernel.

Schema from /root/reference/src/sfast/libs/xformers/xformers_attention.py:46-48; q/k/v are
[B, S, H, D] (possibly strided views, /root/reference/src/sfast/libs/diffusers/xformers_attention.py:66-69).
xformers itself is an external dependency of the reference and is not used here. `attn_bias` is taken in its
tensor form (additive, broadcastable to [B, H, Sq, Skv] -- what diffusers' attention processors pass for
attention_mask / encoder_attention_mask, reference :30-47); dropout is not part of the inference path and is rejected loudly.
"""
from typing import Optional

import torch

from ...hip import functional as F

_lib = torch.library.Library("sfast_xformers", "DEF")


def memory_efficient_attention(query, key, value, attn_bias=None, p=0.0, scale=None, op=None):
    if attn_bias is not None and not torch.is_tensor(attn_bias):
        raise RuntimeError("sfast_xformers::memory_efficient_attention on ROCm: attn_bias must be a tensor (additive bias)")
    if p != 0.0:
        raise RuntimeError("sfast_xformers::memory_efficient_attention on ROCm: dropout is not supported")
    if query.ndim == 3:  # [B, S, D] single-head form
        return F.attention(query.unsqueeze(2), key.unsqueeze(2), value.unsqueeze(2), scale, attn_bias=attn_bias).squeeze(2)
    return F.attention(query, key, value, scale, attn_bias=attn_bias)


_lib.define("memory_efficient_attention(Tensor query, Tensor key, Tensor value, Tensor? attn_bias=None, float p=0.0, "
            "float? scale=None, str? op=None) -> Tensor")
_lib.impl("memory_efficient_attention", memory_efficient_attention, "CUDA")


def xformers_memory_efficient_attention(query, key, value, attn_bias=None, p: float = 0.0,
                                        scale: Optional[float] = None, *, op=None):
    """Python-level entry with the reference wrapper's signature (:51-63)."""
    return torch.ops.sfast_xformers.memory_efficient_attention(query, key, value, attn_bias, p, scale, None)
