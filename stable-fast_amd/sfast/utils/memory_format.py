"""channels_last conversion of a module's 4-D parameters / buffers.

Same contract as /root/reference/src/sfast/utils/memory_format.py:49-57: parameter objects keep
their identity, only their storage layout changes, so conv weights become the K-contiguous
[Cout][kh][kw][Cin] image the implicit-GEMM kernels read directly.
"""
import torch


def suggest_memory_format(x):
    if x.layout != torch.strided:
        return torch.contiguous_format
    if x.ndim == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
        return torch.channels_last
    if x.ndim == 5 and x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous():
        return torch.channels_last_3d
    return torch.contiguous_format


def apply_memory_format(m, memory_format=torch.preserve_format):
    def convert(t):
        if memory_format is None or memory_format == torch.preserve_format:
            return t
        if t.dim() == 4 and memory_format == torch.channels_last:
            return t.to(memory_format=memory_format)
        if t.dim() == 5 and memory_format == torch.channels_last_3d:
            return t.to(memory_format=memory_format)
        if t.dim() in (4, 5) and memory_format == torch.contiguous_format:
            return t.to(memory_format=memory_format)
        return t

    return m._apply(convert)
