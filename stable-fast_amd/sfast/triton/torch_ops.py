"""`torch.ops.sfast_triton.*` on top of libsfast_hip.so.

Operator names and schemas are the ones the reference registers from Python
(/root/reference/src/sfast/triton/torch_ops.py:36-38, :65-67, :105-106, :167-169, :236-238,
:253-255, :294-296); the Triton kernels behind them are replaced by hand-written HIP. Registered
for the CUDA (= ROCm) dispatch key only -- no ATen / CPU fallback. Inference ops (no autograd).
"""
import torch

from ..hip import functional as F

_lib = torch.library.Library("sfast_triton", "DEF")


def _def(schema, fn):
    name = schema.split("(")[0]
    _lib.define(schema)
    _lib.impl(name, fn, "CUDA")


def _strides_for(shape, memory_format):
    return torch.empty(shape, device="meta").to(memory_format=memory_format).stride()


def contiguous(a, memory_format=torch.contiguous_format):
    if a.is_contiguous(memory_format=memory_format):
        return a
    return clone(a, memory_format=memory_format)


def clone(a, memory_format=torch.preserve_format):
    if memory_format == torch.preserve_format:
        out = torch.empty_like(a)
    else:
        out = torch.empty_like(a, memory_format=memory_format)
    if a.ndim <= 4:
        F.strided_copy(a, out)
    else:
        out.copy_(a)
    return out


def reshape(a, shape):
    # a view when strides allow it, otherwise one strided-copy kernel (reference :77-84)
    try:
        return a.view(shape)
    except RuntimeError:
        return contiguous(a).view(shape)


def group_norm(input, num_groups, weight=None, bias=None, eps=1e-5):
    return F.group_norm(input, num_groups, weight, bias, eps, act=None)


def group_norm_silu(input, num_groups, weight=None, bias=None, eps=1e-5):
    return F.group_norm(input, num_groups, weight, bias, eps, act="silu")


def layer_norm(input, normalized_shape, weight=None, bias=None, eps=1e-5):
    return F.layer_norm(input, normalized_shape, weight, bias, eps)


def _convolution(input, weight, bias, stride, padding, dilation, transposed, output_padding, groups, benchmark,
                 deterministic, cudnn_enabled, allow_tf32):
    if transposed:
        raise RuntimeError("sfast_triton::_convolution on ROCm supports non-transposed convolutions only")
    if groups != 1:
        # the reference hands grouped convolutions to ATen (triton/torch_ops.py:116-125); here they stay on the HIP library: one native
        # launch per group writing its channel slice of one output (the wrapper behind sfast::cudnn_convolution_bias, round 5)
        from ..torch_ops import _conv
        return _conv(input, weight, bias, None, None, list(stride), list(padding), list(dilation), False, [0] * len(stride), int(groups), None)
    return F.conv2d(input, weight, bias, stride=tuple(stride), padding=tuple(padding), dilation=tuple(dilation))


_def("contiguous(Tensor a, MemoryFormat memory_format) -> Tensor", contiguous)
_def("clone(Tensor a, MemoryFormat memory_format) -> Tensor", clone)
_def("reshape(Tensor a, int[] shape) -> Tensor", reshape)
_def("group_norm(Tensor input, int num_groups, Tensor? weight, Tensor? bias, float eps) -> Tensor", group_norm)
_def("group_norm_silu(Tensor input, int num_groups, Tensor? weight, Tensor? bias, float eps) -> Tensor", group_norm_silu)
_def("layer_norm(Tensor input, int[] normalized_shape, Tensor? weight, Tensor? bias, float eps) -> Tensor", layer_norm)
_def("_convolution(Tensor input, Tensor weight, Tensor? bias, int[] stride, int[] padding, int[] dilation, "
     "bool transposed, int[] output_padding, int groups, bool benchmark, bool deterministic, bool cudnn_enabled, "
     "bool allow_tf32) -> Tensor", _convolution)
