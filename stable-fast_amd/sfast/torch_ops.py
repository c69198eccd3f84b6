"""`torch.ops.sfast.*` -- the operator names the reference registers in C++
(/root/reference/src/sfast/csrc/main.cpp:18-24) re-registered on top of libsfast_hip.so.

Schemas follow the reference headers:
  cutlass_linear_geglu[_unified]   csrc/operators/cutlass/cutlass_dual_linear_kernel.h:6-15
  cudnn_convolution_bias[_add][_sigmoid|_relu|_tanh]   csrc/operators/cudnn/cudnn_convolution.h:12-78
  cublas_lowp_*                    csrc/operators/cublas/cublas_gemm.h:12-49
  linear_relu / linear_gelu        csrc/operators/fused_linear.h:11-15

The implementations are registered for the CUDA (= ROCm) dispatch key only: calling them with CPU
tensors is a dispatcher error, there is no ATen fallback (the hot path must not silently leave the
HIP kernels). They are inference ops; the conv family additionally carries the reference's backward
formula (its LoRA-training example differentiates through the fused convs).
"""
import torch

from .hip import functional as F

_lib = torch.library.Library("sfast", "DEF")


def _def(schema, fn):
    name = schema.split("(")[0]
    _lib.define(schema)
    _lib.impl(name, fn, "CUDA")


# ---- Linear + GEGLU ------------------------------------------------------------------------------
def cutlass_linear_geglu_unified(input, weight, bias=None):
    return F.linear(input, weight, bias, geglu=True)


def cutlass_linear_geglu(input, weight0, bias0, weight1, bias1):
    n, k = weight0.shape
    # (round 4) the old "adjacent in memory -> one as_strided [2n, k] view" shortcut is gone: two weights can sit back to back in
    # DIFFERENT storages (the caching allocator does that), where as_strided raises "out of bounds for storage" -- found by
    # test_two_weight_geglu_reads_the_weights_in_place. The segment form below covers adjacent weights too, without a view.
    if (weight1.shape == weight0.shape and weight0.stride(1) == 1 and weight1.stride(1) == 1 and weight0.stride(0) == weight1.stride(0)):
        w = [weight0, weight1]   # (hidden, gate) weight segments of ONE launch: the live weights are read in place, no concatenated copy
    else:
        w = torch.cat([weight0, weight1], dim=0)
    if (bias0 is None) != (bias1 is None):
        z = torch.zeros(n, dtype=input.dtype, device=input.device)
        bias0 = z if bias0 is None else bias0
        bias1 = z if bias1 is None else bias1
    b = None if bias0 is None else torch.cat([bias0, bias1], dim=0)
    return F.linear(input, w, b, geglu=True)


_def("cutlass_linear_geglu_unified(Tensor input, Tensor weight, Tensor? bias) -> Tensor", cutlass_linear_geglu_unified)
_def("cutlass_linear_geglu(Tensor input, Tensor weight0, Tensor? bias0, Tensor weight1, Tensor? bias1) -> Tensor",
     cutlass_linear_geglu)


# ---- conv + bias (+ alpha*z) (+ activation) ----------------------------------------------------------
def _conv(input, weight, bias, z, alpha, stride, padding, dilation, transposed, output_padding, groups, act, out=None):
    if transposed or any(int(o) != 0 for o in output_padding):
        raise RuntimeError("sfast conv ops on ROCm support non-transposed convolutions only")
    if input.ndim == 3:  # 1-D conv as 2-D, like the reference (cudnn_convolution_impl.cc:1242-1252)
        y = _conv(input.unsqueeze(2), weight.unsqueeze(2), bias, None if z is None else z.unsqueeze(2), alpha,
                  [1, stride[0]], [0, padding[0]], [1, dilation[0]], transposed, [0, 0], groups, act)
        return y.squeeze(2)
    if groups != 1:
        # grouped / depthwise: one native launch per group on channel-sliced views (the reference leaves these to ATen,
        # cudnn_convolution_impl.cc:1265-1286); every group writes its channel slice of ONE output tensor through the kernels'
        # output strides -- no per-group temporaries, no concatenation
        cin_g, cout_g = input.shape[1] // groups, weight.shape[0] // groups
        if input.shape[1] != cin_g * groups or weight.shape[0] != cout_g * groups or weight.shape[1] != cin_g:
            raise RuntimeError("sfast conv: channels are not divisible by groups")
        y = None
        for g in range(groups):
            zg = None if z is None else (z[:, g * cout_g:(g + 1) * cout_g] if z.shape[1] == weight.shape[0] else z)
            xg, wg = input[:, g * cin_g:(g + 1) * cin_g], weight[g * cout_g:(g + 1) * cout_g]
            bg = None if bias is None else bias[g * cout_g:(g + 1) * cout_g]
            if y is None:   # the first group fixes the output geometry and memory format (one allocation for all groups)
                y0 = _conv(xg, wg, bg, zg, alpha, stride, padding, dilation, False, output_padding, 1, act)
                cl = y0.is_contiguous(memory_format=torch.channels_last) and not y0.is_contiguous()
                y = torch.empty((y0.shape[0], weight.shape[0], y0.shape[2], y0.shape[3]), dtype=y0.dtype, device=y0.device,
                                memory_format=torch.channels_last if cl else torch.contiguous_format)
                y[:, :cout_g].copy_(y0)
                continue
            _conv(xg, wg, bg, zg, alpha, stride, padding, dilation, False, output_padding, 1, act, out=y[:, g * cout_g:(g + 1) * cout_g])
        return y
    a = 1.0 if alpha is None else float(alpha)
    # y = act(conv + alpha*z + bias)   (cudnn_convolution_impl.cc:995-998)
    return F.conv2d(input, weight, bias, z=z, alpha=a, stride=tuple(stride), padding=tuple(padding),
                    dilation=tuple(dilation), act=act, res_before_act=True, out=out)


def _mk_conv_bias(act):
    def op(input, weight, bias, stride, padding, dilation, transposed, output_padding, groups):
        return _conv(input, weight, bias, None, None, stride, padding, dilation, transposed, output_padding, groups, act)
    return op


def _mk_conv_bias_add(act):
    def op(input, weight, bias, z, alpha, stride, padding, dilation, transposed, output_padding, groups):
        return _conv(input, weight, bias, z, alpha, stride, padding, dilation, transposed, output_padding, groups, act)
    return op


def _conv_autograd(name, act, has_z):
    """Backward of the fused conv ops for the reference's LoRA-training use (its autograd Function,
    cudnn_convolution_impl.cc:1289-1399): activation derivative from the saved OUTPUT, then the library convolution backward
    (the reference calls torch::convolution_backward too -- training is not the path this package accelerates), the residual's
    gradient reduced to z's shape and scaled by alpha. The forward stays the HIP kernel."""
    def setup(ctx, inputs, output):
        if has_z:
            x, w, b, z, alpha, stride, padding, dilation, transposed, output_padding, groups = inputs
        else:
            (x, w, b, stride, padding, dilation, transposed, output_padding, groups), z, alpha = inputs, None, None
        ctx.save_for_backward(x, w, output)
        ctx.conv = (None if b is None else list(b.shape), list(stride), list(padding), list(dilation), bool(transposed),
                    list(output_padding), int(groups))
        ctx.z_shape = None if z is None else tuple(z.shape)
        ctx.alpha = 1.0 if alpha is None else float(alpha)

    def backward(ctx, grad):
        x, w, y = ctx.saved_tensors
        if act == "sigmoid":
            g = grad * y * (1 - y)
        elif act == "relu":
            g = grad * (y > 0).to(grad.dtype)
        elif act == "tanh":
            g = grad * (1 - y * y)
        else:
            g = grad
        g = g.contiguous()
        bias_sizes, stride, padding, dilation, transposed, output_padding, groups = ctx.conv
        need = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2] and bias_sizes is not None]
        gx = gw = gb = None
        if any(need):
            gx, gw, gb = torch.ops.aten.convolution_backward(g, x, w, bias_sizes, stride, padding, dilation, transposed, output_padding,
                                                             groups, need)
        if not has_z:
            return gx, gw, gb, None, None, None, None, None, None
        gz = None
        if ctx.z_shape is not None and ctx.needs_input_grad[3]:
            gz = torch.zeros(ctx.z_shape, dtype=g.dtype, device=g.device) if ctx.alpha == 0.0 else g.sum_to_size(ctx.z_shape) * ctx.alpha
        return gx, gw, gb, gz, None, None, None, None, None, None, None

    torch.library.register_autograd(f"sfast::{name}", backward, setup_context=setup)


_CONV_ARGS = "int[] stride, int[] padding, int[] dilation, bool transposed, int[] output_padding, int groups"
for _suffix, _act in (("", None), ("_sigmoid", "sigmoid"), ("_relu", "relu"), ("_tanh", "tanh")):
    _def(f"cudnn_convolution_bias{_suffix}(Tensor input, Tensor weight, Tensor? bias, {_CONV_ARGS}) -> Tensor",
         _mk_conv_bias(_act))
    _def(f"cudnn_convolution_bias_add{_suffix}(Tensor input, Tensor weight, Tensor? bias, Tensor? z, Scalar? alpha, "
         f"{_CONV_ARGS}) -> Tensor", _mk_conv_bias_add(_act))
    _conv_autograd(f"cudnn_convolution_bias{_suffix}", _act, False)
    _conv_autograd(f"cudnn_convolution_bias_add{_suffix}", _act, True)


# ---- low-precision GEMM family ---------------------------------------------------------------------------
def _as_weight(mat2):
    """[K, N] matmul operand -> K-contiguous [N, K] weight view (copy only when unavoidable)."""
    w = mat2.t()
    return w if (w.stride(1) == 1 and w.stride(0) >= w.shape[1]) else w.contiguous()


def cublas_lowp_linear(input, weight, bias=None):
    return F.linear(input, weight, bias)


def cublas_lowp_linear_relu(input, weight, bias=None):
    return F.linear(input, weight, bias, act="relu")


def cublas_lowp_linear_gelu(input, weight, bias=None):
    return F.linear(input, weight, bias, act="gelu")


def cublas_lowp_linear_add(input, weight, bias, other, alpha=1):
    # D = (x W^T + bias) + alpha * other   (cublas_gemm.cpp:900-948)
    return F.linear(input, weight, bias, residual=other, alpha=float(alpha))


def cublas_lowp_mm(self, mat2):
    return F.linear(self, _as_weight(mat2))


def _addmm(self, mat1, mat2, beta, alpha, act=None, other=None, gamma=1.0):
    """act(alpha * mat1 @ mat2 + beta * self) [+ gamma * other] as ONE GEMM launch: alpha is the accumulator scale of the epilogue
    (`out_scale`); `self` is the bias operand when it is a [N] vector with beta == 1, and otherwise the epilogue's scaled residual
    operand, added before the activation (a [N] or [1, N] `self` is a residual with row stride 0 -- no expanded copy is made).
    Reference: csrc/operators/cublas/cublas_gemm.cc:206-330 (cublasGemmEx alpha / beta, then the bias / activation epilogue)."""
    beta, alpha, gamma = float(beta), float(alpha), float(gamma)
    w = _as_weight(mat2)
    n = w.shape[0]
    if alpha == 0.0:
        # torch.addmm semantics: the product is not computed (NaN / inf in mat1 / mat2 do not propagate) -- and the library reads an
        # accumulator scale of 0 as "unset" (= 1), so this degenerate case never reaches it. Pure broadcast arithmetic, no GEMM.
        m = mat1.shape[0]
        base = torch.zeros((m, n), dtype=mat1.dtype, device=mat1.device) if beta == 0.0 else (self * beta).expand(m, n).to(mat1.dtype)
        if act is not None:
            base = torch.relu(base) if act == "relu" else torch.nn.functional.gelu(base)
        return base + gamma * other if other is not None else base.clone()
    if beta == 0.0:
        # torch.addmm ignores `self` when beta == 0 (a NaN there must not reach the result): no residual operand at all
        return F.linear(mat1, w, None, act=act, residual=other, alpha=gamma, out_scale=alpha)
    if beta == 1.0 and self.ndim == 1 and self.shape[0] == n:
        return F.linear(mat1, w, self, act=act, residual=other, alpha=gamma, out_scale=alpha)
    if other is None:
        return F.linear(mat1, w, None, act=act, residual=self, alpha=beta, res_before_act=True, out_scale=alpha)
    # two scaled addends besides the product (addmm_add with a general `self`; the reference's fuser never emits it): the epilogue has
    # one residual slot, so beta * self rides in the GEMM and gamma * other in a second pass over the output
    out = F.linear(mat1, w, None, act=act, residual=self, alpha=beta, res_before_act=True, out_scale=alpha)
    return out.add_(other, alpha=gamma)


def cublas_lowp_addmm(self, mat1, mat2, beta=1, alpha=1):
    return _addmm(self, mat1, mat2, beta, alpha)


def cublas_lowp_addmm_add(self, mat1, mat2, other, beta=1, alpha=1, gamma=1):
    return _addmm(self, mat1, mat2, beta, alpha, other=other, gamma=gamma)


def cublas_lowp_addmm_activation(self, mat1, mat2, beta=1, alpha=1, use_gelu=False):
    return _addmm(self, mat1, mat2, beta, alpha, act="gelu" if use_gelu else "relu")


def cublas_lowp_bmm(self, batch2):
    if self.ndim != 3 or batch2.ndim != 3 or self.shape[0] != batch2.shape[0]:
        raise RuntimeError("cublas_lowp_bmm: expected [B, M, K] x [B, K, N]")
    B, M, K = self.shape
    N = batch2.shape[2]
    if K % 8 == 0 and N % 4 == 0 and self.dtype in (torch.float16, torch.bfloat16):
        # one grouped launch per 64 batch elements: group b = (self[b], batch2[b]^T), outputs are rows of ONE [B, M, N] tensor
        # the GEMM kernels read both operands K-contiguous: batch2 [B, K, N] -> [B, N, K] by ONE native strided copy (unless it already
        # is a transposed view of such a tensor); the outputs are the rows of ONE [B, M, N] tensor -- no stack
        bt = batch2.transpose(1, 2)
        w = bt if bt.is_contiguous() else F.strided_copy(bt, torch.empty((B, N, K), dtype=batch2.dtype, device=batch2.device))
        out = torch.empty((B, M, N), dtype=self.dtype, device=self.device)
        F.linear_grouped([self[b] for b in range(B)], [[w[b]] for b in range(B)], outs=[out[b] for b in range(B)])
        return out
    out = torch.empty((B, M, N), dtype=self.dtype, device=self.device)
    for b in range(B):
        F.linear(self[b], _as_weight(batch2[b]), out=out[b])
    return out


def cublas_lowp_baddbmm(self, batch1, batch2, beta, alpha):
    return cublas_lowp_bmm(batch1, batch2) * float(alpha) + self * float(beta)


def cublas_lowp_matmul(tensor1, tensor2):
    if tensor2.ndim == 2:
        return F.linear(tensor1, _as_weight(tensor2))
    if tensor1.ndim == 3 and tensor2.ndim == 3:
        return cublas_lowp_bmm(tensor1, tensor2)
    lead = torch.broadcast_shapes(tensor1.shape[:-2], tensor2.shape[:-2])
    a = tensor1.expand(*lead, *tensor1.shape[-2:]).reshape(-1, *tensor1.shape[-2:])
    b = tensor2.expand(*lead, *tensor2.shape[-2:]).reshape(-1, *tensor2.shape[-2:])
    return cublas_lowp_bmm(a, b).reshape(*lead, tensor1.shape[-2], tensor2.shape[-1])


_def("cublas_lowp_linear(Tensor input, Tensor weight, Tensor? bias=None) -> Tensor", cublas_lowp_linear)
_def("cublas_lowp_linear_relu(Tensor input, Tensor weight, Tensor? bias=None) -> Tensor", cublas_lowp_linear_relu)
_def("cublas_lowp_linear_gelu(Tensor input, Tensor weight, Tensor? bias=None) -> Tensor", cublas_lowp_linear_gelu)
_def("cublas_lowp_linear_add(Tensor input, Tensor weight, Tensor? bias, Tensor other, Scalar alpha=1) -> Tensor",
     cublas_lowp_linear_add)
_def("cublas_lowp_mm(Tensor self, Tensor mat2) -> Tensor", cublas_lowp_mm)
_def("cublas_lowp_addmm(Tensor self, Tensor mat1, Tensor mat2, Scalar beta=1, Scalar alpha=1) -> Tensor", cublas_lowp_addmm)
_def("cublas_lowp_addmm_add(Tensor self, Tensor mat1, Tensor mat2, Tensor other, Scalar beta=1, Scalar alpha=1, "
     "Scalar gamma=1) -> Tensor", cublas_lowp_addmm_add)
_def("cublas_lowp_addmm_activation(Tensor self, Tensor mat1, Tensor mat2, Scalar beta=1, Scalar alpha=1, "
     "bool use_gelu=False) -> Tensor", cublas_lowp_addmm_activation)
_def("cublas_lowp_bmm(Tensor self, Tensor batch2) -> Tensor", cublas_lowp_bmm)
_def("cublas_lowp_baddbmm(Tensor self, Tensor batch1, Tensor batch2, Scalar beta, Scalar alpha) -> Tensor", cublas_lowp_baddbmm)
_def("cublas_lowp_matmul(Tensor tensor1, Tensor tensor2) -> Tensor", cublas_lowp_matmul)


# ---- fused linear (ATen _addmm_activation in the reference) -------------------------------------------------
def linear_relu(input, weight, bias=None):
    return F.linear(input, weight, bias, act="relu")


def linear_gelu(input, weight, bias=None):
    return F.linear(input, weight, bias, act="gelu")


_def("linear_relu(Tensor input, Tensor weight, Tensor? bias=None) -> Tensor", linear_relu)
_def("linear_gelu(Tensor input, Tensor weight, Tensor? bias=None) -> Tensor", linear_gelu)


# ---- int8 dynamic linear (reference csrc/operators/cutlass/cutlass_qlinear.cc:10-89) ---------------------------------------------
# The reference does not add an operator of its own shape here: it OVERRIDES `quantized::linear_dynamic(X, W_prepack, reduce_range)`
# for the CUDA / QuantizedCUDA dispatch keys (cutlass_qlinear.cc:73-81), so that `torch.quantization.quantize_dynamic(unet)` modules
# reach its kernel, and exposes the same function as `sfast::cutlass_qlinear_dynamic` (:83-88). Same binding here; the kernel behind
# it is libsfast_hip's int8-weight MFMA GEMM (weights stay int8 in HBM, widened while staged into LDS; activations stay 16-bit --
# exactly the reference's arithmetic: its kernel is CUTLASS' mixed-input GEMM, `OpMultiplyAddMixedInputUpcast`, int8 weights upcast
# to the 16-bit activation type, cutlass_qlinear_dynamic_kernel.cu:68,82 -- weight-only despite the "dynamic" in the op name).
def cutlass_qlinear_dynamic_unpacked(input, weight, bias=None):
    """`weight`: a per-tensor-affine quantized qint8 tensor [N, K] (torch.quantize_per_tensor); like the reference the zero point
    is ignored (weight.int_repr() * weight.q_scale(), cutlass_qlinear_dynamic_kernel.cu:272-279). f16 / bf16 inputs run the
    int8-weight MFMA kernel, anything else (fp32 inputs, shapes outside its alignment rules) dequantises the weight and takes the
    ordinary linear kernel -- the reference's own fallback (:231-238)."""
    if not weight.is_quantized:
        raise RuntimeError("weight should be quantized")
    return _qlinear_w8(input, weight.int_repr(), float(weight.q_scale()), bias)


def _qlinear_w8(input, w8, scale, bias):
    N, K = w8.shape
    if bias is not None and bias.dtype != input.dtype:
        bias = bias.to(input.dtype)
    if input.dtype in (torch.float16, torch.bfloat16) and K % 8 == 0 and N % 4 == 0:
        return F.qlinear_w8(input, w8, scale, bias)
    wd = (w8.to(torch.float32) * scale).to(input.dtype)
    return F.linear(input, wd, bias)


class _PackedOnDevice:
    """What PackedLinearWeightCutlass holds in the reference (orig_weight, bias_ -- cutlass_qlinear.cc:21-24): the int8 weight image
    and the bias of one packed-params object, resident on the activation's device. `LinearPackedParamsBase` is a C++ torchbind class
    that Python cannot subclass, so the packed object the dispatcher hands over is PyTorch's own (CPU-packed) one and this record
    hangs off it in a side table: filled by the QuantizedCUDA `linear_prepack` below with the tensors it was given (no copy), or on
    first use from `linear_unpack` for objects packed elsewhere (the reference's `from_native`, :43-51)."""
    __slots__ = ("w8", "scale", "bias")

    def __init__(self, w8, scale, bias):
        self.w8, self.scale, self.bias = w8, scale, bias


# hash(packed) -> (packed, {device: _PackedOnDevice}). The dispatcher hands every call a NEW Python wrapper of the same C++ object: the
# wrappers hash alike (by the C++ pointer) but implement no __eq__, so the table is keyed by the hash itself; the entry keeps one wrapper
# -- and with it the C++ object -- alive, so the pointer cannot be recycled for another weight while its record is cached.
_PACKED = {}
_PACKED_LIMIT = 4096


def _remember(packed, dev, rec):
    if len(_PACKED) >= _PACKED_LIMIT:   # modules re-packed in a loop: keep the table bounded (records are re-derivable)
        _PACKED.clear()
    _PACKED.setdefault(hash(packed), (packed, {}))[1][dev] = rec


def _device_record(packed, device):
    ent = _PACKED.get(hash(packed))
    rec = ent[1].get(device) if ent else None
    if rec is None:
        w, b = torch.ops.quantized.linear_unpack(packed)
        if w.qscheme() not in (torch.per_tensor_affine, torch.per_tensor_symmetric):
            raise RuntimeError(f"Unsupported qscheme: {w.qscheme()}")        # cutlass_qlinear.cc:29-30
        rec = _PackedOnDevice(w.int_repr().to(device).contiguous(), float(w.q_scale()), b.to(device) if b is not None else None)
        _remember(packed, device, rec)
    return rec


def quantized_linear_prepack_cuda(W, B=None):
    """`quantized::linear_prepack` for QuantizedCUDA weights (what `quantize_dynamic(module.cuda())` calls; PyTorch-ROCm registers it
    for QuantizedCPU only). Checks follow PackedLinearWeightCutlass::prepack (cutlass_qlinear.cc:26-41)."""
    if W.qscheme() not in (torch.per_tensor_affine, torch.per_tensor_symmetric):
        raise RuntimeError(f"Unsupported qscheme: {W.qscheme()}")
    if B is not None and (B.dim() != 1 or B.shape[0] != W.shape[0]):
        raise RuntimeError(f"bias should be a vector (1D Tensor) with {W.shape[0]} elements")
    w8 = W.int_repr()
    wq_cpu = torch._make_per_tensor_quantized_tensor(w8.cpu(), float(W.q_scale()), int(W.q_zero_point()))
    packed = torch.ops.quantized.linear_prepack(wq_cpu, B.detach().float().cpu() if B is not None else None)
    _remember(packed, W.device, _PackedOnDevice(w8.contiguous(), float(W.q_scale()), B.detach() if B is not None else None))
    return packed


def cutlass_qlinear_dynamic(X, W_prepack, reduce_range=False):
    """QLinearInt8<false>::run_dynamic (cutlass_qlinear.cc:60-71): `reduce_range` is accepted and ignored, as there (:13-16)."""
    rec = _device_record(W_prepack, X.device)
    return _qlinear_w8(X, rec.w8, rec.scale, rec.bias)


_PACKED_T = "__torch__.torch.classes.quantized.LinearPackedParamsBase"
_lib.define(f"cutlass_qlinear_dynamic(Tensor X, {_PACKED_T} W_prepack, bool reduce_range=False) -> Tensor")
_lib.impl("cutlass_qlinear_dynamic", cutlass_qlinear_dynamic, "CUDA")
# the unpacked form (a QuantizedCUDA weight tensor): registered CompositeImplicitAutograd so that the quantized dispatch key of that
# argument does not hide the implementation
_lib.define("cutlass_qlinear_dynamic_unpacked(Tensor input, Tensor weight, Tensor? bias=None) -> Tensor")
_lib.impl("cutlass_qlinear_dynamic_unpacked", cutlass_qlinear_dynamic_unpacked, "CompositeImplicitAutograd")

_qlib = torch.library.Library("quantized", "IMPL")
for _key in ("CUDA", "QuantizedCUDA"):                      # TORCH_LIBRARY_IMPL(quantized, {QuantizedCUDA, CUDA}) -- cutlass_qlinear.cc:73-81
    _qlib.impl("linear_dynamic", cutlass_qlinear_dynamic, _key)
_qlib.impl("linear_prepack", quantized_linear_prepack_cuda, "QuantizedCUDA")
