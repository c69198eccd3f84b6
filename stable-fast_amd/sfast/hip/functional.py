"""Tensor-level wrappers over the C ABI (include/sfast_hip.h).

Ownership / stream / error conventions mirror the reference operators
(reference src/sfast/csrc/operators/cutlass/cutlass_dual_linear_kernel.cu:309-316,
:364-365): outputs are fresh tensors from the torch caching allocator, inputs are borrowed and never
mutated, work is enqueued on the CURRENT stream of the input's device (so calls are capturable
into a hipGraph), errors surface as RuntimeError. PyTorch is only the allocator / stream provider
here -- every byte of arithmetic runs in libsfast_hip.so.
"""
import ctypes as C
import os
import functools
from typing import Optional, Sequence

import torch

from . import lib as L

_DT = {torch.float16: L.F16, torch.bfloat16: L.BF16, torch.float32: L.F32}
_ACT = {None: L.ACT_NONE, "none": L.ACT_NONE, "identity": L.ACT_NONE, "relu": L.ACT_RELU,
        "gelu": L.ACT_GELU, "gelu_tanh": L.ACT_GELU_TANH, "silu": L.ACT_SILU,
        "sigmoid": L.ACT_SIGMOID, "tanh": L.ACT_TANH}


def _on_device(fn):
    """Run `fn` with the first tensor argument's device current (the reference's ops run under a DeviceGuard,
    cutlass_dual_linear_kernel.cu:364-365): the launch stream, the per-device kernel attributes and the workspace
    allocation all follow the INPUT's device, not whatever device the caller happens to have selected."""
    @functools.wraps(fn)
    def wrapper(first, *args, **kwargs):
        dev = first.device if torch.is_tensor(first) else None
        if dev is not None and dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
            with torch.cuda.device(dev):
                return fn(first, *args, **kwargs)
        return fn(first, *args, **kwargs)
    return wrapper


def _dtype(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise L.SfastHipError(f"sfast HIP kernels support f16/bf16/f32, got {t.dtype}")


def _require_cuda(*ts):
    for t in ts:
        if t is not None and t.device.type != "cuda":
            raise L.SfastHipError(
                "sfast HIP operators need tensors on a ROCm device (got %s); there is no CPU path" % t.device)


def _act(a):
    if isinstance(a, int):
        return a
    return _ACT[a]


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _ws(nbytes, like):
    if nbytes == 0:
        return None, 0
    buf = torch.empty(nbytes, dtype=torch.uint8, device=like.device)
    return buf, nbytes


# SFAST_SPLITK_JOIN=1: split-K problems of linear() / conv2d() finish inside the GEMM kernel (sfast_hip.h SFAST_EXT_WS_TICKETS) instead of
# a separate reduce kernel. Off by default: measured on the SD1.5 step it does not pay (profiles/r03_splitk_join_*).
SPLITK_JOIN = os.environ.get("SFAST_SPLITK_JOIN", "0") not in ("0", "false", "off", "")


def _ws_tickets(nbytes, like):
    """Workspace of a GEMM / conv call whose split-K problems finish inside the kernel (sfast_hip.h SFAST_EXT_WS_TICKETS): the
    ticket block at its end is zeroed here, once per (fresh) workspace. Returns (buffer, bytes, ext flags)."""
    if nbytes == 0:
        return None, 0, 0
    if not SPLITK_JOIN:
        return _ws(nbytes, like) + (0,)
    nbytes = (int(nbytes) + 3) // 4 * 4
    buf = torch.empty(nbytes, dtype=torch.uint8, device=like.device)
    L.check(L.load().sfast_hip_workspace_init(buf.data_ptr(), nbytes, _stream(like)), "sfast_hip_workspace_init")
    return buf, nbytes, L.EXT_WS_TICKETS


def _i64x4(vals):
    return (C.c_int64 * 4)(*[int(v) for v in vals])


def _i64x3(vals):
    return (C.c_int64 * 3)(*[int(v) for v in vals])


# --------------------------------------------------------------------------------------------------
@_on_device
def group_norm(x, num_groups, weight=None, bias=None, eps=1e-5, act=None, x2=None):
    """GroupNorm(+SiLU). x: [N, C, *] (channels_last 4-D is processed natively as NHWC).
    x2: optional second channels_last tensor, normalised as if torch.cat([x, x2], 1)."""
    _require_cuda(x, x2, weight, bias)
    lib = L.init_device()
    if x.ndim < 2:
        raise L.SfastHipError("group_norm: input must be at least 2-D")
    N, C1 = x.shape[0], x.shape[1]
    # tensors that are both NCHW- and NHWC-contiguous (C == 1 or H*W == 1) have one memory image
    nhwc = x.ndim == 4 and x.is_contiguous(memory_format=torch.channels_last)
    if x2 is not None:
        if not (nhwc and x2.ndim == 4 and x2.is_contiguous(memory_format=torch.channels_last)):
            raise L.SfastHipError("group_norm: virtual concat needs two channels_last 4-D tensors")
        if x2.shape[0] != N or x2.shape[2:] != x.shape[2:] or x2.dtype != x.dtype:
            raise L.SfastHipError("group_norm: x2 shape/dtype mismatch")
        Ctot = C1 + x2.shape[1]
    else:
        Ctot = C1
    if not nhwc:
        x = x.contiguous()
    HW = 1
    for s in x.shape[2:]:
        HW *= s
    if Ctot % num_groups != 0:
        raise L.SfastHipError(f"group_norm: {Ctot} channels not divisible by {num_groups} groups")
    if weight is not None:
        weight = weight.to(x.dtype).contiguous()
    if bias is not None:
        bias = bias.to(x.dtype).contiguous()
    out_shape = (N, Ctot) + tuple(x.shape[2:])
    if nhwc:
        y = torch.empty(out_shape, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    else:
        y = torch.empty(out_shape, dtype=x.dtype, device=x.device)
    if y.numel() == 0:
        return y  # empty batch / empty spatial extent: nothing to launch (torch semantics)
    p = L.GnParams(_dtype(x), L.NHWC if nhwc else L.NCHW, N, Ctot, HW, num_groups, C1, _act(act), float(eps))
    nb = lib.sfast_hip_group_norm_workspace_bytes(C.byref(p))
    ws, nb = _ws(nb, x)
    rc = lib.sfast_hip_group_norm(_ptr(x), _ptr(x2), _ptr(weight), _ptr(bias), _ptr(y), C.byref(p),
                                  _ptr(ws), nb, _stream(x))
    L.check(rc, "sfast_hip_group_norm")
    return y


@_on_device
def group_norm_apply(x, num_groups, weight, bias, eps, act, stats1, lay1, x2=None, stats2=None, lay2=None):
    """GroupNorm(+SiLU) of a channels_last tensor (optionally a virtual concat x | x2) from producer-emitted statistics."""
    _require_cuda(x, weight, bias, x2, stats1, stats2)
    lib = L.init_device()
    if not x.is_contiguous(memory_format=torch.channels_last) or (x2 is not None and not x2.is_contiguous(memory_format=torch.channels_last)):
        raise L.SfastHipError("group_norm_apply: channels_last inputs required")
    N, C1, H, W = x.shape
    Ctot = C1 + (x2.shape[1] if x2 is not None else 0)
    y = torch.empty((N, Ctot, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    p = L.GnParams(_dtype(x), L.NHWC, N, Ctot, H * W, int(num_groups), C1, _act(act), float(eps))
    w = weight.to(x.dtype).contiguous() if weight is not None else None
    b = bias.to(x.dtype).contiguous() if bias is not None else None
    L.check(lib.sfast_hip_group_norm_apply(_ptr(x), _ptr(x2), _ptr(w), _ptr(b), _ptr(y), C.byref(p), _ptr(stats1), C.byref(lay1), _ptr(stats2),
                                           C.byref(lay2) if lay2 is not None else None, _stream(x)), "sfast_hip_group_norm_apply")
    return y


@_on_device
def layer_norm(x, normalized_shape: Sequence[int], weight=None, bias=None, eps=1e-5):
    _require_cuda(x, weight, bias)
    lib = L.init_device()
    n = 1
    for s in normalized_shape:
        n *= int(s)
    if tuple(x.shape[x.ndim - len(normalized_shape):]) != tuple(int(s) for s in normalized_shape):
        raise L.SfastHipError("layer_norm: normalized_shape does not match the trailing dims")
    x = x.contiguous()
    m = x.numel() // n
    if weight is not None:
        weight = weight.to(x.dtype).contiguous()
    if bias is not None:
        bias = bias.to(x.dtype).contiguous()
    y = torch.empty_like(x)
    if y.numel() == 0:
        return y
    p = L.LnParams(_dtype(x), m, n, float(eps))
    rc = lib.sfast_hip_layer_norm(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), C.byref(p), _stream(x))
    L.check(rc, "sfast_hip_layer_norm")
    return y


@_on_device
def image_postprocess(image, denormalize=True, to_uint8=True):
    """NCHW image tensor -> NHWC uint8 (round(255 * x)) or float32, optionally denormalised from [-1, 1] first."""
    _require_cuda(image)
    lib = L.init_device()
    if image.ndim != 4:
        raise L.SfastHipError("image_postprocess: need a [B, C, H, W] tensor")
    image = image.contiguous()
    B, Cc, H, W = image.shape
    out = torch.empty((B, H, W, Cc), dtype=torch.uint8 if to_uint8 else torch.float32, device=image.device)
    p = L.ImageParams(_dtype(image), B, Cc, H, W, int(bool(denormalize)), int(bool(to_uint8)))
    rc = lib.sfast_hip_image_postprocess(_ptr(image), _ptr(out), C.byref(p), _stream(image))
    L.check(rc, "sfast_hip_image_postprocess")
    return out


@_on_device
def softmax_rows(x, scale=1.0, out=None):
    """softmax(scale * x) over the last dim of a 2-D f16/bf16 tensor whose rows are 16-byte aligned (fp32 math)."""
    _require_cuda(x)
    lib = L.init_device()
    if x.ndim != 2 or x.stride(1) != 1:
        raise L.SfastHipError("softmax_rows: need a 2-D tensor with contiguous rows")
    y = out if out is not None else torch.empty((x.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
    p = L.SoftmaxParams(_dtype(x), x.shape[0], x.shape[1], x.stride(0), y.stride(0), float(scale))
    rc = lib.sfast_hip_softmax_rows(_ptr(x), _ptr(y), C.byref(p), _stream(x))
    L.check(rc, "sfast_hip_softmax_rows")
    return y


def _check_packed(who, lib, pk, rows, K, like):
    """A packed copy handed to linear() / conv2d() is indexed by the pipe-4 kernels up to ceil(rows / 32) * ceil(K / 64) * 4 KB
    unconditionally: a buffer of another size, dtype or device would be read out of bounds on the GPU. (That it was packed from THIS
    weight cannot be checked here; pack_weight() is the only producer.)"""
    need = int(lib.sfast_hip_packed_weight_bytes(int(rows), int(K)))
    if not (torch.is_tensor(pk) and pk.dtype == torch.uint8 and pk.device == like.device and pk.is_contiguous() and pk.numel() == need):
        got = (tuple(pk.shape), pk.dtype, str(pk.device)) if torch.is_tensor(pk) else type(pk).__name__
        raise L.SfastHipError(f"{who}: w_packed must be the uint8 tensor pack_weight() returns for this [{rows}, {K}] weight "
                              f"({need} bytes on {like.device}), got {got}")


@_on_device
def pack_weight(weight):
    """Packed copy of a [N, K] linear weight or a [Cout, Cin, KH, KW] channels_last conv weight for the pipe-4 kernels
    (sfast_hip_pack_weight: contiguous 1 KB MFMA A-fragments). Pass it as `w_packed=` to linear() / conv2d(); re-pack after the
    weight changed."""
    _require_cuda(weight)
    lib = L.init_device()
    if weight.ndim == 4:
        Cout, Cin, KH, KW = weight.shape
        if not (weight.stride(1) == 1 and (KW == 1 or weight.stride(3) == Cin) and (KH == 1 or weight.stride(2) == KW * Cin)):
            raise L.SfastHipError("pack_weight: conv weight must be [Cout][KH][KW][Cin]-contiguous (channels_last)")
        N, K, ldw = Cout, KH * KW * Cin, weight.stride(0) if Cout > 1 else KH * KW * Cin
    elif weight.ndim == 2 and weight.stride(1) == 1:
        N, K = weight.shape
        ldw = weight.stride(0) if N > 1 else K
    else:
        raise L.SfastHipError("pack_weight: a K-contiguous [N, K] or channels_last [Cout, Cin, KH, KW] weight is required")
    out = torch.empty(lib.sfast_hip_packed_weight_bytes(N, K), dtype=torch.uint8, device=weight.device)
    L.check(lib.sfast_hip_pack_weight(_ptr(weight), _ptr(out), N, K, ldw, _dtype(weight), _stream(weight)), "sfast_hip_pack_weight")
    return out


@_on_device
def linear(x, weight, bias=None, *, act=None, residual=None, alpha=1.0, res_before_act=False,
           geglu=False, rowbias=None, rows_per_batch=0, in_act=None, variant=0, split_k=0, out=None, out_scale=1.0, gn_unit=0,
           rows_per_sample=0, w_packed=None):
    """out[..., N] = epilogue(x[..., K] @ W[N, K]^T). `weight` may be a list of <= 4 equally sized
    [n_i, K] tensors stacked along N (e.g. live to_q / to_k / to_v weights). `w_packed`: pack_weight() of every segment (a tensor
    or a list) -- makes the pipe-4 kernels (variant 41 ..) eligible."""
    ws_list = list(weight) if isinstance(weight, (list, tuple)) else [weight]
    _require_cuda(x, bias, residual, rowbias, *ws_list)
    lib = L.init_device()
    K = x.shape[-1]
    for w in ws_list:
        if w.ndim != 2 or w.shape[1] != K or w.dtype != x.dtype:
            raise L.SfastHipError(f"linear: weight {tuple(w.shape)}/{w.dtype} incompatible with input [..., {K}]/{x.dtype}")
        if w.shape[0] != ws_list[0].shape[0]:
            raise L.SfastHipError("linear: stacked weight segments must have equal row counts")
    ws_list = [w if (w.stride(1) == 1 and w.stride(0) >= K) else w.contiguous() for w in ws_list]
    ldw = ws_list[0].stride(0) if ws_list[0].shape[0] > 1 else K
    for w in ws_list:
        if (w.stride(0) if w.shape[0] > 1 else ldw) != ldw:
            raise L.SfastHipError("linear: stacked weight segments must share a row stride")
    rows = ws_list[0].shape[0] * len(ws_list)
    if geglu:
        if rows % 2:
            raise L.SfastHipError("linear: geglu needs an even number of weight rows")
        N = rows // 2
    else:
        N = rows
    lead = x.shape[:-1]
    x2d = x.reshape(-1, K)
    if x2d.stride(-1) != 1 or (x2d.shape[0] > 1 and x2d.stride(0) < K):
        x2d = x2d.contiguous()
    M = x2d.shape[0]
    ldx = x2d.stride(0) if M > 1 else K
    if out is None:
        out2d = torch.empty((M, N), dtype=x.dtype, device=x.device)
    else:
        out2d = out.reshape(M, N)
        if out2d.data_ptr() != out.data_ptr():
            raise L.SfastHipError("linear: `out` must be viewable as [M, N]")
    if M == 0 and not gn_unit:
        return out2d.reshape(*lead, N) if out is None else out  # zero rows: nothing to launch
    ldo = out2d.stride(0) if M > 1 else N
    res2d = None
    ldr = 0
    if residual is not None:
        res2d = residual.expand(*lead, N).reshape(M, N) if residual.shape != (M, N) else residual
        if res2d.stride(-1) != 1:
            res2d = res2d.contiguous()
        ldr = res2d.stride(0) if M > 1 else N
        if res2d.dtype != x.dtype:
            res2d = res2d.to(x.dtype)
    if bias is not None:
        bias = bias.to(x.dtype).contiguous()
    ld_rb = 0
    if rowbias is not None:
        rowbias = rowbias.to(x.dtype)
        if rowbias.stride(-1) != 1:
            rowbias = rowbias.contiguous()
        ld_rb = rowbias.stride(0) if rowbias.shape[0] > 1 else N
    p = L.GemmParams()
    p.dtype, p.M, p.N, p.K = _dtype(x), M, N, K
    p.ldx, p.ldw, p.ldo, p.ldr = ldx, ldw, ldo, ldr
    p.n_wseg, p.rows_per_seg = len(ws_list), ws_list[0].shape[0]
    p.geglu = 1 if geglu else 0
    p.act, p.res_before_act, p.alpha = _act(act), 1 if res_before_act else 0, float(alpha)
    p.rows_per_batch, p.ld_rowbias = int(rows_per_batch), ld_rb
    p.in_act, p.variant, p.split_k = _act(in_act), int(variant), int(split_k)
    segs = (C.c_void_p * len(ws_list))(*[w.data_ptr() for w in ws_list])
    nb = lib.sfast_hip_gemm_workspace_bytes(C.byref(p))
    wsb, nb, flags = _ws_tickets(nb, x)
    ext = L.EpilogueExt(float(out_scale), int(gn_unit), int(rows_per_sample), flags)
    if w_packed is not None:
        pk_list = list(w_packed) if isinstance(w_packed, (list, tuple)) else [w_packed]
        if len(pk_list) != len(ws_list):
            raise L.SfastHipError("linear: one packed copy per weight segment is required")
        for t in pk_list:
            _check_packed("linear", lib, t, ws_list[0].shape[0], K, x)
        pk_arr = (C.c_void_p * len(pk_list))(*[t.data_ptr() for t in pk_list])
        ext.w_packed = C.cast(pk_arr, C.c_void_p)
    stats, lay = None, None
    if gn_unit:
        lay = L.GnStatsLayout()
        L.check(lib.sfast_hip_gemm_stats_layout(C.byref(p), C.byref(ext), C.byref(lay)), "sfast_hip_gemm_stats_layout")
        stats = torch.full((lay.nbytes() // 4,), float("nan"), dtype=torch.float32, device=x.device)
    rc = lib.sfast_hip_gemm_ex(_ptr(x2d), segs, _ptr(bias), _ptr(rowbias), _ptr(res2d), _ptr(out2d), C.byref(p), C.byref(ext), _ptr(stats),
                               _ptr(wsb), nb, _stream(x))
    L.check(rc, "sfast_hip_gemm")
    res = out2d.reshape(*lead, N) if out is None else out
    if gn_unit:
        return res, stats, lay
    return res


@_on_device
def linear_grouped(x, weight_groups, biases=None, act=None, outs=None):
    """[act(x_g @ cat(W_g)^T + b_g) for g]: every group is a list of <= 2 equally sized [n, K] weights stacked along N; all groups
    have the same shape. `x`: one [M, K] tensor shared by all groups, or a list of per-group [M, K] tensors. Launches of up to
    64 groups each (sfast_hip_gemm_grouped). Returns a list of [M, N] tensors. `outs`: optional preallocated dense [M, N] outputs, one
    per group (e.g. the rows `out[b]` of ONE [B, M, N] tensor: a batched matmul lands in place, no stack)."""
    groups = [list(g) if isinstance(g, (list, tuple)) else [g] for g in weight_groups]
    flat = [w for g in groups for w in g]
    xs = list(x) if isinstance(x, (list, tuple)) else [x] * len(groups)
    _require_cuda(*xs, *flat)
    lib = L.init_device()
    xs2 = []
    for t in xs:
        t2 = t.reshape(-1, t.shape[-1])
        xs2.append(t2 if t2.stride(-1) == 1 and (t2.shape[0] == 1 or t2.stride(0) == xs[0].reshape(-1, xs[0].shape[-1]).stride(0)) else t2.contiguous())
    if any(t.stride(0) != xs2[0].stride(0) for t in xs2 if t.shape[0] > 1):
        xs2 = [t.contiguous() for t in xs2]
    M, K = xs2[0].shape
    nseg, rows = len(groups[0]), groups[0][0].shape[0]
    flat = [w.contiguous() for w in flat]
    N = nseg * rows
    if outs is None:
        outs = [torch.empty((M, N), dtype=xs2[0].dtype, device=xs2[0].device) for _ in groups]
    else:
        outs = list(outs)
        if len(outs) != len(groups) or any(tuple(o.shape) != (M, N) or not o.is_contiguous() or o.dtype != xs2[0].dtype for o in outs):
            raise L.SfastHipError("linear_grouped: `outs` must be one dense [M, N] tensor of the input dtype per group")
    bs = None
    if biases is not None:
        bs = [None if b is None else b.to(xs2[0].dtype).contiguous() for b in biases]
    p = L.GemmParams()
    p.dtype, p.M, p.N, p.K = _dtype(xs2[0]), M, N, K
    p.ldx, p.ldw, p.ldo, p.ldr = xs2[0].stride(0) if M > 1 else K, K, N, 0
    p.n_wseg, p.rows_per_seg, p.act, p.alpha = nseg, rows, _act(act), 1.0
    for g0 in range(0, len(groups), L.MAX_GEMM_GROUPS):
        g1 = min(len(groups), g0 + L.MAX_GEMM_GROUPS)
        n = g1 - g0
        xp = (C.c_void_p * n)(*[t.data_ptr() for t in xs2[g0:g1]])
        wp = (C.c_void_p * (n * nseg))(*[w.data_ptr() for w in flat[g0 * nseg:g1 * nseg]])
        op = (C.c_void_p * n)(*[o.data_ptr() for o in outs[g0:g1]])
        bp = (C.c_void_p * n)(*[_ptr(b) for b in bs[g0:g1]]) if bs is not None else None
        L.check(lib.sfast_hip_gemm_grouped(xp, wp, bp, op, C.byref(p), n, _stream(xs2[0])), "sfast_hip_gemm_grouped")
    return outs


@_on_device
def gemv_grouped(x, weights, biases=None, act=None, in_act=None):
    """out[m, off_g + n] = act(in_act(x)[m] @ W_g[n] + b_g[n]) for every W_g of `weights` ([n_g, K] each, <= 32), ONE launch.
    Returns [M, sum n_g]."""
    weights = list(weights)
    if not 1 <= len(weights) <= L.MAX_GROUPS:
        raise L.SfastHipError(f"gemv_grouped: {len(weights)} groups (1..{L.MAX_GROUPS} per launch)")
    biases = list(biases) if biases is not None else [None] * len(weights)
    _require_cuda(x, *weights, *[b for b in biases if b is not None])
    lib = L.init_device()
    if x.ndim != 2 or x.stride(1) != 1:
        x = x.reshape(-1, x.shape[-1]).contiguous()
    M, K = x.shape
    ws = [w if (w.stride(1) == 1 and w.stride(0) == weights[0].stride(0)) else w.contiguous() for w in weights]
    if any(w.stride(0) != ws[0].stride(0) for w in ws):
        ws = [w.contiguous() for w in ws]
    bs = [None if b is None else b.to(x.dtype).contiguous() for b in biases]
    ntot = sum(w.shape[0] for w in ws)
    out = torch.empty((M, ntot), dtype=x.dtype, device=x.device)
    p = L.GemvGroupedParams()
    p.dtype, p.M, p.K, p.n_groups = _dtype(x), M, K, len(ws)
    for i, w in enumerate(ws):
        p.n_rows[i] = w.shape[0]
    p.ldx, p.ldw, p.ldo = x.stride(0) if M > 1 else K, ws[0].stride(0), ntot
    p.act, p.in_act = _act(act), _act(in_act)
    wp = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * len(ws))(*[_ptr(b) for b in bs])
    L.check(lib.sfast_hip_gemv_grouped(_ptr(x), wp, bp, _ptr(out), C.byref(p), _stream(x)), "sfast_hip_gemv_grouped")
    return out


def _nhwc_strides(t):
    # logical NCHW tensor -> element strides in (n, h, w, c) order
    return (t.stride(0), t.stride(2), t.stride(3), t.stride(1))


@_on_device
def conv2d(x, weight, bias=None, *, z=None, alpha=1.0, stride=1, padding=0, dilation=1, act=None,
           res_before_act=True, x2=None, upsample2x=False, rowbias=None, variant=0, split_k=0,
           channels_last_out: Optional[bool] = None, pad_extra=0, out_scale=1.0, gn_unit=0, out=None, gn=None, w_packed=None):
    """y = act(conv2d(x, w) + bias + rowbias[b] + alpha*z) on logical NCHW tensors of any strides.
    x2: optional tensor concatenated to x along channels (virtual). upsample2x: nearest 2x first.
    pad_extra: additional zero rows / columns at the bottom / right on top of `padding` (F.pad(x, (0, e, 0, e))).
    out_scale: accumulator scale (sfast_epilogue_ext). gn_unit > 0: also emit GroupNorm partial statistics of y; returns
    (y, stats float32 tensor, GnStatsLayout) -- feed them to group_norm_apply().
    out: optional preallocated [B, Cout, Ho, Wo] tensor of ANY strides (e.g. a channel slice of a larger NHWC tensor: the groups of
    a grouped convolution write their slices of one output, no concatenation).
    gn = (num_groups, weight, bias, eps, act): ALSO return act(GroupNorm(y)) -- computed by the split-K reduce launch of this conv
    (sfast_epilogue_ext.gn_out; split-K plans only: pass split_k > 1 or let the planner choose one) -> (y, n), both channels_last."""
    _require_cuda(x, weight, bias, z, x2, rowbias)
    lib = L.init_device()
    if x.ndim != 4 or weight.ndim != 4:
        raise L.SfastHipError("conv2d: 4-D input and weight required")
    pair = lambda v: (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))
    sh, sw = pair(stride)
    ph, pw = pair(padding)
    dh, dw = pair(dilation)
    B, C1, H, W = x.shape
    Cin = C1 + (x2.shape[1] if x2 is not None else 0)
    Cout, Cw, KH, KW = weight.shape
    if Cw != Cin:
        raise L.SfastHipError(f"conv2d: weight expects {Cw} input channels, got {Cin} (groups != 1 unsupported)")
    if weight.dtype != x.dtype:
        raise L.SfastHipError("conv2d: input / weight dtype mismatch")
    Hin, Win = (2 * H, 2 * W) if upsample2x else (H, W)
    eh, ew = pair(pad_extra)
    Ho = (Hin + 2 * ph + eh - dh * (KH - 1) - 1) // sh + 1
    Wo = (Win + 2 * pw + ew - dw * (KW - 1) - 1) // sw + 1
    if channels_last_out is None:
        # cudnn_conv_suggest_memory_format semantics (reference cudnn_convolution_impl.cc:1023-1025)
        def _is_cl(t):
            return t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()
        channels_last_out = _is_cl(x) or _is_cl(weight)
    fmt = torch.channels_last if channels_last_out else torch.contiguous_format
    if out is not None:
        if tuple(out.shape) != (B, Cout, Ho, Wo) or out.dtype != x.dtype or out.device != x.device:
            raise L.SfastHipError(f"conv2d: `out` must be a {(B, Cout, Ho, Wo)} tensor of the input dtype on its device")
        y = out
    else:
        y = torch.empty((B, Cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=fmt)
    if y.numel() == 0 and not gn_unit:
        return y  # empty batch: nothing to launch
    zz = None
    if z is not None:
        zz = z.to(x.dtype).expand(B, Cout, Ho, Wo)
    if bias is not None:
        bias = bias.to(x.dtype).contiguous()
    ld_rb = 0
    if rowbias is not None:
        rowbias = rowbias.to(x.dtype)
        if rowbias.stride(-1) != 1:
            rowbias = rowbias.contiguous()
        ld_rb = rowbias.stride(0) if rowbias.shape[0] > 1 else Cout
    p = L.ConvParams()
    p.dtype, p.B, p.H, p.W, p.Cin, p.Cout, p.KH, p.KW = _dtype(x), B, H, W, Cin, Cout, KH, KW
    p.stride_h, p.stride_w, p.pad_h, p.pad_w, p.dil_h, p.dil_w = sh, sw, ph, pw, dh, dw
    p.pad_h_extra, p.pad_w_extra = eh, ew
    p.upsample2x, p.C1 = 1 if upsample2x else 0, C1
    p.xs = _i64x4(_nhwc_strides(x))
    p.x2s = _i64x4(_nhwc_strides(x2) if x2 is not None else (0, 0, 0, 0))
    p.ws = _i64x4((weight.stride(0), weight.stride(1), weight.stride(2), weight.stride(3)))
    p.os = _i64x4(_nhwc_strides(y))
    p.zs = _i64x4(_nhwc_strides(zz) if zz is not None else (0, 0, 0, 0))
    p.act, p.res_before_act, p.alpha = _act(act), 1 if res_before_act else 0, float(alpha)
    p.ld_rowbias, p.variant, p.split_k = ld_rb, int(variant), int(split_k)
    nb = lib.sfast_hip_conv2d_workspace_bytes(C.byref(p))
    wsb, nb, flags = _ws_tickets(nb, x)
    ext = L.EpilogueExt(float(out_scale), int(gn_unit), Ho * Wo, flags)
    if w_packed is not None:  # pack_weight(weight): the pipe-4 kernels (variant 41 ..) become eligible
        _check_packed("conv2d", lib, w_packed, p.Cout, p.KH * p.KW * p.Cin, x)
        if not (weight.stride(1) == 1 and (p.KW == 1 or weight.stride(3) == p.Cin) and (p.KH == 1 or weight.stride(2) == p.KW * p.Cin)):
            raise L.SfastHipError("conv2d: w_packed needs the channels_last ([Cout][KH][KW][Cin]) weight it was packed from")
        pk_arr = (C.c_void_p * 1)(w_packed.data_ptr())
        ext.w_packed = C.cast(pk_arr, C.c_void_p)
    gn_y = None
    if gn is not None:
        groups, g_w, g_b, g_eps, g_act = gn
        gn_y = torch.empty((B, Cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        g_w = None if g_w is None else g_w.to(x.dtype).contiguous()
        g_b = None if g_b is None else g_b.to(x.dtype).contiguous()
        ext.gn_out, ext.gn_gamma, ext.gn_beta = gn_y.data_ptr(), _ptr(g_w), _ptr(g_b)
        ext.gn_groups, ext.gn_eps, ext.gn_act = int(groups), float(g_eps), _act(g_act)
    stats, lay = None, None
    if gn_unit:
        lay = L.GnStatsLayout()
        L.check(lib.sfast_hip_conv2d_stats_layout(C.byref(p), C.byref(ext), C.byref(lay)), "sfast_hip_conv2d_stats_layout")
        stats = torch.full((lay.nbytes() // 4,), float("nan"), dtype=torch.float32, device=x.device)
    rc = lib.sfast_hip_conv2d_ex(_ptr(x), _ptr(x2), _ptr(weight), _ptr(bias), _ptr(rowbias), _ptr(zz), _ptr(y),
                                 C.byref(p), C.byref(ext), _ptr(stats), _ptr(wsb), nb, _stream(x))
    L.check(rc, "sfast_hip_conv2d")
    if gn is not None:
        return y, gn_y
    if gn_unit:
        return y, stats, lay
    return y


def _gn_conv_params(x, x2, weight, groups, eps, gn_act, z, alpha, act, res_before_act, rowbias, y):
    B, C1, H, W = x.shape
    Cin = C1 + (x2.shape[1] if x2 is not None else 0)
    Cout = weight.shape[0]
    q = L.GnConvParams()
    p = q.conv
    p.dtype, p.B, p.H, p.W, p.Cin, p.Cout, p.KH, p.KW = _dtype(x), B, H, W, Cin, Cout, 3, 3
    p.stride_h = p.stride_w = p.pad_h = p.pad_w = p.dil_h = p.dil_w = 1
    p.upsample2x, p.C1 = 0, C1
    p.xs = _i64x4(_nhwc_strides(x))
    p.x2s = _i64x4(_nhwc_strides(x2) if x2 is not None else (0, 0, 0, 0))
    p.ws = _i64x4((weight.stride(0), weight.stride(1), weight.stride(2), weight.stride(3)))
    p.os = _i64x4(_nhwc_strides(y))
    p.zs = _i64x4(_nhwc_strides(z) if z is not None else (0, 0, 0, 0))
    p.act, p.res_before_act, p.alpha = _act(act), 1 if res_before_act else 0, float(alpha)
    p.ld_rowbias = 0 if rowbias is None else (rowbias.stride(0) if rowbias.shape[0] > 1 else Cout)
    q.groups, q.eps, q.gn_act = int(groups), float(eps), _act(gn_act)
    return q


@_on_device
def gn_conv2d_supported(x, weight, groups, x2=None):
    """Whether sfast_hip_gn_conv2d (GroupNorm(+SiLU) -> 3x3 conv as one weight-streaming launch) covers this problem."""
    lib = L.init_device()
    if x.ndim != 4 or weight.ndim != 4 or tuple(weight.shape[2:]) != (3, 3):
        return False
    y = torch.empty((x.shape[0], weight.shape[0], x.shape[2], x.shape[3]), dtype=x.dtype, device="meta").contiguous(memory_format=torch.channels_last)
    q = _gn_conv_params(x, x2, weight, groups, 1e-5, "silu", None, 1.0, None, True, None, y)
    return bool(lib.sfast_hip_gn_conv2d_supported(C.byref(q)))


@_on_device
def gn_conv2d(x, num_groups, gn_weight, gn_bias, weight, bias=None, *, eps=1e-5, gn_act="silu", x2=None, z=None, alpha=1.0, act=None,
              res_before_act=True, rowbias=None):
    """y = act(conv3x3(gn_act(GroupNorm(cat(x, x2)))) + bias + rowbias[b] + alpha * z), stride 1 / padding 1, ONE weight-streaming launch
    (+ the split-K reduce that carries the epilogue): sfast_hip_gn_conv2d. channels_last tensors; raises when the problem is outside
    the fused kernel's coverage (`gn_conv2d_supported`) -- callers then run group_norm() and conv2d()."""
    _require_cuda(x, x2, gn_weight, gn_bias, weight, bias, z, rowbias)
    lib = L.init_device()
    B, C1, H, W = x.shape
    Cout = weight.shape[0]
    y = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    zz = None if z is None else z.to(x.dtype).expand(B, Cout, H, W)
    if bias is not None:
        bias = bias.to(x.dtype).contiguous()
    if rowbias is not None:
        rowbias = rowbias.to(x.dtype)
        if rowbias.stride(-1) != 1:
            rowbias = rowbias.contiguous()
    gw = None if gn_weight is None else gn_weight.to(x.dtype).contiguous()
    gb = None if gn_bias is None else gn_bias.to(x.dtype).contiguous()
    q = _gn_conv_params(x, x2, weight, num_groups, eps, gn_act, zz, alpha, act, res_before_act, rowbias, y)
    nb = lib.sfast_hip_gn_conv2d_workspace_bytes(C.byref(q))
    ws, nb = _ws(nb, x)
    rc = lib.sfast_hip_gn_conv2d(_ptr(x), _ptr(x2), _ptr(gw), _ptr(gb), _ptr(weight), _ptr(bias), _ptr(rowbias), _ptr(zz), _ptr(y),
                                 C.byref(q), _ptr(ws), nb, _stream(x))
    L.check(rc, "sfast_hip_gn_conv2d")
    return y


@_on_device
def attention(q, k, v, scale: Optional[float] = None, variant=0, attn_bias=None):
    """softmax(q k^T * scale + attn_bias) v with q [B, Sq, H, D], k/v [B, Skv, H, D] (any b/s/h strides).
    attn_bias: additive, broadcastable to [B, H, Sq, Skv] (xformers' tensor-bias form); -inf masks a key."""
    _require_cuda(q, k, v, attn_bias)
    lib = L.init_device()
    if q.ndim != 4 or k.ndim != 4 or v.ndim != 4:
        raise L.SfastHipError("attention: expected [B, S, H, D] tensors")
    B, Sq, H, D = q.shape
    Skv = k.shape[1]
    if k.shape != (B, Skv, H, D) or v.shape != (B, Skv, H, D) or k.dtype != q.dtype or v.dtype != q.dtype:
        raise L.SfastHipError("attention: q/k/v shape or dtype mismatch")
    q, k, v = [t if t.stride(-1) == 1 else t.contiguous() for t in (q, k, v)]
    out = torch.empty((B, Sq, H, D), dtype=q.dtype, device=q.device)
    if out.numel() == 0:
        return out  # no queries
    if Skv == 0:
        raise L.SfastHipError("attention: no keys (softmax over an empty set is undefined)")
    p = L.AttnParams()
    p.dtype, p.B, p.H, p.Sq, p.Skv, p.D = _dtype(q), B, H, Sq, Skv, D
    p.qs = _i64x3(q.stride()[:3])
    p.ks = _i64x3(k.stride()[:3])
    p.vs = _i64x3(v.stride()[:3])
    p.os = _i64x3(out.stride()[:3])
    p.scale = float(scale) if scale is not None else float(D) ** -0.5
    p.variant = int(variant)
    bptr, bstr = None, None
    if attn_bias is not None:
        bias = attn_bias
        if bias.dtype != q.dtype:
            bias = bias.to(q.dtype)
        while bias.ndim < 4:
            bias = bias.unsqueeze(0)
        if bias.shape[-1] == Skv:
            # the MFMA kernel reads 4-key groups with dword loads: rows must start at even element offsets and be readable up to the
            # next multiple of 4 keys. A bias that is not laid out that way (odd row strides, or a dense tensor whose last row ends
            # at an odd key count) is re-laid, un-broadcast, with rows padded to 8 elements -- what xformers demands of its callers
            odd = bias.stride(-1) != 1 or any(st % 2 for st, n in zip(bias.stride()[:3], bias.shape[:3]) if n > 1)
            last = bias.storage_offset() + sum((n - 1) * st for n, st in zip(bias.shape[:3], bias.stride()[:3])) + (Skv + 3) // 4 * 4
            short = last * bias.element_size() > bias.untyped_storage().nbytes()
            if odd or short or bias.storage_offset() % 2:
                padded = torch.zeros(bias.shape[:3] + ((Skv + 7) // 8 * 8,), dtype=bias.dtype, device=bias.device)
                padded[..., :Skv] = bias
                bias = padded[..., :Skv]
        try:
            bias = bias.expand(B, H, Sq, Skv)  # broadcast dims get stride 0: nothing is materialised
        except RuntimeError:
            raise L.SfastHipError(f"attention: attn_bias {tuple(attn_bias.shape)} is not broadcastable to {(B, H, Sq, Skv)}")
        if bias.stride(3) != 1 and Skv > 1:
            bias = bias.contiguous()
        bptr, bstr = bias.data_ptr(), _i64x3(bias.stride()[:3])
    rc = lib.sfast_hip_attention_bias(_ptr(q), _ptr(k), _ptr(v), bptr, bstr, _ptr(out), C.byref(p), _stream(q))
    L.check(rc, "sfast_hip_attention")
    return out


@_on_device
def strided_copy(src, dst):
    """dst[...] = src[...] for equal-shape tensors of rank <= 4 and arbitrary strides."""
    _require_cuda(src, dst)
    lib = L.init_device()
    if src.shape != dst.shape or src.dtype != dst.dtype:
        raise L.SfastHipError("strided_copy: shape/dtype mismatch")
    if src.ndim > 4:
        raise L.SfastHipError("strided_copy: rank <= 4 only")
    if src.numel() == 0:
        return dst
    nd = max(src.ndim, 1)
    shape = list(src.shape) if src.ndim else [1]
    ss = list(src.stride()) if src.ndim else [1]
    ds = list(dst.stride()) if dst.ndim else [1]
    p = L.CopyParams()
    p.elem_bytes, p.ndim = src.element_size(), nd
    p.shape = _i64x4(shape + [1] * (4 - nd))
    p.src_strides = _i64x4(ss + [0] * (4 - nd))
    p.dst_strides = _i64x4(ds + [0] * (4 - nd))
    rc = lib.sfast_hip_strided_copy(_ptr(src), _ptr(dst), C.byref(p), _stream(src))
    L.check(rc, "sfast_hip_strided_copy")
    return dst


@_on_device
def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, downscale_freq_shift=0.0, max_period=10000.0,
                       dtype=torch.float16):
    _require_cuda(timesteps)
    lib = L.init_device()
    t = timesteps.to(torch.float32).contiguous().reshape(-1)
    out = torch.empty((t.numel(), dim), dtype=dtype, device=t.device)
    p = L.TembParams(_DT[dtype], t.numel(), dim, 1 if flip_sin_to_cos else 0, float(downscale_freq_shift),
                     float(max_period))
    rc = lib.sfast_hip_timestep_embedding(_ptr(t), _ptr(out), C.byref(p), _stream(t))
    L.check(rc, "sfast_hip_timestep_embedding")
    return out


@_on_device
def cfg_ddim_step(eps_uc, latents, coef, guidance, latents_out=None, unet_in=None):
    _require_cuda(eps_uc, latents, coef)
    lib = L.init_device()
    numel = latents.numel()
    if eps_uc.numel() != 2 * numel or not eps_uc.is_contiguous() or not latents.is_contiguous():
        raise L.SfastHipError("cfg_ddim_step: eps_uc must be contiguous [2, *latents.shape]")
    if latents_out is None:
        latents_out = torch.empty_like(latents)
    rc = lib.sfast_hip_cfg_ddim_step(_ptr(eps_uc), _ptr(latents), _ptr(latents_out), _ptr(unet_in), _ptr(coef),
                                     float(guidance), numel, _dtype(latents), _stream(latents))
    L.check(rc, "sfast_hip_cfg_ddim_step")
    return latents_out


@_on_device
def linear_step(model_output, sample, coef, index=None, out=None):
    """out = coef[idx, 0] * sample + coef[idx, 1] * model_output (fp32 math). `coef` float32 [n, 2] on the device; `index`: None (row 0),
    a python int (resolved on the host: no sync) or a 0-d int32 / int64 tensor on the device (read by the kernel: no sync)."""
    _require_cuda(model_output, sample, coef)
    lib = L.init_device()
    if coef.dtype != torch.float32 or coef.ndim != 2 or coef.shape[1] != 2 or not coef.is_contiguous():
        raise L.SfastHipError("linear_step: coef must be a contiguous float32 [n, 2] tensor")
    mo = model_output.contiguous()
    x = sample.to(mo.dtype).contiguous()
    if out is None:
        out = torch.empty_like(mo)
    n = coef.shape[0]
    cptr, iptr, i64 = coef.data_ptr(), None, 0
    if index is not None:
        if torch.is_tensor(index) and index.device.type == "cuda":
            if index.dtype not in (torch.int32, torch.int64) or index.numel() != 1:
                raise L.SfastHipError("linear_step: device index must be one int32 / int64 element")
            iptr, i64 = index.data_ptr(), int(index.dtype == torch.int64)
        else:
            i = int(index)
            if not 0 <= i < n:
                raise L.SfastHipError(f"linear_step: index {i} outside the coefficient table [0, {n})")
            cptr, n = cptr + 8 * i, 1
    L.check(lib.sfast_hip_linear_step(_ptr(mo), _ptr(x), _ptr(out), cptr, iptr, i64, n, mo.numel(), _dtype(mo), _stream(mo)),
            "sfast_hip_linear_step")
    return out


@_on_device
def qlinear_w8(x, w_int8, scale, bias=None, act=None):
    """act(scale * (x @ w_int8^T) + bias): weight-only int8 linear (w_int8 [N, K] torch.int8, per-tensor `scale`)."""
    _require_cuda(x, w_int8, bias)
    lib = L.init_device()
    if w_int8.dtype != torch.int8 or w_int8.ndim != 2 or w_int8.shape[1] != x.shape[-1]:
        raise L.SfastHipError("qlinear_w8: weight must be an int8 [N, K] matrix matching the input's last dimension")
    w = w_int8 if w_int8.stride(1) == 1 else w_int8.contiguous()
    K = x.shape[-1]
    lead = x.shape[:-1]
    x2d = x.reshape(-1, K)
    if x2d.stride(-1) != 1:
        x2d = x2d.contiguous()
    M, N = x2d.shape[0], w.shape[0]
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    if bias is not None:
        bias = bias.to(x.dtype).contiguous()
    p = L.GemmParams()
    p.dtype, p.M, p.N, p.K = _dtype(x), M, N, K
    p.ldx, p.ldw, p.ldo = (x2d.stride(0) if M > 1 else K), (w.stride(0) if N > 1 else K), N
    p.n_wseg, p.rows_per_seg, p.act, p.alpha = 1, N, _act(act), 1.0
    L.check(lib.sfast_hip_qlinear_w8(_ptr(x2d), _ptr(w), _ptr(bias), _ptr(out), C.byref(p), float(scale), _stream(x)), "sfast_hip_qlinear_w8")
    return out.reshape(*lead, N)
