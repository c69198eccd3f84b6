"""ctypes binding of libsfast_hip.so -- the C ABI declared in include/sfast_hip.h.

This is the only place Python touches the kernel library. The library is REQUIRED: if it is
missing or fails to load, `load()` raises -- there is no ATen / CPU fallback behind it (the
reference's ops fall back to ATen when their CUDA extension can't take a tensor,
src/sfast/triton/torch_ops.py:116-125; here an unsupported fast-path precondition selects the
generic HIP kernel inside the library instead).
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "_lib", "libsfast_hip.so")
# The probe build (build.py --probes: timing-only experiment kernels, the never-selected patch conv pipe, the split-K join) is a
# separate file that only tools/ and the tests of those candidates ask for, by setting SFAST_HIP_PROBES=1 BEFORE the first load.
PROBES_LIB_PATH = os.path.join(os.path.dirname(_HERE), "_lib", "libsfast_hip_probes.so")
if os.environ.get("SFAST_HIP_PROBES", "0") == "1":
    LIB_PATH = PROBES_LIB_PATH

ABI_VERSION = 9

# enums (include/sfast_hip.h)
F16, BF16, F32 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU, ACT_GELU_TANH, ACT_SILU, ACT_SIGMOID, ACT_TANH = range(7)
NHWC, NCHW = 0, 1
MAX_WSEG = 4
MAX_GROUPS = 32
MAX_GEMM_GROUPS = 64
WS_TICKET_BYTES = 65536   # sfast_hip.h SFAST_WS_TICKET_BYTES: split-K ticket counters at the end of a workspace
EXT_WS_TICKETS = 1        # sfast_epilogue_ext.flags

EXPORTS = [
    "sfast_hip_abi_version", "sfast_hip_init", "sfast_hip_last_error", "sfast_hip_last_kernel",
    "sfast_hip_group_norm_workspace_bytes", "sfast_hip_group_norm", "sfast_hip_layer_norm", "sfast_hip_softmax_rows",
    "sfast_hip_gemm_workspace_bytes", "sfast_hip_gemm", "sfast_hip_gemm_ex", "sfast_hip_gemm_stats_layout",
    "sfast_hip_conv2d_workspace_bytes", "sfast_hip_conv2d", "sfast_hip_conv2d_ex", "sfast_hip_conv2d_stats_layout",
    "sfast_hip_group_norm_apply",
    "sfast_hip_attention", "sfast_hip_attention_bias", "sfast_hip_strided_copy", "sfast_hip_timestep_embedding",
    "sfast_hip_gemv_grouped", "sfast_hip_gemm_grouped", "sfast_hip_qlinear_w8", "sfast_hip_cfg_ddim_step", "sfast_hip_linear_step", "sfast_hip_mix_rows", "sfast_hip_igemm_plan", "sfast_hip_set_trace", "sfast_hip_image_postprocess", "sfast_hip_add_strided",
    "sfast_hip_schedule_advance", "sfast_hip_conv2d_plan", "sfast_hip_workspace_init", "sfast_hip_has_probes",
    "sfast_hip_gn_conv2d_supported", "sfast_hip_gn_conv2d_workspace_bytes", "sfast_hip_gn_conv2d",
    "sfast_hip_lora_merge_plan", "sfast_hip_lora_merge", "sfast_hip_packed_weight_bytes", "sfast_hip_pack_weight",
]
LORA_MAX_RANK = 128


class GnParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("layout", C.c_int32), ("N", C.c_int32), ("C", C.c_int32),
                ("HW", C.c_int32), ("G", C.c_int32), ("C1", C.c_int32), ("act", C.c_int32),
                ("eps", C.c_float)]


class LnParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("eps", C.c_float)]


class AddParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
                ("src_strides", C.c_int64 * 4), ("dst_strides", C.c_int64 * 4)]


class ImageParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("B", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("denormalize", C.c_int32), ("to_uint8", C.c_int32)]


class SoftmaxParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("ldx", C.c_int64), ("ldy", C.c_int64),
                ("scale", C.c_float)]


class GemmParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("ldx", C.c_int64), ("ldw", C.c_int64), ("ldo", C.c_int64), ("ldr", C.c_int64),
                ("n_wseg", C.c_int32), ("rows_per_seg", C.c_int32), ("geglu", C.c_int32),
                ("act", C.c_int32), ("res_before_act", C.c_int32), ("alpha", C.c_float),
                ("rows_per_batch", C.c_int32), ("ld_rowbias", C.c_int64), ("in_act", C.c_int32),
                ("variant", C.c_int32), ("split_k", C.c_int32)]


class EpilogueExt(C.Structure):
    _fields_ = [("out_scale", C.c_float), ("gn_unit", C.c_int32), ("gn_rows_per_sample", C.c_int32), ("flags", C.c_int32),
                # ABI 8: the GroupNorm(+SiLU) that consumes the output, computed by the split-K reduce launch
                ("gn_out", C.c_void_p), ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p), ("gn_groups", C.c_int32), ("gn_eps", C.c_float),
                ("gn_act", C.c_int32), ("reserved", C.c_int32),
                # ABI 9: packed copies of the weight segments (sfast_hip_pack_weight): array of n_wseg pointers, or NULL
                ("w_packed", C.c_void_p)]


class GnStatsLayout(C.Structure):
    _fields_ = [("rb_rows", C.c_int32), ("n_rb", C.c_int32), ("bno", C.c_int32), ("tiles_n", C.c_int32), ("slots", C.c_int32),
                ("unit", C.c_int32)]

    def nbytes(self):
        return int(self.n_rb) * int(self.tiles_n) * int(self.slots) * 8


class MixParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("M", C.c_int64), ("C", C.c_int32), ("vec_rows", C.c_int32), ("vec_mod", C.c_int32),
                ("ld_vec", C.c_int64), ("wx", C.c_float), ("wy", C.c_float), ("switch_spatial_to_temporal", C.c_int32)]


class LoraEntry(C.Structure):
    _fields_ = [("w", C.c_void_p), ("down", C.c_void_p), ("up", C.c_void_p), ("out", C.c_void_p),
                ("N", C.c_int32), ("K", C.c_int32), ("r", C.c_int32), ("tile_begin", C.c_int32),
                ("ldw", C.c_int64), ("ldd", C.c_int64), ("ldu", C.c_int64), ("scale_index", C.c_int32), ("reserved", C.c_int32)]


class GemvGroupedParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("M", C.c_int32), ("K", C.c_int32), ("n_groups", C.c_int32),
                ("n_rows", C.c_int32 * MAX_GROUPS), ("ldx", C.c_int64), ("ldw", C.c_int64), ("ldo", C.c_int64),
                ("act", C.c_int32), ("in_act", C.c_int32)]


class ConvParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("Cin", C.c_int32), ("Cout", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32),
                ("stride_h", C.c_int32), ("stride_w", C.c_int32), ("pad_h", C.c_int32),
                ("pad_w", C.c_int32), ("dil_h", C.c_int32), ("dil_w", C.c_int32),
                ("upsample2x", C.c_int32), ("C1", C.c_int32),
                ("xs", C.c_int64 * 4), ("x2s", C.c_int64 * 4), ("ws", C.c_int64 * 4),
                ("os", C.c_int64 * 4), ("zs", C.c_int64 * 4),
                ("act", C.c_int32), ("res_before_act", C.c_int32), ("alpha", C.c_float),
                ("ld_rowbias", C.c_int64), ("variant", C.c_int32), ("split_k", C.c_int32),
                ("pad_h_extra", C.c_int32), ("pad_w_extra", C.c_int32)]


class GnConvParams(C.Structure):
    _fields_ = [("conv", ConvParams), ("groups", C.c_int32), ("eps", C.c_float), ("gn_act", C.c_int32)]


class AttnParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("Sq", C.c_int32),
                ("Skv", C.c_int32), ("D", C.c_int32),
                ("qs", C.c_int64 * 3), ("ks", C.c_int64 * 3), ("vs", C.c_int64 * 3), ("os", C.c_int64 * 3),
                ("scale", C.c_float), ("variant", C.c_int32)]


class CopyParams(C.Structure):
    _fields_ = [("elem_bytes", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
                ("src_strides", C.c_int64 * 4), ("dst_strides", C.c_int64 * 4)]


class TembParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("B", C.c_int32), ("dim", C.c_int32),
                ("flip_sin_to_cos", C.c_int32), ("downscale_freq_shift", C.c_float),
                ("max_period", C.c_float)]


class SfastHipError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()


def _declare(lib):
    vp, sz = C.c_void_p, C.c_size_t
    lib.sfast_hip_abi_version.restype = C.c_int
    lib.sfast_hip_init.restype = C.c_int
    lib.sfast_hip_last_error.restype = C.c_char_p
    lib.sfast_hip_last_kernel.restype = C.c_char_p
    lib.sfast_hip_group_norm_workspace_bytes.restype = sz
    lib.sfast_hip_group_norm_workspace_bytes.argtypes = [C.POINTER(GnParams)]
    lib.sfast_hip_group_norm.restype = C.c_int
    lib.sfast_hip_group_norm.argtypes = [vp, vp, vp, vp, vp, C.POINTER(GnParams), vp, sz, vp]
    lib.sfast_hip_layer_norm.restype = C.c_int
    lib.sfast_hip_layer_norm.argtypes = [vp, vp, vp, vp, C.POINTER(LnParams), vp]
    lib.sfast_hip_add_strided.restype = C.c_int
    lib.sfast_hip_add_strided.argtypes = [vp, vp, C.POINTER(AddParams), vp]
    lib.sfast_hip_image_postprocess.restype = C.c_int
    lib.sfast_hip_image_postprocess.argtypes = [vp, vp, C.POINTER(ImageParams), vp]
    lib.sfast_hip_softmax_rows.restype = C.c_int
    lib.sfast_hip_softmax_rows.argtypes = [vp, vp, C.POINTER(SoftmaxParams), vp]
    lib.sfast_hip_gemm_workspace_bytes.restype = sz
    lib.sfast_hip_gemm_workspace_bytes.argtypes = [C.POINTER(GemmParams)]
    lib.sfast_hip_gemm.restype = C.c_int
    lib.sfast_hip_gemm.argtypes = [vp, C.POINTER(vp), vp, vp, vp, vp, C.POINTER(GemmParams), vp, sz, vp]
    lib.sfast_hip_gemm_ex.restype = C.c_int
    lib.sfast_hip_gemm_ex.argtypes = [vp, C.POINTER(vp), vp, vp, vp, vp, C.POINTER(GemmParams), C.POINTER(EpilogueExt), vp, vp, sz, vp]
    lib.sfast_hip_conv2d_ex.restype = C.c_int
    lib.sfast_hip_conv2d_ex.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.POINTER(ConvParams), C.POINTER(EpilogueExt), vp, vp, sz, vp]
    lib.sfast_hip_gemm_stats_layout.restype = C.c_int
    lib.sfast_hip_gemm_stats_layout.argtypes = [C.POINTER(GemmParams), C.POINTER(EpilogueExt), C.POINTER(GnStatsLayout)]
    lib.sfast_hip_conv2d_stats_layout.restype = C.c_int
    lib.sfast_hip_conv2d_stats_layout.argtypes = [C.POINTER(ConvParams), C.POINTER(EpilogueExt), C.POINTER(GnStatsLayout)]
    lib.sfast_hip_group_norm_apply.restype = C.c_int
    lib.sfast_hip_group_norm_apply.argtypes = [vp, vp, vp, vp, vp, C.POINTER(GnParams), vp, C.POINTER(GnStatsLayout), vp,
                                               C.POINTER(GnStatsLayout), vp]
    lib.sfast_hip_qlinear_w8.restype = C.c_int
    lib.sfast_hip_qlinear_w8.argtypes = [vp, vp, vp, vp, C.POINTER(GemmParams), C.c_float, vp]
    lib.sfast_hip_gemm_grouped.restype = C.c_int
    lib.sfast_hip_gemm_grouped.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(GemmParams), C.c_int32, vp]
    lib.sfast_hip_gemv_grouped.restype = C.c_int
    lib.sfast_hip_gemv_grouped.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), vp, C.POINTER(GemvGroupedParams), vp]
    lib.sfast_hip_conv2d_workspace_bytes.restype = sz
    lib.sfast_hip_conv2d_workspace_bytes.argtypes = [C.POINTER(ConvParams)]
    lib.sfast_hip_conv2d.restype = C.c_int
    lib.sfast_hip_conv2d.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.POINTER(ConvParams), vp, sz, vp]
    lib.sfast_hip_attention.restype = C.c_int
    lib.sfast_hip_attention.argtypes = [vp, vp, vp, vp, C.POINTER(AttnParams), vp]
    lib.sfast_hip_attention_bias.restype = C.c_int
    lib.sfast_hip_attention_bias.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_int64 * 3), vp, C.POINTER(AttnParams), vp]
    lib.sfast_hip_strided_copy.restype = C.c_int
    lib.sfast_hip_strided_copy.argtypes = [vp, vp, C.POINTER(CopyParams), vp]
    lib.sfast_hip_timestep_embedding.restype = C.c_int
    lib.sfast_hip_timestep_embedding.argtypes = [vp, vp, C.POINTER(TembParams), vp]
    lib.sfast_hip_set_trace.restype = C.c_int
    lib.sfast_hip_set_trace.argtypes = [C.c_void_p]
    lib.sfast_hip_igemm_plan.restype = C.c_int
    lib.sfast_hip_igemm_plan.argtypes = [C.c_int32] * 6 + [C.POINTER(C.c_int32 * 5)]
    lib.sfast_hip_packed_weight_bytes.restype = sz
    lib.sfast_hip_packed_weight_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.sfast_hip_pack_weight.restype = C.c_int
    lib.sfast_hip_pack_weight.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int32, vp]
    lib.sfast_hip_lora_merge_plan.restype = C.c_int
    lib.sfast_hip_lora_merge_plan.argtypes = [C.POINTER(LoraEntry), C.c_int32, C.POINTER(C.c_int32)]
    lib.sfast_hip_lora_merge.restype = C.c_int
    lib.sfast_hip_lora_merge.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_int32, vp]
    lib.sfast_hip_mix_rows.restype = C.c_int
    lib.sfast_hip_mix_rows.argtypes = [vp, vp, vp, vp, vp, C.POINTER(MixParams), vp]
    lib.sfast_hip_linear_step.restype = C.c_int
    lib.sfast_hip_linear_step.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int64, C.c_int64, C.c_int32, vp]
    lib.sfast_hip_workspace_init.restype = C.c_int
    lib.sfast_hip_workspace_init.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sfast_hip_conv2d_plan.restype = C.c_int
    lib.sfast_hip_conv2d_plan.argtypes = [C.POINTER(ConvParams), C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.sfast_hip_schedule_advance.restype = C.c_int
    lib.sfast_hip_schedule_advance.argtypes = [vp, vp, C.c_int32, vp, vp, C.c_int32, vp, C.c_int32, vp]
    lib.sfast_hip_has_probes.restype = C.c_int
    lib.sfast_hip_gn_conv2d_supported.restype = C.c_int
    lib.sfast_hip_gn_conv2d_supported.argtypes = [C.POINTER(GnConvParams)]
    lib.sfast_hip_gn_conv2d_workspace_bytes.restype = sz
    lib.sfast_hip_gn_conv2d_workspace_bytes.argtypes = [C.POINTER(GnConvParams)]
    lib.sfast_hip_gn_conv2d.restype = C.c_int
    lib.sfast_hip_gn_conv2d.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(GnConvParams), vp, sz, vp]
    lib.sfast_hip_cfg_ddim_step.restype = C.c_int
    lib.sfast_hip_cfg_ddim_step.argtypes = [vp, vp, vp, vp, vp, C.c_float, C.c_int64, C.c_int32, vp]


def load():
    """Load (once) and return the ctypes library handle. Raises if it is absent or incompatible."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise SfastHipError(
                f"{LIB_PATH} not found: build it with `python stable-fast_amd/build.py` "
                "(hipcc --offload-arch=gfx950). The HIP kernel library is required; there is no fallback.")
        # torch must be imported first so that libamdhip64.so.7 resolves to the runtime torch uses
        # (streams / device pointers are only valid inside that runtime instance).
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        missing = [s for s in EXPORTS if not hasattr(lib, s)]
        if missing:
            raise SfastHipError(f"libsfast_hip.so lacks symbols: {missing}")
        _declare(lib)
        if lib.sfast_hip_abi_version() != ABI_VERSION:
            raise SfastHipError("libsfast_hip.so ABI version mismatch; rebuild")
        _lib = lib
    return _lib


_probes_lib = None


def load_probes():
    """A SECOND handle, on the probe build (libsfast_hip_probes.so), whatever library load() returns -- for tests / tools that ask the
    host-only queries of a measured-and-not-shipped candidate (sfast_hip_gn_conv2d_supported, ...). Raises when it was not built."""
    global _probes_lib
    if _probes_lib is None:
        with _lock:
            if _probes_lib is None:
                if not os.path.exists(PROBES_LIB_PATH):
                    raise SfastHipError(f"{PROBES_LIB_PATH} not found: build it with `python stable-fast_amd/build.py --probes`")
                import torch  # noqa: F401
                lib = C.CDLL(PROBES_LIB_PATH)
                _declare(lib)
                if lib.sfast_hip_abi_version() != ABI_VERSION or not lib.sfast_hip_has_probes():
                    raise SfastHipError("libsfast_hip_probes.so is stale; rebuild with --probes")
                _probes_lib = lib
    return _probes_lib


_inited = set()


def init_device(device=None):
    """Kernel-attribute setup, once per device (function attributes are per device); needs a visible GPU. Call before any
    graph capture. `device`: torch.device / index; default = the current device."""
    lib = load()
    import torch
    idx = torch.cuda.current_device() if device is None else (device.index if hasattr(device, "index") else int(device))
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _inited:
        with _lock:
            if idx not in _inited:
                with torch.cuda.device(idx):
                    rc = lib.sfast_hip_init()
                if rc != 0:
                    raise SfastHipError(f"sfast_hip_init failed ({rc}): {last_error()}")
                _inited.add(idx)
    return lib


def has_probes():
    """True when the loaded library is the probe build (timing-only instantiations, patch conv pipe, split-K join)."""
    return bool(load().sfast_hip_has_probes())


def last_error():
    return load().sfast_hip_last_error().decode(errors="replace")


def last_kernel():
    return load().sfast_hip_last_kernel().decode(errors="replace")


def check(rc, what):
    if rc != 0:
        raise SfastHipError(f"{what} failed ({rc}): {last_error()}")
