"""The C-ABI library builds, loads and exports what include/sfast_hip.h declares -- runs on CPU
(hipcc cross-compiles gfx950 without a GPU; no kernel is launched here)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sfast_hip.h")


def _declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"\b(sfast_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built_lib):
    from sfast.hip import lib as L
    lib = L.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in sfast_hip.h but not exported"
    assert set(L.EXPORTS) == set(declared)
    assert lib.sfast_hip_abi_version() == L.ABI_VERSION


def test_ctypes_structs_match_the_c_header(built_lib):
    from sfast.hip import lib as L
    structs = {"sfast_gn_params": L.GnParams, "sfast_ln_params": L.LnParams, "sfast_gemm_params": L.GemmParams,
               "sfast_conv_params": L.ConvParams, "sfast_attn_params": L.AttnParams, "sfast_copy_params": L.CopyParams,
               "sfast_temb_params": L.TembParams, "sfast_softmax_params": L.SoftmaxParams, "sfast_image_params": L.ImageParams, "sfast_add_params": L.AddParams,
               "sfast_gemv_grouped_params": L.GemvGroupedParams, "sfast_epilogue_ext": L.EpilogueExt, "sfast_gn_stats_layout": L.GnStatsLayout}
    body = "".join(f'printf("%zu\\n", sizeof({n}));' for n in structs)
    probes = [("sfast_gemm_params", "ld_rowbias"), ("sfast_gemm_params", "split_k"), ("sfast_conv_params", "xs"),
              ("sfast_conv_params", "ld_rowbias"), ("sfast_conv_params", "pad_w_extra"), ("sfast_attn_params", "scale"), ("sfast_copy_params", "dst_strides"),
              ("sfast_softmax_params", "ldx"), ("sfast_softmax_params", "scale"), ("sfast_image_params", "to_uint8"), ("sfast_add_params", "dst_strides"),
              ("sfast_gemv_grouped_params", "ldx"), ("sfast_gemv_grouped_params", "in_act"), ("sfast_epilogue_ext", "gn_rows_per_sample"),
              ("sfast_gn_stats_layout", "unit")]
    body += "".join(f'printf("%zu\\n", offsetof({s}, {f}));' for s, f in probes)
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "abi.c")
        with open(src, "w") as f:
            f.write(f'#include <stdio.h>\n#include <stddef.h>\n#include "{HEADER}"\nint main(){{{body}return 0;}}\n')
        exe = os.path.join(d, "abi")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", src, "-o", exe])  # header is plain C
        out = [int(x) for x in subprocess.check_output([exe]).split()]
    for (name, st), size in zip(structs.items(), out):
        assert C.sizeof(st) == size, name
    for (s, f), off in zip(probes, out[len(structs):]):
        assert getattr(structs[s], f).offset == off, (s, f)


def test_validation_errors_surface_without_a_gpu(built_lib):
    # argument validation happens before any launch -> checkable on CPU
    from sfast.hip import lib as L
    lib = L.load()
    p = L.GemmParams()
    p.M, p.N, p.K = 0, 8, 8
    segs = (C.c_void_p * 1)(1)
    rc = lib.sfast_hip_gemm(1, segs, None, None, None, 1, C.byref(p), None, 0, None)
    assert rc == -2 and b"bad shape" in lib.sfast_hip_last_error()
    g = L.GnParams(L.F16, L.NHWC, 1, 30, 4, 4, 30, L.ACT_NONE, 1e-5)  # 30 channels, 4 groups: not divisible
    assert lib.sfast_hip_group_norm(1, None, None, None, 1, C.byref(g), None, 0, None) == -2


def test_workspace_queries_are_consistent(built_lib):
    from sfast.hip import lib as L
    lib = L.load()
    # 8x8 level conv of SD1.5 at B=2: M=128, K=11520 -> the planner must split K and ask for slabs
    p = L.ConvParams()
    p.dtype, p.B, p.H, p.W, p.Cin, p.Cout, p.KH, p.KW = L.F16, 2, 8, 8, 1280, 1280, 3, 3
    p.stride_h = p.stride_w = p.dil_h = p.dil_w = 1
    p.pad_h = p.pad_w = 1
    p.C1 = 1280
    nb = lib.sfast_hip_conv2d_workspace_bytes(C.byref(p))
    # fp32 slabs (whole tiles of the split problem; here the tiles cover M x N exactly) + the ticket block at the end
    assert nb > L.WS_TICKET_BYTES and (nb - L.WS_TICKET_BYTES) % (128 * 1280 * 4) == 0
    # the workspace query and the plan query agree: bytes >= splits * M * N * 4 + ticket block (0 when unsplit)
    p.H = p.W = 64
    p.Cin = p.C1 = p.Cout = 320
    o = (C.c_int32 * 5)()
    assert lib.sfast_hip_igemm_plan(2 * 64 * 64, 320, 9 * 320, 0, 0, 0, C.byref(o)) == 0
    splits = o[2]
    got = lib.sfast_hip_conv2d_workspace_bytes(C.byref(p))
    assert got == 0 if splits == 1 else got >= splits * (2 * 64 * 64) * 320 * 4 + L.WS_TICKET_BYTES
    # forcing the register pipe without split needs none
    p.variant, p.split_k = 1, 1
    assert lib.sfast_hip_conv2d_workspace_bytes(C.byref(p)) == 0
    p.variant, p.split_k = 0, 0
    g = L.GnParams(L.F16, L.NHWC, 2, 320, 4096, 32, 320, L.ACT_SILU, 1e-5)
    assert lib.sfast_hip_group_norm_workspace_bytes(C.byref(g)) > 0


def test_missing_library_is_a_loud_error(monkeypatch):
    from sfast.hip import lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libsfast_hip.so")
    try:
        L.load()
        raise AssertionError("load() must raise when the kernel library is absent")
    except L.SfastHipError as e:
        assert "no fallback" in str(e)


def test_ops_reject_cpu_tensors(built_lib):
    import torch
    import sfast  # noqa: F401  registers torch.ops.sfast*
    from sfast.hip import functional as F
    x = torch.zeros(2, 8, dtype=torch.float16)
    for call in (lambda: torch.ops.sfast.cublas_lowp_linear(x, x, None),
                 lambda: torch.ops.sfast_triton.layer_norm(x, [8], None, None, 1e-5),
                 lambda: torch.ops.sfast.cutlass_linear_geglu_unified(x, x, None)):
        try:
            call()
            raise AssertionError("CPU call must not succeed")
        except (NotImplementedError, RuntimeError):
            pass
    try:
        F.layer_norm(x, [8])
        raise AssertionError("functional wrapper must reject CPU tensors")
    except RuntimeError as e:
        assert "no CPU path" in str(e)


def test_bench_kernel_symbols_name_real_device_kernels(built_lib):
    """bench.py's `roofline.kernel` must be the symbol rocprofv3 prints: every variant string the library reports for
    the MFMA kernels has to map (bench.kernel_symbol) onto a kernel that is really in the code object."""
    import sys
    from sfast.hip import lib as L
    sys.path.insert(0, ROOT)
    import bench
    with open(L.LIB_PATH, "rb") as f:
        blob = f.read()
    variants = ["attn_fwd[D=160,BQ=128]", "attn_fwd[D=40,BQ=128]", "attn_fwd[D=80,BQ=64]", "attn_fwd[D=64,BQ=128]",
                "igemm_conv_f16[128x128,split=1,ws4]", "igemm_conv_f16[128x128,split=12,reg]", "igemm_conv_f16[128x160,split=2,reg]",
                "igemm_conv_f16[128x160,split=4,ws4]", "igemm_conv_f16[64x64,split=3,ws4]", "igemm_lin_f16[128x128,split=1,dma2]",
                "igemm_lin_f16[128x128,split=4,ws4]", "igemm_lin_f16[64x64,split=1,reg]", "igemm_lin_f16[64x64,split=6,ws4]",
                "igemm_lin_f16_geglu[128x128,split=1,dma2]", "igemm_lin_f16_geglu[64x128,split=1,ws3]", "igemm_lin_bf16[64x64,split=1,ws4]",
                "igemm_conv_f16[128x128,split=1,ws4]+gnstats", "igemm_lin_f16[64x64,split=1,reg]+gnstats", "igemm_conv_f16[128x160,split=1,ws4]+staged",
                "igemm_conv_f16[128x128,split=6,ws4]+join@xcd2x2x2", "igemm_conv_f16[128x128,split=3,ws4]+gnstats+join", "igemm_lin_f16[64x64,split=3,ws4]+join",
                # pipe 4 (packed weights): every tile, linear and conv, staged and not, bf16
                "igemm_lin_f16[64x160,split=1,pk4]@xcd1x8x1", "igemm_lin_f16[64x160,split=1,pk4]+gnstats@xcd1x4x2", "igemm_conv_f16[128x256,split=2,pk4]@xcd2x4x1",
                "igemm_conv_f16[64x320,split=1,pk4]+gnstats", "igemm_lin_f16[128x160,split=1,pk4]", "igemm_conv_f16[64x256,split=6,pk4]",
                "igemm_lin_bf16[128x128,split=1,pk4]+staged", "igemm_conv_bf16[128x128,split=12,pk4]@xcd4x1x2",
                # pipe 5 (256-row tiles): ping-pong (pp), + producer waves (ppw), producers + lockstep consumers (ppl); GEGLU; bf16; split-K
                "igemm_conv_f16[256x160,split=1,pp3]+staged@xcd1x8x1", "igemm_lin_f16[256x128,split=1,pp3]+staged", "igemm_conv_f16[256x256,split=1,pp2]+staged",
                "igemm_lin_f16_geglu[256x256,split=1,pp2]@xcd1x4x2", "igemm_conv_f16[256x128,split=1,ppw3]+staged", "igemm_lin_f16[256x160,split=1,ppw3]+gnstats",
                "igemm_conv_f16[256x160,split=1,ppl3]+gnstats@xcd1x8x1", "igemm_conv_f16[256x128,split=1,ppl3]+staged", "igemm_lin_f16[256x160,split=1,ppl3]+staged",
                "igemm_lin_f16_geglu[256x128,split=1,ppl3]@xcd1x4x2", "igemm_conv_f16[256x160,split=3,ppl3]@xcd1x8x1", "igemm_conv_bf16[256x160,split=1,ppl3]+staged",
                "igemm_lin_bf16_geglu[256x128,split=1,ppl3]"]
    for v in variants:
        sym = bench.kernel_symbol(v)
        assert sym.startswith("_ZN5sfast"), (v, sym)
        assert (sym + ".kd").encode() in blob, f"{v} -> {sym} is not a kernel of libsfast_hip.so"


def test_plain_c_consumer_links_and_calls_the_library(built_lib):
    """The boundary is a C ABI, not a Python one: a C99 program that only includes include/sfast_hip.h links against
    libsfast_hip.so, checks the version, and gets the documented status + message for bad arguments (no GPU needed:
    validation precedes every launch)."""
    from sfast.hip import lib as L
    src = r'''
#include <stdio.h>
#include <string.h>
#include "sfast_hip.h"
int main(void) {
    if (sfast_hip_abi_version() != SFAST_HIP_ABI_VERSION) { printf("abi %d\n", sfast_hip_abi_version()); return 2; }
    sfast_attn_params p;
    memset(&p, 0, sizeof p);
    p.dtype = SFAST_F16; p.B = 1; p.H = 1; p.Sq = 0; p.Skv = 4; p.D = 40;   /* Sq = 0: bad shape */
    int dummy;
    int rc = sfast_hip_attention(&dummy, &dummy, &dummy, &dummy, &p, NULL);
    if (rc != SFAST_ERR_INVALID) { printf("rc %d\n", rc); return 3; }
    if (!strstr(sfast_hip_last_error(), "bad shape")) { printf("msg %s\n", sfast_hip_last_error()); return 4; }
    sfast_softmax_params s;
    memset(&s, 0, sizeof s);
    if (sfast_hip_softmax_rows(NULL, NULL, &s, NULL) == SFAST_OK) return 5;
    printf("ok %d\n", SFAST_HIP_ABI_VERSION);
    return 0;
}
'''
    libdir = os.path.dirname(L.LIB_PATH)
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "consumer.c")
        exe = os.path.join(d, "consumer")
        with open(c, "w") as f:
            f.write(src)
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe,
                        "-L", libdir, "-lsfast_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip() == f"ok {L.ABI_VERSION}", (r.returncode, r.stdout, r.stderr)
