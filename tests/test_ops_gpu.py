"""Per-kernel parity on a real MI355X: HIP path (through the C ABI) vs the fp32 oracle on the same
seeded inputs, plus the committed golden vectors. Shapes: the reference's own test shapes
 and the SD1.5 / SDXL layer shapes, every forced tile variant and split-K.

Tolerances (stated per family): outputs are rounded to f16/bf16 (rel. 2^-11 / 2^-8), accumulation and
epilogues are fp32, so |err| <= atol + rtol*|ref| with rtol = 2e-3 (f16) / 1.6e-2 (bf16) covers the
output rounding plus the f16 rounding of P in attention; the reference's own tolerances are looser
(GEGLU 2e-2, conv 1e-3 in fp32, GroupNorm 1e-2).
"""
import os

import pytest
import torch

import sfast  # noqa: F401  (registers torch.ops.sfast / sfast_triton / sfast_xformers)
from oracle import ops_ref as R
from parity import compare

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def F():
    from sfast.hip import functional
    return functional


def last_kernel():
    from sfast.hip import lib
    return lib.last_kernel()


def _probes_loaded():
    try:
        from sfast.hip import lib
        return lib.has_probes()
    except Exception:
        return False


# Candidates that were measured and never selected (the LDS-patch conv pipe, the in-kernel split-K join) live only in the probe
# build of the library (stable-fast_amd/build.py --probes -> libsfast_hip_probes.so, loaded when SFAST_HIP_PROBES=1 is set before
# the first load): `SFAST_HIP_PROBES=1 pytest tests/test_ops_gpu.py -m gpu -k "patch or join"`. In the default run they skip.
needs_probes = pytest.mark.skipif(not _probes_loaded(), reason="needs the probe build: python stable-fast_amd/build.py --probes; SFAST_HIP_PROBES=1")


def tol(dtype, scale=1.0):
    if dtype == torch.bfloat16:
        return 2e-2 * scale, 1.6e-2
    if dtype == torch.float32:
        return 1e-4 * scale, 1e-4
    return 3e-3 * scale, 2e-3


def rnd(*shape, dtype=torch.float16, seed=0, scale=1.0, shift=0.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale + shift).to(device=DEV, dtype=dtype)


# ---- GroupNorm -------------------------------------------------------------------------------------
@pytest.mark.parametrize("silu", [False, True])
@pytest.mark.parametrize("cl", [False, True])
def test_group_norm_reference_selftest_shape(silu, cl):
    # triton/ops/group_norm.py:485-523: randn(2,320,32,32) fp16, G=32, both layouts
    x = rnd(2, 320, 32, 32, seed=1)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    w, b = rnd(320, seed=2), rnd(320, seed=3)
    y = F().group_norm(x, 32, w, b, 1e-5, "silu" if silu else None)
    assert y.is_contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
    compare(f"gn_ref_shape cl={cl} silu={silu}", y, R.group_norm_ref(x, 32, w, b, 1e-5, silu), *tol(x.dtype), kernel=last_kernel())


@pytest.mark.parametrize("shape", [(2, 320, 64, 64), (2, 640, 32, 32), (2, 1280, 8, 8), (1, 2560, 16, 16), (2, 1920, 32, 32),
                                   (2, 960, 64, 64), (1, 1280, 16, 16), (3, 640, 24, 40),
                                   (1, 128, 64, 64), (2, 256, 32, 32), (1, 512, 16, 16)])  # last row: VAE decoder widths (C/G = 4, 8, 16)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_group_norm_silu_unet_shapes(shape, dtype):
    x = rnd(*shape, dtype=dtype, seed=4, scale=2.0, shift=3.0).contiguous(memory_format=torch.channels_last)
    w, b = rnd(shape[1], dtype=dtype, seed=5, shift=1.0, scale=0.2), rnd(shape[1], dtype=dtype, seed=6, scale=0.2)
    y = F().group_norm(x, 32, w, b, 1e-5, "silu")
    assert "gn_nhwc" in last_kernel() or "gn_small" in last_kernel()
    compare(f"gn_silu {shape} {dtype}", y, R.group_norm_ref(x, 32, w, b, 1e-5, True), *tol(dtype, 2.0), kernel=last_kernel())


@pytest.mark.parametrize("c1,c2,hw", [(640, 320, 64), (1280, 1280, 16), (1280, 640, 32), (320, 320, 64)])
def test_group_norm_virtual_concat(c1, c2, hw):
    x1 = rnd(2, c1, hw, hw, seed=7, shift=-1.0).contiguous(memory_format=torch.channels_last)
    x2 = rnd(2, c2, hw, hw, seed=8, scale=3.0).contiguous(memory_format=torch.channels_last)
    w, b = rnd(c1 + c2, seed=9, shift=1.0, scale=0.1), rnd(c1 + c2, seed=10)
    y = F().group_norm(x1, 32, w, b, 1e-5, "silu", x2=x2)
    want = R.group_norm_ref(torch.cat([x1, x2], 1), 32, w, b, 1e-5, True)
    compare(f"gn_concat {c1}+{c2}@{hw}", y, want, *tol(x1.dtype, 2.0), kernel=last_kernel())


@pytest.mark.parametrize("case", ["cpg6", "f32", "3d", "tiny", "eps1e-6", "noaffine"])
def test_group_norm_generic_and_edges(case):
    if case == "cpg6":  # 6 channels per group -> generic kernel
        x, G = rnd(2, 36, 8, 8, seed=11).contiguous(memory_format=torch.channels_last), 6
    elif case == "f32":
        x, G = rnd(2, 64, 16, 16, dtype=torch.float32, seed=12), 8
    elif case == "3d":
        x, G = rnd(2, 64, 50, seed=13), 4
    elif case == "tiny":
        x, G = rnd(1, 32, 1, 1, seed=14).contiguous(memory_format=torch.channels_last), 4
    else:
        x, G = rnd(2, 320, 16, 16, seed=15).contiguous(memory_format=torch.channels_last), 32
    C = x.shape[1]
    w, b = (None, None) if case == "noaffine" else (rnd(C, dtype=x.dtype, seed=16), rnd(C, dtype=x.dtype, seed=17))
    eps = 1e-6 if case == "eps1e-6" else 1e-5
    y = F().group_norm(x, G, w, b, eps)
    compare(f"gn_edge {case}", y, R.group_norm_ref(x, G, w, b, eps), *tol(x.dtype, 2.0), kernel=last_kernel())


# ---- LayerNorm -------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1151, 8192), (8192, 320), (2048, 640), (512, 1280), (128, 1280), (2, 77, 768), (5, 77), (3, 2048),
                                   (7, 4104), (5, 12288), (3, 32768), (2, 32776), (4, 4096)])   # round 4: the wide-row kernel and its edges
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_layer_norm(shape, dtype):
    # (1151, 8192) is the reference self-test (triton/ops/layer_norm.py:522)
    x = rnd(*shape, dtype=dtype, seed=20, scale=2.0, shift=1.0)
    n = shape[-1]
    w, b = rnd(n, dtype=dtype, seed=21, shift=1.0, scale=0.2), rnd(n, dtype=dtype, seed=22)
    y = F().layer_norm(x, (n,), w, b, 1e-5)
    want_kernel = "ln_wide" if 4096 < n <= 32768 and n % 8 == 0 else ("ln_rows" if n % 8 == 0 and n <= 4096 else "ln_generic")
    assert last_kernel() == want_kernel, last_kernel()
    compare(f"ln {shape} {dtype}", y, R.layer_norm_ref(x, (n,), w, b), *tol(dtype, 2.0), kernel=last_kernel())


def test_layer_norm_f32_and_no_affine():
    x = rnd(33, 320, dtype=torch.float32, seed=23)
    compare("ln f32", F().layer_norm(x, (320,)), R.layer_norm_ref(x, (320,)), *tol(torch.float32), kernel=last_kernel())


# ---- GEMM family -------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("inf", [4, 8, 16])
@pytest.mark.parametrize("outf", [4, 8, 16])
@pytest.mark.parametrize("N", [4, 16])
def test_geglu_reference_grid(dtype, bias, inf, outf, N):
    # /root/reference/tests/operators/test_cutlass_dual_linear.py:42-56 (tolerance 2e-2 there)
    x = rnd(N, inf, dtype=dtype, seed=30)
    w = rnd(2 * outf, inf, dtype=dtype, seed=31, scale=0.5)
    b = rnd(2 * outf, dtype=dtype, seed=32) if bias else None
    y = F().linear(x, w, b, geglu=True)
    compare(f"geglu_grid {dtype} b={bias} {inf}->{outf} N={N}", y, R.linear_ref(x, w, b, geglu=True), 2e-2, 2e-2, kernel=last_kernel())


@pytest.mark.parametrize("M,K,N", [(8192, 320, 1280), (2048, 640, 2560), (512, 1280, 5120), (128, 1280, 5120), (100, 320, 1280)])
@pytest.mark.parametrize("variant", [0, 1, 3, 11, 13, 16, 18, 21, 23])
def test_geglu_unet_shapes(M, K, N, variant):
    x = rnd(M, K, seed=33)
    w = rnd(2 * N, K, seed=34, scale=K ** -0.5)
    b = rnd(2 * N, seed=35, scale=0.1)
    y = F().linear(x, w, b, geglu=True, variant=variant)
    assert "geglu" in last_kernel()
    compare(f"geglu M{M} K{K} N{N} v{variant}", y, R.linear_ref(x, w, b, geglu=True), *tol(x.dtype), kernel=last_kernel())


@pytest.mark.parametrize("split", [2, 4])
def test_geglu_split_k(split):
    x, w, b = rnd(256, 1280, seed=36), rnd(2 * 640, 1280, seed=37, scale=1280 ** -0.5), rnd(2 * 640, seed=38)
    y = F().linear(x, w, b, geglu=True, variant=1, split_k=split)
    assert f"split={split}" in last_kernel()
    y2 = F().linear(x, w, b, geglu=True, variant=11, split_k=split)
    assert f"split={split},dma" in last_kernel()
    compare(f"geglu dma split{split}", y2, R.linear_ref(x, w, b, geglu=True), *tol(x.dtype), kernel=last_kernel())
    y3 = F().linear(x, w, b, geglu=True, variant=21, split_k=split)
    assert f"split={split},ws" in last_kernel()
    compare(f"geglu ws split{split}", y3, R.linear_ref(x, w, b, geglu=True), *tol(x.dtype), kernel=last_kernel())
    compare(f"geglu split{split}", y, R.linear_ref(x, w, b, geglu=True), *tol(x.dtype), kernel=last_kernel())


@pytest.mark.parametrize("M,K,N", [(8192, 320, 320), (8192, 320, 960), (2048, 2560, 640), (512, 1280, 1280), (154, 768, 640),
                                   (4095, 328, 324), (77, 64, 36), (130, 1280, 1280)])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 11, 12, 13, 14, 15, 16, 17, 18, 21, 22, 23, 24, 25, 26])
def test_linear_variants(M, K, N, variant):
    x = rnd(M, K, seed=40)
    w = rnd(N, K, seed=41, scale=K ** -0.5)
    b = rnd(N, seed=42, scale=0.1)
    y = F().linear(x, w, b, variant=variant)
    assert "igemm_lin" in last_kernel() and (variant < 10 or ("ws" if variant > 20 else "dma") in last_kernel())
    compare(f"linear M{M} K{K} N{N} v{variant}", y, R.linear_ref(x, w, b), *tol(x.dtype), kernel=last_kernel())


@pytest.mark.parametrize("split", [2, 3, 8])
@pytest.mark.parametrize("variant", [1, 2, 4, 11, 13, 15, 21, 22, 23])
def test_linear_split_k(split, variant):
    x, w, b = rnd(128, 5120, seed=43), rnd(1280, 5120, seed=44, scale=5120 ** -0.5), rnd(1280, seed=45)
    r = rnd(128, 1280, seed=46)
    y = F().linear(x, w, b, residual=r, variant=variant, split_k=split)
    assert f"split={split}" in last_kernel()
    compare(f"linear split{split} v{variant}", y, R.linear_ref(x, w, b, residual=r), *tol(x.dtype), kernel=last_kernel())


@pytest.mark.parametrize("variant,split,want", [(3, 1, "@xcd1x2x4"), (3, 8, "@xcd8x1x1"), (3, 2, "@xcd2x1x4"), (13, 4, "@xcd4x1x2"),
                                                (21, 1, "@xcd1x4x2"), (23, 4, "@xcd4x1x2"), (1, 3, "@xcd1x4x2")])
def test_linear_xcd_box_map(variant, split, want):
    """Block -> (tile, K-split) boxes per XCD (igemm.hip choose_xcd_map): weight-heavy problems put the XCD factor on the K-splits first,
    then on the cheaper of row / column boxes; the map must stay a bijection whatever the factorisation."""
    x, w, b = rnd(512, 1024, seed=47), rnd(1280, 1024, seed=48, scale=1024 ** -0.5), rnd(1280, seed=49)
    r = rnd(512, 1280, seed=50)
    y = F().linear(x, w, b, residual=r, variant=variant, split_k=split)
    assert last_kernel().endswith(want), last_kernel()  # tags: K-split x row x column boxes
    compare(f"linear xcd map v{variant} split{split}", y, R.linear_ref(x, w, b, residual=r), *tol(x.dtype), kernel=last_kernel())


def test_geglu_and_conv_xcd_box_map():
    x, w, b = rnd(512, 1280, seed=51), rnd(2 * 5120, 1280, seed=52, scale=1280 ** -0.5), rnd(2 * 5120, seed=53)
    for variant in (1, 11, 21):
        y = F().linear(x, w, b, geglu=True, variant=variant, split_k=1)
        assert "@xcd1x1x8" in last_kernel(), last_kernel()  # 26 MB of weights: every XCD owns an eighth of the columns
        compare(f"geglu xcd map v{variant}", y, R.linear_ref(x, w, b, geglu=True), *tol(x.dtype), kernel=last_kernel())
    x, w, b = cl(rnd(2, 1280, 16, 16, seed=54)), cl(rnd(1280, 1280, 3, 3, seed=55, scale=11520 ** -0.5)), rnd(1280, seed=56)
    want = R.conv2d_ref(x, w, b, None, 1.0, 1, 1)
    for variant, split, tag in ((21, 4, "@xcd4x1x2"), (21, 8, "@xcd8x1x1"), (4, 2, "@xcd2x1x4"), (23, 1, "@xcd1x2x4"), (13, 6, "@xcd2x1x4")):
        y = F().conv2d(x, w, b, padding=1, variant=variant, split_k=split)
        assert last_kernel().endswith(tag), last_kernel()
        compare(f"conv xcd map v{variant} split{split}", y, want, *tol(x.dtype, 2.0), kernel=last_kernel())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("epi", ["res_after", "res_before_relu", "gelu", "silu", "rowbias", "segs3", "inplace_res", "strided_out"])
def test_linear_epilogues(dtype, epi):
    M, K, N = 600, 640, 640
    x = rnd(M, K, dtype=dtype, seed=50)
    w = rnd(N, K, dtype=dtype, seed=51, scale=K ** -0.5)
    b = rnd(N, dtype=dtype, seed=52, scale=0.1)
    r = rnd(M, N, dtype=dtype, seed=53)
    kw, ref_kw = {}, {}
    if epi == "res_after":
        kw = dict(residual=r, alpha=0.5)
    elif epi == "res_before_relu":
        kw = dict(residual=r, alpha=2.0, res_before_act=True, act="relu")
    elif epi in ("gelu", "silu"):
        kw = dict(act=epi)
    elif epi == "rowbias":
        rb = rnd(3, N, dtype=dtype, seed=54)
        kw = dict(rowbias=rb, rows_per_batch=200)
    if epi == "segs3":
        ws = [rnd(N, K, dtype=dtype, seed=55 + i, scale=K ** -0.5) for i in range(3)]
        y = F().linear(x, ws, None)
        want = R.linear_ref(x, torch.cat(ws, 0))
    elif epi == "inplace_res":
        buf = r.clone()
        y = F().linear(x, w, b, residual=buf, out=buf)
        want = R.linear_ref(x, w, b, residual=r)
    elif epi == "strided_out":
        big = torch.zeros(M, 3 * N, dtype=dtype, device=DEV)
        y = F().linear(x, w, b, out=big[:, N:2 * N])
        want = R.linear_ref(x, w, b)
        assert float(big[:, :N].abs().max()) == 0 and float(big[:, 2 * N:].abs().max()) == 0
    else:
        y = F().linear(x, w, b, **kw)
        want = R.linear_ref(x, w, b, **kw)
    compare(f"linear_epi {epi} {dtype}", y, want, *tol(dtype, 2.0), kernel=last_kernel())


@pytest.mark.parametrize("variant", [11, 12, 13, 15, 21, 22, 23, 24, 25, 26])
def test_dma_pipe_epilogues_bf16(variant):
    M, K, N = 700, 1280, 640
    dt = torch.bfloat16
    x, w, b = rnd(M, K, dtype=dt, seed=56), rnd(N, K, dtype=dt, seed=57, scale=K ** -0.5), rnd(N, dtype=dt, seed=58, scale=0.1)
    r, rb = rnd(M, N, dtype=dt, seed=59), rnd(4, N, dtype=dt, seed=60)
    y = F().linear(x, w, b, residual=r, alpha=0.5, act="gelu", rowbias=rb, rows_per_batch=175, variant=variant)
    assert ("ws" if variant >= 21 else "dma") in last_kernel()  # 11..18 = LDS-DMA ring, 21..23 = wave-specialised ring
    want = R.linear_ref(x, w, b, residual=r, alpha=0.5, act="gelu", rowbias=rb, rows_per_batch=175)
    compare(f"dma epilogue bf16 v{variant}", y, want, *tol(dt, 2.0), kernel=last_kernel())
    buf = r.clone()
    F().linear(x, w, b, residual=buf, out=buf, variant=variant)
    compare(f"dma inplace residual bf16 v{variant}", buf, R.linear_ref(x, w, b, residual=r), *tol(dt, 2.0), kernel=last_kernel())


@pytest.mark.parametrize("M", [1, 2, 5, 8, 16])
@pytest.mark.parametrize("case", ["plain", "silu_out", "in_silu", "res", "segs2", "ldo"])
def test_gemv_small_m(M, case):
    K, N = 1280, 1280
    x, w, b = rnd(M, K, seed=60), rnd(N, K, seed=61, scale=K ** -0.5), rnd(N, seed=62, scale=0.1)
    if case == "plain":
        y, want = F().linear(x, w, b), R.linear_ref(x, w, b)
    elif case == "silu_out":
        y, want = F().linear(x, w, b, act="silu"), R.linear_ref(x, w, b, act="silu")
    elif case == "in_silu":
        y, want = F().linear(x, w, b, in_act="silu"), R.linear_ref(x, w, b, in_act="silu")
    elif case == "res":
        r = rnd(M, N, seed=63)
        y = F().linear(x, w, b, residual=r, res_before_act=True, act="silu")
        want = R.linear_ref(x, w, b, residual=r, res_before_act=True, act="silu")
    elif case == "segs2":
        w2 = rnd(N, K, seed=64, scale=K ** -0.5)
        y, want = F().linear(x, [w, w2], None), R.linear_ref(x, torch.cat([w, w2], 0))
    else:
        big = torch.zeros(M, 4 * N, dtype=x.dtype, device=DEV)
        y = F().linear(x, w, b, out=big[:, 2 * N:3 * N])
        want = R.linear_ref(x, w, b)
    assert last_kernel() == "gemv_small_m"
    compare(f"gemv M{M} {case}", y, want, *tol(x.dtype), kernel=last_kernel())


def test_linear_naive_paths():
    # unaligned K (not a multiple of 8) and fp32 go to the generic kernel
    x, w, b = rnd(33, 12, seed=65), rnd(20, 12, seed=66), rnd(20, seed=67)
    compare("naive K12", F().linear(x, w, b, act="gelu"), R.linear_ref(x, w, b, act="gelu"), *tol(x.dtype), kernel=last_kernel())
    assert last_kernel() == "gemm_naive"
    x, w = rnd(40, 64, dtype=torch.float32, seed=68), rnd(48, 64, dtype=torch.float32, seed=69)
    compare("naive f32", F().linear(x, w), R.linear_ref(x, w), 1e-4, 1e-4, kernel=last_kernel())


# ---- conv family ------------------------------------------------------------------------------------------
def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("act", [None, "sigmoid", "relu", "tanh"])
def test_conv_reference_test_model(act):
    # /root/reference/tests/operators/test_cudnn_convolution.py:39-69 (+ the activation variants):
    # Conv2d(2,2,3) on ones(1,2,256,256) fp32, z = ones(1,1,254,254) broadcast, alpha 0.5, tol 1e-3
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(2, 2, 3).to(DEV)
    x = torch.ones(1, 2, 256, 256, device=DEV)
    z = torch.ones(1, 1, 254, 254, device=DEV)
    with torch.no_grad():
        y = F().conv2d(x, conv.weight, conv.bias, z=z, alpha=0.5, act=act)
        want = R.conv2d_ref(x, conv.weight, conv.bias, z, 0.5, act=act)
    compare(f"conv_ref_model act={act}", y, want, 1e-3, 1e-3, kernel=last_kernel())


CONV_CASES = [
    # name, B, Cin, H, W, Cout, k, stride, pad, extras
    ("res 320@64", 2, 320, 64, 64, 320, 3, 1, 1, dict(rowbias=True)),
    ("res2 320@64 +z", 2, 320, 64, 64, 320, 3, 1, 1, dict(z=True)),
    ("down 320@64 s2", 2, 320, 64, 64, 320, 3, 2, 1, {}),
    ("640@32", 2, 640, 32, 32, 640, 3, 1, 1, dict(z=True)),
    ("1280@16", 2, 1280, 16, 16, 1280, 3, 1, 1, dict(rowbias=True)),
    ("1280@8", 2, 1280, 8, 8, 1280, 3, 1, 1, dict(z=True)),
    ("up 1280@16 ups", 1, 1280, 16, 16, 1280, 3, 1, 1, dict(ups=True)),
    ("cat 640+320@64", 1, 640, 64, 64, 320, 3, 1, 1, dict(c2=320)),
    ("cat1x1 1280+640@32", 2, 1280, 32, 32, 640, 1, 1, 0, dict(c2=640)),
    ("1x1 320@64", 2, 320, 64, 64, 320, 1, 1, 0, dict(z=True)),
    ("cin32 odd", 1, 32, 17, 13, 64, 3, 1, 1, {}),
    ("cin8", 2, 8, 20, 20, 32, 3, 2, 1, dict(z=True)),
    ("5x5 pad2", 1, 64, 12, 12, 96, 5, 1, 2, {}),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 11, 12, 13, 14, 15, 16, 17, 18, 21, 22, 23, 24, 25, 26])
def test_conv_igemm(case, variant):
    name, B, Cin, H, W, Cout, k, stride, pad, ex = case
    x = cl(rnd(B, Cin, H, W, seed=70))
    c2 = ex.get("c2", 0)
    x2 = cl(rnd(B, c2, H, W, seed=71)) if c2 else None
    w = cl(rnd(Cout, Cin + c2, k, k, seed=72, scale=((Cin + c2) * k * k) ** -0.5))
    b = rnd(Cout, seed=73, scale=0.1)
    ups = ex.get("ups", False)
    Hin, Win = (2 * H, 2 * W) if ups else (H, W)
    Ho, Wo = (Hin + 2 * pad - k) // stride + 1, (Win + 2 * pad - k) // stride + 1
    z = cl(rnd(B, Cout, Ho, Wo, seed=74)) if ex.get("z") else None
    rb = rnd(B, Cout, seed=75) if ex.get("rowbias") else None
    y = F().conv2d(x, w, b, z=z, stride=stride, padding=pad, x2=x2, upsample2x=ups, rowbias=rb, variant=variant)
    assert "igemm_conv" in last_kernel(), last_kernel()
    assert y.is_contiguous(memory_format=torch.channels_last)
    want = R.conv2d_ref(x, w, b, z, 1.0, stride, pad, x2=x2, upsample2x=ups, rowbias=rb)
    compare(f"conv {name} v{variant}", y, want, *tol(x.dtype, 2.0), kernel=last_kernel())


PATCH_CASES = [
    # name, B, Cin, H, W, Cout, extras: every image size of the UNet levels (a 128-pixel tile = 2 rows at 64 wide ... two whole images at
    # 8x8), odd batch, two-source concat (slices of both sources), non-square images, Cout that leaves a ragged last column tile
    ("320@64", 2, 320, 64, 64, 320, dict(rowbias=True)),
    ("320@64 +z", 1, 320, 64, 64, 320, dict(z=True)),
    ("640@32", 2, 640, 32, 32, 640, dict(z=True)),
    ("1280@16", 2, 1280, 16, 16, 1280, dict(rowbias=True)),
    ("1280@8 two images per tile", 2, 1280, 8, 8, 1280, dict(z=True)),
    ("1280@8 B=4", 4, 1280, 8, 8, 640, {}),
    ("cat 640+320@64", 1, 640, 64, 64, 320, dict(c2=320)),
    ("cat 1280+1280@16", 2, 1280, 16, 16, 1280, dict(c2=1280, z=True)),
    ("64@32x16", 3, 64, 32, 16, 96, dict(z=True)),
    ("128@16x32", 2, 128, 16, 32, 200, {}),
]


@needs_probes
@pytest.mark.parametrize("case", PATCH_CASES, ids=[c[0] for c in PATCH_CASES])
@pytest.mark.parametrize("variant,split", [(31, 1), (32, 1), (34, 1), (31, 2), (32, 5), (34, 4), (31, 20)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_conv_patch_pipe(case, variant, split, dtype):
    """conv_patch.hip (pipe 3): the input patch of a 64-channel slice stays in LDS for all nine taps. Same results as the implicit-
    im2col pipes (and the oracle) for every tile / split / image geometry; a problem the pipe cannot take is planned on another one."""
    name, B, Cin, H, W, Cout, ex = case
    x = cl(rnd(B, Cin, H, W, seed=80, dtype=dtype))
    c2 = ex.get("c2", 0)
    x2 = cl(rnd(B, c2, H, W, seed=81, dtype=dtype)) if c2 else None
    w = cl(rnd(Cout, Cin + c2, 3, 3, seed=82, scale=((Cin + c2) * 9) ** -0.5, dtype=dtype))
    b = rnd(Cout, seed=83, scale=0.1, dtype=dtype)
    z = cl(rnd(B, Cout, H, W, seed=84, dtype=dtype)) if ex.get("z") else None
    rb = rnd(B, Cout, seed=85, dtype=dtype) if ex.get("rowbias") else None
    nslices = (Cin + c2) // 64
    y = F().conv2d(x, w, b, z=z, padding=1, x2=x2, rowbias=rb, variant=variant, split_k=split)
    k = last_kernel()
    assert "igemm_conv" in k
    bm = 128
    fits = (B * H * W) % bm == 0 and bm % W == 0 and ((H * W) % bm == 0 or bm % (H * W) == 0)
    if fits:
        assert "patch" in k, k
        if split <= nslices:
            assert f"split={-(-nslices // -(-nslices // split))}" in k, k   # K is cut between 64-channel slices
    else:
        assert "patch" not in k, k
    want = R.conv2d_ref(x, w, b, z, 1.0, 1, 1, x2=x2, rowbias=rb)
    compare(f"conv patch {name} v{variant} s{split} {dtype}", y, want, *tol(dtype, 2.0), kernel=k)


@needs_probes
@pytest.mark.parametrize("variant,split", [(31, 1), (32, 1), (34, 1), (31, 5), (32, 2)])
@pytest.mark.parametrize("cin,cout,hw,unit", [(320, 320, 32, 10), (640, 1280, 16, 20)])
def test_conv_patch_pipe_emits_groupnorm_statistics(variant, split, cin, cout, hw, unit):
    import numpy as np
    x = rnd(2, cin, hw, hw, seed=200, shift=0.5).contiguous(memory_format=torch.channels_last)
    w = rnd(cout, cin, 3, 3, seed=201, scale=(9 * cin) ** -0.5).contiguous(memory_format=torch.channels_last)
    b = rnd(cout, seed=202, shift=2.0)
    z = rnd(2, cout, hw, hw, seed=203).contiguous(memory_format=torch.channels_last)
    y, stats, lay = F().conv2d(x, w, b, z=z, padding=1, variant=variant, split_k=split, gn_unit=unit)
    k = last_kernel()
    assert "patch" in k and "+gnstats" in k, k
    plain = F().conv2d(x, w, b, z=z, padding=1, variant=variant, split_k=split)
    assert torch.equal(y, plain), k
    want = _stats_reference(y.permute(0, 2, 3, 1).reshape(-1, cout), lay)
    got = stats.double().cpu().numpy().reshape(want.shape)
    used = ~np.isnan(want)
    assert np.allclose(got[..., 0][used[..., 0]], want[..., 0][used[..., 0]], rtol=1e-4, atol=1e-4), k
    assert np.allclose(got[..., 1][used[..., 1]], want[..., 1][used[..., 1]], rtol=2e-3, atol=1e-2), k
    gam, bet = rnd(cout, seed=204, shift=1.0, scale=0.2), rnd(cout, seed=205, scale=0.2)
    yn = F().group_norm_apply(y, 32, gam, bet, 1e-5, "silu", stats, lay)
    compare(f"gn_apply conv patch {cin}->{cout}@{hw} v{variant} s{split}", yn, R.group_norm_ref(y, 32, gam, bet, 1e-5, True), *tol(y.dtype, 2.0), kernel=k)


# ---- pipe 5 (round 6): 256-row tiles, csrc/igemm_pp.h -- ids 51 - 53 (8 waves, ping-pong), 55 / 56 (+ 4 producer waves), 57 / 58 (producers + lockstep consumers)
PP_VARIANTS = [51, 52, 53, 55, 56, 57, 58]


def _pp_applies(M, K):
    return M >= 256 and K % 64 == 0


@pytest.mark.parametrize("M,K,N", [(8192, 320, 320), (8192, 320, 960), (2048, 2560, 640), (512, 1280, 1280), (4095, 320, 324), (300, 64, 36),
                                   (256, 1280, 1280), (8192, 328, 320), (130, 1280, 1280)])
@pytest.mark.parametrize("variant", PP_VARIANTS)
def test_linear_pp_variants(M, K, N, variant):
    x = rnd(M, K, seed=40)
    w = rnd(N, K, seed=41, scale=K ** -0.5)
    b = rnd(N, seed=42, scale=0.1)
    y = F().linear(x, w, b, variant=variant)
    k = last_kernel()
    assert "igemm_lin" in k and (("pp" in k.split(",")[-1]) == _pp_applies(M, K)), k  # K tail / fewer than 256 rows: another pipe takes the call
    compare(f"linear M{M} K{K} N{N} v{variant}", y, R.linear_ref(x, w, b), *tol(x.dtype), kernel=k)


@pytest.mark.parametrize("M,K,N", [(8192, 320, 1280), (2048, 640, 2560), (512, 1280, 5120), (700, 1280, 320), (256, 64, 64)])
@pytest.mark.parametrize("variant,tag", [(53, "geglu[256x256,"), (57, "geglu[256x128,")])
def test_geglu_pp_variant(M, K, N, variant, tag):
    x = rnd(M, K, seed=33)
    w = rnd(2 * N, K, seed=34, scale=K ** -0.5)
    b = rnd(2 * N, seed=35, scale=0.1)
    y = F().linear(x, w, b, geglu=True, variant=variant)
    assert tag in last_kernel() and ",pp" in last_kernel(), last_kernel()  # (the planner may add a split-K of its own)
    compare(f"geglu M{M} K{K} N{N} v{variant}", y, R.linear_ref(x, w, b, geglu=True), *tol(x.dtype), kernel=last_kernel())
    ws = [w[:N].contiguous(), w[N:].contiguous()]  # two live weight tensors (h rows, g rows), as the engine passes them
    y2 = F().linear(x, ws, b, geglu=True, variant=variant)
    assert torch.equal(y, y2), last_kernel()
    yb = F().linear(x.bfloat16(), w.bfloat16(), b.bfloat16(), geglu=True, variant=variant)
    compare(f"geglu bf16 M{M} K{K} N{N} v{variant}", yb, R.linear_ref(x.bfloat16(), w.bfloat16(), b.bfloat16(), geglu=True), *tol(torch.bfloat16, 2.0), kernel=last_kernel())


@pytest.mark.parametrize("variant,split", [(51, 2), (52, 2), (52, 3), (53, 2), (53, 5), (55, 2), (56, 4), (57, 3), (58, 2)])
def test_linear_pp_split_k(variant, split):
    x, w, b = rnd(600, 5120, seed=43), rnd(1280, 5120, seed=44, scale=5120 ** -0.5), rnd(1280, seed=45)
    r = rnd(600, 1280, seed=46)
    y = F().linear(x, w, b, residual=r, variant=variant, split_k=split)
    assert f"split={split},pp" in last_kernel(), last_kernel()
    compare(f"linear split{split} v{variant}", y, R.linear_ref(x, w, b, residual=r), *tol(x.dtype), kernel=last_kernel())


@pytest.mark.parametrize("variant", PP_VARIANTS)
def test_pp_epilogues_bf16(variant):
    M, K, N = 700, 1280, 640
    dt = torch.bfloat16
    x, w, b = rnd(M, K, dtype=dt, seed=56), rnd(N, K, dtype=dt, seed=57, scale=K ** -0.5), rnd(N, dtype=dt, seed=58, scale=0.1)
    r, rb = rnd(M, N, dtype=dt, seed=59), rnd(4, N, dtype=dt, seed=60)
    y = F().linear(x, w, b, residual=r, alpha=0.5, act="gelu", rowbias=rb, rows_per_batch=175, variant=variant)
    assert "pp" in last_kernel().split(",")[-1], last_kernel()
    want = R.linear_ref(x, w, b, residual=r, alpha=0.5, act="gelu", rowbias=rb, rows_per_batch=175)
    compare(f"pp epilogue bf16 v{variant}", y, want, *tol(dt, 2.0), kernel=last_kernel())
    buf = r.clone()
    F().linear(x, w, b, residual=buf, out=buf, variant=variant)
    compare(f"pp inplace residual bf16 v{variant}", buf, R.linear_ref(x, w, b, residual=r), *tol(dt, 2.0), kernel=last_kernel())
    big = torch.zeros(M, 3 * N, dtype=dt, device=DEV)
    ws = [rnd(N // 2, K, dtype=dt, seed=61 + i, scale=K ** -0.5) for i in range(2)]  # stacked weight segments + a strided output
    y = F().linear(x, ws, b, out=big[:, N:2 * N], variant=variant)
    compare(f"pp segments / strided out bf16 v{variant}", y, R.linear_ref(x, torch.cat(ws, 0), b), *tol(dt, 2.0), kernel=last_kernel())
    assert float(big[:, :N].abs().max()) == 0 and float(big[:, 2 * N:].abs().max()) == 0


PP_CONV_CASES = [c for c in CONV_CASES if c[0] in ("res 320@64", "res2 320@64 +z", "down 320@64 s2", "640@32", "1280@16", "1280@8", "cat 640+320@64",
                                                   "cat1x1 1280+640@32", "1x1 320@64", "5x5 pad2", "up 1280@16 ups")] + [
    ("up 320@24 ups B3", 3, 320, 24, 24, 192, 3, 1, 1, dict(ups=True, z=True)),      # fused nearest-2x upsample: the 12-wave forms only (55 ..)
    ("3 images 128@24 ragged", 3, 128, 24, 24, 200, 3, 1, 1, dict(z=True)),          # M = 1728: a ragged last row tile, Cout % 32 != 0
    ("dilated 64@40", 1, 64, 40, 40, 96, 3, 1, 2, dict(dil=2)),
]


@pytest.mark.parametrize("case", PP_CONV_CASES, ids=[c[0] for c in PP_CONV_CASES])
@pytest.mark.parametrize("variant", PP_VARIANTS)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_conv_pp_variants(case, variant, dtype):
    name, B, Cin, H, W, Cout, k, stride, pad, ex = case
    if dtype == torch.bfloat16 and variant not in (52, 53, 56, 58):
        pytest.skip("bf16: one tile shape per wave layout")
    x = cl(rnd(B, Cin, H, W, seed=70, dtype=dtype))
    c2 = ex.get("c2", 0)
    dil = ex.get("dil", 1)
    x2 = cl(rnd(B, c2, H, W, seed=71, dtype=dtype)) if c2 else None
    w = cl(rnd(Cout, Cin + c2, k, k, seed=72, scale=((Cin + c2) * k * k) ** -0.5, dtype=dtype))
    b = rnd(Cout, seed=73, scale=0.1, dtype=dtype)
    ups = ex.get("ups", False)
    Hin, Win = (2 * H, 2 * W) if ups else (H, W)
    Ho, Wo = (Hin + 2 * pad - dil * (k - 1) - 1) // stride + 1, (Win + 2 * pad - dil * (k - 1) - 1) // stride + 1
    z = cl(rnd(B, Cout, Ho, Wo, seed=74, dtype=dtype)) if ex.get("z") else None
    rb = rnd(B, Cout, seed=75, dtype=dtype) if ex.get("rowbias") else None
    y = F().conv2d(x, w, b, z=z, stride=stride, padding=pad, dilation=dil, x2=x2, upsample2x=ups, rowbias=rb, variant=variant)
    kname = last_kernel()
    assert "igemm_conv" in kname and (("pp" in kname.split(",")[-1]) == (B * Ho * Wo >= 256 and (not ups or variant >= 55))), kname
    want = R.conv2d_ref(x, w, b, z, 1.0, stride, pad, dil, x2=x2, upsample2x=ups, rowbias=rb)
    compare(f"conv {name} v{variant} {dtype}", y, want, *tol(dtype, 2.0), kernel=kname)


@pytest.mark.parametrize("variant,split", [(52, 1), (53, 1), (56, 1), (51, 1), (55, 1), (52, 2), (56, 3), (57, 1), (58, 1), (58, 2)])
@pytest.mark.parametrize("cin,cout,hw,unit", [(320, 320, 32, 10), (640, 1280, 16, 20), (320, 640, 64, 20)])
def test_conv_pp_emits_groupnorm_statistics(variant, split, cin, cout, hw, unit):
    import numpy as np
    x = rnd(2, cin, hw, hw, seed=200, shift=0.5).contiguous(memory_format=torch.channels_last)
    w = rnd(cout, cin, 3, 3, seed=201, scale=(9 * cin) ** -0.5).contiguous(memory_format=torch.channels_last)
    b = rnd(cout, seed=202, shift=2.0)
    z = rnd(2, cout, hw, hw, seed=203).contiguous(memory_format=torch.channels_last)
    y, stats, lay = F().conv2d(x, w, b, z=z, padding=1, variant=variant, split_k=split, gn_unit=unit)
    k = last_kernel()
    assert "pp" in k.split(",")[-1] and "+gnstats" in k, k
    plain = F().conv2d(x, w, b, z=z, padding=1, variant=variant, split_k=split)
    assert torch.equal(y, plain), k
    want = _stats_reference(y.permute(0, 2, 3, 1).reshape(-1, cout), lay)
    got = stats.double().cpu().numpy().reshape(want.shape)
    used = ~np.isnan(want)
    assert np.allclose(got[..., 0][used[..., 0]], want[..., 0][used[..., 0]], rtol=1e-4, atol=1e-4), k
    assert np.allclose(got[..., 1][used[..., 1]], want[..., 1][used[..., 1]], rtol=2e-3, atol=1e-2), k
    gam, bet = rnd(cout, seed=204, shift=1.0, scale=0.2), rnd(cout, seed=205, scale=0.2)
    yn = F().group_norm_apply(y, 32, gam, bet, 1e-5, "silu", stats, lay)
    compare(f"gn_apply conv pp {cin}->{cout}@{hw} v{variant} s{split}", yn, R.group_norm_ref(y, 32, gam, bet, 1e-5, True), *tol(y.dtype, 2.0), kernel=k)


def test_pp_matches_the_other_pipes_on_identical_inputs():
    # same products, different fp32 summation order (channel-slice-major K traversal): results agree to f16 rounding, and a second call is bit-equal
    x, w, b = cl(rnd(4, 320, 64, 64, seed=90)), cl(rnd(320, 320, 3, 3, seed=91, scale=2880 ** -0.5)), rnd(320, seed=92)
    y_ws = F().conv2d(x, w, b, padding=1, variant=22)
    for v in PP_VARIANTS:
        y1 = F().conv2d(x, w, b, padding=1, variant=v)
        k = last_kernel()
        assert "pp" in k.split(",")[-1], k
        assert torch.equal(y1, F().conv2d(x, w, b, padding=1, variant=v)), k  # deterministic
        compare(f"pp v{v} vs ws 128x160", y1, y_ws.float(), *tol(x.dtype), kernel=k)


@pytest.mark.parametrize("split", [2, 4, 12])
def test_conv_split_k(split):
    x, w, b = cl(rnd(2, 1280, 8, 8, seed=76)), cl(rnd(1280, 1280, 3, 3, seed=77, scale=11520 ** -0.5)), rnd(1280, seed=78)
    z = cl(rnd(2, 1280, 8, 8, seed=79))
    y = F().conv2d(x, w, b, z=z, padding=1, split_k=split, variant=4)
    assert f"split={split}" in last_kernel()
    compare(f"conv split{split}", y, R.conv2d_ref(x, w, b, z, 1.0, 1, 1), *tol(x.dtype, 2.0), kernel=last_kernel())
    for v in (11, 13, 21, 22, 23):
        y = F().conv2d(x, w, b, z=z, padding=1, split_k=split, variant=v)
        assert f"split={split},{'ws' if v > 20 else 'dma'}" in last_kernel()
        compare(f"conv dma v{v} split{split}", y, R.conv2d_ref(x, w, b, z, 1.0, 1, 1), *tol(x.dtype, 2.0), kernel=last_kernel())


@needs_probes
@pytest.mark.parametrize("variant,split", [(1, 3), (4, 12), (2, 4), (21, 6), (22, 8), (23, 12), (24, 3), (25, 5), (32, 5), (34, 10)])
def test_split_k_join_matches_the_reduce_kernel(variant, split, monkeypatch):
    """Split-K finished inside the GEMM kernel (ticket counters, sfast_hip.h SFAST_EXT_WS_TICKETS) against the two-launch form (fp32
    row slabs + splitk_reduce_kernel): both sum the splits in order 0 .. S-1, so the outputs agree to the last bit -- whichever
    workgroup happened to draw the last ticket, launch after launch."""
    x, w, b = cl(rnd(2, 1280, 16, 16, seed=176)), cl(rnd(1280, 1280, 3, 3, seed=177, scale=11520 ** -0.5)), rnd(1280, seed=178)
    z = cl(rnd(2, 1280, 16, 16, seed=179))
    fm = F()
    monkeypatch.setattr(fm, "SPLITK_JOIN", True)
    run = lambda: fm.conv2d(x, w, b, z=z, padding=1, act="silu", res_before_act=True, split_k=split, variant=variant)
    y = run()
    k = last_kernel()
    assert f"split={split}," in k and "+join" in k, k
    for _ in range(4):
        assert torch.equal(run(), y), k
    monkeypatch.setattr(fm, "SPLITK_JOIN", False)
    y2 = run()
    k2 = last_kernel()
    assert f"split={split}," in k2 and "+join" not in k2, k2
    ref = R.conv2d_ref(x, w, b, z, 1.0, 1, 1, act="silu", res_before_act=True)
    compare(f"conv join v{variant} split{split}", y, ref, *tol(x.dtype, 2.0), kernel=k)
    compare(f"conv reduce v{variant} split{split}", y2, ref, *tol(x.dtype, 2.0), kernel=k2)
    assert float((y.float() - y2.float()).abs().max()) <= 2e-3, (k, k2)


@needs_probes
@pytest.mark.parametrize("variant,split", [(21, 3), (23, 6), (1, 4), (3, 2)])
def test_split_k_join_linear_geglu_and_ragged(variant, split, monkeypatch):
    """GEGLU (both halves travel through the slabs) and a problem whose tiles hang over M and N."""
    monkeypatch.setattr(F(), "SPLITK_JOIN", True)
    x, w, b = rnd(256, 1280, seed=36), rnd(2 * 640, 1280, seed=37, scale=1280 ** -0.5), rnd(2 * 640, seed=38)
    y = F().linear(x, w, b, geglu=True, variant=variant, split_k=split)
    k = last_kernel()
    if "geglu" in k and f"split={split}," in k:
        assert "+join" in k, k
        compare(f"geglu join v{variant} split{split}", y, R.linear_ref(x, w, b, geglu=True), *tol(x.dtype), kernel=k)
    x, w, b, r = rnd(130, 5128, seed=43), rnd(1284, 5128, seed=44, scale=5128 ** -0.5), rnd(1284, seed=45), rnd(130, 1284, seed=46)
    y = F().linear(x, w, b, residual=r, variant=variant, split_k=split)
    k = last_kernel()
    assert "+join" in k, k
    compare(f"linear ragged join v{variant} split{split}", y, R.linear_ref(x, w, b, residual=r), *tol(x.dtype), kernel=k)


@needs_probes
@pytest.mark.parametrize("variant,split", [(21, 4), (23, 6), (3, 3)])
def test_split_k_join_emits_groupnorm_statistics(variant, split, monkeypatch):
    """With the join the workgroup that finishes a tile runs the staged epilogue, statistics included (no reduce-rows kernel)."""
    import numpy as np
    monkeypatch.setattr(F(), "SPLITK_JOIN", True)
    cin, cout, hw, unit = 640, 1280, 16, 20
    x = cl(rnd(2, cin, hw, hw, seed=200, shift=0.5))
    w = cl(rnd(cout, cin, 3, 3, seed=201, scale=(9 * cin) ** -0.5))
    b, z = rnd(cout, seed=202, shift=2.0), cl(rnd(2, cout, hw, hw, seed=203))
    y, stats, lay = F().conv2d(x, w, b, z=z, padding=1, variant=variant, split_k=split, gn_unit=unit)
    k = last_kernel()
    assert "+gnstats+join" in k and f"split={split}," in k, k
    want = _stats_reference(y.permute(0, 2, 3, 1).reshape(-1, cout), lay)
    got = stats.double().cpu().numpy().reshape(want.shape)
    used = ~np.isnan(want)
    assert np.allclose(got[..., 0][used[..., 0]], want[..., 0][used[..., 0]], rtol=1e-4, atol=1e-4), k
    assert np.allclose(got[..., 1][used[..., 1]], want[..., 1][used[..., 1]], rtol=2e-3, atol=1e-2), k
    compare(f"conv join+stats v{variant} s{split}", y, R.conv2d_ref(x, w, b, z, 1.0, 1, 1), *tol(x.dtype, 2.0), kernel=k)


def test_conv_broadcast_z_folds_to_rowbias():
    x, w, b = cl(rnd(2, 320, 32, 32, seed=80)), cl(rnd(320, 320, 3, 3, seed=81, scale=2880 ** -0.5)), rnd(320, seed=82)
    z = rnd(2, 320, 1, 1, seed=83)  # time-embedding broadcast (cudnn_convolution_impl.cc:1045-1053)
    y = F().conv2d(x, w, b, z=z, padding=1)
    assert "igemm_conv" in last_kernel()
    compare("conv z[B,C,1,1]", y, R.conv2d_ref(x, w, b, z, 1.0, 1, 1), *tol(x.dtype, 2.0), kernel=last_kernel())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_conv_in_and_conv_out_kernels(dtype):
    # conv_in: NCHW [B,4,64,64] sample -> NHWC 320 channels (conv_small_c)
    x = rnd(2, 4, 64, 64, dtype=dtype, seed=84)
    w = cl(rnd(320, 4, 3, 3, dtype=dtype, seed=85, scale=1 / 6))
    b = rnd(320, dtype=dtype, seed=86)
    y = F().conv2d(x, w, b, padding=1, channels_last_out=True)
    assert last_kernel() == "conv_small_c"
    compare(f"conv_in {dtype}", y, R.conv2d_ref(x, w, b, None, 1.0, 1, 1), *tol(dtype, 2.0), kernel=last_kernel())
    # conv_out: NHWC 320 -> NCHW 4 (conv_small_n)
    x = cl(rnd(2, 320, 64, 64, dtype=dtype, seed=87))
    w = cl(rnd(4, 320, 3, 3, dtype=dtype, seed=88, scale=2880 ** -0.5))
    b = rnd(4, dtype=dtype, seed=89)
    y = F().conv2d(x, w, b, padding=1, channels_last_out=False)
    assert last_kernel() == "conv_small_n" and y.is_contiguous()
    compare(f"conv_out {dtype}", y, R.conv2d_ref(x, w, b, None, 1.0, 1, 1), *tol(dtype, 2.0), kernel=last_kernel())


def test_conv_naive_nchw_and_dilation():
    x, w, b = rnd(2, 6, 15, 15, seed=90), rnd(10, 6, 3, 3, seed=91, scale=0.2), rnd(10, seed=92)
    y = F().conv2d(x, w, b, padding=2, dilation=2, act="relu")
    assert last_kernel() == "conv_naive" and y.is_contiguous()
    compare("conv naive dil2", y, R.conv2d_ref(x, w, b, None, 1.0, 1, 2, 2, act="relu"), *tol(x.dtype, 2.0), kernel=last_kernel())


# ---- attention -----------------------------------------------------------------------------------------------
ATTN_CASES = [
    # B, H, Sq, Skv, D
    (1, 8, 4096, 4096, 40), (2, 8, 1024, 1024, 80), (2, 8, 256, 256, 160), (2, 8, 64, 64, 160),
    (2, 8, 4096, 77, 40), (2, 8, 1024, 77, 80), (2, 8, 256, 77, 160), (2, 8, 64, 77, 160),
    (1, 10, 4096, 4096, 64), (2, 20, 1024, 77, 64), (1, 4, 70, 130, 64), (1, 2, 100, 100, 128), (2, 3, 33, 1, 40),
    # odd / even tile counts with ragged tails through the software-pipelined loop (K fragments run two tiles ahead)
    (1, 2, 200, 320, 80), (1, 3, 97, 193, 40), (1, 1, 32, 448, 160), (1, 2, 129, 65, 128), (2, 2, 64, 128, 64),
]


def _q64_covers(D, Skv):
    """attention_q64.hip (64 query rows per wave): head dims 40 / 64 / 80, at least two 64-key tiles, no bias."""
    return D in (40, 64, 80) and Skv >= 128


@pytest.mark.parametrize("case", ATTN_CASES, ids=[str(c) for c in ATTN_CASES])
@pytest.mark.parametrize("variant", [0, 2, 4, 32, 62, 64])   # 2 / 4 / 32: the 32-row kernel; 62 / 64: the 64-row kernel, 2 / 4 waves per workgroup
def test_attention(case, variant):
    B, H, Sq, Skv, D = case
    q, k, v = rnd(B, Sq, H, D, seed=100), rnd(B, Skv, H, D, seed=101), rnd(B, Skv, H, D, seed=102)
    o = F().attention(q, k, v, variant=variant)
    if variant in (62, 64) and _q64_covers(D, Skv):
        assert last_kernel() == f"attn_q64[D={D},BQ={(variant - 60) * 64}]", last_kernel()
    elif variant in (2, 4, 32):
        assert "attn_fwd" in last_kernel()
    else:
        assert "attn_fwd" in last_kernel() or "attn_q64" in last_kernel()
    compare(f"attn {case} v{variant}", o, R.attention_ref(q, k, v), *tol(q.dtype), kernel=last_kernel())


Q64_CASES = [
    # B, H, Sq, Skv, D: ragged query blocks (rows past Sq inside a wave's second 32-row block, inside its first, whole idle waves),
    # ragged / odd / even key-tile counts (2 .. 9 tiles), every head dim of the kernel
    (1, 2, 64, 128, 40), (1, 2, 65, 129, 64), (1, 3, 31, 192, 80), (2, 2, 200, 320, 80), (1, 3, 97, 193, 40), (1, 2, 129, 191, 64),
    (1, 1, 257, 448, 40), (1, 2, 300, 576, 64), (2, 4, 512, 512, 80), (1, 8, 1024, 1024, 40), (3, 5, 130, 257, 64),
]


@pytest.mark.parametrize("case", Q64_CASES, ids=[str(c) for c in Q64_CASES])
@pytest.mark.parametrize("variant", [62, 64])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_attention_q64_shapes(case, variant, dtype):
    B, H, Sq, Skv, D = case
    q, k, v = (rnd(B, s_, H, D, dtype=dtype, seed=140 + i) for i, s_ in enumerate((Sq, Skv, Skv)))
    o = F().attention(q, k, v, variant=variant)
    assert "attn_q64" in last_kernel(), last_kernel()
    compare(f"attn q64 {case} v{variant} {dtype}", o, R.attention_ref(q, k, v), *tol(dtype), kernel=last_kernel())
    # identical to rounding with the first-generation kernel on the same inputs (two independent implementations of one contract)
    o32 = F().attention(q, k, v, variant=32)
    assert "attn_fwd" in last_kernel()
    assert rel(o, o32) < (2e-3 if dtype == torch.float16 else 1.5e-2)


def test_attention_q64_on_fused_qkv_views_and_scale():
    B, S, H, D = 2, 1024, 8, 80
    C = H * D
    qkv = rnd(B, S, 3 * C, seed=146)
    q, k, v = (qkv[:, :, i * C:(i + 1) * C].unflatten(2, (H, D)) for i in range(3))
    o = F().attention(q, k, v, variant=64)
    assert "attn_q64" in last_kernel()
    compare("attn q64 fused-qkv views", o, R.attention_ref(q, k, v), *tol(qkv.dtype), kernel=last_kernel())
    o2 = F().attention(q, k, v, scale=0.05, variant=62)
    compare("attn q64 scale", o2, R.attention_ref(q, k, v, 0.05), *tol(qkv.dtype), kernel=last_kernel())


@pytest.mark.parametrize("variant", [32, 62, 64])
def test_attention_reference_maximum_moves(variant):
    """The 64-row kernel keeps a REFERENCE maximum per row that only moves when a tile's maximum exceeds it 64-fold (2^6); the
    32-row kernel rescales whenever any row's maximum grows. Inputs that force the move at chosen tiles -- after O and the denominator
    have accumulated -- for rows in both query blocks of a wave, once and repeatedly (scores growing tile after tile), plus rows that
    never move. (Guide rule 26: a rare data-dependent branch needs an input that takes it and a full-tensor reference.)"""
    B, H, S, D = 1, 2, 1024, 64
    q, k, v = rnd(B, S, H, D, seed=150), rnd(B, S, H, D, seed=151), rnd(B, S, H, D, seed=152)
    k[:, 300] = q[:, 5] * 4.0          # query 5 (first block of wave 0): jump at tile 4
    k[:, 77] = q[:, 200] * 6.0         # query 200: jump at tile 1
    k[:, 1000] = q[:, 40] * 8.0        # query 40 (second block of wave 0): jump at the last tile
    for t in range(2, 16):             # query 700: a new, larger maximum in every tile from tile 2 on
        k[:, 64 * t + 3] = q[:, 700] * (0.5 * t)
    o = F().attention(q, k, v, variant=variant)
    want = R.attention_ref(q, k, v)
    compare(f"attn moving reference v{variant}", o, want, *tol(q.dtype), kernel=last_kernel())
    for row in (5, 40, 200, 700):
        assert float((o[0, row].float() - want[0, row].float()).abs().max()) < 2e-2


def test_attention_bf16_and_scale():
    q, k, v = (rnd(2, 300, 8, 64, dtype=torch.bfloat16, seed=103 + i) for i in range(3))
    o = F().attention(q, k, v, scale=0.2)
    compare("attn bf16 scale", o, R.attention_ref(q, k, v, 0.2), *tol(torch.bfloat16), kernel=last_kernel())


def test_attention_on_fused_qkv_views():
    # q/k/v as strided [B,S,H,D] views of one [B,S,3C] projection buffer (what the engine does)
    B, S, H, D = 2, 1024, 8, 80
    C = H * D
    qkv = rnd(B, S, 3 * C, seed=106)
    q, k, v = (qkv[:, :, i * C:(i + 1) * C].unflatten(2, (H, D)) for i in range(3))
    o = F().attention(q, k, v)
    assert "attn_fwd" in last_kernel()
    compare("attn fused-qkv views", o, R.attention_ref(q, k, v), *tol(qkv.dtype), kernel=last_kernel())


def test_attention_peaked_softmax_rows():
    # forces large running-max jumps between K tiles (online-softmax rescale path)
    B, H, S, D = 1, 2, 512, 64
    q, k, v = rnd(B, S, H, D, seed=107), rnd(B, S, H, D, seed=108), rnd(B, S, H, D, seed=109)
    k[:, 300] = q[:, 5] * 4.0
    k[:, 77] = q[:, 200] * 6.0
    o = F().attention(q, k, v)
    compare("attn peaked", o, R.attention_ref(q, k, v), *tol(q.dtype), kernel=last_kernel())


def test_attention_naive_path():
    q, k, v = (rnd(2, 50, 3, 24, seed=110 + i) for i in range(3))
    o = F().attention(q, k, v)
    assert last_kernel() == "attn_naive"
    compare("attn naive D24", o, R.attention_ref(q, k, v), *tol(q.dtype), kernel=last_kernel())
    q, k, v = (rnd(1, 20, 2, 16, dtype=torch.float32, seed=113 + i) for i in range(3))
    compare("attn naive f32", F().attention(q, k, v), R.attention_ref(q, k, v), 1e-4, 1e-4, kernel=last_kernel())


# ---- elementwise ---------------------------------------------------------------------------------------------------
def test_strided_copy_reference_case():
    # /root/reference/tests/triton/test_torch_ops.py:11-30: permuted 1x4x256x512, contiguous + channels_last
    a = rnd(1, 4, 256, 512, seed=120).permute(0, 1, 3, 2)
    out = torch.ops.sfast_triton.contiguous(a, torch.contiguous_format)
    assert out.is_contiguous() and torch.equal(out, a.contiguous())
    out = torch.ops.sfast_triton.contiguous(a, torch.channels_last)
    assert out.is_contiguous(memory_format=torch.channels_last) and torch.equal(out, a)
    out = torch.ops.sfast_triton.clone(a, torch.preserve_format)
    assert torch.equal(out, a)
    out = torch.ops.sfast_triton.reshape(a, [4, 512 * 256])
    assert torch.equal(out, a.reshape(4, -1))


@pytest.mark.parametrize("shape,perm,tile", [
    ((512, 256), (1, 0), True), ((4096, 4096), (1, 0), True), ((16, 256, 4096), (0, 2, 1), True), ((16, 32, 128, 256), (0, 1, 3, 2), True),
    ((2, 320, 64, 64), "to_cl", True), ((2, 64, 64, 320), "to_nchw", True), ((3, 70, 33), (0, 2, 1), True), ((5, 17, 130, 66), (0, 3, 2, 1), True),
    ((2, 3, 100, 20), (1, 3, 0, 2), None),   # source-contiguous dim of extent 20 lands last: a plain (non-transposing) copy
    ((7, 16, 16), (2, 0, 1), None), ((9, 8, 40), (0, 2, 1), False)])   # last: extent 8 < 16 along the source-contiguous dim -> generic kernel
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_strided_copy_transposes_bit_exact(shape, perm, tile, dtype):
    """Round 4: transposing copies run through an LDS tile (transpose_tile_kernel); every layout the reference's Triton copy is written
    for (triton/ops/copy.py:303-311 sizes + NCHW <-> NHWC), ragged extents and permutations the tile path must NOT take -- bit exact."""
    x = rnd(*shape, dtype=dtype, seed=121)
    if perm == "to_cl":
        src, dst = x, torch.empty_like(x, memory_format=torch.channels_last)
    elif perm == "to_nchw":
        src = x.permute(0, 3, 1, 2)                                    # logical NCHW view of an NHWC tensor
        dst = torch.empty(src.shape, dtype=dtype, device=DEV)
    else:
        src = x.permute(*perm)
        dst = torch.empty(src.shape, dtype=dtype, device=DEV)
    F().strided_copy(src, dst)
    assert torch.equal(dst, src), last_kernel()
    if tile is not None:
        assert ("transpose_tile" in last_kernel()) == tile, last_kernel()


def test_timestep_embedding():
    from oracle.unet_ref import timestep_embedding as ref
    t = torch.tensor([981.0, 1.0, 500.0], device=DEV)
    for dim, flip, shift in ((320, True, 0.0), (256, True, 0.0), (128, False, 1.0)):
        e = F().timestep_embedding(t, dim, flip, shift, dtype=torch.float32)
        compare(f"temb {dim}", e, ref(t, dim, flip, shift), 2e-4, 1e-4)
        e16 = F().timestep_embedding(t, dim, flip, shift, dtype=torch.float16)
        compare(f"temb16 {dim}", e16, ref(t, dim, flip, shift), 1e-3, 1e-3)


def test_cfg_ddim_step():
    ts, coefs = R.ddim_schedule(50)
    lat = rnd(2, 4, 64, 64, seed=121)
    eps = rnd(2, 2, 4, 64, 64, seed=122)
    coef = torch.tensor(coefs[3], dtype=torch.float32, device=DEV)
    unet_in = torch.empty(2, 2, 4, 64, 64, dtype=lat.dtype, device=DEV)
    out = F().cfg_ddim_step(eps, lat, coef, 7.5, unet_in=unet_in)
    want = R.cfg_ddim_ref(eps, lat, coefs[3], 7.5)
    compare("cfg_ddim", out, want, 5e-3, 2e-3)
    assert torch.equal(unet_in[0], out) and torch.equal(unet_in[1], out)


# ---- committed golden vectors ----------------------------------------------------------------------------------------
def test_golden_ops():
    g = torch.load(os.path.join(GOLDEN, "ops.pt"))
    d = lambda t: t.to(DEV)
    c = g["group_norm_silu"]
    compare("golden gn_silu nchw", F().group_norm(d(c["x"]), c["groups"], d(c["weight"]), d(c["bias"]), c["eps"], "silu"), c["y"], 3e-3, 2e-3)
    compare("golden gn_silu nhwc", F().group_norm(cl(d(c["x"])), c["groups"], d(c["weight"]), d(c["bias"]), c["eps"], "silu"), c["y"], 3e-3, 2e-3)
    c = g["group_norm"]
    compare("golden gn", F().group_norm(cl(d(c["x"])), c["groups"], d(c["weight"]), d(c["bias"]), c["eps"]), c["y"], 3e-3, 2e-3)
    c = g["layer_norm"]
    compare("golden ln", F().layer_norm(d(c["x"]), (320,), d(c["weight"]), d(c["bias"]), c["eps"]), c["y"], 3e-3, 2e-3)
    c = g["geglu"]
    compare("golden geglu", torch.ops.sfast.cutlass_linear_geglu_unified(d(c["x"]), d(c["weight"]), d(c["bias"])), c["y"], 3e-3, 2e-3)
    c = g["linear_add"]
    compare("golden linear_add", torch.ops.sfast.cublas_lowp_linear_add(d(c["x"]), d(c["weight"]), d(c["bias"]), d(c["other"]), c["alpha"]), c["y"], 3e-3, 2e-3)
    c = g["conv3x3_bias_add"]
    y = torch.ops.sfast.cudnn_convolution_bias_add(cl(d(c["x"])), cl(d(c["weight"])), d(c["bias"]), cl(d(c["z"])), c["alpha"], [1, 1], [1, 1], [1, 1], False, [0, 0], 1)
    compare("golden conv_bias_add", y, c["y"], 3e-3, 2e-3, kernel=last_kernel())
    c = g["conv3x3_s2_relu"]
    y = torch.ops.sfast.cudnn_convolution_bias_relu(d(c["x"]), d(c["weight"]), d(c["bias"]), [2, 2], [1, 1], [1, 1], False, [0, 0], 1)
    compare("golden conv_s2_relu (NCHW)", y, c["y"], 3e-3, 2e-3, kernel=last_kernel())
    c = g["attention_d40_kv77"]
    y = torch.ops.sfast_xformers.memory_efficient_attention(d(c["q"]), d(c["k"]), d(c["v"]), None, 0.0, None, None)
    compare("golden attention", y, c["y"], 3e-3, 2e-3, kernel=last_kernel())

# ---- grouped GEMV (the 22 time_emb_proj layers in one launch) ------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,rows", [(2, 1280, [320, 320, 640, 640, 1280, 1280, 1280, 1280, 640, 320]), (1, 1280, [1280] * 22),
                                      (16, 1280, [320] * 32), (3, 256, [8, 24, 40]), (2, 1280, [320])])
def test_gemv_grouped_matches_per_layer_linear(dtype, M, K, rows):
    x = rnd(M, K, dtype=dtype, seed=70)
    ws = [rnd(n, K, dtype=dtype, seed=71 + i, scale=K ** -0.5) for i, n in enumerate(rows)]
    bs = [rnd(n, dtype=dtype, seed=171 + i) if i % 3 else None for i, n in enumerate(rows)]
    y = F().gemv_grouped(x, ws, bs)
    want = torch.cat([R.linear_ref(x, w, b) for w, b in zip(ws, bs)], dim=1)
    compare(f"gemv_grouped M={M} K={K} G={len(rows)} {dtype}", y, want, *tol(dtype), kernel=last_kernel())
    assert "gemv_grouped" in last_kernel()
    # bitwise equal to the per-layer GEMV path (same per-column summation order)
    one = torch.cat([F().linear(x, w, b, variant=101) for w, b in zip(ws, bs)], dim=1)
    assert torch.equal(y, one)


def test_gemv_grouped_activations_and_validation():
    x = rnd(2, 512, seed=80)
    ws = [rnd(64, 512, seed=81, scale=0.05), rnd(128, 512, seed=82, scale=0.05)]
    y = F().gemv_grouped(x, ws, None, act="silu", in_act="silu")
    want = torch.cat([R.linear_ref(x, w, None, "silu", in_act="silu") for w in ws], dim=1)
    compare("gemv_grouped silu/silu", y, want, *tol(x.dtype), kernel=last_kernel())
    from sfast.hip.lib import SfastHipError
    with pytest.raises(SfastHipError):
        F().gemv_grouped(x, [ws[0]] * 33)

# ---- grouped GEMM (cross-attention K/V projections of one UNet level in one launch) -------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,C_,G", [(154, 768, 320, 5), (154, 768, 1280, 6), (77, 768, 640, 5), (154, 2048, 1280, 60), (154, 2048, 640, 10),
                                     (1232, 768, 320, 5), (50, 1024, 160, 3)])
def test_linear_grouped_matches_per_block_linear(dtype, M, K, C_, G):
    x = rnd(M, K, dtype=dtype, seed=90)
    groups = [[rnd(C_, K, dtype=dtype, seed=100 + 2 * g, scale=K ** -0.5), rnd(C_, K, dtype=dtype, seed=101 + 2 * g, scale=K ** -0.5)]
              for g in range(G)]
    outs = F().linear_grouped(x, groups)
    k = last_kernel()
    assert "igemm_grouped" in k
    for g in (0, G // 2, G - 1):
        want = R.linear_ref(x, torch.cat(groups[g], 0))
        compare(f"linear_grouped M={M} K={K} C={C_} g={g}/{G} {dtype}", outs[g], want, *tol(dtype), kernel=k)
    # same arithmetic as the single-problem register pipe with the same tile
    one = F().linear(x, groups[G - 1], variant=3 if "64x64" in k else 1, split_k=1)
    assert torch.equal(outs[G - 1], one)


def test_linear_grouped_bias_act_single_segment():
    x = rnd(96, 256, seed=110)
    ws = [[rnd(192, 256, seed=111 + g, scale=0.06)] for g in range(4)]
    bs = [rnd(192, seed=121 + g) for g in range(4)]
    outs = F().linear_grouped(x, ws, bs, act="relu")
    for g in range(4):
        compare(f"linear_grouped bias relu g={g}", outs[g], R.linear_ref(x, ws[g][0], bs[g], "relu"), *tol(x.dtype), kernel=last_kernel())


# ---- GroupNorm statistics emitted by the producing GEMM / conv epilogue -> one-pass GroupNorm --------------------------------
def _stats_reference(y_nhwc_2d, lay):
    """{mean, M2} records of a [M, N] output in the layout the library reported."""
    import numpy as np
    M, N = y_nhwc_2d.shape
    o = y_nhwc_2d.double().cpu()
    rec = np.full((lay.n_rb, lay.tiles_n, lay.slots, 2), np.nan)
    for rb in range(lay.n_rb):
        blk = o[rb * lay.rb_rows:(rb + 1) * lay.rb_rows]
        for tn in range(lay.tiles_n):
            for j in range(lay.slots):
                U = (tn * lay.bno) // lay.unit + j
                lo, hi = max(tn * lay.bno, U * lay.unit), min(min((tn + 1) * lay.bno, N), (U + 1) * lay.unit)
                if hi > lo:
                    v = blk[:, lo:hi]
                    rec[rb, tn, j, 0] = float(v.mean())
                    rec[rb, tn, j, 1] = float(((v - v.mean()) ** 2).sum())
    return rec


@pytest.mark.parametrize("variant,split", [(0, 0), (1, 1), (2, 1), (3, 1), (5, 1), (11, 1), (12, 1), (13, 1), (21, 1), (22, 1), (23, 1), (24, 1), (25, 1),
                                           (21, 4), (23, 6), (3, 3)])
@pytest.mark.parametrize("cin,cout,hw,unit", [(320, 320, 32, 10), (640, 1280, 16, 20)])
def test_conv_epilogue_emits_groupnorm_statistics(variant, split, cin, cout, hw, unit):
    import numpy as np
    x = rnd(2, cin, hw, hw, seed=200, shift=0.5).contiguous(memory_format=torch.channels_last)
    w = rnd(cout, cin, 3, 3, seed=201, scale=(9 * cin) ** -0.5).contiguous(memory_format=torch.channels_last)
    b = rnd(cout, seed=202, shift=2.0)
    z = rnd(2, cout, hw, hw, seed=203).contiguous(memory_format=torch.channels_last)
    try:
        y, stats, lay = F().conv2d(x, w, b, z=z, padding=1, variant=variant, split_k=split, gn_unit=unit)
    except Exception as e:  # a tile that does not divide H*W cannot emit statistics: the library says so instead of guessing
        assert (hw * hw) % {1: 128, 2: 128, 5: 256, 11: 128, 12: 128, 21: 128, 22: 128, 24: 128}.get(variant, 64) != 0 or "statistics" in str(e), e
        pytest.skip(f"variant {variant}: {e}")
    k = last_kernel()
    assert "+gnstats" in k
    plain = F().conv2d(x, w, b, z=z, padding=1, variant=variant, split_k=split)
    assert torch.equal(y, plain), k  # staged stores / the statistics reduce change nothing in the output
    want = _stats_reference(y.permute(0, 2, 3, 1).reshape(-1, cout), lay)
    got = stats.double().cpu().numpy().reshape(want.shape)
    used = ~np.isnan(want)
    assert np.allclose(got[..., 0][used[..., 0]], want[..., 0][used[..., 0]], rtol=1e-4, atol=1e-4), k
    assert np.allclose(got[..., 1][used[..., 1]], want[..., 1][used[..., 1]], rtol=2e-3, atol=1e-2), k
    # ... and the one-pass GroupNorm over them equals the oracle (and the library's own two-pass GroupNorm to rounding)
    G = 32
    gam, bet = rnd(cout, seed=204, shift=1.0, scale=0.2), rnd(cout, seed=205, scale=0.2)
    yn = F().group_norm_apply(y, G, gam, bet, 1e-5, "silu", stats, lay)
    compare(f"gn_apply conv {cin}->{cout}@{hw} v{variant} s{split}", yn, R.group_norm_ref(y, G, gam, bet, 1e-5, True), *tol(y.dtype, 2.0), kernel=k)


@pytest.mark.parametrize("M,N,K,hw,unit", [(8192, 320, 320, 4096, 10), (2048, 640, 2560, 1024, 10), (512, 1280, 1280, 256, 20)])
@pytest.mark.parametrize("variant", [0, 3, 13, 23, 1, 21])
def test_gemm_epilogue_statistics_and_concat_groupnorm(M, N, K, hw, unit, variant):
    """proj_out-style GEMM + residual emits statistics; consumed alone (C/G = N/32) and as the second half of a virtual concat."""
    x = rnd(M, K, seed=210)
    w = rnd(N, K, seed=211, scale=K ** -0.5)
    b = rnd(N, seed=212)
    res = rnd(M, N, seed=213, shift=-1.0)
    try:
        y, stats, lay = F().linear(x, w, b, residual=res, variant=variant, split_k=1, gn_unit=unit, rows_per_sample=hw)
    except Exception as e:
        pytest.skip(str(e))
    k = last_kernel()
    assert torch.equal(y, F().linear(x, w, b, residual=res, variant=variant, split_k=1))
    B = M // hw
    side = int(hw ** 0.5)
    y4 = y.reshape(B, side, side, N).permute(0, 3, 1, 2)  # channels_last view
    gam, bet = rnd(N, seed=214, shift=1.0, scale=0.1), rnd(N, seed=215)
    yn = F().group_norm_apply(y4, 32, gam, bet, 1e-6, None, stats, lay)
    compare(f"gn_apply gemm {M}x{N}x{K} v{variant}", yn, R.group_norm_ref(y4, 32, gam, bet, 1e-6, False), *tol(y.dtype, 2.0), kernel=k)
    # concat [first | y]: the first source's statistics come from a conv launch
    c1 = 2 * N
    xa = rnd(B, N, side, side, seed=216).contiguous(memory_format=torch.channels_last)
    wa = rnd(c1, N, 1, 1, seed=217, scale=N ** -0.5).contiguous(memory_format=torch.channels_last)
    first, st1, lay1 = F().conv2d(xa, wa, None, gn_unit=unit)
    g2, b2 = rnd(c1 + N, seed=218, shift=1.0, scale=0.1), rnd(c1 + N, seed=219)
    yc = F().group_norm_apply(first, 32, g2, b2, 1e-5, "silu", st1, lay1, x2=y4, stats2=stats, lay2=lay)
    want = R.group_norm_ref(torch.cat([first, y4], 1), 32, g2, b2, 1e-5, True)
    compare(f"gn_apply concat {c1}+{N}@{side} v{variant}", yc, want, *tol(y.dtype, 2.0), kernel=last_kernel())


def test_groupnorm_statistics_survive_a_large_offset():
    """mean >> std: shifted sums inside a tile + pairwise merges across tiles keep the variance (a raw sum / sum-of-squares would not)."""
    x = rnd(1, 320, 64, 64, seed=220, scale=0.05).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(320, 320, 1, 1, device=DEV, dtype=torch.float16)
    w[torch.arange(320), torch.arange(320), 0, 0] = 1.0
    w = w.contiguous(memory_format=torch.channels_last)
    b = torch.full((320,), 60.0, device=DEV, dtype=torch.float16)
    y, stats, lay = F().conv2d(x, w, b, gn_unit=10)
    yn = F().group_norm_apply(y, 32, None, None, 1e-5, None, stats, lay)
    compare("gn_apply offset 60 / std 0.05", yn, R.group_norm_ref(y, 32, None, None, 1e-5, False), 3e-2, 2e-2, kernel=last_kernel())


def test_out_scale_applies_before_the_output_rounding():
    q, kk = rnd(256, 512, seed=230, scale=4.0), rnd(256, 512, seed=231, scale=4.0)
    y = F().linear(q, kk, out_scale=512 ** -0.5)
    want = (q.float() @ kk.float().t()) * 512 ** -0.5
    compare("linear out_scale", y, want, *tol(y.dtype, 4.0), kernel=last_kernel())
    assert torch.isfinite(y).all() and not torch.isfinite(F().linear(q * 8, kk * 8)).all()  # unscaled logits overflow f16 ...
    assert torch.isfinite(F().linear(q * 8, kk * 8, out_scale=2.0 ** -10)).all()            # ... scaled in fp32 they do not


# ---- attention bias / masks (xformers attn_bias, diffusers attention_mask) ----------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,Sq,Skv,D", [(2, 8, 1024, 77, 80), (2, 8, 4096, 77, 40), (1, 10, 1024, 1024, 64), (2, 5, 300, 333, 64),
                                          (1, 8, 256, 256, 160), (2, 4, 64, 130, 128)])
def test_attention_additive_bias_full_tensor(dtype, B, H, Sq, Skv, D):
    q, k, v = (rnd(B, s_, H, D, dtype=dtype, seed=300 + i) for i, s_ in enumerate((Sq, Skv, Skv)))
    # rows of the bias padded to a multiple of 8 elements, as xformers requires of its attn_bias (and diffusers pads its masks):
    # the MFMA kernel reads 4-key groups with dword-aligned loads
    pad = (Skv + 7) // 8 * 8
    bias = rnd(B, H, Sq, pad, dtype=dtype, seed=303, scale=2.0)[..., :Skv]
    y = F().attention(q, k, v, attn_bias=bias)
    kname = last_kernel()
    assert "attn_fwd" in kname and "+bias" in kname
    compare(f"attention bias {(B, H, Sq, Skv, D)} {dtype}", y, R.attention_ref(q, k, v, None, bias), *tol(dtype, 2.0), kernel=kname)
    if Skv % 2:  # a dense odd-width bias is re-laid with padded rows by the wrapper: still the MFMA kernel, same result
        y2 = F().attention(q[:1, :64], k[:1], v[:1], attn_bias=bias[:1, :, :64].contiguous())
        assert "+bias" in last_kernel()
        compare(f"attention bias unaligned {(B, H, Sq, Skv, D)} {dtype}", y2, R.attention_ref(q[:1, :64], k[:1], v[:1], None, bias[:1, :, :64]),
                *tol(dtype, 2.0), kernel=last_kernel())


@pytest.mark.parametrize("Skv,valid", [(77, 60), (128, 64), (200, 1), (77, 77)])
def test_attention_key_padding_mask_broadcast(Skv, valid):
    """diffusers' encoder_attention_mask: [B, Skv] -> additive (1 - mask) * -10000 broadcast over heads and queries; and the hard
    form with -inf, including fully masked 64-key tiles at the tail."""
    B, H, Sq, D = 2, 8, 640, 40
    q, k, v = (rnd(B, s_, H, D, seed=310 + i) for i, s_ in enumerate((Sq, Skv, Skv)))
    keep = torch.zeros(B, Skv, device=DEV)
    keep[0, :valid] = 1
    keep[1, :max(1, valid // 2)] = 1
    for neg in (-10000.0, float("-inf")):
        bias = ((1 - keep) * 1.0).masked_fill(keep == 0, neg).masked_fill(keep == 1, 0.0).half()[:, None, None, :]
        y = F().attention(q, k, v, attn_bias=bias)
        want = R.attention_ref(q, k, v, None, bias)
        compare(f"attention key mask Skv={Skv} valid={valid} neg={neg}", y, want, *tol(q.dtype, 2.0), kernel=last_kernel())
        # equals attention over the kept keys only
        only = F().attention(q[:1], k[:1, :valid], v[:1, :valid])
        assert rel(y[:1], only) < 2e-3
    op = torch.ops.sfast_xformers.memory_efficient_attention(q, k, v, bias, 0.0, None, None)
    assert torch.equal(op, y)


def test_attention_bias_leading_masked_tiles_and_generic_path():
    """-inf over the FIRST tiles of a row (running max still -inf when real keys arrive) and the generic kernel (odd head dim)."""
    B, H, Sq, Skv, D = 1, 2, 128, 256, 64
    q, k, v = (rnd(B, s_, H, D, seed=320 + i) for i, s_ in enumerate((Sq, Skv, Skv)))
    bias = torch.zeros(B, H, Sq, Skv, device=DEV, dtype=torch.float16)
    bias[:, :, :, :192] = float("-inf")
    bias[:, :, 5, :] = 0.0
    y = F().attention(q, k, v, attn_bias=bias)
    compare("attention bias leading -inf tiles", y, R.attention_ref(q, k, v, None, bias), *tol(q.dtype, 2.0), kernel=last_kernel())
    q2, k2, v2 = (rnd(1, s_, 3, 24, seed=330 + i) for i, s_ in enumerate((50, 70, 70)))
    b2 = rnd(1, 3, 50, 70, seed=333)
    y2 = F().attention(q2, k2, v2, attn_bias=b2)
    assert last_kernel() == "attn_naive"
    compare("attention bias generic", y2, R.attention_ref(q2, k2, v2, None, b2), *tol(q2.dtype, 2.0), kernel="attn_naive")


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm())


def test_empty_and_degenerate_inputs():
    """Zero-row / zero-batch inputs return empty tensors of the right shape without a launch (torch semantics; the reference's
    ops inherit them from ATen); an attention without keys is an error, not a NaN tensor."""
    from sfast.hip.lib import SfastHipError
    f = F()
    w, b = rnd(64, 32, seed=1), rnd(64, seed=2)
    y = f.linear(torch.empty(0, 32, device=DEV, dtype=torch.float16), w, b)
    assert y.shape == (0, 64)
    y = f.linear(torch.empty(3, 0, 32, device=DEV, dtype=torch.float16), w, b, act="gelu")
    assert y.shape == (3, 0, 64)
    x0 = torch.empty(0, 32, 8, 8, device=DEV, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    assert f.group_norm(x0, 4, rnd(32, seed=3), rnd(32, seed=4), act="silu").shape == (0, 32, 8, 8)
    assert f.layer_norm(torch.empty(0, 5, 32, device=DEV, dtype=torch.float16), [32], rnd(32, seed=5), rnd(32, seed=6)).shape == (0, 5, 32)
    assert f.conv2d(x0, cl(rnd(16, 32, 3, 3, seed=7)), rnd(16, seed=8), padding=1).shape == (0, 16, 8, 8)
    q = torch.empty(2, 0, 4, 40, device=DEV, dtype=torch.float16)
    k = rnd(2, 77, 4, 40, seed=9)
    assert f.attention(q, k, k).shape == (2, 0, 4, 40)
    with pytest.raises(SfastHipError):
        f.attention(rnd(2, 16, 4, 40, seed=10), k[:, :0], k[:, :0])
    # one row, one key, one channel group: smallest non-empty problems
    y = f.attention(rnd(1, 1, 1, 40, seed=11), k[:1, :1, :1], k[:1, :1, :1])
    compare("attention 1x1", y, k[:1, :1, :1].float(), *tol(torch.float16))


def test_golden_ops_round2():
    """The round-2 entry points against the committed fixtures (tests/golden/ops_r2.pt): no oracle involved at run time."""
    g = torch.load(os.path.join(GOLDEN, "ops_r2.pt"))
    d = lambda t: t.to(DEV)  # noqa: E731
    for name in ("attention_full_bias", "attention_key_padding_mask"):
        c = g[name]
        y = F().attention(d(c["q"]), d(c["k"]), d(c["v"]), attn_bias=d(c["bias"]))
        compare(f"golden {name}", y, c["y"], 3e-3, 2e-3, kernel=last_kernel())
        y = torch.ops.sfast_xformers.memory_efficient_attention(d(c["q"]), d(c["k"]), d(c["v"]), d(c["bias"]).expand(2, 3, 70, 77), 0.0, None, None)
        compare(f"golden {name} via sfast_xformers", y, c["y"], 3e-3, 2e-3)
    for name in ("ddim_step_eps", "euler_step_eps"):
        c = g[name]
        coef = torch.tensor([c["coef"]], dtype=torch.float32, device=DEV)
        y = F().linear_step(d(c["model_output"]), d(c["sample"]), coef, 0)
        compare(f"golden {name}", y, c["y"], 2e-3, 2e-3)
    c = g["euler_step_eps"]
    coef = torch.tensor([[c["scale"], 0.0]], dtype=torch.float32, device=DEV)
    compare("golden euler scale_model_input", F().linear_step(d(c["sample"]), d(c["sample"]), coef, 0), c["y_scaled"], 2e-3, 2e-3)


# ---- round 4: GroupNorm(+SiLU) -> 3x3 conv as one weight-streaming launch (gnconv.hip, sfast_hip_gn_conv2d) --------------------------------
GNCONV_CASES = [
    # name, B, C1, C2, H, W, Cout, extras
    ("sd15 8x8 1280->1280 (resnet conv2 + residual)", 2, 1280, 0, 8, 8, 1280, dict(z=True)),
    ("sd15 8x8 1280->1280 (conv1 + temb row bias)", 2, 1280, 0, 8, 8, 1280, dict(rowbias=True)),
    ("sd15 8x8 cat 1280+1280 -> 1280", 2, 1280, 1280, 8, 8, 1280, dict(rowbias=True)),
    ("literal B=1 8x8 (two 32-pixel blocks)", 1, 1280, 0, 8, 8, 1280, dict(z=True)),
    ("640 -> 320, cpg 40 via 16 groups", 2, 640, 0, 8, 8, 320, dict(groups=16)),
    ("cat 640+640 -> 640, 4x4, B=8", 8, 640, 640, 4, 4, 640, dict(groups=16, z=True, rowbias=True)),
    ("ragged pixels: 2 x 7 x 6 = 84", 2, 320, 0, 7, 6, 64, dict(groups=8)),
    ("no SiLU, eps 1e-6, no affine bias", 2, 1280, 0, 8, 8, 96, dict(gn_act=None, eps=1e-6)),
]


@needs_probes
@pytest.mark.parametrize("case", GNCONV_CASES, ids=[c[0] for c in GNCONV_CASES])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gn_conv2d_fused(case, dtype):
    """One launch against (a) the fp32 oracle of the reference's pair group_norm_silu -> conv + bias + residual, with the normalised
    tensor rounded to the I/O dtype between the two (what the two-operator path stores), and (b) the two HIP operators it replaces."""
    name, B, C1, C2, H, W, Cout, ex = case
    G, eps, gn_act = ex.get("groups", 32), ex.get("eps", 1e-5), ex.get("gn_act", "silu")
    Cin = C1 + C2
    x = cl(rnd(B, C1, H, W, dtype=dtype, seed=301, scale=2.0, shift=0.7))
    x2 = cl(rnd(B, C2, H, W, dtype=dtype, seed=302, scale=0.5, shift=-1.0)) if C2 else None
    gw, gb = rnd(Cin, dtype=dtype, seed=303, scale=0.2, shift=1.0), rnd(Cin, dtype=dtype, seed=304, scale=0.2)
    w = cl(rnd(Cout, Cin, 3, 3, dtype=dtype, seed=305, scale=(9 * Cin) ** -0.5))
    b = rnd(Cout, dtype=dtype, seed=306)
    z = cl(rnd(B, Cout, H, W, dtype=dtype, seed=307)) if ex.get("z") else None
    rb = rnd(B, Cout, dtype=dtype, seed=308) if ex.get("rowbias") else None
    assert F().gn_conv2d_supported(x, w, G, x2=x2)
    y = F().gn_conv2d(x, G, gw, gb, w, b, eps=eps, gn_act=gn_act, x2=x2, z=z, rowbias=rb)
    k = last_kernel()
    assert "gnconv" in k, k
    assert torch.equal(y, F().gn_conv2d(x, G, gw, gb, w, b, eps=eps, gn_act=gn_act, x2=x2, z=z, rowbias=rb)), "not reproducible"
    xc = x if x2 is None else torch.cat([x, x2], 1)
    n32 = R.group_norm_ref(xc, G, gw, gb, eps, gn_act == "silu")
    want = R.conv2d_ref(n32.to(dtype), w, b, z, 1.0, 1, 1, rowbias=rb)
    compare(f"gn_conv2d {name} {dtype}", y, want, *tol(dtype, 3.0), kernel=k)
    n = F().group_norm(x, G, gw, gb, eps, gn_act, x2=x2)
    y2 = F().conv2d(n, w, b, z=z, padding=1, rowbias=rb)
    compare(f"gn_conv2d vs two operators {name} {dtype}", y, y2.float(), *tol(dtype, 3.0), kernel=k)


@needs_probes
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gn_conv2d_without_groupnorm_is_the_plain_conv(dtype):
    """groups = 0: the same weight-streaming launch without the normalisation ("wsconv") -- measured against the tuned ring kernels and not
    selected anywhere (profiles/r04_wdirect_wsconv_run15.log: 20.6 vs 16.5 us at the 8x8 level); kept as the probe it is."""
    x = cl(rnd(2, 1280, 8, 8, dtype=dtype, seed=311))
    x2 = cl(rnd(2, 1280, 8, 8, dtype=dtype, seed=312))
    w = cl(rnd(640, 2560, 3, 3, dtype=dtype, seed=313, scale=(9 * 2560) ** -0.5))
    b, rb = rnd(640, dtype=dtype, seed=314), rnd(2, 640, dtype=dtype, seed=315)
    y = F().gn_conv2d(x, 0, None, None, w, b, x2=x2, rowbias=rb)
    assert "wsconv" in last_kernel(), last_kernel()
    compare(f"wsconv {dtype}", y, R.conv2d_ref(x, w, b, None, 1.0, 1, 1, x2=x2, rowbias=rb), *tol(dtype, 2.0), kernel=last_kernel())


def test_gn_conv2d_refuses_what_it_does_not_cover():
    """Probe build: shapes outside the fused kernel's coverage are refused. Product build (round 5: the fused launch measured slower and
    lives in the probe library only): EVERYTHING is refused -- supported() == 0, the call raises SFAST_ERR_UNSUPPORTED, nothing launches."""
    from sfast.hip import lib
    if not _probes_loaded():
        x8, w8 = cl(rnd(2, 1280, 8, 8, seed=318)), cl(rnd(1280, 1280, 3, 3, seed=319, scale=0.01))
        assert not F().gn_conv2d_supported(x8, w8, 32)
        with pytest.raises(lib.SfastHipError):
            F().gn_conv2d(x8, 32, None, None, w8)
    x = cl(rnd(2, 1280, 16, 16, seed=311))               # 512 pixels
    w = cl(rnd(1280, 1280, 3, 3, seed=312, scale=0.01))
    assert not F().gn_conv2d_supported(x, w, 32)
    with pytest.raises(lib.SfastHipError):
        F().gn_conv2d(x, 32, None, None, w)
    x = cl(rnd(2, 1920, 8, 8, seed=313))                 # 60 channels per group: not a multiple of 8
    assert not F().gn_conv2d_supported(x, cl(rnd(1280, 1920, 3, 3, seed=314, scale=0.01)), 32)
    x1, x2 = cl(rnd(2, 1280, 8, 8, seed=315)), cl(rnd(2, 640, 8, 8, seed=316))   # concat 1280 + 640: groups of 60 straddle the sources
    assert not F().gn_conv2d_supported(x1, cl(rnd(1280, 1920, 3, 3, seed=317, scale=0.01)), 32, x2=x2)


@needs_probes
# ---- round 4: the consumer GroupNorm(+SiLU) computed by the split-K reduce launch (sfast_epilogue_ext.gn_out) -------------------------------
@pytest.mark.parametrize("case", [
    # name, B, Cin, H, W, Cout, split, groups, act, extras
    ("8x8 1280->1280 split 12 + temb", 2, 1280, 8, 8, 1280, 12, 32, "silu", dict(rowbias=True)),
    ("8x8 1280->1280 split 12 + residual, plain GN eps 1e-6", 2, 1280, 8, 8, 1280, 12, 32, None, dict(z=True, eps=1e-6)),
    ("16x16 1280->1280 split 6", 2, 1280, 16, 16, 1280, 6, 32, "silu", dict(z=True)),
    ("16x16 640->1280 split 3", 2, 640, 16, 16, 1280, 3, 32, "silu", dict(rowbias=True)),
    ("stride-2 downsampler 16->8, split 8", 2, 1280, 16, 16, 1280, 8, 32, "silu", dict(stride=2)),
    ("B=1 8x8 split 5, 16 groups", 1, 640, 8, 8, 640, 5, 16, "silu", {}),
    ("ragged 7x6, Cout 96 (3 channels... 24 per group)", 3, 320, 7, 6, 96, 4, 4, "silu", dict(z=True)),
], ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_conv_split_k_reduce_carries_the_consumer_groupnorm(case, dtype):
    """(y, n) from ONE conv call: y bit-identical to the same conv without the fused GroupNorm, n == GroupNorm(+SiLU) of the STORED y
    within one output rounding (both against the fp32 oracle on y and against the separate HIP GroupNorm launch)."""
    name, B, Cin, H, W, Cout, split, G, act, ex = case
    stride, eps = ex.get("stride", 1), ex.get("eps", 1e-5)
    x = cl(rnd(B, Cin, H, W, dtype=dtype, seed=401))
    w = cl(rnd(Cout, Cin, 3, 3, dtype=dtype, seed=402, scale=(9 * Cin) ** -0.5))
    b = rnd(Cout, dtype=dtype, seed=403)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    z = cl(rnd(B, Cout, Ho, Wo, dtype=dtype, seed=404)) if ex.get("z") else None
    rb = rnd(B, Cout, dtype=dtype, seed=405) if ex.get("rowbias") else None
    gw, gb = rnd(Cout, dtype=dtype, seed=406, scale=0.2, shift=1.0), rnd(Cout, dtype=dtype, seed=407, scale=0.2)
    y, n = F().conv2d(x, w, b, z=z, stride=stride, padding=1, rowbias=rb, split_k=split, gn=(G, gw, gb, eps, act))
    k = last_kernel()
    assert "+gn" in k and f"split={split}," in k, k
    y0 = F().conv2d(x, w, b, z=z, stride=stride, padding=1, rowbias=rb, split_k=split)
    assert torch.equal(y, y0), "the fused reduce changed the conv output"
    y_again, n_again = F().conv2d(x, w, b, z=z, stride=stride, padding=1, rowbias=rb, split_k=split, gn=(G, gw, gb, eps, act))
    assert torch.equal(n, n_again) and torch.equal(y, y_again), "not reproducible"
    compare(f"conv+gn y {name} {dtype}", y, R.conv2d_ref(x, w, b, z, 1.0, stride, 1, rowbias=rb), *tol(dtype, 2.0), kernel=k)
    compare(f"conv+gn n {name} {dtype}", n, R.group_norm_ref(y, G, gw, gb, eps, act == "silu"), *tol(dtype, 2.0), kernel=k)
    n2 = F().group_norm(y, G, gw, gb, eps, act)
    compare(f"conv+gn n vs separate launch {name} {dtype}", n, n2.float(), *tol(dtype, 1.0), kernel=k)


def test_fused_groupnorm_epilogue_refuses_unsplit_plans():
    """sfast_epilogue_ext.gn_out is refused BEFORE the conv is launched (no partial side effects): `out` stays untouched. In the
    product build (the fused reduce lives in the probe library) that holds for split plans too."""
    from sfast.hip import lib
    x, w = cl(rnd(2, 1280, 8, 8, seed=415)), cl(rnd(1280, 1280, 3, 3, seed=416, scale=0.02))
    sentinel = torch.full((2, 1280, 8, 8), 7.0, dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
    with pytest.raises(lib.SfastHipError):   # split 1: no reduce launch to ride in, whichever build
        F().conv2d(x, w, None, padding=1, split_k=1, gn=(32, None, None, 1e-5, "silu"), out=sentinel)
    torch.cuda.synchronize()
    assert bool((sentinel == 7.0).all()), "the conv was launched before the fused GroupNorm epilogue was refused"
    if not _probes_loaded():
        with pytest.raises(lib.SfastHipError):
            F().conv2d(x, w, None, padding=1, split_k=12, gn=(32, None, None, 1e-5, "silu"), out=sentinel)
        torch.cuda.synchronize()
        assert bool((sentinel == 7.0).all())
    x, w = cl(rnd(2, 320, 32, 32, seed=411)), cl(rnd(320, 320, 3, 3, seed=412, scale=0.02))
    with pytest.raises(lib.SfastHipError):   # 32x32: 1024 pixels x 10 channels per group: fine, but split 1 has no reduce launch to ride in
        F().conv2d(x, w, None, padding=1, split_k=1, variant=4, gn=(32, None, None, 1e-5, "silu"))
    x, w = cl(rnd(2, 640, 64, 64, seed=413)), cl(rnd(640, 640, 3, 3, seed=414, scale=0.02))
    with pytest.raises(lib.SfastHipError):   # 4096 pixels x 20 channels per group > 16384 values per workgroup
        F().conv2d(x, w, None, padding=1, split_k=2, gn=(32, None, None, 1e-5, "silu"))
