"""ORACLE -- test infrastructure only. The case list shared by `oracle/ref_triton_run.py` (which runs the reference's own
Triton kernels, staged by `oracle/make_ref.py`, in a subprocess) and by the tests that compare the oracle restatement and
the HIP kernels with those outputs. Inputs are generated from CPU generators with fixed seeds, so both sides build
bit-identical tensors without shipping them.

Shapes: the reference's own self-check shapes (triton/ops/group_norm.py:485-523 `randn(2,320,32,32)` both layouts;
layer_norm.py:522 `(1151, 8192)`; copy.py:303-311 transposes; tests/triton/test_torch_ops.py:16 `ones(1,4,256,512)`
permuted) plus the SD1.5 / SDXL / VAE layer shapes of the hot path. `small=True` rows are the ones whose reference OUTPUTS
are committed as the fixture `tests/golden/ref_triton_small.pt` (kept under 1 MB).
"""
import torch

DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def _randn(shape, seed, scale=1.0, shift=0.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale + shift


GN_CASES = [
    # name, (N, C, H, W), groups, eps, silu, channels_last, dtype, seed, scale, shift, small
    dict(name="gn_selfcheck_nchw", shape=(2, 320, 32, 32), groups=32, eps=1e-5, silu=False, cl=False, dtype="f16", seed=101),
    dict(name="gn_selfcheck_cl", shape=(2, 320, 32, 32), groups=32, eps=1e-5, silu=False, cl=True, dtype="f16", seed=101),
    dict(name="gn_silu_selfcheck_nchw", shape=(2, 320, 32, 32), groups=32, eps=1e-5, silu=True, cl=False, dtype="f16", seed=102),
    dict(name="gn_silu_selfcheck_cl", shape=(2, 320, 32, 32), groups=32, eps=1e-5, silu=True, cl=True, dtype="f16", seed=102),
    dict(name="gn_silu_sd15_64", shape=(2, 320, 64, 64), groups=32, eps=1e-5, silu=True, cl=True, dtype="f16", seed=103, scale=2.0, shift=1.0),
    dict(name="gn_silu_sd15_32", shape=(2, 640, 32, 32), groups=32, eps=1e-5, silu=True, cl=True, dtype="f16", seed=104, scale=2.0, shift=-1.0),
    dict(name="gn_silu_sd15_16", shape=(2, 1280, 16, 16), groups=32, eps=1e-5, silu=True, cl=True, dtype="f16", seed=105, scale=3.0),
    dict(name="gn_silu_sd15_8", shape=(2, 1280, 8, 8), groups=32, eps=1e-5, silu=True, cl=True, dtype="f16", seed=106, scale=3.0),
    dict(name="gn_silu_sd15_cat_8", shape=(2, 2560, 8, 8), groups=32, eps=1e-5, silu=True, cl=True, dtype="f16", seed=107),
    dict(name="gn_silu_sd15_cat_32", shape=(2, 960, 32, 32), groups=32, eps=1e-5, silu=True, cl=True, dtype="f16", seed=108),
    dict(name="gn_tfm_eps1e6_64", shape=(2, 320, 64, 64), groups=32, eps=1e-6, silu=False, cl=True, dtype="f16", seed=109),
    dict(name="gn_tfm_eps1e6_16", shape=(2, 1280, 16, 16), groups=32, eps=1e-6, silu=False, cl=True, dtype="f16", seed=110),
    dict(name="gn_silu_sdxl_128", shape=(2, 320, 128, 128), groups=32, eps=1e-5, silu=True, cl=True, dtype="f16", seed=111),
    dict(name="gn_silu_vae_c128", shape=(1, 128, 64, 64), groups=32, eps=1e-6, silu=True, cl=True, dtype="f16", seed=112),
    dict(name="gn_silu_bf16", shape=(2, 640, 16, 16), groups=32, eps=1e-5, silu=True, cl=True, dtype="bf16", seed=113),
    dict(name="gn_small_cl", shape=(2, 64, 8, 8), groups=32, eps=1e-5, silu=False, cl=True, dtype="f16", seed=114, small=True),
    dict(name="gn_silu_small_cl", shape=(2, 64, 8, 8), groups=32, eps=1e-5, silu=True, cl=True, dtype="f16", seed=115, shift=0.5, small=True),
    dict(name="gn_silu_small_nchw", shape=(2, 64, 6, 10), groups=16, eps=1e-6, silu=True, cl=False, dtype="f16", seed=116, scale=2.0, small=True),
    dict(name="gn_small_wide_groups", shape=(1, 160, 8, 8), groups=32, eps=1e-5, silu=False, cl=True, dtype="f16", seed=117, shift=-2.0, small=True),
]

LN_CASES = [
    # name, (M, N), eps, dtype, seed  -- inputs follow layer_norm.py:410-413: x = -2.3 + 0.5 randn, w/b = rand
    dict(name="ln_selfcheck", shape=(1151, 8192), eps=1e-5, dtype="f16", seed=201),
    dict(name="ln_sd15_64", shape=(8192, 320), eps=1e-5, dtype="f16", seed=202),
    dict(name="ln_sd15_32", shape=(2048, 640), eps=1e-5, dtype="f16", seed=203),
    dict(name="ln_sd15_16", shape=(512, 1280), eps=1e-5, dtype="f16", seed=204),
    dict(name="ln_sd15_8", shape=(128, 1280), eps=1e-5, dtype="f16", seed=205),
    dict(name="ln_sdxl", shape=(8192, 640), eps=1e-5, dtype="f16", seed=206),
    dict(name="ln_bf16", shape=(512, 1280), eps=1e-5, dtype="bf16", seed=207),
    dict(name="ln_small", shape=(24, 320), eps=1e-5, dtype="f16", seed=208, small=True),
    dict(name="ln_small_odd", shape=(7, 200), eps=1e-6, dtype="f16", seed=209, small=True),
]

COPY_CASES = [
    # name, shape of the SOURCE before transpose(-1,-2) / permute, dtype, seed
    dict(name="copy_2d_t", shape=(512, 256), op="t", dtype="f16", seed=301),
    dict(name="copy_2d_t_f32", shape=(512, 256), op="t", dtype="f32", seed=302),
    dict(name="copy_2d_big", shape=(4096, 4096), op="t", dtype="f16", seed=303),
    dict(name="copy_3d_t", shape=(16, 256, 4096), op="t", dtype="f16", seed=304),
    dict(name="copy_4d_t", shape=(16, 32, 128, 256), op="t", dtype="f16", seed=305),
    dict(name="copy_4d_perm", shape=(1, 4, 256, 512), op="perm0132", dtype="f32", seed=306),  # tests/triton/test_torch_ops.py:16
    dict(name="copy_nchw_to_nhwc", shape=(2, 320, 64, 64), op="to_cl", dtype="f16", seed=307),
    dict(name="copy_small", shape=(3, 20, 12), op="t", dtype="f16", seed=308, small=True),
]

CONV_CASES = [
    # name, x (N,C,H,W), w (O,I,kh,kw), stride, padding, bias, dtype, seed, channels_last
    dict(name="conv3x3_320", x=(2, 320, 32, 32), w=(320, 320, 3, 3), stride=1, padding=1, bias=True, dtype="f16", seed=401, cl=True),
    dict(name="conv1x1_640", x=(2, 640, 16, 16), w=(320, 640, 1, 1), stride=1, padding=0, bias=True, dtype="f16", seed=402, cl=True),
    dict(name="conv3x3_s2", x=(1, 64, 16, 16), w=(128, 64, 3, 3), stride=2, padding=1, bias=True, dtype="f16", seed=403, cl=True, small=True),
    dict(name="conv3x3_nchw", x=(1, 32, 12, 12), w=(48, 32, 3, 3), stride=1, padding=1, bias=False, dtype="f16", seed=404, cl=False, small=True),
]


def gn_inputs(c):
    x = _randn(c["shape"], c["seed"], c.get("scale", 1.0), c.get("shift", 0.0)).to(DT[c["dtype"]])
    C = c["shape"][1]
    w = _randn((C,), c["seed"] + 1000, 0.3, 1.0).to(DT[c["dtype"]])
    b = _randn((C,), c["seed"] + 2000, 0.3, 0.0).to(DT[c["dtype"]])
    if c["cl"]:
        x = x.contiguous(memory_format=torch.channels_last)
    return x, w, b


def ln_inputs(c):
    M, N = c["shape"]
    g = torch.Generator(device="cpu").manual_seed(c["seed"])
    w = torch.rand(N, generator=g).to(DT[c["dtype"]])
    b = torch.rand(N, generator=g).to(DT[c["dtype"]])
    x = (-2.3 + 0.5 * torch.randn(M, N, generator=g)).to(DT[c["dtype"]])
    return x, w, b


def copy_inputs(c):
    x = _randn(c["shape"], c["seed"]).to(DT[c["dtype"]])
    return x


def copy_view(c, x):
    """The strided SOURCE view the copy reads, and the memory format of the dense destination."""
    if c["op"] == "t":
        return x.transpose(-1, -2), torch.contiguous_format
    if c["op"] == "perm0132":
        return x.permute(0, 1, 3, 2), torch.contiguous_format
    if c["op"] == "to_cl":
        return x, torch.channels_last
    raise ValueError(c["op"])


def conv_inputs(c):
    x = _randn(c["x"], c["seed"]).to(DT[c["dtype"]])
    fan_in = c["w"][1] * c["w"][2] * c["w"][3]
    w = _randn(c["w"], c["seed"] + 1000, fan_in ** -0.5).to(DT[c["dtype"]])
    b = _randn((c["w"][0],), c["seed"] + 2000, 0.1).to(DT[c["dtype"]]) if c["bias"] else None
    if c["cl"]:
        x = x.contiguous(memory_format=torch.channels_last)
        w = w.contiguous(memory_format=torch.channels_last)
    return x, w, b


ALL = {"gn": GN_CASES, "ln": LN_CASES, "copy": COPY_CASES, "conv": CONV_CASES}
