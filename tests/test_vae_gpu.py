"""VAE decoder path on a real MI355X: row softmax kernel, the native decoder engine
(eager plan and hipGraph replay, and behind sfast.compilers.compile_vae()) vs the fp32 oracle decoder.

Tolerances: as for the UNet (tests/test_unet_gpu.py) -- relative L2 vs the fp32 oracle <= 4e-3 and not worse than 1.5x
the error of the same restatement run eagerly in fp16 on the GPU; every measured value is logged to
gpurun_out/parity.jsonl.
"""
import os
import types

import pytest
import torch

from oracle import vae_ref as V
from parity import compare, log_value, rel_l2

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N", [(37, 64), (64, 64), (256, 1024), (300, 4096), (8, 8192)])
def test_softmax_rows(M, N, dtype):
    from sfast.hip import functional as F
    from sfast.hip.lib import last_kernel
    g = torch.Generator().manual_seed(M * 7 + N)
    x = (torch.randn(M, N, generator=g) * 6).to(DEV, dtype)
    scale = N ** -0.5 * 3
    y = F.softmax_rows(x, scale)
    want = torch.softmax(x.float() * scale, dim=-1)
    tol = (2e-3, 2e-2) if dtype == torch.float16 else (1e-2, 4e-2)
    compare(f"softmax_rows {M}x{N} {dtype}", y, want, *tol, kernel=last_kernel())
    assert abs(float(y.float().sum(-1).mean()) - 1.0) < 2e-2
    # in place, with a row stride larger than N
    buf = torch.zeros(M, N + 64, device=DEV, dtype=dtype)
    buf[:, :N] = x
    view = buf[:, :N]
    F.softmax_rows(view, scale, out=view)
    compare(f"softmax_rows in place {M}x{N} {dtype}", buf[:, :N], want, *tol, kernel=last_kernel())
    assert float(buf[:, N:].abs().max()) == 0.0


def test_softmax_rows_rejects_bad_layouts():
    from sfast.hip import functional as F
    from sfast.hip.lib import SfastHipError
    with pytest.raises(SfastHipError):
        F.softmax_rows(torch.zeros(4, 12, device=DEV, dtype=torch.float16))  # N % 8 != 0
    with pytest.raises(SfastHipError):
        F.softmax_rows(torch.zeros(4, 16))  # CPU tensor: there is no CPU path


def _engine(model):
    from sfast.engine import VaeDecoderEngine
    return VaeDecoderEngine.from_module(model)


def test_tiny_decoder_matches_golden():
    gold = torch.load(os.path.join(GOLDEN, "vae_tiny.pt"))
    m = V.build("tiny", seed=gold["seed"], dtype=torch.float16, device=DEV, **gold["config"])
    eng = _engine(m)
    y = eng.forward(gold["z"].to(DEV))
    err = rel_l2(y.float().cpu(), gold["y"])
    log_value("vae_tiny vs golden", rel_l2=err)
    assert torch.isfinite(y).all() and err < 4e-3, err


def test_sd_decoder_parity_256px():
    """Full SD VAE decoder (49.5 M parameters), 32x32 latent -> 256x256 image, batch 1."""
    m = V.build("sd", seed=11, dtype=torch.float16, device=DEV)
    ref = V.build("sd", seed=11)
    ref.load_state_dict({k: v.float().cpu() for k, v in m.state_dict().items()})
    z = torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(3)).to(DEV, torch.float16)
    eng = _engine(m)
    y = eng.forward(z)
    with torch.no_grad():
        want = ref(z.float().cpu())
        eager16 = m(z)
    e_eng, e_eager = rel_l2(y.float().cpu(), want), rel_l2(eager16.float().cpu(), want)
    log_value("sd vae decoder 32x32 latent", engine_vs_fp32=e_eng, eager16_vs_fp32=e_eager,
              engine_vs_eager16=rel_l2(y.float(), eager16.float()))
    assert torch.isfinite(y).all() and y.shape == (1, 3, 256, 256)
    assert e_eng < 4e-3 and e_eng < 1.5 * e_eager + 1e-3, (e_eng, e_eager)
    kinds = {op.kind for op in eng.get_plan(1, 32, 32).ops}
    assert {"attn_vae", "softmax", "conv3x3", "gn_silu"} <= kinds


def test_decoder_graph_replay_and_batch():
    from sfast.engine import capture_plan_graph
    cfg = dict(block_out_channels=(64, 128, 128), norm_num_groups=16, layers_per_block=1)
    m = V.build("tiny", seed=21, dtype=torch.float16, device=DEV, **cfg)
    ref = V.build("tiny", seed=21, device=DEV, **cfg)
    ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
    eng = _engine(m)
    z = torch.randn(3, 4, 16, 8, generator=torch.Generator().manual_seed(5)).to(DEV, torch.float16)
    y = eng.forward(z)
    with torch.no_grad():
        want = ref(z.float())
    err = rel_l2(y.float(), want)
    log_value("vae decoder B=3 16x8 latent", rel_l2=err)
    assert err < 4e-3, err
    plan = eng.get_plan(3, 16, 8)
    s = torch.cuda.Stream()
    graph, _ = capture_plan_graph(plan, s)
    plan.static_out.zero_()
    with torch.cuda.stream(s):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(plan.static_out, y)


@pytest.mark.parametrize("graph", [False, True])
def test_compile_vae_drop_in(graph):
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_vae
    cfg = dict(block_out_channels=(64, 128), norm_num_groups=8, layers_per_block=1)
    dec = V.build("tiny", seed=31, dtype=torch.float16, device=DEV, **cfg)
    eager = V.build("tiny", seed=31, dtype=torch.float16, device=DEV, **cfg)
    vae = types.SimpleNamespace(decoder=dec, device=torch.device(DEV), config=types.SimpleNamespace(norm_num_groups=8),
                                parameters=lambda: dec.parameters(), named_parameters=lambda: dec.named_parameters())
    c = CompilationConfig.Default()
    c.enable_cuda_graph = graph
    c.memory_format = None
    out = compile_vae(vae, c)
    assert out is vae and hasattr(vae, "_sfast_vae_engine")
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(9)).to(DEV, torch.float16)
    with torch.no_grad():
        y = vae.decoder(z)
        want = eager(z)
        y_again = vae.decoder(z)
    err = rel_l2(y.float(), want.float())
    log_value(f"compile_vae() tiny decoder graph={graph} vs eager fp16", rel_l2=err)
    assert err < 1e-2 and torch.equal(y, y_again)
    # a different latent through the cached plan / graph (a stale or empty graph would return the previous image)
    z2 = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(10)).to(DEV, torch.float16)
    with torch.no_grad():
        err2 = rel_l2(vae.decoder(z2).float(), eager(z2).float())
    assert err2 < 1e-2, err2
    # a call the engine does not cover (extra latent_embeds argument) is routed to the original forward
    with pytest.raises(TypeError):
        vae.decoder(z, torch.zeros(1, device=DEV))  # the oracle Decoder takes no latent_embeds: proves the fallback ran


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_image_postprocess_kernel(dtype):
    from sfast.hip import functional as F
    img = (torch.rand(2, 3, 40, 24, generator=torch.Generator().manual_seed(1)) * 2.4 - 1.2).to(DEV, dtype)
    den = (img.float() / 2 + 0.5).clamp(0, 1)
    f = F.image_postprocess(img, denormalize=True, to_uint8=False)
    assert f.shape == (2, 40, 24, 3) and f.dtype == torch.float32
    torch.testing.assert_close(f, den.permute(0, 2, 3, 1), rtol=0, atol=1e-6)
    u = F.image_postprocess(img, denormalize=True, to_uint8=True)
    want = den.permute(0, 2, 3, 1).mul(255).round().to(torch.uint8)
    assert u.dtype == torch.uint8 and torch.equal(u, want)
    raw = F.image_postprocess(den.to(dtype), denormalize=False, to_uint8=True)
    assert torch.equal(raw, den.to(dtype).float().permute(0, 2, 3, 1).mul(255).round().to(torch.uint8))


def test_image_processor_patch_gpu():
    from sfast.libs.diffusers.image_processor import patch_image_prcessor
    proc = patch_image_prcessor(types.SimpleNamespace(config=types.SimpleNamespace(do_normalize=True), postprocess=None,
                                                      pt_to_numpy=None, pt_to_pil=None))
    img = (torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(2)) * 2 - 1).to(DEV, torch.float16)
    pil = proc.postprocess(img, "pil")
    import numpy as np
    got = np.asarray(pil[0])
    # the reference denormalises in the tensor's own dtype (fp16) before the fp32 mul(255).round(); fp32 throughout here
    want = ((img / 2 + 0.5).clamp(0, 1)).permute(0, 2, 3, 1).float().mul(255).round().to(torch.uint8).cpu().numpy()[0]
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert got.shape == (64, 64, 3) and diff.max() <= 1 and (diff > 0).mean() < 0.02
    arr = proc.postprocess(img, "np")
    assert arr.shape == (1, 64, 64, 3) and abs(float(arr.mean()) - 0.5) < 0.05


def test_conv_pad_extra_matches_f_pad():
    """ABI 2: `pad_extra` = F.pad(x, (0, e, 0, e)) before the conv (diffusers Downsample2D in the VAE encoder)."""
    import torch.nn.functional as TF
    from sfast.hip import functional as F
    from sfast.hip.lib import last_kernel
    g = torch.Generator().manual_seed(4)
    for cin, cout, hw in ((64, 64, 16), (128, 128, 33), (8, 16, 10)):
        x = torch.randn(2, cin, hw, hw + 2, generator=g).to(DEV, torch.float16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(DEV, torch.float16).contiguous(memory_format=torch.channels_last)
        b = torch.randn(cout, generator=g).to(DEV, torch.float16)
        y = F.conv2d(x, w, b, stride=2, padding=0, pad_extra=1)
        want = TF.conv2d(TF.pad(x.float(), (0, 1, 0, 1)), w.float(), b.float(), stride=2)
        compare(f"conv pad_extra {cin}->{cout} @{hw}", y, want, 2e-2, 2e-2, kernel=last_kernel())


def test_sd_encoder_parity_and_compile_vae():
    """Full SD VAE encoder (34.2 M parameters), 256x256 image -> 32x32 moments; then both halves behind compile_vae()."""
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_vae
    from sfast.engine import VaeEncoderEngine
    enc = V.build_encoder("sd", seed=12, dtype=torch.float16, device=DEV)
    ref = V.build_encoder("sd", seed=12)
    ref.load_state_dict({k: v.float().cpu() for k, v in enc.state_dict().items()})
    x = (torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(6)) * 2 - 1).to(DEV, torch.float16)
    y = VaeEncoderEngine.from_module(enc).forward(x)
    with torch.no_grad():
        want = ref(x.float().cpu())
        eager16 = enc(x)
    e_eng, e_eager = rel_l2(y.float().cpu(), want), rel_l2(eager16.float().cpu(), want)
    log_value("sd vae encoder 256x256 image", engine_vs_fp32=e_eng, eager16_vs_fp32=e_eager)
    assert y.shape == (1, 8, 32, 32) and torch.isfinite(y).all()
    assert e_eng < 4e-3 and e_eng < 1.5 * e_eager + 1e-3, (e_eng, e_eager)
    # compile_vae patches encoder and decoder
    cfg = dict(block_out_channels=(64, 128), norm_num_groups=8, layers_per_block=1)
    vae = types.SimpleNamespace(encoder=V.build_encoder("tiny", seed=5, dtype=torch.float16, device=DEV, **cfg),
                                decoder=V.build("tiny", seed=6, dtype=torch.float16, device=DEV, **cfg), device=torch.device(DEV),
                                config=types.SimpleNamespace(norm_num_groups=8))
    eager_enc = V.build_encoder("tiny", seed=5, dtype=torch.float16, device=DEV, **cfg)
    c = CompilationConfig.Default()
    c.memory_format = None
    compile_vae(vae, c)
    assert hasattr(vae, "_sfast_vae_encoder_engine") and hasattr(vae, "_sfast_vae_engine")
    img = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(7)).to(DEV, torch.float16)
    with torch.no_grad():
        assert rel_l2(vae.encoder(img).float(), eager_enc(img).float()) < 1e-2
