"""VAE decoder path on CPU: the oracle pins and the engine's host logic against the C-ABI
emulator. No GPU, no kernel launches."""
import os

import pytest
import torch

from abi_emulator import EmuLib, EmuHost
from oracle import vae_ref as V
from parity import rel_l2
from sfast.engine import UnsupportedVae, VaeDecoderEngine

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TINY = dict(block_out_channels=(64, 128), norm_num_groups=8, layers_per_block=1)


def test_decoder_topology_known_answer_param_count():
    # AutoencoderKL (SD1.x/2.x/SDXL VAE): 83,653,863 parameters = encoder 34,163,592 + decoder 49,490,179 + 72 + 20
    assert V.param_count(V.build("sd")) == V.SD_VAE_DECODER_PARAMS == 49_490_179


def test_decoder_state_dict_keys_follow_diffusers_naming():
    keys = set(V.build("sd").state_dict().keys())
    for k in ("conv_in.weight", "mid_block.resnets.0.norm1.weight", "mid_block.attentions.0.group_norm.weight",
              "mid_block.attentions.0.to_q.bias", "mid_block.attentions.0.to_out.0.weight", "up_blocks.0.upsamplers.0.conv.weight",
              "up_blocks.2.resnets.0.conv_shortcut.weight", "up_blocks.3.resnets.2.conv2.bias", "conv_norm_out.bias", "conv_out.weight"):
        assert k in keys, k
    assert "up_blocks.3.upsamplers.0.conv.weight" not in keys and "up_blocks.0.resnets.0.conv_shortcut.weight" not in keys


def test_oracle_reproduces_golden_vae():
    gold = torch.load(os.path.join(GOLDEN, "vae_tiny.pt"))
    d = V.build("tiny", seed=gold["seed"], **gold["config"])
    d.load_state_dict({k: v.half().float() for k, v in d.state_dict().items()})
    with torch.no_grad():
        y = d(gold["z"].float())
    torch.testing.assert_close(y, gold["y"], rtol=1e-3, atol=1e-4)


def _pair(seed, **cfg):
    m16 = V.build("tiny", seed=seed, dtype=torch.float16, **cfg)
    m32 = V.build("tiny", seed=seed, **cfg)
    m32.load_state_dict({k: v.float() for k, v in m16.state_dict().items()})
    return m16, m32


def test_plan_executes_tiny_decoder(built_lib):
    m16, m32 = _pair(5, **TINY)
    emu = EmuLib()
    eng = VaeDecoderEngine.from_module(m16, _host=EmuHost(emu))
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(0)).half()
    y = eng.forward(z)
    with torch.no_grad():
        want = m32(z.float())
    assert y.shape == (2, 3, 16, 16) and rel_l2(y, want) < 3e-3
    # second signature: new plan, shared (live) parameters, non-square latent
    z2 = torch.randn(1, 4, 8, 16, generator=torch.Generator().manual_seed(1)).half()
    with torch.no_grad():
        want2 = m32(z2.float())
    assert rel_l2(eng.forward(z2), want2) < 3e-3 and len(eng._plans) == 2
    vplan = eng.get_plan(2, 8, 8)
    inv = vplan.summary()
    # conv_in + 2 mid resnets + 2x2 up resnets (2 convs each) + 1 upsampler conv + conv_out; one attention with 2 samples.
    # (round 4) GroupNorms behind a split-K conv run inside that conv's reduce launch and leave the op list: vplan.gn_in_reduce
    n_gn = inv.get("gn_silu", {"count": 0})["count"] + inv.get("gn", {"count": 0})["count"] + vplan.gn_in_reduce
    assert inv["conv3x3"]["count"] == 13 and n_gn == 13 + 1
    assert inv["softmax"]["count"] == 2 and inv["attn_vae"]["count"] == 4 and inv["conv1x1"]["count"] == 1
    assert {"softmax_rows", "strided_copy", "conv2d", "gemm"} <= set(emu.calls) and ({"group_norm", "fused_gn"} & set(emu.calls))


def test_live_parameters_are_read_at_every_run(built_lib):
    m16, m32 = _pair(6, **TINY)
    eng = VaeDecoderEngine.from_module(m16, _host=EmuHost())
    z = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(2)).half()
    y0 = eng.forward(z)
    with torch.no_grad():
        m16.conv_out.bias.add_(1.0)  # in-place update (the LoRA / fine-tune contract of the reference, README.md:228-265)
        m16.mid_block.attentions[0].to_v.bias.mul_(0.0)
    y1 = eng.forward(z)
    m32.load_state_dict({k: v.float() for k, v in m16.state_dict().items()})
    with torch.no_grad():
        want = m32(z.float())
    assert rel_l2(y1, want) < 3e-3 and rel_l2(y1, y0) > 1e-2


def test_unsupported_decoders_are_rejected(built_lib):
    m32 = V.build("tiny", **TINY)
    with pytest.raises(UnsupportedVae):
        VaeDecoderEngine.from_module(m32, _host=EmuHost())  # fp32 parameters
    bad = V.build("tiny", dtype=torch.float16, block_out_channels=(36, 72), norm_num_groups=4, layers_per_block=1)
    with pytest.raises(UnsupportedVae):
        VaeDecoderEngine.from_module(bad, _host=EmuHost())  # channel counts not multiples of 8


class _FakeProcessor:
    """Duck-typed stand-in for diffusers' VaeImageProcessor (not installable here)."""

    def __init__(self):
        import types
        self.config = types.SimpleNamespace(do_normalize=True)

    def postprocess(self, image, output_type="pil", do_denormalize=None):
        raise AssertionError("patched away")

    @staticmethod
    def pt_to_numpy(images):
        raise AssertionError("patched away")

    @staticmethod
    def pt_to_pil(images):
        raise AssertionError("patched away")


def test_image_processor_patch_cpu_tensors_keep_reference_semantics():
    from sfast.libs.diffusers.image_processor import patch_image_prcessor
    proc = patch_image_prcessor(_FakeProcessor())
    img = torch.rand(2, 3, 8, 6, generator=torch.Generator().manual_seed(0)) * 2 - 1
    want = (img / 2 + 0.5).clamp(0, 1)
    assert torch.equal(proc.postprocess(img, "pt"), want)
    assert proc.postprocess(img, "latent") is img
    arr = proc.postprocess(img, "np")
    assert arr.shape == (2, 8, 6, 3) and arr.dtype.name == "float32"
    torch.testing.assert_close(torch.from_numpy(arr), want.permute(0, 2, 3, 1))
    pil = proc.postprocess(img, "pil")
    assert len(pil) == 2 and pil[0].size == (6, 8) and pil[0].mode == "RGB"
    with pytest.raises(ValueError):
        proc.postprocess([1, 2, 3])
    # unsupported processors are left alone (reference :18-20)
    other = object()
    assert patch_image_prcessor(other) is other


def test_encoder_topology_and_plan(built_lib):
    from sfast.engine import VaeEncoderEngine
    assert V.param_count(V.build_encoder("sd")) == V.SD_VAE_ENCODER_PARAMS == 34_163_592
    cfg = dict(block_out_channels=(64, 128, 128), norm_num_groups=8, layers_per_block=1)
    m16 = V.build_encoder("tiny", seed=8, dtype=torch.float16, **cfg)
    m32 = V.build_encoder("tiny", seed=8, **cfg)
    m32.load_state_dict({k: v.float() for k, v in m16.state_dict().items()})
    emu = EmuLib()
    eng = VaeEncoderEngine.from_module(m16, _host=EmuHost(emu))
    x = torch.randn(2, 3, 16, 24, generator=torch.Generator().manual_seed(3)).half()
    y = eng.forward(x)
    with torch.no_grad():
        want = m32(x.float())
    # two stride-2 convs over F.pad(x, (0,1,0,1)) (conv `pad_extra`): 16x24 -> 4x6, 2*latent channels
    assert y.shape == (2, 8, 4, 6) and rel_l2(y, want) < 3e-3
    with pytest.raises(UnsupportedVae):
        eng.forward(torch.zeros(1, 3, 18, 16).half())  # not divisible by 4
