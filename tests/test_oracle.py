"""Pins the oracle (oracle/) -- runs on CPU.

The reference's own tests are differential against eager PyTorch ops and keep no stored vectors
(SURVEY.md section 8c), so the pins are: (1) the public parameter counts of the two UNets whose
topology is restated, (2) agreement of every per-op oracle with the ATen op the reference's tests
use as ground truth, (3) stability against the committed golden vectors.
"""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import ops_ref as R
from oracle import unet_ref as U

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_topology_known_answer_param_counts():
    # publicly known UNet sizes
    with torch.device("meta"):
        sd15 = U.UNet2DConditionModel(**U.SD15_CONFIG)
        sdxl = U.UNet2DConditionModel(**U.SDXL_CONFIG)
    assert U.param_count(sd15) == 859_520_964
    assert U.param_count(sdxl) == 2_567_463_684
    # SD2.1 (stabilityai/stable-diffusion-2-1): the SD1.5 topology with linear proj_in/out, 64-wide heads (5/10/20/20 per
    # level) and OpenCLIP's 1024-wide text states -- 865,910,724 parameters
    sd21_cfg = dict(U.SD15_CONFIG, attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True)
    with torch.device("meta"):
        sd21 = U.UNet2DConditionModel(**sd21_cfg)
    assert U.param_count(sd21) == 865_910_724


def test_state_dict_keys_follow_diffusers_naming():
    with torch.device("meta"):
        m = U.UNet2DConditionModel(**U.SD15_CONFIG)
    keys = set(m.state_dict().keys())
    for k in ("conv_in.weight", "time_embedding.linear_1.weight", "down_blocks.0.resnets.0.time_emb_proj.bias",
              "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight",
              "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_out.0.bias",
              "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
              "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.2.bias",
              "down_blocks.0.downsamplers.0.conv.weight", "mid_block.attentions.0.proj_in.weight",
              "up_blocks.1.upsamplers.0.conv.bias", "up_blocks.3.resnets.2.conv_shortcut.weight",
              "conv_norm_out.weight", "conv_out.bias"):
        assert k in keys, k
    assert "down_blocks.3.attentions.0.norm.weight" not in keys  # DownBlock2D has no attention
    assert m.state_dict()["up_blocks.0.resnets.0.conv1.weight"].shape == (1280, 2560, 3, 3)
    assert m.state_dict()["up_blocks.3.resnets.2.conv1.weight"].shape == (320, 640, 3, 3)
    assert m.state_dict()["down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k.weight"].shape == (640, 768)


def test_group_norm_oracle_matches_independent_restatement():
    # reference self-test: randn(2,320,32,32), G=32, eps 1e-5 (triton/ops/group_norm.py:485-523)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 320, 16, 16, generator=g) * 3 + 1
    w, b = torch.randn(320, generator=g), torch.randn(320, generator=g)
    a = R.group_norm_ref(x, 32, w, b, 1e-5)
    m = R.group_norm_manual(x, 32, w, b, 1e-5)
    torch.testing.assert_close(a, m, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(R.group_norm_ref(x, 32, w, b, 1e-5, silu=True), F.silu(m), rtol=1e-4, atol=1e-4)


def test_geglu_oracle_is_the_reference_test_model():
    # tests/operators/test_cutlass_dual_linear.py:37-40: proj -> chunk(2, -1) -> hidden * gelu(gate)
    g = torch.Generator().manual_seed(1)
    x, w, b = torch.randn(16, 8, generator=g), torch.randn(32, 8, generator=g), torch.randn(32, generator=g)
    h, gate = F.linear(x, w, b).chunk(2, dim=-1)
    torch.testing.assert_close(R.linear_ref(x, w, b, geglu=True), h * F.gelu(gate))


def test_conv_oracle_is_the_reference_test_model():
    # tests/operators/test_cudnn_convolution.py:14-27,50-69: act(conv(x) + alpha*y) with broadcast y
    conv = torch.nn.Conv2d(2, 2, 3)
    x = torch.ones(1, 2, 32, 32)
    y = torch.ones(1, 1, 30, 30)
    with torch.no_grad():
        want = conv(x) + 0.5 * y
        got = R.conv2d_ref(x, conv.weight, conv.bias, y, 0.5)
        torch.testing.assert_close(got, want)
        torch.testing.assert_close(R.conv2d_ref(x, conv.weight, conv.bias, y, 0.5, act="tanh"), torch.tanh(want))


def test_attention_oracle_matches_sdpa():
    g = torch.Generator().manual_seed(2)
    q, k, v = (torch.randn(2, 33, 4, 40, generator=g) for _ in range(3))
    want = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2)
    torch.testing.assert_close(R.attention_ref(q, k, v), want, rtol=1e-4, atol=1e-5)


def test_ddim_schedule_constants():
    ts, coefs = R.ddim_schedule(50)
    assert ts[0] == 981 and ts[1] == 961 and ts[-1] == 1 and len(ts) == 50
    for sa, s1a, sp, s1p in coefs:
        assert abs(sa * sa + s1a * s1a - 1) < 1e-6 and abs(sp * sp + s1p * s1p - 1) < 1e-6
        assert sp >= sa  # alpha_bar grows towards t = 0


def test_oracle_reproduces_golden_ops():
    gold = torch.load(os.path.join(GOLDEN, "ops.pt"))
    c = gold["group_norm_silu"]
    torch.testing.assert_close(R.group_norm_ref(c["x"], c["groups"], c["weight"], c["bias"], c["eps"], True), c["y"], rtol=1e-4, atol=1e-4)
    c = gold["layer_norm"]
    torch.testing.assert_close(R.layer_norm_ref(c["x"], (320,), c["weight"], c["bias"], c["eps"]), c["y"], rtol=1e-4, atol=1e-4)
    c = gold["geglu"]
    torch.testing.assert_close(R.linear_ref(c["x"], c["weight"], c["bias"], geglu=True), c["y"], rtol=1e-4, atol=1e-4)
    c = gold["conv3x3_bias_add"]
    torch.testing.assert_close(R.conv2d_ref(c["x"], c["weight"], c["bias"], c["z"], c["alpha"], 1, 1), c["y"], rtol=1e-4, atol=1e-4)
    c = gold["attention_d40_kv77"]
    torch.testing.assert_close(R.attention_ref(c["q"], c["k"], c["v"]), c["y"], rtol=1e-4, atol=1e-4)


def test_oracle_reproduces_golden_unet():
    gold = torch.load(os.path.join(GOLDEN, "unet_tiny.pt"))
    m = U.build(gold["config"], seed=gold["seed"])
    m.load_state_dict({k: v.half().float() for k, v in m.state_dict().items()})
    with torch.no_grad():
        y = m(gold["sample"].float(), gold["timestep"], gold["encoder_hidden_states"].float()).sample
        yb = m(gold["sample"].float(), torch.tensor(gold["timesteps_b"]), gold["encoder_hidden_states"].float()).sample
    torch.testing.assert_close(y, gold["y"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(yb, gold["y_b"], rtol=1e-3, atol=1e-4)


def test_controlnet_and_vae_topology_known_answers():
    from oracle import controlnet_ref as CN
    assert CN.param_count(CN.build("sd15")) == CN.SD15_CONTROLNET_PARAMS == 361_279_120
    keys = set(CN.build("tiny").state_dict().keys())
    for k in ("controlnet_cond_embedding.conv_in.weight", "controlnet_cond_embedding.blocks.3.bias", "controlnet_cond_embedding.conv_out.weight",
              "controlnet_down_blocks.0.weight", "controlnet_down_blocks.5.bias", "controlnet_mid_block.weight", "mid_block.resnets.1.conv2.weight"):
        assert k in keys, k
    assert not any(k.startswith("up_blocks") or k.startswith("conv_out") for k in keys)


def test_closed_form_known_answers():
    """Pins that do not lean on another implementation: inputs whose result is known in closed form."""
    import math
    from oracle.unet_ref import timestep_embedding
    # attention: identical keys -> uniform softmax -> the mean of V; one dominant key -> that V row
    g = torch.Generator().manual_seed(7)
    q, v = torch.randn(1, 5, 2, 8, generator=g), torch.randn(1, 6, 2, 8, generator=g)
    k = torch.randn(1, 1, 2, 8, generator=g).expand(1, 6, 2, 8)
    torch.testing.assert_close(R.attention_ref(q, k, v), v.mean(1, keepdim=True).expand(1, 5, 2, 8), rtol=1e-5, atol=1e-6)
    k2 = torch.zeros(1, 6, 2, 8)
    k2[:, 3] = 100.0 * q[:, 0].sign()  # score of key 3 for query 0 = 100 * |q|_1 * scale >> the others (0)
    torch.testing.assert_close(R.attention_ref(q, k2, v)[:, 0], v[:, 3], rtol=1e-4, atol=1e-5)
    # GroupNorm: a group holding the values {a, b} equally often normalises them to -/+ 1 (eps -> 0); affine applies per channel
    x = torch.tensor([1.0, 3.0]).repeat(4)[None, :, None].repeat(1, 1, 5)  # [1, 8, 5], groups of 4 channels
    y = R.group_norm_ref(x, 2, weight=torch.full((8,), 2.0), bias=torch.full((8,), 0.5), eps=0.0)
    torch.testing.assert_close(y, torch.tensor([-1.5, 2.5]).repeat(4)[None, :, None].repeat(1, 1, 5))
    assert float(R.group_norm_ref(x, 2, eps=1e-5, silu=True)[0, 1, 0]) == pytest.approx(1.0 / (1.0 + math.exp(-1.0)), rel=1e-4)
    # LayerNorm of an arithmetic progression 0..n-1: mean (n-1)/2, biased variance (n^2-1)/12
    n = 16
    ln = R.layer_norm_ref(torch.arange(n, dtype=torch.float32)[None], (n,), eps=0.0)
    torch.testing.assert_close(ln[0], (torch.arange(n) - (n - 1) / 2) / math.sqrt((n * n - 1) / 12.0))
    # GEGLU: value * gelu(gate); gelu(0) = 0 kills the output, gelu(g) = g for large g
    w = torch.zeros(4, 2)
    w[0, 0], w[1, 1] = 1.0, 1.0          # value half: identity on the two inputs
    w[2, 0], w[3, 1] = 0.0, 50.0         # gate half: 0 for column 0, large for column 1
    out = R.linear_ref(torch.tensor([[3.0, 2.0]]), w, geglu=True)
    torch.testing.assert_close(out, torch.tensor([[0.0, 2.0 * 100.0]]))
    # sinusoidal embedding at t = 0 is cos = 1 / sin = 0; at frequency 1 (index 0) it is cos t / sin t
    e = timestep_embedding(torch.tensor([0.0, 2.0]), 8, flip_sin_to_cos=True)
    torch.testing.assert_close(e[0], torch.tensor([1.0, 1, 1, 1, 0, 0, 0, 0]))
    assert float(e[1, 0]) == pytest.approx(math.cos(2.0), abs=1e-6) and float(e[1, 4]) == pytest.approx(math.sin(2.0), abs=1e-6)
    # DDIM with eta = 0: with identical t and t_prev constants the latents are unchanged, whatever the noise prediction
    lat, eps_uc = torch.randn(1, 4, 3, 3, generator=g), torch.randn(2, 4, 3, 3, generator=g)
    sa, s1a = math.sqrt(0.7), math.sqrt(0.3)
    torch.testing.assert_close(R.cfg_ddim_ref(eps_uc, lat, (sa, s1a, sa, s1a), 7.5), lat, rtol=1e-5, atol=1e-6)
    # ... and guidance 1 ignores the unconditional half: x_prev = sp * (x - s1a e_c) / sa + s1p e_c
    sp, s1p = math.sqrt(0.9), math.sqrt(0.1)
    want = sp * (lat - s1a * eps_uc[1:]) / sa + s1p * eps_uc[1:]
    torch.testing.assert_close(R.cfg_ddim_ref(eps_uc, lat, (sa, s1a, sp, s1p), 1.0), want, rtol=1e-5, atol=1e-6)
    # conv: a 3x3 kernel that is 1 at the centre tap is the identity; stride 2 picks the even pixels
    img = torch.randn(1, 2, 6, 6, generator=g)
    wc = torch.zeros(2, 2, 3, 3)
    wc[0, 0, 1, 1], wc[1, 1, 1, 1] = 1.0, 1.0
    torch.testing.assert_close(R.conv2d_ref(img, wc, padding=1), img)
    torch.testing.assert_close(R.conv2d_ref(img, wc, padding=1, stride=2), img[:, :, ::2, ::2])


def test_oracle_reproduces_golden_round2():
    """Round-2 fixtures (tests/golden/make_golden_r2.py): attention bias / key-padding mask, the scheduler rows, the tiny SVD UNet."""
    from oracle import svd_ref as S
    gold = torch.load(os.path.join(GOLDEN, "ops_r2.pt"))
    c = gold["attention_full_bias"]
    torch.testing.assert_close(R.attention_ref(c["q"].float(), c["k"].float(), c["v"].float(), attn_bias=c["bias"].float()), c["y"], rtol=1e-4, atol=1e-4)
    c = gold["attention_key_padding_mask"]
    torch.testing.assert_close(R.attention_ref(c["q"].float(), c["k"].float(), c["v"].float(), attn_bias=c["bias"].float()), c["y"], rtol=1e-4, atol=1e-4)
    for b, valid in enumerate(c["valid"]):  # independent of the bias code path: masked keys == keys that are not there
        want = R.attention_ref(c["q"][b:b + 1].float(), c["k"][b:b + 1, :valid].float(), c["v"][b:b + 1, :valid].float())
        torch.testing.assert_close(c["y"][b:b + 1], want, rtol=1e-4, atol=1e-4)
    for name in ("ddim_step_eps", "euler_step_eps"):
        c = gold[name]
        a, bb = c["coef"]
        torch.testing.assert_close(a * c["sample"].float() + bb * c["model_output"].float(), c["y"], rtol=1e-5, atol=1e-5)
    c = gold["euler_step_eps"]
    torch.testing.assert_close(c["scale"] * c["sample"].float(), c["y_scaled"], rtol=1e-5, atol=1e-6)
    gold = torch.load(os.path.join(GOLDEN, "svd_tiny.pt"))
    m = S.build(gold["config"], seed=gold["seed"])
    m.load_state_dict({k: v.half().float() for k, v in m.state_dict().items()})
    with torch.no_grad():
        y = m(gold["sample"].float(), gold["timesteps"], gold["encoder_hidden_states"].float(), gold["added_time_ids"]).sample
    torch.testing.assert_close(y, gold["y"], rtol=1e-3, atol=1e-4)


def test_oracle_reproduces_golden_round3():
    """Round-3 fixtures (tests/golden/make_golden_r3.py): the tiny UNet with time_cond_proj / class embeddings / the stacked embedding
    chain (what round 3 added to oracle/unet_ref.py), and the rows the device-side schedule cursor hands out."""
    import importlib.util
    from oracle import unet_ref as U
    spec = importlib.util.spec_from_file_location("make_golden_r3", os.path.join(GOLDEN, "make_golden_r3.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gold = torch.load(os.path.join(GOLDEN, "unet_tiny_r3.pt"))
    assert set(gold) == set(gen.CASES)
    for name, c in gold.items():
        cfg = U.tiny_config(**c["over"])
        m = U.build(cfg, seed=c["seed"])
        m.load_state_dict({k: v.half().float() for k, v in m.state_dict().items()})
        s, e, kw = gen.inputs(name, cfg, c["seed"] + 1000)
        with torch.no_grad():
            y = m(s, c["timestep"], e, **kw).sample
        torch.testing.assert_close(y, c["y"], rtol=1e-3, atol=1e-4, msg=lambda t, name=name: f"{name}: {t}")
        if "class_labels" in kw:   # the condition really reaches the output
            kw2 = dict(kw, class_labels=kw["class_labels"].flip(0))
            with torch.no_grad():
                assert float((m(s, c["timestep"], e, **kw2).sample - c["y"]).abs().max()) > 1e-3, name
    ops = torch.load(os.path.join(GOLDEN, "ops_r3.pt"))["schedule_advance"]
    ts, coef = R.ddim_schedule(ops["n_steps"])
    for j, row in enumerate(ops["rows"]):
        assert row == (ops["start"] + j) % ops["n_steps"]
        assert float(ops["ts_out"][j, 0]) == float(ts[row])
        torch.testing.assert_close(ops["coef_out"][j], torch.tensor(coef[row], dtype=torch.float32))
    assert ops["cursor_after"] == (ops["start"] + len(ops["rows"])) % ops["n_steps"]


def test_oracle_vs_reference_triton_fixture():
    """tests/golden/ref_triton_small.pt holds OUTPUTS OF THE REFERENCE'S OWN TRITON KERNELS (group_norm.py, layer_norm.py, copy.py and,
    if it compiled, conv.py of /root/reference/src/sfast/triton/ops), run on an MI355X through oracle/ref_triton_run.py
    (tests/golden/make_golden_ref_triton.py). Unlike the other fixtures it is not minted from this oracle, so this pins
    `oracle.ops_ref` for SURVEY section-8 rows a6-a9 (a15) against the reference itself. Bars as in tests/test_ref_triton_gpu.py:
    one output ulp, plus -- GroupNorm only -- the reference's f16-rounded statistics (its mean / rstd tensors take the input dtype)."""
    import torch
    from oracle import ops_ref as R
    from oracle import ref_cases as RC
    path = os.path.join(GOLDEN, "ref_triton_small.pt")
    assert os.path.exists(path), "the committed fixture of the reference's own outputs is missing (tests/golden/make_golden_ref_triton.py)"
    fx = torch.load(path, weights_only=False)
    assert fx["status"]["gn"].startswith("ok") and fx["status"]["ln"].startswith("ok") and fx["status"]["copy"].startswith("ok"), fx["status"]
    u = 2.0 ** -10
    n = 0
    for c in RC.GN_CASES:
        if not c.get("small"):
            continue
        x, w, b = RC.gn_inputs(c)
        want = R.group_norm_ref(x, c["groups"], w, b, c["eps"], c["silu"])
        got = fx["out"][c["name"]]["y"].float()
        N, C = x.shape[:2]
        xs = x.float().reshape(N, c["groups"], -1)
        mean = xs.mean(2, keepdim=True)
        rstd = (xs.var(2, unbiased=False, keepdim=True) + c["eps"]).rsqrt()
        stat = ((u / 2) * ((xs - mean).abs() * rstd + mean.abs() * rstd)).reshape(x.shape) * w.float().abs().reshape(1, C, 1, 1)
        lim = u * (1.0 + want.abs()) + 1.65 * stat
        assert int(((got - want).abs() > lim).sum()) == 0, (c["name"], float((got - want).abs().max()))
        assert torch.allclose(fx["out"][c["name"]]["mean"].float(), mean.reshape(N, -1), atol=u, rtol=u)
        assert torch.allclose(fx["out"][c["name"]]["rstd"].float(), rstd.reshape(N, -1), atol=u, rtol=2 * u)
        n += 1
    for c in RC.LN_CASES:
        if not c.get("small"):
            continue
        x, w, b = RC.ln_inputs(c)
        want = R.layer_norm_ref(x, (x.shape[-1],), w, b, c["eps"])
        got = fx["out"][c["name"]]["y"].float()
        assert torch.allclose(got, want, atol=u, rtol=u), (c["name"], float((got - want).abs().max()))
        n += 1
    for c in RC.COPY_CASES:
        if not c.get("small"):
            continue
        x = RC.copy_inputs(c)
        src, fmt = RC.copy_view(c, x)
        assert fx["out"][c["name"]]["equal_to_torch_copy"]
        assert torch.equal(fx["out"][c["name"]]["y"], src.contiguous(memory_format=fmt))
        n += 1
    if fx["status"].get("conv", "").startswith("ok"):
        for c in RC.CONV_CASES:
            if not c.get("small"):
                continue
            x, w, b = RC.conv_inputs(c)
            want = R.conv2d_ref(x, w, b, stride=c["stride"], padding=c["padding"])
            got = fx["out"][c["name"]]["y"].float()
            K = c["w"][1] * c["w"][2] * c["w"][3]
            f = max(2.0, 0.75 * (K / 32.0) ** 0.5)   # the reference's Triton conv accumulates f16 inputs in f16 (conv.py:844-845)
            assert torch.allclose(got, want, atol=f * u, rtol=f * u), (c["name"], float((got - want).abs().max()))
            n += 1
    assert n >= 7
