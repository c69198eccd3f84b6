"""Parity evidence the round-2 review asked for (VERDICT r02 "What's weak" #1, "Next round" #1):

  (a) StableVideoDiffusion-XT at FULL size (BASELINE.json configs[4]: 576x1024 -> 72x128 latent, 25 frames, `SVD_CONFIG` unmodified,
      1.52 B parameters) against the fp32 oracle -- same assertion form as tests/test_sdxl_gpu.py;
  (b) a "trained-statistics" whole-UNet case: every default-init whole-model test has near-uniform softmax rows (logit sigma ~ 1) and
      near-Gaussian GroupNorm inputs. Here `to_q` / `to_k` are rescaled so the attention logits have sigma ~ 5 (peaked rows: the
      online-softmax rescale path, f16 probabilities spanning their whole range) and a few output channels of the convs that feed
      GroupNorms are multiplied by 30 (outlier channels dominating their group's statistics), on the full SD1.5 UNet and the tiny one;
  (c) bounds relative to the f16-STORAGE floor measured in the test itself (tools/error_budget.storage_hooks): `e <= 1.1 * floor + 1e-4`
      where the floor is exact (SD1.5), the measured analogue elsewhere.

Every value is logged to gpurun_out/parity.jsonl (copied to profiles/r03_parity_*.jsonl).
"""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import svd_ref as S
from oracle import unet_ref as U
from parity import log_value, rel_l2, storage_floor

pytestmark = pytest.mark.gpu
DEV = "cuda"


# ---- (b) trained statistics ---------------------------------------------------------------------------------------------------------
def make_trained_like(m, logit_sigma=5.0, outlier=30.0, n_out=3, seed=0):
    """In place, on an oracle UNet with default-init weights (q, k ~ N(0, 1) per element -> logits q.k / sqrt(D) ~ N(0, 1)):
      * to_q and to_k of EVERY attention scaled by sqrt(logit_sigma) each -> logit sigma ~ logit_sigma (row max ~ 3-4 sigma above the
        mean over 4096 keys: a handful of keys carry each row, as in a trained model);
      * `n_out` output channels of conv_in and of every resnet conv1 / conv2 (the tensors GroupNorms read, directly or through the
        residual stream) multiplied by `outlier`."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, mod in m.named_modules():
            if isinstance(mod, U.Attention):
                mod.to_q.weight.mul_(math.sqrt(logit_sigma))
                mod.to_k.weight.mul_(math.sqrt(logit_sigma))
            elif isinstance(mod, nn.Conv2d) and (name == "conv_in" or name.endswith((".conv1", ".conv2"))):
                idx = torch.randperm(mod.weight.shape[0], generator=g)[:n_out].to(mod.weight.device)
                mod.weight[idx] *= outlier
                mod.bias[idx] *= outlier
    return m


def _attention_logit_sigma(m32, sample, ehs, t):
    """Measured on the module: std of the self-attention logits of the first 64x64-level block (what the rescale is meant to produce)."""
    seen = {}
    attn = m32.down_blocks[0].attentions[0].transformer_blocks[0].attn1

    def hook(mod, inp):
        x = inp[0]
        B, Sq, C = x.shape
        H = mod.heads
        q = mod.to_q(x).view(B, Sq, H, C // H).transpose(1, 2)
        k = mod.to_k(x).view(B, Sq, H, C // H).transpose(1, 2)
        s = (q[:, :, :512] @ k.transpose(-1, -2)) * (C // H) ** -0.5
        seen["sigma"] = float(s.std())
        seen["rowmax_minus_mean"] = float((s.amax(-1) - s.mean(-1)).mean())
        p = torch.softmax(s, -1)
        seen["top1_mass"] = float(p.amax(-1).mean())

    h = attn.register_forward_pre_hook(hook)
    try:
        with torch.no_grad():
            m32(sample.float(), t, ehs.float())
    finally:
        h.remove()
    return seen


def _trained_case(cfg_name, cfg, B, seed, S_ctx=77):
    from sfast.engine import UNet2DEngine
    m = U.build(cfg_name if isinstance(cfg_name, str) and cfg_name != "tiny" else cfg, seed=seed, dtype=torch.float32, device=DEV)
    make_trained_like(m, seed=seed)
    m = m.half()  # the engine and the oracle share f16-representable weights
    g = torch.Generator().manual_seed(seed + 1)
    hw = cfg["sample_size"]
    sample = torch.randn(B, cfg["in_channels"], hw, hw, generator=g).to(DEV, torch.float16)
    ehs = torch.randn(B, S_ctx, cfg["cross_attention_dim"], generator=g).to(DEV, torch.float16)
    eng = UNet2DEngine.from_module(m)
    y = eng.forward(sample, 981, ehs)
    assert torch.isfinite(y).all()
    with torch.no_grad():
        y16 = m(sample, 981, ehs).sample
    del eng
    ref = m.float()
    with torch.no_grad():
        y32 = ref(sample.float(), 981, ehs.float()).sample
    stats = _attention_logit_sigma(ref, sample, ehs, 981)
    floor = storage_floor(ref, lambda: ref(sample.float(), 981, ehs.float()).sample, y32)
    return dict(engine_vs_fp32=rel_l2(y, y32), eager16_vs_fp32=rel_l2(y16, y32), engine_vs_eager16=rel_l2(y, y16), f16_storage_floor=floor,
                max_abs_engine=float((y.float() - y32).abs().max()), ref_absmax=float(y32.abs().max()), **stats)


def test_sd15_trained_statistics_parity():
    r = _trained_case("sd15", U.SD15_CONFIG, 2, seed=21)
    log_value("sd15 B=2 trained-statistics parity (logit sigma ~5, x30 outlier channels)", **r)
    assert r["sigma"] > 3.0 and r["top1_mass"] > 0.05, r        # the rows really are peaked (default init: sigma ~ 1, top-1 mass ~ 1e-3)
    assert r["engine_vs_fp32"] <= 1.1 * r["f16_storage_floor"] + 1e-4, r
    assert r["engine_vs_fp32"] < r["eager16_vs_fp32"], r


def test_tiny_trained_statistics_parity():
    r = _trained_case("tiny", U.tiny_config(), 2, seed=22)
    log_value("tiny B=2 trained-statistics parity (logit sigma ~5, x30 outlier channels)", **r)
    assert r["sigma"] > 3.0, r
    assert r["engine_vs_fp32"] <= 1.1 * r["f16_storage_floor"] + 1e-4, r


def test_peaked_attention_rows_force_the_rescale_path():
    """Operator level (guide rule: a rare data-dependent branch needs an input that FORCES it): one key per query block spiked far above
    the rest at a LATE tile, so the running max jumps after O and l have accumulated -- self-attention shapes of SD1.5 / SDXL."""
    from sfast.hip import functional as Fn
    for (S_, H, D, dt) in ((4096, 8, 40, torch.float16), (1024, 8, 80, torch.float16), (4096, 10, 64, torch.float16), (1024, 5, 64, torch.bfloat16)):
        g = torch.Generator().manual_seed(S_ + D)
        q = torch.randn(2, S_, H, D, generator=g).to(DEV, dt)
        k = torch.randn(2, S_, H, D, generator=g).to(DEV, dt)
        v = torch.randn(2, S_, H, D, generator=g).to(DEV, dt)
        for late in (S_ - 5, S_ // 2 + 17, 70):
            k[:, late] = (q[:, 100] * 3.0).to(dt)   # key `late` aligned with query 100 (and strongly correlated with nothing else)
        k[:, S_ - 9] = (q[:, S_ - 1] * 6.0).to(dt)
        want = F.scaled_dot_product_attention(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)).transpose(1, 2)
        for variant in (32, 64):   # the 32-row kernel (exact lazy rescale) and the 64-row kernel (reference maximum, 64-fold threshold)
            y = Fn.attention(q, k, v, variant=variant)
            err = rel_l2(y, want)
            worst = float((y.float() - want).abs().max())
            log_value(f"attention spiked keys S={S_} H={H} D={D} {dt} variant={variant}", rel_l2=err, max_abs=worst)
            tol = 2e-3 if dt == torch.float16 else 1.5e-2
            assert torch.isfinite(y).all() and err < tol and worst < 20 * tol, (S_, D, variant, err, worst)


# ---- (a) SVD-XT, full size ------------------------------------------------------------------------------------------------------------
def _bounded_sdpa(real, limit_bytes=6 << 30):
    """fp32 attention of the ORACLE at S = 9216: ROCm's fused SDPA kernels take 16-bit inputs only, the math fallback materialises
    [batch, heads, S, S] fp32 scores (42 GB for 25 frames). Same arithmetic in batch slices so that the test's memory stays bounded
    (an out-of-memory box is a strike): test plumbing, the product path has no such thing."""
    def sdpa(q, k, v, *a, **kw):
        per = q.shape[1] * q.shape[-2] * k.shape[-2] * 4 * 3
        if q.dtype != torch.float32 or per * q.shape[0] <= limit_bytes:
            return real(q, k, v, *a, **kw)
        step = max(1, limit_bytes // per)
        return torch.cat([real(q[i:i + step], k[i:i + step], v[i:i + step], *a, **kw) for i in range(0, q.shape[0], step)])
    return sdpa


def test_svd_xt_full_size_parity():
    """BASELINE.json configs[4] / the reference's examples/optimize_stable_video_diffusion_pipeline.py workload: SVD-XT UNet
    (`SVD_CONFIG` unmodified: 2 layers per block, widths 320..1280, 1,524,623,082 parameters), 25 frames, 72x128 latent, B = 1."""
    import time
    from sfast.engine import SVDUNetEngine
    marks = [("start", time.time())]

    def mark(name):
        torch.cuda.synchronize()
        marks.append((name, time.time()))
    cfg = S.SVD_CONFIG
    m = S.build(cfg, seed=61, dtype=torch.float16, device=DEV)
    mark("build")
    assert sum(p.numel() for p in m.parameters()) == 1_524_623_082
    B, Fr, H, W = 1, 25, 72, 128
    g = torch.Generator().manual_seed(62)
    sample = torch.randn(B, Fr, cfg["in_channels"], H, W, generator=g).to(DEV, torch.float16)
    ehs = torch.randn(B, 1, cfg["cross_attention_dim"], generator=g).to(DEV, torch.float16)
    tids = torch.tensor([[6.0, 127.0, 0.02]] * B, device=DEV)
    t = torch.tensor([500.0], device=DEV)
    eng = SVDUNetEngine.from_module(m)
    mark("engine")
    y = eng.forward(sample, t, ehs, tids)
    mark("plan+forward")
    assert y.shape == (B, Fr, cfg["out_channels"], H, W) and torch.isfinite(y).all()
    plan = eng.get_plan(B, Fr, H, W)
    launches = len(plan.ops)
    del eng, plan
    # (the eager legs run without MIOpen: tests/conftest.py eager_references_without_miopen -- first-use solver compilation was
    #  245 s + 249 s of this test's 534 s)
    with torch.no_grad():
        y16 = m(sample, t, ehs, tids.half()).sample
    mark("eager16")
    ref = m.float()  # in place: f16 weights are exactly representable
    real = F.scaled_dot_product_attention
    F.scaled_dot_product_attention = _bounded_sdpa(real)
    try:
        def fwd():
            with torch.no_grad():
                return ref(sample.float(), t, ehs.float(), tids).sample
        y32 = fwd()
        mark("fp32 oracle")
        floor = storage_floor(ref, fwd, y32, extra_leaf=(nn.Conv3d,),
                              extra_comp=(S.TemporalResnetBlock, S.SpatioTemporalResBlock, S.TemporalBasicTransformerBlock, S.AlphaBlender,
                                          S.TransformerSpatioTemporalModel))
    finally:
        F.scaled_dot_product_attention = real
    mark("floor")
    e, e16 = rel_l2(y, y32), rel_l2(y16, y32)
    log_value("svd-xt FULL size test phases (seconds)", **{b[0]: round(b[1] - a[1], 1) for a, b in zip(marks, marks[1:])})
    log_value("svd-xt FULL size B=1 F=25 72x128 vs fp32 oracle", engine_vs_fp32=e, eager16_vs_fp32=e16, engine_vs_eager16=rel_l2(y, y16),
              f16_storage_floor=floor, launches=launches, max_abs_engine=float((y.float() - y32).abs().max()), ref_absmax=float(y32.abs().max()),
              peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30)
    assert e < 2.5e-3, (e, e16, floor)
    assert e < e16 + 2e-4, (e, e16)                      # not worse than the eager-fp16 stand-in
    assert e <= 1.25 * floor + 2e-4, (e, floor)          # at the f16-storage floor of this network
