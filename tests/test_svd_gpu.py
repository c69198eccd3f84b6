"""SVD-XT spatio-temporal UNet on a real MI355X: native plan (HIP kernels through the C ABI) vs the fp32 oracle restatement."""
import pytest
import torch

from oracle import svd_ref as S
from parity import compare, log_value, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(cfg, B, Fr, hw, seed):
    from sfast.engine import SVDUNetEngine
    m = S.build(cfg, seed=seed, dtype=torch.float16, device=DEV)
    g = torch.Generator().manual_seed(seed + 1)
    sample = torch.randn(B, Fr, cfg["in_channels"], hw, hw, generator=g).to(DEV, torch.float16)
    ehs = torch.randn(B, 1, cfg["cross_attention_dim"], generator=g).to(DEV, torch.float16)
    tids = torch.tensor([[6.0, 127.0, 0.02]] * B, device=DEV)
    t = torch.tensor([500.0, 321.0][:B], device=DEV)
    eng = SVDUNetEngine.from_module(m)
    y = eng.forward(sample, t, ehs, tids)
    plan = eng.get_plan(B, Fr, hw, hw)
    # hipGraph replay of the plan == eager plan, bitwise
    gph, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(gph, stream=s):
            plan.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.current_stream().wait_stream(s)
    gph.replay()
    torch.cuda.synchronize()
    assert torch.equal(plan.static_out, y)
    with torch.no_grad():
        y16 = m(sample, t, ehs, tids.half()).sample
        y32 = m.float()(sample.float(), t, ehs.float(), tids).sample
    return y, y16, y32, plan


def test_tiny_svd_unet_parity_and_graph():
    y, y16, y32, plan = _run(S.tiny_svd_config(), 2, 5, 16, 51)
    e, e16 = rel_l2(y, y32), rel_l2(y16, y32)
    log_value("svd tiny B=2 F=5 vs fp32 oracle", engine_vs_fp32=e, eager16_vs_fp32=e16, launches=len(plan.ops))
    assert torch.isfinite(y).all() and e < 4e-3, (e, e16)


def test_svd_width_unet_parity_25_frames():
    """SVD-XT channel widths / head counts / 25 frames at a reduced latent (32x32) and one layer per block: every kernel shape
    class of the full model (temporal conv at 320..1280 channels, temporal attention S=25 D=64, 1024-wide context) on real sizes."""
    cfg = dict(S.SVD_CONFIG)
    cfg.update(layers_per_block=1, sample_size=32)
    y, y16, y32, plan = _run(cfg, 1, 25, 32, 52)
    e, e16 = rel_l2(y, y32), rel_l2(y16, y32)
    log_value("svd-xt widths B=1 F=25 32x32 vs fp32 oracle", engine_vs_fp32=e, eager16_vs_fp32=e16, launches=len(plan.ops),
              gn_fused=plan.gn_fused)
    assert torch.isfinite(y).all() and e < 4e-3 and e < 1.5 * e16 + 2e-4, (e, e16)


@pytest.mark.parametrize("M,C_,rows,mod", [(4096, 320, 64, 8), (1000, 64, 1, 7), (50, 1280, 25, 2)])
def test_mix_rows_kernel(M, C_, rows, mod):
    import ctypes as C
    from sfast.hip import lib as L
    lib = L.init_device()
    g = torch.Generator().manual_seed(60)
    x, yv = (torch.randn(M, C_, generator=g).to(DEV, torch.float16) for _ in range(2))
    vec = torch.randn(mod, C_, generator=g).to(DEV, torch.float16)
    mix = torch.tensor([0.3], device=DEV, dtype=torch.float16)
    out = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    p = L.MixParams(L.F16, M, C_, rows, mod, C_, 1.0, 0.0, 1)
    L.check(lib.sfast_hip_mix_rows(x.data_ptr(), yv.data_ptr(), None, mix.data_ptr(), out.data_ptr(), C.byref(p), st), "mix")
    a = 1 - torch.sigmoid(mix.float())
    compare(f"mix_rows blend {M}x{C_}", out, a * x.float() + (1 - a) * yv.float(), 2e-3, 2e-3, kernel="mix_rows")
    p2 = L.MixParams(L.F16, M, C_, rows, mod, C_, 1.0, 0.0, 0)
    L.check(lib.sfast_hip_mix_rows(x.data_ptr(), None, vec.data_ptr(), None, out.data_ptr(), C.byref(p2), st), "mix")
    idx = (torch.arange(M, device=DEV) // rows) % mod
    compare(f"mix_rows rowvec {M}x{C_}", out, x.float() + vec.float()[idx], 2e-3, 2e-3, kernel="mix_rows")


def test_compile_unet_recognises_the_spatio_temporal_unet():
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    from sfast.engine import SVDUNetEngine
    cfg = S.tiny_svd_config()
    m = S.build(cfg, seed=53, dtype=torch.float16, device=DEV)
    ref = S.build(cfg, seed=53, dtype=torch.float16, device=DEV)
    config = CompilationConfig.Default()
    config.enable_cuda_graph = True
    m = compile_unet(m, config)
    assert isinstance(m._sfast_engine, SVDUNetEngine)
    g = torch.Generator().manual_seed(54)
    sample = torch.randn(1, 5, 8, 16, 16, generator=g).to(DEV, torch.float16)
    ehs = torch.randn(1, 1, cfg["cross_attention_dim"], generator=g).to(DEV, torch.float16)
    tids = torch.tensor([[6.0, 127.0, 0.02]], device=DEV, dtype=torch.float16)
    out = m(sample, torch.tensor(400.0, device=DEV), ehs, tids, return_dict=False)[0]
    out2 = m(sample, torch.tensor(400.0, device=DEV), ehs, tids).sample  # graph replay
    want = SVDUNetEngine.from_module(ref).forward(sample, 400.0, ehs, tids)
    assert torch.equal(out, want) and torch.equal(out2, want) and len(m.forward._cached) == 1


def test_tiny_svd_unet_vs_golden_fixture():
    """tests/golden/svd_tiny.pt: seeded inputs + the fp32 oracle's output, committed -- the HIP plan against it without running the oracle."""
    import os
    from sfast.engine import SVDUNetEngine
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "svd_tiny.pt"))
    m = S.build(gold["config"], seed=gold["seed"])
    m.load_state_dict({k: v.half().float() for k, v in m.state_dict().items()})
    m = m.half().to(DEV)
    eng = SVDUNetEngine.from_module(m)
    y = eng.forward(gold["sample"].to(DEV), gold["timesteps"].to(DEV), gold["encoder_hidden_states"].to(DEV), gold["added_time_ids"].to(DEV))
    e = rel_l2(y, gold["y"].to(DEV))
    log_value("svd tiny vs golden fixture", engine_vs_golden=e)
    assert torch.isfinite(y).all() and e < 4e-3, e
