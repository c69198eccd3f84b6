"""SVD-XT spatio-temporal UNet (BASELINE.json configs[4]) on CPU: the oracle restatement against the
published parameter count, the engine's parameter inventory against the oracle, and the planner (temporal GroupNorm / conv as
reshaped 2-D ops, strided temporal attention, single-key cross-attention as a row bias, AlphaBlender) executed on the ABI emulator."""
import torch

from abi_emulator import EmuLib, EmuHost
from oracle import svd_ref as S
from sfast.engine import SVDUNetEngine
from sfast.engine.unet_spec import SVD_CONFIG, svd_param_shapes


def test_svd_oracle_parameter_count_known_answer():
    with torch.device("meta"):
        m = S.UNetSpatioTemporalConditionModel(**S.SVD_CONFIG)
    assert S.param_count(m) == 1_524_623_082  # the published size of the SVD / SVD-XT UNet


def test_svd_param_inventory_matches_oracle_state_dict():
    with torch.device("meta"):
        m = S.UNetSpatioTemporalConditionModel(**S.SVD_CONFIG)
    want = {k: tuple(v.shape) for k, v in m.named_parameters()}
    got = svd_param_shapes(SVD_CONFIG)
    assert set(got) == set(want)
    assert all(tuple(got[k]) == want[k] for k in want)


def test_single_key_cross_attention_is_value_projection():
    """The identity the plan relies on: attention over ONE key returns the value row for every query."""
    a = S.Attention(64, 4, 48).eval()
    x, ctx = torch.randn(3, 10, 64), torch.randn(3, 1, 48)
    with torch.no_grad():
        want = a(x, ctx)
        got = a.to_out[0](a.to_v(ctx)).expand(3, 10, 64)
    assert torch.allclose(want, got, atol=1e-6)


def test_svd_plan_executes_tiny_topology(built_lib):
    cfg = S.tiny_svd_config()
    m = S.build(cfg, seed=41, dtype=torch.float16)
    emu = EmuLib()
    eng = SVDUNetEngine.from_module(m, _host=EmuHost(emu))
    g = torch.Generator().manual_seed(42)
    B, Fr = 2, cfg["num_frames"]
    sample = torch.randn(B, Fr, 8, 16, 16, generator=g).half()
    ehs = torch.randn(B, 1, cfg["cross_attention_dim"], generator=g).half()
    tids = torch.tensor([[6.0, 127.0, 0.02], [7.0, 100.0, 0.1]])
    y = eng.forward(sample, torch.tensor([500.0, 321.0]), ehs, tids)
    with torch.no_grad():
        want = m.float()(sample.float(), torch.tensor([500.0, 321.0]), ehs.float(), tids).sample
    assert y.shape == want.shape == (B, Fr, 4, 16, 16)
    err = float((y.float() - want).norm() / want.norm())
    assert err < 5e-3, err
    plan = eng.get_plan(B, Fr, 16, 16)
    kinds = plan.summary()
    assert kinds["attn_temporal"]["count"] == 7 * B and kinds["conv_temporal"]["count"] == 2 * 11  # 7 transformers x B launches; 11 resnets
    assert "attn_cross" not in kinds  # every cross-attention collapsed to to_out(to_v(context))
    assert emu.calls.count("mix_rows") >= 11 + 7 * 4
