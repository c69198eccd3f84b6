#!/usr/bin/env python3
"""Golden vectors for the round-2 entry points, minted from the fp32 CPU oracle like make_golden.py (same caveat: the reference
holds no stored vectors, so these pin the ORACLE against drift and give the GPU tests an oracle-free check):

    python tests/golden/make_golden_r2.py   ->   tests/golden/ops_r2.pt, tests/golden/svd_tiny.pt

  attention with an additive bias (full [B,H,Sq,Skv] tensor) and with a key-padding mask (-inf entries)   oracle/ops_ref.attention_ref
  the scheduler updates trace_scheduler runs as one kernel: DDIM (eta = 0) and Euler rows `prev = A x + B e`       closed form
  tiny spatio-temporal UNet (SVD topology: temporal resnets, temporal attention, AlphaBlender, added time ids)    oracle/svd_ref
"""
import math
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ops_ref as R  # noqa: E402
from oracle import svd_ref as S  # noqa: E402


def g(seed):
    return torch.Generator().manual_seed(seed)


def h(t):
    return t.half().float()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    out = {}
    q = h(torch.randn(2, 70, 3, 40, generator=g(101)))
    k = h(torch.randn(2, 77, 3, 40, generator=g(102)))
    v = h(torch.randn(2, 77, 3, 40, generator=g(103)))
    bias = h(torch.randn(2, 3, 70, 77, generator=g(104)))
    out["attention_full_bias"] = dict(q=q, k=k, v=v, bias=bias, y=R.attention_ref(q, k, v, attn_bias=bias))
    keep = torch.ones(2, 77)
    keep[0, 60:] = 0
    keep[1, 33:] = 0
    mask = torch.zeros(2, 1, 1, 77).masked_fill(keep[:, None, None, :] == 0, float("-inf"))
    out["attention_key_padding_mask"] = dict(q=q, k=k, v=v, bias=mask, valid=[60, 33], y=R.attention_ref(q, k, v, attn_bias=mask))

    # scheduler rows: SD's scaled-linear schedule, 10 inference steps
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2
    acp = torch.cumprod(1.0 - betas, dim=0)
    x, e = h(torch.randn(2, 4, 8, 8, generator=g(105))), h(torch.randn(2, 4, 8, 8, generator=g(106)))
    t, prev = 901, 801
    a_t, a_p = float(acp[t]), float(acp[prev])
    x0 = (x.double() - math.sqrt(1 - a_t) * e.double()) / math.sqrt(a_t)
    ddim = math.sqrt(a_p) * x0 + math.sqrt(1 - a_p) * e.double()
    out["ddim_step_eps"] = dict(sample=x, model_output=e, coef=[math.sqrt(a_p / a_t), math.sqrt(1 - a_p) - math.sqrt(a_p) * math.sqrt(1 - a_t) / math.sqrt(a_t)],
                                y=ddim.float())
    sig = ((1 - acp) / acp) ** 0.5
    s0, s1 = float(sig[901]), float(sig[801])
    xs = h(x * s0)
    out["euler_step_eps"] = dict(sample=xs, model_output=e, coef=[1.0, s1 - s0], y=(xs.double() + (s1 - s0) * e.double()).float(),
                                 scale=1.0 / math.sqrt(s0 * s0 + 1.0), y_scaled=(xs.double() / math.sqrt(s0 * s0 + 1.0)).float())
    for case in out.values():
        for k_, v_ in list(case.items()):
            if torch.is_tensor(v_) and not k_.startswith("y"):
                case[k_] = v_.half()
    torch.save(out, os.path.join(HERE, "ops_r2.pt"))

    cfg = S.tiny_svd_config()
    m = S.build(cfg, seed=777)
    m.load_state_dict({k_: h(v_) for k_, v_ in m.state_dict().items()})
    B, Fr, hw = 2, 3, 8
    sample = h(torch.randn(B, Fr, cfg["in_channels"], hw, hw, generator=g(110)))
    ehs = h(torch.randn(B, 1, cfg["cross_attention_dim"], generator=g(111)))
    tids = torch.tensor([[6.0, 127.0, 0.02]] * B)
    t = torch.tensor([500.0, 321.0])
    with torch.no_grad():
        y = m(sample, t, ehs, tids).sample
    torch.save(dict(config=cfg, seed=777, sample=sample.half(), encoder_hidden_states=ehs.half(), added_time_ids=tids, timesteps=t, y=y),
               os.path.join(HERE, "svd_tiny.pt"))
    print("wrote ops_r2.pt, svd_tiny.pt", {k_: tuple(v_["y"].shape) for k_, v_ in out.items()}, tuple(y.shape))


if __name__ == "__main__":
    main()
