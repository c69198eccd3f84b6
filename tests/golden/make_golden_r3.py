#!/usr/bin/env python3
"""Golden vectors for what round 3 added to the oracle and to the C ABI, minted from the fp32 CPU oracle like make_golden{,_r2}.py
(same caveat: the reference holds no stored vectors, so these pin the ORACLE against drift and give engine tests an oracle-free check):

    python tests/golden/make_golden_r3.py   ->   tests/golden/unet_tiny_r3.pt, tests/golden/ops_r3.pt

  tiny UNet with `time_cond_proj_dim` (LCM w-embedding), `class_embed_type` "timestep" / "projection", text_time + class + cond stacked,
  `cross_attention_kwargs={"scale": s}` (a no-op without LoRA layers)                                            oracle/unet_ref.py
  the device-side schedule cursor (sfast_hip_schedule_advance): row i of the timestep / coefficient tables, cursor wrap     closed form
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import unet_ref as U  # noqa: E402
from oracle.ops_ref import ddim_schedule  # noqa: E402


def g(seed):
    return torch.Generator().manual_seed(seed)


def h(t):
    return t.half().float()


CASES = {
    "lcm_timestep_cond": dict(over=dict(time_cond_proj_dim=32), seed=301),
    "class_timestep": dict(over=dict(class_embed_type="timestep"), seed=302),
    "class_projection": dict(over=dict(class_embed_type="projection", projection_class_embeddings_input_dim=40), seed=303),
    "stacked": dict(over=dict(class_embed_type="timestep", addition_embed_type="text_time", addition_time_embed_dim=32,
                              projection_class_embeddings_input_dim=64 + 6 * 32, time_cond_proj_dim=16), seed=304),
}


def inputs(name, cfg, seed):
    gen = g(seed)
    s = h(torch.randn(2, 4, 16, 16, generator=gen))
    e = h(torch.randn(2, 20, cfg["cross_attention_dim"], generator=gen))
    kw = {}
    if cfg.get("time_cond_proj_dim"):
        kw["timestep_cond"] = h(torch.randn(2, cfg["time_cond_proj_dim"], generator=gen))
    if cfg.get("class_embed_type") == "timestep":
        kw["class_labels"] = torch.tensor([3.0, 977.0])
    if cfg.get("class_embed_type") == "projection":
        kw["class_labels"] = h(torch.randn(2, cfg["projection_class_embeddings_input_dim"], generator=gen))
    if cfg.get("addition_embed_type") == "text_time":
        kw["added_cond_kwargs"] = dict(text_embeds=h(torch.randn(2, 64, generator=gen)), time_ids=torch.tensor([[512., 512, 0, 0, 512, 512]] * 2))
    return s, e, kw


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    out = {}
    for name, c in CASES.items():
        cfg = U.tiny_config(**c["over"])
        m = U.build(cfg, seed=c["seed"])
        m.load_state_dict({k: h(v) for k, v in m.state_dict().items()})
        s, e, kw = inputs(name, cfg, c["seed"] + 1000)
        with torch.no_grad():
            y = m(s, 321, e, **kw).sample
            y_scale = m(s, 321, e, cross_attention_kwargs={"scale": 0.7}, **kw).sample
        assert torch.equal(y, y_scale), "cross_attention_kwargs.scale must be a no-op without LoRA layers"
        out[name] = dict(over=c["over"], seed=c["seed"], timestep=321, y=y)
    torch.save(out, os.path.join(HERE, "unet_tiny_r3.pt"))

    # schedule cursor: tables of a 7-step DDIM schedule, the rows three consecutive advances hand out starting at cursor 5 (wraps)
    ts, coef = ddim_schedule(7)
    ts_table = torch.tensor([[float(t)] for t in ts], dtype=torch.float32)
    coef_table = torch.tensor(coef, dtype=torch.float32)
    start = 5
    rows = [(start + i) % 7 for i in range(3)]
    ops = dict(schedule_advance=dict(ts_table=ts_table, coef_table=coef_table, start=start, n_steps=7, rows=rows,
                                     ts_out=ts_table[rows].clone(), coef_out=coef_table[rows].clone(), cursor_after=(start + 3) % 7))
    torch.save(ops, os.path.join(HERE, "ops_r3.pt"))
    print("wrote unet_tiny_r3.pt, ops_r3.pt", {k: tuple(v["y"].shape) for k, v in out.items()}, rows)


if __name__ == "__main__":
    main()
