#!/usr/bin/env python3
"""Ablation timing of the LDS-DMA igemm pipe: full / no-MFMA / no-refill / neither, per selected op."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-fast_amd"))
os.environ.setdefault("SFAST_AUTOTUNE", "0")
import torch  # noqa: E402

from sfast.engine import UNet2DEngine  # noqa: E402
from sfast.engine.unet_spec import SD15_CONFIG, random_params  # noqa: E402
from sfast.hip import lib as L  # noqa: E402

PROBES = [
    ("down_blocks.0.resnets.0.conv2", [(12, 2), (17, 2), (17, 4), (16, 2), (15, 2), (2, 2), (2, 4)]),
    ("up_blocks.3.resnets.0.conv1", [(12, 4), (15, 4)]),
    ("down_blocks.0.attentions.0.transformer_blocks.0.ff.out", [(13, 1), (18, 1), (16, 1), (3, 1)]),
    ("down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out", [(13, 1), (18, 1), (3, 1)]),
    ("down_blocks.0.attentions.0.transformer_blocks.0.ff.geglu", [(11, 1), (16, 1), (18, 1), (1, 1), (3, 1)]),
    ("down_blocks.2.resnets.1.conv2", [(15, 12), (11, 6)]),
]


def timeit(fn, n=10):
    st = torch.cuda.current_stream()
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(n):
        fn()
    b.record(st)
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda")
    eng = UNet2DEngine(SD15_CONFIG, random_params(SD15_CONFIG, device=dev))
    plan = eng.build_plan(2, 64, 64, 77)
    lib = L.load()
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    sp = torch.cuda.current_stream().cuda_stream
    ops = {op.name: op for op in plan.ops}
    for name, cfgs in PROBES:
        op = ops[name]
        p, launch_with = op.tune
        for v, s in cfgs:
            p.variant, p.split_k = v, s
            row = []
            for dbg in (0, 1, 2, 3):
                lib.sfast_hip_set_debug(dbg)
                assert launch_with(sp, ws.data_ptr(), ws.numel()) == 0
                row.append(timeit(lambda: launch_with(sp, ws.data_ptr(), ws.numel())))
            lib.sfast_hip_set_debug(0)
            kern = L.last_kernel()
            print(f"{name[-40:]:40s} {kern:44s} full {row[0]:7.1f} | no-mfma {row[1]:7.1f} | no-refill {row[2]:7.1f} | neither {row[3]:7.1f} us  ({op.flops / row[0] / 1e6:6.0f} TF)")
        p.variant, p.split_k = 0, 0


if __name__ == "__main__":
    main()
