#!/usr/bin/env python3
"""Launch a few representative hot-path kernels in isolation (for rocprofv3 --pmc runs).

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES ... -- python tools/pmc_probe.py
Each probe = one op of the SD1.5 B=2 plan re-launched REPS times with a forced (variant, split).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-fast_amd"))

os.environ.setdefault("SFAST_AUTOTUNE", "0")
import torch  # noqa: E402

from sfast.engine import UNet2DEngine  # noqa: E402
from sfast.engine.unet_spec import SD15_CONFIG, random_params  # noqa: E402

REPS = int(os.environ.get("PROBE_REPS", "5"))
PROBES = [
    # op name, variant, split  (variants: include/sfast_hip.h / igemm.hip kVariants)
    ("up_blocks.3.resnets.0.conv1", 15, 4),   # 256x128 dma3
    ("up_blocks.3.resnets.0.conv1", 17, 4),   # 128x160 dma2
    ("up_blocks.3.resnets.0.conv1", 2, 4),    # 128x160 reg
    ("down_blocks.0.resnets.0.conv2", 18, 1),  # 64x64 dma3
    ("down_blocks.0.resnets.0.conv2", 12, 2),  # 128x160 dma4
    ("down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_qkv", 18, 1),
    ("down_blocks.0.attentions.0.transformer_blocks.0.ff.out", 18, 1),
    ("down_blocks.0.attentions.0.transformer_blocks.0.ff.geglu", 16, 1),
    ("down_blocks.2.resnets.1.conv2", 15, 12),
    ("down_blocks.0.attentions.0.transformer_blocks.0.attn1", 0, 0),
    ("down_blocks.0.resnets.0.norm1", 0, 0),
    ("down_blocks.0.attentions.0.transformer_blocks.0.norm1", 0, 0),
]


def main():
    dev = torch.device("cuda")
    eng = UNet2DEngine(SD15_CONFIG, random_params(SD15_CONFIG, device=dev))
    plan = eng.build_plan(2, 64, 64, 77)
    for t in plan.static_in.values():
        if t.dtype == torch.float16:
            t.normal_()
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    sp = torch.cuda.current_stream().cuda_stream
    plan.run(sp)  # populate every activation buffer with realistic data
    torch.cuda.synchronize()
    ops = {op.name: op for op in plan.ops}
    for name, v, s in PROBES:
        op = ops[name]
        for _ in range(REPS):
            if op.tune is not None:
                p, launch_with = op.tune
                p.variant, p.split_k = v, s
                rc = launch_with(sp, ws.data_ptr(), ws.numel())
                assert rc == 0, (name, rc)
                p.variant, p.split_k = 0, 0
            else:
                op.launch(sp)
        torch.cuda.synchronize()
    print("probe done")


if __name__ == "__main__":
    main()
