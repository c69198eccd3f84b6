"""Workgroup-size choice of sfast_hip_attention (variant 2 = 64 query rows per workgroup, 4 = 128) at the UNet's shapes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402

SHAPES = [(2, 4096, 4096, 8, 40), (2, 1024, 1024, 8, 80), (2, 256, 256, 8, 160), (2, 64, 64, 8, 160), (2, 4096, 77, 8, 40),
          (2, 1024, 77, 8, 80), (2, 4096, 4096, 10, 64), (2, 1024, 1024, 20, 64), (2, 4096, 77, 10, 64), (2, 1024, 77, 20, 64)]
for B, Sq, Skv, H, D in SHAPES:
    q = torch.randn(B, Sq, H, D, device="cuda", dtype=torch.float16)
    k = torch.randn(B, Skv, H, D, device="cuda", dtype=torch.float16)
    v = torch.randn(B, Skv, H, D, device="cuda", dtype=torch.float16)
    res = []
    for nw in (0, 2, 4):
        for _ in range(5):
            F.attention(q, k, v, variant=nw)
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                F.attention(q, k, v, variant=nw)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
        res.append(best)
    print(f"D={D:3d} Sq={Sq:4d} Skv={Skv:4d} H={H:2d}: auto {res[0]:7.1f} us   nw=2 {res[1]:7.1f} us   nw=4 {res[2]:7.1f} us", flush=True)
