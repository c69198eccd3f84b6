"""Occupancy / overlap probe: attention D=40 S=4096 at growing batch (workgroups per CU) for both workgroup sizes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402

D = int(os.environ.get("ATTN_D", "40"))
H = 8
S = 4096
print("D", D, flush=True)
for nw in (4, 2):
    for B in (1, 2, 3, 4, 6, 8):
        q, k, v = [torch.randn(B, S, H, D, device="cuda", dtype=torch.float16) for _ in range(3)]
        for _ in range(3):
            F.attention(q, k, v, variant=nw)
        best = 1e9
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                F.attention(q, k, v, variant=nw)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        wgs = B * H * S // (nw * 32)
        print(f"nw={nw} B={B} wgs={wgs:5d} ({wgs / 256:.1f}/CU, {wgs * nw / 1024:.1f} waves/SIMD): {best:7.1f} us  "
              f"{4.0 * B * H * S * S * D / best / 1e6:6.1f} TF", flush=True)
