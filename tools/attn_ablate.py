"""Where the tile loop of attention_q64.hip spends its time: timing-only instantiations with parts of the loop removed (ABL bit mask,
see the kernel), all in one process, interleaved rounds. Shapes with exactly 256 workgroups (one per CU, one round)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probes  # noqa: E402

_probes.use_probe_build()  # experiment / ablation / patch-pipe instantiations exist only in libsfast_hip_probes.so
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402

NAMES = {0: "full kernel", 1: "no v_exp", 3: "no v_exp, no convert", 4: "no row maxima", 7: "no softmax VALU at all", 8: "no V^T fragment reads",
         16: "no K fragment reads", 24: "no fragment reads", 32: "no staging stores", 64: "no global prefetch", 96: "no staging (loads + stores)",
         128: "no barrier", 224: "no staging, no barrier", 256: "no QK^T MFMAs", 512: "no PV MFMAs", 768: "no MFMAs", 255: "MFMAs only",
         1023: "loop skeleton"}
torch.manual_seed(0)
for (B, S, H, D) in ((2, 4096, 8, 40), (2, 4096, 8, 64)):
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
    best = {m: 1e9 for m in NAMES}
    for m in NAMES:
        F.attention(q, k, v, variant=64 if m == 0 else 1000 + m)
    for rep in range(4):
        for m in NAMES:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                F.attention(q, k, v, variant=64 if m == 0 else 1000 + m)
            e1.record()
            torch.cuda.synchronize()
            best[m] = min(best[m], e0.elapsed_time(e1) / 10 * 1e3)
    tiles = S // 64
    print(f"D={D} S={S} B*H={B * H} (256 workgroups, {tiles} tiles each; MFMAs per tile: {28 if D == 40 else 40} x 32 cycles)")
    for m in NAMES:
        print(f"  {NAMES[m]:34s} {best[m]:7.1f} us   {best[m] / tiles * 1e3:7.0f} ns/tile   delta vs full {best[m] - best[0]:+7.1f} us", flush=True)
