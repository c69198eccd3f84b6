"""Timing (+ error vs torch SDPA) of sfast_hip_attention at the UNet's shapes; `attn_ab.py <other libsfast_hip.so>` times another
build for a same-box A/B, `--trace` adds the per-phase s_memtime split of the tile loop (sfast_hip_set_trace)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
import torch  # noqa: E402

from sfast.hip import lib as L  # noqa: E402

if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
    L.LIB_PATH = os.path.abspath(sys.argv[1])
from sfast.hip import functional as F  # noqa: E402

SHAPES = [  # (B, Sq, Skv, H, D)
    (2, 4096, 4096, 8, 40), (2, 1024, 1024, 8, 80), (2, 256, 256, 8, 160), (2, 4096, 77, 8, 40),
    (2, 4096, 4096, 10, 64), (2, 1024, 1024, 20, 64), (2, 1024, 77, 20, 64),
]
torch.manual_seed(0)
out = []
# `--variants 0,32,62,64`: the kernels to time per shape, interleaved rounds in ONE process (guide rule 24): 0 = the library's own choice,
# 32 = first-generation kernel (32 query rows per wave), 62 / 64 = attention_q64.hip with 2 / 4 waves per workgroup
VARIANTS = [0]
if "--variants" in sys.argv:
    VARIANTS = [int(x) for x in sys.argv[sys.argv.index("--variants") + 1].split(",")]
if "--fixed-cost" in sys.argv:  # same query set, 2 .. 64 key tiles: the intercept is launch + prologue + epilogue, the slope the tile loop
    SHAPES = [(2, 4096, skv, 8, d) for d in (40, 64) for skv in (128, 512, 2048, 4096)]
if "--more-shapes" in sys.argv:
    SHAPES += [(16, 4096, 4096, 8, 40), (2, 9216, 9216, 5, 64), (8, 2304, 2304, 10, 64), (2, 1024, 1024, 8, 40), (4, 1024, 1024, 10, 64)]
for B, Sq, Skv, H, D in SHAPES:
    q = torch.randn(B, Sq, H, D, device="cuda", dtype=torch.float16)
    k = torch.randn(B, Skv, H, D, device="cuda", dtype=torch.float16)
    v = torch.randn(B, Skv, H, D, device="cuda", dtype=torch.float16)
    ref = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).float() \
        if B * H * Sq * Skv > (1 << 31) else torch.nn.functional.scaled_dot_product_attention(
            q.transpose(1, 2).float(), k.transpose(1, 2).float(), v.transpose(1, 2).float()).transpose(1, 2)
    errs, names, best = {}, {}, {vr: 1e9 for vr in VARIANTS}
    for vr in VARIANTS:
        for _ in range(3):
            o = F.attention(q, k, v, variant=vr)
        names[vr] = L.last_kernel()
        errs[vr] = (o.float() - ref).abs().max().item()
    for rep in range(5):
        for vr in VARIANTS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                F.attention(q, k, v, variant=vr)
            e1.record()
            torch.cuda.synchronize()
            best[vr] = min(best[vr], e0.elapsed_time(e1) / 20 * 1e3)
    fl = 4.0 * B * H * Sq * Skv * D
    for vr in VARIANTS:
        out.append(f"B={B:2d} D={D:3d} Sq={Sq:4d} Skv={Skv:4d} H={H:2d} variant={vr:2d} {names[vr]:24s}: {best[vr]:7.1f} us  {fl / best[vr] / 1e6:6.1f} TF  "
                   f"({fl / best[vr] / 1e6 / 25.0:4.1f} % of 2.5 PF)  err {errs[vr]:.1e}")
if "--trace" in sys.argv:
    import numpy as np
    lib = L.load()
    for B, Sq, Skv, H, D in [(1, 4096, 4096, 8, 40), (2, 4096, 4096, 8, 40), (4, 4096, 4096, 8, 40), (2, 4096, 4096, 10, 64)]:
        q, k, v = [torch.randn(B, S_, H, D, device="cuda", dtype=torch.float16) for S_ in (Sq, Skv, Skv)]
        trace = torch.zeros(16 * 8192, dtype=torch.int64, device="cuda")
        lib.sfast_hip_set_trace(trace.data_ptr())
        F.attention(q, k, v, variant=4)
        torch.cuda.synchronize()
        lib.sfast_hip_set_trace(None)
        nwg = B * H * Sq // 128
        r = trace.cpu().numpy().reshape(-1, 16)[:nwg].astype(np.float64)
        t = r[:, 5:6]
        names = ["top", "phase1 QK||exp", "phase2 PV||max,stage", "barrier"]
        per = r[:, :4] / t
        line = "  ".join(f"{n} {per[:, i].mean():6.0f}" for i, n in enumerate(names))
        out.append(f"trace B={B} D={D} Sq={Sq}: cycles/tile (wave 0 of each WG, mean over {nwg} WGs): {line}  | sum {per.sum(1).mean():6.0f}  "
                   f"kernel {r[:, 4].mean() / t.mean():6.0f}/tile incl. prologue+epilogue")
print(os.path.basename(L.LIB_PATH), flush=True)
print("\n".join(out), flush=True)
