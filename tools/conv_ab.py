"""Same-process A/B of the conv pipes on the 3x3 convs of the SD1.5 / SDXL UNets: for every shape, every (variant, split) the library
accepts is timed (interleaved rounds, hipGraph of 8 back-to-back calls so that host launch overhead does not mask 15-40 us kernels);
prints the best implicit-im2col choice (pipes 0-2: variants 1..26), the best LDS-patch choice (pipe 3: variants 31..34) and what the
traffic model predicts: bytes into LDS / 13.3 TB/s."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probes  # noqa: E402

_probes.use_probe_build()  # experiment / ablation / patch-pipe instantiations exist only in libsfast_hip_probes.so
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402

SHAPES = [  # B, Cin, C2, H, W, Cout
    (2, 320, 0, 64, 64, 320), (2, 640, 0, 32, 32, 640), (2, 1280, 0, 16, 16, 1280), (2, 1280, 0, 8, 8, 1280),
    (2, 320, 0, 32, 32, 640), (2, 640, 0, 16, 16, 1280), (2, 1280, 1280, 16, 16, 1280), (2, 640, 320, 64, 64, 320),
    (2, 1280, 640, 32, 32, 640), (2, 1280, 1280, 8, 8, 1280),
]
if "--sdxl" in sys.argv:
    SHAPES = [(2, 320, 0, 128, 128, 320), (2, 640, 0, 64, 64, 640), (2, 1280, 0, 32, 32, 1280), (2, 1280, 1280, 32, 32, 1280)]
OLD = [(v, s) for v in (2, 12, 17, 21, 22, 23, 24, 25, 26, 3, 13) for s in (1, 2, 3, 4, 6, 8, 12, 16)]
NEW = [(v, s) for v in (31, 32, 34) for s in (1, 2, 3, 4, 5, 8, 10, 20)]
torch.manual_seed(0)
stream = torch.cuda.Stream()


def graph_of(fn, n=8):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(n):
                fn()
    return g


for (B, Cin, C2, H, W, Cout) in SHAPES:
    cl = torch.channels_last
    x = torch.randn(B, Cin, H, W, device="cuda", dtype=torch.float16).contiguous(memory_format=cl)
    x2 = torch.randn(B, C2, H, W, device="cuda", dtype=torch.float16).contiguous(memory_format=cl) if C2 else None
    w = (torch.randn(Cout, Cin + C2, 3, 3, device="cuda", dtype=torch.float16) * ((Cin + C2) * 9) ** -0.5).contiguous(memory_format=cl)
    b = torch.randn(Cout, device="cuda", dtype=torch.float16)
    z = torch.randn(B, Cout, H, W, device="cuda", dtype=torch.float16).contiguous(memory_format=cl)
    out = torch.empty_like(z)
    ref = None
    res = {}
    cands = []
    for (v, s) in OLD + NEW:
        try:
            y = F.conv2d(x, w, b, z=z, padding=1, x2=x2, variant=v, split_k=s)
        except Exception:
            continue
        k = L.last_kernel()
        tag = "patch" if v >= 30 else "ws" if v >= 20 else "dma" if v >= 10 else "reg"
        if tag not in k.split(",")[-1] or f"split={s}," not in k:
            continue
        if ref is None:
            ref = y.float()
        err = float((y.float() - ref).abs().max())
        cands.append((v, s, k, err, graph_of(lambda v=v, s=s: F.conv2d(x, w, b, z=z, padding=1, x2=x2, variant=v, split_k=s))))
    best = {c[:2]: 1e9 for c in cands}
    for rep in range(3):
        for (v, s, k, err, g) in cands:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record(stream)
                g.replay()
                e1.record(stream)
            torch.cuda.synchronize()
            best[(v, s)] = min(best[(v, s)], e0.elapsed_time(e1) / 8 * 1e3)
    M, K = B * H * W, (Cin + C2) * 9
    flops = 2.0 * M * Cout * K
    old = min(((best[c[:2]], c) for c in cands if c[0] < 30), default=None)
    new = min(((best[c[:2]], c) for c in cands if c[0] >= 30), default=None)
    print(f"conv3x3 B={B} {Cin}+{C2}->{Cout} @{H}x{W}  ({flops / 1e9:.1f} GFLOP, weights {Cout * K * 2 / 1e6:.1f} MB)")
    for label, r in (("  im2col pipes", old), ("  patch pipe  ", new)):
        if r:
            t, (v, s, k, err, _) = r
            print(f"{label}: {t:7.1f} us  {flops / t / 1e6:6.1f} TF ({flops / t / 1e6 / 25:4.1f} % of 2.5 PF)  {k}  max|diff| vs first {err:.1e}")
    if old and new:
        print(f"  speed-up {old[0] / new[0]:.2f}x")
    allp = sorted((best[c[:2]], c[0], c[1]) for c in cands if c[0] >= 30)
    print("  patch candidates:", ", ".join(f"v{v}/s{s} {t:.1f}" for t, v, s in allp[:8]), flush=True)
