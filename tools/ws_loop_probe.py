"""Where the K loop of the wave-specialised LDS-DMA conv kernel (igemm_glds_ws.hip) spends its time: the kernel's timing-only
experiment instantiations (SFAST_IGEMM_EXP, results are garbage) against the full kernel, same launch geometry, hipGraph of 8 calls.
bit 0 no MFMAs | bit 1 no fragment reads | bit 2 no LDS-DMA requests inside the loop | bit 3 only the weight half of the requests."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probes  # noqa: E402

_probes.use_probe_build()  # experiment / ablation / patch-pipe instantiations exist only in libsfast_hip_probes.so
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402

CASES = [  # B, Cin, C2, H, W, Cout, variant, split
    (2, 320, 0, 64, 64, 320, 21, 1), (2, 640, 0, 32, 32, 640, 21, 3), (2, 640, 320, 64, 64, 320, 22, 2), (2, 1280, 640, 32, 32, 640, 22, 4),
    (2, 1280, 1280, 16, 16, 1280, 22, 8),
]
if "--patch" in sys.argv:  # the LDS-patch kernel (conv_patch.hip): bit 2 = no weight requests in the loop, bit 3 = no patch requests in the loop
    CASES = [(2, 320, 0, 64, 64, 320, 32, 1), (2, 640, 0, 32, 32, 640, 32, 3), (2, 640, 320, 64, 64, 320, 31, 2), (2, 1280, 1280, 16, 16, 1280, 31, 8)]
NAMES = {0: "full", 1: "no MFMA", 2: "no fragment reads", 3: "no MFMA, no fragment reads", 4: "no in-loop DMA", 7: "loop skeleton",
         8: "weight requests only (x requests read the zero block)", 16: "full, s_setprio 3 in the producer waves",
         32: "full, s_setprio 3 in the consumer waves", 6: "MFMAs + barriers only (no fragment reads, no in-loop DMA)",
         5: "fragment reads + barriers only (no MFMA, no in-loop DMA)", 64: "full, accumulators in AGPRs",
         68: "no in-loop DMA, accumulators in AGPRs", 128: "full, loop header behind the barrier (exact LDS waits)",
         132: "no in-loop DMA, loop header behind the barrier"}
if "--patch" in sys.argv:
    NAMES = {0: "full", 1: "no MFMA", 2: "no fragment reads", 3: "no MFMA, no fragment reads", 4: "no in-loop weight DMA", 8: "no in-loop patch DMA",
             12: "no in-loop DMA at all", 15: "loop skeleton"}
SKEL = 15 if "--patch" in sys.argv else 7
stream = torch.cuda.Stream()
lib = L.load()
trace = torch.zeros(16 * 65536, dtype=torch.int64, device="cuda")


def graph_of(fn, n=8):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(n):
                fn()
    return g


for (B, Cin, C2, H, W, Cout, v, s) in CASES:
    cl = torch.channels_last
    x = torch.randn(B, Cin, H, W, device="cuda", dtype=torch.float16).contiguous(memory_format=cl)
    x2 = torch.randn(B, C2, H, W, device="cuda", dtype=torch.float16).contiguous(memory_format=cl) if C2 else None
    w = (torch.randn(Cout, Cin + C2, 3, 3, device="cuda", dtype=torch.float16) * ((Cin + C2) * 9) ** -0.5).contiguous(memory_format=cl)
    b = torch.randn(Cout, device="cuda", dtype=torch.float16)
    run = lambda: F.conv2d(x, w, b, padding=1, x2=x2, variant=v, split_k=s)
    graphs = {}
    for ex in NAMES:
        os.environ["SFAST_IGEMM_EXP"] = str(ex)
        lib.sfast_hip_set_trace(trace.data_ptr())
        run()
        if ex == 0:
            kname = L.last_kernel()
        graphs[ex] = graph_of(run)
    lib.sfast_hip_set_trace(None)
    best = {ex: 1e9 for ex in NAMES}
    for rep in range(4):
        for ex, g in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record(stream)
                g.replay()
                e1.record(stream)
            torch.cuda.synchronize()
            best[ex] = min(best[ex], e0.elapsed_time(e1) / 8 * 1e3)
    M, K = B * H * W, (Cin + C2) * 9
    flops = 2.0 * M * Cout * K
    tiles_k = K // 64 // s
    print(f"conv3x3 B={B} {Cin}+{C2}->{Cout} @{H}x{W}  {kname}  K-tiles per workgroup {tiles_k}  pure MFMA time at 2.5 PF {flops / 2.5e15 * 1e6:.1f} us")
    for ex, t in best.items():
        print(f"   {NAMES[ex]:58s} {t:7.1f} us   ({(t - best[SKEL]) / tiles_k * 1e3:6.0f} ns per K-tile above the skeleton)", flush=True)
