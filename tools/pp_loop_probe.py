"""Where the K loop of the ping-pong conv kernel (igemm_pp.h) spends its time: the kernel's timing-only experiment instantiations
(SFAST_IGEMM_EXP, results are garbage) against the full kernel, same launch geometry, hipGraph of 8 calls.
bit 0 no MFMAs | bit 1 no fragment reads | bit 2 no LDS-DMA requests inside the loop | bit 3 no s_setprio (results correct)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probes  # noqa: E402

_probes.use_probe_build()  # experiment instantiations exist only in libsfast_hip_probes.so
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402

CASES = [  # B, Cin, C2, H, W, Cout, variant, split
    (16, 320, 0, 64, 64, 320, 52, 1), (16, 320, 0, 64, 64, 320, 56, 1), (16, 320, 0, 64, 64, 320, 58, 1), (16, 640, 0, 32, 32, 640, 58, 1), (16, 640, 0, 32, 32, 640, 53, 1),
]
NAMES = {0: "full", 1: "no MFMA", 2: "no fragment reads", 3: "no MFMA, no fragment reads", 4: "no in-loop DMA", 7: "loop skeleton (barriers only)",
         6: "MFMAs + barriers only (no fragment reads, no in-loop DMA)", 5: "fragment reads + barriers only (no MFMA, no in-loop DMA)",
         8: "full, s_setprio 1 in the memory part"}
SKEL = 7
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


def timeline(run, nblocks):
    """EXP 16: shader-clock stamps of K-tile 8 in waves 0 (group 0) and 4 (group 1) of every workgroup; prints mean cycles between points"""
    os.environ["SFAST_IGEMM_EXP"] = "16"
    trace.zero_()
    lib.sfast_hip_set_trace(trace.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.sfast_hip_set_trace(None)
    rec = trace[: nblocks * 16].view(nblocks, 16).double()
    okr = rec[:, 0] > 0
    ph = ["setup", "prologue issue", "first tile landed", "K loop", "epilogue issued", "stores drained"]
    d = (rec[okr][:, 1:7] - rec[okr][:, 0:6]) / 100.0  # 100 MHz ticks -> us
    print("   workgroup phases (median us): " + ", ".join(f"{n} {d[:, i].median():.2f}" for i, n in enumerate(ph)) +
          f" | total {((rec[okr][:, 6] - rec[okr][:, 0]) / 100.0).median():.2f}; launch span {(rec[okr][:, 6].max() - rec[okr][:, 0].min()) / 100.0:.1f} us")
    t = trace[16 * 32768: 16 * 32768 + nblocks * 32].view(nblocks, 2, 16).double()
    names = ["reads+requests issued", "vmcnt wait", "lgkmcnt wait", "barrier 1", "MFMAs issued", "barrier 2"]
    for g in range(2):
        d = t[:, g, 1:7] - t[:, g, 0:6]
        ok = (t[:, g, 0] > 0) & (t[:, g, 6] > t[:, g, 0])
        d = d[ok]
        print(f"   timeline group {g} ({int(ok.sum())} workgroups), mean shader clocks: " + ", ".join(f"{n} {d[:, i].mean():.0f}" for i, n in enumerate(names)) +
              f"  | K-tile total {(t[ok][:, g, 6] - t[ok][:, g, 0]).mean():.0f}")
    # offset of group 1 against group 0 at point 0
    okb = (t[:, 0, 0] > 0) & (t[:, 1, 0] > 0)
    print(f"   group 1 starts its memory part {(t[okb][:, 1, 0] - t[okb][:, 0, 0]).mean():.0f} clocks after group 0")


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
        print(f"   {NAMES[ex]:58s} {t:7.1f} us   {flops / t / 1e6:6.0f} 'TF/s'", flush=True)
    bm = 256
    bn = 160 if v in (52, 56, 58) else 256 if v == 53 else 128
    timeline(run, (M // bm) * ((Cout + bn - 1) // bn))
