#!/usr/bin/env python3
"""Same-process A/B of sfast_hip_gn_conv2d (csrc/gnconv.hip) against the two operators it replaces (GroupNorm+SiLU, then the autotuned
3x3 conv) on SD1.5's 8x8-level shapes at CFG batch 2. Each form is captured into a hipGraph of REPS back-to-back calls (the Python
wrappers cost more than the kernels) and timed over replays. Run under rocprofv3 --kernel-trace --stats for per-kernel durations."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402

REPS = 20
SHAPES = [("1280->1280 conv1 (+temb)", 2, 1280, 0, 1280, dict(rowbias=True)),
          ("1280->1280 conv2 (+residual)", 2, 1280, 0, 1280, dict(z=True)),
          ("cat 1280+1280->1280 conv1", 2, 1280, 1280, 1280, dict(rowbias=True)),
          ("literal B=1 1280->1280", 1, 1280, 0, 1280, dict(z=True))]


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(REPS):
                fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / REPS)
    return best


def main():
    dev = "cuda"
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    for name, B, C1, C2, Cout, ex in SHAPES:
        gen = torch.Generator(device=dev).manual_seed(1)
        Cin = C1 + C2
        x = cl(torch.randn(B, C1, 8, 8, generator=gen, device=dev).half())
        x2 = cl(torch.randn(B, C2, 8, 8, generator=gen, device=dev).half()) if C2 else None
        gw, gb = torch.randn(Cin, generator=gen, device=dev).half(), torch.randn(Cin, generator=gen, device=dev).half()
        w = cl((torch.randn(Cout, Cin, 3, 3, generator=gen, device=dev) * (9 * Cin) ** -0.5).half())
        b = torch.randn(Cout, generator=gen, device=dev).half()
        z = cl(torch.randn(B, Cout, 8, 8, generator=gen, device=dev).half()) if ex.get("z") else None
        rb = torch.randn(B, Cout, generator=gen, device=dev).half() if ex.get("rowbias") else None
        fused = lambda: F.gn_conv2d(x, 32, gw, gb, w, b, x2=x2, z=z, rowbias=rb)
        two = lambda: F.conv2d(F.group_norm(x, 32, gw, gb, 1e-5, "silu", x2=x2), w, b, z=z, padding=1, rowbias=rb)
        y1 = fused()
        k1 = L.last_kernel()
        y2 = two()
        k2 = L.last_kernel()
        err = float((y1.float() - y2.float()).abs().max())
        t1, t2 = timed(fused), timed(two)
        wbytes = Cout * Cin * 9 * 2
        # the same weight-streaming launch WITHOUT the normalisation (groups = 0: "wsconv") against the autotuned conv alone
        xn = F.group_norm(x, 32, gw, gb, 1e-5, "silu", x2=x2)
        plain = lambda: F.gn_conv2d(xn, 0, None, None, w, b, z=z, rowbias=rb)
        conv = lambda: F.conv2d(xn, w, b, z=z, padding=1, rowbias=rb)
        y3 = plain()
        k3 = L.last_kernel()
        y4 = conv()
        k4 = L.last_kernel()
        t3, t4 = timed(plain), timed(conv)
        print(f"{name:32s} wsconv {t3:6.1f} us  conv alone {t4:6.1f} us  ratio {t4 / t3:4.2f}x  max|diff| {float((y3.float() - y4.float()).abs().max()):.3g}  "
              f"[{k3}] vs [{k4}]", flush=True)
        print(f"{name:32s} fused {t1:6.1f} us ({wbytes / t1 / 1e6:5.2f} TB/s of weights)  two operators {t2:6.1f} us ({wbytes / t2 / 1e6:5.2f} TB/s)  "
              f"ratio {t2 / t1:4.2f}x  max|diff| {err:.3g}  [{k1}] vs [gn + {k2}]", flush=True)


if __name__ == "__main__":
    main()
