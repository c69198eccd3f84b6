#!/usr/bin/env python3
"""Per-workgroup phase timeline of gnconv_kernel (sfast_hip_set_trace): slots 0 entry, 1 weight requests issued, 2 slice in LDS,
3 statistics, 4 normalised, 5 k loop done, 6 accumulators added, 7 slab stored (100 MHz wall clock)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402


def main():
    dev = "cuda"
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    lib = L.load()
    trace = torch.zeros(16 * 4096, dtype=torch.int64, device=dev)
    for name, B, C1, C2 in (("1280->1280", 2, 1280, 0), ("cat 2560->1280", 2, 1280, 1280)):
        gen = torch.Generator(device=dev).manual_seed(1)
        Cin = C1 + C2
        x = cl(torch.randn(B, C1, 8, 8, generator=gen, device=dev).half())
        x2 = cl(torch.randn(B, C2, 8, 8, generator=gen, device=dev).half()) if C2 else None
        gw, gb = torch.randn(Cin, generator=gen, device=dev).half(), torch.randn(Cin, generator=gen, device=dev).half()
        w = cl((torch.randn(1280, Cin, 3, 3, generator=gen, device=dev) * (9 * Cin) ** -0.5).half())
        b = torch.randn(1280, generator=gen, device=dev).half()
        F.gn_conv2d(x, 32, gw, gb, w, b, x2=x2)
        for rep in range(3):
            flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev).zero_()   # weights out of L2 / Infinity Cache
            del flush
            trace.zero_()
            lib.sfast_hip_set_trace(trace.data_ptr())
            F.gn_conv2d(x, 32, gw, gb, w, b, x2=x2)
            torch.cuda.synchronize()
            lib.sfast_hip_set_trace(None)
            t = trace.cpu().numpy().reshape(-1, 16)
            t = t[t[:, 0] != 0][:, :8].astype(np.int64)
            t0 = t[:, 0].min()
            names = ["issue W", "slice->LDS", "stats", "normalise", "k loop", "add waves", "slab store"]
            d = np.diff(t, axis=1) / 100.0
            print(f"{name} [{L.last_kernel()}] rep {rep}: {len(t)} workgroups; first entry -> last exit {(t[:, 7].max() - t0) / 100.0:.1f} us; "
                  f"entry spread {(t[:, 0].max() - t0) / 100.0:.1f} us")
            print("   phase (us)  median / p90 / max : " + "  ".join(f"{n} {np.median(d[:, i]):.2f}/{np.percentile(d[:, i], 90):.2f}/{d[:, i].max():.2f}"
                                                                     for i, n in enumerate(names)))
            print(f"   per-workgroup total median {np.median(t[:, 7] - t[:, 0]) / 100.0:.2f} us, max {(t[:, 7] - t[:, 0]).max() / 100.0:.2f} us", flush=True)


if __name__ == "__main__":
    main()
