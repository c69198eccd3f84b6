#!/usr/bin/env python3
"""Kernel-only rates of the pipe-5 kernels IN STEADY STATE: the chip needs seconds of sustained load to reach its steady clocks
(profiles/r06_warmup_ramp_run43.log), and tools/pp_ab.py times bursts of 50 launches on an idle chip. Each case here is launched
back to back for PREHEAT seconds first, then timed as a hipGraph of 10 launches (best of 5, like pp_ab.py) -- and, when the script runs
under `rocprofv3 --kernel-trace`, the last 50 dispatches of every symbol are what tools/rocpd_summary.py averages.

    python tools/pp_steady.py            # VERDICT r05 item 1's two shapes + the other batch-16 convs
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402
from pp_ab import timed  # noqa: E402

PREHEAT = float(os.environ.get("PP_STEADY_PREHEAT", "3.0"))


def steady(label, flops, fn):
    fn()
    name = L.last_kernel()
    cold, _ = timed(fn)
    t_end = time.perf_counter() + PREHEAT
    while time.perf_counter() < t_end:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    hot, _ = timed(fn)
    for _ in range(50):   # the dispatches a kernel trace should average
        fn()
    torch.cuda.synchronize()
    print(f"{label:40s} idle chip {cold:8.1f} us {flops / cold / 1e6:6.0f} TF/s | after {PREHEAT:.0f} s of load {hot:8.1f} us {flops / hot / 1e6:6.0f} TF/s "
          f"({flops / hot / 1e6 / 2500:.2f} of 2.5 PF)  [{name}]", flush=True)


def main():
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(1)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    for label, B, Cin, hw, Cout, k, v in (("conv 320->320 @64^2 B16 (M=65536 N=320 K=2880)", 16, 320, 64, 320, 3, 58), ("conv 640->640 @32^2 B16", 16, 640, 32, 640, 3, 58),
                                          ("conv 960->320 @64^2 B16", 16, 960, 64, 320, 3, 58), ("conv 1280->1280 @16^2 B16", 16, 1280, 16, 1280, 3, 58)):
        x = cl(torch.randn(B, Cin, hw, hw, generator=gen, device=dev).half())
        w = cl((torch.randn(Cout, Cin, k, k, generator=gen, device=dev) * (k * k * Cin) ** -0.5).half())
        b = torch.randn(Cout, generator=gen, device=dev).half()
        steady(label, 2.0 * B * hw * hw * Cout * Cin * k * k, lambda: F.conv2d(x, w, b, padding=k // 2, variant=v, split_k=1))
    for label, M, N, K, variants in (("geglu 8192x640->2560 (weight rows 5120)", 8192, 2560, 640, (57, 53, 16)), ("geglu 65536x320->1280", 65536, 1280, 320, (57, 53)),
                                     ("geglu 16384x640->2560", 16384, 2560, 640, (57, 53))):
        x = torch.randn(M, K, generator=gen, device=dev).half()
        w = (torch.randn(2 * N, K, generator=gen, device=dev) * K ** -0.5).half()
        b = torch.randn(2 * N, generator=gen, device=dev).half()
        for v in variants:
            steady(f"{label} v{v}", 2.0 * M * 2 * N * K, lambda: F.linear(x, w, b, geglu=True, variant=v, split_k=1))


if __name__ == "__main__":
    main()
