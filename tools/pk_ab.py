#!/usr/bin/env python3
"""Pipe 4 (packed weights, csrc/igemm_pk.h) against the ring kernels on SD1.5 / SDXL layer shapes: every tile variant x split-K,
each timed as a hipGraph of REPS back-to-back launches. Prints the best of each family and every pipe-4 variant's best split."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402

REPS = 10
RING = [21, 22, 23, 24, 25, 26, 1, 2, 3]
PK = [41, 42, 43, 44, 45, 46]
SPLITS = [1, 2, 3, 4, 6, 8, 12]
CONVS = [("conv 320->320 @64^2", 2, 320, 64, 320, 3), ("conv 640->640 @32^2", 2, 640, 32, 640, 3), ("conv 1280->1280 @16^2", 2, 1280, 16, 1280, 3),
         ("conv 1280->1280 @8^2", 2, 1280, 8, 1280, 3), ("conv1x1 320->320 @64^2", 2, 320, 64, 320, 1), ("conv 960->320 @64^2", 2, 960, 64, 320, 3)]
GEMMS = [("ff.out 8192x320x1280", 8192, 320, 1280), ("qkv 8192x960x320", 8192, 960, 320), ("ff.out 2048x640x2560", 2048, 640, 2560),
         ("qkv 2048x1920x640", 2048, 1920, 640), ("sdxl ff.out 8192x1280x5120", 8192, 1280, 5120), ("sdxl qkv 8192x3840x1280", 8192, 3840, 1280),
         ("sdxl 2048x640x2560", 2048 * 4, 640, 2560)]


def timed(fn):
    try:
        fn()
    except Exception:
        return None, None
    name = L.last_kernel()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(REPS):
                fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / REPS)
    return best, name


def sweep(label, flops, call, pk):
    res = {}
    for v in RING + PK:
        for s in SPLITS:
            t, name = timed(lambda: call(v, s, pk if v >= 40 else None))
            if t is None or f"split={s}," not in name or ((",pk" in name) != (v >= 40)):
                continue
            if v not in res or t < res[v][0]:
                res[v] = (t, s, name)
    ring = min((res[v] for v in RING if v in res), key=lambda r: r[0])
    print(f"{label:32s} best ring {ring[0]:7.1f} us {flops / ring[0] / 1e6:6.0f} TF/s  [{ring[2]}]", flush=True)
    for v in PK:
        if v in res:
            t, s, name = res[v]
            print(f"{'':32s}   pk v{v} {t:7.1f} us {flops / t / 1e6:6.0f} TF/s  x{ring[0] / t:4.2f}  [{name}]", flush=True)


def main():
    dev = "cuda"
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    gen = torch.Generator(device=dev).manual_seed(1)
    for label, B, Cin, hw, Cout, k in CONVS:
        x = cl(torch.randn(B, Cin, hw, hw, generator=gen, device=dev).half())
        w = cl((torch.randn(Cout, Cin, k, k, generator=gen, device=dev) * (k * k * Cin) ** -0.5).half())
        b = torch.randn(Cout, generator=gen, device=dev).half()
        pk = F.pack_weight(w)
        sweep(label, 2.0 * B * hw * hw * Cout * Cin * k * k, lambda v, s, p: F.conv2d(x, w, b, padding=k // 2, variant=v, split_k=s, w_packed=p), pk)
    for label, M, N, K in GEMMS:
        x = torch.randn(M, K, generator=gen, device=dev).half()
        w = (torch.randn(N, K, generator=gen, device=dev) * K ** -0.5).half()
        b = torch.randn(N, generator=gen, device=dev).half()
        pk = F.pack_weight(w)
        sweep(label, 2.0 * M * N * K, lambda v, s, p: F.linear(x, w, b, variant=v, split_k=s, w_packed=p), pk)


if __name__ == "__main__":
    main()
