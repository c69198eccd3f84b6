#!/usr/bin/env python3
"""Pipe 5 (256-row ping-pong tiles, csrc/igemm_pp.h) against the older pipes on large-M layer shapes (SD1.5 at 8 images per GPU,
SDXL, SVD-XT): correctness of every pp variant against torch first (max abs err vs an fp32 matmul / conv of the same f16 inputs),
then every tile variant x split-K timed as a hipGraph of REPS back-to-back launches. Prints the best older kernel and every pp
variant with its TF/s (algorithmic flops: 2 M N K, GEGLU 2 M K 2N).

    python tools/pp_ab.py [--quick] [--only conv|gemm|geglu]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402

REPS = 10
OLD = [21, 22, 23, 24, 25, 1, 2, 5, 11, 12, 15]
OLD_GEGLU = [1, 11, 16, 21, 23]
PK = [41, 44, 46]
PP = [51, 52, 53, 55, 56, 57, 58]
PP_GEGLU = [53, 57]
SPLITS = [1, 2, 3]

# name, B, Cin, hw, Cout, k
CONVS = [("conv 320->320 @64^2 B16", 16, 320, 64, 320, 3), ("conv 640->640 @32^2 B16", 16, 640, 32, 640, 3),
         ("conv 1280->1280 @16^2 B16", 16, 1280, 16, 1280, 3), ("conv 960->320 @64^2 B16", 16, 960, 64, 320, 3),
         ("conv1x1 320->320 @64^2 B16", 16, 320, 64, 320, 1), ("conv 320->320 @128^2 B2 (sdxl)", 2, 320, 128, 320, 3),
         ("conv 640->640 @64^2 B2 (sdxl)", 2, 640, 64, 640, 3), ("conv 320->320 @64^2 B2", 2, 320, 64, 320, 3)]
GEMMS = [("qkv 65536x960x320", 65536, 960, 320), ("ff.out 65536x320x1280", 65536, 320, 1280), ("to_out 65536x320x320", 65536, 320, 320),
         ("qkv 16384x1920x640", 16384, 1920, 640), ("ff.out 16384x640x2560", 16384, 640, 2560),
         ("sdxl ff.out 8192x640x2560", 8192, 640, 2560), ("sdxl qkv 2048x3840x1280", 2048, 3840, 1280), ("sdxl ff.out 2048x1280x5120", 2048, 1280, 5120),
         ("svd ff.out 28800x1280x5120", 28800, 1280, 5120)]
GEGLUS = [("geglu 65536x320->1280", 65536, 1280, 320), ("geglu 16384x640->2560", 16384, 2560, 640), ("geglu 8192x640->2560 (sdxl)", 8192, 2560, 640),
          ("geglu 2048x1280->5120 (sdxl)", 2048, 5120, 1280), ("geglu 4096x1280->5120", 4096, 5120, 1280)]


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
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / REPS)
    return best, name


def tag(v):
    return "pp" if v >= 50 else "pk" if v >= 40 else "patch" if v >= 30 else "ws" if v >= 20 else "dma" if v >= 10 else "reg"


def sweep(label, flops, call, pk, old, quick):
    res = {}
    for v in old + (PK if pk is not None else []) + (PP_GEGLU if "geglu" in label else PP):
        for s in ([1] if (quick or v >= 50) else SPLITS):
            t, name = timed(lambda: call(v, s, pk if 40 <= v < 50 else None))
            if t is None or f"split={s}," not in name or tag(v) not in name.split(",")[-1]:
                continue
            if v not in res or t < res[v][0]:
                res[v] = (t, s, name)
    olds = [res[v] for v in res if v < 50]
    base = min(olds, key=lambda r: r[0]) if olds else None
    if base:
        print(f"{label:34s} best old {base[0]:8.1f} us {flops / base[0] / 1e6:6.0f} TF/s  [{base[2]}]", flush=True)
    for v in (PP_GEGLU if "geglu" in label else PP):
        if v in res:
            t, s, name = res[v]
            print(f"{'':34s}   pp v{v} {t:8.1f} us {flops / t / 1e6:6.0f} TF/s  x{(base[0] / t) if base else 0:4.2f}  [{name}]", flush=True)


def check(label, got, want):
    err = (got.float() - want).abs().max().item()
    ref = want.abs().max().item()
    ok = err <= 2e-2 * max(ref, 1.0)
    print(f"  check {label:48s} max|err| {err:.3e} (max|ref| {ref:.2f}) {'ok' if ok else 'FAIL'}  [{L.last_kernel()}]", flush=True)
    return ok


def correctness(dev):
    gen = torch.Generator(device=dev).manual_seed(3)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    ok = True
    for (B, Cin, hw, Cout, k, c2) in [(2, 320, 64, 320, 3, 0), (1, 640, 32, 640, 3, 0), (1, 64, 40, 96, 3, 0), (2, 320, 32, 320, 1, 0), (1, 640, 32, 320, 3, 320),
                                      (3, 128, 24, 200, 3, 0), (16, 320, 64, 320, 3, 0), (9, 64, 64, 160, 3, 0), (5, 128, 96, 96, 1, 64)]:
        x = cl(torch.randn(B, Cin, hw, hw, generator=gen, device=dev).half())
        x2 = cl(torch.randn(B, c2, hw, hw, generator=gen, device=dev).half()) if c2 else None
        w = cl((torch.randn(Cout, Cin + c2, k, k, generator=gen, device=dev) * (k * k * (Cin + c2)) ** -0.5).half())
        b = torch.randn(Cout, generator=gen, device=dev).half()
        z = cl(torch.randn(B, Cout, hw, hw, generator=gen, device=dev).half())
        xin = x if x2 is None else torch.cat([x, x2], 1)
        want = torch.nn.functional.conv2d(xin.float(), w.float(), b.float(), padding=k // 2) + z.float()
        for v in PP:
            for s in (1, 2):
                y = F.conv2d(x, w, b, z=z, padding=k // 2, x2=x2, variant=v, split_k=s)
                if "pp" not in L.last_kernel():
                    continue
                ok &= check(f"conv B{B} {Cin}+{c2}->{Cout} @{hw} k{k} v{v} s{s}", y, want)
    for (M, N, K) in [(8192, 320, 320), (4096, 960, 320), (1000, 640, 2560), (256, 1280, 1280), (333, 200, 128), (65536, 320, 320), (40000, 408, 128)]:
        x = torch.randn(M, K, generator=gen, device=dev).half()
        w = (torch.randn(N, K, generator=gen, device=dev) * K ** -0.5).half()
        b = torch.randn(N, generator=gen, device=dev).half()
        r = torch.randn(M, N, generator=gen, device=dev).half()
        want = x.float() @ w.float().t() + b.float() + r.float()
        for v in PP:
            for s in (1, 2):
                y = F.linear(x, w, b, residual=r, variant=v, split_k=s)
                if "pp" not in L.last_kernel():
                    continue
                ok &= check(f"linear {M}x{N}x{K} v{v} s{s}", y, want)
    for (M, N, K) in [(2048, 1280, 320), (700, 2560, 640), (256, 128, 64), (8192, 2560, 640), (33000, 200, 128)]:
        x = torch.randn(M, K, generator=gen, device=dev).half()
        w = (torch.randn(2 * N, K, generator=gen, device=dev) * K ** -0.5).half()
        b = torch.randn(2 * N, generator=gen, device=dev).half()
        h = x.float() @ w.float().t() + b.float()
        want = h[:, :N] * torch.nn.functional.gelu(h[:, N:])
        for v in PP_GEGLU:
            y = F.linear(x, w, b, geglu=True, variant=v)
            if "pp" in L.last_kernel():
                ok &= check(f"geglu {M}x{N}x{K} v{v}", y, want)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    dev = "cuda"
    if not a.no_check:
        ok = correctness(dev)
        print("CORRECTNESS", "ok" if ok else "FAILED", flush=True)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    gen = torch.Generator(device=dev).manual_seed(1)
    if a.only in ("", "conv"):
        for label, B, Cin, hw, Cout, k in CONVS:
            x = cl(torch.randn(B, Cin, hw, hw, generator=gen, device=dev).half())
            w = cl((torch.randn(Cout, Cin, k, k, generator=gen, device=dev) * (k * k * Cin) ** -0.5).half())
            b = torch.randn(Cout, generator=gen, device=dev).half()
            pk = F.pack_weight(w)
            sweep(label, 2.0 * B * hw * hw * Cout * Cin * k * k, lambda v, s, p: F.conv2d(x, w, b, padding=k // 2, variant=v, split_k=s, w_packed=p), pk, OLD, a.quick)
    if a.only in ("", "gemm"):
        for label, M, N, K in GEMMS:
            x = torch.randn(M, K, generator=gen, device=dev).half()
            w = (torch.randn(N, K, generator=gen, device=dev) * K ** -0.5).half()
            b = torch.randn(N, generator=gen, device=dev).half()
            pk = F.pack_weight(w)
            sweep(label, 2.0 * M * N * K, lambda v, s, p: F.linear(x, w, b, variant=v, split_k=s, w_packed=p), pk, OLD, a.quick)
    if a.only in ("", "geglu"):
        for label, M, N, K in GEGLUS:
            x = torch.randn(M, K, generator=gen, device=dev).half()
            w = (torch.randn(2 * N, K, generator=gen, device=dev) * K ** -0.5).half()
            b = torch.randn(2 * N, generator=gen, device=dev).half()
            sweep(label, 2.0 * M * 2 * N * K, lambda v, s, p: F.linear(x, w, b, geglu=True, variant=v, split_k=s), None, OLD_GEGLU, a.quick)


if __name__ == "__main__":
    main()
