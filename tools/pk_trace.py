#!/usr/bin/env python3
"""Per-workgroup phase times of a pipe-4 launch (sfast_hip_set_trace): slots 0 entry, 1 prologue requests issued, 2/3 first barrier
passed, 4 K loop done, then epilogue / exit as igemm_device.h's trace_finish writes them."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402


def trace(label, call, nblocks):
    lib = L.load()
    buf = torch.zeros(nblocks * 16, dtype=torch.int64, device="cuda")
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    lib.sfast_hip_set_trace(C.c_void_p(buf.data_ptr()))
    call()
    torch.cuda.synchronize()
    lib.sfast_hip_set_trace(None)
    name = L.last_kernel()
    t = buf.cpu().numpy().reshape(-1, 16).astype(np.float64)
    t = t[t[:, 0] > 0]
    base = t[:, 0].min()
    us = lambda a: a / 100.0  # 100 MHz wall clock
    ph = [("issue", 0, 1), ("1st barrier", 1, 3), ("k loop", 3, 4), ("epilogue", 4, 5)]
    print(f"{label}: {name}: {len(t)} workgroups; first entry -> last exit {us(t[:, 5].max() - base):.1f} us; entry spread {us(t[:, 0].max() - base):.1f} us")
    print("   phase (us) median / p90 / max: " + "  ".join(
        f"{n} {np.median(us(t[:, b] - t[:, a])):.2f}/{np.percentile(us(t[:, b] - t[:, a]), 90):.2f}/{us(t[:, b] - t[:, a]).max():.2f}" for n, a, b in ph))
    print(f"   per-workgroup total median {np.median(us(t[:, 5] - t[:, 0])):.2f} us", flush=True)


def main():
    dev = "cuda"
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    gen = torch.Generator(device=dev).manual_seed(1)
    x = cl(torch.randn(2, 320, 64, 64, generator=gen, device=dev).half())
    w = cl((torch.randn(320, 320, 3, 3, generator=gen, device=dev) * 2880 ** -0.5).half())
    pk = F.pack_weight(w)
    for v in (26, 45, 44, 43):
        trace(f"conv 320->320@64 v{v}", lambda: F.conv2d(x, w, None, padding=1, variant=v, split_k=1, w_packed=pk if v >= 40 else None), 4096)
    x = cl(torch.randn(2, 640, 32, 32, generator=gen, device=dev).half())
    w = cl((torch.randn(640, 640, 3, 3, generator=gen, device=dev) * 5760 ** -0.5).half())
    pk = F.pack_weight(w)
    for v, s in ((21, 3), (43, 4), (45, 2), (41, 6)):
        trace(f"conv 640->640@32 v{v} split{s}", lambda: F.conv2d(x, w, None, padding=1, variant=v, split_k=s, w_packed=pk if v >= 40 else None), 4096)


if __name__ == "__main__":
    main()
