#!/usr/bin/env python3
"""Launches for the operand-delivery PMC passes (tools/gpu_pmc_delivery.sh): the ring kernels and the packed-weight kernels on the
shapes DESIGN.md section 9 round 4 item 4b argues about, a few launches each, nothing else in the process."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402


def main():
    dev = "cuda"
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    gen = torch.Generator(device=dev).manual_seed(1)
    cases = []
    x = cl(torch.randn(2, 640, 32, 32, generator=gen, device=dev).half())
    w = cl((torch.randn(640, 640, 3, 3, generator=gen, device=dev) * 5760 ** -0.5).half())
    pk = F.pack_weight(w)
    cases += [("conv 640->640@32 ring 128x128 split 3", lambda: F.conv2d(x, w, None, padding=1, variant=21, split_k=3)),
              ("conv 640->640@32 pk 128x160 split 4", lambda: F.conv2d(x, w, None, padding=1, variant=44, split_k=4, w_packed=pk))]
    x2 = cl(torch.randn(2, 320, 64, 64, generator=gen, device=dev).half())
    w2 = cl((torch.randn(320, 320, 3, 3, generator=gen, device=dev) * 2880 ** -0.5).half())
    pk2 = F.pack_weight(w2)
    cases += [("conv 320->320@64 ring 64x64", lambda: F.conv2d(x2, w2, None, padding=1, variant=26, split_k=1)),
              ("conv 320->320@64 pk 64x160", lambda: F.conv2d(x2, w2, None, padding=1, variant=45, split_k=1, w_packed=pk2))]
    x3 = torch.randn(8192, 5120, generator=gen, device=dev).half()
    w3 = (torch.randn(1280, 5120, generator=gen, device=dev) * 5120 ** -0.5).half()
    pk3 = F.pack_weight(w3)
    cases += [("gemm 8192x1280x5120 ring 128x160", lambda: F.linear(x3, w3, None, variant=22, split_k=1)),
              ("gemm 8192x1280x5120 pk 128x256", lambda: F.linear(x3, w3, None, variant=41, split_k=1, w_packed=pk3))]
    for name, fn in cases:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        print(f"{name}: {L.last_kernel()}", flush=True)


if __name__ == "__main__":
    main()
