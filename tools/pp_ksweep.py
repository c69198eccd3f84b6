#!/usr/bin/env python3
"""Fixed per-tile cost of the pipe-5 kernels: the same M x N problem timed at several K (time = a + b K per launch; a = prologue +
epilogue that no main loop hides when one workgroup owns a CU). GEGLU 8192 x K -> 2560 and linear 65536 x K -> 320.

    python tools/pp_ksweep.py

Round 6 result (profiles/r06_pp_ksweep_run28.log; _run30 / _run31 with a PERSISTENT form of the lockstep kernel, variant ids 61 / 62, that
streamed the next tile's K-tiles under the epilogue and is not in the tree): 5 - 10 us of fixed cost per tile round, marginal rates of
1040 - 1050 TF/s. The persistent form did not move the fixed part (GEGLU 8192 x 640 -> 2560: 74.1 vs 77.2 us; staged linears were slower,
the two-pass 128-row staging in one ring stage cost more than the hidden prologue returned): the fixed part is the epilogue's own
execution time in the consumer waves, not an exposed prologue or the store drain -- which the hardware already overlaps with the next
workgroup's prologue.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402
from pp_ab import timed  # noqa: E402


def main():
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(1)
    for label, M, N, geglu, variants in (("geglu 8192 x K -> 2560", 8192, 2560, True, (57, 53, 16)),
                                         ("geglu 65536 x K -> 1280", 65536, 1280, True, (57, 53, 16)),
                                         ("linear 65536 x K -> 320", 65536, 320, False, (58, 52, 12)),
                                         ("linear 16384 x K -> 640", 16384, 640, False, (58, 52, 12))):
        for v in variants:
            row = []
            for K in (320, 640, 1280, 2560):
                x = torch.randn(M, K, generator=gen, device=dev).half()
                w = (torch.randn(2 * N if geglu else N, K, generator=gen, device=dev) * K ** -0.5).half()
                b = torch.randn(2 * N if geglu else N, generator=gen, device=dev).half()
                t, name = timed(lambda: F.linear(x, w, b, geglu=geglu, variant=v, split_k=1))
                if t is None:
                    continue
                flops = 2.0 * M * (2 * N if geglu else N) * K
                row.append((K, t, flops / t / 1e6, name))
            if len(row) >= 2:
                (k0, t0, _, _), (k1, t1, _, _) = row[0], row[-1]
                slope = (t1 - t0) / (k1 - k0)
                fixed = t0 - slope * k0
                print(f"{label:26s} v{v:<3d} " + "  ".join(f"K={k}: {t:6.1f} us {tf:5.0f} TF/s" for k, t, tf, _ in row)
                      + f"   fixed {fixed:5.1f} us + {slope * 64:5.2f} us per K-tile  [{row[0][3]}]", flush=True)


if __name__ == "__main__":
    main()
