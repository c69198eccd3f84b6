#!/usr/bin/env python3
"""Experiment: CFG batch-2 UNet as ONE batch-2 plan vs TWO batch-1 plans on two graph branches (kernels are
latency-bound; do two concurrent chains hide each other's per-kernel floor?)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-fast_amd"))
import torch  # noqa: E402

from sfast.engine import UNet2DEngine  # noqa: E402
from sfast.engine.unet_spec import SD15_CONFIG, random_params  # noqa: E402


def timed(g, stream, n=30):
    with torch.cuda.stream(stream):
        for _ in range(5):
            g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(n):
            g.replay()
        b.record(stream)
        b.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = torch.device("cuda")
    eng = UNet2DEngine(SD15_CONFIG, random_params(SD15_CONFIG, device=dev))
    p2 = eng.build_plan(2, 64, 64, 77)
    pa = eng.build_plan(1, 64, 64, 77)
    pb = eng.build_plan(1, 64, 64, 77)
    for p in (p2, pa, pb):
        for t in p.static_in.values():
            if t.dtype == torch.float16:
                t.normal_()
        p.static_in["timestep"].fill_(500.0)
    s = torch.cuda.Stream()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        for p in (p2, pa, pb):
            p.run(s.cuda_stream)
    torch.cuda.synchronize()

    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g2, stream=s):
            p2.run(torch.cuda.current_stream().cuda_stream)
    gs = torch.cuda.CUDAGraph()  # two batch-1 plans back to back on one branch
    with torch.cuda.stream(s):
        with torch.cuda.graph(gs, stream=s):
            cur = torch.cuda.current_stream()
            pa.run(cur.cuda_stream)
            pb.run(cur.cuda_stream)
    gd = torch.cuda.CUDAGraph()  # two batch-1 plans on two branches
    with torch.cuda.stream(s):
        with torch.cuda.graph(gd, stream=s):
            cur = torch.cuda.current_stream()
            e0 = torch.cuda.Event()
            e0.record(cur)
            s1.wait_event(e0)
            s2.wait_event(e0)
            with torch.cuda.stream(s1):
                pa.run(s1.cuda_stream)
                e1 = torch.cuda.Event()
                e1.record(s1)
            with torch.cuda.stream(s2):
                pb.run(s2.cuda_stream)
                e2 = torch.cuda.Event()
                e2.record(s2)
            cur.wait_event(e1)
            cur.wait_event(e2)
    torch.cuda.synchronize()
    print(f"batch-2 plan, one branch        : {timed(g2, s):.3f} ms")
    print(f"two batch-1 plans, one branch   : {timed(gs, s):.3f} ms")
    print(f"two batch-1 plans, two branches : {timed(gd, s):.3f} ms")


if __name__ == "__main__":
    main()
