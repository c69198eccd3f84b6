#!/usr/bin/env python3
"""Where do the ~0.3 ms between back-to-back replays of the step's hipGraph and bench.py's loop go? Times, with HIP events around 200
iterations each: (a) DenoiseLoop.step (2 small device-to-device copies + graph replay), (b) the loop's graph alone, (c) the copies
alone, (d) the UNet plan captured as a serial graph. usage: tools/replay_gap_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-fast_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

from sfast.engine import UNet2DEngine  # noqa: E402
from sfast.engine.denoise import DenoiseLoop  # noqa: E402
from sfast.engine.unet_spec import SD15_CONFIG, random_params  # noqa: E402

dev = torch.device("cuda", 0)
eng = UNet2DEngine(SD15_CONFIG, random_params(SD15_CONFIG, seed=0, dtype=torch.float16, device=dev))
loop = DenoiseLoop(eng, images=1, height=64, width=64, ctx_len=77, guidance=7.5, num_steps=50, use_graph=True)
g = torch.Generator(device=dev).manual_seed(1234)
loop.set_inputs(torch.randn(1, 4, 64, 64, generator=g, device=dev).half(), torch.randn(2, 77, 768, generator=g, device=dev).half())
loop.capture(warmups=3)
plan = loop.plan
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    serial = torch.cuda.CUDAGraph()
    with torch.cuda.graph(serial, stream=s):
        plan.run(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()


def timeit(fn, n=200):
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def copies(i):
    idx = i % 50
    plan.static_in["timestep"].copy_(loop.ts_table[idx], non_blocking=True)
    loop.coef.copy_(loop.coef_table[idx], non_blocking=True)


for rnd in range(2):
    print(f"round {rnd}: loop.step {timeit(loop.step):.3f} ms | loop graph only {timeit(lambda i: loop.graph.replay()):.3f} ms | "
          f"copies only {timeit(copies) * 1e3:.1f} us | plan-only serial graph {timeit(lambda i: serial.replay()):.3f} ms | "
          f"graph + 1 copy {timeit(lambda i: (loop.coef.copy_(loop.coef_table[i % 50], non_blocking=True), loop.graph.replay())):.3f} ms")
