#!/usr/bin/env python3
"""Can HIP events time single kernels INSIDE a hipGraph replay? Captures the SD1.5 step with `external` timing events around the
D = 40 attention launches (event-record nodes in the graph), replays it back to back and prints what the events say, next to the
eager in-order replay bench.py used before. usage: tools/graph_event_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-fast_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from sfast.engine import UNet2DEngine  # noqa: E402
from sfast.engine.unet_spec import SD15_CONFIG, random_params  # noqa: E402
from sfast.hip import lib as L  # noqa: E402

dev = torch.device("cuda", 0)
eng = UNet2DEngine(SD15_CONFIG, random_params(SD15_CONFIG, seed=0, dtype=torch.float16, device=dev))
plan = eng.get_plan(2, 64, 64, 77)
s = torch.cuda.Stream()
names = []
with torch.cuda.stream(s):
    for op in plan.ops:
        op.launch(s.cuda_stream)
        names.append(L.last_kernel())
torch.cuda.synchronize()
idx = [i for i, n in enumerate(names) if n.startswith("attn_fwd[D=40")]
print("marked launches:", len(idx))
try:
    evs = {i: (torch.cuda.Event(enable_timing=True, external=True), torch.cuda.Event(enable_timing=True, external=True)) for i in idx}
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            sp = torch.cuda.current_stream().cuda_stream
            for i, op in enumerate(plan.ops):
                if i in evs:
                    evs[i][0].record(torch.cuda.current_stream())
                    op.launch(sp)
                    evs[i][1].record(torch.cuda.current_stream())
                else:
                    op.launch(sp)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t = [evs[i][0].elapsed_time(evs[i][1]) * 1e3 for i in idx]
    print("in-graph us:", [round(x, 1) for x in t], "mean", round(sum(t) / len(t), 2))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    print("graph with event nodes: ms/step", a.elapsed_time(b) / 50)
except Exception as e:  # noqa: BLE001
    print("in-graph event timing failed:", type(e).__name__, e)
with torch.cuda.stream(s):
    situ, ov = bench.in_situ_timing(plan, idx)
print("eager in-situ us:", [round(situ[i] * 1e6, 1) for i in idx], "overhead", round(ov * 1e6, 1))
