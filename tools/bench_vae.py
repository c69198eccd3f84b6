#!/usr/bin/env python3
"""VAE decode latency: SD VAE decoder, 64x64 latent -> 512x512 image, batch 1, fp16.
Native engine (eager plan and hipGraph replay) beside the same restatement run eagerly by PyTorch-ROCm (MIOpen /
hipBLASLt), seeded random-init weights. Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-fast_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import vae_ref as V  # noqa: E402  (the eager PyTorch stand-in; not the product path)
from sfast.engine import VaeDecoderEngine, capture_plan_graph  # noqa: E402


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


def main():
    hw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda")
    m = V.build("sd", seed=0, dtype=torch.float16, device=dev)
    eager = V.build("sd", seed=0, dtype=torch.float16, device=dev).to(memory_format=torch.channels_last)
    z = torch.randn(1, 4, hw, hw, generator=torch.Generator().manual_seed(0)).to(dev, torch.float16)
    eng = VaeDecoderEngine.from_module(m)
    plan = eng.get_plan(1, hw, hw)
    eng.load_inputs(plan, z)
    y = eng.forward(z)
    with torch.no_grad():
        want = eager(z)
    err = float((y.float() - want.float()).norm() / want.float().norm())
    s = torch.cuda.Stream()
    graph, _ = capture_plan_graph(plan, s)
    with torch.cuda.stream(s):
        t_graph = timed(graph.replay)
    t_plan = timed(lambda: plan.run(torch.cuda.current_stream().cuda_stream))
    with torch.no_grad():
        t_eager = timed(lambda: eager(z), n=10)
    # per-op timing (HIP events on the launch stream, 3-launch bursts), aggregated by family and by kernel variant
    from sfast.hip import lib as L
    st = torch.cuda.current_stream()
    fam, kern = {}, {}
    for op in plan.ops:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        op.launch(st.cuda_stream)
        a.record(st)
        for _ in range(3):
            op.launch(st.cuda_stream)
        b.record(st)
        b.synchronize()
        t = a.elapsed_time(b) / 3
        f = fam.setdefault(op.kind, [0, 0.0, 0.0])
        f[0] += 1
        f[1] += t
        f[2] += op.flops
        k = kern.setdefault(L.last_kernel(), [0, 0.0, 0.0])
        k[0] += 1
        k[1] += t
        k[2] += op.flops
    inv = plan.summary()
    gflop = sum(v["gflop"] for v in inv.values())
    print(json.dumps({"metric": f"SD VAE decode {hw}x{hw} latent -> {8 * hw}x{8 * hw} image, bs=1 fp16", "unit": "ms",
                      "native_graph_ms": t_graph, "native_eager_plan_ms": t_plan, "pytorch_rocm_eager_ms": t_eager,
                      "speedup_vs_pytorch_eager": t_eager / t_graph, "rel_l2_vs_pytorch_eager_fp16": err,
                      "gflop": gflop, "tflops": gflop / t_graph, "launches": len(plan.ops),
                      "families_ms": {k: dict(n=v[0], ms=round(v[1], 3), tflops=round(v[2] / v[1] / 1e9, 1) if v[2] else None)
                                      for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])},
                      "kernels_ms": {k: dict(n=v[0], ms=round(v[1], 3), tflops=round(v[2] / v[1] / 1e9, 1) if v[2] else None)
                                     for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])[:14]}}))


if __name__ == "__main__":
    main()
