#!/usr/bin/env python3
"""Sweep tile variant x split-K for every distinct GEMM / conv problem of a UNet plan on the GPU.

    python tools/tune_igemm.py [--config sd15] [--batch 2] > gpurun_out/tune.json

For each problem prints the planner's own choice and the measured time of every forced
(variant, split) so the analytic cost model in csrc/igemm.hip can be calibrated from one GPU run.
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-fast_amd"))

os.environ.setdefault("SFAST_AUTOTUNE", "0")
import torch  # noqa: E402

from sfast.engine import UNet2DEngine  # noqa: E402
from sfast.engine.unet_spec import SD15_CONFIG, SDXL_CONFIG, random_params  # noqa: E402
from sfast.hip import lib as L  # noqa: E402


def time_launch(fn, reps=5):
    st = torch.cuda.current_stream()
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        fn()
        b.record(st)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="sd15")
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    cfg = SD15_CONFIG if a.config == "sd15" else SDXL_CONFIG
    dev = torch.device("cuda")
    eng = UNet2DEngine(cfg, random_params(cfg, device=dev))
    hw = cfg["sample_size"]
    plan = eng.build_plan(a.batch, hw, hw, 77)
    lib = L.load()
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    sp = torch.cuda.current_stream().cuda_stream
    seen = {}
    out = []
    for op in plan.ops:
        if op.tune is None or op.kind in ("temb",):
            continue
        p, launch_with = op.tune
        if isinstance(p, L.GemmParams):
            if p.M <= 16:
                continue
            key = ("gemm", p.M, p.N, p.K, p.geglu)
            M, N, K, geglu = p.M, p.N, p.K, p.geglu
        else:
            if p.Cout < 16 or p.Cin < 8:
                continue
            Hin = 2 * p.H if p.upsample2x else p.H
            Ho = (Hin + 2 * p.pad_h - (p.KH - 1) - 1) // p.stride_h + 1
            M, N, K, geglu = p.B * Ho * Ho, p.Cout, p.KH * p.KW * p.Cin, 0
            key = ("conv", M, N, K, p.KH, p.stride_h, p.upsample2x, p.C1 != p.Cin)
        if key in seen:
            seen[key]["count"] += 1
            continue
        rec = dict(key=list(key), name=op.name, kind=op.kind, flops=op.flops, count=1, results=[])
        seen[key] = rec
        out.append(rec)
        o = (C.c_int32 * 5)()
        lib.sfast_hip_igemm_plan(M, N, K, geglu, 0, 0, C.byref(o))
        rec["auto"] = list(o)
        p.variant, p.split_k = 0, 0
        rec["auto_us"] = time_launch(lambda: launch_with(sp, ws.data_ptr(), ws.numel()))
        variants = [1, 3, 11, 13] if geglu else [1, 2, 3, 5, 11, 12, 13, 15]
        ktiles = (K + 63) // 64
        for v in variants:
            for s in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
                if s > 1 and (ktiles // s) < 2:
                    continue
                lib.sfast_hip_igemm_plan(M, N, K, geglu, v, s, C.byref(o))
                bm, bn, splits, ktps, vid = list(o)
                if vid != v:
                    continue
                if splits != s:
                    continue
                bno = bn // 2 if geglu else bn
                tiles = -(-M // bm) * -(-N // bno)
                if tiles * splits > 8192 or (s > 1 and tiles * splits > 2048):
                    continue
                p.variant, p.split_k = v, s
                rc = launch_with(sp, ws.data_ptr(), ws.numel())
                if rc != 0:
                    continue
                if (v >= 10) != ("dma" in L.last_kernel()):
                    continue  # the library substituted another pipe (problem not eligible)
                us = time_launch(lambda: launch_with(sp, ws.data_ptr(), ws.numel()))
                rec["results"].append(dict(variant=v, bm=bm, bn=bn, split=s, us=us, tflops=op.flops / us / 1e6))
        p.variant, p.split_k = 0, 0
        best = min(rec["results"], key=lambda r: r["us"]) if rec["results"] else None
        rec["best"] = best
        print(f"{op.kind:8s} {str(key):60s} auto {rec['auto']} {rec['auto_us']:8.1f}us | best v{best['variant']} s{best['split']} "
              f"{best['us']:8.1f}us {best['tflops']:7.1f} TF x{rec['count']}", file=sys.stderr)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
