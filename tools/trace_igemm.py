#!/usr/bin/env python3
"""Per-workgroup phase timeline of the MFMA GEMM kernels (sfast_hip_set_trace): where a workgroup's time goes.

Slots (100 MHz wall clock): 0 entry, 1 metadata done, 2 prologue loads issued, 3 first K-tile usable,
4 K loop done, 5 epilogue issued, 6 stores drained, 7 HW_ID | XCC_ID << 32.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-fast_amd"))
os.environ.setdefault("SFAST_AUTOTUNE", "0")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probes  # noqa: E402

_probes.use_probe_build()  # experiment / ablation / patch-pipe instantiations exist only in libsfast_hip_probes.so
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sfast.engine import UNet2DEngine  # noqa: E402
from sfast.engine.unet_spec import SD15_CONFIG, random_params  # noqa: E402
from sfast.hip import lib as L  # noqa: E402

# (op, variant, split): each also traced under the kernel's experiment instantiations (SFAST_IGEMM_EXP)
EXPERIMENTS = [
    ("down_blocks.0.resnets.0.conv2", 12, 2),
    ("down_blocks.0.attentions.0.transformer_blocks.0.ff.out", 18, 1),
]
if "--ws" in sys.argv:  # the wave-specialised kernel's experiment instantiations (igemm_glds_ws.hip): shader clock under each ablation
    EXPERIMENTS = [("down_blocks.0.resnets.0.conv2", 21, 1), ("up_blocks.3.resnets.0.conv1", 22, 2)]
EXP_NAMES = {0: "full", 1: "no MFMA", 2: "no LDS reads", 3: "no MFMA, no LDS reads", 4: "no in-loop DMA", 8: "no barrier",
             7: "loop skeleton (wait + barrier only)"}

PROBES = [
    ("down_blocks.0.attentions.0.transformer_blocks.0.ff.geglu", [(16, 1), (21, 1), (23, 1), (1, 1)]),
    ("down_blocks.0.attentions.0.transformer_blocks.0.ff.out", [(18, 1), (23, 1), (3, 1)]),
    ("down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_qkv", [(18, 1), (23, 1), (3, 1)]),
    ("down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out", [(13, 1), (23, 1), (3, 1)]),
    ("down_blocks.0.resnets.0.conv2", [(12, 2), (22, 2), (23, 1), (18, 1), (2, 4)]),
    ("up_blocks.3.resnets.0.conv1", [(15, 4), (22, 4), (22, 2), (2, 4)]),
    ("down_blocks.2.resnets.1.conv2", [(15, 12), (22, 12), (21, 12), (5, 12)]),
]


def pct(a, q):
    return float(np.percentile(a, q)) / 100.0  # ticks (10 ns) -> us


def main():
    dev = torch.device("cuda")
    eng = UNet2DEngine(SD15_CONFIG, random_params(SD15_CONFIG, device=dev))
    plan = eng.build_plan(2, 64, 64, 77)
    lib = L.load()
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    trace = torch.zeros(16 * 65536, dtype=torch.int64, device=dev)
    sp = torch.cuda.current_stream().cuda_stream
    ops = {op.name: op for op in plan.ops}
    names = ["setup", "issue", "1st-tile", "k-loop", "epi", "drain", "total"]
    for name, v, sk in EXPERIMENTS:
        op = ops[name]
        p, launch_with = op.tune
        p.variant, p.split_k = v, sk
        for ex in (0, 1, 2, 3, 4, 8, 7):
            if "--ws" in sys.argv and ex == 8 and False:
                continue
            os.environ["SFAST_IGEMM_EXP"] = str(ex)
            trace.zero_()
            lib.sfast_hip_set_trace(trace.data_ptr())
            for _ in range(2):
                assert launch_with(sp, ws.data_ptr(), ws.numel()) == 0
            torch.cuda.synchronize()
            kern = L.last_kernel()
            lib.sfast_hip_set_trace(None)
            t = trace.cpu().numpy().reshape(-1, 16)
            t = t[t[:, 0] != 0]
            st = t[:, :7].astype(np.int64)
            kl = st[:, 4] - st[:, 3]
            tot = st[:, 6] - st[:, 0]
            wall = tot.astype(np.float64)
            cyc = (t[:, 9] - t[:, 8]).astype(np.float64)
            ok = wall > 50
            mhz = float(np.median(cyc[ok] / wall[ok]) * 100.0) if ok.any() else 0.0
            print(f"EXP {name[-28:]:28s} {kern:40s} {EXP_NAMES[ex]:36s} k-loop med {pct(kl, 50):6.2f} us  WG total {pct(tot, 50):6.2f} us  "
                  f"shader clock {mhz:5.0f} MHz", flush=True)
        os.environ.pop("SFAST_IGEMM_EXP", None)
        p.variant, p.split_k = 0, 0
    for name, cfgs in ([] if "--ws" in sys.argv else PROBES):
        op = ops[name]
        p, launch_with = op.tune
        for v, s in cfgs:
            p.variant, p.split_k = v, s
            for _ in range(3):
                assert launch_with(sp, ws.data_ptr(), ws.numel()) == 0
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                launch_with(sp, ws.data_ptr(), ws.numel())
            b.record()
            b.synchronize()
            t_us = a.elapsed_time(b) * 100
            kern = L.last_kernel()
            trace.zero_()
            lib.sfast_hip_set_trace(trace.data_ptr())
            launch_with(sp, ws.data_ptr(), ws.numel())
            torch.cuda.synchronize()
            lib.sfast_hip_set_trace(None)
            t = trace.cpu().numpy().reshape(-1, 16)
            t = t[t[:, 0] != 0]
            n = len(t)
            st = t[:, :7].astype(np.int64)
            t0 = st[:, 0].min()
            span = (st[:, 6].max() - t0) / 100.0
            d = np.stack([st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 4] - st[:, 3],
                          st[:, 5] - st[:, 4], st[:, 6] - st[:, 5], st[:, 6] - st[:, 0]], 1)
            hw = t[:, 7].astype(np.uint64)
            cu = ((hw >> np.uint64(32)) & np.uint64(0xF)) * np.uint64(256) + ((hw >> np.uint64(8)) & np.uint64(0xFF))
            ncu = len(np.unique(cu))
            # peak concurrent workgroups on one CU
            peak = 0
            for c in np.unique(cu)[:32]:
                m = cu == c
                ev = sorted([(x, 1) for x in st[m, 0]] + [(x, -1) for x in st[m, 6]])
                cur = 0
                for _, dlt in ev:
                    cur += dlt
                    peak = max(peak, cur)
            start = st[:, 0] - t0
            print(f"{name[-34:]:34s} {kern:42s} {t_us:6.1f} us/launch | traced span {span:6.1f} us, {n} WGs on {ncu} CUs, "
                  f"peak {peak} WG/CU")
            print("      phase med/p90 us: " + "  ".join(
                f"{nm} {pct(d[:, i], 50):.2f}/{pct(d[:, i], 90):.2f}" for i, nm in enumerate(names)))
            wall = (st[:, 6] - st[:, 0]).astype(np.float64)  # 10 ns ticks
            cyc = (t[:, 9] - t[:, 8]).astype(np.float64)
            ok = wall > 50
            mhz = float(np.median(cyc[ok] / wall[ok]) * 100.0) if ok.any() else 0.0
            print(f"      clock64 ticks per us over a workgroup's life (median): {mhz:.0f}")
            print(f"      WG start offsets us: p10 {pct(start, 10):.2f} p50 {pct(start, 50):.2f} p90 {pct(start, 90):.2f} "
                  f"max {pct(start, 100):.2f}; first-round WGs (start < 1 us): {(start < 100).sum()}")
        p.variant, p.split_k = 0, 0


if __name__ == "__main__":
    main()
