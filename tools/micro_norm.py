"""Where do the normalisation launches of the SD1.5 step spend their time? (round 5)

Per shape: the one-pass GroupNorm apply over producer-emitted statistics (what the plan runs behind a conv), the two-launch
GroupNorm (statistics + apply), LayerNorm over the same bytes, and a plain copy of the same bytes as the floor of a dependent
read-modify-write launch. Every candidate is captured 40 times back to back into ONE hipGraph (same stream: each launch waits for its
predecessor, as in the step) and the graph is replayed between HIP events: microseconds per launch, no host time inside.
`SFAST_GN_APPLY_WGS`, `SFAST_GN_FINAL` ... are read by the library at load: run the script once per setting.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stable-fast_amd")]

import torch  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402

DEV = "cuda"
REP = 40


def timed(fn, label, nbytes):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP):
                fn()
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            g.replay()
            b.record(s)
            b.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / REP)
    us = sorted(ts)[len(ts) // 2]
    rec = dict(op=label, us=round(us, 2), gbs=round(nbytes / us / 1e3, 1), kernel=L.last_kernel())
    print(json.dumps(rec), flush=True)
    del g
    return us


def main():
    torch.manual_seed(0)
    cases = [(2, 320, 64, 64, 32), (2, 640, 32, 32, 32), (2, 1280, 16, 16, 32), (2, 1280, 8, 8, 32), (2, 960, 64, 64, 32), (16, 320, 64, 64, 32)]
    for (B, C, H, W, G) in cases:
        x = torch.randn(B, C, H, W, device=DEV, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        gw = torch.randn(C, device=DEV, dtype=torch.float16)
        gb = torch.randn(C, device=DEV, dtype=torch.float16)
        nbytes = 2 * x.numel() * 2
        print(json.dumps(dict(case=[B, C, H, W], mbytes=nbytes / 1e6)), flush=True)
        # producer with statistics: a 1x1 conv C -> C (the cheapest producer that emits the same record layout family)
        w = (torch.randn(C, C, 1, 1, device=DEV, dtype=torch.float16) * C ** -0.5).contiguous(memory_format=torch.channels_last)
        try:
            y, stats, lay = F.conv2d(x, w, None, gn_unit=C // G)
            timed(lambda: F.group_norm_apply(y, G, gw, gb, 1e-5, "silu", stats, lay), "gn_apply(pre)+silu", nbytes)
        except L.SfastHipError as e:
            print(json.dumps(dict(op="gn_apply(pre)", error=str(e)[:200])), flush=True)
            y = x
        timed(lambda: F.group_norm(y, G, gw, gb, 1e-5, "silu"), "group_norm 2-launch +silu", nbytes)
        dst = torch.empty_like(y)
        timed(lambda: F.strided_copy(y, dst), "copy", nbytes)
        rows = y.permute(0, 2, 3, 1).reshape(B * H * W, C)
        timed(lambda: F.layer_norm(rows, (C,), gw, gb), "layer_norm rows", nbytes)
        timed(lambda: dst.copy_(y), "torch copy_", nbytes)


if __name__ == "__main__":
    main()
