"""Quick parity pass over every tile of the wave-specialised LDS-DMA kernel (variants 21..26, linear / conv / GEGLU, 1-, 2- and
many-tile K loops, split-K, ragged M, residual, staged stores with GroupNorm statistics) against fp32 torch: a few seconds, run
before the full suite after a change of the K loop. Exits non-zero on the first mismatch."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-fast_amd"))
import torch  # noqa: E402
import torch.nn.functional as TF  # noqa: E402

from sfast.hip import functional as F  # noqa: E402
from sfast.hip import lib as L  # noqa: E402

torch.backends.cudnn.enabled = False
torch.manual_seed(0)
dev = "cuda"
bad = 0


def check(tag, y, ref, tol=3e-3):
    global bad
    e = float((y.float() - ref).norm() / ref.norm())
    ok = bool(torch.isfinite(y).all()) and e < tol
    if not ok:
        bad += 1
    print(f"{'ok ' if ok else 'BAD'} {tag:58s} {L.last_kernel():52s} rel_l2 {e:.2e}", flush=True)


for dt in (torch.float16, torch.bfloat16):
    tol = 3e-3 if dt == torch.float16 else 2e-2
    for v in (21, 22, 23, 24, 25, 26, 27):  # (27 = the 256x128 tile with 128x64 per consumer wave of run 22: skipped unless it is built in)
        for (M, K, N, s) in ((8192, 320, 320, 1), (2048, 64, 640, 1), (2048, 128, 640, 1), (130, 1280, 1280, 1), (512, 2560, 1280, 3), (512, 2560, 1280, 8),
                             (128, 5120, 1280, 20)):
            if dt == torch.bfloat16 and (M, K) not in ((8192, 320), (512, 2560)):
                continue
            x = torch.randn(M, K, device=dev).to(dt)
            w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
            b = torch.randn(N, device=dev).to(dt)
            r = torch.randn(M, N, device=dev).to(dt)
            try:
                y = F.linear(x, w, b, residual=r, variant=v, split_k=s)
            except L.SfastHipError as e:
                print("skip", v, M, K, N, s, str(e)[:60])
                continue
            check(f"{dt} linear v{v} M{M} K{K} N{N} split{s}", y, x.float() @ w.float().t() + b.float() + r.float(), tol)
        for (B, C1, C2, H, Co, s) in ((2, 320, 0, 64, 320, 1), (2, 64, 0, 32, 128, 1), (2, 1280, 0, 16, 1280, 6), (2, 640, 320, 32, 320, 2), (2, 1280, 1280, 8, 1280, 12)):
            if dt == torch.bfloat16 and C1 != 320:
                continue
            cl = torch.channels_last
            x = torch.randn(B, C1, H, H, device=dev).to(dt).contiguous(memory_format=cl)
            x2 = torch.randn(B, C2, H, H, device=dev).to(dt).contiguous(memory_format=cl) if C2 else None
            w = (torch.randn(Co, C1 + C2, 3, 3, device=dev) * ((C1 + C2) * 9) ** -0.5).to(dt).contiguous(memory_format=cl)
            b = torch.randn(Co, device=dev).to(dt)
            z = torch.randn(B, Co, H, H, device=dev).to(dt).contiguous(memory_format=cl)
            xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], 1)
            ref = TF.conv2d(xin, w.float(), b.float(), padding=1) + z.float()
            try:
                y = F.conv2d(x, w, b, z=z, padding=1, x2=x2, variant=v, split_k=s)
            except L.SfastHipError as e:
                print("skip conv", v, C1, C2, H, s, str(e)[:60])
                continue
            check(f"{dt} conv v{v} {C1}+{C2}->{Co}@{H} split{s}", y, ref, tol)
            if s == 1 and dt == torch.float16:
                try:
                    y2, stats, lay = F.conv2d(x, w, b, z=z, padding=1, x2=x2, variant=v, split_k=s, gn_unit=8)
                    check(f"{dt} conv+gnstats v{v} {C1}+{C2}->{Co}@{H}", y2, ref, tol)
                    assert torch.equal(y2, y), "staged stores changed the output"
                except L.SfastHipError:
                    pass
    for v in (21, 26):
        x = torch.randn(2048, 640, device=dev).to(dt)
        w = (torch.randn(2 * 2560, 640, device=dev) * 640 ** -0.5).to(dt)
        b = torch.randn(2 * 2560, device=dev).to(dt)
        y = F.linear(x, w, b, geglu=True, variant=v)
        full = x.float() @ w.float().t() + b.float()
        check(f"{dt} geglu v{v} M2048 K640 N2560", y, full[:, :2560] * TF.gelu(full[:, 2560:]), tol)
print("FAILED" if bad else "ALL OK", bad, flush=True)
if "--time" in sys.argv and not bad:
    stream = torch.cuda.Stream()

    def timed(fn, n=8):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=stream):
                for _ in range(n):
                    fn()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record(stream)
                g.replay()
                e1.record(stream)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n * 1e3)
        return best

    dt = torch.float16
    cl = torch.channels_last
    for (B, C1, C2, H, Co) in ((2, 320, 0, 64, 320), (2, 640, 320, 64, 320), (2, 640, 0, 32, 640), (2, 640, 0, 64, 640), (2, 1280, 0, 32, 1280)):
        x = torch.randn(B, C1, H, H, device=dev).to(dt).contiguous(memory_format=cl)
        x2 = torch.randn(B, C2, H, H, device=dev).to(dt).contiguous(memory_format=cl) if C2 else None
        w = (torch.randn(Co, C1 + C2, 3, 3, device=dev) * ((C1 + C2) * 9) ** -0.5).to(dt).contiguous(memory_format=cl)
        b = torch.randn(Co, device=dev).to(dt)
        line = []
        for v in (21, 22, 27):
            for sp in (1, 2, 3, 4, 6):
                try:
                    F.conv2d(x, w, b, padding=1, x2=x2, variant=v, split_k=sp)
                except L.SfastHipError:
                    continue
                k = L.last_kernel()
                if f"split={sp}," not in k:
                    continue
                line.append(f"v{v}/s{sp} {timed(lambda: F.conv2d(x, w, b, padding=1, x2=x2, variant=v, split_k=sp)):.1f}")
        print(f"conv {C1}+{C2}->{Co}@{H}: " + ", ".join(line), flush=True)
    for (M, K, N) in ((8192, 320, 1280), (8192, 1280, 320), (32768, 640, 640), (32768, 640, 2560)):
        x = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        b = torch.randn(N, device=dev).to(dt)
        line = []
        for v in (21, 22, 25, 27):
            for sp in (1, 2):
                try:
                    F.linear(x, w, b, variant=v, split_k=sp)
                except L.SfastHipError:
                    continue
                if f"split={sp}," not in L.last_kernel():
                    continue
                line.append(f"v{v}/s{sp} {timed(lambda: F.linear(x, w, b, variant=v, split_k=sp)):.1f}")
        print(f"linear M{M} K{K} N{N}: " + ", ".join(line), flush=True)
sys.exit(1 if bad else 0)
