#!/usr/bin/env python3
"""What does ONE node of a captured hipGraph cost on this box? Chains of N dependent tiny kernels, replayed and timed with events:
   (a) the library's cfg_ddim kernel on 16 K elements (64 workgroups), (b) its LayerNorm on [128, 1280], (c) alternating (a)/(b)
   (different code objects back to back), (d) a torch elementwise kernel, (e) cfg_ddim on 4 M elements (dirty-line write-back),
   (f) two independent chains on two captured branches (does the graph overlap them?).
Prints us per node. Context for the per-step dispatch count: node cost x nodes = the floor of a step."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-fast_amd"))
import torch  # noqa: E402

from sfast.hip import lib as L  # noqa: E402

dev = torch.device("cuda")
lib = L.init_device(dev)
N = 400


def timed(build, reps=20, two_branch=False):
    s = torch.cuda.Stream()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        build(s.cuda_stream, 4)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            cur = torch.cuda.current_stream()
            if two_branch:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    build(side.cuda_stream, N // 2, 1)
                build(cur.cuda_stream, N // 2, 0)
                cur.wait_stream(side)
            else:
                build(cur.cuda_stream, N)
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(reps):
            g.replay()
        b.record(s)
        b.synchronize()
    return a.elapsed_time(b) / reps / N * 1e3


def main():
    n_small, n_big = 16384, 4 << 20
    bufs = {}
    for tag, n in (("s", n_small), ("b", n_big)):
        bufs[tag] = [(torch.randn(2 * n, device=dev).half(), torch.randn(n, device=dev).half(), torch.empty(n, device=dev).half()) for _ in range(2)]
    coef = torch.tensor([0.9, 0.4, 0.95, 0.3], device=dev)
    x = [torch.randn(128, 1280, device=dev).half() for _ in range(2)]
    y = [torch.empty_like(x[0]) for _ in range(2)]
    gam, bet = torch.ones(1280, device=dev).half(), torch.zeros(1280, device=dev).half()
    lp = L.LnParams(L.F16, 128, 1280, 1e-5)

    def ddim(tag):
        def build(st, n, br=0):
            e, l, o = bufs[tag][br]
            for _ in range(n):
                lib.sfast_hip_cfg_ddim_step(e.data_ptr(), l.data_ptr(), o.data_ptr(), None, coef.data_ptr(), C.c_float(7.5), l.numel(), L.F16, st)
        return build

    def ln(st, n, br=0):
        for _ in range(n):
            lib.sfast_hip_layer_norm(x[br].data_ptr(), gam.data_ptr(), bet.data_ptr(), y[br].data_ptr(), C.byref(lp), st)

    def alt(st, n, br=0):
        e, l, o = bufs["s"][br]
        for i in range(n):
            if i & 1:
                lib.sfast_hip_layer_norm(x[br].data_ptr(), gam.data_ptr(), bet.data_ptr(), y[br].data_ptr(), C.byref(lp), st)
            else:
                lib.sfast_hip_cfg_ddim_step(e.data_ptr(), l.data_ptr(), o.data_ptr(), None, coef.data_ptr(), C.c_float(7.5), l.numel(), L.F16, st)

    t = [torch.zeros(16384, device=dev) for _ in range(2)]

    def tch(st, n, br=0):
        for _ in range(n):
            t[br].add_(1.0)

    print(f"cfg_ddim 16K elements      : {timed(ddim('s')):6.2f} us / node")
    print(f"layer_norm [128,1280]      : {timed(ln):6.2f} us / node")
    print(f"alternating ddim / ln      : {timed(alt):6.2f} us / node")
    print(f"torch add_ 16K             : {timed(tch):6.2f} us / node")
    print(f"cfg_ddim 4M elements       : {timed(ddim('b')):6.2f} us / node   (8 MB written per node)")
    print(f"two branches, ddim 16K     : {timed(ddim('s'), two_branch=True):6.2f} us / node   (ideal overlap = half the single-chain figure)")
    print(f"two branches, ln           : {timed(ln, two_branch=True):6.2f} us / node")


if __name__ == "__main__":
    main()
