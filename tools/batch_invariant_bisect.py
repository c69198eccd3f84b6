#!/usr/bin/env python3
"""SFAST_BATCH_INVARIANT=1 debugging aid: run the SD1.5 plans of two batch sizes op by op (serial, one stream) on the SAME leading
samples and report the first ops after which sample 0's slice of the buffers the op wrote differs between the two plans, with the
kernel each plan launched for it. Buffers of a plan are batch-major, so sample 0 is the first 1/B of every pooled buffer.

    SFAST_BATCH_INVARIANT=1 python tools/batch_invariant_bisect.py [--a 1] [--b 2] [--model sd15] [--max 12]
"""
import argparse
import os
import sys

os.environ.setdefault("SFAST_BATCH_INVARIANT", "1")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "stable-fast_amd"))
import torch  # noqa: E402

from sfast.engine import UNet2DEngine  # noqa: E402
from sfast.engine import unet_spec as U  # noqa: E402
from sfast.hip import lib as L  # noqa: E402


def sig(t, B):
    n = t.numel() // B
    v = t[:n].view(torch.int16).to(torch.int64)
    return (int(v.sum()), int((v * v).sum()), n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", type=int, default=1)
    ap.add_argument("--b", type=int, default=2)
    ap.add_argument("--model", default="sd15")
    ap.add_argument("--max", type=int, default=12)
    a = ap.parse_args()
    dev = torch.device("cuda")
    cfg = {"sd15": U.SD15_CONFIG, "sdxl": U.SDXL_CONFIG}[a.model]
    eng = UNet2DEngine(cfg, U.random_params(cfg, seed=0, dtype=torch.float16, device=dev))
    g = torch.Generator(device="cpu").manual_seed(3)
    Bm = max(a.a, a.b)
    hw = cfg.get("sample_size", 64)
    sample = torch.randn(Bm, cfg["in_channels"], hw, hw, generator=g).half().to(dev)
    ehs = torch.randn(Bm, 77, cfg["cross_attention_dim"], generator=g).half().to(dev)
    plans = {}
    for B in (a.a, a.b):
        p = eng.get_plan(B, hw, hw, 77)
        eng.load_inputs(p, sample[:B], 981, ehs[:B])
        plans[B] = p
    pa, pb = plans[a.a], plans[a.b]
    print(f"ops: B={a.a}: {len(pa.ops)}  B={a.b}: {len(pb.ops)}", flush=True)
    na, nb = [o.name for o in pa.ops], [o.name for o in pb.ops]
    if na != nb:
        print("OP LISTS DIFFER:")
        for x in sorted(set(na) ^ set(nb)):
            print("   ", x, "(only B=%d)" % (a.a if x in na else a.b))
    s = torch.cuda.current_stream(dev).cuda_stream

    def snapshot(p, B):
        return [sig(t, B) for t in p.pool.all]

    common = [n for n in na if n in set(nb)]
    ia = {o.name: o for o in pa.ops}
    ib = {o.name: o for o in pb.ops}
    # run both plans in their own order, but compare by op name: after each op of plan a, the multiset of sample-0 signatures of the
    # buffers that CHANGED; then the same for plan b
    def trace(p, B):
        out = {}
        prev = snapshot(p, B)
        for op in p.ops:
            op.launch(s)
            torch.cuda.synchronize()
            k = L.last_kernel()
            cur = snapshot(p, B)
            changed = sorted(c for c, q in zip(cur, prev) if c != q)
            data = [t[: t.numel() // B].clone() for t, c, q in zip(p.pool.all, cur, prev) if c != q]
            out[op.name] = (k, changed, op, data)
            prev = cur
        return out
    ta, tb = trace(pa, a.a), trace(pb, a.b)
    shown = 0
    for n in common:
        ka, ca, oa, da = ta[n]
        kb, cb, ob, db = tb[n]
        same = ca == cb
        va = (int(oa.tune[0].variant), int(oa.tune[0].split_k)) if oa.tune else None
        vb = (int(ob.tune[0].variant), int(ob.tune[0].split_k)) if ob.tune else None
        if not same or va != vb:
            print(f"{'DIFF' if not same else 'same'} {n:60s} kind={oa.kind}\n      B={a.a}: {va} [{ka}] wrote {len(ca)}\n      B={a.b}: {vb} [{kb}] wrote {len(cb)}", flush=True)
            if not same and len(da) == 1 and len(db) == 1 and da[0].numel() == db[0].numel():
                ne = (da[0] != db[0]).nonzero().flatten()
                width = int(getattr(oa.tune[0], "N", 0) or getattr(oa.tune[0], "Cout", 0)) if oa.tune else 0
                print(f"      {ne.numel()} of {da[0].numel()} elements differ; first flat indices {ne[:8].tolist()} last {ne[-3:].tolist()}"
                      f" max|d| {(da[0].float() - db[0].float()).abs().max().item():.3e}" + (f" (row width {width})" if width else ""), flush=True)
            shown += 1
            if shown >= a.max:
                break
    # statistics hand-over: layouts and sample-0 records of every fused GroupNorm
    ca = {c["name"]: c for c in pa.gn_candidates}
    cb = {c["name"]: c for c in pb.gn_candidates}
    nshow = 0
    for n in ca:
        if n not in cb:
            continue
        x, y = ca[n], cb[n]
        fa, fb = x["pre"][0] is not None, y["pre"][0] is not None
        line = f"gn {n:50s} fused B={a.a}:{fa} B={a.b}:{fb}"
        bad = fa != fb
        if fa and fb:
            for key, Ba, Bb in (("xw", a.a, a.b), ("x2w", a.a, a.b)):
                wa, wb = x.get(key), y.get(key)
                if wa is None or wb is None or wa["stats"][0] is None or wb["stats"][0] is None:
                    continue
                la, lb = x["pre"][0][1 if key == "xw" else 3], y["pre"][0][1 if key == "xw" else 3]
                ga = tuple(int(getattr(la, f)) for f in ("rb_rows", "n_rb", "bno", "tiles_n", "slots", "unit"))
                gb = tuple(int(getattr(lb, f)) for f in ("rb_rows", "n_rb", "bno", "tiles_n", "slots", "unit"))
                sa, sb = wa["stats"][0], wb["stats"][0]
                na_, nb_ = la.nbytes() // 4 // Ba, lb.nbytes() // 4 // Bb
                eq = na_ == nb_ and torch.equal(sa[:na_], sb[:nb_])
                line += f"\n      {key}: producer {wa['name']} layout {ga} vs {gb} records {'equal' if eq else 'DIFFER'}"
                bad |= not eq
        if bad and nshow < a.max:
            print(line, flush=True)
            nshow += 1
    print("done; differing/unequal-choice ops shown:", shown)


if __name__ == "__main__":
    main()
