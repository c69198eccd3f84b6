"""Shared helpers for the GPU parity tests: error metrics + a persistent log under gpurun_out/ so one
GPU run leaves the measured error of every case behind (pass or fail)."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG_DIR = os.path.join(ROOT, "gpurun_out")
LOG = os.path.join(LOG_DIR, "parity.jsonl")


def _log(rec):
    try:
        os.makedirs(LOG_DIR, exist_ok=True)
        with open(LOG, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


def rel_l2(got, want):
    got, want = got.double().flatten(), want.double().flatten()
    return float((got - want).norm() / want.norm().clamp_min(1e-30))


def compare(name, got, want, atol, rtol, kernel=None):
    """assert |got - want| <= atol + rtol*|want| elementwise; always logs max-abs / rel-L2 errors and,
    on failure, where the worst element sits (row/col patterns reveal layout bugs)."""
    assert got.shape == want.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    g, w = got.double(), want.double().to(got.device)
    finite = bool(torch.isfinite(g).all())
    diff = (g - w).abs()
    tol = atol + rtol * w.abs()
    bad = diff > tol
    nbad = int(bad.sum())
    worst = int(torch.argmax(diff - tol))
    idx = [int(i) for i in torch.unravel_index(torch.tensor(worst), got.shape)] if got.ndim else []
    rec = dict(case=name, kernel=kernel, shape=list(got.shape), max_abs=float(diff.max()), rel_l2=rel_l2(g, w), nbad=nbad,
               frac_bad=nbad / max(1, got.numel()), worst_idx=idx, got=float(g.flatten()[worst]), want=float(w.flatten()[worst]),
               finite=finite, atol=atol, rtol=rtol)
    _log(rec)
    assert finite, f"{name}: non-finite values in output ({kernel})"
    assert nbad == 0, (f"{name} [{kernel}]: {nbad}/{got.numel()} elements off; max_abs={rec['max_abs']:.4g} "
                       f"rel_l2={rec['rel_l2']:.4g} worst@{idx}: got {rec['got']:.6g} want {rec['want']:.6g}")
    return rec


def log_value(name, **kw):
    _log(dict(case=name, **kw))


def storage_floor(ref, fwd, y32, dtype=torch.float16, extra_leaf=(), extra_comp=()):
    """rel. L2 of the fp32 oracle `ref` re-run (through `fwd()`) with every op output rounded to `dtype` once -- exact arithmetic,
    16-bit activation STORAGE only: the error floor of any engine that keeps 16-bit activations (tools/error_budget.py)."""
    import sys
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import error_budget as EB
    hs = EB.storage_hooks(ref, dtype, blocks=True, extra_leaf=extra_leaf, extra_comp=extra_comp)
    try:
        with torch.no_grad():
            y = fwd()
    finally:
        for h in hs:
            h.remove()
    return rel_l2(y, y32)
