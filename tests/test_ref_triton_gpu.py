"""Rows a6-a9 / a15 of SURVEY section 8 pinned against THE REFERENCE ITSELF: the reference's own Triton kernels
(/root/reference/src/sfast/triton/ops/{group_norm,layer_norm,copy,conv}.py, staged as oracle/_ref/sfast_ref_triton.zip by
oracle/make_ref.py) run on this MI355X in a subprocess (oracle/ref_triton_run.py), and BOTH the oracle restatement
(`oracle.ops_ref`) and the HIP kernels (through the C ABI) are compared with their outputs on the same seeded inputs:
the reference's self-check shapes (group_norm.py:485-523, layer_norm.py:522, copy.py:303-311, tests/triton/test_torch_ops.py:16)
and the SD1.5 / SDXL / VAE layer shapes.

Tolerances. All three computations do fp32 arithmetic on identical 16-bit inputs and round the result to 16 bits once, so
they agree to the output rounding plus the difference in reduction order: |a - b| <= atol + rtol |b| with rtol = 2^-10 (f16: one
ulp) / 2^-7 (bf16) and atol one ulp at the output scale. The reference's own checks use 1e-2 (group_norm.py:499, layer_norm.py:431).
One exception, by construction of the REFERENCE kernel and asserted as such: its statistics tensors `mean` / `rstd` are
allocated in the INPUT dtype (group_norm.py:404-415), so the channels_last apply pass normalises with an f16-rounded mean and
rstd (relative 2^-11 each) -- the outputs then differ from an fp32-statistics GroupNorm by up to ~2^-10 * (|x - mean| * rstd + |mean| * rstd),
which the tolerance below carries as an explicit term computed from the case's own statistics.

The archive cannot exist without /root/reference having been staged (it is git-ignored); without it these tests SKIP and the
committed fixture tests/golden/ref_triton_small.pt (outputs of the same kernels, tests/test_oracle.py) is what pins the oracle.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

import sfast  # noqa: F401
from oracle import ops_ref as R
from oracle import ref_cases as RC
from oracle import make_ref
from parity import compare, log_value

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


def F():
    from sfast.hip import functional
    return functional


def last_kernel():
    from sfast.hip import lib
    return lib.last_kernel()


@pytest.fixture(scope="module")
def ref():
    if not make_ref.available():
        pytest.skip("oracle/_ref/sfast_ref_triton.zip not staged (run oracle/make_ref.py where /root/reference exists)")
    import tempfile
    out = os.path.join(tempfile.gettempdir(), "sfast_ref_triton_full.pt")  # ~100 MB of outputs: not into gpurun_out/ (64 MiB cap)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_triton_run.py"), "--out", out, "--time"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    with open(os.path.join(ROOT, "gpurun_out", "ref_triton_run.log"), "w") as f:
        f.write(p.stdout + "\n--- stderr ---\n" + p.stderr[-20000:])
    assert p.returncode == 0, p.stderr[-3000:]
    res = torch.load(out, weights_only=False)
    log_value("ref_triton_status", **{k: str(v) for k, v in res["status"].items()}, triton=res["triton"], device=res["device"])
    return res


def _ulp(dtype):
    return 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7


def _time(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def _got(ref, name):
    rec = ref["out"].get(name)
    if rec is None or "error" in rec:
        pytest.fail(f"the reference's Triton kernel did not run for {name}: {(rec or {}).get('error', 'absent')} "
                    f"(family status: {ref['status']})")
    return rec


@pytest.mark.parametrize("case", RC.GN_CASES, ids=[c["name"] for c in RC.GN_CASES])
def test_group_norm_vs_reference_triton(ref, case):
    rec = _got(ref, case["name"])
    x, w, b = (t.to(DEV) for t in RC.gn_inputs(case))
    y_ref = rec["y"].to(DEV)
    dt = x.dtype
    u = _ulp(dt)
    want32 = R.group_norm_ref(x, case["groups"], w, b, case["eps"], case["silu"])
    # the reference rounds mean / rstd to the input dtype before the apply pass (docstring): carry that as an explicit term
    N, C = x.shape[:2]
    xs = x.float().reshape(N, case["groups"], -1)
    mean = xs.mean(2, keepdim=True)
    rstd = (xs.var(2, unbiased=False, keepdim=True) + case["eps"]).rsqrt()
    stat_term = (u / 2) * ((xs - mean).abs() * rstd + mean.abs() * rstd)          # |d y_norm| from rounding mean and rstd
    cpg = C // case["groups"]
    if x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
        stat_term = stat_term.reshape(N, case["groups"], cpg, -1).reshape(N, C, *x.shape[2:])
    else:
        stat_term = stat_term.reshape(x.shape)
    stat_term = stat_term * w.float().abs().reshape(1, C, *([1] * (x.ndim - 2))) * (1.1 if case["silu"] else 1.0)
    # diagnostics first: the statistics the reference returns against the exact ones (logged whether or not the case passes)
    ref_mean, ref_rstd = rec["mean"].float().to(DEV), rec["rstd"].float().to(DEV)
    log_value(f"ref_triton_stats {case['name']}", mean_max_err=float((ref_mean - mean.reshape(N, -1)).abs().max()),
              rstd_max_rel_err=float(((ref_rstd - rstd.reshape(N, -1)) / rstd.reshape(N, -1)).abs().max()),
              y_finite=bool(torch.isfinite(y_ref.float()).all()))
    # A reference-side platform deviation is recorded as such, not papered over: when the statistics the REFERENCE'S kernel returns are
    # themselves off by more than 8 output ulps, the case cannot pin anything. Seen on exactly one row -- [2, 2560, 8, 8], C/G = 80 ->
    # ROW_SIZE 128 x BLOCK_SIZE 32, 16 warps: mean off by 2.1e-2, rstd by 9.3 % (profiles/r04_ref_triton_tests_run2.log) -- while the
    # same kernel run through Triton's CPU interpreter (TRITON_INTERPRET=1, in this container) returns the exact statistics (mean err
    # 7e-6, rstd 4.8e-4): a Triton-ROCm 3.6 code-generation problem of that tile shape, not the reference's algorithm and not the
    # oracle. The HIP kernel is still checked against the oracle for the shape, then the case is reported as skipped.
    ref_stats_bad = (float(((ref_rstd - rstd.reshape(N, -1)) / rstd.reshape(N, -1)).abs().max()) > 8 * u
                     or float((ref_mean - mean.reshape(N, -1)).abs().max()) > 8 * u * float(1.0 + mean.abs().max()))
    if ref_stats_bad:
        y = F().group_norm(x, case["groups"], w, b, case["eps"], "silu" if case["silu"] else None)
        compare(f"hip_vs_oracle (reference deviates) {case['name']}", y.float(), want32, 2 * u, 2 * u, kernel=last_kernel())
        pytest.skip("the reference's Triton kernel returns wrong statistics for this tile shape on this Triton-ROCm build "
                    "(exact under TRITON_INTERPRET=1); HIP kernel checked against the oracle instead")
    # (a) oracle restatement vs the reference's kernel
    d = (want32 - y_ref.float()).abs()
    lim = u * (1.0 + want32.abs()) + 1.5 * stat_term
    nbad = int((d > lim).sum())
    log_value(f"oracle_vs_ref_triton {case['name']}", max_abs=float(d.max()), nbad=nbad,
              rel_l2=float((want32 - y_ref.float()).norm() / want32.norm()))
    assert nbad == 0, f"oracle.ops_ref.group_norm_ref differs from the reference's Triton kernel: {nbad} elements, max {float(d.max()):.4g}"
    # (b) HIP kernel vs the reference's kernel
    y = F().group_norm(x, case["groups"], w, b, case["eps"], "silu" if case["silu"] else None)
    assert y.is_contiguous(memory_format=torch.channels_last) == rec["cl"] or y.is_contiguous()
    d = (y.float() - y_ref.float()).abs()
    nbad = int((d > lim + u * y_ref.float().abs()).sum())
    log_value(f"hip_vs_ref_triton {case['name']}", kernel=last_kernel(), max_abs=float(d.max()), nbad=nbad,
              rel_l2=float((y.float() - y_ref.float()).norm() / y_ref.float().norm()))
    assert nbad == 0, f"HIP GroupNorm [{last_kernel()}] differs from the reference's Triton kernel: {nbad} elements, max {float(d.max()):.4g}"
    # statistics the reference returns (f16): mean / rstd of the fp32 restatement within one rounding
    assert torch.allclose(rec["mean"].float().to(DEV), mean.reshape(N, -1), atol=u, rtol=u)
    assert torch.allclose(rec["rstd"].float().to(DEV), rstd.reshape(N, -1), atol=u, rtol=2 * u)
    t = ref["timing"].get(case["name"])
    if t:
        us = _time(lambda: F().group_norm(x, case["groups"], w, b, case["eps"], "silu" if case["silu"] else None))
        log_value(f"time_gn {case['name']}", ref_triton_us=t["us"], hip_us=us, bytes=t["bytes"],
                  ref_triton_gbps=t["bytes"] / t["us"] / 1e3, hip_gbps=t["bytes"] / us / 1e3)


@pytest.mark.parametrize("case", RC.LN_CASES, ids=[c["name"] for c in RC.LN_CASES])
def test_layer_norm_vs_reference_triton(ref, case):
    rec = _got(ref, case["name"])
    x, w, b = (t.to(DEV) for t in RC.ln_inputs(case))
    y_ref = rec["y"].to(DEV)
    u = _ulp(x.dtype)
    want32 = R.layer_norm_ref(x, (x.shape[-1],), w, b, case["eps"])
    compare(f"oracle_vs_ref_triton {case['name']}", y_ref.float(), want32, u, u)
    y = F().layer_norm(x, (x.shape[-1],), w, b, case["eps"])
    compare(f"hip_vs_ref_triton {case['name']}", y.float(), y_ref.float(), 2 * u, 2 * u, kernel=last_kernel())
    t = ref["timing"].get(case["name"])
    if t:
        us = _time(lambda: F().layer_norm(x, (x.shape[-1],), w, b, case["eps"]))
        log_value(f"time_ln {case['name']}", ref_triton_us=t["us"], hip_us=us, bytes=t["bytes"],
                  ref_triton_gbps=t["bytes"] / t["us"] / 1e3, hip_gbps=t["bytes"] / us / 1e3)


@pytest.mark.parametrize("case", RC.COPY_CASES, ids=[c["name"] for c in RC.COPY_CASES])
def test_strided_copy_vs_reference_triton(ref, case):
    rec = _got(ref, case["name"])
    assert rec["equal_to_torch_copy"], "the reference's Triton copy itself disagrees with torch.copy_ on this box"
    x = RC.copy_inputs(case).to(DEV)
    src, fmt = RC.copy_view(case, x)
    dst = torch.empty(src.shape, dtype=src.dtype, device=DEV).contiguous(memory_format=fmt)
    F().strided_copy(src, dst)
    assert abs(float(dst.double().sum()) - rec["sum"]) <= 1e-9 * max(1.0, abs(rec["sum"])), "digest differs from the reference's copy"
    assert torch.equal(dst, torch.empty_like(dst).copy_(src))
    if rec.get("y") is not None:
        assert torch.equal(dst.cpu(), rec["y"])
    t = ref["timing"].get(case["name"])
    if t:
        us = _time(lambda: F().strided_copy(src, dst))
        log_value(f"time_copy {case['name']}", ref_triton_us=t["us"], hip_us=us, bytes=t["bytes"],
                  ref_triton_gbps=t["bytes"] / t["us"] / 1e3, hip_gbps=t["bytes"] / us / 1e3)


@pytest.mark.parametrize("case", RC.CONV_CASES, ids=[c["name"] for c in RC.CONV_CASES])
def test_conv_vs_reference_triton(ref, case):
    rec = ref["out"].get(case["name"])
    if rec is None or "error" in rec:
        # the reference disables its Triton conv by default (compilers/diffusion_pipeline_compiler.py: enable_triton = False
        # guards it) and the kernel targets Triton 2.0; a compile failure under Triton-ROCm 3.x is recorded, not hidden
        log_value(f"ref_triton_conv_unavailable {case['name']}", error=(rec or {}).get("error", ref["status"].get("conv", "absent")))
        pytest.skip("the reference's Triton conv does not compile / run under this Triton-ROCm: " + str((rec or {}).get("error", ref["status"].get("conv")))[:300])
    x, w, b = RC.conv_inputs(case)
    x, w = x.to(DEV), w.to(DEV)
    b = b.to(DEV) if b is not None else None
    y_ref = rec["y"].to(DEV)
    want32 = R.conv2d_ref(x, w, b, stride=case["stride"], padding=case["padding"])
    K = case["w"][1] * case["w"][2] * case["w"][3]
    # The reference's Triton conv accumulates f16 inputs IN F16 (conv.py:844-845 `ACC_TYPE = tl.float16`, `acc += tl.dot(..., out_dtype=ACC_TYPE)`
    # :449): every one of the K / BLOCK_K partial dot products is rounded into an f16 running sum, a random walk of ~sqrt(K / 32) half-ulps
    # of the sum's magnitude. The oracle (and the HIP kernels: fp32 MFMA accumulate) do not reproduce that loss, so the bar carries it as
    # an explicit factor: u * max(2, 0.75 * sqrt(K / 32)) -- 7 ulp at K = 2880, 3.4 at K = 640 (measured on the 320->320 3x3 case:
    # max 1.0e-2 at |y| = 1.27, rel. L2 1.0e-3, profiles/r04_ref_triton_tests_run1.log). bf16 / fp32 inputs accumulate in fp32 there too.
    u = _ulp(x.dtype)
    f = max(2.0, 0.75 * (K / 32.0) ** 0.5) if x.dtype == torch.float16 else 2.0
    rec1 = compare(f"oracle_vs_ref_triton {case['name']}", y_ref.float(), want32, f * u, f * u)
    y = F().conv2d(x, w, b, stride=case["stride"], padding=case["padding"])
    rec2 = compare(f"hip_vs_ref_triton {case['name']}", y.float(), y_ref.float(), (f + 1) * u, (f + 1) * u, kernel=last_kernel())
    rec3 = compare(f"hip_vs_oracle {case['name']}", y.float(), want32, 2 * u, 2 * u, kernel=last_kernel())
    assert rec3["rel_l2"] <= rec1["rel_l2"] + 1e-4, "the HIP conv (fp32 accumulate) should be at least as close to the fp32 oracle as the reference's f16-accumulating kernel"
