"""Pipe 4 of the MFMA implicit GEMM (csrc/igemm_pk.h): weights read from a packed copy (sfast_hip_pack_weight) straight into
registers. Parity vs the fp32 oracle AND bit-equality with the ring kernels on the same problem (same K order, same fp32
accumulation, same epilogue code), over the UNet's shapes, ragged edges, split-K, stacked weight segments, every epilogue form,
both conv sources, strided / 1x1 convs and the statistics-emitting staged epilogue."""
import numpy as np
import pytest
import torch

import sfast  # noqa: F401
from oracle import ops_ref as R
from parity import compare
from test_ops_gpu import CONV_CASES, F, _stats_reference, cl, last_kernel, rnd, tol

pytestmark = pytest.mark.gpu
DEV = "cuda"
PK = [41, 42, 43, 44, 45, 46]


def test_pack_weight_layout_bit_exact():
    """packed[(nb * KS + s) * 64 + lane] = w[nb * 32 + lane % 32][s * 16 + (lane // 32) * 8 : + 8], zeros outside [N) x [K)."""
    for N, K, ld in ((320, 320, 320), (77, 72, 72), (1280, 2880, 2880), (33, 200, 264)):
        for dtype in (torch.float16, torch.bfloat16):
            base = rnd(N, ld, dtype=dtype, seed=N + K)
            w = base[:, :K]
            pk = F().pack_weight(w)
            KS = (K + 63) // 64 * 4
            NB = (N + 31) // 32
            assert pk.numel() == NB * KS * 1024
            got = pk.view(torch.int16).reshape(NB, KS, 2, 32, 8).cpu()          # [nb][s][g][r][8]
            wp = torch.zeros(NB * 32, KS * 16, dtype=dtype)
            wp[:N, :K] = w.cpu()
            want = wp.view(torch.int16).reshape(NB, 32, KS, 2, 8).permute(0, 2, 3, 1, 4)
            assert torch.equal(got, want), (N, K, dtype)
    wc = cl(rnd(64, 128, 3, 3, seed=5))                                           # conv weight = its [Cout][9 * Cin] view
    assert torch.equal(F().pack_weight(wc), F().pack_weight(wc.permute(0, 2, 3, 1).reshape(64, -1)))


@pytest.mark.parametrize("M,K,N", [(8192, 320, 320), (8192, 320, 960), (2048, 2560, 640), (512, 1280, 1280), (154, 768, 640),
                                   (4095, 328, 324), (77, 64, 36), (130, 1280, 1280), (8192, 1280, 320)])
@pytest.mark.parametrize("variant", PK)
def test_linear_packed_variants(M, K, N, variant):
    x = rnd(M, K, seed=40)
    w = rnd(N, K, seed=41, scale=K ** -0.5)
    b = rnd(N, seed=42, scale=0.1)
    pk = F().pack_weight(w)
    y = F().linear(x, w, b, variant=variant, split_k=1, w_packed=pk)
    k = last_kernel()
    assert "igemm_lin" in k and ",pk" in k, k
    compare(f"linear packed M{M} K{K} N{N} v{variant}", y, R.linear_ref(x, w, b), *tol(x.dtype), kernel=k)
    assert torch.equal(y, F().linear(x, w, b, variant=21, split_k=1)), k         # same arithmetic as the ring kernel
    # without a packed copy the variant is not eligible: another pipe runs, same result
    y0 = F().linear(x, w, b, variant=variant, split_k=1)
    assert ",pk" not in last_kernel() and torch.equal(y0, y)


@pytest.mark.parametrize("split", [2, 3, 8])
@pytest.mark.parametrize("variant", PK)
def test_linear_packed_split_k(split, variant):
    x, w, b = rnd(128, 5120, seed=43), rnd(1280, 5120, seed=44, scale=5120 ** -0.5), rnd(1280, seed=45)
    r = rnd(128, 1280, seed=46)
    y = F().linear(x, w, b, residual=r, variant=variant, split_k=split, w_packed=F().pack_weight(w))
    k = last_kernel()
    assert f"split={split},pk" in k, k
    compare(f"linear packed split{split} v{variant}", y, R.linear_ref(x, w, b, residual=r), *tol(x.dtype), kernel=k)
    assert torch.equal(y, F().linear(x, w, b, residual=r, variant=21, split_k=split))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("epi", ["res_after", "res_before_relu", "gelu", "silu", "rowbias", "segs3", "inplace_res", "strided_out"])
@pytest.mark.parametrize("variant", [41, 43, 45])
def test_linear_packed_epilogues(dtype, epi, variant):
    M, K, N = 600, 640, 640
    x = rnd(M, K, dtype=dtype, seed=50)
    w = rnd(N, K, dtype=dtype, seed=51, scale=K ** -0.5)
    b = rnd(N, dtype=dtype, seed=52, scale=0.1)
    r = rnd(M, N, dtype=dtype, seed=53)
    pk = F().pack_weight(w)
    kw = {}
    if epi == "res_after":
        kw = dict(residual=r, alpha=0.5)
    elif epi == "res_before_relu":
        kw = dict(residual=r, alpha=2.0, res_before_act=True, act="relu")
    elif epi in ("gelu", "silu"):
        kw = dict(act=epi)
    elif epi == "rowbias":
        kw = dict(rowbias=rnd(3, N, dtype=dtype, seed=54), rows_per_batch=200)
    if epi == "segs3":  # stacked live segments (to_q / to_k / to_v): one packed copy each; 640 rows per segment = 20 row blocks
        ws = [rnd(N, K, dtype=dtype, seed=55 + i, scale=K ** -0.5) for i in range(3)]
        y = F().linear(x, ws, None, variant=variant, w_packed=[F().pack_weight(t) for t in ws])
        want = R.linear_ref(x, torch.cat(ws, 0))
    elif epi == "inplace_res":
        buf = r.clone()
        y = F().linear(x, w, b, residual=buf, out=buf, variant=variant, w_packed=pk)
        want = R.linear_ref(x, w, b, residual=r)
    elif epi == "strided_out":
        big = torch.zeros(M, 3 * N, dtype=dtype, device=DEV)
        y = F().linear(x, w, b, out=big[:, N:2 * N], variant=variant, w_packed=pk)
        want = R.linear_ref(x, w, b)
        assert float(big[:, :N].abs().max()) == 0 and float(big[:, 2 * N:].abs().max()) == 0
    else:
        y = F().linear(x, w, b, variant=variant, w_packed=pk, **kw)
        want = R.linear_ref(x, w, b, **kw)
    assert ",pk" in last_kernel(), last_kernel()
    compare(f"linear packed epi {epi} {dtype} v{variant}", y, want, *tol(dtype, 2.0), kernel=last_kernel())


def test_linear_packed_segments_of_320_rows_straddle_tiles():
    """to_q / to_k / to_v of the 320-wide level: 10 row blocks per segment, a 256-row tile covers blocks of two segments."""
    x = rnd(8192, 320, seed=60)
    ws = [rnd(320, 320, seed=61 + i, scale=320 ** -0.5) for i in range(3)]
    pks = [F().pack_weight(t) for t in ws]
    want = R.linear_ref(x, torch.cat(ws, 0))
    for variant in PK:
        y = F().linear(x, ws, None, variant=variant, w_packed=pks)
        assert ",pk" in last_kernel(), last_kernel()
        compare(f"linear packed qkv320 v{variant}", y, want, *tol(x.dtype), kernel=last_kernel())
    # a segment height that is not a multiple of 32 cannot use the pipe: the launch runs another one
    ws2 = [rnd(72, 320, seed=64 + i, scale=320 ** -0.5) for i in range(2)]
    y = F().linear(x, ws2, None, variant=41, w_packed=[F().pack_weight(t) for t in ws2])
    assert ",pk" not in last_kernel()
    compare("linear packed ineligible segments", y, R.linear_ref(x, torch.cat(ws2, 0)), *tol(x.dtype), kernel=last_kernel())


PK_CONV_CASES = [c for c in CONV_CASES if c[0] not in ("up 1280@16 ups", "cin32 odd", "cin8")]  # LDS-DMA pipes: 64-channel slices, no fused upsample


@pytest.mark.parametrize("case", PK_CONV_CASES, ids=[c[0] for c in PK_CONV_CASES])
@pytest.mark.parametrize("variant", PK)
def test_conv_packed(case, variant):
    name, B, Cin, H, W, Cout, k, stride, pad, ex = case
    x = cl(rnd(B, Cin, H, W, seed=70))
    c2 = ex.get("c2", 0)
    x2 = cl(rnd(B, c2, H, W, seed=71)) if c2 else None
    w = cl(rnd(Cout, Cin + c2, k, k, seed=72, scale=((Cin + c2) * k * k) ** -0.5))
    b = rnd(Cout, seed=73, scale=0.1)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    z = cl(rnd(B, Cout, Ho, Wo, seed=74)) if ex.get("z") else None
    rb = rnd(B, Cout, seed=75) if ex.get("rowbias") else None
    y = F().conv2d(x, w, b, z=z, stride=stride, padding=pad, x2=x2, rowbias=rb, variant=variant, split_k=1, w_packed=F().pack_weight(w))
    kname = last_kernel()
    assert "igemm_conv" in kname and ",pk" in kname, kname
    want = R.conv2d_ref(x, w, b, z, 1.0, stride, pad, x2=x2, rowbias=rb)
    compare(f"conv packed {name} v{variant}", y, want, *tol(x.dtype, 2.0), kernel=kname)
    assert torch.equal(y, F().conv2d(x, w, b, z=z, stride=stride, padding=pad, x2=x2, rowbias=rb, variant=21, split_k=1)), kname


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("variant,split", [(41, 4), (42, 6), (43, 3), (44, 2), (45, 2), (46, 12)])
def test_conv_packed_split_k(variant, split, dtype):
    x = cl(rnd(2, 1280, 16, 16, dtype=dtype, seed=80))
    w = cl(rnd(1280, 1280, 3, 3, dtype=dtype, seed=81, scale=11520 ** -0.5))
    b = rnd(1280, dtype=dtype, seed=82)
    z = cl(rnd(2, 1280, 16, 16, dtype=dtype, seed=83))
    y = F().conv2d(x, w, b, z=z, padding=1, variant=variant, split_k=split, w_packed=F().pack_weight(w))
    k = last_kernel()
    assert f"split={split},pk" in k, k
    compare(f"conv packed split{split} v{variant} {dtype}", y, R.conv2d_ref(x, w, b, z, 1.0, 1, 1), *tol(dtype, 2.0), kernel=k)


@pytest.mark.parametrize("variant,split", [(41, 1), (42, 1), (43, 1), (44, 1), (45, 1), (46, 1), (41, 4), (43, 6)])
@pytest.mark.parametrize("cin,cout,hw,unit", [(320, 320, 32, 10), (640, 1280, 16, 20)])
def test_conv_packed_emits_groupnorm_statistics(variant, split, cin, cout, hw, unit):
    x = cl(rnd(2, cin, hw, hw, seed=200, shift=0.5))
    w = cl(rnd(cout, cin, 3, 3, seed=201, scale=(9 * cin) ** -0.5))
    b = rnd(cout, seed=202, shift=2.0)
    z = cl(rnd(2, cout, hw, hw, seed=203))
    pk = F().pack_weight(w)
    try:
        y, stats, lay = F().conv2d(x, w, b, z=z, padding=1, variant=variant, split_k=split, gn_unit=unit, w_packed=pk)
    except Exception as e:  # a tile that does not divide H*W cannot emit statistics: the library says so instead of guessing
        assert "statistics" in str(e) or "tile" in str(e), e
        pytest.skip(f"variant {variant}: {e}")
    k = last_kernel()
    assert "+gnstats" in k and ",pk" in k, k
    assert torch.equal(y, F().conv2d(x, w, b, z=z, padding=1, variant=variant, split_k=split, w_packed=pk)), k
    want = _stats_reference(y.permute(0, 2, 3, 1).reshape(-1, cout), lay)
    got = stats.double().cpu().numpy().reshape(want.shape)
    used = ~np.isnan(want)
    assert np.allclose(got[..., 0][used[..., 0]], want[..., 0][used[..., 0]], rtol=1e-4, atol=1e-4), k
    assert np.allclose(got[..., 1][used[..., 1]], want[..., 1][used[..., 1]], rtol=2e-3, atol=1e-2), k
    gam, bet = rnd(cout, seed=204, shift=1.0, scale=0.2), rnd(cout, seed=205, scale=0.2)
    yn = F().group_norm_apply(y, 32, gam, bet, 1e-5, "silu", stats, lay)
    compare(f"gn_apply conv packed {cin}->{cout}@{hw} v{variant} s{split}", yn, R.group_norm_ref(y, 32, gam, bet, 1e-5, True), *tol(y.dtype, 2.0), kernel=k)


def test_repacking_follows_the_live_weight():
    x = rnd(512, 640, seed=90)
    w = rnd(640, 640, seed=91, scale=640 ** -0.5)
    pk = F().pack_weight(w)
    y1 = F().linear(x, w, None, variant=41, w_packed=pk)
    w.mul_(-2.0)
    stale = F().linear(x, w, None, variant=41, w_packed=pk)
    assert torch.equal(stale, y1)                                   # the kernel reads the packed copy, not `w`
    lib = __import__("sfast.hip.lib", fromlist=["x"]).load()
    rc = lib.sfast_hip_pack_weight(w.data_ptr(), pk.data_ptr(), 640, 640, 640, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    y2 = F().linear(x, w, None, variant=41, w_packed=pk)           # same buffer, re-packed in place: no new pointer for a captured graph
    compare("linear packed after re-pack", y2, R.linear_ref(x, w, None), *tol(x.dtype), kernel=last_kernel())


def test_lora_merge_matches_the_unfused_forward():
    """sfast_hip_lora_merge: W_eff = W + s * up @ down for a table of linears in one launch, fp32 math, one rounding."""
    import ctypes as C
    from sfast.hip import lib as L
    lib = L.init_device()
    shapes = [(320, 320, 4), (640, 768, 16), (1280, 1280, 64), (96, 72, 128), (1280, 768, 8)]
    for dtype, dt in ((torch.float16, 0), (torch.bfloat16, 1)):
        ents = (L.LoraEntry * len(shapes))()
        keep, scales = [], []
        for i, (N, K, r) in enumerate(shapes):
            w = rnd(N, K, dtype=dtype, seed=300 + i, scale=K ** -0.5)
            down = rnd(r, K, dtype=dtype, seed=310 + i, scale=K ** -0.5)
            up = rnd(N, r, dtype=dtype, seed=320 + i, scale=0.3)
            out = torch.full((N, K), float("nan"), dtype=dtype, device=DEV)
            keep.append((w, down, up, out))
            scales.append(0.25 * (i + 1))
            e = ents[i]
            e.w, e.down, e.up, e.out = w.data_ptr(), down.data_ptr(), up.data_ptr(), out.data_ptr()
            e.N, e.K, e.r, e.ldw, e.ldd, e.ldu, e.scale_index = N, K, r, K, K, r, i
        total = C.c_int32()
        L.check(lib.sfast_hip_lora_merge_plan(ents, len(shapes), C.byref(total)), "lora_merge_plan")
        tab = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(DEV)
        sc = torch.tensor(scales, dtype=torch.float32, device=DEV)
        L.check(lib.sfast_hip_lora_merge(tab.data_ptr(), len(shapes), total.value, sc.data_ptr(), dt, torch.cuda.current_stream().cuda_stream), "lora_merge")
        for (w, down, up, out), s in zip(keep, scales):
            want = w.double() + s * (up.double() @ down.double())
            compare(f"lora_merge {tuple(w.shape)} r{down.shape[0]} {dtype}", out, want.float(), *tol(dtype), kernel=L.last_kernel())


def test_engine_repacks_after_an_in_place_update():
    """sync_packed() refreshes a packed copy when the parameter's version counter moved -- also for an engine built from a raw
    parameter dict (the tensors' own counters) -- and is a no-op otherwise; a forward after it equals a fresh engine's."""
    from oracle import unet_ref as U
    from sfast.engine import UNet2DEngine
    from sfast.engine import unet2d as E
    from sfast.engine.unet_spec import random_params
    if not E.PACKED_WEIGHTS:
        pytest.skip("SFAST_PACKED_WEIGHTS=0: the engine keeps no packed copies")
    cfg = U.tiny_config()
    params = random_params(cfg, seed=3, dtype=torch.float16, device=DEV)
    eng = UNet2DEngine(cfg, params)
    g = torch.Generator().manual_seed(4)
    s = torch.randn(2, 4, 16, 16, generator=g).to(DEV, torch.float16)
    e = torch.randn(2, 77, 64, generator=g).to(DEV, torch.float16)
    y0 = eng.forward(s, 500, e)
    name = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.weight"
    w = params[name]
    rec = eng._packed_for(w)                                        # a record as the planner makes it for an op on pipe 4
    assert rec is not None and rec["name"] == name
    rec["users"] += 1
    assert torch.equal(rec["buf"], F().pack_weight(w)) and eng.sync_packed() == 0
    before = rec["buf"].clone()
    w.mul_(-1.0)                                                    # in place, through the tensor the engine was given
    assert eng.sync_packed() == 1 and eng.sync_packed() == 0
    assert not torch.equal(rec["buf"], before) and torch.equal(rec["buf"], F().pack_weight(w))
    y1 = eng.forward(s, 500, e)
    assert torch.equal(y1, UNet2DEngine(cfg, params).forward(s, 500, e)) and not torch.equal(y1, y0)
    assert eng._packed_for(torch.zeros(64, 64, dtype=torch.float16, device=DEV)) is None   # not one of the engine's parameters
