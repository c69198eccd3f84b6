"""Full-size parity of the two configurations BASELINE.json names beside SD1.5 bs=1 (VERDICT r01 "What's weak" #2, #9):

  * SDXL-base UNet, B=2 (the CFG batch of "bs=1"), 128x128 latent: 10-deep transformer stacks at head dim 64, add_embedding
    2816 -> 1280, linear proj_in / proj_out, 2.57 B parameters -- engine vs the fp32 oracle on the same inputs;
  * SD1.5 at B=16: the per-GPU shape of BASELINE configs[3] (bs=64 sharded 8-way = 8 images per GPU, x2 for CFG).

Same assertion form as tests/test_unet_gpu.py::test_sd15_unet_parity_and_graph: relative L2 vs the fp32 oracle bounded, and not
worse than 1.5x what the same module run eagerly in fp16 by PyTorch-ROCm (the diffusers-fp16 stand-in) loses. The f16-storage
floor of the network (tools/error_budget.py: exact arithmetic, activations rounded to f16 between ops) is logged beside it.
"""
import os
import sys

import pytest
import torch

from oracle import unet_ref as U
from parity import log_value, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _storage_floor(ref, fwd, y32):
    import error_budget as EB
    hs = EB.storage_hooks(ref, torch.float16, blocks=True)
    try:
        y = fwd()
    finally:
        for h in hs:
            h.remove()
    return rel_l2(y, y32)


def test_sdxl_full_size_unet_parity():
    from sfast.engine import UNet2DEngine
    cfg = U.SDXL_CONFIG
    m = U.build("sdxl", seed=0, dtype=torch.float16, device=DEV)
    assert sum(p.numel() for p in m.parameters()) == 2_567_463_684
    B = 2
    g = torch.Generator().manual_seed(5)
    sample = torch.randn(B, 4, 128, 128, generator=g).to(DEV, torch.float16)
    ehs = torch.randn(B, 77, 2048, generator=g).to(DEV, torch.float16)
    added = dict(text_embeds=torch.randn(B, 1280, generator=g).to(DEV, torch.float16),
                 time_ids=torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * B, device=DEV, dtype=torch.float16))
    eng = UNet2DEngine.from_module(m)
    y = eng.forward(sample, 981, ehs, added)
    assert y.shape == (B, 4, 128, 128) and torch.isfinite(y).all()
    # hipGraph replay of the same plan is bitwise identical to the eager plan
    plan = eng.get_plan(B, 128, 128, 77)
    gph, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(gph, stream=s):
            plan.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.current_stream().wait_stream(s)
    gph.replay()
    torch.cuda.synchronize()
    assert torch.equal(plan.static_out, y)
    with torch.no_grad():
        y16 = m(sample, 981, ehs, added_cond_kwargs=added).sample
    del eng, plan, gph
    ref = m.float()  # in place: the f16 weights are exactly representable, one 10 GB copy instead of two resident models
    a32 = {k: v.float() for k, v in added.items()}

    def fwd():
        with torch.no_grad():
            return ref(sample.float(), 981, ehs.float(), added_cond_kwargs=a32).sample

    y32 = fwd()
    floor = _storage_floor(ref, fwd, y32)
    e_engine, e_eager = rel_l2(y, y32), rel_l2(y16, y32)
    log_value("sdxl B=2 128x128 full-size parity", engine_vs_fp32=e_engine, eager16_vs_fp32=e_eager, engine_vs_eager16=rel_l2(y, y16),
              f16_storage_floor=floor, max_abs_engine=float((y.float() - y32).abs().max()), ref_absmax=float(y32.abs().max()),
              dispatches=None)
    assert e_engine < 2e-3, (e_engine, e_eager, floor)   # measured 1.16e-3
    assert e_engine < e_eager, (e_engine, e_eager)        # eager fp16: 2.2e-3
    assert e_engine < 1.5 * floor + 2e-4, (e_engine, floor)  # the engine sits at the f16-storage floor of this network


def test_sd15_batch16_parity():
    """BASELINE configs[3]: 8 images per GPU -> UNet batch 16. Every op is per-sample, so besides the oracle comparison the
    first two rows must reproduce the B=2 plan's output for the same inputs to rounding (other tile shapes, same arithmetic)."""
    from sfast.engine import UNet2DEngine
    m = U.build("sd15", seed=0, dtype=torch.float16, device=DEV)
    B = 16
    g = torch.Generator().manual_seed(6)
    sample = torch.randn(B, 4, 64, 64, generator=g).to(DEV, torch.float16)
    ehs = torch.randn(B, 77, 768, generator=g).to(DEV, torch.float16)
    t = torch.tensor([981.0, 961.0] * 8, device=DEV)
    eng = UNet2DEngine.from_module(m)
    y = eng.forward(sample, t, ehs)
    assert torch.isfinite(y).all()
    y2 = eng.forward(sample[:2], t[:2], ehs[:2])
    e_batch = rel_l2(y[:2], y2)
    with torch.no_grad():
        y16 = m(sample, t, ehs).sample
    ref = m.float()
    with torch.no_grad():
        y32 = ref(sample.float(), t, ehs.float()).sample
    e_engine, e_eager = rel_l2(y, y32), rel_l2(y16, y32)
    log_value("sd15 B=16 parity", engine_vs_fp32=e_engine, eager16_vs_fp32=e_eager, rows01_vs_B2_plan=e_batch)
    assert e_engine < 2.5e-3 and e_engine < e_eager, (e_engine, e_eager)   # SD1.5 B=16: measured 1.60e-3 vs 2.98e-3
    # Two plans with different kernel choices (tile shapes, split-K factors, K-tile order of pipe 5) are two independent f16-rounded
    # evaluations of the same function: each sits ~1.6e-3 from the fp32 forward, so they sit up to sqrt(2) x that apart from each other
    # (measured 1.98e-3 - 2.00e-3 over rounds 3 - 6; the fixed 2e-3 this line used to assert was a coincidence of that, and the re-chosen
    # SD1.5 kernels of round 6 crossed it at 2.002e-3). Bound: 1.5 x the larger of the two rows' own distances to the fp32 forward.
    # (Bit-equal rows across batch sizes are what SFAST_BATCH_INVARIANT=1 is for: tests/test_unet_gpu.py.)
    e_rows16, e_rows2 = rel_l2(y[:2], y32[:2]), rel_l2(y2, y32[:2])
    assert e_batch < 1.5 * max(e_rows16, e_rows2), (e_batch, e_rows16, e_rows2)
