"""Whole-path parity on a real MI355X: the native engine (HIP kernels through the C ABI, eager and
hipGraph replay, and behind sfast.compilers.compile()) vs the oracle UNet.

Tolerance: BASELINE.json asks for <= 1e-3 relative vs "the diffusers fp16 UNet". diffusers is not
installable, so the stand-ins are the oracle restatement in fp32 (ground truth) and the same
restatement run eagerly in fp16 on the GPU (= what diffusers fp16 computes). The error budget
(tools/error_budget.py, profiles/r02_error_budget_sd15_gpu.jsonl) splits the engine's distance to
the fp32 truth: rounding every activation to f16 ONCE per op with otherwise exact fp32 arithmetic
already costs 1.63e-3 on SD1.5 (1.06e-3 on SDXL) -- the floor of ANY pipeline that stores f16
activations -- the engine measures 1.60e-3 (1.16e-3), its f16 attention probabilities contribute
7.6e-5, and eager PyTorch fp16 sits at 3.35e-3 (2.2e-3). 1e-3 against fp32 is therefore not
reachable with f16 activations; against the fp16 pipeline the target names, the engine is the
more accurate of the two. Asserted on the full-size models: <= 2.5e-3 vs the fp32 oracle AND
better than the eager-fp16 stand-in's own error (SDXL additionally: within 1.5x of the measured
storage floor); the tiny topologies (random weights, no averaging over width) keep 4e-3 where
they exercise plumbing. Every measured value is logged to gpurun_out/parity.jsonl and quoted in
DESIGN.md.
"""
import os
import types

import pytest
import torch

from oracle import unet_ref as U
from parity import compare, log_value, rel_l2, storage_floor

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def _engine(model):
    from sfast.engine import UNet2DEngine
    return UNet2DEngine.from_module(model)


def _inputs(cfg, B, seed=0, S=77, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    hw = cfg["sample_size"]
    sample = torch.randn(B, cfg["in_channels"], hw, hw, generator=g).to(DEV, dtype)
    ehs = torch.randn(B, S, cfg["cross_attention_dim"], generator=g).to(DEV, dtype)
    return sample, ehs


def test_tiny_unet_matches_golden():
    gold = torch.load(os.path.join(GOLDEN, "unet_tiny.pt"))
    m = U.build(gold["config"], seed=gold["seed"], dtype=torch.float16, device=DEV)
    eng = _engine(m)
    y = eng.forward(gold["sample"].to(DEV), gold["timestep"], gold["encoder_hidden_states"].to(DEV))
    err = rel_l2(y.float().cpu(), gold["y"])
    log_value("unet_tiny vs golden", rel_l2=err)
    assert torch.isfinite(y).all() and err < 4e-3, err
    yb = eng.forward(gold["sample"].to(DEV), torch.tensor(gold["timesteps_b"], device=DEV), gold["encoder_hidden_states"].to(DEV))
    errb = rel_l2(yb.float().cpu(), gold["y_b"])
    log_value("unet_tiny per-sample timesteps vs golden", rel_l2=errb)
    assert errb < 4e-3, errb


def test_tiny_sdxl_style_unet():
    cfg = U.tiny_config(down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                        transformer_layers_per_block=(1, 2, 2), attention_head_dim=(1, 2, 2), use_linear_projection=True,
                        addition_embed_type="text_time", addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32)
    m = U.build(cfg, seed=7, dtype=torch.float16, device=DEV)
    ref = U.build(cfg, seed=7, dtype=torch.float32, device=DEV)
    ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
    sample, ehs = _inputs(cfg, 2, seed=1, S=50)
    added = dict(text_embeds=torch.randn(2, 64, device=DEV, dtype=torch.float16),
                 time_ids=torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * 2, device=DEV, dtype=torch.float16))
    y = _engine(m).forward(sample, 500, ehs, added)
    with torch.no_grad():
        want = ref(sample.float(), 500, ehs.float(), added_cond_kwargs={k: v.float() for k, v in added.items()}).sample
    err = rel_l2(y, want)
    log_value("unet_tiny_sdxl vs fp32 oracle", rel_l2=err)
    assert err < 4e-3, err


@pytest.fixture(scope="module")
def sd15():
    """Seeded random-init weights of the published architecture -- or, when SFAST_SD15_DIR names a diffusers model directory (or a unet
    .safetensors file), the REAL runwayml/stable-diffusion-v1-5 weights, loaded into the module every test of this file compares against
    (engine, fp32 oracle and eager fp16 all receive the same state dict; sfast.engine.unet_spec.load_params checks names and shapes)."""
    torch.manual_seed(0)
    m = U.build("sd15", seed=0, dtype=torch.float16, device=DEV)
    real = os.environ.get("SFAST_SD15_DIR")
    if real:
        from sfast.engine.unet_spec import SD15_CONFIG, load_params
        m.load_state_dict(load_params(real, SD15_CONFIG, dtype=torch.float16, device=DEV))
        log_value("sd15 fixture", weights=real)
    return m


def test_sd15_unet_parity_and_graph(sd15):
    """Full-size SD1.5 UNet, B=2 (the CFG batch of 'bs=1'): engine vs fp32 oracle vs eager fp16."""
    m = sd15
    sample, ehs = _inputs(U.SD15_CONFIG, 2, seed=3)
    eng = _engine(m)
    y = eng.forward(sample, 981, ehs)
    with torch.no_grad():
        y16 = m(sample, 981, ehs).sample  # eager PyTorch-ROCm fp16 = diffusers-fp16 stand-in
        ref = U.build("sd15", seed=0, dtype=torch.float32, device=DEV)
        ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
        y32 = ref(sample.float(), 981, ehs.float()).sample
        # f16-STORAGE floor of this network, measured here: the fp32 oracle with every op output rounded to f16 once (exact arithmetic)
        floor = storage_floor(ref, lambda: ref(sample.float(), 981, ehs.float()).sample, y32)
        del ref
    e_engine, e_eager, e_cross = rel_l2(y, y32), rel_l2(y16, y32), rel_l2(y, y16)
    log_value("sd15 B=2 parity", engine_vs_fp32=e_engine, eager16_vs_fp32=e_eager, engine_vs_eager16=e_cross, f16_storage_floor=floor,
              max_abs_engine=float((y.float() - y32).abs().max()), ref_absmax=float(y32.abs().max()))
    assert torch.isfinite(y).all()
    # floor-relative: no arithmetic error beyond what storing f16 activations costs ANY engine (measured: engine 1.60e-3,
    # floor 1.63e-3); and closer to the truth than the fp16 pipeline it replaces (3.35e-3)
    assert e_engine <= 1.1 * floor + 1e-4, (e_engine, floor, e_eager)
    assert e_engine < e_eager, (e_engine, e_eager)

    # hipGraph replay reproduces the eager plan bit for bit, and is deterministic
    plan = eng.get_plan(2, 64, 64, 77)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.load_inputs(plan, sample, 981, ehs)
        with torch.cuda.graph(g, stream=s):
            plan.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    r1 = plan.static_out.clone()
    g.replay()
    r2 = plan.static_out.clone()
    torch.cuda.synchronize()
    assert torch.equal(r1, r2) and torch.equal(r1, y)

    # forked capture (time-embedding chain and text K/V projections as parallel graph branches) computes the same
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g2, stream=s):
            plan.run_forked(torch.cuda.current_stream())
    torch.cuda.current_stream().wait_stream(s)
    plan.static_out.zero_()
    g2.replay()
    r3 = plan.static_out.clone()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(r3, y) and torch.equal(plan.static_out, y)
    # side lanes: sinusoid + time MLP (2) + ONE grouped GEMV over the 22 projections + one grouped K/V GEMM per level width (3)
    assert sum(op.lane != 0 for op in plan.ops) == 1 + 2 + 1 + 3
    log_value("sd15 B=2 plan", launches=len(plan.ops))

    # batch independence: every op is per-sample, so B=1 must reproduce row 0 of the B=2 run
    y1 = eng.forward(sample[:1], 981, ehs[:1])
    e_b = rel_l2(y1, y[:1])
    log_value("sd15 B=1 vs row 0 of B=2", rel_l2=e_b)
    assert e_b < 2e-3, e_b


class MiniPipeline:
    """Just enough of a diffusers pipeline for compile(): .unet/.vae/.scheduler attributes and a
    CFG + DDIM denoise loop that calls unet(sample, t, encoder_hidden_states=..., return_dict=False)."""

    def __init__(self, unet):
        self.unet = unet
        self.vae = None
        self.device = torch.device(DEV)

    @torch.no_grad()
    def __call__(self, latents, ehs_uc, steps=3, guidance=7.5):
        from oracle.ops_ref import ddim_schedule
        ts, coefs = ddim_schedule(50)
        for t, c in list(zip(ts, coefs))[:steps]:
            inp = torch.cat([latents, latents])
            eps = self.unet(inp, t, encoder_hidden_states=ehs_uc, return_dict=False)[0]
            eu, ec = eps.float().chunk(2)
            e = eu + guidance * (ec - eu)
            x0 = (latents.float() - c[1] * e) / c[0]
            latents = (c[2] * x0 + c[3] * e).to(latents.dtype)
        return latents


def test_compile_drop_in_surface():
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile
    cfg = U.tiny_config()
    unet = U.build(cfg, seed=11, dtype=torch.float16, device=DEV)
    eager = U.build(cfg, seed=11, dtype=torch.float16, device=DEV)
    pipe = MiniPipeline(unet)
    config = CompilationConfig.Default()
    config.enable_xformers = True
    config.enable_triton = True
    config.enable_cuda_graph = True
    pipe2 = compile(pipe, config)
    assert pipe2 is pipe and hasattr(pipe.unet.forward, "_cached") and pipe.unet.forward.__self__ is pipe.unet
    lat = torch.randn(1, 4, 16, 16, device=DEV, dtype=torch.float16)
    ehs = torch.randn(2, 77, 64, device=DEV, dtype=torch.float16)
    out = pipe(lat, ehs)
    want = MiniPipeline(eager)(lat, ehs)
    err = rel_l2(out, want)
    log_value("compile() tiny pipeline 3 DDIM steps vs eager fp16", rel_l2=err)
    assert err < 1e-2, err
    assert len(pipe.unet.forward._cached) == 1
    # return_dict=True form and a second input signature (dynamic shape by recapture)
    o = pipe.unet(torch.cat([lat, lat]), 981, encoder_hidden_states=ehs)
    assert o.sample.shape == (2, 4, 16, 16)
    lat2 = torch.randn(2, 4, 32, 16, device=DEV, dtype=torch.float16)
    o2 = pipe.unet(lat2, torch.tensor(500, device=DEV), encoder_hidden_states=ehs[:, :40], return_dict=False)[0]
    w2 = eager(lat2, 500, ehs[:, :40]).sample
    assert len(pipe.unet.forward._cached) == 2 and rel_l2(o2, w2) < 1e-2
    # cross_attention_kwargs={"scale": s} (the LoRA scale diffusers pipelines pass along) has nothing to act on in a module without
    # LoRA layers: accepted natively, same plan, same result
    o2s = pipe.unet(lat2, torch.tensor(500, device=DEV), encoder_hidden_states=ehs[:, :40], cross_attention_kwargs={"scale": 0.5}, return_dict=False)[0]
    assert torch.equal(o2s, o2) and not pipe.unet.forward._warned and len(pipe.unet.forward._cached) == 2
    # unsupported call forms fall back to the original forward instead of computing something else (T2I-Adapter residuals are not a plan
    # input; the oracle module ignores the keyword, so both eager runs agree)
    o3 = pipe.unet(lat2, 500, encoder_hidden_states=ehs[:, :40], down_intrablock_additional_residuals=[torch.zeros(1, device=DEV)],
                   return_dict=False)[0]
    assert pipe.unet.forward._warned  # took the eager path (two eager fp16 runs differ by conv algorithm noise)
    assert rel_l2(o3, w2) < 1e-2 and len(pipe.unet.forward._cached) == 2
    # a UNet-level attention_mask whose length cannot match every self-attention level (512 keys at the top, 128 / 32 below) has no
    # plan: that signature is handed to the module's own forward, which refuses it as diffusers does (round 5)
    with pytest.raises(RuntimeError):
        pipe.unet(lat2, 500, encoder_hidden_states=ehs[:, :40], attention_mask=torch.ones(2, 512, device=DEV), return_dict=False)


@pytest.mark.parametrize("option", ["timestep_cond", "class_timestep", "class_projection", "all"])
def test_lcm_and_class_conditioned_calls_stay_native(option):
    """VERDICT r02 "eager cliffs": `timestep_cond` (LCM-distilled UNets, /root/reference/examples/optimize_lcm_pipeline.py),
    `class_labels` (class_embed_type "timestep" / "projection") and `cross_attention_kwargs={"scale": s}` are inputs of the native
    plan: tiny UNet through compile_unet() + hipGraph replay vs the fp32 oracle, and the fallback warning does not fire."""
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    over = {}
    if option in ("timestep_cond", "all"):
        over["time_cond_proj_dim"] = 256       # LCM's w-embedding width
    if option in ("class_timestep", "all"):
        over["class_embed_type"] = "timestep"
    if option == "class_projection":
        over.update(class_embed_type="projection", projection_class_embeddings_input_dim=96)
    cfg = U.tiny_config(**over)
    unet = U.build(cfg, seed=31, dtype=torch.float16, device=DEV)
    ref = U.build(cfg, seed=31, dtype=torch.float32, device=DEV)
    ref.load_state_dict({k: v.float() for k, v in unet.state_dict().items()})
    config = CompilationConfig.Default()
    config.enable_cuda_graph = True
    unet = compile_unet(unet, config)
    g = torch.Generator().manual_seed(32)
    x = torch.randn(2, 4, 16, 16, generator=g).to(DEV, torch.float16)
    ehs = torch.randn(2, 77, 64, generator=g).to(DEV, torch.float16)
    kw = {}
    if "time_cond_proj_dim" in over:
        kw["timestep_cond"] = torch.randn(2, 256, generator=g).to(DEV, torch.float16)
    if over.get("class_embed_type") == "timestep":
        kw["class_labels"] = torch.tensor([7.0, 912.0], device=DEV)
    if over.get("class_embed_type") == "projection":
        kw["class_labels"] = torch.randn(2, 96, generator=g).to(DEV, torch.float16)
    y = unet(x, 441, encoder_hidden_states=ehs, cross_attention_kwargs={"scale": 0.7}, return_dict=False, **kw)[0]
    y2 = unet(x, 441, encoder_hidden_states=ehs, return_dict=False, **kw)[0]   # graph replay, no cross_attention_kwargs: same plan
    with torch.no_grad():
        want = ref(x.float(), 441, ehs.float(), **{k: v.float() for k, v in kw.items()}).sample
        blind = ref(x.float(), 441, ehs.float(), **{k: torch.zeros_like(v.float()) for k, v in kw.items()}).sample
    err = rel_l2(y, want)
    log_value(f"tiny UNet {option} through compile_unet vs fp32 oracle", rel_l2=err, effect_of_the_input=rel_l2(blind, want))
    assert not unet.forward._warned and len(unet.forward._cached) == 1 and torch.equal(y, y2)
    assert err < 4e-3 and rel_l2(blind, want) > 1e-2, (err, rel_l2(blind, want))


def test_live_weight_update_without_recapture():
    """preserve_parameters / LoRA contract (reference README.md:228-265): an in-place parameter update
    must show up in the next replay of the already captured graph."""
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    cfg = U.tiny_config()
    unet = U.build(cfg, seed=12, dtype=torch.float16, device=DEV)
    config = CompilationConfig.Default()
    config.enable_cuda_graph = True
    unet = compile_unet(unet, config)
    x = torch.randn(2, 4, 16, 16, device=DEV, dtype=torch.float16)
    ehs = torch.randn(2, 77, 64, device=DEV, dtype=torch.float16)
    y0 = unet(x, 981, encoder_hidden_states=ehs, return_dict=False)[0]
    with torch.no_grad():
        for n, p in unet.named_parameters():
            if n.endswith("attn2.to_v.weight") or n.endswith("conv2.weight"):
                p.add_(0.02 * torch.randn_like(p))
    y1 = unet(x, 981, encoder_hidden_states=ehs, return_dict=False)[0]
    ref = U.build(cfg, seed=12, dtype=torch.float32, device=DEV)
    ref.load_state_dict({k: v.float() for k, v in unet.state_dict().items()})
    with torch.no_grad():
        want = ref(x.float(), 981, ehs.float()).sample
    assert len(unet.forward._cached) == 1
    assert rel_l2(y0, want) > 1e-2  # the update really changed the function
    assert rel_l2(y1, want) < 4e-3


_RCCL_SCRIPT = r"""
import os, sys
sys.path[:0] = [%(root)r, %(pkg)r]
import torch
import torch.distributed as dist
from oracle import unet_ref as U
from sfast.engine.replicas import broadcast_parameters
from sfast.engine.unet_spec import random_params
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    params = random_params(U.tiny_config(), seed=3, device="cuda")
    ref = {k: v.clone() for k, v in params.items()}
    ptrs = {k: v.data_ptr() for k, v in params.items()}
    n = broadcast_parameters(params, src=0, bucket_bytes=1 << 20, force=True)
    torch.cuda.synchronize()
    assert n == sum(v.numel() * v.element_size() for v in params.values())
    assert all(torch.equal(params[k], ref[k]) and params[k].data_ptr() == ptrs[k] for k in params)
    print("RCCL_OK", n)
finally:
    dist.destroy_process_group()
"""


def test_rccl_weight_broadcast_single_rank():
    """The RCCL leg of the replica path (bucketed in-place broadcast) on the one GPU this box has. In a process of its own: a
    process group that is created and destroyed inside the long-lived pytest process leaves RCCL / watchdog threads behind, and a
    later test of this file (the ControlNet -> UNet chain, right after its hipGraph captures) died with SIGSEGV / SIGABRT in a
    non-Python thread in 5 of 22 fresh runs of the file, with and without packed weights (profiles/r04_flaky_chain_test_run30.log);
    bench.py keeps its process group for the life of the process and has not shown it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _RCCL_SCRIPT % dict(root=root, pkg=os.path.join(root, "stable-fast_amd"))
    if os.environ.get("SFAST_TEST_INPROC"):  # crash hunt only (tools/gpu_r5.sh): the round-4 arrangement, process group inside pytest
        return exec(compile(script, "<rccl>", "exec"), {"__name__": "__rccl__"})
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


def test_add_strided_kernel():
    from sfast.hip import lib as L
    import ctypes as C
    lib = L.init_device()
    for dt, code in ((torch.float16, L.F16), (torch.bfloat16, L.BF16), (torch.float32, L.F32)):
        g = torch.Generator().manual_seed(3)
        src = torch.randn(2, 24, 5, 7, generator=g).to(DEV, dt)          # NCHW
        dst = torch.randn(2, 5, 7, 24, generator=g).to(DEV, dt)          # NHWC
        want = (dst.float() + src.float().permute(0, 2, 3, 1)).to(dt)
        ap = L.AddParams()
        ap.dtype, ap.ndim = code, 4
        ap.shape = (C.c_int64 * 4)(2, 5, 7, 24)
        ap.src_strides = (C.c_int64 * 4)(24 * 35, 7, 1, 35)
        ap.dst_strides = (C.c_int64 * 4)(35 * 24, 7 * 24, 24, 1)
        L.check(lib.sfast_hip_add_strided(src.data_ptr(), dst.data_ptr(), C.byref(ap), torch.cuda.current_stream().cuda_stream), "add")
        torch.cuda.synchronize()
        assert torch.equal(dst, want), dt


def test_controlnet_residuals_native_and_through_compile():
    """ControlNet residual inputs stay on the native engine (eager plan, hipGraph, compile())."""
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    cfg = U.tiny_config()
    m = U.build(cfg, seed=13, dtype=torch.float16, device=DEV)
    ref = U.build(cfg, seed=13, device=DEV)
    ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
    eng = _engine(m)
    sample, ehs = _inputs(cfg, 2, seed=5)
    plan = eng.get_plan(2, cfg["sample_size"], cfg["sample_size"], 77, True)
    g = torch.Generator().manual_seed(6)
    down = [(0.3 * torch.randn(*t.shape, generator=g)).to(DEV, torch.float16) for t in plan.static_in["down_block_additional_residuals"]]
    mid = (0.3 * torch.randn(*plan.static_in["mid_block_additional_residual"].shape, generator=g)).to(DEV, torch.float16)
    y = eng.forward(sample, 321, ehs, down_block_additional_residuals=down, mid_block_additional_residual=mid)
    with torch.no_grad():
        want = ref(sample.float(), 321, ehs.float(), down_block_additional_residuals=[d.float() for d in down],
                   mid_block_additional_residual=mid.float()).sample
        base = ref(sample.float(), 321, ehs.float()).sample
    err = rel_l2(y.float(), want)
    log_value("tiny unet + controlnet residuals vs fp32 oracle", rel_l2=err)
    assert err < 4e-3 and rel_l2(y.float(), base) > 5e-2
    # behind compile_unet(): same call signature as diffusers, graph replay, and it must NOT take the eager fallback
    c = CompilationConfig.Default()
    c.enable_cuda_graph = True
    unet = compile_unet(m, c)
    for _ in range(2):
        o = unet(sample, 321, encoder_hidden_states=ehs, down_block_additional_residuals=down, mid_block_additional_residual=mid,
                 return_dict=False)[0]
        assert rel_l2(o.float(), want) < 4e-3
    assert not unet.forward._warned and len(unet.forward._cached) == 1
    down2 = [d * 0 for d in down]
    o0 = unet(sample, 321, encoder_hidden_states=ehs, down_block_additional_residuals=down2, mid_block_additional_residual=mid * 0,
              return_dict=False)[0]
    assert rel_l2(o0.float(), base) < 4e-3  # zero residuals == plain UNet, through the same cached graph


def test_tiny_unet_bf16():
    cfg = U.tiny_config()
    m = U.build(cfg, seed=17, dtype=torch.bfloat16, device=DEV)
    ref = U.build(cfg, seed=17, device=DEV)
    ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
    sample, ehs = _inputs(cfg, 2, seed=8, dtype=torch.bfloat16)
    y = _engine(m).forward(sample, 700, ehs)
    with torch.no_grad():
        want = ref(sample.float(), 700, ehs.float()).sample
        eager = m(sample, 700, ehs).sample
    e_eng, e_eager = rel_l2(y.float(), want), rel_l2(eager.float(), want)
    log_value("tiny unet bf16", engine_vs_fp32=e_eng, eager_bf16_vs_fp32=e_eager)
    assert torch.isfinite(y).all() and e_eng < 3e-2 and e_eng < 1.5 * e_eager + 5e-3, (e_eng, e_eager)


_INNER = "SFAST_ISOLATED_TEST_INNER"


def _run_isolated(name):
    """Run test `name` of this file in a pytest process of its own, ONCE: any failure -- a death by SIGSEGV / SIGABRT included -- fails
    the calling test with the child's output. (Round 4 retried signal deaths once while the intermittent crash of the compiled
    ControlNet -> UNet chain was open; round 5 root-caused it -- DESIGN.md section 9, round 5, item 1 -- and the retry is gone. The chain
    test keeps its own process because it is the one test of this file that holds two compiled models with graphs at once.)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", f"{os.path.abspath(__file__)}::{name}", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       env=dict(os.environ, **{_INNER: name}), capture_output=True, text=True, timeout=1200,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if r.returncode != 0:
        raise AssertionError(f"{name} (isolated) failed: rc={r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}")


def test_isolation_helper_selftest(tmp_path):
    """The helper above, on a test that needs no GPU work: a passing child passes through, a child that dies with SIGSEGV fails the
    caller -- no retry."""
    if os.environ.get(_INNER) == "test_isolation_helper_selftest":
        if os.environ.get("SFAST_SELFTEST_DIE"):
            import signal
            os.kill(os.getpid(), signal.SIGSEGV)
        return
    _run_isolated("test_isolation_helper_selftest")
    os.environ["SFAST_SELFTEST_DIE"] = "1"
    try:
        with pytest.raises(AssertionError, match="rc=-11|rc=139"):
            _run_isolated("test_isolation_helper_selftest")
    finally:
        del os.environ["SFAST_SELFTEST_DIE"]


def test_controlnet_engine_and_compiled_chain():
    """ControlNetModel on the native engine, behind compile_unet(), chained into the compiled UNet."""
    # round 6: runs INLINE again -- the arrangement that crashed in round 4 (two compiled models with graphs behind the earlier tests of
    # this file); graph teardown is deferred by OwnedGraph now (sfast/engine/unet2d.py), see also the teardown test below
    from oracle import controlnet_ref as CN
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    ccfg, ucfg = CN.tiny_config(), U.tiny_config()
    cnet = CN.build(ccfg, seed=41, dtype=torch.float16, device=DEV)
    cref = CN.build(ccfg, seed=41, device=DEV)
    cref.load_state_dict({k: v.float() for k, v in cnet.state_dict().items()})
    unet = U.build(ucfg, seed=42, dtype=torch.float16, device=DEV)
    uref = U.build(ucfg, seed=42, device=DEV)
    uref.load_state_dict({k: v.float() for k, v in unet.state_dict().items()})
    sample, ehs = _inputs(ucfg, 2, seed=9)
    cond = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(10)).to(DEV, torch.float16)
    with torch.no_grad():
        wd, wm = cref(sample.float(), 444, ehs.float(), cond.float(), return_dict=False)
        want = uref(sample.float(), 444, ehs.float(), down_block_additional_residuals=wd, mid_block_additional_residual=wm).sample
    c = CompilationConfig.Default()
    c.enable_cuda_graph = True
    cnet = compile_unet(cnet, c)
    unet = compile_unet(unet, c)
    assert type(cnet._sfast_engine).__name__ == "ControlNetEngine"
    for _ in range(2):
        out = cnet(sample, 444, encoder_hidden_states=ehs, controlnet_cond=cond, return_dict=True)
        down, mid = out.down_block_res_samples, out.mid_block_res_sample
        errs = [rel_l2(a.float(), b) for a, b in zip(down, wd)] + [rel_l2(mid.float(), wm)]
        y = unet(sample, 444, encoder_hidden_states=ehs, down_block_additional_residuals=down, mid_block_additional_residual=mid,
                 return_dict=False)[0]
    log_value("tiny controlnet vs fp32 oracle", max_rel_l2=max(errs), chained_unet=rel_l2(y.float(), want))
    assert max(errs) < 4e-3 and rel_l2(y.float(), want) < 4e-3
    assert not cnet.forward._warned and not unet.forward._warned
    d2, m2 = cnet(sample, 444, encoder_hidden_states=ehs, controlnet_cond=cond, conditioning_scale=0.5, return_dict=False)
    assert rel_l2(m2.float(), 0.5 * wm) < 4e-3
    # guess_mode stays on the native plan (round 4): logspace(-1, 0) residual weights, as diffusers' ControlNetModel.forward applies them
    with torch.no_grad():
        gd, gm = cref(sample.float(), 444, ehs.float(), cond.float(), conditioning_scale=0.7, guess_mode=True, return_dict=False)
    d3, m3 = cnet(sample, 444, encoder_hidden_states=ehs, controlnet_cond=cond, conditioning_scale=0.7, guess_mode=True,
                  cross_attention_kwargs={"scale": 1.0}, return_dict=False)
    assert max(rel_l2(a.float(), b) for a, b in zip(d3, gd)) < 4e-3 and rel_l2(m3.float(), gm) < 4e-3
    assert not cnet.forward._warned


def test_batch_invariant_mode_is_bit_exact_across_batch_sizes():
    """with SFAST_BATCH_INVARIANT=1 (opt-in, read when the library and the tuner are first loaded -- hence a
    process of its own) every kernel choice and every statistics partition follows the PER-SAMPLE problem at the reference batch, so
    the same image gives the same latents whether it runs alone (B = 1), as the CFG pair (B = 2) or as one of 8 images per GPU
    (B = 16, the per-GPU shape of BASELINE configs[3]): bit-equal rows, full-size SD1.5 UNet. Without the mode the rows differ by
    ~2e-3 (different split-K factors per batch: `test_sd15_unet_parity_and_graph` logs it)."""
    name = "test_batch_invariant_mode_is_bit_exact_across_batch_sizes"
    if os.environ.get(_INNER) != name:
        old = os.environ.get("SFAST_BATCH_INVARIANT")
        os.environ["SFAST_BATCH_INVARIANT"] = "1"
        try:
            return _run_isolated(name)
        finally:
            if old is None:
                os.environ.pop("SFAST_BATCH_INVARIANT", None)
            else:
                os.environ["SFAST_BATCH_INVARIANT"] = old
    from sfast.engine import autotune
    assert autotune.BATCH_INVARIANT
    m = U.build("sd15", seed=0, dtype=torch.float16, device=DEV)
    eng = _engine(m)
    sample, ehs = _inputs(U.SD15_CONFIG, 16, seed=3)
    outs = {}
    for B in (2, 1, 16):
        outs[B] = eng.forward(sample[:B], 981, ehs[:B]).clone()
    with torch.no_grad():
        ref = U.build("sd15", seed=0, dtype=torch.float32, device=DEV)
        ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
        y32 = ref(sample[:2].float(), 981, ehs[:2].float()).sample
    log_value("sd15 batch-invariant mode", engine_vs_fp32_B2=rel_l2(outs[2], y32), b1_vs_b2_row0=rel_l2(outs[1], outs[2][:1]),
              b16_vs_b2_rows01=rel_l2(outs[16][:2], outs[2]))
    assert rel_l2(outs[2], y32) < 2.5e-3
    assert torch.equal(outs[1][0], outs[2][0])
    assert torch.equal(outs[16][:2], outs[2])


def test_graph_teardown_survives_del_and_gc():
    """a user who drops a compiled model right after its last replay (`del pipe; gc.collect()`) hit the trigger of
    the round-4 crash -- the winning graph of every plan was destroyed by plain reference counting, a few microseconds after its last
    launch. 100 compile -> replay -> del -> gc cycles in ONE fresh process (under the crash-backtrace shim when it is built), plus a
    DenoiseLoop re-capture per cycle; the retired queue must stay bounded and the process must exit cleanly."""
    name = "test_graph_teardown_survives_del_and_gc"
    if os.environ.get(_INNER) != name:
        ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        shim = os.path.join(ROOT, "tools", "_crashbt.so")
        if not os.path.exists(shim) and os.path.exists(os.path.join(ROOT, "tools", "crashbt.c")):
            import subprocess
            subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", shim, os.path.join(ROOT, "tools", "crashbt.c"), "-ldl", "-lpthread"], check=False)
        old = os.environ.get("LD_PRELOAD")
        if os.path.exists(shim):
            os.environ["LD_PRELOAD"] = shim
        try:
            return _run_isolated(name)
        finally:
            if old is None:
                os.environ.pop("LD_PRELOAD", None)
            else:
                os.environ["LD_PRELOAD"] = old
    import gc
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    from sfast.engine import UNet2DEngine
    from sfast.engine import unet2d as E
    from sfast.engine.denoise import DenoiseLoop
    ucfg = U.tiny_config()
    sample, ehs = _inputs(ucfg, 2, seed=9)
    c = CompilationConfig.Default()
    c.enable_cuda_graph = True
    first = None
    for i in range(100):
        unet = compile_unet(U.build(ucfg, seed=42, dtype=torch.float16, device=DEV), c)
        y = unet(sample, 444, encoder_hidden_states=ehs, return_dict=False)[0]
        y = unet(sample, 444, encoder_hidden_states=ehs, return_dict=False)[0]  # a graph replay ...
        if first is None:
            first = y.clone()
        assert torch.equal(y, first)
        del unet, y                                                               # ... and the model is gone in the same breath
        gc.collect()
        if i % 10 == 0:
            eng = UNet2DEngine.from_module(U.build(ucfg, seed=42, dtype=torch.float16, device=DEV))
            loop = DenoiseLoop(eng, images=1, height=16, width=16, ctx_len=77, num_steps=4)
            loop.set_inputs(sample[:1], ehs)
            loop.capture(warmups=1)
            loop.step(0)
            loop.capture(warmups=1)   # re-capture: the first graph retires
            loop.step(1)
            del loop, eng
            gc.collect()
        assert len(E._RETIRED) <= 2 * E._RETIRED_KEEP + 4, len(E._RETIRED)
    torch.cuda.synchronize()


# ---- trace_scheduler: the scheduler update as one HIP kernel behind diffusers' step() signature -------------------------
class _DDIMLike:
    """diffusers DDIMScheduler call surface (public semantics): eager reference for the native step."""
    init_noise_sigma = 1.0
    _sfast_ddim_like = True  # DDIM is recognised by class name or this opt-in, never by attributes

    def __init__(self, prediction_type="epsilon"):
        self.config = types.SimpleNamespace(num_train_timesteps=1000, prediction_type=prediction_type, clip_sample=False,
                                            thresholding=False, steps_offset=1)
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.num_inference_steps = None

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        self.timesteps = ((torch.arange(0, n) * (1000 // n)).flip(0) + 1).to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None, variance_noise=None,
             return_dict=True):
        t = int(timestep)
        prev = t - 1000 // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        mo, x = model_output.double(), sample.double()
        if self.config.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * x - (1 - a_t) ** 0.5 * mo
            eps = a_t ** 0.5 * mo + (1 - a_t) ** 0.5 * x
        else:
            x0 = (x - (1 - a_t) ** 0.5 * mo) / a_t ** 0.5
            eps = mo
        out = (a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps).to(sample.dtype)
        return (out,) if not return_dict else types.SimpleNamespace(prev_sample=out)


@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_native_scheduler_step_matches_ddim(pred, dtype):
    from sfast.libs.diffusers.scheduler import NativeDDIMStep, patch_scheduler
    ref, s = _DDIMLike(pred), _DDIMLike(pred)
    assert patch_scheduler(s) and isinstance(s.step, NativeDDIMStep) and s.step.__self__ is s
    for sc in (ref, s):
        sc.set_timesteps(50, device=DEV)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 4, 64, 64, generator=g).to(DEV, dtype)
    e = torch.randn(2, 4, 64, 64, generator=g).to(DEV, dtype)
    for i in (0, 17, 49):  # first step, a middle one, the last one (prev timestep < 0 -> final_alpha_cumprod)
        t = s.timesteps[i]  # 0-d int64 CUDA tensor, read by the kernel: no host sync
        got = s.step(e, t, x, return_dict=False)[0]
        want = ref.step(e, t, x, return_dict=False)[0]
        compare(f"native ddim step {pred} {dtype} i={i}", got, want, 2e-3 if dtype == torch.float16 else 1e-5, 1e-3 if dtype == torch.float16 else 1e-5)
        got_int = s.step(e, int(t), x).prev_sample  # python-int timestep: resolved on the host
        assert torch.equal(got_int, got)
    assert s.step.native_calls == 6
    # anything outside the deterministic linear form keeps the original method
    out = s.step(e, int(s.timesteps[3]), x, eta=0.5, return_dict=False)[0]
    assert s.step.native_calls == 6 and torch.isfinite(out).all()


def test_compile_with_trace_scheduler_patches_the_step():
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile
    from sfast.libs.diffusers.scheduler import NativeDDIMStep
    cfg = U.tiny_config()
    pipe = MiniPipeline(U.build(cfg, seed=11, dtype=torch.float16, device=DEV))
    pipe.scheduler = _DDIMLike()
    config = CompilationConfig.Default()
    config.enable_cuda_graph = True
    config.trace_scheduler = True
    compile(pipe, config)
    assert isinstance(pipe.scheduler.step, NativeDDIMStep)
    # a scheduler of an unknown family is left alone
    other = types.SimpleNamespace(step=lambda *a, **k: None, config=types.SimpleNamespace())
    pipe2 = MiniPipeline(U.build(cfg, seed=11, dtype=torch.float16, device=DEV))
    pipe2.scheduler = other
    keep = other.step
    compile(pipe2, config)
    assert other.step is keep


def test_module_from_params_drives_compile_unet():
    """Bare state dict -> parameter-container module -> compile_unet(): same output as the engine built from the oracle module."""
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    from sfast.engine.unet_spec import module_from_params
    cfg = U.tiny_config()
    m = U.build(cfg, seed=13, dtype=torch.float16, device=DEV)
    params = {k: v.detach().clone() for k, v in m.named_parameters()}
    shell = module_from_params(dict(cfg), params)
    config = CompilationConfig.Default()
    config.enable_cuda_graph = True
    shell = compile_unet(shell, config)
    sample, ehs = _inputs(cfg, 2, seed=4, S=40)
    got = shell(sample, 321, encoder_hidden_states=ehs, return_dict=False)[0]
    want = _engine(m).forward(sample, 321, ehs)
    assert torch.equal(got, want)
    with pytest.raises(RuntimeError):
        shell.conv_in(sample)


def test_encoder_attention_mask_native_and_through_compile():
    """encoder_attention_mask no longer drops the UNet to eager: it is a static input of the native plan (attention bias kernel)."""
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    cfg = U.tiny_config()
    m = U.build(cfg, seed=14, dtype=torch.float16, device=DEV)
    ref = U.build(cfg, seed=14, dtype=torch.float32, device=DEV)
    sample, ehs = _inputs(cfg, 2, seed=5, S=77)
    mask = torch.ones(2, 77, device=DEV)
    mask[0, 30:] = 0
    mask[1, 9:] = 0
    with torch.no_grad():
        want = ref(sample.float(), 400, ehs.float(), encoder_attention_mask=mask).sample
    y = _engine(m).forward(sample, 400, ehs, encoder_attention_mask=mask)
    err = rel_l2(y, want)
    log_value("tiny unet encoder_attention_mask vs fp32 oracle", rel_l2=err)
    assert err < 4e-3, err
    config = CompilationConfig.Default()
    config.enable_cuda_graph = True
    cm = compile_unet(U.build(cfg, seed=14, dtype=torch.float16, device=DEV), config)
    out = cm(sample, 400, encoder_hidden_states=ehs, encoder_attention_mask=mask.bool(), return_dict=False)[0]
    assert not cm.forward._warned and torch.equal(out, y)
    out2 = cm(sample, 400, encoder_hidden_states=ehs, encoder_attention_mask=mask.bool(), return_dict=False)[0]  # graph replay
    assert torch.equal(out2, y) and len(cm.forward._cached) == 1


def test_self_attention_mask_native_and_through_compile():
    """the UNet-level `attention_mask` (keep-mask over the self-attention keys) is a static input of the native plan
    (the BIAS instantiation of the flash kernel on every attn1 launch) -- no whole-call eager fallback, no warning. Topology: one
    resolution level, the only kind on which diffusers' own forward accepts such a mask (every self-attention layer must have exactly
    `mask.shape[1]` tokens)."""
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    cfg = U.tiny_config(block_out_channels=(64,), down_block_types=("CrossAttnDownBlock2D",), up_block_types=("CrossAttnUpBlock2D",),
                        sample_size=16)
    m = U.build(cfg, seed=15, dtype=torch.float16, device=DEV)
    ref = U.build(cfg, seed=15, dtype=torch.float32, device=DEV)
    sample, ehs = _inputs(cfg, 2, seed=6, S=77)
    mask = torch.ones(2, 256, device=DEV)
    mask[0, 100:] = 0
    mask[1, ::3] = 0
    with torch.no_grad():
        want = ref(sample.float(), 400, ehs.float(), attention_mask=mask).sample
        plain = ref(sample.float(), 400, ehs.float()).sample
    y = _engine(m).forward(sample, 400, ehs, attention_mask=mask)
    err = rel_l2(y, want)
    log_value("tiny one-level unet attention_mask vs fp32 oracle", rel_l2=err, mask_effect=rel_l2(plain, want))
    assert err < 4e-3 and rel_l2(plain, want) > 1e-2, (err, rel_l2(plain, want))
    config = CompilationConfig.Default()
    config.enable_cuda_graph = True
    cm = compile_unet(U.build(cfg, seed=15, dtype=torch.float16, device=DEV), config)
    out = cm(sample, 400, encoder_hidden_states=ehs, attention_mask=mask.bool(), return_dict=False)[0]
    assert not cm.forward._warned and torch.equal(out, y)
    out2 = cm(sample, 400, encoder_hidden_states=ehs, attention_mask=mask.bool(), return_dict=False)[0]  # graph replay
    assert torch.equal(out2, y) and len(cm.forward._cached) == 1


@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_native_scheduler_step_matches_euler(pred, dtype):
    """trace_scheduler for EulerDiscreteScheduler (the default of diffusers' SDXL pipelines): `scale_model_input` and `step` as one
    kernel launch each, against the scheduler's own eager arithmetic over a whole schedule; step index, `is_scale_input_called`,
    tuple return and the fall-back for stochastic churn behave as in diffusers."""
    from test_host_cpu import _EulerRef
    from sfast.libs.diffusers.scheduler import NativeEulerScale, NativeEulerStep, patch_scheduler
    ref, nat = _EulerRef(pred, device=DEV), _EulerRef(pred, device=DEV)
    assert patch_scheduler(nat) and isinstance(nat.step, NativeEulerStep) and isinstance(nat.scale_model_input, NativeEulerScale)
    assert patch_scheduler(nat)  # idempotent
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(2, 4, 32, 32, generator=g) * float(ref.sigmas[0])).to(DEV, dtype)
    xr = x.clone()
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    for t in ref.timesteps:
        e = torch.randn(2, 4, 32, 32, generator=g).to(DEV, dtype)
        a, b = nat.scale_model_input(x, t), ref.scale_model_input(xr, t)
        assert rel_l2(a, b) < tol and nat.is_scale_input_called
        out = nat.step(e, t, x)
        want = ref.step(e, t, xr)
        assert nat.step_index == ref.step_index
        assert rel_l2(out.prev_sample, want.prev_sample) < tol, (int(t), rel_l2(out.prev_sample, want.prev_sample))
        assert rel_l2(out.pred_original_sample, want.pred_original_sample.to(dtype)) < tol
        x, xr = out.prev_sample, want.prev_sample
    assert nat.step.native_calls == len(ref.timesteps)
    # tuple form, and s_churn > 0 keeps the original method (stochastic: not a two-term update)
    n2 = _EulerRef(pred, device=DEV)
    patch_scheduler(n2)
    t0 = n2.timesteps[0]
    (p,) = n2.step(e, t0, x, return_dict=False)
    assert p.shape == x.shape and n2.step.native_calls == 1
    n2.step(e, n2.timesteps[1], x, s_churn=0.5)
    assert n2.step.native_calls == 1 and n2.step_index == 2


def _ip_models(cfg, seed, scale, image_embed_dim, dev=DEV):
    m = U.build(cfg, seed=seed, dtype=torch.float16, device=dev)
    m.load_ip_adapter(image_embed_dim=image_embed_dim, num_tokens=4, scale=scale, seed=seed + 1)
    ref = U.build(cfg, seed=seed, dtype=torch.float32, device=dev)
    ref.load_ip_adapter(image_embed_dim=image_embed_dim, num_tokens=4, scale=scale, seed=seed + 1)
    ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
    return m, ref


def test_ip_adapter_native_and_through_compile():
    """A UNet after `load_ip_adapter()` (encoder_hid_dim_type "ip_image_proj"; ImageProjection + to_k_ip / to_v_ip on every attn2) stays
    on the native plan: image embeddings are a static input, the decoupled image cross-attention is a second attention launch."""
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    cfg = U.tiny_config()
    m, ref = _ip_models(cfg, 21, 0.7, 32)
    sample, ehs = _inputs(cfg, 2, seed=6, S=77)
    g = torch.Generator().manual_seed(7)
    ie = torch.randn(2, 1, 32, generator=g).to(DEV, torch.float16)
    ie3 = torch.randn(2, 3, 32, generator=g).to(DEV, torch.float16)
    with torch.no_grad():
        want = ref(sample.float(), 400, ehs.float(), added_cond_kwargs={"image_embeds": [ie.float()]}).sample
        want3 = ref(sample.float(), 400, ehs.float(), added_cond_kwargs={"image_embeds": [ie3.float()]}).sample
        ref.set_ip_adapter_scale(0.0)
        text_only = ref(sample.float(), 400, ehs.float(), added_cond_kwargs={"image_embeds": [ie.float()]}).sample
    eng = _engine(m)
    y = eng.forward(sample, 400, ehs, added_cond_kwargs={"image_embeds": [ie]})
    y3 = eng.forward(sample, 400, ehs, added_cond_kwargs={"image_embeds": [ie3]})   # 12 image tokens
    log_value("tiny unet + ip-adapter vs fp32 oracle", rel_l2=rel_l2(y, want), rel_l2_3_images=rel_l2(y3, want3),
              adapter_effect=rel_l2(text_only, want))
    assert rel_l2(y, want) < 4e-3 and rel_l2(y3, want3) < 4e-3 and rel_l2(text_only, want) > 1e-2
    config = CompilationConfig.Default()
    config.enable_cuda_graph = True
    cm = compile_unet(m, config)
    kw = dict(encoder_hidden_states=ehs, added_cond_kwargs={"image_embeds": [ie]}, return_dict=False)
    out = cm(sample, 400, **kw)[0]
    assert not cm.forward._warned and torch.equal(out, y)
    assert torch.equal(cm(sample, 400, **kw)[0], y) and len(cm.forward._cached) == 1            # graph replay
    m.set_ip_adapter_scale(0.0)                                                                   # pipe.set_ip_adapter_scale(0)
    out0 = cm(sample, 400, **kw)[0]
    assert len(cm.forward._cached) == 2 and rel_l2(out0, text_only) < 4e-3 and not cm.forward._warned
    m.set_ip_adapter_scale(0.7)
    assert torch.equal(cm(sample, 400, **kw)[0], y) and len(cm.forward._cached) == 2             # back on the first graph
    # without image_embeds the original forward raises diffusers' error
    with pytest.raises(ValueError, match="image_embeds"):
        cm(sample, 400, encoder_hidden_states=ehs, return_dict=False)


def test_sd15_ip_adapter_parity(sd15):
    """Full-size SD1.5 + IP-Adapter geometry (CLIP image embedding 1024 -> 4 tokens of 768; 16 cross-attention blocks)."""
    import copy
    m = copy.deepcopy(sd15)
    m.load_ip_adapter(image_embed_dim=1024, num_tokens=4, scale=1.0, seed=5)
    sample, ehs = _inputs(U.SD15_CONFIG, 2, seed=3)
    ie = torch.randn(2, 1, 1024, generator=torch.Generator().manual_seed(1)).to(DEV, torch.float16)
    eng = _engine(m)
    y = eng.forward(sample, 981, ehs, added_cond_kwargs={"image_embeds": [ie]})
    with torch.no_grad():
        y16 = m(sample, 981, ehs, added_cond_kwargs={"image_embeds": [ie]}).sample
        ref = U.build("sd15", seed=0, dtype=torch.float32, device=DEV)
        ref.load_ip_adapter(image_embed_dim=1024, num_tokens=4, scale=1.0, seed=5)
        ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
        y32 = ref(sample.float(), 981, ehs.float(), added_cond_kwargs={"image_embeds": [ie.float()]}).sample
        ref.set_ip_adapter_scale(0.0)
        y32_text = ref(sample.float(), 981, ehs.float(), added_cond_kwargs={"image_embeds": [ie.float()]}).sample
        del ref
    e_engine, e_eager = rel_l2(y, y32), rel_l2(y16, y32)
    plan = next(iter(eng._plans.values()))
    log_value("sd15 B=2 + ip-adapter parity", engine_vs_fp32=e_engine, eager16_vs_fp32=e_eager, adapter_effect=rel_l2(y32_text, y32),
              launches=len(plan.ops))
    assert torch.isfinite(y).all() and e_engine < 2.5e-3 and e_engine < e_eager, (e_engine, e_eager)
    assert sum(1 for op in plan.ops if op.name.endswith(".ip_adapter.0")) == 16


def test_sdxl_style_controlnet_native_and_through_compile():
    """ControlNetModel with addition_embed_type "text_time" (SDXL ControlNets): native plan, eager and behind compile_controlnet-style
    wrapping, guess_mode included."""
    from oracle import controlnet_ref as CN
    from sfast.engine import ControlNetEngine
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    ccfg = CN.tiny_config(down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                          transformer_layers_per_block=(1, 2, 2), attention_head_dim=(1, 2, 2), use_linear_projection=True,
                          addition_embed_type="text_time", addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32)
    c16 = CN.build(ccfg, seed=33, dtype=torch.float16, device=DEV)
    c32 = CN.build(ccfg, seed=33, device=DEV)
    c32.load_state_dict({k: v.float() for k, v in c16.state_dict().items()})
    sample, ehs = _inputs(ccfg, 2, seed=2, S=77)
    cond = torch.rand(2, 3, 64, 64, device=DEV, dtype=torch.float16)
    added = dict(text_embeds=torch.randn(2, 64, device=DEV, dtype=torch.float16),
                 time_ids=torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * 2, device=DEV, dtype=torch.float16))
    with torch.no_grad():
        wd, wm = c32(sample.float(), 300, ehs.float(), cond.float(), guess_mode=True, conditioning_scale=0.9,
                     added_cond_kwargs={k: v.float() for k, v in added.items()}, return_dict=False)
    eng = ControlNetEngine.from_module(c16)
    down, mid = eng.forward(sample, 300, ehs, cond, conditioning_scale=0.9, guess_mode=True, added_cond_kwargs=added)
    errs = [rel_l2(a, b) for a, b in zip(down + [mid], list(wd) + [wm])]
    log_value("tiny sdxl-style controlnet vs fp32 oracle", worst_rel_l2=max(errs))
    assert max(errs) < 4e-3, errs
    config = CompilationConfig.Default()
    config.enable_cuda_graph = True
    cm = compile_unet(c16, config)
    for _ in range(2):
        d2, m2 = cm(sample, 300, encoder_hidden_states=ehs, controlnet_cond=cond, conditioning_scale=0.9, guess_mode=True,
                    added_cond_kwargs=added, return_dict=False)
        assert not cm.forward._warned and torch.equal(m2, mid) and all(torch.equal(a, b) for a, b in zip(d2, down))


@pytest.mark.parametrize("style", ["diffusers", "peft"])
def test_unfused_lora_native_and_switched_in_place(style):
    """A UNet with LoRA factors loaded and NOT fused (the reference's own LoRA test loads the adapter before compile(),
    /root/reference/tests/compilers/test_stable_diffusion_pipeline_compiler.py:327-328) runs on the native plan: one merge launch per
    step rebuilds the effective weights from the live tensors. cross_attention_kwargs["scale"] and the README's in-place adapter
    switch (README.md:228-265) act on the captured graph without re-capture."""
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    cfg = U.tiny_config()
    m = U.build(cfg, seed=31, dtype=torch.float16, device=DEV)
    ref = U.build(cfg, seed=31, dtype=torch.float32, device=DEV)
    for mm in (m, ref):
        mm.load_lora(rank=8, network_alpha=4.0, seed=9, style=style, up_scale=0.05)
    ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
    sample, ehs = _inputs(cfg, 2, seed=8, S=77)
    with torch.no_grad():
        want = ref(sample.float(), 300, ehs.float()).sample
        want_half = ref(sample.float(), 300, ehs.float(), cross_attention_kwargs={"scale": 0.5}).sample
        want_off = ref(sample.float(), 300, ehs.float(), cross_attention_kwargs={"scale": 0.0}).sample
    eng = _engine(m)
    y = eng.forward(sample, 300, ehs)
    log_value(f"tiny unet + un-fused lora ({style}) vs fp32 oracle", rel_l2=rel_l2(y, want), adapter_effect=rel_l2(want_off, want), linears=len(eng.lora))
    assert rel_l2(y, want) < 4e-3 and rel_l2(want_off, want) > 1e-2
    config = CompilationConfig.Default()
    config.enable_cuda_graph = True
    cm = compile_unet(m, config)
    out = cm(sample, 300, encoder_hidden_states=ehs, return_dict=False)[0]
    assert not cm.forward._warned and torch.equal(out, y)
    half = cm(sample, 300, encoder_hidden_states=ehs, cross_attention_kwargs={"scale": 0.5}, return_dict=False)[0]
    off = cm(sample, 300, encoder_hidden_states=ehs, cross_attention_kwargs={"scale": 0.0}, return_dict=False)[0]
    assert rel_l2(half, want_half) < 4e-3 and rel_l2(off, want_off) < 4e-3 and len(cm.forward._cached) == 1     # one graph, three scales
    # switch the adapter: other factors into the same tensors
    g = torch.Generator().manual_seed(77)
    sd, sdr = m.state_dict(), ref.state_dict()
    with torch.no_grad():
        for k in [k for k in sd if "lora" in k]:
            sd[k].copy_((torch.randn(sd[k].shape, generator=g) * (0.05 if ("up" in k or "lora_B" in k) else sd[k].shape[1] ** -0.5)).to(DEV))
            sdr[k].copy_(sd[k].float())
        want2 = ref(sample.float(), 300, ehs.float()).sample
    out2 = cm(sample, 300, encoder_hidden_states=ehs, return_dict=False)[0]
    assert rel_l2(out2, want2) < 4e-3 and rel_l2(want2, want) > 1e-2 and len(cm.forward._cached) == 1


def test_sd15_lora_step_cost(sd15):
    """Full-size SD1.5 with rank-16 LoRA on all 128 attention projections: parity, and what the per-step merge launch costs."""
    import copy
    m = copy.deepcopy(sd15)
    m.load_lora(rank=16, network_alpha=16.0, seed=2, up_scale=0.02)
    sample, ehs = _inputs(U.SD15_CONFIG, 2, seed=3)
    eng = _engine(m)
    assert len(eng.lora) == 128
    y = eng.forward(sample, 981, ehs)
    with torch.no_grad():
        y16 = m(sample, 981, ehs).sample
        ref = U.build("sd15", seed=0, dtype=torch.float32, device=DEV)
        ref.load_lora(rank=16, network_alpha=16.0, seed=2, up_scale=0.02)
        ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
        y32 = ref(sample.float(), 981, ehs.float()).sample
        y32_off = ref(sample.float(), 981, ehs.float(), cross_attention_kwargs={"scale": 0.0}).sample
        del ref
    plan = next(iter(eng._plans.values()))
    merge = plan.ops[0]
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.current_stream().cuda_stream
    merge.launch(s)
    a.record()
    for _ in range(20):
        merge.launch(s)
    b.record()
    b.synchronize()
    e_engine, e_eager = rel_l2(y, y32), rel_l2(y16, y32)
    log_value("sd15 B=2 + un-fused lora r16 parity", engine_vs_fp32=e_engine, eager16_vs_fp32=e_eager, adapter_effect=rel_l2(y32_off, y32),
              merge_launch_us=a.elapsed_time(b) * 1e3 / 20, merged_mbytes=merge.bytes / 2e6)
    assert merge.name.startswith("lora.merge[128") and torch.isfinite(y).all() and e_engine < 2.5e-3, (e_engine, e_eager)


def test_ip_adapter_plus_through_compile_projects_once():
    """IP-Adapter Plus: the resampler is not a plan op -- the compiled forward runs the module's encoder_hid_proj once per distinct
    image_embeds and hands the plan the projected tokens; replays with the same embeddings do not run it again."""
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    cfg = U.tiny_config()
    m = U.build(cfg, seed=41, dtype=torch.float16, device=DEV)
    ref = U.build(cfg, seed=41, dtype=torch.float32, device=DEV)
    for mm in (m, ref):
        mm.load_ip_adapter_plus(image_embed_dim=32, num_tokens=6, scale=0.7, seed=42)
    ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
    sample, ehs = _inputs(cfg, 2, seed=9, S=77)
    g = torch.Generator().manual_seed(10)
    patches = torch.randn(2, 1, 9, 32, generator=g).to(DEV, torch.float16)
    with torch.no_grad():
        want = ref(sample.float(), 400, ehs.float(), added_cond_kwargs={"image_embeds": [patches.float()]}).sample
        ref.set_ip_adapter_scale(0.0)
        text_only = ref(sample.float(), 400, ehs.float(), added_cond_kwargs={"image_embeds": [patches.float()]}).sample
    config = CompilationConfig.Default()
    config.enable_cuda_graph = True
    cm = compile_unet(m, config)
    calls = []
    hook = m.encoder_hid_proj.register_forward_hook(lambda *a: calls.append(1))
    kw = dict(encoder_hidden_states=ehs, added_cond_kwargs={"image_embeds": [patches]}, return_dict=False)
    out = cm(sample, 400, **kw)[0]
    out2 = cm(sample, 380, **kw)[0]
    assert not cm.forward._warned and len(calls) == 1 and len(cm.forward._cached) == 1
    log_value("tiny unet + ip-adapter plus (external projection) vs fp32 oracle", rel_l2=rel_l2(out, want), adapter_effect=rel_l2(text_only, want))
    assert rel_l2(out, want) < 4e-3 and rel_l2(text_only, want) > 1e-2 and torch.isfinite(out2).all()
    patches.mul_(-1.0)                                                            # new image: version counter moved -> projected again
    with torch.no_grad():
        ref.set_ip_adapter_scale(0.7)
        want_neg = ref(sample.float(), 400, ehs.float(), added_cond_kwargs={"image_embeds": [patches.float()]}).sample
    out3 = cm(sample, 400, **kw)[0]
    hook.remove()
    assert len(calls) == 2 and rel_l2(out3, want_neg) < 4e-3
