"""Host-side pieces that need no GPU: graph-cache keys, tree copies, config surface, batch sharding
and the weight broadcast of the multi-GPU path (world_size 2 over gloo)."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_hash_arg_semantics():
    # /root/reference/src/sfast/cuda/graphs.py:225-241
    from sfast.cuda.graphs import hash_arg
    a = torch.zeros(2, 3)
    assert hash_arg(a) == hash_arg(torch.ones(2, 3))           # tensors hash by type/shape, not value
    assert hash_arg(a) != hash_arg(torch.zeros(2, 4))
    assert hash_arg(a) != hash_arg(a.half())
    assert hash_arg(torch.tensor(3)) != hash_arg(torch.tensor(4))  # CPU scalars hash by value
    assert hash_arg((1, "x", [a])) == hash_arg((1, "x", [a.clone()]))
    assert hash_arg({"k": a, "j": 2}) == hash_arg({"j": 2, "k": a})
    assert hash_arg(object()) == object


def test_tree_copy_roundtrip():
    from sfast.utils.copy import tree_copy, tree_copy_
    src = {"a": torch.arange(4.0), "b": (torch.ones(2), [torch.zeros(1), 5]), "c": "s"}
    dst = tree_copy(src)
    assert dst["a"] is not src["a"] and torch.equal(dst["a"], src["a"]) and dst["b"][1][1] == 5
    src["a"].add_(1)
    tree_copy_(dst, src)
    assert torch.equal(dst["a"], src["a"])

    import collections
    import dataclasses
    import pytest
    from sfast.utils.copy import can_be_perfectly_copied

    @dataclasses.dataclass
    class Out(collections.OrderedDict):  # the shape of diffusers' BaseOutput: a dict AND a dataclass
        sample: torch.Tensor = None

    Pair = collections.namedtuple("Pair", "x y")
    tree = [Out(sample=torch.ones(3)), Pair(torch.zeros(2), 7)]
    clone = tree_copy(tree, detach=True)
    assert isinstance(clone[0], Out) and torch.equal(clone[0].sample, tree[0].sample) and clone[0].sample is not tree[0].sample
    assert isinstance(clone[1], Pair) and clone[1].y == 7
    tree[0].sample.mul_(3)
    tree_copy_(clone, tree)
    assert torch.equal(clone[0].sample, torch.full((3,), 3.0))
    with pytest.raises(ValueError):
        tree_copy_([torch.zeros(1)], [torch.zeros(1), torch.zeros(1)])
    with pytest.raises(ValueError):
        tree_copy_({"a": 1}, {"a": "1"})
    assert can_be_perfectly_copied(tree) and not can_be_perfectly_copied([object()])


def test_compilation_config_surface():
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile, compile_unet, compile_vae  # noqa: F401
    import sfast.compilers.stable_diffusion_pipeline_compiler as alias
    c = CompilationConfig.Default()
    names = ["memory_format", "enable_jit", "enable_jit_freeze", "preserve_parameters", "enable_cnn_optimization",
             "enable_fused_linear_geglu", "prefer_lowp_gemm", "enable_xformers", "enable_cuda_graph", "enable_triton",
             "trace_scheduler"]
    assert [f for f in c.__dataclass_fields__] == names
    assert c.enable_jit and c.preserve_parameters and not c.enable_cuda_graph
    assert alias.compile is compile and alias.CompilationConfig is CompilationConfig


def test_compile_unet_on_cpu_keeps_eager_forward():
    # no GPU -> the native engine is not engaged and nothing silently emulates it
    from oracle import unet_ref as U
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile_unet
    m = U.build("tiny", seed=0)
    fwd = m.forward
    m2 = compile_unet(m, CompilationConfig.Default())
    assert m2 is m and m.forward == fwd and not hasattr(m, "_sfast_engine")
    assert m.conv_in.weight.is_contiguous()  # default memory_format on a CPU-only box is contiguous


def test_shard_batch():
    from sfast.engine.replicas import shard_batch
    assert [shard_batch(64, r, 8) for r in range(8)] == [(8 * r, 8 * r + 8) for r in range(8)]
    parts = [shard_batch(10, r, 4) for r in range(4)]
    assert parts == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_batch(1, 0, 1) == (0, 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sfast.engine.replicas import broadcast_parameters, gather_latents, shard_batch
        torch.manual_seed(100 + rank)  # different weights on every rank before the broadcast
        params = {
            "conv.weight": torch.randn(8, 4, 3, 3).contiguous(memory_format=torch.channels_last),
            "lin.weight": torch.randn(16, 8),
            "lin.bias": torch.randn(16),
            "norm.weight": torch.randn(8).half(),
        }
        ptrs = {k: v.data_ptr() for k, v in params.items()}
        nbytes = broadcast_parameters(params, src=0, bucket_bytes=256)  # tiny buckets: several collectives
        assert all(params[k].data_ptr() == ptrs[k] for k in params)  # in place: captured pointers stay valid
        assert params["conv.weight"].is_contiguous(memory_format=torch.channels_last)
        torch.manual_seed(100)
        want = {"conv.weight": torch.randn(8, 4, 3, 3), "lin.weight": torch.randn(16, 8), "lin.bias": torch.randn(16),
                "norm.weight": torch.randn(8).half()}
        ok = all(torch.equal(params[k], want[k]) for k in params)
        lo, hi = shard_batch(6, rank, world)
        local = torch.full((hi - lo, 2), float(rank))
        outs = gather_latents(local, dst=0)
        if rank == 0:
            ok = ok and len(outs) == world and all(float(o[0, 0]) == r for r, o in enumerate(outs))
        q.put((rank, ok, nbytes))
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_and_sharding_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert res[0][2] == res[1][2] > 0


def test_native_unet_forward_fallback_passes_every_given_argument():
    """A call the engine does not cover must reach the ORIGINAL forward with everything the caller passed -- dropping e.g.
    `down_intrablock_additional_residuals` (T2I-Adapter) would silently change the result."""
    from types import SimpleNamespace
    from sfast.compilers.diffusion_pipeline_compiler import _NativeUNetForward
    seen = {}

    def orig(sample, timestep, **kw):
        seen.update(kw)
        return "eager"

    fwd = _NativeUNetForward(module=None, engine=SimpleNamespace(dtype=torch.float16, device=torch.device("cpu")),
                             orig_forward=orig, enable_graph=False)
    x = torch.zeros(1, 4, 8, 8)                       # a CPU tensor: never the native path
    intr = [torch.ones(1)]
    out = fwd(x, 10, encoder_hidden_states=torch.zeros(1, 77, 8), down_intrablock_additional_residuals=intr,
              cross_attention_kwargs={"scale": 0.5}, return_dict=False)
    assert out == "eager"
    assert seen["down_intrablock_additional_residuals"] is intr and seen["cross_attention_kwargs"] == {"scale": 0.5}
    assert seen["return_dict"] is False and "class_labels" not in seen and "encoder_attention_mask" not in seen


def test_bench_contract_defaults_and_self_launch(monkeypatch):
    """bench.py: no flags = 1 GPU with a K / W that finish in minutes; `--gpus N` typed without the launcher re-launches
    itself as N ranks on 127.0.0.1 (the driver does that itself and sets RANK, in which case nothing is re-launched)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup, a.config, a.images) == (1, 200, 150, "sd15", 1)   # warm-up: three 50-step images (the reference's protocol)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "svd"])
    assert bench.parse().warmup == 20   # 0.2 s per step
    monkeypatch.setattr(sys, "argv", ["bench.py", "--warmup", "7"])
    assert bench.parse().warmup == 7
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    cmd = bench.torchrun_argv(4, ["--gpus", "4", "--steps", "7"], port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    # a free port is picked when none is given
    assert int(bench.torchrun_argv(2, [])[bench.torchrun_argv(2, []).index("--master-port") + 1]) > 0


def _loop_worker(rank, world, port, q):
    """Every rank: tiny UNet engine on the ABI emulator, weights broadcast from rank 0, its own image (seed 1234 + rank) through
    three fused CFG + DDIM iterations of DenoiseLoop -- the per-GPU loop of bench.py --gpus N, on CPU."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from abi_emulator import EmuLib, emulated_denoise_loop, EmuHost
        from oracle import unet_ref as U
        from sfast.engine import UNet2DEngine, autotune
        from sfast.engine.replicas import broadcast_parameters, gather_latents, share_tune_cache
        cfg = U.tiny_config()
        m = U.build(cfg, seed=50 + rank, dtype=torch.float16)  # different weights per rank until the broadcast
        params = {k: (v.data.contiguous(memory_format=torch.channels_last) if v.ndim == 4 else v.data) for k, v in m.named_parameters()}
        broadcast_parameters(params, src=0)
        if rank == 0:
            autotune.import_cache({"gfx950|f16|gemm|1x2x3|(0, 0, 1)": [3, 1]})
        got = share_tune_cache(src=0)
        eng = UNet2DEngine(m.config, params, device=torch.device("cpu"), dtype=torch.float16, _host=EmuHost())
        loop = emulated_denoise_loop(eng, images=1, height=16, width=16, ctx_len=20, guidance=7.5, num_steps=50)
        g = torch.Generator().manual_seed(1234 + rank)
        lat = torch.randn(1, 4, 16, 16, generator=g).half()
        ehs = torch.randn(2, 20, cfg["cross_attention_dim"], generator=g).half()
        loop.set_inputs(lat, ehs)
        for i in range(3):
            loop.step(i)
        outs = gather_latents(loop.latents.float(), dst=0)
        q.put((rank, got, [o.clone() for o in outs] if rank == 0 else None, autotune.export_cache().get("gfx950|f16|gemm|1x2x3|(0, 0, 1)")))
    finally:
        dist.destroy_process_group()


def test_denoise_loop_replicas_gloo_world2(built_lib):
    """Rank r's latents after three fused steps == a single-process run with rank 0's weights and seed 1234 + r."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1][1] >= 1 and res[1][3] == [3, 1]  # rank 1 received rank 0's kernel choices
    from abi_emulator import EmuHost, EmuLib
    from oracle import unet_ref as U
    from oracle.ops_ref import cfg_ddim_ref, ddim_schedule
    from sfast.engine import UNet2DEngine
    cfg = U.tiny_config()
    m = U.build(cfg, seed=50, dtype=torch.float16)
    eng = UNet2DEngine.from_module(m, _host=EmuHost())
    ts, coefs = ddim_schedule(50)
    for r in range(2):
        g = torch.Generator().manual_seed(1234 + r)
        lat = torch.randn(1, 4, 16, 16, generator=g).half()
        ehs = torch.randn(2, 20, cfg["cross_attention_dim"], generator=g).half()
        for i in range(3):
            eps = eng.forward(torch.cat([lat, lat]), float(ts[i]), ehs)
            c32 = torch.tensor(coefs[i], dtype=torch.float32).tolist()  # the loop keeps its coefficient table in float32
            lat = cfg_ddim_ref(eps.flatten(), lat.flatten(), c32, 7.5).to(torch.float16).reshape(lat.shape)
        assert torch.equal(res[0][2][r], lat.float()), r


def test_packaged_tune_cache_is_well_formed():
    """The (variant, split-K) choices shipped with the package (measured on an MI355X, profiles/r02_tune_cache_run6.json): every entry
    names a known kernel variant and a split the planner accepts; importing them is what spares a fresh process the timing runs."""
    import json
    from sfast.engine import autotune
    with open(autotune.PACKAGED_CACHE) as f:
        d = json.load(f)
    assert len(d) > 100
    n_pk = n_pp = 0
    for k, (v, s) in d.items():
        arch, dt, kind, mnk, epi = k.split("|")[:5]
        packed = k.endswith("|pk")   # the same problem with a packed copy of the weight at hand: pipe-4 variants are candidates too
        assert len(k.split("|")) == (6 if packed else 5), k
        assert arch == "gfx950" and dt in ("f16", "bf16") and kind in ("gemm", "conv"), k
        assert len(mnk.split("x")) == 3
        geglu = kind == "gemm" and epi.startswith("(1,")
        allowed = autotune.GEGLU_VARIANTS if geglu else (autotune.VARIANTS + autotune.CONV_PATCH_VARIANTS + (autotune.PK_VARIANTS if packed else ()))
        M, N, K = (int(x) for x in mnk.split("x"))
        allowed = allowed + autotune._pp_candidates(M, N, K, geglu)   # round 6: the 256-row tiles, only where their tiles fill >= 96 CUs
        assert v in allowed and s in autotune.SPLITS, (k, v, s)
        n_pk += packed
        n_pp += v >= 50
    assert n_pk > 50   # round 4: the SD1.5 / SDXL problems were re-timed with the packed-weight pipe among the candidates
    assert n_pp > 50   # round 6: ... and the large-M problems (8 images per GPU, SDXL 128^2, SVD-XT, VAE) against the 256-row tiles
    n0 = len(autotune.export_cache())
    assert autotune.import_cache(d) >= 0 and len(autotune.export_cache()) >= max(n0, len(d))


class _EulerRef:
    """The arithmetic of diffusers' EulerDiscreteScheduler (s_churn = 0), restated for the tests: Karras-free sigma schedule from
    SD's scaled-linear betas, `scale_model_input`, and the first-order step for epsilon / v-prediction models."""
    _sfast_euler_like = True
    init_noise_sigma = 1.0

    def __init__(self, prediction_type="epsilon", n=10, device="cpu"):
        import types
        self.config = types.SimpleNamespace(prediction_type=prediction_type, num_train_timesteps=1000)
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2
        acp = torch.cumprod(1.0 - betas, dim=0)
        sig = ((1 - acp) / acp) ** 0.5
        ts = torch.linspace(999, 0, n).round().long()
        self.timesteps = ts.to(device)
        self.sigmas = torch.cat([sig[ts], torch.zeros(1, dtype=torch.float64)]).to(torch.float32).to(device)
        self._step_index = None
        self.is_scale_input_called = False

    @property
    def step_index(self):
        return self._step_index

    def _init_step_index(self, timestep):
        self._step_index = int((self.timesteps == int(timestep)).nonzero()[0])

    def scale_model_input(self, sample, timestep):
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self._step_index]
        self.is_scale_input_called = True
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, generator=None, return_dict=True):
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self._step_index]
        x = sample.to(torch.float32)
        if self.config.prediction_type == "epsilon":
            x0 = x - sigma * model_output
        else:
            x0 = model_output * (-sigma / (sigma ** 2 + 1) ** 0.5) + (x / (sigma ** 2 + 1))
        derivative = (x - x0) / sigma
        dt = self.sigmas[self._step_index + 1] - sigma
        prev = (x + derivative * dt).to(model_output.dtype)
        self._step_index += 1
        import types
        return (prev,) if not return_dict else types.SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
def test_euler_coefficient_tables_reproduce_the_scheduler_arithmetic(pred):
    """trace_scheduler for EulerDiscreteScheduler (SDXL's default): the per-step-index (A, B) rows of `prev = A x + B model_output`
    and the input scale, checked against the scheduler's own arithmetic on the host (the kernel that applies them is GPU-tested)."""
    from sfast.libs.diffusers.scheduler import _EulerTables, euler_like, ddim_like
    s = _EulerRef(pred)
    assert euler_like(s) and not ddim_like(s)
    step_tab, scale_tab = _EulerTables(s).get("cpu")
    assert step_tab.shape == (10, 2) and scale_tab.shape == (10, 2) and step_tab.dtype == torch.float32
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 8, generator=g)
    for i, t in enumerate(s.timesteps):
        e = torch.randn(2, 4, 8, 8, generator=g)
        want_in = s.scale_model_input(x, t)
        torch.testing.assert_close(scale_tab[i, 0] * x, want_in, rtol=1e-5, atol=1e-6)
        assert float(scale_tab[i, 1]) == 0.0
        want = s.step(e, t, x).prev_sample
        torch.testing.assert_close(step_tab[i, 0] * x + step_tab[i, 1] * e, want, rtol=1e-4, atol=1e-4)
        x = want
    assert s.step_index == 10
    # a multistep solver carries the same attributes with different arithmetic: never recognised by attributes alone
    other = _EulerRef(pred)
    other._sfast_euler_like = False
    assert not euler_like(other)


def test_ddim_is_recognised_by_class_never_by_attributes():
    """PNDMScheduler (the SD1.5 default), LCMScheduler, TCDScheduler and DDIMInverseScheduler carry DDIM's
    attribute surface (`alphas_cumprod`, `final_alpha_cumprod`, `step(eta=..., generator=...)`) with multistep / stochastic /
    inverse arithmetic. `patch_scheduler` must leave them eager (the reference's lazy_trace keeps each scheduler's own math) and
    take only DDIMScheduler / DDIMParallelScheduler (by class name in the MRO) or a declared look-alike."""
    import types
    from sfast.libs.diffusers.scheduler import NativeDDIMStep, ddim_like, patch_scheduler

    def make(name, bases=(), **attrs):
        def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None, variance_noise=None,
                 return_dict=True):
            return ("own arithmetic",)
        cls = type(name, bases, dict(step=step, **attrs))
        s = cls()
        s.config = types.SimpleNamespace(num_train_timesteps=1000, prediction_type="epsilon", clip_sample=False, thresholding=False)
        s.alphas_cumprod = torch.linspace(0.999, 0.01, 1000)
        s.final_alpha_cumprod = torch.tensor(1.0)
        s.num_inference_steps = 50
        return s

    for name in ("PNDMScheduler", "LCMScheduler", "TCDScheduler", "DDIMInverseScheduler", "DPMSolverMultistepScheduler"):
        s = make(name)
        keep = s.step
        assert not ddim_like(s), name
        assert patch_scheduler(s) is False and s.step == keep, name
        assert s.step(None, 0, None) == ("own arithmetic",)
    for name in ("DDIMScheduler", "DDIMParallelScheduler"):
        s = make(name)
        assert ddim_like(s) and patch_scheduler(s) and isinstance(s.step, NativeDDIMStep), name
    base = type("DDIMScheduler", (), {})
    sub = make("MyTunedDDIM", (base,))
    assert ddim_like(sub)                                  # subclass of DDIMScheduler
    assert ddim_like(make("Custom", _sfast_ddim_like=True))  # declared look-alike
    clip = make("DDIMScheduler")
    clip.config.clip_sample = True
    assert not ddim_like(clip)                             # clipping is not the two-term linear form


def test_euler_tables_follow_a_rebound_schedule_of_equal_length():
    """the table key must not depend on id() / data_ptr() of a freed tensor. Rebinding `sigmas` to a new tensor of the same
    length (custom sigmas, karras toggled) rebuilds the rows; in-place edits (version bump) do too; an unchanged schedule does not."""
    from sfast.libs.diffusers.scheduler import _EulerTables
    s = _EulerRef("epsilon")
    tabs = _EulerTables(s)
    a0, _ = tabs.get("cpu")
    a0 = a0.clone()
    assert tabs.get("cpu")[0] is tabs.get("cpu")[0]        # cached while the schedule object is unchanged
    n = len(s.sigmas)
    for _ in range(4):                                     # allocate / free same-sized tensors: ids and storage get recycled
        s.sigmas = (torch.rand(n) + 0.5).sort(descending=True).values
        want = (s.sigmas[1:] - s.sigmas[:-1]).to(torch.float32)
        got, _ = tabs.get("cpu")
        torch.testing.assert_close(got[:, 1], want, rtol=1e-6, atol=1e-7)
    s.sigmas.mul_(2.0)                                     # in-place edit of the live schedule
    got, _ = tabs.get("cpu")
    torch.testing.assert_close(got[:, 1], (s.sigmas[1:] - s.sigmas[:-1]).to(torch.float32), rtol=1e-6, atol=1e-7)
    assert not torch.equal(got, a0)


def test_auto_graph_compiler_walk_matches_patch_module():
    """the reference's patch_module (utils/patch.py:1-19) patches an accepted ROOT and still walks its children; an accepted
    child is patched and not descended into. AutoGraphCraphCompiler rejects keyword arguments it could only fail on later."""
    import torch.nn as nn
    from sfast.cuda.graphs import AutoGraphCraphCompiler, _LazyCompiledForward, apply_auto_graph_compiler_to_all_modules

    def wrapped(m):
        return isinstance(m.forward, _LazyCompiledForward)

    net = nn.Sequential(nn.Linear(4, 4), nn.Sequential(nn.Linear(4, 4), nn.ReLU()))
    apply_auto_graph_compiler_to_all_modules(net)                      # default filter: accepts everything
    assert wrapped(net) and wrapped(net[0]) and wrapped(net[1])        # root AND its direct children
    assert not wrapped(net[1][0])                                      # ... but nothing below an accepted child

    net = nn.Sequential(nn.Linear(4, 4), nn.Sequential(nn.Linear(4, 4), nn.ReLU()))
    apply_auto_graph_compiler_to_all_modules(net, filter_func=lambda stack: stack[-1][0] is None or isinstance(stack[-1][1], nn.Linear))
    assert wrapped(net) and wrapped(net[0]) and not wrapped(net[1]) and wrapped(net[1][0]) and not wrapped(net[1][1])

    net = nn.Sequential(nn.Linear(4, 4))
    apply_auto_graph_compiler_to_all_modules(net, recursive=False)
    assert wrapped(net) and not wrapped(net[0])
    with pytest.raises(TypeError):
        AutoGraphCraphCompiler(warmups=1)


def test_bench_quotes_pmc_traffic_only_with_matching_provenance(tmp_path, monkeypatch):
    """BENCH_r04's roofline.traffic silently came from a round-3 file. bench.roofline_from() quotes a PMC traffic
    file only when its `_meta` block names the kernel choices in use (sha256 of the packaged tune cache) and says which round / commit /
    command produced it; otherwise `traffic` is null and `traffic_note` says why. `frac_kernel_only` (rocprofv3 per-dispatch average
    from the same file) sits beside `frac_interval` (live HIP events around the C-ABI call, reduce launch included)."""
    import hashlib
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sym_variant = "igemm_conv_f16[128x128,split=3,ws4]+gnstats@xcd1x8x1"
    rows = [dict(kind="conv3x3", name=f"c{i}", kernel=sym_variant, seconds=40e-6, flops=18.874e9, bytes=25.5e6) for i in range(4)]
    rows.append(dict(kind="ln", name="ln", kernel="ln_rows", seconds=5e-6, flops=0.0, bytes=10.5e6))
    sym = bench.kernel_symbol(sym_variant)
    with open(os.path.join(root, "stable-fast_amd", "sfast", "engine", "tune_gfx950.json"), "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()[:16]
    path = tmp_path / "r05_pmc_traffic_by_symbol.json"
    monkeypatch.setenv("SFAST_TRAFFIC_PROFILE", str(path))

    def write(meta):
        doc = {sym: dict(bytes_per_launch=85.6e6, avg_us=36.0, launches=192, avg_us_trace=31.2, launches_trace=48)}
        if meta is not None:
            doc["_meta"] = meta
        path.write_text(json.dumps(doc))

    write(None)                                             # a pre-round-5 file: no provenance
    roof, _, _ = bench.roofline_from(rows, None, "sd15")
    assert roof["kernel"] == sym and roof["traffic"] is None and "no _meta" in roof["traffic_note"] and roof["frac_kernel_only"] is None
    write(dict(round=5, commit="abc1234", tune_cache_sha256="0" * 16, command="python bench.py"))   # other kernel choices
    roof, _, _ = bench.roofline_from(rows, None, "sd15")
    assert roof["traffic"] is None and "other kernel choices" in roof["traffic_note"]
    write(dict(round=5, commit="abc1234", tune_cache_sha256=sha, command="python bench.py"))
    roof, _, _ = bench.roofline_from(rows, None, "sd15")
    assert roof["traffic"] == 85.6e6 and "round 5, commit abc1234" in roof["traffic_source"]
    assert abs(roof["traffic_over_algorithmic"] - 85.6 / 25.5) < 1e-6
    assert abs(roof["frac_interval"] - 18.874e9 / 40e-6 / 1e12 / 2500.0) < 1e-9 and roof["frac"] == roof["frac_interval"]
    assert abs(roof["frac_kernel_only"] - 18.874e9 / 31.2e-6 / 1e12 / 2500.0) < 1e-9   # what rocprofv3 --stats reproduces (0.24 in round 4)


def test_unwanted_graphs_are_retired_oldest_first_after_a_synchronise(monkeypatch):
    """Round 5 (the fix of the round-4 crash): a hipGraphExec that is no longer wanted is never destroyed in the same breath as its last
    replay. `retire_graph` parks it; `_trim_retired` -- called from capture_plan_graph's own context only -- lets the oldest entries go
    once more than 2 x _RETIRED_KEEP have piled up, and only after it synchronised every device that owns one."""
    import sfast.engine.unet2d as E
    synced = []
    monkeypatch.setattr(E.torch.cuda, "synchronize", lambda dev=None: synced.append(dev))
    monkeypatch.setattr(E, "_RETIRED", type(E._RETIRED)())
    dropped = []

    class G:
        def __init__(self, i):
            self.i = i

        def __del__(self):
            dropped.append((self.i, len(synced)))

    for i in range(2 * E._RETIRED_KEEP):
        E.retire_graph(G(i), "dev0")
        E._trim_retired()
    assert not dropped and not synced and len(E._RETIRED) == 2 * E._RETIRED_KEEP      # nothing leaves while the queue is short
    E.retire_graph(G(99), "dev1")
    E._trim_retired()
    assert sorted(synced) == ["dev0", "dev1"]                                              # every owning device, before anything is dropped
    assert [i for i, _ in dropped] == list(range(E._RETIRED_KEEP + 1))                      # oldest first ...
    assert all(n == 2 for _, n in dropped) and len(E._RETIRED) == E._RETIRED_KEEP          # ... and only after both synchronisations
    assert E._RETIRED[-1][0].i == 99


def test_real_weight_hook_round_trip(tmp_path):
    """`load_params` (what bench.py --weights and SFAST_SD15_DIR use) reads a diffusers-layout checkpoint into the
    engines' {state-dict name: tensor} form and refuses one that does not match the config. No real checkpoint is reachable offline,
    so the file is written here from the oracle module of the tiny config -- names come from the MODULE, shapes from the inventory."""
    from safetensors.torch import save_file
    from oracle import unet_ref as U
    from sfast.engine.unet_spec import find_unet_weights, load_params
    cfg = U.tiny_config()
    m = U.build(cfg, seed=3)
    sd = {k: v.detach().clone().contiguous() for k, v in m.state_dict().items()}
    (tmp_path / "unet").mkdir()
    f = tmp_path / "unet" / "diffusion_pytorch_model.safetensors"
    save_file(sd, str(f))
    assert find_unet_weights(str(tmp_path)) == str(f) and find_unet_weights(str(f)) == str(f)
    p = load_params(str(tmp_path), cfg, dtype=torch.float32, device="cpu")
    assert set(p) == set(sd) and all(torch.equal(p[k], sd[k]) for k in sd)
    assert all(v.is_contiguous(memory_format=torch.channels_last) for v in p.values() if v.ndim == 4)
    m2 = U.build(cfg, seed=99)
    m2.load_state_dict(p)   # the loaded dict drops into the oracle (and, on a GPU, into UNet2DEngine(cfg, params))
    bad = dict(sd)
    bad.pop("conv_in.bias")
    bad["conv_out.weight"] = torch.zeros(1, 2, 3, 3)
    save_file(bad, str(f))
    with pytest.raises(ValueError, match="1 missing.*conv_in.bias.*1 shape mismatches"):
        load_params(str(tmp_path), cfg, dtype=torch.float32, device="cpu")
    with pytest.raises(FileNotFoundError):
        find_unet_weights(str(tmp_path / "nothing"))
