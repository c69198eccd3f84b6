"""MI355X-native executor of the diffusion UNet forward -- the hot path behind
`sfast.compilers.compile()`.

The reference reaches its steady state (one CUDA-graph replay per denoise step,
/root/reference/src/sfast/cuda/graphs.py:153-157) by tracing the diffusers module with TorchScript,
rewriting the traced graph with pattern passes (/root/reference/src/sfast/compilers/
diffusion_pipeline_compiler.py:193-252) and capturing the result. This engine owns the forward
instead: from the module's `config` + live parameters it builds, per input signature, a static
PLAN -- an ordered list of C-ABI kernel launches on preallocated NHWC buffers -- and captures that
plan into a hipGraph. The fusion spec is the reference's pass list, applied by construction:

    GroupNorm+SiLU fused, LayerNorm, Linear+GEGLU (dual GEMM), conv+bias+(time-emb | residual) add,
    Linear+bias+residual add, fused QKV projection feeding [B,S,H,D] flash attention in place,
    nearest-2x upsample and the up-block channel concat folded into the conv / GroupNorm gathers
    (never materialised), NHWC everywhere so [B,C,H,W] <-> [B,HW,C] is free.

Weights are read from the live parameter storage at every launch (the reference's
`preserve_parameters=True` / LoRA in-place update contract, README.md:228-265) -- with ONE exception,
the packed-weight pipe (pipe 4, on by default): a GEMM / conv whose autotuned kernel reads a packed
copy of its weight depends on `sync_packed()`, which every engine `forward`, every compiled forward and
`DenoiseLoop.set_inputs / step / refresh_text_kv` call before they launch; it re-packs a parameter
whose autograd VERSION COUNTER moved since the copy was made. In-place updates through the parameter
(`p.add_`, `p.copy_`, `state_dict()[k].copy_`) are therefore seen at the next call; writes that bypass
the counter (`p.data.copy_`, raw pointers, foreign kernels) need `engine.sync_packed(force=True)`,
re-assigned parameters `engine.refresh_parameters(module)`. `SFAST_PACKED_WEIGHTS=0`: no copies at all.
"""
import ctypes as C
import logging
import os
import atexit
import collections
import weakref
import threading
from collections import defaultdict

import torch

from ..hip import lib as L


def _cfg_get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


def _per_block(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def _as2d(w, rows, cols):
    """View a live 1x1-conv / linear weight as [rows, cols] WITHOUT copying (the plan keeps raw
    pointers into parameter storage, so a silent copy would detach it from in-place updates)."""
    v = w.reshape(rows, cols)
    if v.data_ptr() != w.data_ptr() or v.stride() != (cols, 1):
        raise UnsupportedUNet("weight is not viewable as a dense [out, in] matrix")
    return v


class UnsupportedUNet(NotImplementedError):
    pass


def live_norm_eps(m):
    """eps of every normalisation layer, read from the LIVE module (module path -> eps) instead of a config default or a constant:
    diffusers hands SVD's temporal resnets the spatial block's eps unless `temporal_eps` is set, custom UNets override `norm_eps`
    per block, ... The planner's own values are only the fallback for parameter sets that come without a module."""
    out = {}
    for name, mod in m.named_modules():
        e = getattr(mod, "eps", None)
        if name and isinstance(e, (int, float)) and not isinstance(e, bool):
            out[name] = float(e)
    return out


def lora_multiplier_sources(m):
    """base weight name -> callable giving the LIVE factor that multiplies up @ down besides the call's `scale`: diffusers
    LoRALinearLayer: network_alpha / rank (or 1); peft lora.Linear: scaling[adapter], 0 while the layer is merged or disabled."""
    out = {}
    for name, mod in m.named_modules():
        ll = getattr(mod, "lora_layer", None)
        if ll is not None and hasattr(ll, "down") and hasattr(ll, "up"):
            out[name + ".weight"] = (lambda ll=ll: (float(ll.network_alpha) / float(ll.rank)) if getattr(ll, "network_alpha", None) is not None else 1.0)
        elif hasattr(mod, "base_layer") and hasattr(mod, "lora_A") and hasattr(mod, "lora_B"):
            def peft_factor(mod=mod):
                if getattr(mod, "merged", False) or getattr(mod, "disable_adapters", False):
                    return 0.0
                act = [a for a in getattr(mod, "active_adapters", []) if a in mod.lora_A]
                return float(mod.scaling[act[0]]) if act else 0.0
            out[name + ".weight"] = peft_factor
        for t in ("to_q", "to_k", "to_v", "to_out"):  # diffusers <= 0.20 LoRAAttnProcessor: <attn>.processor.to_q_lora.{down, up}
            ll = getattr(getattr(mod, "processor", None), t + "_lora", None)
            if ll is not None and hasattr(ll, "down"):
                out[f"{name}.{'to_out.0' if t == 'to_out' else t}.weight"] = (
                    lambda ll=ll: (float(ll.network_alpha) / float(ll.rank)) if getattr(ll, "network_alpha", None) is not None else 1.0)
    return out


class _Pool:
    """Size-keyed free list of device buffers; a plan is executed in order on one stream, so a
    buffer released by the planner can be handed to any later op."""

    def __init__(self, device, dtype):
        self.device, self.dtype = device, dtype
        self.free = defaultdict(list)
        self.all = []
        self.writer = None  # the plan's buffer -> producer map (UNetPlan.writer), kept consistent with buffer recycling

    def get(self, numel):
        numel = int(numel)
        lst = self.free[numel]
        if lst:
            t = lst.pop()
            if self.writer is not None:
                self.writer.pop(id(t), None)  # recycled: whatever its last producer knew about it is void
            return t
        t = torch.empty(numel, dtype=self.dtype, device=self.device)
        self.all.append(t)
        return t

    def put(self, t):
        self.free[t.numel()].append(t)

    def total_bytes(self):
        return sum(t.numel() * t.element_size() for t in self.all)


logger = logging.getLogger(__name__)
LANE_MAIN, LANE_TEMB, LANE_KV = 0, 1, 2


class OpRecord:
    __slots__ = ("kind", "name", "flops", "bytes", "kernel", "launch", "tune", "lane", "needs", "packed")

    def __init__(self, kind, name, flops, nbytes, kernel, launch, tune=None, lane=LANE_MAIN, needs=None):
        self.kind, self.name, self.flops, self.bytes, self.kernel, self.launch = kind, name, flops, nbytes, kernel, launch
        self.tune = tune  # (params struct, launch_with(stream, ws_ptr, ws_bytes)) for tools/tune_igemm.py
        self.lane = lane  # ops that depend only on (timestep | text context) form side lanes of the graph
        self.needs = needs  # LANE_TEMB / LANE_KV: first consumer of a side lane's results joins it
        self.packed = None  # (ext structs, packed-weight records) when the op was handed packed copies of its weights (pipe 4)


class UNetPlan:
    """Static launch list + static I/O buffers for one (B, H, W, S_ctx) signature."""

    def __init__(self, engine, B, H, W, S_ctx):
        self.engine = engine
        self.B, self.H, self.W, self.S_ctx = B, H, W, S_ctx
        self.ops = []
        self.ws = [None, 0]  # shared workspace [tensor, nbytes]; main-lane ops run serially on one stream
        self.ws_side = [None, 0]  # workspace of the side lanes (they may overlap main-lane kernels)
        self.side_stream = None
        self._fork_events = None  # fork / lane / join events of run_forked: created once, never destroyed while the plan lives
        self.graph = None
        self.static_in = {}
        self.static_out = None
        self.pool = None
        self.keep = []  # ctypes objects / tensors that must outlive the launches
        self.kv_requests = []  # (name, Wk, Wv, kv buffer, C) of every cross-attention block, grouped at the end of build_plan
        self.P = None     # parameter name -> tensor the plan's launches read: the engine's parameters, LoRA'd weights replaced by merged copies
        self.lora = None  # un-fused LoRA: dict(table, n, total, eff, eff_ids, last) (UNet2DEngine._emit_lora)
        self.ip = None  # IP-Adapter: dict(tokens=[(buffer [B, S_ip, ctx], S_ip)], requests=[[kv request] per adapter], scales={block: [float]})
        # id(buffer) -> producer record of the op that last wrote the whole buffer (GroupNorm statistics hand-over). A record holds its
        # buffer (`buf`), so the id cannot be recycled by another tensor while the record exists, and writer_of() checks identity.
        self.writer = {}
        self.gn_candidates = []
        self.gn_fused = 0      # GroupNorms that run as ONE normalisation pass over statistics their producers emit
        self.gn_in_reduce = 0  # GroupNorms computed by their producer's split-K reduce launch (no launch of their own)
        self.packed_ops = 0    # GEMM / conv launches that read packed weight copies (pipe 4)

    def writer_of(self, t):
        rec = self.writer.get(id(t)) if t is not None else None
        return rec if rec is not None and rec["buf"] is t else None

    def run(self, stream_ptr):
        """Serial execution in program order on one stream (eager mode, tuning, per-op timing)."""
        for op in self.ops:
            op.launch(stream_ptr)

    def run_forked(self, main):
        """Execution for hipGraph capture: the time-embedding chain (depends only on t) and the cross-attention
        K/V projections (depend only on the text context) run on a side stream and become parallel branches of the
        captured graph; the main chain joins them at their first consumer. `main` is a torch.cuda.Stream."""
        side = self.side_stream
        if side is None or not any(op.lane != LANE_MAIN for op in self.ops):
            return self.run(main.cuda_stream)
        # The fork / join events live as long as the plan. They are recorded on streams that are being CAPTURED: an event object that is
        # destroyed before hipStreamEndCapture (a local `torch.cuda.Event()`, or the temporary inside `Stream.wait_stream`) leaves the
        # capturing streams' bookkeeping pointing at freed host memory.
        ev = self._fork_events
        if ev is None:
            ev = self._fork_events = {k: torch.cuda.Event() for k in ("fork", LANE_TEMB, LANE_KV, "join")}
        if FORK_EVENTS_LOCAL:  # A/B knob: the round-4 behaviour (events die inside the capture)
            ev = {k: torch.cuda.Event() for k in ev}
        ev["fork"].record(main)
        side.wait_event(ev["fork"])
        with torch.cuda.stream(side):
            sp = side.cuda_stream
            for lane in (LANE_TEMB, LANE_KV):
                for op in self.ops:
                    if op.lane == lane:
                        op.launch(sp)
                ev[lane].record(side)
        mp = main.cuda_stream
        joined = set()
        for op in self.ops:
            if op.lane != LANE_MAIN:
                continue
            if op.needs is not None and op.needs not in joined:
                for lane in (LANE_TEMB, LANE_KV):  # joining KV implies the earlier temb work too
                    if lane <= op.needs and lane not in joined:
                        main.wait_event(ev[lane])
                        joined.add(lane)
            op.launch(mp)
        ev["join"].record(side)
        main.wait_event(ev["join"])

    def summary(self):
        agg = defaultdict(lambda: [0, 0.0, 0.0])
        for op in self.ops:
            a = agg[op.kind]
            a[0] += 1
            a[1] += op.flops
            a[2] += op.bytes
        return {k: {"count": v[0], "gflop": v[1] / 1e9, "mbytes": v[2] / 1e6} for k, v in agg.items()}


# hipGraphExec objects that are no longer wanted. A graph is never destroyed in the same breath as its last replay: the HIP runtime releases
# the kernel commands of a launch on its asynchronous completion thread (ROCclr NDRangeKernelCommand::releaseResources -> free of the
# captured kernel arguments), and a graph torn down under it ends in an invalid free() on that thread a few milliseconds later -- the
# backtrace of the round-4 crash (profiles/r05_crash_backtrace_run2.log). Retired graphs are dropped oldest-first, only when more than
# 2 * _RETIRED_KEEP have piled up, only from capture_plan_graph's own (non-capturing, non-GC) context and only after a device synchronise.
_RETIRED = collections.deque()
_RETIRED_KEEP = 8


def retire_graph(g, device=None):
    _RETIRED.append((g, device))


class OwnedGraph:
    """A captured hipGraph whose teardown is deferred (round 6). Everything that keeps a live graph -- the plan /
    graph cache of a compiled wrapper, DenoiseLoop, the engines' own plans -- holds it through this handle. When the last reference
    goes (`del pipe`, garbage collection of a compiled model, interpreter exit) the finalizer does NOT destroy the hipGraphExec: it
    moves it to the retired queue, which is trimmed only from a non-capturing context after a device synchronise (`_trim_retired`)
    or, at interpreter exit, by `_drain_retired_at_exit`. Before this the WINNING graph of every plan was destroyed by plain
    reference counting -- the trigger of the round-4 crash for any user who dropped a pipeline right after its last replay -- and
    bench.py kept its graphs alive by hand (`_KEEP_GRAPHS`). The reference never evicts a captured graph at all
    (/root/reference/src/sfast/cuda/graphs.py:31-49, :139-164: a per-callable cache under one lock)."""
    __slots__ = ("_g", "_fin", "__weakref__")

    def __init__(self, g, device=None):
        self._g = g
        self._fin = weakref.finalize(self, retire_graph, g, device)

    def replay(self):
        self._g.replay()

    def pool(self):
        return self._g.pool()

    @property
    def raw(self):
        return self._g

    def __getattr__(self, name):  # anything else of torch.cuda.CUDAGraph (debug dumps, ...)
        return getattr(object.__getattribute__(self, "_g"), name)


def _drain_retired_at_exit():
    """atexit (registered at import, so it runs AFTER the weakref finalizers of still-live OwnedGraphs): the device is idle before the
    interpreter tears the retired graphs down."""
    try:
        if _RETIRED and torch.cuda.is_initialized():
            for dev in {d for _, d in _RETIRED}:
                torch.cuda.synchronize(dev)
    except Exception:
        pass
    _RETIRED.clear()


atexit.register(_drain_retired_at_exit)


def _trim_retired():
    if len(_RETIRED) > 2 * _RETIRED_KEEP:
        for dev in {d for _, d in _RETIRED}:
            torch.cuda.synchronize(dev)
        while len(_RETIRED) > _RETIRED_KEEP:
            _RETIRED.popleft()


def capture_plan_graph(plan, stream, pool=None, tail=None, calibrate=True):
    """Capture `plan` (+ an optional `tail(stream_ptr)` launch) into a hipGraph on `stream`.

    Two graph shapes are possible: a single chain, or the chain with the time-embedding / text-K/V side lanes
    forked into parallel branches (`UNetPlan.run_forked`). Which one replays faster depends on how much idle
    capacity the main chain leaves, so both are captured and the faster one is kept -- the same measure-then-commit
    policy as the kernel autotuner. The measure is the steady state of a denoise loop: 8 replays queued back to back
    between one event pair (median of 3). A single replay followed by a host synchronisation charges every graph its
    launch latency and ranked the forked shape first (5.64 vs 5.70 ms) although back to back it is the slower one
    (5.60 vs 5.42 ms: with the side lanes reduced to ~8 grouped launches the fork / join edges cost more than the
    overlap returns; tools/replay_gap_probe.py, profiles/r02_replay_gap_probe.log)."""
    dev = plan.engine.device
    _trim_retired()
    calibrate = calibrate and GRAPH_CALIBRATE

    def cap(forked):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            kw = {"pool": pool} if pool is not None else {}
            with torch.cuda.graph(g, stream=stream, **kw):
                cur = torch.cuda.current_stream(dev)
                if forked:
                    plan.run_forked(cur)
                else:
                    plan.run(cur.cuda_stream)
                if tail is not None:
                    tail(cur.cuda_stream)
        return g

    has_side = plan.side_stream is not None and any(op.lane != LANE_MAIN for op in plan.ops)
    graphs = [(False, cap(False))]
    if has_side:
        graphs.append((True, cap(True)))
    # The candidate that loses the calibration below is RETIRED, not destroyed on the spot (see retire_graph): hipGraphExecDestroy in the
    # same breath as the graph's last replay is one of the two triggers of the round-4 crash.
    if len(graphs) == 1 or not calibrate:
        for _, g in graphs[:-1]:  # (the serial candidate of an uncalibrated forked capture)
            retire_graph(g, dev)
        return OwnedGraph(graphs[-1][1], dev), graphs[-1][0]
    best = None
    with torch.cuda.stream(stream):
        for forked, g in graphs:
            g.replay()
            g.replay()
            ts = []
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                for _ in range(8):
                    g.replay()
                b.record(stream)
                b.synchronize()
                ts.append(a.elapsed_time(b) / 8)
            t = sorted(ts)[1]
            cal = getattr(plan, "graph_calibration_ms", None) or {}
            cal["forked" if forked else "serial"] = t
            plan.graph_calibration_ms = cal
            if best is None or t < best[0]:
                best = (t, forked, g)
    if not GRAPH_DESTROY_LOSER:
        for _, g in graphs:
            if g is not best[2]:
                retire_graph(g, dev)
    return OwnedGraph(best[2], dev), best[1]


# SFAST_GRAPH_CALIBRATE=0: keep the forked candidate without timing the two graph shapes against each other (52 extra replays of the step:
# what made the counter passes over the SVD-XT step -- 930 launches each -- run into their timeouts, tools/gpu_pmc_bench.sh)
GRAPH_CALIBRATE = os.environ.get("SFAST_GRAPH_CALIBRATE", "1") not in ("0", "false", "off", "")
# A/B knobs of the round-5 crash hunt (tools/crash_repro.py): 1 = the round-4 behaviour
FORK_EVENTS_LOCAL = os.environ.get("SFAST_FORK_EVENTS_LOCAL", "0") not in ("0", "false", "off", "")
GRAPH_DESTROY_LOSER = os.environ.get("SFAST_GRAPH_DESTROY_LOSER", "0") not in ("0", "false", "off", "")
# sfast_epilogue_ext.flags of every GEMM / conv launch of a plan. SFAST_SPLITK_JOIN=1: split-K problems finish inside the GEMM kernel
# (ticket counters at the end of the plan's workspace) instead of a reduce launch -- off by default, it measured slower on the SD1.5
# step (one workgroup per tile re-reads all slabs; DESIGN.md round 3, profiles/r03_splitk_join_*.json.log)
EXT_FLAGS = L.EXT_WS_TICKETS if os.environ.get("SFAST_SPLITK_JOIN", "0") not in ("0", "false", "off", "") else 0
# Two fusions that were built to parity in round 4, measured SLOWER in the SD1.5 step than what they replace, and therefore live in the
# PROBE build of the library only (build.py --probes, loaded with SFAST_HIP_PROBES=1): the product library answers
# SFAST_ERR_UNSUPPORTED for both, and without the probe library these knobs do nothing.
#  * SFAST_FUSE_GN_CONV=1: GroupNorm+SiLU -> 3x3 conv as one weight-streaming launch (B*H*W <= 128: SD1.5's 8x8 level; csrc/gnconv.hip,
#    sfast_hip_gn_conv2d): 180.1 vs 182.2 it/s (profiles/r04_gnconv_step_ab_run6.log).
#  * SFAST_GN_IN_REDUCE=1: the GroupNorm(+SiLU) behind a split-K conv / GEMM rides in that problem's reduce launch
#    (UNet2DEngine._fuse_gn_into_reduce, sfast_epilogue_ext.gn_out): 180.2 vs 181.2 it/s (profiles/r04_reduce_gn_ab_run{7,8,9}.log).
_PROBE_LIB = os.environ.get("SFAST_HIP_PROBES", "0") == "1"
FUSE_GN_CONV = _PROBE_LIB and os.environ.get("SFAST_FUSE_GN_CONV", "0") not in ("0", "false", "off", "")
GN_IN_REDUCE = _PROBE_LIB and os.environ.get("SFAST_GN_IN_REDUCE", "0") not in ("0", "false", "off", "")
# Packed weights (pipe 4, csrc/igemm_pk.h): every GEMM / conv whose weight is a parameter of the engine is offered a packed copy (1 KB
# MFMA fragments, sfast_hip_pack_weight); the autotuner then times the pipe-4 kernels beside the ring kernels and keeps a copy only
# where one of them won. The copies are re-packed when the parameter's version counter moved (UNet2DEngine.sync_packed, called by
# every compiled forward before it replays): in-place updates through the parameter or its state_dict() tensor -- the reference's
# "Dynamically Switch LoRA" recipe, README.md:228-265 -- are seen at the next call; writes that bypass autograd's version counter
# (`p.data.copy_`, raw pointers) need `engine.sync_packed(force=True)`. SFAST_PACKED_WEIGHTS=0: no copies, weights read live only.
PACKED_WEIGHTS = os.environ.get("SFAST_PACKED_WEIGHTS", "1") not in ("0", "false", "off", "")


class DeviceHost:
    """What an engine asks of its surroundings: the C-ABI library, a ROCm device, streams, whether to measure kernel choices.
    This is the only implementation the package has -- the real library on a real device, no CPU path. (The planner tests replace
    it from outside: tests/abi_emulator.EmuHost hands the engines a host emulator of the C ABI so that pointer / stride /
    buffer-reuse logic can be checked without a GPU.)"""

    def library(self):
        return L.load()

    def require_device(self, device, who):
        if device.type != "cuda":
            raise L.SfastHipError(f"{who} needs parameters on a ROCm device; there is no CPU path")

    def init_device(self, device):
        L.init_device(device)

    def stream_ptr(self, device):
        return torch.cuda.current_stream(device).cuda_stream

    def new_stream(self, device):
        return torch.cuda.Stream(device=device)

    def tuning(self):
        from . import autotune
        return autotune.enabled()


class UNet2DEngine:
    """Executor for SD1.5 / SD2.x / SDXL-family `UNet2DConditionModel` parameter sets."""

    lora = ()         # [(base weight name, down name, up name)] of the un-fused LoRA factors found among the parameters (_parse_lora)
    _lora_mult = {}   # base weight name -> callable: the live network_alpha / rank or peft scaling (from_module)
    ip_proj = None  # [(prefix, tokens per image, image embedding width)] when an IP-Adapter is loaded (_parse_ip_adapter)
    _ip_processors = {}
    _param_objs = {}  # parameter name -> nn.Parameter (set per instance by from_module): the version counters sync_packed() watches

    @property
    def _pk(self):
        """(data_ptr, N, K, ldw) -> packed-weight record (dict: w, buf, N, K, ldw, users, name, version); see _packed_for."""
        return self.__dict__.setdefault("_pk_records", {})

    def __init__(self, config, params, device=None, dtype=None, _host=None):
        self.host = _host if _host is not None else DeviceHost()
        self.lib = self.host.library()
        self.cfg = config
        self.params = params
        first = params["conv_in.weight"]
        self.device = device or first.device
        self.dtype = dtype or first.dtype
        if self.dtype not in (torch.float16, torch.bfloat16):
            raise UnsupportedUNet(f"UNet2DEngine runs f16/bf16 parameters, got {self.dtype}")
        self.host.require_device(self.device, type(self).__name__)
        self.dt = L.F16 if self.dtype == torch.float16 else L.BF16
        self.esize = 2
        self.norm_eps = {}  # module path -> eps of the live normalisation layer (from_module); empty: the planner's defaults
        self._parse_config()
        self._plans = {}
        self._lock = threading.Lock()

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_module(cls, m, _host=None):
        """Build from a diffusers-style module: `m.config` + `m.named_parameters()` (live storage)."""
        cfg = getattr(m, "config", None)
        if cfg is None:
            raise UnsupportedUNet("module has no .config")
        params = {}
        with torch.no_grad():
            for name, p in m.named_parameters():
                d = p.data
                if d.ndim == 4 and not d.is_contiguous(memory_format=torch.channels_last):
                    # K-contiguous [Cout][kh][kw][Cin] image for the implicit-GEMM kernels; same effect as the reference's
                    # apply_memory_format (utils/memory_format.py:49-57). 1x1 kernels and single-input-channel weights are
                    # K-contiguous in either format: their storage is left alone. Otherwise the parameter's storage pointer
                    # CHANGES here (like `module.to(memory_format=channels_last)`): anything holding the old storage -- an
                    # optimizer state, an external LoRA merger -- must re-read `p.data` after compile().
                    if d.shape[2] * d.shape[3] == 1 and d.is_contiguous():
                        pass
                    else:
                        p.data = d = d.contiguous(memory_format=torch.channels_last)
                # peft lora.Linear keeps the wrapped layer's tensors under `<linear>.base_layer.*`: the plan knows them by the plain name
                params[name.replace(".base_layer.", ".")] = d
        eng = cls(cfg, params, _host=_host)
        eng.norm_eps = live_norm_eps(m)
        eng._param_objs = {n.replace(".base_layer.", "."): p for n, p in m.named_parameters()}
        if eng.lora:
            eng._lora_mult = lora_multiplier_sources(m)
        if getattr(eng, "ip_proj", None):
            # IP-Adapter: `scale` is a python attribute of each attn2 processor (pipe.set_ip_adapter_scale); read live by ip_scales()
            eng._ip_processors = {name: mod.processor for name, mod in m.named_modules()
                                  if name.endswith(".attn2") and hasattr(getattr(mod, "processor", None), "to_k_ip")}
        return eng

    def refresh_parameters(self, m):
        """Re-bind after parameters were re-assigned (not needed for in-place `copy_` updates)."""
        new = type(self).from_module(m, _host=self.host)
        with self._lock:
            self.params = new.params
            self.norm_eps = new.norm_eps
            # everything that is keyed by the OLD storage goes with it: version sources, LoRA factor sources, IP-Adapter processors, and the
            # packed-weight caches (data_ptr -> name map, records). A record that survived would pin the freed tensor and its packed copy,
            # keep being re-packed from it, and a plan-owned buffer that later lands on the freed address would be taken for a parameter.
            self._param_objs = new._param_objs
            self.lora, self._lora_mult = new.lora, new._lora_mult
            self._ip_processors = new._ip_processors
            self._plans.clear()
            self._pk.clear()
            self.__dict__.pop("_ptr_names", None)
            # plans held OUTSIDE the engine (a compiled forward's per-signature cache, a DenoiseLoop's private plan) compare this counter at
            # their next call and rebuild: they would otherwise replay graphs over the old storage and the old packed copies
            self.generation = getattr(self, "generation", 0) + 1

    def _parse_config(self):
        g = lambda k, d=None: _cfg_get(self.cfg, k, d)
        self.boc = tuple(g("block_out_channels"))
        n = len(self.boc)
        self.layers = g("layers_per_block", 2)
        if isinstance(self.layers, (tuple, list)):
            if len(set(self.layers)) != 1:
                raise UnsupportedUNet("per-block layers_per_block")
            self.layers = self.layers[0]
        self.down_types = tuple(g("down_block_types"))
        self.up_types = tuple(g("up_block_types") or ())
        self.is_controlnet = "controlnet_mid_block.weight" in self.params  # diffusers ControlNetModel: no up path
        for t in self.down_types:
            if t not in ("CrossAttnDownBlock2D", "DownBlock2D"):
                raise UnsupportedUNet(f"down block {t}")
        for t in self.up_types:
            if t not in ("CrossAttnUpBlock2D", "UpBlock2D"):
                raise UnsupportedUNet(f"up block {t}")
        if g("mid_block_type", "UNetMidBlock2DCrossAttn") not in (None, "UNetMidBlock2DCrossAttn"):
            raise UnsupportedUNet(f"mid block {g('mid_block_type')}")
        heads = g("num_attention_heads") or g("attention_head_dim")
        self.heads = _per_block(heads, n)
        self.depth = _per_block(g("transformer_layers_per_block", 1), n)
        self.groups = g("norm_num_groups", 32)
        self.eps = float(g("norm_eps", 1e-5))
        self.linear_proj = bool(g("use_linear_projection", False))
        self.flip = bool(g("flip_sin_to_cos", True))
        self.freq_shift = float(g("freq_shift", 0))
        self.ctx_dim = g("cross_attention_dim")
        if isinstance(self.ctx_dim, (tuple, list)):
            if len(set(self.ctx_dim)) != 1:
                raise UnsupportedUNet("per-block cross_attention_dim")
            self.ctx_dim = self.ctx_dim[0]
        self.in_ch, self.out_ch = g("in_channels", 4), g("out_channels", 4)
        self.add_type = g("addition_embed_type")
        if self.add_type not in (None, "text_time"):
            raise UnsupportedUNet(f"addition_embed_type {self.add_type}")
        self.add_time_dim = g("addition_time_embed_dim")
        if g("time_embedding_type") not in (None, "positional"):
            raise UnsupportedUNet(f"time_embedding_type={g('time_embedding_type')}")
        # encoder_hid_dim_type "ip_image_proj" = an IP-Adapter is loaded (diffusers loaders/unet.py _load_ip_adapter_weights): the image
        # embeddings of added_cond_kwargs pass through `encoder_hid_proj` and feed a second, decoupled cross-attention per block
        hid = g("encoder_hid_dim_type")
        if hid not in (None, "ip_image_proj"):
            raise UnsupportedUNet(f"encoder_hid_dim_type={hid}")
        self.ip_proj = self._parse_ip_adapter() if hid == "ip_image_proj" else None
        self._ip_processors = {}
        # class conditioning: "timestep" (sinusoid -> MLP) and "projection" (float vector -> MLP) are plan inputs; an nn.Embedding
        # table (class_embed_type None + num_class_embeds), "identity" and "simple_projection" are not built
        self.class_type = g("class_embed_type")
        if self.class_type not in (None, "timestep", "projection"):
            raise UnsupportedUNet(f"class_embed_type={self.class_type}")
        # LCM guidance embedding: `timestep_cond` [B, time_cond_proj_dim] -> bias-free Linear added to the sinusoid before the MLP
        self.tcond_dim = g("time_cond_proj_dim")
        if g("act_fn", "silu") not in ("silu", "swish"):
            raise UnsupportedUNet("act_fn")
        if g("resnet_time_scale_shift", "default") != "default":
            raise UnsupportedUNet("resnet_time_scale_shift")
        if g("dual_cross_attention", False) or g("only_cross_attention", False) or g("upcast_attention", False):
            raise UnsupportedUNet("dual/only_cross/upcast attention")
        self._check_config_whitelist()
        self.temb_dim = self.params["time_embedding.linear_1.weight"].shape[0]
        self._validate_params()

    # Every config option the planner does NOT implement must hold the value the plan silently assumes; anything else would
    # engage the native engine and compute a different network than the module defines (diffusers option names).
    _ASSUMED = {
        "center_input_sample": (False,), "downsample_padding": (1,), "mid_block_scale_factor": (1, 1.0), "dropout": (0, 0.0),
        "reverse_transformer_layers_per_block": (None,), "encoder_hid_dim": (None,), "num_class_embeds": (None,),
        "resnet_skip_time_act": (False,), "resnet_out_scale_factor": (1, 1.0), "time_embedding_dim": (None,),
        "time_embedding_act_fn": (None,), "timestep_post_act": (None,), "conv_in_kernel": (3,),
        "conv_out_kernel": (3,), "attention_type": ("default", None), "class_embeddings_concat": (False,),
        "mid_block_only_cross_attention": (None, False), "cross_attention_norm": (None,), "attention_bias": (False, None),
        "global_pool_conditions": (False,), "controlnet_conditioning_channel_order": ("rgb", None),
        "norm_elementwise_affine": (True, None), "use_timestep_embedding": (True, None), "only_cross_attention": (False, None),
    }
    _HANDLED = {
        "sample_size", "in_channels", "out_channels", "flip_sin_to_cos", "freq_shift", "down_block_types", "mid_block_type",
        "up_block_types", "block_out_channels", "layers_per_block", "act_fn", "norm_num_groups", "norm_eps", "cross_attention_dim",
        "transformer_layers_per_block", "encoder_hid_dim_type", "attention_head_dim", "num_attention_heads", "dual_cross_attention",
        "use_linear_projection", "class_embed_type", "addition_embed_type", "addition_time_embed_dim", "upcast_attention",
        "resnet_time_scale_shift", "time_embedding_type", "projection_class_embeddings_input_dim", "addition_embed_type_num_heads",
        "conditioning_embedding_out_channels", "conditioning_channels", "time_cond_proj_dim",
    }

    def _config_items(self):
        cfg = self.cfg
        if isinstance(cfg, dict):
            return list(cfg.items())
        if hasattr(cfg, "items"):
            try:
                return list(cfg.items())
            except Exception:
                pass
        return [(k, getattr(cfg, k)) for k in dir(cfg) if not k.startswith("_") and not callable(getattr(cfg, k, None))]

    def _check_config_whitelist(self):
        for k, v in self._config_items():
            if k.startswith("_") or k in self._HANDLED:
                continue
            if isinstance(v, list):
                v = tuple(v)
            ok = self._ASSUMED.get(k)
            if ok is not None:
                if v not in ok:
                    raise UnsupportedUNet(f"config option {k}={v!r} is not implemented by the native plan (it assumes {ok[0]!r})")
            elif v not in (None, False):
                raise UnsupportedUNet(f"unknown config option {k}={v!r}: refusing to guess its meaning")

    _LORA_PATTERNS = (  # (regex on the parameter name, base weight name template): the three layouts diffusers has used
        (r"^(.*)\.lora_layer\.down\.weight$", "{0}.weight", "{0}.lora_layer.up.weight"),                       # LoRACompatibleLinear (0.21 - 0.24)
        (r"^(.*)\.lora_A\.([^.]+)\.weight$", "{0}.weight", "{0}.lora_B.{1}.weight"),                             # peft lora.Linear (>= 0.25)
        (r"^(.*)\.processor\.(to_q|to_k|to_v)_lora\.down\.weight$", "{0}.{1}.weight", "{0}.processor.{1}_lora.up.weight"),  # LoRAAttnProcessor
        (r"^(.*)\.processor\.to_out_lora\.down\.weight$", "{0}.to_out.0.weight", "{0}.processor.to_out_lora.up.weight"),
    )

    def _parse_lora(self, want):
        """Un-fused LoRA factors among the parameters -> [(base, down, up)]. Native for the linears of the transformer blocks (attention
        projections, feed-forward, linear proj_in / proj_out): their effective weight W + s * up @ down is rebuilt by ONE launch per
        step (sfast_hip_lora_merge) from the live tensors, so the reference's in-place adapter switch (README.md:228-265) needs no
        re-capture. Anything else (conv LoRA, two adapters on one layer, DoRA, rank > 128) keeps the module's own forward."""
        import re
        found, have = {}, self.params
        for k in have:
            if "lora" not in k:
                continue
            for pat, base_t, up_t in self._LORA_PATTERNS:
                mt = re.match(pat, k)
                if mt:
                    base, up = base_t.format(*mt.groups()), up_t.format(*mt.groups())
                    if base in found:
                        raise UnsupportedUNet(f"more than one LoRA adapter on {base}")
                    found[base] = (k, up)
                    break
        out = []
        for base, (dn, un) in sorted(found.items()):
            w = have.get(base)
            ok = (base in want and un in have and w is not None and w.ndim == 2 and have[dn].ndim == 2 and have[un].ndim == 2
                  and (".attn1." in base or ".attn2." in base or ".ff.net." in base or base.endswith((".proj_in.weight", ".proj_out.weight")))
                  and have[dn].shape[1] == w.shape[1] and have[un].shape[0] == w.shape[0] and have[un].shape[1] == have[dn].shape[0]
                  and have[dn].shape[0] <= L.LORA_MAX_RANK and w.shape[1] % 8 == 0 and hasattr(self.lib, "sfast_hip_lora_merge"))
            if not ok:
                raise UnsupportedUNet(f"LoRA factors on {base} are outside the native plan (linear layers of the transformer blocks, rank <= {L.LORA_MAX_RANK})")
            out.append((base, dn, un))
        return tuple(out)

    def lora_multipliers(self, scale=1.0):
        """scales[i] of sfast_hip_lora_merge for this call: cross_attention_kwargs["scale"] times the layer's own live factor."""
        return [float(scale) * float(self._lora_mult.get(base, lambda: 1.0)()) for base, _, _ in self.lora]

    def _emit_lora(self, plan):
        lib, n = self.lib, len(self.lora)
        ents = (L.LoraEntry * n)()
        eff = {}
        for i, (base, dn, un) in enumerate(self.lora):
            w, d, u = self.params[base], self.params[dn], self.params[un]
            if d.stride(1) != 1 or u.stride(1) != 1 or w.stride(1) != 1:
                raise UnsupportedUNet(f"LoRA factors of {base} are not row-major")
            out = torch.empty((w.shape[0], w.shape[1]), dtype=self.dtype, device=self.device)
            e = ents[i]
            e.w, e.down, e.up, e.out = w.data_ptr(), d.data_ptr(), u.data_ptr(), out.data_ptr()
            e.N, e.K, e.r = w.shape[0], w.shape[1], d.shape[0]
            e.ldw, e.ldd, e.ldu, e.scale_index = w.stride(0), d.stride(0), u.stride(0), i
            eff[base] = out
        total = C.c_int32()
        rc = lib.sfast_hip_lora_merge_plan(ents, n, C.byref(total))
        if rc != 0:
            raise UnsupportedUNet(f"LoRA merge table refused ({rc}): {L.last_error() if hasattr(lib, 'sfast_hip_last_error') else ''}")
        table = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(self.device)
        scales = torch.ones(n, dtype=torch.float32, device=self.device)
        plan.static_in["lora_scale"] = scales
        plan.lora = dict(table=table, n=n, total=int(total.value), eff=eff, eff_ids={id(t) for t in eff.values()}, last=None)
        plan.P = dict(self.params)
        plan.P.update(eff)

    def _op_lora_merge(self, plan):
        lib, lo = self.lib, plan.lora
        tp, sp, n, total = lo["table"].data_ptr(), plan.static_in["lora_scale"].data_ptr(), lo["n"], lo["total"]
        nbytes = sum(2.0 * t.numel() * self.esize for t in lo["eff"].values())
        self._add(plan, "misc", f"lora.merge[{n} linears]", 0.0, nbytes,
                  lambda s: L.check(lib.sfast_hip_lora_merge(tp, n, total, sp, self.dt, s), "sfast_hip_lora_merge"), lane=LANE_TEMB)

    def _parse_ip_adapter(self):
        """[(parameter prefix, image tokens per image T, image embedding width)] of the ImageProjection layers under `encoder_hid_proj`
        (one per loaded adapter; `MultiIPAdapterImageProjection.image_projection_layers.{i}` or, in older diffusers, the bare layer)."""
        P, ctx = self.params, _cfg_get(self.cfg, "cross_attention_dim")
        pres = []
        if "encoder_hid_proj.image_embeds.weight" in P:
            pres = ["encoder_hid_proj"]
        else:
            while f"encoder_hid_proj.image_projection_layers.{len(pres)}.image_embeds.weight" in P:
                pres.append(f"encoder_hid_proj.image_projection_layers.{len(pres)}")
        if not isinstance(ctx, int):
            raise UnsupportedUNet("IP-Adapter with a per-block cross_attention_dim")
        if not pres:
            # another projection (IP-Adapter Plus' resampler, a user module): the plan does not run it -- it takes the PROJECTED image
            # tokens (`ip_hidden_states`, one [B, tokens, cross_attention_dim] tensor per adapter) as its input; the compiled forward
            # computes them with the module's own encoder_hid_proj, once per distinct image_embeds (they do not change along a denoise
            # loop). One entry (None, None, None) per adapter = per `to_k_ip.{i}` of the attention processors.
            n = 0
            while any(k.endswith(f".attn2.processor.to_k_ip.{n}.weight") for k in P):
                n += 1
            if n == 0:
                raise UnsupportedUNet("encoder_hid_dim_type 'ip_image_proj' without to_k_ip / to_v_ip parameters")
            return [(None, None, None)] * n
        out = []
        for pre in pres:
            w = P[pre + ".image_embeds.weight"]
            if w.ndim != 2 or w.shape[0] % ctx or w.shape[1] % 8 or (pre + ".norm.weight") not in P:
                raise UnsupportedUNet(f"{pre} is not an ImageProjection the plan knows")
            out.append((pre, w.shape[0] // ctx, w.shape[1]))
        return out

    @property
    def ip_external(self):
        """True when the image projection runs outside the plan (anything but a plain ImageProjection)."""
        return bool(self.ip_proj) and self.ip_proj[0][0] is None

    def ip_scales(self):
        """Snapshot of the live IP-Adapter scales: ((block path, (scale per adapter, ...)), ...). Part of the plan signature -- a scale
        is a launch constant of the plan (and of its captured graph); set_ip_adapter_scale() selects / builds another plan."""
        if not self.ip_proj:
            return None
        n = len(self.ip_proj)
        out = []
        for name in sorted(self._ip_processors):
            s = getattr(self._ip_processors[name], "scale", 1.0)
            s = list(s) if isinstance(s, (list, tuple)) else [s] * n
            if len(s) != n or not all(isinstance(v, (int, float)) for v in s):
                raise UnsupportedUNet("IP-Adapter scale that is not one number per adapter (masked / per-image scales)")
            out.append((name, tuple(float(v) for v in s)))
        return tuple(out)

    def ip_signature(self, added_cond_kwargs):
        """The `ip` argument of get_plan / build_plan for one call: (images per adapter, live scales), or None without an adapter."""
        if not self.ip_proj:
            return None
        return (tuple(int(t.shape[1]) for t in self._ip_embeds(added_cond_kwargs, None)), self.ip_scales())

    def _ip_embeds(self, added_cond_kwargs, B):
        """added_cond_kwargs["image_embeds"] as diffusers' MultiIPAdapterImageProjection takes it: one [B, images, D] tensor per adapter
        (a bare tensor = one adapter; [B, D] = one image)."""
        if not added_cond_kwargs or added_cond_kwargs.get("image_embeds") is None:
            raise ValueError("encoder_hid_dim_type 'ip_image_proj' requires `image_embeds` in added_cond_kwargs")  # diffusers' own error
        if self.ip_external:
            ie = added_cond_kwargs.get("ip_hidden_states")
            if ie is None:
                raise UnsupportedUNet("this IP-Adapter's image projection is not part of the plan: pass the projected tokens as "
                                      "added_cond_kwargs['ip_hidden_states'] (the compiled forward does, with module.encoder_hid_proj)")
            ie = list(ie) if isinstance(ie, (list, tuple)) else [ie]
            if len(ie) != len(self.ip_proj) or any(t.ndim != 3 or t.shape[2] != self.ctx_dim or (B is not None and t.shape[0] != B) for t in ie):
                raise UnsupportedUNet(f"ip_hidden_states: want {len(self.ip_proj)} tensors [B, tokens, {self.ctx_dim}]")
            return ie
        ie = added_cond_kwargs["image_embeds"]
        ie = list(ie) if isinstance(ie, (list, tuple)) else [ie]
        if len(ie) != len(self.ip_proj):
            raise ValueError(f"image_embeds holds {len(ie)} tensors, the UNet has {len(self.ip_proj)} IP-Adapters")
        out = []
        for t, (_, _, dimg) in zip(ie, self.ip_proj):
            t = t[:, None] if t.ndim == 2 else t
            if t.ndim != 3 or t.shape[2] != dimg or (B is not None and t.shape[0] != B):
                raise UnsupportedUNet(f"image_embeds of shape {tuple(t.shape)} (want [B, images, {dimg}])")
            out.append(t)
        return out

    def _validate_params(self):
        """Parameter inventory check: every tensor the plan will read exists with the expected shape, and the module holds no
        parameter the plan would ignore (attention biases, gated-attention fusers, LoRA wrappers, class embeddings, ...)."""
        from .unet_spec import unet2d_param_shapes
        n = len(self.boc)
        cfg = dict(block_out_channels=self.boc, layers_per_block=self.layers, down_block_types=self.down_types,
                   up_block_types=self.up_types if not self.is_controlnet else (), cross_attention_dim=self.ctx_dim,
                   transformer_layers_per_block=self.depth, use_linear_projection=self.linear_proj, in_channels=self.in_ch,
                   out_channels=self.out_ch, addition_embed_type=self.add_type,
                   projection_class_embeddings_input_dim=_cfg_get(self.cfg, "projection_class_embeddings_input_dim"),
                   time_cond_proj_dim=self.tcond_dim, class_embed_type=self.class_type)
        try:
            want = unet2d_param_shapes(cfg)
        except (KeyError, TypeError, IndexError) as e:
            raise UnsupportedUNet(f"config does not describe a UNet2DConditionModel the plan knows ({e})")
        if self.is_controlnet:
            want = {k: v for k, v in want.items() if not k.startswith(("up_blocks.", "conv_norm_out.", "conv_out."))}
        T = self.temb_dim
        have = self.params
        missing = [k for k in want if k not in have]
        if missing:
            raise UnsupportedUNet(f"parameters missing for the native plan: {missing[:3]}{' ...' if len(missing) > 3 else ''} "
                                  "(wrapped / renamed modules such as peft LoRA layers keep the eager forward)")
        if self.ip_proj:
            # IP-Adapter parameters: ImageProjection (Linear + LayerNorm) per adapter, to_k_ip / to_v_ip per cross-attention and adapter
            ctx = self.ctx_dim
            for pre, T_, dimg in self.ip_proj:
                if pre is None:  # external projection: its parameters belong to the module's encoder_hid_proj, which the plan never reads
                    want.update({k: tuple(v.shape) for k, v in have.items() if k.startswith("encoder_hid_proj.")})
                    continue
                want.update({pre + ".image_embeds.weight": (T_ * ctx, dimg), pre + ".image_embeds.bias": (T_ * ctx,),
                             pre + ".norm.weight": (ctx,), pre + ".norm.bias": (ctx,)})
            for k in [k for k in want if k.endswith(".attn2.to_k.weight")]:
                for i in range(len(self.ip_proj)):
                    for kv in ("to_k_ip", "to_v_ip"):
                        want[k[:-len("to_k.weight")] + f"processor.{kv}.{i}.weight"] = (want[k][0], ctx)
            missing = [k for k in want if k not in have]
            if missing:
                raise UnsupportedUNet(f"IP-Adapter parameters missing for the native plan: {missing[:3]}")
        self.lora = self._parse_lora(want)
        for _, dn, un in self.lora:
            want[dn], want[un] = tuple(have[dn].shape), tuple(have[un].shape)
        extra = [k for k in have if k not in want and not k.startswith("controlnet_")]
        if extra:
            raise UnsupportedUNet(f"module has parameters the native plan would ignore: {extra[:3]}{' ...' if len(extra) > 3 else ''}")
        for k, shp in want.items():
            got = tuple(have[k].shape)
            if "time_emb" in k or "add_embedding" in k or "time_embedding" in k or "class_embedding" in k:
                continue  # widths derive from the embedding dim, checked by the GEMM launches themselves
            if got != tuple(shp):
                raise UnsupportedUNet(f"parameter {k} has shape {got}, the plan expects {tuple(shp)}")
        del n, T

    # ------------------------------------------------------------------------------------------
    # plan construction helpers
    # ------------------------------------------------------------------------------------------
    # ------------------------------------------------------------------------------------------
    # packed weights (pipe 4)
    # ------------------------------------------------------------------------------------------
    def _packed_for(self, w):
        """Packed copy of a weight that is one of this engine's parameters (else None): a [N, K] view or a [Cout][KH][KW][Cin] conv
        weight. One record per distinct (storage offset, geometry); packed now, on the current stream."""
        if not PACKED_WEIGHTS or not hasattr(self.lib, "sfast_hip_pack_weight"):
            return None
        if w.ndim == 4:
            Cout, Cin, KH, KW = w.shape
            if not (w.stride(1) == 1 and (KW == 1 or w.stride(3) == Cin) and (KH == 1 or w.stride(2) == KW * Cin)):
                return None
            N, K, ldw = Cout, KH * KW * Cin, (w.stride(0) if Cout > 1 else KH * KW * Cin)
        elif w.ndim == 2 and w.stride(1) == 1:
            N, K = w.shape
            ldw = w.stride(0) if N > 1 else K
        else:
            return None
        if K % 8 or ldw % 8 or w.data_ptr() % 16:
            return None
        key = (w.data_ptr(), N, K, ldw)
        rec = self._pk.get(key)
        if rec is None:
            names = getattr(self, "_ptr_names", None)
            if names is None:
                names = self._ptr_names = {t.data_ptr(): n for n, t in self.params.items() if torch.is_tensor(t)}
            name = names.get(w.data_ptr())
            if name is None:
                return None  # a plan-owned buffer (padded conv_in image, merged LoRA weight, ...): rewritten per step, never packed here
            buf = torch.empty(self.lib.sfast_hip_packed_weight_bytes(N, K), dtype=torch.uint8, device=self.device)
            rec = self._pk[key] = dict(w=w, buf=buf, N=N, K=K, ldw=ldw, users=0, name=name, version=None)
            self._pack(rec)
        return rec

    def _pack(self, rec):
        L.check(self.lib.sfast_hip_pack_weight(rec["w"].data_ptr(), rec["buf"].data_ptr(), rec["N"], rec["K"], rec["ldw"], self.dt,
                                               self.host.stream_ptr(self.device)), "sfast_hip_pack_weight")
        p = self._version_source(rec["name"])
        rec["version"] = p._version if p is not None else None

    def _version_source(self, name):
        """The object whose autograd version counter tells that parameter `name` was written: the module's nn.Parameter (from_module;
        the engine's own `p.data` alias has a counter of its own), else the tensor the engine was constructed with."""
        p = self._param_objs.get(name)
        return p if p is not None else self.params.get(name)

    def sync_packed(self, force=False):
        """Re-pack every packed weight whose parameter changed since it was packed (version counter), or all of them (`force`).
        Launches on the current stream, outside any graph; a no-op costing a few microseconds when nothing changed."""
        n = 0
        for rec in list(self._pk.values()):  # a snapshot: another thread may be building a plan (new records) while this one replays
            if rec["users"] <= 0:
                continue
            p = self._version_source(rec["name"])
            if p is not None and p.data_ptr() != rec["w"].data_ptr() and not rec.get("warned"):
                # `p.data = other` / load_state_dict(assign=True): the plan still reads the OLD storage (packed or not)
                rec["warned"] = True
                logger.warning("sfast: parameter %s was re-assigned after the engine was built; call refresh_parameters() "
                               "(the plan reads the storage it was built on)", rec["name"])
            if force or (p is not None and p._version != rec["version"]):
                self._pack(rec)
                n += 1
        return n

    def _offer_packed(self, plan, weights, exts):
        """Give the op's launches packed copies of `weights` (all or nothing). Returns the records, or None."""
        recs = [self._packed_for(w) for w in weights]
        if any(r is None for r in recs):
            return None
        arr = (C.c_void_p * len(recs))(*[r["buf"].data_ptr() for r in recs])
        plan.keep.append(arr)
        for x in exts:
            x.w_packed = C.cast(arr, C.c_void_p)
        for r in recs:
            r["users"] += 1
        return recs

    def _settle_packed(self, plan):
        """After tuning: an op that did not choose a pipe-4 kernel gives its packed copies back; copies nobody uses are freed."""
        for op in plan.ops:
            if op.packed is None:
                continue
            exts, recs = op.packed
            if int(op.tune[0].variant) < 40 or int(op.tune[0].variant) >= 100:
                for x in exts:
                    x.w_packed = None
                for r in recs:
                    r["users"] -= 1
                op.packed = None
        for key in [k for k, r in self._pk.items() if r["users"] <= 0]:
            del self._pk[key]
        plan.packed_ops = sum(1 for op in plan.ops if op.packed is not None)

    def _add(self, plan, kind, name, flops, nbytes, launch, tune=None, lane=LANE_MAIN, needs=None):
        plan.ops.append(OpRecord(kind, name, flops, nbytes, None, launch, tune, lane, needs))

    def _need_ws(self, plan, nbytes, lane=LANE_MAIN):
        holder = plan.ws if lane == LANE_MAIN else plan.ws_side
        if nbytes > holder[1]:
            holder[1] = int(nbytes)

    def _op_gn(self, plan, name, x, x2, C1, Ctot, N, HW, y, eps, silu, prefix):
        lib = self.lib
        gamma, beta = self.params[prefix + ".weight"], self.params[prefix + ".bias"]
        eps = self.norm_eps.get(prefix, eps)
        p = L.GnParams(self.dt, L.NHWC, N, Ctot, HW, self.groups, C1, L.ACT_SILU if silu else L.ACT_NONE, float(eps))
        self._need_ws(plan, lib.sfast_hip_group_norm_workspace_bytes(C.byref(p)))
        xp, x2p, gp, bp, yp = x.data_ptr(), (x2.data_ptr() if x2 is not None else None), gamma.data_ptr(), beta.data_ptr(), y.data_ptr()
        ws = plan.ws
        plan.keep.append(p)
        pre = [None]  # set by _fuse_gn_statistics: (stats1 ptr, layout1, stats2 ptr, layout2)

        def launch(stream, p=p, pre=pre):
            if pre[0] is not None:
                s1, l1, s2, l2 = pre[0]
                L.check(lib.sfast_hip_group_norm_apply(xp, x2p, gp, bp, yp, C.byref(p), s1, C.byref(l1), s2, C.byref(l2) if l2 is not None else None,
                                                       stream), name)
                return
            L.check(lib.sfast_hip_group_norm(xp, x2p, gp, bp, yp, C.byref(p), ws[0].data_ptr() if ws[0] is not None else None, ws[1], stream), name)

        self._add(plan, "gn_silu" if silu else "gn", name, 0.0, (2.0 * N * HW * Ctot + 2 * Ctot) * self.esize, launch)
        plan.gn_candidates.append(dict(name=name, p=p, pre=pre, xw=plan.writer_of(x), x2w=plan.writer_of(x2),
                                       concat=x2 is not None, HW=HW, gp=gp, bp=bp, yp=yp, eps=float(eps), silu=bool(silu)))
        plan.writer.pop(id(y), None)

    def _op_ln(self, plan, name, x, y, M, N, prefix, lane=LANE_MAIN):
        lib = self.lib
        gamma, beta = self.params[prefix + ".weight"], self.params[prefix + ".bias"]
        p = L.LnParams(self.dt, M, N, float(self.norm_eps.get(prefix, 1e-5)))
        xp, gp, bp, yp = x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr()
        plan.keep.append(p)
        plan.writer.pop(id(y), None)

        def launch(stream, p=p):
            L.check(lib.sfast_hip_layer_norm(xp, gp, bp, yp, C.byref(p), stream), name)

        self._add(plan, "ln", name, 0.0, (2.0 * M * N + 2 * N) * self.esize, launch, lane=lane)

    def _op_mix(self, plan, name, x, y, vec, out, M, Cc, *, mix=None, switch=False, vec_rows=1, vec_mod=1, ld_vec=0, wx=1.0, wy=0.0,
                lane=LANE_MAIN, needs=None):
        """out[r] = wx * x[r] + wy * y[r] + vec[(r / vec_rows) % vec_mod]  (sfast_hip_mix_rows; `mix`: a live AlphaBlender factor)."""
        lib = self.lib
        p = L.MixParams(self.dt, int(M), int(Cc), int(vec_rows), int(vec_mod), int(ld_vec), float(wx), float(wy), int(switch))
        plan.keep.append(p)
        xp, yp, vp = x.data_ptr(), (y.data_ptr() if y is not None else None), (vec.data_ptr() if vec is not None else None)
        mp, op = (mix.data_ptr() if mix is not None else None), out.data_ptr()
        plan.writer.pop(id(out), None)
        self._add(plan, "misc", name, 0.0, (2 + (y is not None)) * float(M) * Cc * self.esize,
                  lambda s, p=p: L.check(lib.sfast_hip_mix_rows(xp, yp, vp, mp, op, C.byref(p), s), name), lane=lane, needs=needs)

    def _op_gemm(self, plan, name, x, weights, bias, out, M, N, K, ldx, ldo, *, residual=None, ldr=0, act=L.ACT_NONE,
                 geglu=False, res_before_act=False, out_offset=0, kind=None, lane=LANE_MAIN):
        lib = self.lib
        p = L.GemmParams()
        p.dtype, p.M, p.N, p.K = self.dt, M, N, K
        w0 = weights[0]
        ldw = w0.stride(0) if w0.ndim == 2 else K
        p.ldx, p.ldw, p.ldo, p.ldr = ldx, ldw, ldo, ldr
        p.n_wseg, p.rows_per_seg = len(weights), w0.shape[0]
        p.geglu, p.act, p.res_before_act, p.alpha = int(geglu), act, int(res_before_act), 1.0
        p.rows_per_batch, p.ld_rowbias, p.in_act, p.variant, p.split_k = 0, 0, 0, 0, 0
        self._need_ws(plan, lib.sfast_hip_gemm_workspace_bytes(C.byref(p)), lane)
        segs = (C.c_void_p * len(weights))(*[w.data_ptr() for w in weights])
        xp = x.data_ptr()
        bp = bias.data_ptr() if bias is not None else None
        rp = residual.data_ptr() if residual is not None else None
        op = out.data_ptr() + out_offset * self.esize
        ws = plan.ws if lane == LANE_MAIN else plan.ws_side
        ext, text = L.EpilogueExt(0.0, 0, 0, EXT_FLAGS), L.EpilogueExt(0.0, 0, 0, EXT_FLAGS)  # (text: the tuner's launches, never statistics)
        stats = [None]  # device buffer of GroupNorm partial statistics once a consumer asks for them (_fuse_gn_statistics)
        plan.keep += [p, segs, ext, text]

        def launch(stream, p=p, segs=segs, ext=ext, stats=stats):
            L.check(lib.sfast_hip_gemm_ex(xp, segs, bp, None, rp, op, C.byref(p), C.byref(ext), stats[0].data_ptr() if stats[0] is not None else None,
                                          ws[0].data_ptr() if ws[0] is not None else None, ws[1], stream), name)

        def launch_with(stream, ws_ptr, ws_bytes, p=p, segs=segs, text=text):
            return lib.sfast_hip_gemm_ex(xp, segs, bp, None, rp, op, C.byref(p), C.byref(text), None, ws_ptr, ws_bytes, stream)

        if out_offset == 0 and ldo == N and lane == LANE_MAIN:
            plan.writer[id(out)] = dict(buf=out, name=name, p=p, ext=ext, stats=stats, conv=False, lane=lane)
        else:
            plan.writer.pop(id(out), None)

        wrows = (2 * N if geglu else N)
        flops = 2.0 * M * wrows * K
        nbytes = (M * K + wrows * K + wrows + M * N + (M * N if residual is not None else 0)) * self.esize
        # merged LoRA weights are written by the lora.merge launch at the head of the time-embedding lane: their first reader joins it
        needs = LANE_TEMB if (lane == LANE_MAIN and plan.lora is not None and any(id(w) in plan.lora["eff_ids"] for w in weights)) else None
        self._add(plan, kind or ("geglu" if geglu else ("gemv" if M <= 16 else "linear")), name, flops, nbytes, launch,
                  tune=(p, launch_with), lane=lane, needs=needs)
        if not geglu and M > 16:
            recs = self._offer_packed(plan, weights, (ext, text))
            if recs is not None:
                plan.ops[-1].packed = ((ext, text), recs)

    def _op_gemv_grouped(self, plan, name, x, weights, biases, out, M, K, ldx, ldo, *, out_offset=0, act=L.ACT_NONE, lane=LANE_MAIN):
        """out[m][out_offset + off_g + n] = act(x[m] . W_g[n] + b_g[n]) for every weight matrix of `weights`, one launch."""
        lib = self.lib
        p = L.GemvGroupedParams()
        p.dtype, p.M, p.K, p.n_groups = self.dt, M, K, len(weights)
        for i, w in enumerate(weights):
            if w.ndim != 2 or w.shape[1] != K or w.stride() != (w.stride(0), 1) or w.stride(0) != weights[0].stride(0):
                raise UnsupportedUNet(f"{name}: weight {i} is not a dense [N, {K}] matrix with the common row stride")
            p.n_rows[i] = w.shape[0]
        p.ldx, p.ldw, p.ldo, p.act, p.in_act = ldx, weights[0].stride(0), ldo, act, L.ACT_NONE
        wp = (C.c_void_p * len(weights))(*[w.data_ptr() for w in weights])
        bp = (C.c_void_p * len(weights))(*[(b.data_ptr() if b is not None else None) for b in biases])
        xp = x.data_ptr()
        op = out.data_ptr() + out_offset * self.esize
        plan.keep += [p, wp, bp]

        def launch(stream, p=p, wp=wp, bp=bp):
            L.check(lib.sfast_hip_gemv_grouped(xp, wp, bp, op, C.byref(p), stream), name)

        ntot = sum(w.shape[0] for w in weights)
        self._add(plan, "temb", name, 2.0 * M * ntot * K, (M * K + ntot * K + ntot + M * ntot) * self.esize, launch, lane=lane)

    def _op_conv(self, plan, name, x, x2, w, bias, out, B, H, W, C1, C2, Cout, k, stride, pad, *, ups=False, rowbias=None,
                 ld_rowbias=0, rowbias_offset=0, z=None, xs=None, os_=None, kind=None, act=L.ACT_NONE, pad_extra=0, kw=None, pad_w=None):
        lib = self.lib
        Cin = C1 + C2
        kw = k if kw is None else kw          # kernel height k, width kw (the temporal (3,1,1) convs run as 3 x 1 over [frames, pixels])
        pad_w = pad if pad_w is None else pad_w
        p = L.ConvParams()
        p.dtype, p.B, p.H, p.W, p.Cin, p.Cout, p.KH, p.KW = self.dt, B, H, W, Cin, Cout, k, kw
        p.stride_h = p.stride_w = stride
        p.pad_h, p.pad_w = pad, pad_w
        p.dil_h = p.dil_w = 1
        p.upsample2x, p.C1 = int(ups), C1
        Hin, Win = (2 * H, 2 * W) if ups else (H, W)
        Ho = (Hin + 2 * pad + pad_extra - (k - 1) - 1) // stride + 1
        Wo = (Win + 2 * pad_w + pad_extra - (kw - 1) - 1) // stride + 1
        p.pad_h_extra = p.pad_w_extra = pad_extra
        p.xs = (C.c_int64 * 4)(*(xs or (H * W * C1, W * C1, C1, 1)))
        p.x2s = (C.c_int64 * 4)(*((H * W * C2, W * C2, C2, 1) if C2 else (0, 0, 0, 0)))
        p.ws = (C.c_int64 * 4)(w.stride(0), w.stride(1), w.stride(2), w.stride(3))
        p.os = (C.c_int64 * 4)(*(os_ or (Ho * Wo * Cout, Wo * Cout, Cout, 1)))
        p.zs = (C.c_int64 * 4)(*((Ho * Wo * Cout, Wo * Cout, Cout, 1) if z is not None else (0, 0, 0, 0)))
        p.act, p.res_before_act, p.alpha = act, 1, 1.0
        p.ld_rowbias, p.variant, p.split_k = ld_rowbias, 0, 0
        self._need_ws(plan, lib.sfast_hip_conv2d_workspace_bytes(C.byref(p)))
        xp = x.data_ptr()
        x2p = x2.data_ptr() if x2 is not None else None
        wp = w.data_ptr()
        bp = bias.data_ptr() if bias is not None else None
        rbp = (rowbias.data_ptr() + rowbias_offset * self.esize) if rowbias is not None else None
        zp = z.data_ptr() if z is not None else None
        op = out.data_ptr()
        ws = plan.ws
        ext, text = L.EpilogueExt(0.0, 0, 0, EXT_FLAGS), L.EpilogueExt(0.0, 0, 0, EXT_FLAGS)
        stats = [None]
        plan.keep += [p, ext, text]

        def launch(stream, p=p, ext=ext, stats=stats):
            L.check(lib.sfast_hip_conv2d_ex(xp, x2p, wp, bp, rbp, zp, op, C.byref(p), C.byref(ext), stats[0].data_ptr() if stats[0] is not None else None,
                                            ws[0].data_ptr() if ws[0] is not None else None, ws[1], stream), name)

        def launch_with(stream, ws_ptr, ws_bytes, p=p, text=text):
            return lib.sfast_hip_conv2d_ex(xp, x2p, wp, bp, rbp, zp, op, C.byref(p), C.byref(text), None, ws_ptr, ws_bytes, stream)

        if os_ is None:
            plan.writer[id(out)] = dict(buf=out, name=name, p=p, ext=ext, stats=stats, conv=True, lane=LANE_MAIN)
        else:
            plan.writer.pop(id(out), None)

        M = B * Ho * Wo
        flops = 2.0 * M * Cout * Cin * k * kw
        nbytes = (B * H * W * Cin + Cout * Cin * k * kw + Cout + M * Cout + (M * Cout if z is not None else 0)) * self.esize
        self._add(plan, kind or ("conv3x3" if k == 3 else "conv1x1"), name, flops, nbytes, launch, tune=(p, launch_with),
                  needs=LANE_TEMB if rowbias is not None else None)
        if not ups and C1 % 64 == 0 and C2 % 64 == 0:
            recs = self._offer_packed(plan, [w], (ext, text))
            if recs is not None:
                plan.ops[-1].packed = ((ext, text), recs)
        return Ho, Wo

    def _gnconv_params(self, x2, w, B, H, W, C1, C2, Cout, eps, ld_rowbias, z):
        q = L.GnConvParams()
        p = q.conv
        Cin = C1 + C2
        p.dtype, p.B, p.H, p.W, p.Cin, p.Cout, p.KH, p.KW = self.dt, B, H, W, Cin, Cout, 3, 3
        p.stride_h = p.stride_w = p.pad_h = p.pad_w = p.dil_h = p.dil_w = 1
        p.upsample2x, p.C1 = 0, C1
        p.xs = (C.c_int64 * 4)(H * W * C1, W * C1, C1, 1)
        p.x2s = (C.c_int64 * 4)(*((H * W * C2, W * C2, C2, 1) if C2 else (0, 0, 0, 0)))
        p.ws = (C.c_int64 * 4)(w.stride(0), w.stride(1), w.stride(2), w.stride(3))
        p.os = (C.c_int64 * 4)(H * W * Cout, W * Cout, Cout, 1)
        p.zs = (C.c_int64 * 4)(*((H * W * Cout, W * Cout, Cout, 1) if z is not None else (0, 0, 0, 0)))
        p.act, p.res_before_act, p.alpha = L.ACT_NONE, 1, 1.0
        p.ld_rowbias, p.variant, p.split_k = ld_rowbias, 0, 0
        q.groups, q.eps, q.gn_act = self.groups, float(eps), L.ACT_SILU
        return q

    def _op_gnconv(self, plan, name, x, x2, norm_prefix, w, bias, out, B, H, W, C1, C2, Cout, *, rowbias=None, ld_rowbias=0,
                   rowbias_offset=0, z=None):
        """GroupNorm+SiLU -> 3x3 conv as ONE launch (sfast_hip_gn_conv2d, csrc/gnconv.hip) when the library covers the shape -- the
        low-resolution levels, where the conv streams weights and the separate normalisation launch costs as much as a third of that
        stream. Returns False (and emits nothing) otherwise: the caller then builds the GroupNorm op and the conv op."""
        if not FUSE_GN_CONV:
            return False
        lib = self.lib
        eps = self.norm_eps.get(norm_prefix, self.eps)
        q = self._gnconv_params(x2, w, B, H, W, C1, C2, Cout, eps, ld_rowbias, z)
        if not lib.sfast_hip_gn_conv2d_supported(C.byref(q)):
            return False
        gamma, beta = self.params[norm_prefix + ".weight"], self.params[norm_prefix + ".bias"]
        self._need_ws(plan, lib.sfast_hip_gn_conv2d_workspace_bytes(C.byref(q)))
        xp, x2p = x.data_ptr(), (x2.data_ptr() if x2 is not None else None)
        gp, bep, wp, bp = gamma.data_ptr(), beta.data_ptr(), w.data_ptr(), (bias.data_ptr() if bias is not None else None)
        rbp = (rowbias.data_ptr() + rowbias_offset * self.esize) if rowbias is not None else None
        zp = z.data_ptr() if z is not None else None
        op = out.data_ptr()
        ws = plan.ws
        plan.keep.append(q)

        def launch(stream, q=q):
            L.check(lib.sfast_hip_gn_conv2d(xp, x2p, gp, bep, wp, bp, rbp, zp, op, C.byref(q), ws[0].data_ptr() if ws[0] is not None else None,
                                            ws[1], stream), name)

        plan.writer.pop(id(out), None)   # this producer emits no GroupNorm statistics: a consumer GroupNorm computes its own
        Cin, M = C1 + C2, B * H * W
        flops = 2.0 * M * Cout * Cin * 9
        nbytes = (2 * M * Cin + 2 * Cin + Cout * Cin * 9 + Cout + M * Cout + (M * Cout if z is not None else 0)) * self.esize
        self._add(plan, "gnconv3x3", name, flops, nbytes, launch, needs=LANE_TEMB if rowbias is not None else None)
        return True

    def _conv_in(self, plan, sample, h, B, H, W, c0):
        """conv_in on the NCHW latent. With fewer than 8 input channels (4 for SD) the conv read through NCHW strides runs on the
        generic small-channel kernel (37 us at 2 x 64 x 64: 0.7 % of the SD1.5 step); padded to 8 channels it is an MFMA implicit
        GEMM with K = 72. Two strided copies per step make the padded operands: the sample into a zero-initialised NHWC8 buffer
        and the LIVE weight (the contract: parameters are read at every launch) into a zero-initialised [Cout][3][3][8] image; the
        padding lanes are written once, at plan build, and belong to no pool (nothing else may touch them)."""
        lib, P, cin = self.lib, self.params, self.in_ch
        w = P["conv_in.weight"]
        if not (cin < 8 and w.shape[2] == 3 and w.shape[3] == 3 and w.stride(2) == 3 * w.stride(3)) or os.environ.get("SFAST_CONV_IN_PAD", "1") == "0":
            self._op_conv(plan, "conv_in", sample, None, w, P["conv_in.bias"], h, B, H, W, cin, 0, c0, 3, 1, 1,
                          xs=(cin * H * W, W, 1, H * W), kind="conv_in")
            return
        x8 = torch.zeros(B * H * W * 8, dtype=self.dtype, device=self.device)
        w8 = torch.zeros(c0 * 9 * 8, dtype=self.dtype, device=self.device)
        plan.keep += [x8, w8]

        def copy(name, src, dst, shape, sst, dst_st, lane=LANE_MAIN):
            cp = L.CopyParams()
            cp.elem_bytes, cp.ndim = self.esize, 3
            cp.shape = (C.c_int64 * 4)(*shape, 1)
            cp.src_strides = (C.c_int64 * 4)(*sst, 0)
            cp.dst_strides = (C.c_int64 * 4)(*dst_st, 0)
            plan.keep.append(cp)
            sp_, dp_ = src.data_ptr(), dst.data_ptr()
            nbytes = 2.0 * shape[0] * shape[1] * shape[2] * self.esize
            self._add(plan, "misc", name, 0.0, nbytes, lambda s, cp=cp: L.check(lib.sfast_hip_strided_copy(sp_, dp_, C.byref(cp), s), name), lane=lane)

        copy("conv_in.weight.pad8", w, w8, (c0, 9, cin), (w.stride(0), w.stride(3), w.stride(1)), (72, 8, 1))
        copy("sample.to_nhwc8", sample, x8, (B, H * W, cin), (cin * H * W, 1, H * W), (H * W * 8, 8, 1))
        w8v = torch.as_strided(w8, (c0, 8, 3, 3), (72, 1, 24, 8))
        self._op_conv(plan, "conv_in", x8, None, w8v, P["conv_in.bias"], h, B, H, W, 8, 0, c0, 3, 1, 1, kind="conv_in")

    def _op_add_nchw(self, plan, name, src_nchw, dst_nhwc, B, Cc, Hh, Ww):
        """dst[b][h][w][c] += src[b][c][h][w] (dense NHWC buffer += NCHW tensor)."""
        lib = self.lib
        ap = L.AddParams()
        ap.dtype, ap.ndim = self.dt, 4
        ap.shape = (C.c_int64 * 4)(B, Hh, Ww, Cc)
        ap.src_strides = (C.c_int64 * 4)(Cc * Hh * Ww, Ww, 1, Hh * Ww)
        ap.dst_strides = (C.c_int64 * 4)(Hh * Ww * Cc, Ww * Cc, Cc, 1)
        plan.keep.append(ap)
        sp, dp = src_nchw.data_ptr(), dst_nhwc.data_ptr()
        plan.writer.pop(id(dst_nhwc), None)  # modified in place: statistics of its producer no longer describe it
        self._add(plan, "misc", name, 0.0, 3.0 * B * Cc * Hh * Ww * self.esize,
                  lambda s, ap=ap: L.check(lib.sfast_hip_add_strided(sp, dp, C.byref(ap), s), name))

    def _op_attn(self, plan, name, q, k, v, out, B, Hh, Sq, Skv, D, qs, ks, vs, os_, q_off=0, k_off=0, v_off=0, bias=None, bias_strides=None,
                 out_off=0, kind=None, variant=0):
        lib = self.lib
        p = L.AttnParams()
        p.dtype, p.B, p.H, p.Sq, p.Skv, p.D = self.dt, B, Hh, Sq, Skv, D
        p.qs = (C.c_int64 * 3)(*qs)
        p.ks = (C.c_int64 * 3)(*ks)
        p.vs = (C.c_int64 * 3)(*vs)
        p.os = (C.c_int64 * 3)(*os_)
        p.scale = float(D) ** -0.5
        p.variant = variant
        qp = q.data_ptr() + q_off * self.esize
        kp = k.data_ptr() + k_off * self.esize
        vp = v.data_ptr() + v_off * self.esize
        op = out.data_ptr() + out_off * self.esize
        plan.keep.append(p)
        plan.writer.pop(id(out), None)

        bptr = bias.data_ptr() if bias is not None else None
        bstr = (C.c_int64 * 3)(*bias_strides) if bias is not None else None
        plan.keep.append(bstr)

        def launch(stream, p=p):
            if bptr is not None:  # additive bias broadcast over heads / queries (diffusers encoder_attention_mask)
                L.check(lib.sfast_hip_attention_bias(qp, kp, vp, bptr, bstr, op, C.byref(p), stream), name)
            else:
                L.check(lib.sfast_hip_attention(qp, kp, vp, op, C.byref(p), stream), name)

        flops = 4.0 * B * Hh * Sq * Skv * D
        nbytes = (2.0 * B * Sq * Hh * D + 2.0 * B * Skv * Hh * D) * self.esize
        cross = not (Sq == Skv and q is k)
        self._add(plan, kind or ("attn_cross" if cross else "attn_self"), name, flops, nbytes, launch, needs=LANE_KV if cross else None)

    # ------------------------------------------------------------------------------------------
    # network pieces
    # ------------------------------------------------------------------------------------------
    def _resnet(self, plan, pre, x, x2, C1, C2, Cout, B, H, W, temb_all, temb_ld, temb_off):
        pool, P = plan.pool, self.params
        M = B * H * W
        Cin = C1 + C2
        h1 = pool.get(M * Cout)
        if not self._op_gnconv(plan, pre + ".norm1+conv1", x, x2, pre + ".norm1", P[pre + ".conv1.weight"], P[pre + ".conv1.bias"], h1,
                               B, H, W, C1, C2, Cout, rowbias=temb_all, ld_rowbias=temb_ld, rowbias_offset=temb_off):
            n1 = pool.get(M * Cin)
            self._op_gn(plan, pre + ".norm1", x, x2, C1, Cin, B, H * W, n1, self.eps, True, pre + ".norm1")
            self._op_conv(plan, pre + ".conv1", n1, None, P[pre + ".conv1.weight"], P[pre + ".conv1.bias"], h1, B, H, W, Cin, 0,
                          Cout, 3, 1, 1, rowbias=temb_all, ld_rowbias=temb_ld, rowbias_offset=temb_off)
            pool.put(n1)
        # conv2 with norm2 fused: decided before the shortcut is emitted (plan order stays norm2, shortcut, conv2)
        q2 = self._gnconv_params(None, P[pre + ".conv2.weight"], B, H, W, Cout, 0, Cout, self.norm_eps.get(pre + ".norm2", self.eps), 0, x)
        fuse2 = FUSE_GN_CONV and bool(self.lib.sfast_hip_gn_conv2d_supported(C.byref(q2)))
        n2 = None
        if not fuse2:
            n2 = pool.get(M * Cout)
            self._op_gn(plan, pre + ".norm2", h1, None, Cout, Cout, B, H * W, n2, self.eps, True, pre + ".norm2")
            pool.put(h1)
        if (pre + ".conv_shortcut.weight") in P:
            sc = pool.get(M * Cout)
            wsc, bsc = P[pre + ".conv_shortcut.weight"], P[pre + ".conv_shortcut.bias"]
            if x2 is None:
                self._op_gemm(plan, pre + ".conv_shortcut", x, [_as2d(wsc, Cout, Cin)], bsc, sc, M, Cout, Cin, Cin, Cout,
                              kind="conv1x1")
            else:
                self._op_conv(plan, pre + ".conv_shortcut", x, x2, wsc, bsc, sc, B, H, W, C1, C2, Cout, 1, 1, 0)
            res, own = sc, True
        else:
            if x2 is not None or Cin != Cout:
                raise UnsupportedUNet(f"{pre}: no conv_shortcut for {Cin}->{Cout}")
            res, own = x, False
        out = pool.get(M * Cout)
        if fuse2:
            ok = self._op_gnconv(plan, pre + ".norm2+conv2", h1, None, pre + ".norm2", P[pre + ".conv2.weight"], P[pre + ".conv2.bias"], out,
                                 B, H, W, Cout, 0, Cout, z=res)
            assert ok
            pool.put(h1)
        else:
            self._op_conv(plan, pre + ".conv2", n2, None, P[pre + ".conv2.weight"], P[pre + ".conv2.bias"], out, B, H, W, Cout, 0,
                          Cout, 3, 1, 1, z=res)
            pool.put(n2)
        if own:
            pool.put(res)
        return out

    def _transformer(self, plan, pre, x, Cc, B, H, W, heads, depth, ctx, S_ctx):
        pool, P = plan.pool, (plan.P if plan.P is not None else self.params)
        M = B * H * W
        S = H * W
        D = Cc // heads
        g = pool.get(M * Cc)
        self._op_gn(plan, pre + ".norm", x, None, Cc, Cc, B, S, g, 1e-6, False, pre + ".norm")
        t = pool.get(M * Cc)
        w_in = _as2d(P[pre + ".proj_in.weight"], Cc, Cc)
        self._op_gemm(plan, pre + ".proj_in", g, [w_in], P[pre + ".proj_in.bias"], t, M, Cc, Cc, Cc, Cc,
                      kind="linear" if self.linear_proj else "conv1x1")
        pool.put(g)
        for d in range(depth):
            bp = f"{pre}.transformer_blocks.{d}"
            n = pool.get(M * Cc)
            # --- self attention: LN -> fused QKV GEMM -> flash attention (in-place [B,S,H,D] view) -> out proj + residual
            self._op_ln(plan, bp + ".norm1", t, n, M, Cc, bp + ".norm1")
            qkv = pool.get(M * 3 * Cc)
            self._op_gemm(plan, bp + ".attn1.to_qkv", n, [P[bp + ".attn1.to_q.weight"], P[bp + ".attn1.to_k.weight"],
                                                          P[bp + ".attn1.to_v.weight"]], None, qkv, M, 3 * Cc, Cc, Cc, 3 * Cc)
            a = pool.get(M * Cc)
            st = (S * 3 * Cc, 3 * Cc, D)
            sbias = plan.static_in.get("attention_bias")
            if sbias is not None and sbias.shape[1] != S:
                # diffusers hands the UNet-level attention_mask to EVERY self-attention layer; one whose token count differs pads it to the
                # sum of both lengths and fails in scaled_dot_product_attention (Attention.prepare_attention_mask) -- nothing to restate
                raise UnsupportedUNet(f"attention_mask of {sbias.shape[1]} keys on a self-attention layer with {S} tokens")
            self._op_attn(plan, bp + ".attn1", qkv, qkv, qkv, a, B, heads, S, S, D, st, st, st, (S * Cc, Cc, D),
                          q_off=0, k_off=Cc, v_off=2 * Cc, bias=sbias, bias_strides=(sbias.stride(0), 0, 0) if sbias is not None else None,
                          kind="attn_self")
            pool.put(qkv)
            self._op_gemm(plan, bp + ".attn1.to_out", a, [P[bp + ".attn1.to_out.0.weight"]], P[bp + ".attn1.to_out.0.bias"], t,
                          M, Cc, Cc, Cc, Cc, residual=t, ldr=Cc)
            # --- cross attention
            self._op_ln(plan, bp + ".norm2", t, n, M, Cc, bp + ".norm2")
            q = pool.get(M * Cc)
            self._op_gemm(plan, bp + ".attn2.to_q", n, [P[bp + ".attn2.to_q.weight"]], None, q, M, Cc, Cc, Cc, Cc)
            # K/V of the text context depend on nothing inside the UNet: dedicated buffer (never recycled) so the
            # projection can be hoisted onto the side lane of the graph
            kv = torch.empty(B * S_ctx * 2 * Cc, dtype=self.dtype, device=self.device)
            plan.keep.append(kv)
            # deferred: all K/V projections of equal width become ONE grouped launch at the top of the plan (_emit_kv_groups)
            plan.kv_requests.append((bp + ".attn2.to_kv", P[bp + ".attn2.to_k.weight"], P[bp + ".attn2.to_v.weight"], kv, Cc))
            skv = (S_ctx * 2 * Cc, 2 * Cc, D)
            ebias = plan.static_in.get("encoder_attention_bias")
            self._op_attn(plan, bp + ".attn2", q, kv, kv, a, B, heads, S, S_ctx, D, (S * Cc, Cc, D), skv, skv, (S * Cc, Cc, D),
                          k_off=0, v_off=Cc, bias=ebias, bias_strides=(ebias.stride(0), 0, 0) if ebias is not None else None)
            if plan.ip is not None:
                # IP-Adapter (diffusers IPAdapterAttnProcessor2_0): the same queries attend to the projected image tokens with the
                # adapter's own to_k_ip / to_v_ip -- a second softmax, not more keys -- and a += scale * that, before to_out
                for i, (_, S_ip) in enumerate(plan.ip["tokens"]):
                    sc = plan.ip["scales"].get(bp + ".attn2", (1.0,) * len(plan.ip["tokens"]))[i]
                    if sc == 0.0:
                        continue  # diffusers skips the adapter as well
                    kvi = torch.empty(B * S_ip * 2 * Cc, dtype=self.dtype, device=self.device)
                    plan.keep.append(kvi)
                    plan.ip["requests"][i].append((f"{bp}.attn2.processor.to_kv_ip.{i}", P[f"{bp}.attn2.processor.to_k_ip.{i}.weight"],
                                                   P[f"{bp}.attn2.processor.to_v_ip.{i}.weight"], kvi, Cc))
                    a2 = pool.get(M * Cc)
                    si_ = (S_ip * 2 * Cc, 2 * Cc, D)
                    self._op_attn(plan, f"{bp}.attn2.ip_adapter.{i}", q, kvi, kvi, a2, B, heads, S, S_ip, D, (S * Cc, Cc, D), si_, si_,
                                  (S * Cc, Cc, D), k_off=0, v_off=Cc, kind="attn_cross")
                    self._op_mix(plan, f"{bp}.attn2.ip_adapter.{i}.add", a, a2, None, a, M, Cc, wx=1.0, wy=sc)
                    pool.put(a2)
            pool.put(q)
            self._op_gemm(plan, bp + ".attn2.to_out", a, [P[bp + ".attn2.to_out.0.weight"]], P[bp + ".attn2.to_out.0.bias"], t,
                          M, Cc, Cc, Cc, Cc, residual=t, ldr=Cc)
            pool.put(a)
            # --- feed forward: LN -> Linear+GEGLU (dual GEMM) -> Linear + residual
            self._op_ln(plan, bp + ".norm3", t, n, M, Cc, bp + ".norm3")
            gg = pool.get(M * 4 * Cc)
            self._op_gemm(plan, bp + ".ff.geglu", n, [P[bp + ".ff.net.0.proj.weight"]], P[bp + ".ff.net.0.proj.bias"], gg,
                          M, 4 * Cc, Cc, Cc, 4 * Cc, geglu=True)
            pool.put(n)
            self._op_gemm(plan, bp + ".ff.out", gg, [P[bp + ".ff.net.2.weight"]], P[bp + ".ff.net.2.bias"], t,
                          M, Cc, 4 * Cc, 4 * Cc, Cc, residual=t, ldr=Cc)
            pool.put(gg)
        out = pool.get(M * Cc)
        w_out = _as2d(P[pre + ".proj_out.weight"], Cc, Cc)
        self._op_gemm(plan, pre + ".proj_out", t, [w_out], P[pre + ".proj_out.bias"], out, M, Cc, Cc, Cc, Cc, residual=x, ldr=Cc,
                      kind="linear" if self.linear_proj else "conv1x1")
        pool.put(t)
        return out

    def _resnet_names(self):
        names = []
        for i in range(len(self.down_types)):
            names += [f"down_blocks.{i}.resnets.{j}" for j in range(self.layers)]
        names += ["mid_block.resnets.0", "mid_block.resnets.1"]
        for i in range(len(self.up_types)):
            names += [f"up_blocks.{i}.resnets.{j}" for j in range(self.layers + 1)]
        return names

    # ------------------------------------------------------------------------------------------
    def build_plan(self, B, H, W, S_ctx, ctrl=False, enc_mask=False, tcond=False, ip=None, self_mask=0):
        """`ctrl`: the plan also takes ControlNet residuals (one NCHW tensor per skip connection + one for the mid block,
        diffusers `down_block_additional_residuals` / `mid_block_additional_residual`) as static inputs.
        `tcond`: the plan takes `timestep_cond` [B, time_cond_proj_dim] (LCM-distilled UNets).
        `ip`: (images per adapter, scales) as ip_signature() returns it -- required when an IP-Adapter is loaded: the plan then takes
        `image_embeds` (one [B, images, D] tensor per adapter) and runs ImageProjection + the decoupled image cross-attention."""
        if (ip is not None) != bool(self.ip_proj):
            if ip is None:
                if self.ip_external:
                    raise UnsupportedUNet("an IP-Adapter with an external image projection needs the `ip` signature (tokens per adapter)")
                ip = ((1,) * len(self.ip_proj), self.ip_scales())  # one image per adapter, live scales
            else:
                raise UnsupportedUNet("image_embeds given, but no IP-Adapter is loaded (encoder_hid_dim_type is not 'ip_image_proj')")
        if tcond and self.tcond_dim is None:
            raise UnsupportedUNet("timestep_cond given, but the UNet has no time_embedding.cond_proj (time_cond_proj_dim is None)")
        self.host.init_device(self.device)
        nlev = len(self.boc)
        if H % (1 << (nlev - 1)) or W % (1 << (nlev - 1)):
            raise UnsupportedUNet(f"latent {H}x{W} not divisible by {1 << (nlev - 1)}")
        if self_mask:
            # checked BEFORE any op is emitted: the same refusal from inside _transformer left the packed-weight records of
            # the ops emitted so far with users > 0 -- pinned buffers that sync_packed() kept re-packing for a plan that never existed
            lv = {i for i, t in enumerate(self.down_types) if t.startswith("CrossAttn")} | {nlev - 1 - j for j, t in enumerate(self.up_types) if t.startswith("CrossAttn")}
            lv.add(nlev - 1)  # the mid block
            for i in sorted(lv):
                S = (H >> i) * (W >> i)
                if S != self_mask:
                    raise UnsupportedUNet(f"attention_mask of {self_mask} keys on a self-attention layer with {S} tokens")
        P = self.params
        dev, dt = self.device, self.dtype
        plan = UNetPlan(self, B, H, W, S_ctx)
        pool = plan.pool = _Pool(dev, dt)
        pool.writer = plan.writer
        plan.P = self.params
        if self.lora:
            self._emit_lora(plan)
        # static inputs / output
        sample = torch.zeros((B, self.in_ch, H, W), dtype=dt, device=dev)
        tbuf = torch.zeros((B,), dtype=torch.float32, device=dev)
        ctx = torch.zeros((B, S_ctx, self.ctx_dim), dtype=dt, device=dev)
        out = torch.zeros((B, self.out_ch, H, W), dtype=dt, device=dev)
        plan.static_in.update({"sample": sample, "timestep": tbuf, "encoder_hidden_states": ctx})
        if enc_mask:
            # additive key bias of the cross-attention layers: (1 - encoder_attention_mask) * -10000 as diffusers'
            # UNet2DConditionModel.forward builds it, [B, S_ctx] broadcast over heads and queries (row padded to 8 halves)
            ld = (S_ctx + 7) // 8 * 8
            plan.static_in["encoder_attention_bias"] = torch.zeros((B, ld), dtype=dt, device=dev)[:, :S_ctx]
        if self_mask:
            # additive key bias of the SELF-attention layers (diffusers `attention_mask`, a keep-mask of `self_mask` keys): same form
            ld = (int(self_mask) + 7) // 8 * 8
            plan.static_in["attention_bias"] = torch.zeros((B, ld), dtype=dt, device=dev)[:, :int(self_mask)]
        plan.static_out = out
        lib = self.lib
        if ip is not None:
            n_img, scales = ip
            if len(n_img) != len(self.ip_proj):
                raise UnsupportedUNet(f"{len(n_img)} image_embeds tensors for {len(self.ip_proj)} IP-Adapters")
            if self.ip_external:  # n_img = projected tokens per adapter; the token buffers themselves are the static inputs
                toks = [(torch.zeros((B, int(n), self.ctx_dim), dtype=dt, device=dev), int(n)) for n in n_img]
                plan.static_in["ip_hidden_states"] = [t for t, _ in toks]
            else:
                plan.static_in["image_embeds"] = [torch.zeros((B, int(n), dimg), dtype=dt, device=dev) for n, (_, _, dimg) in zip(n_img, self.ip_proj)]
                toks = [(torch.zeros((B, int(n) * T_, self.ctx_dim), dtype=dt, device=dev), int(n) * T_) for n, (_, T_, _) in zip(n_img, self.ip_proj)]
            plan.ip = dict(tokens=toks, requests=[[] for _ in self.ip_proj], scales=dict(scales or ()))

        # ---- time embedding --------------------------------------------------------------------
        c0 = self.boc[0]
        T = self.temb_dim
        t_emb = pool.get(B * c0)
        tp = L.TembParams(self.dt, B, c0, int(self.flip), self.freq_shift, 10000.0)
        plan.keep.append(tp)
        tb_ptr, te_ptr = tbuf.data_ptr(), t_emb.data_ptr()
        self._add(plan, "misc", "timestep_embedding", 0.0, B * c0 * 2.0,
                  lambda s, tp=tp: L.check(lib.sfast_hip_timestep_embedding(tb_ptr, te_ptr, C.byref(tp), s), "timestep_embedding"),
                  lane=LANE_TEMB)
        if tcond:
            # TimestepEmbedding.forward: sample = sample + cond_proj(condition) -- one more GEMV whose residual operand is the sinusoid
            tc = torch.zeros((B, int(self.tcond_dim)), dtype=dt, device=dev)
            plan.static_in["timestep_cond"] = tc
            t_emb2 = pool.get(B * c0)
            self._op_gemm(plan, "time_embedding.cond_proj", tc, [P["time_embedding.cond_proj.weight"]], None, t_emb2, B, c0, int(self.tcond_dim),
                          int(self.tcond_dim), c0, residual=t_emb, ldr=c0, kind="temb", lane=LANE_TEMB)
            t_emb = t_emb2
        e1 = pool.get(B * T)
        self._op_gemm(plan, "time_embedding.linear_1", t_emb, [P["time_embedding.linear_1.weight"]], P["time_embedding.linear_1.bias"],
                      e1, B, T, c0, c0, T, act=L.ACT_SILU, kind="temb", lane=LANE_TEMB)
        act_emb = pool.get(B * T)
        aug = None  # sum of the embeddings diffusers adds to `emb` after the time MLP: class embedding, then text_time
        if self.add_type == "text_time":
            Din = P["add_embedding.linear_1.weight"].shape[1]
            td = self.add_time_dim
            text_embeds = torch.zeros((B, Din - 6 * td), dtype=dt, device=dev)
            time_ids = torch.zeros((B * 6,), dtype=torch.float32, device=dev)
            plan.static_in["text_embeds"] = text_embeds
            plan.static_in["time_ids"] = time_ids
            tide = pool.get(B * 6 * td)
            tp2 = L.TembParams(self.dt, B * 6, td, int(self.flip), self.freq_shift, 10000.0)
            plan.keep.append(tp2)
            ti_ptr, tide_ptr = time_ids.data_ptr(), tide.data_ptr()
            self._add(plan, "misc", "add_time_ids_embedding", 0.0, B * 6 * td * 2.0,
                      lambda s, tp2=tp2: L.check(lib.sfast_hip_timestep_embedding(ti_ptr, tide_ptr, C.byref(tp2), s), "time_ids"),
                      lane=LANE_TEMB)
            add_in = pool.get(B * Din)
            ntext = Din - 6 * td

            def copy2d(name, src_ptr, rows, cols, src_ld, dst_ptr, dst_ld):
                cp = L.CopyParams()
                cp.elem_bytes, cp.ndim = 2, 2
                cp.shape = (C.c_int64 * 4)(rows, cols, 1, 1)
                cp.src_strides = (C.c_int64 * 4)(src_ld, 1, 0, 0)
                cp.dst_strides = (C.c_int64 * 4)(dst_ld, 1, 0, 0)
                plan.keep.append(cp)
                self._add(plan, "misc", name, 0.0, rows * cols * 4.0,
                          lambda s, cp=cp: L.check(lib.sfast_hip_strided_copy(src_ptr, dst_ptr, C.byref(cp), s), name),
                          lane=LANE_TEMB)

            copy2d("add_in.text", text_embeds.data_ptr(), B, ntext, ntext, add_in.data_ptr(), Din)
            copy2d("add_in.time", tide.data_ptr(), B, 6 * td, 6 * td, add_in.data_ptr() + ntext * 2, Din)
            a1 = pool.get(B * T)
            self._op_gemm(plan, "add_embedding.linear_1", add_in, [P["add_embedding.linear_1.weight"]], P["add_embedding.linear_1.bias"],
                          a1, B, T, Din, Din, T, act=L.ACT_SILU, kind="temb", lane=LANE_TEMB)
            aug = pool.get(B * T)
            self._op_gemm(plan, "add_embedding.linear_2", a1, [P["add_embedding.linear_2.weight"]], P["add_embedding.linear_2.bias"],
                          aug, B, T, T, T, T, kind="temb", lane=LANE_TEMB)
        if self.class_type is not None:
            # emb = emb + class_embedding(class_labels): "timestep" embeds the labels through the same sinusoid as t, "projection"
            # takes a float vector; the MLP's second Linear carries the text_time embedding (if any) as its residual operand
            Dc = P["class_embedding.linear_1.weight"].shape[1]
            if self.class_type == "timestep":
                clab = torch.zeros((B,), dtype=torch.float32, device=dev)
                ce_in = pool.get(B * Dc)
                tp3 = L.TembParams(self.dt, B, Dc, int(self.flip), self.freq_shift, 10000.0)
                plan.keep.append(tp3)
                cl_ptr, ce_ptr = clab.data_ptr(), ce_in.data_ptr()
                self._add(plan, "misc", "class_labels_embedding", 0.0, B * Dc * 2.0,
                          lambda s, tp3=tp3: L.check(lib.sfast_hip_timestep_embedding(cl_ptr, ce_ptr, C.byref(tp3), s), "class_labels"),
                          lane=LANE_TEMB)
            else:
                clab = torch.zeros((B, Dc), dtype=dt, device=dev)
                ce_in = clab
            plan.static_in["class_labels"] = clab
            c1 = pool.get(B * T)
            self._op_gemm(plan, "class_embedding.linear_1", ce_in, [P["class_embedding.linear_1.weight"]], P["class_embedding.linear_1.bias"],
                          c1, B, T, Dc, Dc, T, act=L.ACT_SILU, kind="temb", lane=LANE_TEMB)
            cemb = pool.get(B * T)
            self._op_gemm(plan, "class_embedding.linear_2", c1, [P["class_embedding.linear_2.weight"]], P["class_embedding.linear_2.bias"],
                          cemb, B, T, T, T, T, residual=aug, ldr=T if aug is not None else 0, kind="temb", lane=LANE_TEMB)
            aug = cemb
        # act_emb = silu(emb [+ aug]): the embedding is only ever consumed through SiLU (ResnetBlock2D)
        self._op_gemm(plan, "time_embedding.linear_2", e1, [P["time_embedding.linear_2.weight"]], P["time_embedding.linear_2.bias"],
                      act_emb, B, T, T, T, T, act=L.ACT_SILU, residual=aug, ldr=T if aug is not None else 0,
                      res_before_act=aug is not None, kind="temb", lane=LANE_TEMB)
        # every resnet's time_emb_proj depends only on t: hoisted to the top of the graph
        rnames = self._resnet_names()
        offs, tot = {}, 0
        for rn in rnames:
            offs[rn] = tot
            tot += P[rn + ".time_emb_proj.weight"].shape[0]
        temb_all = pool.get(B * tot)
        if B <= 64 and T % 8 == 0:
            # ONE grouped GEMV launch per <= 32 resnets: same input, live weights read in place
            for g0 in range(0, len(rnames), L.MAX_GROUPS):
                grp = rnames[g0:g0 + L.MAX_GROUPS]
                self._op_gemv_grouped(plan, f"time_emb_proj[{g0}:{g0 + len(grp)}]", act_emb,
                                      [P[rn + ".time_emb_proj.weight"] for rn in grp], [P[rn + ".time_emb_proj.bias"] for rn in grp],
                                      temb_all, B, T, T, tot, out_offset=offs[grp[0]], lane=LANE_TEMB)
        else:
            for rn in rnames:
                w = P[rn + ".time_emb_proj.weight"]
                self._op_gemm(plan, rn + ".time_emb_proj", act_emb, [w], P[rn + ".time_emb_proj.bias"], temb_all, B, w.shape[0], T, T, tot,
                              out_offset=offs[rn], kind="temb", lane=LANE_TEMB)

        # ---- conv_in (reads the NCHW sample through strides, writes NHWC) -----------------------
        h = pool.get(B * H * W * c0)
        self._conv_in(plan, sample, h, B, H, W, c0)
        if self.is_controlnet:
            h = self._controlnet_cond_embedding(plan, h, B, H, W, c0)
        skips = [(h, c0)]
        skip_dims = [(c0, H, W)]  # (channels, height, width) of every skip tensor, in creation order
        ch = c0
        cH, cW = H, W
        # ---- down ---------------------------------------------------------------------------------
        for i, t in enumerate(self.down_types):
            co = self.boc[i]
            for j in range(self.layers):
                rn = f"down_blocks.{i}.resnets.{j}"
                hn = self._resnet(plan, rn, h, None, ch, 0, co, B, cH, cW, temb_all, tot, offs[rn])
                ch = co
                if t == "CrossAttnDownBlock2D":
                    ha = self._transformer(plan, f"down_blocks.{i}.attentions.{j}", hn, co, B, cH, cW, self.heads[i], self.depth[i], ctx, S_ctx)
                    pool.put(hn)
                    hn = ha
                h = hn
                skips.append((h, ch))
                skip_dims.append((ch, cH, cW))
            if i < nlev - 1:
                dn = f"down_blocks.{i}.downsamplers.0.conv"
                hd = pool.get(B * (cH // 2) * (cW // 2) * ch)
                self._op_conv(plan, dn, h, None, P[dn + ".weight"], P[dn + ".bias"], hd, B, cH, cW, ch, 0, ch, 3, 2, 1)
                cH, cW = cH // 2, cW // 2
                h = hd
                skips.append((h, ch))
                skip_dims.append((ch, cH, cW))
        # ---- mid -----------------------------------------------------------------------------------
        rn = "mid_block.resnets.0"
        hm = self._resnet(plan, rn, h, None, ch, 0, ch, B, cH, cW, temb_all, tot, offs[rn])
        ha = self._transformer(plan, "mid_block.attentions.0", hm, ch, B, cH, cW, self.heads[-1], self.depth[-1], ctx, S_ctx)
        pool.put(hm)
        rn = "mid_block.resnets.1"
        h = self._resnet(plan, rn, ha, None, ch, 0, ch, B, cH, cW, temb_all, tot, offs[rn])
        pool.put(ha)
        # (the last skip tensor is the mid-block input; it stays alive in `skips`)
        if self.is_controlnet:
            # ControlNetModel: one 1x1 "zero conv" per skip tensor + one on the mid-block output, written NCHW (what
            # UNet2DConditionModel.forward takes as down_block_additional_residuals / mid_block_additional_residual)
            outs = []
            for k, ((buf, _), (cc, sh_, sw_)) in enumerate(zip(skips, skip_dims)):
                o = torch.zeros((B, cc, sh_, sw_), dtype=dt, device=dev)
                zn = f"controlnet_down_blocks.{k}"
                self._op_conv(plan, zn, buf, None, P[zn + ".weight"], P[zn + ".bias"], o, B, sh_, sw_, cc, 0, cc, 1, 1, 0,
                              os_=(cc * sh_ * sw_, sw_, 1, sh_ * sw_))
                outs.append(o)
            om = torch.zeros((B, ch, cH, cW), dtype=dt, device=dev)
            self._op_conv(plan, "controlnet_mid_block", h, None, P["controlnet_mid_block.weight"], P["controlnet_mid_block.bias"], om,
                          B, cH, cW, ch, 0, ch, 1, 1, 0, os_=(ch * cH * cW, cW, 1, cH * cW))
            plan.static_out = {"down_block_res_samples": outs, "mid_block_res_sample": om}
            self._emit_kv_groups(plan, ctx, B, S_ctx)
            self._finish_plan(plan)
            return plan
        if ctrl:
            # ControlNet residuals. diffusers adds them to COPIES of the skip tensors after the down path, so the mid
            # block still sees the un-augmented activation: here the in-place adds are emitted after the mid block
            # (which has consumed the last skip by now), then the mid residual goes onto the mid-block output.
            ctrl_in = []
            for k, ((buf, _), (cc, sh_, sw_)) in enumerate(zip(skips, skip_dims)):
                r = torch.zeros((B, cc, sh_, sw_), dtype=dt, device=dev)
                ctrl_in.append(r)
                self._op_add_nchw(plan, f"controlnet.down_residual.{k}", r, buf, B, cc, sh_, sw_)
            rm = torch.zeros((B, ch, cH, cW), dtype=dt, device=dev)
            self._op_add_nchw(plan, "controlnet.mid_residual", rm, h, B, ch, cH, cW)
            plan.static_in["down_block_additional_residuals"] = ctrl_in
            plan.static_in["mid_block_additional_residual"] = rm
        # ---- up ------------------------------------------------------------------------------------
        rboc = self.boc[::-1]
        rheads, rdepth = self.heads[::-1], self.depth[::-1]
        for i, t in enumerate(self.up_types):
            co = rboc[i]
            for j in range(self.layers + 1):
                rn = f"up_blocks.{i}.resnets.{j}"
                sk, sc_ = skips.pop()
                hn = self._resnet(plan, rn, h, sk, ch, sc_, co, B, cH, cW, temb_all, tot, offs[rn])
                pool.put(h)
                pool.put(sk)
                ch = co
                if t == "CrossAttnUpBlock2D":
                    ha = self._transformer(plan, f"up_blocks.{i}.attentions.{j}", hn, co, B, cH, cW, rheads[i], rdepth[i], ctx, S_ctx)
                    pool.put(hn)
                    hn = ha
                h = hn
            if i < nlev - 1:
                un = f"up_blocks.{i}.upsamplers.0.conv"
                hu = pool.get(B * (2 * cH) * (2 * cW) * ch)
                self._op_conv(plan, un, h, None, P[un + ".weight"], P[un + ".bias"], hu, B, cH, cW, ch, 0, ch, 3, 1, 1, ups=True)
                pool.put(h)
                h = hu
                cH, cW = 2 * cH, 2 * cW
        assert not skips and (cH, cW) == (H, W)
        # ---- out -----------------------------------------------------------------------------------
        nout = pool.get(B * H * W * ch)
        self._op_gn(plan, "conv_norm_out", h, None, ch, ch, B, H * W, nout, self.eps, True, "conv_norm_out")
        pool.put(h)
        self._op_conv(plan, "conv_out", nout, None, P["conv_out.weight"], P["conv_out.bias"], out, B, H, W, ch, 0, self.out_ch, 3, 1, 1,
                      os_=(self.out_ch * H * W, W, 1, H * W), kind="conv_out")
        pool.put(nout)
        self._emit_kv_groups(plan, ctx, B, S_ctx)
        self._finish_plan(plan)
        return plan

    def _emit_kv_groups(self, plan, ctx, B, S_ctx):
        """Cross-attention K/V projections: they depend on the text context only, so every block's `to_k` / `to_v` pair runs at the
        top of the plan on the side lane -- blocks of equal width (same [M, 2C, ctx_dim] problem) share one grouped GEMM launch."""
        reqs, plan.kv_requests = plan.kv_requests, []
        new_ops = []
        saved, plan.ops = plan.ops, new_ops
        if plan.lora is not None:
            self._op_lora_merge(plan)  # first launch of the step: every LoRA'd linear's effective weight from the live tensors
        self._emit_kv_projections(plan, reqs, ctx, B * S_ctx, "attn2.to_kv")
        if plan.ip is not None:
            P = self.params
            embs = plan.static_in.get("image_embeds") or [None] * len(self.ip_proj)
            for i, ((pre, T_, dimg), emb, (tok, S_ip), rq) in enumerate(zip(self.ip_proj, embs, plan.ip["tokens"], plan.ip["requests"])):
                if not rq:
                    continue  # every scale of this adapter is 0
                if pre is None:  # projected tokens are a plan input
                    self._emit_kv_projections(plan, rq, tok, B * S_ip, f"attn2.to_kv_ip.{i}")
                    continue
                # ImageProjection: Linear(D_img -> T * ctx) per image, the row re-read as T tokens, LayerNorm over ctx
                rows = B * (S_ip // T_)
                lin = torch.empty(rows * T_ * self.ctx_dim, dtype=self.dtype, device=self.device)
                plan.keep.append(lin)
                self._op_gemm(plan, pre + ".image_embeds", emb, [P[pre + ".image_embeds.weight"]], P[pre + ".image_embeds.bias"], lin,
                              rows, T_ * self.ctx_dim, dimg, dimg, T_ * self.ctx_dim, lane=LANE_KV)
                self._op_ln(plan, pre + ".norm", lin, tok, B * S_ip, self.ctx_dim, pre + ".norm", lane=LANE_KV)
                self._emit_kv_projections(plan, rq, tok, B * S_ip, f"attn2.to_kv_ip.{i}")
        plan.ops = new_ops + saved

    def _emit_kv_projections(self, plan, reqs, ctx, M, label):
        if not reqs:
            return
        lib = self.lib
        K = self.ctx_dim
        by_c = defaultdict(list)
        for r in reqs:
            by_c[r[4]].append(r)
        for Cc, rs in by_c.items():
            groupable = K % 8 == 0 and all(w.stride() == (K, 1) for r in rs for w in (r[1], r[2]))
            if not groupable:
                for (name, wk, wv, kv, _) in rs:
                    self._op_gemm(plan, name, ctx, [wk, wv], None, kv, M, 2 * Cc, K, K, 2 * Cc, lane=LANE_KV)
                continue
            for g0 in range(0, len(rs), L.MAX_GEMM_GROUPS):
                grp = rs[g0:g0 + L.MAX_GEMM_GROUPS]
                p = L.GemmParams()
                p.dtype, p.M, p.N, p.K = self.dt, M, 2 * Cc, K
                p.ldx, p.ldw, p.ldo, p.ldr = K, K, 2 * Cc, 0
                p.n_wseg, p.rows_per_seg = 2, Cc
                p.geglu, p.act, p.res_before_act, p.alpha = 0, L.ACT_NONE, 0, 1.0
                p.rows_per_batch, p.ld_rowbias, p.in_act, p.variant, p.split_k = 0, 0, 0, 0, 0
                n = len(grp)
                wp = (C.c_void_p * (2 * n))(*[w.data_ptr() for r in grp for w in (r[1], r[2])])
                op = (C.c_void_p * n)(*[r[3].data_ptr() for r in grp])
                xp = (C.c_void_p * n)(*([ctx.data_ptr()] * n))
                plan.keep += [p, wp, op, xp]
                name = f"{label}[C={Cc},x{n}]"

                def launch(stream, p=p, wp=wp, op=op, xp=xp, n=n, name=name):
                    L.check(lib.sfast_hip_gemm_grouped(xp, wp, None, op, C.byref(p), n, stream), name)

                self._add(plan, "linear", name, 2.0 * M * 2 * Cc * K * n, (M * K + n * (2 * Cc * K + M * 2 * Cc)) * self.esize, launch,
                          lane=LANE_KV)

    def _fuse_gn_statistics(self, plan):
        """GroupNorm as ONE pass: every large GroupNorm whose input tensor(s) were written by MFMA GEMM / conv launches gets its
        statistics from those launches' epilogues (sfast_epilogue_ext / sfast_hip_group_norm_apply) instead of running its own
        statistics kernel. Runs after autotuning: the record layout follows the tile shape that was chosen for the producer."""
        mode = os.environ.get("SFAST_GN_FUSE", "1")
        if mode in ("0", "false", "off", ""):
            return
        import math
        from . import autotune
        lib = self.lib
        # one statistics unit for the whole plan: the widest channel count that divides every GroupNorm's channels-per-group and
        # every concat boundary (SD / SDXL: 320 / 32 = 10), so that the records of a tensor serve all of its consumers
        unit = 0
        for c in plan.gn_candidates:
            unit = math.gcd(unit, c["p"].C // c["p"].G)
            if c["concat"]:
                unit = math.gcd(unit, c["p"].C1)
        if unit < 8 or not hasattr(lib, "sfast_hip_group_norm_apply"):
            return
        for c in plan.gn_candidates:
            if c.get("in_reduce"):
                continue  # already computed by its producer's split-K reduce launch
            p = c["p"]
            cpg = p.C // p.G
            srcs = [c["xw"]] + ([c["x2w"]] if c["concat"] else [])
            if any(w is None for w in srcs) or cpg % unit or p.C1 % unit or p.C % 8 or p.C1 % 8:
                continue
            n_small = autotune.BATCH_REF if autotune.BATCH_INVARIANT else p.N  # (the library's gn_small_ok sees the same number)
            if p.HW * cpg * 2 <= 32 * 1024 and n_small * p.HW * p.C * 2 <= (4 << 20):
                continue  # the library runs these as one single-pass kernel already (norm.hip gn_small)
            lays = []
            for w in srcs:
                ext = L.EpilogueExt(0.0, unit, p.HW, w["ext"].flags)
                lay = L.GnStatsLayout()
                q = lib.sfast_hip_conv2d_stats_layout if w["conv"] else lib.sfast_hip_gemm_stats_layout
                if q(C.byref(w["p"]), C.byref(ext), C.byref(lay)) != 0:
                    lays = None
                    break
                lays.append(lay)
            if lays is None:
                continue
            if mode == "nosplit" and any(l.rb_rows < 64 for l in lays):
                continue  # A/B knob: only producers whose own epilogue writes the records (no split-K reduce in between)
            for w, lay in zip(srcs, lays):
                if w["stats"][0] is None:
                    w["ext"].gn_unit, w["ext"].gn_rows_per_sample = unit, p.HW
                    w["stats"][0] = torch.zeros(max(lay.nbytes() // 4, 2), dtype=torch.float32, device=self.device)
                    plan.keep.append(lay)
                elif w["ext"].gn_rows_per_sample != p.HW:
                    lays = None
                    break
            if lays is None:
                continue
            s2 = (srcs[1]["stats"][0].data_ptr(), lays[1]) if c["concat"] else (None, None)
            c["pre"][0] = (srcs[0]["stats"][0].data_ptr(), lays[0], s2[0], s2[1])
            plan.gn_fused += 1

    def _fuse_gn_into_reduce(self, plan):
        """Round 4: a GroupNorm(+SiLU) whose input was written by the split-K reduce launch of the op RIGHT BEFORE it in the plan (the
        16x16 / 8x8 levels: every resnet's norm2 behind conv1, the next block's norm behind conv2) is computed BY that launch
        (sfast_epilogue_ext.gn_out, csrc/igemm.hip splitk_reduce_gn_kernel) and leaves the plan: one launch of 4.5 - 6 us less per
        pair. Decided after tuning (the producer's K-split count is a tuning result); a producer that runs unsplit keeps its
        separate GroupNorm. SFAST_GN_IN_REDUCE=0 switches the pass off (A/B knob)."""
        if not GN_IN_REDUCE:
            return
        lib = self.lib
        index = {op.name: i for i, op in enumerate(plan.ops)}
        drop = set()
        for c in plan.gn_candidates:
            w, p = c["xw"], c["p"]
            if c["concat"] or w is None or c["name"] not in index or w["name"] not in index:
                continue
            if index[c["name"]] != index[w["name"]] + 1 or w["stats"][0] is not None or w["ext"].gn_unit or w["ext"].gn_out:
                continue   # something runs in between, or this producer already serves another consumer
            cpg = p.C // p.G
            if p.C1 != p.C or cpg % 4 or p.HW * cpg > 16384 or p.layout != L.NHWC:
                continue
            o5 = (C.c_int32 * 5)()
            wp = w["p"]
            if w["conv"]:
                Hin, Win = (2 * wp.H, 2 * wp.W) if wp.upsample2x else (wp.H, wp.W)
                Ho = (Hin + 2 * wp.pad_h + wp.pad_h_extra - wp.dil_h * (wp.KH - 1) - 1) // wp.stride_h + 1
                Wo = (Win + 2 * wp.pad_w + wp.pad_w_extra - wp.dil_w * (wp.KW - 1) - 1) // wp.stride_w + 1
                if Ho * Wo != p.HW or wp.B != p.N:
                    continue   # the GroupNorm's samples are not the conv's images (e.g. SVD's temporal norms over [B, C, F*H*W])
                if wp.Cout != p.C or lib.sfast_hip_conv2d_plan(C.byref(wp), wp.variant, wp.split_k, o5) != 0:
                    continue
            else:
                if wp.M != p.N * p.HW:
                    continue
                if wp.N != p.C or wp.geglu or wp.ldo != p.C or \
                        lib.sfast_hip_igemm_plan(wp.M, wp.N, wp.K, 0, wp.variant if wp.variant < 100 else 0, wp.split_k, o5) != 0:
                    continue
                if wp.rows_per_batch not in (0, p.HW):
                    continue
            if int(o5[2]) <= 1:
                continue
            ext = w["ext"]
            ext.gn_out, ext.gn_gamma, ext.gn_beta = c["yp"], c["gp"], c["bp"]
            ext.gn_groups, ext.gn_eps, ext.gn_act = p.G, c["eps"], (L.ACT_SILU if c["silu"] else L.ACT_NONE)
            ext.gn_rows_per_sample = p.HW
            c["in_reduce"] = True
            drop.add(c["name"])
        if drop:
            plan.ops = [op for op in plan.ops if op.name not in drop]
            plan.gn_in_reduce = len(drop)

    def _finish_plan(self, plan):
        # measured tile / pipe / split-K selection per distinct GEMM / conv problem (cuDNN-benchmark style)
        from . import autotune
        lib, dev = self.lib, self.device
        if self.host.tuning():
            autotune.tune_plan(plan, dev, "f16" if self.dtype == torch.float16 else "bf16")
            for op in plan.ops:
                if op.tune is not None:
                    p = op.tune[0]
                    q = lib.sfast_hip_gemm_workspace_bytes if isinstance(p, L.GemmParams) else lib.sfast_hip_conv2d_workspace_bytes
                    self._need_ws(plan, q(C.byref(p)), op.lane)
        self._settle_packed(plan)
        self._fuse_gn_into_reduce(plan)
        self._fuse_gn_statistics(plan)
        # one scratch buffer per lane, shared by all its operators; the split-K ticket counters of the GEMM / conv kernels live in a
        # block of their own behind the largest need (sfast_hip.h SFAST_EXT_WS_TICKETS), zeroed here once
        for holder in (plan.ws, plan.ws_side):
            if holder[1]:
                holder[1] = (holder[1] + 255) // 256 * 256 + L.WS_TICKET_BYTES
                holder[0] = torch.empty(holder[1], dtype=torch.uint8, device=dev)
                L.check(lib.sfast_hip_workspace_init(holder[0].data_ptr(), holder[1], self.host.stream_ptr(dev)), "sfast_hip_workspace_init")
        plan.side_stream = self.host.new_stream(dev)

    def _controlnet_cond_embedding(self, plan, h, B, H, W, c0):
        """ControlNetConditioningEmbedding on the NCHW conditioning image; its conv_out is fused with the add onto
        conv_in(sample) (residual operand of the conv epilogue). Returns the new hidden-state buffer."""
        pool, P = plan.pool, self.params
        dev, dt = self.device, self.dtype
        pre = "controlnet_cond_embedding"
        nblk = 0
        while f"{pre}.blocks.{nblk}.weight" in P:
            nblk += 1
        f = 1 << (nblk // 2)
        cin = P[f"{pre}.conv_in.weight"].shape[1]
        cond = torch.zeros((B, cin, H * f, W * f), dtype=dt, device=dev)
        plan.static_in["controlnet_cond"] = cond
        cH, cW = H * f, W * f
        c = P[f"{pre}.conv_in.weight"].shape[0]
        cur = pool.get(B * cH * cW * c)
        self._op_conv(plan, f"{pre}.conv_in", cond, None, P[f"{pre}.conv_in.weight"], P[f"{pre}.conv_in.bias"], cur, B, cH, cW, cin, 0, c,
                      3, 1, 1, xs=(cin * cH * cW, cW, 1, cH * cW), act=L.ACT_SILU, kind="conv_in")
        for i in range(nblk):
            w = P[f"{pre}.blocks.{i}.weight"]
            co = w.shape[0]
            stride = 2 if i % 2 else 1
            oH, oW = (cH + 2 - 3) // stride + 1, (cW + 2 - 3) // stride + 1
            nxt = pool.get(B * oH * oW * co)
            self._op_conv(plan, f"{pre}.blocks.{i}", cur, None, w, P[f"{pre}.blocks.{i}.bias"], nxt, B, cH, cW, c, 0, co, 3, stride, 1,
                          act=L.ACT_SILU)
            pool.put(cur)
            cur, c, cH, cW = nxt, co, oH, oW
        if (cH, cW) != (H, W):
            raise UnsupportedUNet("conditioning image size does not reduce to the latent size")
        out = pool.get(B * H * W * c0)
        self._op_conv(plan, f"{pre}.conv_out", cur, None, P[f"{pre}.conv_out.weight"], P[f"{pre}.conv_out.bias"], out, B, H, W, c, 0, c0,
                      3, 1, 1, z=h)
        pool.put(cur)
        pool.put(h)
        return out

    # ------------------------------------------------------------------------------------------
    def get_plan(self, B, H, W, S_ctx, ctrl=False, enc_mask=False, tcond=False, ip=None, self_mask=0):
        if ip is None and self.ip_proj and not self.ip_external:
            ip = ((1,) * len(self.ip_proj), self.ip_scales())
        key = (B, H, W, S_ctx, bool(ctrl), bool(enc_mask), bool(tcond), ip) + ((int(self_mask),) if self_mask else ())
        plan = self._plans.get(key)
        if plan is None:
            with self._lock:
                plan = self._plans.get(key)
                if plan is None:
                    kw = {}
                    if ctrl:
                        kw["ctrl"] = True
                    if enc_mask:
                        kw["enc_mask"] = True
                    if tcond:
                        kw["tcond"] = True
                    if ip is not None:
                        kw["ip"] = ip
                    if self_mask:
                        kw["self_mask"] = int(self_mask)
                    plan = self.build_plan(B, H, W, S_ctx, **kw)
                    self._plans[key] = plan
        return plan

    @staticmethod
    def encoder_attention_bias(mask, dtype):
        """diffusers UNet2DConditionModel.forward: a 2-D keep-mask [B, S] (bool / 0-1) becomes the additive bias
        (1 - mask) * -10000; a 3-D tensor [B, 1, S] is already a bias. Returns [B, S] in `dtype`."""
        if mask.ndim == 3 and mask.shape[1] == 1:
            return mask[:, 0].to(dtype)
        if mask.ndim != 2:
            raise UnsupportedUNet(f"encoder_attention_mask of shape {tuple(mask.shape)}")
        return ((1 - mask.to(dtype)) * -10000.0).to(dtype)

    def load_inputs(self, plan, sample, timestep, encoder_hidden_states, added_cond_kwargs=None,
                    down_block_additional_residuals=None, mid_block_additional_residual=None, encoder_attention_mask=None,
                    timestep_cond=None, class_labels=None, lora_scale=1.0, attention_mask=None):
        si = plan.static_in
        if "attention_bias" in si:
            if attention_mask is None or attention_mask.ndim != 2:
                raise ValueError("this plan takes an attention_mask [B, keys] (1 = keep, 0 = discard)")
            si["attention_bias"].copy_(((1 - attention_mask.to(self.dtype)) * -10000.0).to(self.dtype))
        if "lora_scale" in si:
            vals = self.lora_multipliers(lora_scale)
            if vals != plan.lora["last"]:  # cross_attention_kwargs["scale"] / set_adapters() moved: one small copy, no re-capture
                si["lora_scale"].copy_(torch.tensor(vals, dtype=torch.float32))
                plan.lora["last"] = vals
        if "timestep_cond" in si:
            if timestep_cond is None:
                raise ValueError("this plan takes a timestep_cond")
            si["timestep_cond"].copy_(timestep_cond)
        if "class_labels" in si:
            if class_labels is None:
                raise ValueError("class_labels should be provided when the UNet has a class embedding")  # diffusers' own error
            cl = si["class_labels"]
            cl.copy_(class_labels.reshape(cl.shape) if cl.ndim == 2 else class_labels.reshape(-1).to(torch.float32).expand(plan.B))
        if "encoder_attention_bias" in si:
            if encoder_attention_mask is None:
                raise ValueError("this plan takes an encoder_attention_mask")
            si["encoder_attention_bias"].copy_(self.encoder_attention_bias(encoder_attention_mask, self.dtype))
        for key in ("image_embeds", "ip_hidden_states"):
            if key in si:
                for dst, src in zip(si[key], self._ip_embeds(added_cond_kwargs, plan.B)):
                    if src.shape[1] != dst.shape[1]:
                        raise ValueError(f"this plan takes {dst.shape[1]} image(s) / token(s) per adapter, got {src.shape[1]}")
                    dst.copy_(src)
        si["sample"].copy_(sample)
        if torch.is_tensor(timestep):
            si["timestep"].copy_(timestep.reshape(-1).to(torch.float32).expand(plan.B), non_blocking=True)
        else:
            si["timestep"].fill_(float(timestep))
        si["encoder_hidden_states"].copy_(encoder_hidden_states)
        if self.add_type == "text_time":
            if not added_cond_kwargs or "text_embeds" not in added_cond_kwargs or "time_ids" not in added_cond_kwargs:
                raise ValueError("added_cond_kwargs with text_embeds and time_ids is required (addition_embed_type=text_time)")
            si["text_embeds"].copy_(added_cond_kwargs["text_embeds"])
            si["time_ids"].copy_(added_cond_kwargs["time_ids"].reshape(-1).to(torch.float32))
        if "down_block_additional_residuals" in si:
            want = si["down_block_additional_residuals"]
            if (down_block_additional_residuals is None or mid_block_additional_residual is None
                    or len(down_block_additional_residuals) != len(want)):
                raise ValueError(f"this plan takes {len(want)} ControlNet down residuals and one mid residual")
            for dst, src in zip(want, down_block_additional_residuals):
                dst.copy_(src)
            si["mid_block_additional_residual"].copy_(mid_block_additional_residual)

    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, encoder_attention_mask=None, timestep_cond=None, class_labels=None, lora_scale=1.0,
                attention_mask=None):
        """Eager (no graph) execution on the current stream; returns a fresh NCHW tensor."""
        B, _, H, W = sample.shape
        ctrl = down_block_additional_residuals is not None
        plan = self.get_plan(B, H, W, encoder_hidden_states.shape[1], ctrl, encoder_attention_mask is not None, timestep_cond is not None,
                             self.ip_signature(added_cond_kwargs), self_mask=attention_mask.shape[1] if attention_mask is not None else 0)
        self.sync_packed()
        self.load_inputs(plan, sample, timestep, encoder_hidden_states, added_cond_kwargs, down_block_additional_residuals,
                         mid_block_additional_residual, encoder_attention_mask, timestep_cond, class_labels, lora_scale, attention_mask)
        plan.run(self.host.stream_ptr(self.device))
        return plan.static_out.clone()


class ControlNetEngine(UNet2DEngine):
    """Executor for diffusers `ControlNetModel` parameter sets: the UNet's down path + mid block, the conditioning
    embedding and the 1x1 output convs, as one plan of the same C-ABI launches."""

    def __init__(self, config, params, device=None, dtype=None, _host=None):
        super().__init__(config, params, device=device, dtype=dtype, _host=_host)
        if not self.is_controlnet:
            raise UnsupportedUNet("parameters do not look like a ControlNetModel (no controlnet_mid_block)")
        if self.ip_proj:
            raise UnsupportedUNet("ControlNet with an IP-Adapter image projection")

    def load_inputs(self, plan, sample, timestep, encoder_hidden_states, controlnet_cond=None, added_cond_kwargs=None, **_):
        # SDXL ControlNets (addition_embed_type "text_time") take text_embeds / time_ids exactly as the SDXL UNet does
        super().load_inputs(plan, sample, timestep, encoder_hidden_states, added_cond_kwargs)
        if controlnet_cond is None:
            raise ValueError("controlnet_cond is required")
        plan.static_in["controlnet_cond"].copy_(controlnet_cond)

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0, guess_mode=False,
                added_cond_kwargs=None):
        """Eager (no graph) execution on the current stream; returns (down_block_res_samples, mid_block_res_sample)
        as fresh NCHW tensors, ready to be passed to UNet2DConditionModel.forward / UNet2DEngine.forward."""
        B, _, H, W = sample.shape
        plan = self.get_plan(B, H, W, encoder_hidden_states.shape[1])
        self.sync_packed()  # pipe-4 launches read packed copies: follow the live parameters' version counters
        self.load_inputs(plan, sample, timestep, encoder_hidden_states, controlnet_cond, added_cond_kwargs)
        plan.run(self.host.stream_ptr(self.device))
        return self.outputs(plan, conditioning_scale, guess_mode)

    @staticmethod
    def residual_scales(n_down, conditioning_scale=1.0, guess_mode=False):
        """Weights of the n_down skip residuals and the mid residual. guess_mode (diffusers ControlNetModel.forward, without
        global_pool_conditions): torch.logspace(-1, 0, n_down + 1) * conditioning_scale, i.e. 0.1 for the shallowest skip up to 1.0 for the
        mid block; otherwise conditioning_scale everywhere. Host floats -- the residuals are scaled as they are copied out of the plan."""
        if guess_mode:
            return [float(conditioning_scale) * 10.0 ** (-1.0 + i / n_down) for i in range(n_down + 1)]
        return [float(conditioning_scale)] * (n_down + 1)

    @classmethod
    def outputs(cls, plan, conditioning_scale=1.0, guess_mode=False):
        so = plan.static_out
        down = so["down_block_res_samples"]
        sc = cls.residual_scales(len(down), conditioning_scale, guess_mode)
        return ([t.clone() if s == 1.0 else t * s for t, s in zip(down, sc[:-1])],
                so["mid_block_res_sample"].clone() if sc[-1] == 1.0 else so["mid_block_res_sample"] * sc[-1])
