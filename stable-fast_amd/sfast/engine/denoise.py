"""One denoise iteration as ONE hipGraph: CFG batch-2 UNet forward + guidance combine + DDIM update.

The reference measures "it/s" as pipeline iterations where each iteration is a classifier-free
guidance batch-2 UNet forward plus the scheduler step
(/root/reference/examples/optimize_stable_diffusion_pipeline.py:127-151), with the UNet replayed from a
CUDA graph and the scheduler optionally traced (`trace_scheduler`,
/root/reference/src/sfast/compilers/diffusion_pipeline_compiler.py:103-107). Here the guidance combine
and the DDIM update are one HIP kernel recorded in the same graph as the UNet plan, and its output is
written straight into the UNet's static input buffer; the timestep and the DDIM coefficients of the step come
from device tables through a cursor the graph advances itself (`sfast_hip_schedule_advance`), so a step is one
graph launch.
"""
import ctypes as C

import torch

from ..hip import lib as L


def ddim_schedule(num_steps=50, num_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
    """SD1.5 DDIM constants: scaled-linear betas, 'leading' spacing with offset 1, eta = 0,
    set_alpha_to_one = False. Returns (timesteps, rows of [sqrt a_t, sqrt(1-a_t), sqrt a_prev, sqrt(1-a_prev)])."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float64) ** 2
    acp = torch.cumprod(1.0 - betas, dim=0)
    ratio = num_train // num_steps
    ts = (torch.arange(0, num_steps) * ratio).flip(0) + steps_offset
    rows = []
    for t in ts.tolist():
        a_t = acp[t]
        prev = t - ratio
        a_p = acp[prev] if prev >= 0 else acp[0]
        rows.append([float(a_t.sqrt()), float((1 - a_t).sqrt()), float(a_p.sqrt()), float((1 - a_p).sqrt())])
    return ts.tolist(), rows


class DenoiseLoop:
    """Independent per-GPU denoise loop over `images` latents with classifier-free guidance.

    One iteration = ONE graph launch and nothing else: the graph's first node (`sfast_hip_schedule_advance`) moves the current
    row of the timestep / coefficient tables into the plan's static inputs and advances a device-side cursor, so the host issues
    no per-step copies.

    `hoist_text_kv` (default False): the cross-attention K/V projections depend on the text context only -- constant over the 50
    steps. With the flag they run ONCE, in `set_inputs`, and are left out of the step graph (a pipeline-level loop-invariant
    hoist that `compile()` cannot do: there the context may change per call). The default keeps them IN the step graph, so a
    step is literally one `unet.forward` + guidance + scheduler update -- the unit the reference's it/s counts and what
    `bench.py`'s headline measures; the hoisted form is reported beside it. With the hoist, in-place updates of `to_k` / `to_v`
    weights (the live-weight / LoRA contract) become visible at the next `set_inputs()` or `refresh_text_kv()`, not at the next
    step -- every other weight stays live: read from the parameter's storage by the launch itself, or (packed-weight pipe, the
    default) re-packed by `step()` when the parameter's autograd version counter moved (`UNet2DEngine.sync_packed`; writes that
    bypass the counter, e.g. `p.data.copy_`, need `engine.sync_packed(force=True)`).

    The loop owns a PRIVATE plan (`engine.build_plan`, not the engine's cached plan of the same signature): its static inputs
    and K/V buffers cannot be overwritten by `engine.forward()` / `compile()` calls that share the engine between two steps."""

    def __init__(self, engine, images=1, height=64, width=64, ctx_len=77, guidance=7.5, num_steps=50, use_graph=True,
                 hoist_text_kv=False):
        self.engine = engine
        self.images = images
        self._shape = (height, width, ctx_len)
        self._gen = getattr(engine, "generation", 0)
        self.guidance = float(guidance)
        self.lib = self._library(engine)
        dev, dt = engine.device, engine.dtype
        self.plan = engine.build_plan(2 * images, height, width, ctx_len)
        ts, rows = ddim_schedule(num_steps)
        self.num_steps = num_steps
        self.ts_table = torch.tensor(ts, dtype=torch.float32, device=dev).reshape(-1, 1).expand(-1, 2 * images).contiguous()
        self.coef_table = torch.tensor(rows, dtype=torch.float32, device=dev)
        self.coef = torch.zeros(4, dtype=torch.float32, device=dev)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=dev)   # device-side step index, advanced by the graph itself
        self._next = 0                                                # host mirror of the cursor
        self.latents = torch.zeros((images, engine.in_ch, height, width), dtype=dt, device=dev)
        self.use_graph = use_graph
        self.graph = None
        from .unet2d import LANE_KV
        self._kv_lane = LANE_KV
        self.hoist_text_kv = bool(hoist_text_kv)
        if self.hoist_text_kv:
            self._step_ops = [op for op in self.plan.ops if op.lane != LANE_KV]   # what a step launches
            self._ctx_ops = [op for op in self.plan.ops if op.lane == LANE_KV]    # what a new text context launches, once
        else:
            self._step_ops = list(self.plan.ops)                                  # the whole forward, every step
            self._ctx_ops = []

    @staticmethod
    def _library(engine):
        return L.init_device(engine.device)

    def _stream_ptr(self):
        return torch.cuda.current_stream(self.engine.device).cuda_stream

    def set_inputs(self, latents, ehs_uncond_cond, lora_scale=1.0):
        """latents [images,4,H,W]; ehs_uncond_cond [2*images, ctx, dim] ordered [uncond..., cond...]. Runs the text-side launches
        (every cross-attention block's K/V projection) for this context. `lora_scale`: diffusers' cross_attention_kwargs["scale"]
        for a UNet with un-fused LoRA factors (an entry of the merge launch's scale table; ignored otherwise)."""
        self.latents.copy_(latents)
        si = self.plan.static_in
        self.engine.sync_packed()
        if "lora_scale" in si:
            vals = self.engine.lora_multipliers(lora_scale)
            if vals != self.plan.lora["last"]:
                si["lora_scale"].copy_(torch.tensor(vals, dtype=torch.float32))
                self.plan.lora["last"] = vals
        si["sample"][: self.images].copy_(latents)
        si["sample"][self.images:].copy_(latents)
        si["encoder_hidden_states"].copy_(ehs_uncond_cond)
        self.refresh_text_kv()

    def refresh_text_kv(self):
        """hoist_text_kv only: re-run the text-side launches (every cross-attention block's K/V projection) on the current stream --
        after a new context (`set_inputs` calls this) or after an in-place update of to_k / to_v weights. A no-op otherwise."""
        sp = self._stream_ptr()
        if self._ctx_ops:
            self.engine.sync_packed()
        if self._ctx_ops and self.plan.lora is not None:
            for op in self.plan.ops:  # the hoisted K/V projections read merged LoRA weights: rebuild those first
                if op.name.startswith("lora.merge"):
                    op.launch(sp)
        for op in self._ctx_ops:
            op.launch(sp)

    def _follow_engine(self):
        """`engine.refresh_parameters()` re-bound the engine to re-assigned parameters since this loop built its private plan: rebuild the
        plan against the new storage (static inputs and the schedule position carried over), re-run the text-side launches and re-capture
        the graph when there was one -- the old graph retires through its OwnedGraph handle."""
        gen = getattr(self.engine, "generation", 0)
        if gen == self._gen:
            return
        self._gen = gen
        old_in = {k: v.clone() for k, v in self.plan.static_in.items() if torch.is_tensor(v)}
        h, w, ctx = self._shape
        self.plan = self.engine.build_plan(2 * self.images, h, w, ctx)
        if self.hoist_text_kv:
            self._step_ops = [op for op in self.plan.ops if op.lane != self._kv_lane]
            self._ctx_ops = [op for op in self.plan.ops if op.lane == self._kv_lane]
        else:
            self._step_ops, self._ctx_ops = list(self.plan.ops), []
        for k, v in old_in.items():
            if k in self.plan.static_in and self.plan.static_in[k].shape == v.shape:
                self.plan.static_in[k].copy_(v)
        self.refresh_text_kv()
        if self.graph is not None:
            nxt = self._next
            self.capture(warmups=1)
            self.set_step(nxt)

    def set_step(self, i):
        """Next iteration to run (0-based, modulo the schedule length)."""
        self._next = i % self.num_steps
        self.cursor.fill_(self._next)

    def _head(self, stream):
        rc = self.lib.sfast_hip_schedule_advance(self.cursor.data_ptr(), self.ts_table.data_ptr(), self.ts_table.shape[1],
                                                self.plan.static_in["timestep"].data_ptr(), self.coef_table.data_ptr(), 4,
                                                self.coef.data_ptr(), self.num_steps, stream)
        L.check(rc, "sfast_hip_schedule_advance")

    def _tail(self, stream):
        rc = self.lib.sfast_hip_cfg_ddim_step(self.plan.static_out.data_ptr(), self.latents.data_ptr(), self.latents.data_ptr(),
                                             self.plan.static_in["sample"].data_ptr(), self.coef.data_ptr(),
                                             C.c_float(self.guidance), self.latents.numel(), self.engine.dt, stream)
        L.check(rc, "sfast_hip_cfg_ddim_step")

    def _launch_all(self, stream):
        self._head(stream)
        for op in self._step_ops:
            op.launch(stream)
        self._tail(stream)

    def capture(self, warmups=3):
        dev = self.engine.device
        keep_lat = self.latents.clone()
        keep_in = self.plan.static_in["sample"].clone()
        torch.cuda.synchronize(dev)
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            for _ in range(warmups):
                self._launch_all(side.cuda_stream)
        torch.cuda.synchronize(dev)
        if self.use_graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    self._launch_all(torch.cuda.current_stream(dev).cuda_stream)
            from .unet2d import OwnedGraph, _trim_retired
            # a re-capture: dropping the old handle RETIRES its graph (OwnedGraph), never destroys it in the same breath as its last replay;
            # the retired queue is trimmed here as well (a process that only re-captures loops never reached capture_plan_graph)
            self.graph = None
            _trim_retired()
            self.graph = OwnedGraph(g, dev)
            torch.cuda.synchronize(dev)
        self.latents.copy_(keep_lat)
        self.plan.static_in["sample"].copy_(keep_in)
        self.set_step(0)

    def step(self, i):
        """Run denoise iteration i (0-based) on the current stream: one graph launch. The host touches the device cursor only when
        `i` is not the iteration the cursor already points at (a restart, a skipped step)."""
        self._follow_engine()
        idx = i % self.num_steps
        if idx != self._next:
            self.cursor.fill_(idx)
        self._next = (idx + 1) % self.num_steps
        # pipe-4 launches read PACKED copies of their weights: an in-place parameter update (the live-LoRA switch) between two steps is
        # picked up here through the parameters' version counters (a host-side compare per packed weight when nothing changed)
        self.engine.sync_packed()
        if self.graph is not None:
            self.graph.replay()
        else:
            self._launch_all(self._stream_ptr())
