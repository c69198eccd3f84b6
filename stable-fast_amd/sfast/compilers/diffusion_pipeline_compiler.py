"""Drop-in `compile()` surface of stable-fast for MI355X.

Same names, arguments and in-place semantics as the reference module
/root/reference/src/sfast/compilers/diffusion_pipeline_compiler.py
  CompilationConfig.Default  (:22-78)   the same 11 fields
  compile(m, config)         (:81-124)  mutates and returns the pipeline (duck-typed on .unet, .vae, ...)
  compile_unet(m, config)    (:127-151)
  compile_vae(m, config)     (:154-190)  the DECODER runs on the native engine too (sfast.engine.VaeDecoderEngine)

What changed is what happens behind `unet.forward`: instead of TorchScript trace + pattern passes +
CUDA graph, the UNet is handed to `sfast.engine.UNet2DEngine`, which runs the forward as a static
plan of hand-written gfx950 kernels (libsfast_hip.so) captured in a hipGraph. The per-signature
cache keeps the reference's "dynamic shape by recapture" behaviour (cuda/graphs.py:31,
jit/trace_helper.py:44): a new (batch, latent size, context length) builds and captures a new plan.

Config flags: `enable_jit` gates the native engine (the reference does "most optimizations" under
it), `enable_cuda_graph` gates hipGraph capture, `memory_format` converts 4-D parameters exactly as
before. `enable_xformers`, `enable_triton`, `enable_cnn_optimization`, `enable_fused_linear_geglu`
and `prefer_lowp_gemm` select individual fusions in the reference; the native engine always runs
the fused HIP kernels (fp32 accumulate -- CDNA4 MFMA has no fp16-accumulate mode, so the
"quality degradation" caveat of the last two does not apply), so they are accepted and recorded
but do not change the executed kernels.
"""
import logging
import threading
from dataclasses import dataclass

import torch

from ..cuda.graphs import get_per_device_graph_execution_env, make_dynamic_graphed_callable
from ..utils import gpu_device
from ..utils.memory_format import apply_memory_format

logger = logging.getLogger()


class CompilationConfig:

    @dataclass
    class Default:
        """Default compilation config (field meanings as in the reference, see module docstring)."""
        memory_format: torch.memory_format = (
            torch.channels_last if gpu_device.device_has_tensor_core() else torch.contiguous_format)
        enable_jit: bool = True
        enable_jit_freeze: bool = True
        preserve_parameters: bool = True
        enable_cnn_optimization: bool = gpu_device.device_has_tensor_core()
        enable_fused_linear_geglu: bool = gpu_device.device_has_capability(8, 0)
        prefer_lowp_gemm: bool = True
        enable_xformers: bool = False
        enable_cuda_graph: bool = False
        enable_triton: bool = False
        trace_scheduler: bool = False


class UNet2DConditionOutput:
    """Stand-in for diffusers' output dataclass when diffusers is not importable."""

    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]

    def __iter__(self):
        return iter((self.sample,))


def _make_output(sample):
    try:
        from diffusers.models.unets.unet_2d_condition import UNet2DConditionOutput as _Out  # type: ignore
        return _Out(sample=sample)
    except Exception:
        return UNet2DConditionOutput(sample)


def _device_of(m):
    if hasattr(m, "device"):
        d = m.device
        return d if isinstance(d, torch.device) else torch.device(d)
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


_FALLBACK = object()  # plan-cache entry of a signature that must run the module's original forward


def _drop_stale_plans(fwd):
    """`engine.refresh_parameters(module)` re-binds the engine to RE-ASSIGNED parameters (new storage) and bumps `engine.generation`.
    A compiled forward's (plan, graph) entries built before it would go on replaying the old storage and the old packed copies,
    silently: they are dropped here, at the next call -- the graphs retire through their OwnedGraph handles, the next call
    of each signature rebuilds its plan against the new parameters."""
    gen = getattr(fwd.engine, "generation", 0)
    if fwd.__dict__.get("_gen", gen) != gen:
        with fwd._lock:
            if fwd._cached:
                logger.info("sfast: engine parameters were re-bound (generation %d): dropping %d cached plan(s)", gen, len(fwd._cached))
            fwd._cached.clear()
    fwd._gen = gen


class _NativeUNetForward:
    """Replacement for `unet.forward`: per-signature plan cache + hipGraph replay."""

    def __init__(self, module, engine, orig_forward, enable_graph, warmups=3):
        self.module = module
        self.engine = engine
        self.orig_forward = orig_forward
        self.enable_graph = enable_graph
        self.warmups = warmups
        self._cached = {}
        self._lock = threading.Lock()
        self._warned = False
        self.__self__ = module
        self.__name__ = "forward"

    def _projected_image_tokens(self, image_embeds):
        """IP-Adapters whose image projection is not a plain ImageProjection (IP-Adapter Plus' resampler, ...): the module's own
        `encoder_hid_proj` runs eagerly, ONCE per distinct image_embeds -- pipelines pass the same tensors at every denoise step -- and
        the plan takes the projected tokens. The cache key is (storage, version counter, shape) of every tensor."""
        ie = list(image_embeds) if isinstance(image_embeds, (list, tuple)) else [image_embeds]
        key = tuple((t.data_ptr(), t._version, tuple(t.shape), t.dtype) for t in ie)
        hit = getattr(self, "_ip_tokens", None)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                out = self.module.encoder_hid_proj(image_embeds if isinstance(image_embeds, (list, tuple)) else ie[0])
            out = list(out) if isinstance(out, (list, tuple)) else [out]
            hit = self._ip_tokens = (key, [t.to(self.engine.dtype).contiguous() for t in out])
        return hit[1]

    def _fallback(self, why, *args, **kwargs):
        if not self._warned:
            logger.warning("sfast: UNet call not handled by the native engine (%s); running the original forward", why)
            self._warned = True
        return self.orig_forward(*args, **kwargs)

    def _prepare(self, key, sample, timestep, ehs, added, down_res, mid_res, emask=None, tcond=None, clabels=None, lora_scale=1.0, amask=None):
        eng = self.engine
        B, H, W, S, ctrl, has_mask, has_tcond, ip, amask_len = key
        # everything below (kernel-attribute setup, autotune launches and their event timing, warm-up, capture) must run with
        # the MODEL's device current, whatever device the caller has selected (reference: graphs.py wraps capture and replay
        # in torch.cuda.device(execution_env.device))
        with torch.cuda.device(eng.device):
            plan = eng.get_plan(B, H, W, S, ctrl, has_mask, has_tcond, ip, self_mask=amask_len)
        env = get_per_device_graph_execution_env(eng.device)
        graph = None
        # warm-up: runs the whole plan eagerly (also validates every launch before capture)
        torch.cuda.synchronize(eng.device)
        with torch.cuda.device(eng.device), torch.cuda.stream(torch.cuda.Stream(device=eng.device)):
            eng.load_inputs(plan, sample, timestep, ehs, added, down_res, mid_res, emask, tcond, clabels, lora_scale, amask)
            for _ in range(self.warmups if self.enable_graph else 1):
                plan.run(torch.cuda.current_stream(eng.device).cuda_stream)
        torch.cuda.synchronize(eng.device)
        if self.enable_graph:
            from ..engine import capture_plan_graph
            with env.lock:
                with torch.cuda.device(eng.device):
                    graph, _ = capture_plan_graph(plan, env.stream, pool=env.mempool)
                torch.cuda.synchronize(eng.device)
        return plan, graph, env

    def __call__(self, sample, timestep, encoder_hidden_states=None, class_labels=None, timestep_cond=None,
                 attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                 down_block_additional_residuals=None, mid_block_additional_residual=None,
                 down_intrablock_additional_residuals=None, encoder_attention_mask=None, return_dict=True):
        extra = dict(down_intrablock_additional_residuals=down_intrablock_additional_residuals)
        bad = [k for k, v in extra.items() if v is not None]
        eng = self.engine
        _drop_stale_plans(self)
        # attention_mask (a keep-mask over the SELF-attention keys) is an input of the native plan: an additive key bias of every attn1
        # launch, as diffusers builds it; the reference hands attn_bias straight to the kernel (libs/xformers/xformers_attention.py:26-48).
        # A mask whose length differs from some layer's token count has no plan (diffusers itself fails on it): that signature keeps the
        # module's own forward, which raises diffusers' own error.
        amask = attention_mask
        if amask is not None and not (torch.is_tensor(amask) and amask.device.type == "cuda" and amask.ndim == 2
                                      and torch.is_tensor(sample) and amask.shape[0] == sample.shape[0]):
            bad.append("attention_mask (need a [B, keys] keep-mask on the GPU)")
        # encoder_attention_mask (text padding) is an input of the native plan: an additive key bias of every cross-attention
        # launch (reference passes attn_bias through, libs/xformers/xformers_attention.py:30-47)
        emask = encoder_attention_mask
        if emask is not None and not (torch.is_tensor(emask) and emask.device.type == "cuda" and emask.shape[0] == sample.shape[0]
                                      and ((emask.ndim == 2) or (emask.ndim == 3 and emask.shape[1] == 1))):
            bad.append("encoder_attention_mask (need [B, S] or [B, 1, S] on the GPU)")
        # cross_attention_kwargs: diffusers pops "scale" and applies it to LoRA layers only: un-fused LoRA factors the engine took over
        # (UNet2DEngine.lora) are scaled by it through the merge launch's scale table, without them it has nothing to act on; any
        # other key (ip_adapter_masks, gligen, ...) selects an attention processor the plan does not implement
        if cross_attention_kwargs and set(cross_attention_kwargs) - {"scale"}:
            bad.append("cross_attention_kwargs " + str(sorted(set(cross_attention_kwargs) - {"scale"})))
        # timestep_cond (LCM-distilled UNets: the guidance-scale embedding w) and class_labels (class_embed_type "timestep" /
        # "projection") are inputs of the native plan (examples/optimize_lcm_pipeline.py in the reference runs on the compiled UNet)
        tcond, tdim, ctype = timestep_cond, getattr(eng, "tcond_dim", None), getattr(eng, "class_type", None)
        if tcond is not None and not (torch.is_tensor(tcond) and tcond.device.type == "cuda" and tcond.ndim == 2 and tdim is not None
                                      and tuple(tcond.shape) == (sample.shape[0], int(tdim))):
            bad.append("timestep_cond (need [B, time_cond_proj_dim] on the GPU)")
        # diffusers ignores class_labels when the UNet has no class embedding (get_class_embed returns None): so does the plan
        clabels = class_labels if ctype is not None else None
        if clabels is not None and not (torch.is_tensor(clabels) and clabels.device.type == "cuda"):
            bad.append("class_labels (need a tensor on the GPU)")
        if clabels is None and ctype is not None:
            bad.append("class_labels missing")  # let the original forward raise diffusers' own error
        # ControlNet residuals (reference keeps ControlNet pipelines on the compiled UNet, :89-90): taken natively when
        # both kinds are present as tensors of the engine's dtype on its device
        ctrl = down_block_additional_residuals is not None or mid_block_additional_residual is not None
        if ctrl:
            ok = (down_block_additional_residuals is not None and mid_block_additional_residual is not None
                  and torch.is_tensor(mid_block_additional_residual)
                  and all(torch.is_tensor(r) and r.device.type == "cuda" and r.dtype == eng.dtype
                          for r in list(down_block_additional_residuals) + [mid_block_additional_residual]))
            if not ok:
                bad.append("controlnet residuals (need both kinds, engine dtype, on the GPU)")
        # IP-Adapter (load_ip_adapter: encoder_hid_dim_type "ip_image_proj"): the image embeddings of added_cond_kwargs are plan inputs;
        # the number of images per adapter and the live processor scales are part of the plan signature
        ip = None
        if getattr(eng, "ip_proj", None) and not bad:
            try:
                if eng.ip_external and added_cond_kwargs and added_cond_kwargs.get("image_embeds") is not None:
                    added_cond_kwargs = dict(added_cond_kwargs, ip_hidden_states=self._projected_image_tokens(added_cond_kwargs["image_embeds"]))
                ip = eng.ip_signature(added_cond_kwargs)
                if not all(t.device.type == "cuda" for t in eng._ip_embeds(added_cond_kwargs, sample.shape[0])):
                    bad.append("image_embeds (need tensors on the GPU)")
            except (NotImplementedError, ValueError, AttributeError) as e:
                bad.append(f"image_embeds ({e})")  # the original forward raises diffusers' own error for a missing input
        if (bad or encoder_hidden_states is None or not torch.is_tensor(sample) or sample.device.type != "cuda"
                or sample.dtype != eng.dtype or sample.ndim != 4):
            # hand the original forward exactly what the caller passed (only the arguments that were given, so older
            # diffusers signatures without the newer keywords keep working)
            given = dict(class_labels=class_labels, timestep_cond=timestep_cond, attention_mask=attention_mask,
                         cross_attention_kwargs=cross_attention_kwargs, added_cond_kwargs=added_cond_kwargs,
                         down_block_additional_residuals=down_block_additional_residuals,
                         mid_block_additional_residual=mid_block_additional_residual,
                         down_intrablock_additional_residuals=down_intrablock_additional_residuals,
                         encoder_attention_mask=encoder_attention_mask)
            return self._fallback(", ".join(bad) or "input device/dtype", sample, timestep,
                                  encoder_hidden_states=encoder_hidden_states, return_dict=return_dict,
                                  **{k: v for k, v in given.items() if v is not None})
        B, _, H, W = sample.shape
        lora_scale = float((cross_attention_kwargs or {}).get("scale", 1.0))
        key = (B, H, W, encoder_hidden_states.shape[1], ctrl, emask is not None, tcond is not None, ip, int(amask.shape[1]) if amask is not None else 0)
        entry = self._cached.get(key)
        if entry is None:
            with self._lock:
                entry = self._cached.get(key)
                if entry is None:
                    logger.info("sfast: building native UNet plan for %s (graph=%s)", key, self.enable_graph)
                    try:
                        entry = self._prepare(key, sample, timestep, encoder_hidden_states, added_cond_kwargs,
                                              down_block_additional_residuals, mid_block_additional_residual, emask, tcond, clabels, lora_scale, amask)
                    except (NotImplementedError, KeyError) as e:
                        # this signature is outside the plan's coverage (e.g. a latent size the levels do not divide, a
                        # parameter the planner expected but a wrapper renamed): keep the module's own forward for it
                        logger.warning("sfast: no native plan for UNet call %s (%s: %s); this signature runs the original forward",
                                       key, type(e).__name__, e)
                        entry = _FALLBACK
                    self._cached[key] = entry
        if entry is _FALLBACK:
            given = dict(added_cond_kwargs=added_cond_kwargs, down_block_additional_residuals=down_block_additional_residuals,
                         mid_block_additional_residual=mid_block_additional_residual, encoder_attention_mask=encoder_attention_mask,
                         timestep_cond=timestep_cond, class_labels=class_labels, cross_attention_kwargs=cross_attention_kwargs,
                         attention_mask=attention_mask)
            return self.orig_forward(sample, timestep, encoder_hidden_states=encoder_hidden_states, return_dict=return_dict,
                                     **{k: v for k, v in given.items() if v is not None})
        plan, graph, env = entry
        with env.lock, torch.cuda.device(eng.device):
            eng.sync_packed()  # packed weight copies follow the live parameters (version counters); a no-op when nothing changed
            eng.load_inputs(plan, sample, timestep, encoder_hidden_states, added_cond_kwargs,
                            down_block_additional_residuals, mid_block_additional_residual, emask, tcond, clabels, lora_scale, amask)
            if graph is not None:
                graph.replay()
            else:
                plan.run(torch.cuda.current_stream(eng.device).cuda_stream)
            out = plan.static_out.clone()
        if not return_dict:
            return (out,)
        return _make_output(out)


class ControlNetOutput:
    """Stand-in for diffusers' ControlNetOutput when diffusers is not importable."""

    def __init__(self, down_block_res_samples, mid_block_res_sample):
        self.down_block_res_samples = down_block_res_samples
        self.mid_block_res_sample = mid_block_res_sample

    def __getitem__(self, i):
        return (self.down_block_res_samples, self.mid_block_res_sample)[i]

    def __iter__(self):
        return iter((self.down_block_res_samples, self.mid_block_res_sample))


class _NativeControlNetForward:
    """Replacement for `controlnet.forward` (diffusers ControlNetModel): per-signature plan cache + hipGraph replay."""

    def __init__(self, module, engine, orig_forward, enable_graph):
        self.module, self.engine, self.orig_forward, self.enable_graph = module, engine, orig_forward, enable_graph
        self._cached = {}
        self._lock = threading.Lock()
        self._warned = False
        self.__self__ = module
        self.__name__ = "forward"

    def _prepare(self, key, sample, timestep, ehs, cond, added=None):
        eng = self.engine
        B, H, W, S = key
        with torch.cuda.device(eng.device):
            plan = eng.get_plan(B, H, W, S)
        graph = None
        torch.cuda.synchronize(eng.device)
        with torch.cuda.device(eng.device), torch.cuda.stream(torch.cuda.Stream(device=eng.device)):
            eng.load_inputs(plan, sample, timestep, ehs, cond, added)
            plan.run(torch.cuda.current_stream(eng.device).cuda_stream)
        torch.cuda.synchronize(eng.device)
        env = get_per_device_graph_execution_env(eng.device)
        if self.enable_graph:
            from ..engine import capture_plan_graph
            with env.lock:
                with torch.cuda.device(eng.device):
                    graph, _ = capture_plan_graph(plan, env.stream, pool=env.mempool)
                torch.cuda.synchronize(eng.device)
        return plan, graph, env

    def __call__(self, sample, timestep, encoder_hidden_states=None, controlnet_cond=None, conditioning_scale=1.0,
                 class_labels=None, timestep_cond=None, attention_mask=None, added_cond_kwargs=None, cross_attention_kwargs=None,
                 guess_mode=False, return_dict=True):
        eng = self.engine
        _drop_stale_plans(self)
        bad = [k for k, v in dict(class_labels=class_labels, timestep_cond=timestep_cond, attention_mask=attention_mask).items() if v is not None]
        # added_cond_kwargs: SDXL ControlNets (addition_embed_type "text_time") take text_embeds + time_ids natively; any other key, or
        # extra conditioning handed to a ControlNet without the addition embedding (diffusers ignores it there), keeps diffusers' forward
        if added_cond_kwargs is not None:
            if eng.add_type != "text_time" or set(added_cond_kwargs) - {"text_embeds", "time_ids"} or not all(
                    torch.is_tensor(v) and v.device.type == "cuda" for v in added_cond_kwargs.values()):
                bad.append("added_cond_kwargs")
        elif eng.add_type == "text_time":
            bad.append("added_cond_kwargs missing")  # the original forward raises diffusers' own error
        if cross_attention_kwargs and set(cross_attention_kwargs) - {"scale"}:   # "scale" acts on LoRA layers only; the engine holds none
            bad.append("cross_attention_kwargs " + str(sorted(set(cross_attention_kwargs) - {"scale"})))
        # guess_mode is native (per-residual logspace weights, ControlNetEngine.residual_scales) unless the ControlNet pools its
        # conditions globally (global_pool_conditions: the engine's config whitelist already refuses such a module)
        if (bad or encoder_hidden_states is None or controlnet_cond is None or not torch.is_tensor(sample)
                or sample.device.type != "cuda" or sample.dtype != eng.dtype or sample.ndim != 4
                or not isinstance(conditioning_scale, (int, float))):
            if not self._warned:
                logger.warning("sfast: ControlNet call not handled by the native engine (%s); running the original forward",
                               ", ".join(bad) or "inputs")
                self._warned = True
            return self.orig_forward(sample, timestep, encoder_hidden_states=encoder_hidden_states, controlnet_cond=controlnet_cond,
                                     conditioning_scale=conditioning_scale, class_labels=class_labels, timestep_cond=timestep_cond,
                                     attention_mask=attention_mask, added_cond_kwargs=added_cond_kwargs,
                                     cross_attention_kwargs=cross_attention_kwargs, guess_mode=guess_mode, return_dict=return_dict)
        B, _, H, W = sample.shape
        key = (B, H, W, encoder_hidden_states.shape[1])
        entry = self._cached.get(key)
        if entry is None:
            with self._lock:
                entry = self._cached.get(key)
                if entry is None:
                    try:
                        entry = self._prepare(key, sample, timestep, encoder_hidden_states, controlnet_cond, added_cond_kwargs)
                    except (NotImplementedError, KeyError) as e:
                        logger.warning("sfast: no native plan for ControlNet call %s (%s: %s); this signature runs the original forward",
                                       key, type(e).__name__, e)
                        entry = _FALLBACK
                    self._cached[key] = entry
        if entry is _FALLBACK:
            return self.orig_forward(sample, timestep, encoder_hidden_states=encoder_hidden_states, controlnet_cond=controlnet_cond,
                                     conditioning_scale=conditioning_scale, guess_mode=guess_mode, return_dict=return_dict,
                                     **({"added_cond_kwargs": added_cond_kwargs} if added_cond_kwargs is not None else {}))
        plan, graph, env = entry
        with env.lock, torch.cuda.device(eng.device):
            eng.sync_packed()  # packed weight copies follow the live parameters (version counters); a no-op when nothing changed
            eng.load_inputs(plan, sample, timestep, encoder_hidden_states, controlnet_cond.to(eng.dtype), added_cond_kwargs)
            if graph is not None:
                graph.replay()
            else:
                plan.run(torch.cuda.current_stream(eng.device).cuda_stream)
            down, mid = eng.outputs(plan, float(conditioning_scale), bool(guess_mode))
        if not return_dict:
            return down, mid
        return ControlNetOutput(down, mid)


class _NativeSVDForward:
    """Replacement for `UNetSpatioTemporalConditionModel.forward` (Stable Video Diffusion): per-shape plan cache + hipGraph replay
    (reference: examples/optimize_stable_video_diffusion_pipeline.py hands the SVD pipeline to the same compile())."""

    def __init__(self, module, engine, orig_forward, enable_graph):
        self.module, self.engine, self.orig_forward, self.enable_graph = module, engine, orig_forward, enable_graph
        self._cached = {}
        self._lock = threading.Lock()
        self._warned = False
        self.__self__ = module
        self.__name__ = "forward"

    def _prepare(self, key, sample, timestep, ehs, tids):
        eng = self.engine
        with torch.cuda.device(eng.device):
            plan = eng.get_plan(*key)
        graph = None
        torch.cuda.synchronize(eng.device)
        with torch.cuda.device(eng.device), torch.cuda.stream(torch.cuda.Stream(device=eng.device)):
            eng.load_inputs(plan, sample, timestep, ehs, tids)
            plan.run(torch.cuda.current_stream(eng.device).cuda_stream)
        torch.cuda.synchronize(eng.device)
        env = get_per_device_graph_execution_env(eng.device)
        if self.enable_graph:
            from ..engine import capture_plan_graph
            with env.lock:
                with torch.cuda.device(eng.device):
                    graph, _ = capture_plan_graph(plan, env.stream, pool=env.mempool)
                torch.cuda.synchronize(eng.device)
        return plan, graph, env

    def __call__(self, sample, timestep, encoder_hidden_states, added_time_ids, return_dict=True):
        eng = self.engine
        _drop_stale_plans(self)
        ok = (torch.is_tensor(sample) and sample.ndim == 5 and sample.device.type == "cuda" and sample.dtype == eng.dtype
              and torch.is_tensor(encoder_hidden_states) and encoder_hidden_states.ndim == 3 and encoder_hidden_states.shape[1] == 1
              and torch.is_tensor(added_time_ids))
        entry = None
        if ok:
            B, Fr, _, H, W = sample.shape
            key = (B, Fr, H, W)
            entry = self._cached.get(key)
            if entry is None:
                with self._lock:
                    entry = self._cached.get(key)
                    if entry is None:
                        try:
                            entry = self._prepare(key, sample, timestep, encoder_hidden_states, added_time_ids)
                        except (NotImplementedError, KeyError) as e:
                            logger.warning("sfast: no native plan for SVD UNet call %s (%s: %s); this shape runs the original forward",
                                           key, type(e).__name__, e)
                            entry = _FALLBACK
                        self._cached[key] = entry
        if not ok or entry is _FALLBACK:
            if not ok and not self._warned:
                logger.warning("sfast: SVD UNet call not handled by the native engine; running the original forward")
                self._warned = True
            return self.orig_forward(sample, timestep, encoder_hidden_states, added_time_ids, return_dict=return_dict)
        plan, graph, env = entry
        with env.lock, torch.cuda.device(eng.device):
            eng.sync_packed()  # packed weight copies follow the live parameters (version counters); a no-op when nothing changed
            eng.load_inputs(plan, sample, timestep, encoder_hidden_states, added_time_ids)
            if graph is not None:
                graph.replay()
            else:
                plan.run(torch.cuda.current_stream(eng.device).cuda_stream)
            out = plan.static_out.clone()
        if not return_dict:
            return (out,)
        return _make_output(out)


def _looks_like_svd_unet(m):
    if not all(hasattr(m, a) for a in ("conv_in", "time_embedding", "add_embedding", "down_blocks", "mid_block", "up_blocks", "conv_out", "config")):
        return False
    try:
        return hasattr(m.down_blocks[0].resnets[0], "temporal_res_block")
    except Exception:
        return False


def _looks_like_controlnet(m):
    return all(hasattr(m, a) for a in ("conv_in", "time_embedding", "down_blocks", "mid_block", "controlnet_cond_embedding",
                                       "controlnet_down_blocks", "controlnet_mid_block", "config"))


def _looks_like_unet2d_condition(m):
    return all(hasattr(m, a) for a in ("conv_in", "time_embedding", "down_blocks", "mid_block", "up_blocks",
                                       "conv_norm_out", "conv_out", "config"))


def compile(m, config):
    device = _device_of(m)
    enable_cuda_graph = config.enable_cuda_graph and device.type == "cuda"

    m.unet = compile_unet(m.unet, config)
    if hasattr(m, "controlnet"):
        m.controlnet = compile_unet(m.controlnet, config)
    if getattr(m, "vae", None) is not None:
        m.vae = compile_vae(m.vae, config)

    if config.enable_jit and config.trace_scheduler and device.type == "cuda" and getattr(m, "scheduler", None) is not None:
        # reference :103-107 wraps scheduler.scale_model_input / scheduler.step in lazy_trace; here the step of a supported
        # scheduler family becomes one HIP kernel with the same call signature (libs/diffusers/scheduler.py)
        from ..libs.diffusers.scheduler import patch_scheduler
        patch_scheduler(m.scheduler)

    if getattr(m, "image_processor", None) is not None:
        # reference :117-122: post-processing moved onto the GPU
        from ..libs.diffusers.image_processor import patch_image_prcessor
        patch_image_prcessor(m.image_processor)

    if enable_cuda_graph:
        for name in ("text_encoder", "text_encoder_2", "image_encoder"):
            enc = getattr(m, name, None)
            if enc is not None:
                enc.forward = _graphed_with_fallback(enc.forward)
    return m


def _graphed_with_fallback(forward):
    graphed = make_dynamic_graphed_callable(forward)
    state = {"ok": True}

    def call(*args, **kwargs):
        if state["ok"]:
            try:
                return graphed(*args, **kwargs)
            except Exception as e:  # capture can fail on data-dependent host code; behave like the eager module
                logger.warning("sfast: hipGraph capture of %s failed (%s); running eagerly",
                               getattr(forward, "__qualname__", forward), e)
                state["ok"] = False
        return forward(*args, **kwargs)

    call.__self__ = getattr(forward, "__self__", None)
    call._cached = graphed._cached
    return call


def compile_unet(m, config):
    device = _device_of(m)
    enable_cuda_graph = config.enable_cuda_graph and device.type == "cuda"

    if config.memory_format is not None:
        apply_memory_format(m, memory_format=config.memory_format)

    native = None
    if config.enable_jit and device.type == "cuda" and _looks_like_controlnet(m):
        from ..engine import ControlNetEngine, UnsupportedUNet
        try:
            cn = ControlNetEngine.from_module(m)
        except UnsupportedUNet as e:
            logger.warning("sfast: %s is outside the native ControlNet engine's coverage (%s); keeping the eager forward",
                           type(m).__name__, e)
            cn = None
        if cn is not None:
            m.forward = _NativeControlNetForward(m, cn, m.forward, enable_cuda_graph)
            m._sfast_engine = cn
            return m
    if config.enable_jit and device.type == "cuda" and _looks_like_svd_unet(m):
        from ..engine import SVDUNetEngine, UnsupportedUNet
        try:
            svd = SVDUNetEngine.from_module(m)
        except UnsupportedUNet as e:
            logger.warning("sfast: %s is outside the native spatio-temporal engine's coverage (%s); keeping the eager forward",
                           type(m).__name__, e)
            svd = None
        if svd is not None:
            m.forward = _NativeSVDForward(m, svd, m.forward, enable_cuda_graph)
            m._sfast_engine = svd
            return m
    if config.enable_jit and device.type == "cuda" and _looks_like_unet2d_condition(m):
        from ..engine import UNet2DEngine, UnsupportedUNet
        try:
            native = UNet2DEngine.from_module(m)
        except UnsupportedUNet as e:
            logger.warning("sfast: %s is outside the native engine's coverage (%s); keeping the eager forward",
                           type(m).__name__, e)
    if native is not None:
        fwd = _NativeUNetForward(m, native, m.forward, enable_cuda_graph)
        m.forward = fwd
        m._sfast_engine = native
    elif enable_cuda_graph:
        m.forward = _graphed_with_fallback(m.forward)
    return m


class _NativeVaeDecoderForward:
    """Replacement for `vae.decoder.forward`: per-shape plan cache (+ hipGraph replay when enabled)."""

    def __init__(self, module, engine, orig_forward, enable_graph):
        self.module, self.engine, self.orig_forward, self.enable_graph = module, engine, orig_forward, enable_graph
        self._cached = {}
        self._lock = threading.Lock()
        self._warned = False
        self.__self__ = module
        self.__name__ = "forward"

    def _prepare(self, key, z):
        eng = self.engine
        B, H, W = key
        with torch.cuda.device(eng.device):
            plan = eng.get_plan(B, H, W)
        graph = None
        torch.cuda.synchronize(eng.device)
        with torch.cuda.device(eng.device), torch.cuda.stream(torch.cuda.Stream(device=eng.device)):
            eng.sync_packed()  # packed weight copies follow the live parameters (version counters); a no-op when nothing changed
            eng.load_inputs(plan, z)
            plan.run(torch.cuda.current_stream(eng.device).cuda_stream)  # validates every launch before capture
        torch.cuda.synchronize(eng.device)
        env = get_per_device_graph_execution_env(eng.device)
        if self.enable_graph:
            from ..engine import capture_plan_graph
            with env.lock:
                with torch.cuda.device(eng.device):
                    graph, _ = capture_plan_graph(plan, env.stream, pool=env.mempool)
                torch.cuda.synchronize(eng.device)
        return plan, graph, env

    def __call__(self, sample, latent_embeds=None, *args, **kwargs):
        eng = self.engine
        _drop_stale_plans(self)
        if (latent_embeds is not None or args or kwargs or not torch.is_tensor(sample) or sample.device.type != "cuda"
                or sample.dtype != eng.dtype or sample.ndim != 4 or sample.shape[1] != eng.in_ch):
            if not self._warned:
                logger.warning("sfast: VAE decoder call not handled by the native engine; running the original forward")
                self._warned = True
            if latent_embeds is not None:
                return self.orig_forward(sample, latent_embeds, *args, **kwargs)
            return self.orig_forward(sample, *args, **kwargs)
        B, _, H, W = sample.shape
        key = (B, H, W)
        entry = self._cached.get(key)
        if entry is None:
            with self._lock:
                entry = self._cached.get(key)
                if entry is None:
                    try:
                        entry = self._prepare(key, sample)
                    except (NotImplementedError, KeyError) as e:
                        logger.warning("sfast: no native plan for VAE call %s (%s: %s); this shape runs the original forward",
                                       key, type(e).__name__, e)
                        entry = _FALLBACK
                    self._cached[key] = entry
        if entry is _FALLBACK:
            return self.orig_forward(sample)
        plan, graph, env = entry
        # static buffers and the workspace are shared by every call of this shape: serialised per device, like the UNet
        # wrapper and the reference's graphed callables (cuda/graphs.py:148)
        with env.lock, torch.cuda.device(eng.device):
            eng.sync_packed()  # packed weight copies follow the live parameters (version counters); a no-op when nothing changed
            eng.load_inputs(plan, sample)
            if graph is not None:
                graph.replay()
            else:
                plan.run(torch.cuda.current_stream(eng.device).cuda_stream)
            return plan.static_out.clone()


def _looks_like_vae_decoder(d):
    return d is not None and all(hasattr(d, a) for a in ("conv_in", "mid_block", "up_blocks", "conv_norm_out", "conv_out"))


def compile_vae(m, config):
    # reference: compilers/diffusion_pipeline_compiler.py:154-190 (memory format, xformers patch, TorchScript fusion of the
    # whole VAE; CUDA graphs deliberately left off there). Here the decoder (every text-to-image call) and the encoder
    # (img2img / inpainting) are handed to the native engine; quant_conv / post_quant_conv
    # (1x1, 8 / 4 channels) and the Gaussian sampling stay on PyTorch-ROCm.
    device = _device_of(m)
    enable_cuda_graph = config.enable_cuda_graph and device.type == "cuda"
    if config.memory_format is not None:
        apply_memory_format(m, memory_format=config.memory_format)
    dec = getattr(m, "decoder", None)
    if config.enable_jit and device.type == "cuda" and _looks_like_vae_decoder(dec):
        from ..engine import UnsupportedUNet, VaeDecoderEngine
        try:
            native = VaeDecoderEngine.from_module(dec, config=getattr(m, "config", None))
        except UnsupportedUNet as e:
            logger.warning("sfast: %s is outside the native VAE engine's coverage (%s); keeping the eager decoder",
                           type(dec).__name__, e)
            native = None
        if native is not None:
            dec.forward = _NativeVaeDecoderForward(dec, native, dec.forward, enable_cuda_graph)
            m._sfast_vae_engine = native
    enc = getattr(m, "encoder", None)
    if config.enable_jit and device.type == "cuda" and enc is not None and all(
            hasattr(enc, a) for a in ("conv_in", "down_blocks", "mid_block", "conv_norm_out", "conv_out")):
        from ..engine import UnsupportedUNet, VaeEncoderEngine
        try:
            native_enc = VaeEncoderEngine.from_module(enc, config=getattr(m, "config", None))
        except UnsupportedUNet as e:
            logger.warning("sfast: %s is outside the native VAE engine's coverage (%s); keeping the eager encoder",
                           type(enc).__name__, e)
            native_enc = None
        if native_enc is not None:
            # same wrapper: one NCHW tensor in, one NCHW tensor out, per-shape plan cache (+ hipGraph)
            enc.forward = _NativeVaeDecoderForward(enc, native_enc, enc.forward, enable_cuda_graph)
            m._sfast_vae_encoder_engine = native_enc
    return m
