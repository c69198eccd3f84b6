"""`trace_scheduler` for MI355X: the scheduler update of the denoise loop as ONE HIP kernel.

The reference's `compile(..., trace_scheduler=True)` wraps `scheduler.scale_model_input` and `scheduler.step` in
`lazy_trace` (/root/reference/src/sfast/compilers/diffusion_pipeline_compiler.py:103-107), turning the dozen small eager
ops of a diffusers scheduler step into one fused TorchScript graph. Here the same two methods are replaced, in place and with
the same call signature, by a native implementation whenever the update is a deterministic two-term linear form

    prev_sample = A(t) * sample + B(t) * model_output

which covers DDIM-family schedulers at eta = 0 for `epsilon` and `v_prediction` models (the configuration BASELINE.json's
metric is quoted on: 50-step DDIM). The coefficient table over all training timesteps is built on the host once per
`set_timesteps()`; the timestep stays wherever the pipeline keeps it -- a 0-d CUDA tensor is read by the kernel itself, so the
step introduces no host synchronisation (diffusers' own DDIM step indexes `alphas_cumprod` with the device timestep, which does).
Anything else -- eta > 0, clipping / thresholding, variance noise, schedulers of another family -- keeps the original method
for that call.
"""
import logging
import math

import torch

logger = logging.getLogger()


class SchedulerOutput:
    """Stand-in for diffusers' `DDIMSchedulerOutput`: `.prev_sample`, tuple-style access, and `pred_original_sample` computed only
    if somebody asks for it (callbacks)."""

    def __init__(self, prev_sample, pred_fn=None):
        self.prev_sample = prev_sample
        self._pred_fn = pred_fn

    @property
    def pred_original_sample(self):
        return self._pred_fn() if self._pred_fn is not None else None

    def __getitem__(self, i):
        return (self.prev_sample,)[i]

    def __iter__(self):
        return iter((self.prev_sample,))


def _cfg(s, name, default=None):
    c = getattr(s, "config", None)
    if c is None:
        return default
    if isinstance(c, dict):
        return c.get(name, default)
    return getattr(c, name, default)


def ddim_like(s):
    """Duck-typed recognition of a DDIM-family scheduler (diffusers DDIMScheduler and subclasses / look-alikes)."""
    if not (hasattr(s, "alphas_cumprod") and hasattr(s, "step") and hasattr(s, "final_alpha_cumprod")):
        return False
    if _cfg(s, "prediction_type", "epsilon") not in ("epsilon", "v_prediction"):
        return False
    if _cfg(s, "clip_sample", False) or _cfg(s, "thresholding", False):
        return False
    return _cfg(s, "num_train_timesteps") is not None


class NativeDDIMStep:
    """Replacement for `scheduler.step` (same signature as diffusers' DDIMScheduler.step)."""

    def __init__(self, scheduler, orig_step):
        self.scheduler = scheduler
        self.orig_step = orig_step
        self._tables = {}  # (device, num_inference_steps) -> float32 [num_train, 2]
        self.__self__ = scheduler
        self.__name__ = "step"
        self.native_calls = 0

    def _table(self, device):
        s = self.scheduler
        n_inf = getattr(s, "num_inference_steps", None)
        if n_inf is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        key = (str(device), int(n_inf))
        t = self._tables.get(key)
        if t is None:
            n_train = int(_cfg(s, "num_train_timesteps"))
            acp = torch.as_tensor(s.alphas_cumprod).double().cpu()
            final = float(torch.as_tensor(s.final_alpha_cumprod))
            ratio = n_train // int(n_inf)
            v_pred = _cfg(s, "prediction_type", "epsilon") == "v_prediction"
            rows = []
            for ts in range(n_train):
                a_t = float(acp[ts])
                prev = ts - ratio
                a_p = float(acp[prev]) if prev >= 0 else final
                sa, s1a, sp, s1p = math.sqrt(a_t), math.sqrt(1 - a_t), math.sqrt(a_p), math.sqrt(1 - a_p)
                if v_pred:   # x0 = sa x - s1a v ; eps = sa v + s1a x
                    rows.append((sp * sa + s1p * s1a, s1p * sa - sp * s1a))
                else:        # x0 = (x - s1a e) / sa ; eps = e
                    rows.append((sp / sa, s1p - sp * s1a / sa))
            t = torch.tensor(rows, dtype=torch.float32, device=device)
            self._tables = {key: t}  # one live schedule at a time
        return t

    def __call__(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None, variance_noise=None,
                 return_dict=True):
        native = (eta == 0.0 and not use_clipped_model_output and variance_noise is None and torch.is_tensor(model_output)
                  and model_output.device.type == "cuda" and torch.is_tensor(sample) and sample.shape == model_output.shape
                  and model_output.dtype in (torch.float16, torch.bfloat16, torch.float32))
        if not native:
            return self.orig_step(model_output, timestep, sample, eta=eta, use_clipped_model_output=use_clipped_model_output,
                                  generator=generator, variance_noise=variance_noise, return_dict=return_dict)
        from ...hip import functional as F
        table = self._table(model_output.device)
        idx = timestep
        if torch.is_tensor(timestep):
            if timestep.device.type == "cuda" and timestep.dtype in (torch.int32, torch.int64) and timestep.numel() == 1:
                idx = timestep.reshape(())
            else:
                idx = int(timestep)
        prev = F.linear_step(model_output, sample, table, idx)
        self.native_calls += 1
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev)


def patch_scheduler(scheduler):
    """In-place: route `scheduler.step` to the native kernel when the scheduler is of a supported family. Returns True if patched."""
    if scheduler is None or isinstance(getattr(scheduler, "step", None), NativeDDIMStep):
        return scheduler is not None
    if ddim_like(scheduler):
        scheduler.step = NativeDDIMStep(scheduler, scheduler.step)
        # DDIM's scale_model_input is the identity (diffusers DDIMScheduler.scale_model_input returns `sample`): nothing to fuse
        return True
    logger.info("sfast: trace_scheduler: %s is not a supported scheduler family; its step stays eager", type(scheduler).__name__)
    return False
