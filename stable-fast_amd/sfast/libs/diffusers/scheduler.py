"""`trace_scheduler` for MI355X: the scheduler update of the denoise loop as ONE HIP kernel.

The reference's `compile(..., trace_scheduler=True)` wraps `scheduler.scale_model_input` and `scheduler.step` in
`lazy_trace` (/root/reference/src/sfast/compilers/diffusion_pipeline_compiler.py:103-107), turning the dozen small eager
ops of a diffusers scheduler step into one fused TorchScript graph. Here the same two methods are replaced, in place and with
the same call signature, by a native implementation whenever the update is a deterministic two-term linear form

    prev_sample = A(t) * sample + B(t) * model_output

which covers DDIM-family schedulers at eta = 0 for `epsilon` and `v_prediction` models (the configuration BASELINE.json's
metric is quoted on: 50-step DDIM) and the first-order Euler update of `EulerDiscreteScheduler` (the default of diffusers' SDXL
pipelines; there `scale_model_input` is a kernel of the same form too). The coefficient table over all training timesteps is built on the host once per
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


_DDIM_CLASSES = ("DDIMScheduler", "DDIMParallelScheduler")


def ddim_like(s):
    """diffusers' DDIMScheduler / DDIMParallelScheduler, a subclass, or a declared look-alike (`_sfast_ddim_like = True`).
    Recognised by class name, never by attributes: PNDMScheduler (the SD1.5 default), LCMScheduler, TCDScheduler and
    DDIMInverseScheduler carry the same `alphas_cumprod` / `final_alpha_cumprod` / `step` surface with multistep, stochastic or
    inverse arithmetic -- the reference's lazy_trace keeps each scheduler's own math, so anything else stays eager."""
    names = {c.__name__ for c in type(s).__mro__}
    if not (names & set(_DDIM_CLASSES)) and not getattr(s, "_sfast_ddim_like", False):
        return False
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


def euler_like(s):
    """diffusers' EulerDiscreteScheduler (SDXL's default) or a subclass / declared look-alike: a sigma schedule, a host-side step index
    and a deterministic first-order update. Recognised by class name, not by attributes: the multistep / ancestral / Heun families
    carry the same attributes (`sigmas`, `_step_index`) with different arithmetic."""
    names = {c.__name__ for c in type(s).__mro__}
    if "EulerDiscreteScheduler" not in names and not getattr(s, "_sfast_euler_like", False):
        return False
    if not (hasattr(s, "sigmas") and hasattr(s, "_init_step_index") and hasattr(s, "scale_model_input") and hasattr(s, "step")):
        return False
    return _cfg(s, "prediction_type", "epsilon") in ("epsilon", "v_prediction")


class _EulerTables:
    """(A, B) rows of the update and the input scale per step index, rebuilt whenever `set_timesteps` installs new sigmas.
        epsilon:       prev = x + (s' - s) e                                   x_in = x / sqrt(s^2 + 1)
        v_prediction:  prev = x (1 + (s' - s) s / (s^2 + 1)) + v (s' - s) / sqrt(s^2 + 1)
    (diffusers EulerDiscreteScheduler.step with s_churn = 0: sigma_hat = sigma, derivative = (x - x0) / sigma, dt = s' - s)."""

    def __init__(self, scheduler):
        self.scheduler = scheduler
        self._key, self._sig, self._step, self._scale = None, None, None, None

    def get(self, device):
        sig = self.scheduler.sigmas
        # identity of the installed schedule without touching its values (a device-resident `sigmas` would cost a sync per step):
        # `set_timesteps` binds a new tensor, in-place edits bump `_version`. The keyed tensor is held by a strong reference and
        # compared with `is`, so a freed schedule's id / storage address cannot be recycled into a false match.
        key = (int(len(sig)), getattr(sig, "_version", 0), str(device))
        if sig is not self._sig or key != self._key:
            self._sig = sig
            sg = torch.as_tensor(sig).double().cpu()
            s0, s1 = sg[:-1], sg[1:]
            dt = s1 - s0
            if _cfg(self.scheduler, "prediction_type", "epsilon") == "v_prediction":
                a, b = 1.0 + dt * s0 / (s0 * s0 + 1.0), dt / torch.sqrt(s0 * s0 + 1.0)
            else:
                a, b = torch.ones_like(dt), dt
            self._step = torch.stack([a, b], dim=1).to(torch.float32).to(device).contiguous()
            self._scale = torch.stack([1.0 / torch.sqrt(s0 * s0 + 1.0), torch.zeros_like(s0)], dim=1).to(torch.float32).to(device).contiguous()
            self._key = key
        return self._step, self._scale


def _native_tensor(t, like=None):
    return (torch.is_tensor(t) and t.device.type == "cuda" and t.dtype in (torch.float16, torch.bfloat16, torch.float32)
            and (like is None or t.shape == like.shape))


class NativeEulerStep:
    """Replacement for `EulerDiscreteScheduler.step` (same signature): one kernel, `prev = A[i] x + B[i] model_output` in fp32, the
    step index kept on the host exactly as diffusers keeps it (`_init_step_index` on first use, `+= 1` per call). Stochastic churn
    (s_churn > 0) and anything else the two-term form does not cover take the original method."""

    def __init__(self, scheduler, orig_step, tables):
        self.scheduler, self.orig_step, self.tables = scheduler, orig_step, tables
        self.__self__ = scheduler
        self.__name__ = "step"
        self.native_calls = 0

    def __call__(self, model_output, timestep, sample, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, generator=None,
                 return_dict=True, **kw):
        s = self.scheduler
        if (kw or s_churn != 0.0 or not (_native_tensor(model_output) and _native_tensor(sample, model_output))
                or sample.dtype != model_output.dtype):  # diffusers upcasts an fp32 sample before the update: keep its arithmetic
            return self.orig_step(model_output, timestep, sample, s_churn=s_churn, s_tmin=s_tmin, s_tmax=s_tmax, s_noise=s_noise,
                                  generator=generator, return_dict=return_dict, **kw)
        from ...hip import functional as F
        if getattr(s, "step_index", None) is None:
            s._init_step_index(timestep)
        i = int(s.step_index)
        step_tab, _ = self.tables.get(model_output.device)
        prev = F.linear_step(model_output, sample, step_tab, i)
        s._step_index = i + 1
        self.native_calls += 1
        if not return_dict:
            return (prev,)

        def pred_x0():  # only if a callback asks: x0 = x - s e   /   x / (s^2 + 1) - v s / sqrt(s^2 + 1)
            sg = float(torch.as_tensor(s.sigmas)[i])
            if _cfg(s, "prediction_type", "epsilon") == "v_prediction":
                return (sample.float() / (sg * sg + 1.0) - model_output.float() * sg / math.sqrt(sg * sg + 1.0)).to(model_output.dtype)
            return (sample.float() - sg * model_output.float()).to(model_output.dtype)

        return SchedulerOutput(prev, pred_x0)


class NativeEulerScale:
    """Replacement for `EulerDiscreteScheduler.scale_model_input`: `sample / sqrt(sigma_i^2 + 1)` as one launch of the same kernel."""

    def __init__(self, scheduler, orig, tables):
        self.scheduler, self.orig, self.tables = scheduler, orig, tables
        self.__self__ = scheduler
        self.__name__ = "scale_model_input"

    def __call__(self, sample, timestep, *a, **kw):
        s = self.scheduler
        if a or kw or not _native_tensor(sample) or sample.dtype == torch.float32:
            return self.orig(sample, timestep, *a, **kw)
        from ...hip import functional as F
        if getattr(s, "step_index", None) is None:
            s._init_step_index(timestep)
        _, scale_tab = self.tables.get(sample.device)
        out = F.linear_step(sample, sample, scale_tab, int(s.step_index))
        s.is_scale_input_called = True
        return out


def patch_scheduler(scheduler):
    """In-place: route `scheduler.step` to the native kernel when the scheduler is of a supported family. Returns True if patched."""
    if scheduler is None or isinstance(getattr(scheduler, "step", None), (NativeDDIMStep, NativeEulerStep)):
        return scheduler is not None
    if ddim_like(scheduler):
        scheduler.step = NativeDDIMStep(scheduler, scheduler.step)
        # DDIM's scale_model_input is the identity (diffusers DDIMScheduler.scale_model_input returns `sample`): nothing to fuse
        return True
    if euler_like(scheduler):
        tables = _EulerTables(scheduler)
        scheduler.step = NativeEulerStep(scheduler, scheduler.step, tables)
        scheduler.scale_model_input = NativeEulerScale(scheduler, scheduler.scale_model_input, tables)
        return True
    logger.info("sfast: trace_scheduler: %s is not a supported scheduler family; its step stays eager", type(scheduler).__name__)
    return False
