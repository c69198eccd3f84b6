"""Device capability probes behind the CompilationConfig defaults (names and meaning of
/root/reference/src/sfast/utils/gpu_device.py:4-15).

PyTorch-ROCm reports gfx9xx parts as compute capability (9, x), so the reference's two gates -- `major >= 7` ("has tensor
cores") and `>= (8, 0)` -- both read as "has MFMA matrix cores" on an Instinct GPU, which is what they select for here.
"""
import torch


def _capability():
    return tuple(torch.cuda.get_device_capability()) if torch.cuda.is_available() else None


def device_has_capability(major, minor):
    cap = _capability()
    return cap is not None and cap >= (major, minor)


def device_has_tensor_core():
    return device_has_capability(7, 0)


def device_is_gfx950():
    if _capability() is None:
        return False
    return "gfx950" in getattr(torch.cuda.get_device_properties(0), "gcnArchName", "")
