"""Device capability probes used by CompilationConfig defaults.

Mirrors /root/reference/src/sfast/utils/gpu_device.py:4-15 (same names / meaning). On ROCm,
`torch.cuda.get_device_capability()` reports (9, x) for gfx9xx, so the reference's `major >= 7`
(tensor cores) and `>= (8, 0)` gates read as "has MFMA matrix cores", which is what they select
for here.
"""
import torch


def device_has_tensor_core():
    if torch.cuda.is_available():
        major, _ = torch.cuda.get_device_capability()
        return major >= 7
    return False


def device_has_capability(major, minor):
    if torch.cuda.is_available():
        return tuple(torch.cuda.get_device_capability()) >= (major, minor)
    return False


def device_is_gfx950():
    if not torch.cuda.is_available():
        return False
    name = getattr(torch.cuda.get_device_properties(0), "gcnArchName", "")
    return "gfx950" in name
