"""GPU-side image post-processing for diffusers' `VaeImageProcessor`.

Mirror of /root/reference/src/sfast/libs/diffusers/image_processor.py (:13-20 `patch_image_prcessor` -- the reference's
own spelling, kept so callers port unchanged -- and :23-108 `postprocess` / `pt_to_numpy` / `pt_to_pil`): the reference
moves denormalize + permute + (for PIL) mul(255).round().to(uint8) onto the GPU with TorchScript so only the final bytes
cross PCIe. Here those steps are ONE HIP kernel (`sfast_hip_image_postprocess`: NCHW f16 -> NHWC uint8 / float32,
denormalisation fused), for CUDA(=ROCm) tensors; CPU tensors keep the processor's original methods.
"""
import logging
from typing import List, Optional

import numpy as np
import torch

logger = logging.getLogger()


def patch_image_prcessor(processor):
    # the reference only patches exactly `VaeImageProcessor` (:14); diffusers is not importable here, so the check is
    # structural: the three methods it replaces must exist
    if all(hasattr(processor, a) for a in ("postprocess", "pt_to_numpy", "pt_to_pil")):
        processor._sfast_orig = (processor.postprocess, processor.pt_to_numpy, processor.pt_to_pil)
        processor.postprocess = postprocess.__get__(processor)
        processor.pt_to_numpy = pt_to_numpy
        processor.pt_to_pil = pt_to_pil
    else:
        logger.warning(f"Image processor {type(processor)} is not supported for patching")
    return processor


def _native(images):
    return torch.is_tensor(images) and images.device.type == "cuda" and images.ndim == 4


def postprocess(self, image, output_type: str = "pil", do_denormalize: Optional[List[bool]] = None):
    if not isinstance(image, torch.Tensor):
        raise ValueError(f"Input for postprocessing is in incorrect format: {type(image)}. We only support pytorch tensor")
    if output_type not in ["latent", "pt", "np", "pil"]:
        output_type = "np"  # the reference deprecates unknown types to `np` (:31-41)
    if output_type == "latent":
        return image
    cfg = getattr(self, "config", None)
    norm = bool(getattr(cfg, "do_normalize", True))
    if do_denormalize is None:
        do_denormalize = [norm] * image.shape[0]
    uniform = all(do_denormalize) or not any(do_denormalize)
    if output_type in ("np", "pil") and _native(image) and uniform:
        from ...hip import functional as F
        out = F.image_postprocess(image, denormalize=bool(do_denormalize[0]), to_uint8=(output_type == "pil"))
        arr = out.cpu().numpy()
        return arr if output_type == "np" else _to_pil(arr)
    image = torch.stack([(image[i] / 2 + 0.5).clamp(0, 1) if do_denormalize[i] else image[i] for i in range(image.shape[0])])
    if output_type == "pt":
        return image
    if output_type == "pil":
        return pt_to_pil(image)
    return pt_to_numpy(image)


def pt_to_numpy(images) -> np.ndarray:
    if _native(images):
        from ...hip import functional as F
        return F.image_postprocess(images, denormalize=False, to_uint8=False).cpu().numpy()
    return images.permute(0, 2, 3, 1).contiguous().float().cpu().numpy()


def _to_pil(arr):
    from PIL import Image
    if arr.shape[-1] == 1:
        return [Image.fromarray(a.squeeze(), mode="L") for a in arr]
    return [Image.fromarray(a) for a in arr]


def pt_to_pil(images):
    if images.ndim == 3:
        images = images[None, ...]
    if _native(images):
        from ...hip import functional as F
        arr = F.image_postprocess(images, denormalize=False, to_uint8=True).cpu().numpy()
    else:
        arr = images.permute(0, 2, 3, 1).contiguous().float().mul(255).round().to(dtype=torch.uint8).cpu().numpy()
    return _to_pil(arr)
