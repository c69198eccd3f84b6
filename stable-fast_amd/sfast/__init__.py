"""sfast -- stable-fast's drop-in surface, rebuilt MI355X-native (gfx950 / CDNA4).

`import sfast` registers the reference's operator namespaces (`torch.ops.sfast`, `sfast_triton`,
`sfast_xformers`) on top of the hand-written HIP kernel library libsfast_hip.so and exposes
`sfast.compilers.diffusion_pipeline_compiler.{compile, CompilationConfig}`. The kernel library is
loaded lazily on first use and is mandatory on a GPU box (no eager / ATen fallback).
"""
__version__ = "0.1.0+mi355x"

from . import torch_ops  # noqa: F401,E402  (torch.ops.sfast.*)
from .triton import torch_ops as _triton_ops  # noqa: F401,E402  (torch.ops.sfast_triton.*)
from .libs.xformers import xformers_attention as _xf_ops  # noqa: F401,E402  (torch.ops.sfast_xformers.*)
