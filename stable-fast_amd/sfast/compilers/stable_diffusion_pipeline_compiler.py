"""Deprecated alias kept for import compatibility
(/root/reference/src/sfast/compilers/stable_diffusion_pipeline_compiler.py:1-8)."""
import logging

from .diffusion_pipeline_compiler import *  # noqa: F401,F403
from .diffusion_pipeline_compiler import CompilationConfig, compile, compile_unet, compile_vae  # noqa: F401

logging.getLogger().warning(
    "sfast.compilers.stable_diffusion_pipeline_compiler is deprecated, use sfast.compilers.diffusion_pipeline_compiler")
