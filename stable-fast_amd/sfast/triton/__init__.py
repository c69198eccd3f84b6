"""Kept for import-path compatibility with the reference (`sfast.triton.torch_ops`). There is no
Triton here: the `sfast_triton::*` operator names are served by the gfx950 HIP kernels."""
