"""Old module path of the pipeline compiler. The reference keeps it as a deprecated alias
(/root/reference/src/sfast/compilers/stable_diffusion_pipeline_compiler.py:1-8); here every attribute is forwarded
lazily (PEP 562) to `diffusion_pipeline_compiler`, with one DeprecationWarning per process."""
import importlib
import warnings

_TARGET = "sfast.compilers.diffusion_pipeline_compiler"
_warned = False


def _target():
    global _warned
    if not _warned:
        _warned = True
        warnings.warn(f"{__name__} is deprecated; import {_TARGET}", DeprecationWarning, stacklevel=3)
    return importlib.import_module(_TARGET)


def __getattr__(name):
    try:
        return getattr(_target(), name)
    except AttributeError:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None


def __dir__():
    return sorted(set(globals()) | set(dir(_target())))
