"""sfast.hip -- binding of the gfx950 kernel library (libsfast_hip.so)."""
from .lib import SfastHipError, load, init_device, last_error, last_kernel, LIB_PATH  # noqa: F401
from . import functional  # noqa: F401
