"""sfast.engine -- static-plan executors for the hot path (UNet forward)."""
from .unet2d import ControlNetEngine, UNet2DEngine, UNetPlan, UnsupportedUNet, capture_plan_graph  # noqa: F401
from .vae import UnsupportedVae, VaeDecoderEngine, VaeEncoderEngine  # noqa: F401
from .svd import SVDUNetEngine  # noqa: F401
from .denoise import DenoiseLoop, ddim_schedule  # noqa: F401
