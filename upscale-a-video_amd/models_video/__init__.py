"""Drop-in `models_video` package: the reference's public classes, re-built on libuav_hip.so."""
from .autoencoder_kl_cond_video import AutoencoderKLVideo  # noqa: F401
from .unet_video import UNetVideoModel  # noqa: F401
from .propagation_module import Propagation  # noqa: F401
