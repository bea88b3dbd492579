"""uav — host-side binding of libuav_hip.so, the MI355X kernel library of the Upscale-A-Video hot path."""
from . import _lib, build, ops  # noqa: F401
from ._lib import UavError, load  # noqa: F401
