"""MI355X-native captioner forward path (BLIP / CoCa style ViT encoder + autoregressive text decoder)
behind the reference's Captioner plugin API.  Hot path = hand-written HIP (gfx950) in csrc/, reached
through the C ABI declared in include/captioner_hip.h.  Importing this package never imports `oracle/`.
"""
from .config import BlipArch  # noqa: F401

__version__ = "0.1.0"
