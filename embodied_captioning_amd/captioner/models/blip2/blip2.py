"""`BLIP2(cfg)` entry of the plugin factory (reference ``captioner/models/blip2/blip2.py:16-29``).

The reference's BLIP-2 (ViT-g/14 + Q-Former + OPT-2.7B, 8-bit) is listed as a later tier in SURVEY.md §8(f)-4; the
MI355X-native kernels of this package cover the ViT encoder / cross-attention decoder family the north star names
(BLIP-base).  Until the Q-Former and OPT decoder are wired to the same kernels, selecting ``arch_name: blip2`` with a
BLIP(-base) checkpoint runs it through the BLIP path; a genuine BLIP-2 checkpoint raises."""
from ..blip.blip import BLIP


class BLIP2(BLIP):
    def __init__(self, cfg=None):
        name = (cfg.model_name or "").lower()
        if "blip2" in name:
            raise NotImplementedError("BLIP-2 (Q-Former + OPT) checkpoints are not supported by the HIP path yet; "
                                      "use arch_name 'blip' with a BLIP captioning checkpoint")
        super().__init__(cfg)
