"""`CoCa(cfg)` entry of the plugin factory (reference ``captioner/models/coca/coca.py:19-33``).

CoCa ViT-L/14 (attentional pooler + unimodal/multimodal text towers) reuses the GEMM / attention / LayerNorm kernels
of this package; its tower wiring is not built yet (no importable oracle for open_clip in the build
container), so construction raises instead of silently running something else."""
from ...captioning_predictor import CaptioningPredictor


class CoCa(CaptioningPredictor):
    def __init__(self, cfg=None):
        super().__init__(cfg)
        raise NotImplementedError("CoCa is not wired to the HIP path yet (open_clip is not available to pin an oracle); "
                                  "use arch_name 'blip'")
