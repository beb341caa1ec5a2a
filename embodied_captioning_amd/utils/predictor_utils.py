"""`Captioner` plugin - mirror of the reference's ``experimenting_env/utils/predictor_utils.py:166-208``: same
constructor signature, `.forward(x) -> str`, `get_captioner(cfg)` dispatch on ``cfg.arch_name``; plus a batched
`caption_batch` for the pseudo-labeler driver."""
from __future__ import annotations

import logging

import torch

from ..captioner.captioning_predictor import _Base
from ..captioner.utils.utils import Configuration
from ..captioner.utils.utils_captioner import select_captioner

logger = logging.getLogger(__name__)


class Captioner(_Base):
    def __init__(self, cfg=None, input_format=None, load_checkpoint=False, checkpoint_path=None, metadata=None,
                 model=None):
        super().__init__()
        if model is None:
            self.model = self.get_captioner(cfg.captioner)
        else:
            self.model = model
        self.model.eval()
        assert self.model is not None, "No model provided"
        logger.info("Captioner model loaded successfully")
        if load_checkpoint and checkpoint_path is not None:
            checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
            res = self.model.load_checkpoint_state_dict(checkpoint["model"], strict=False)
            logger.info(f"Captioner model checkpoint loaded successfully from {checkpoint_path} "
                        f"({res['matched']} tensors, {len(res['unknown'])} not stored by this architecture)")

    def get_captioner(self, cfg):
        """Get the captioner model based on the configuration settings (reference :190-202)."""
        extra = {k: getattr(cfg, k) for k in ("num_beams", "max_length", "max_new_tokens", "dtype", "batch_size", "device",
                                              "streams", "early_exit_poll", "image_size", "num_beam_groups", "tokenizer_dir",
                                              "coalesce_rows", "cross_cache", "strict_range")
                 if hasattr(cfg, k)}
        captioner_cfg = Configuration(arch_name=cfg.arch_name, model_name=cfg.model_name,
                                      checkpoint_name=getattr(cfg, "checkpoint_name", None), height=cfg.height,
                                      width=cfg.width, **extra)
        return select_captioner(captioner_cfg.captioner).eval()

    def forward(self, x):
        out = self.model(x)
        caption = out["text"]
        return caption

    def caption_batch(self, images):
        """Batched extension: list of PIL images / uint8 tensor -> list of captions."""
        return self.model.generate_batch(images)["texts"]

    @property
    def direct_resize_size(self):
        """Side of the processor's plain square resize (BLIP / BLIP-2), or None (CoCa: aspect-preserving resize + centre
        crop): lets the box driver crop + resize on the device (pseudolabeler.BatchedBoxCaptioner)."""
        return getattr(self.model, "direct_resize_size", None)

    @property
    def shorter_side_resize_size(self):
        """CoCa: side of the aspect-preserving resize + centre crop (None for the square-resize processors)."""
        return getattr(self.model, "shorter_side_resize_size", None)

    @property
    def crop_device(self):
        return getattr(self.model, "device", "cuda:0")
