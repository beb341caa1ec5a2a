"""Batched counterpart of the reference's per-box captioning loop
(``experimenting_env/detector/pseudolabeler.py:629-711``: `expand_box`, `predict_captions`, `predict_caption`).

The reference crops one box at a time and calls the captioner with batch 1; here every box of a dataloader batch is
expanded (+20 %), cropped, and captioned in ONE `caption_batch` call (micro-batched inside the engine), then the
captions are handed back per frame in box order - same crop arithmetic, same BGR->RGB swap, same caption order.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
import torch


def expand_box(box, expand_factor: float, image_size) -> np.ndarray:
    """Reference `expand_box` (:629-643), including its naming quirk: `image_size` is `image.shape` = (H, W, C), and
    the reference clamps x to image_size[0] and y to image_size[1].  fp32 arithmetic like the torch scalars it uses;
    `int()` truncates toward zero."""
    x1, y1, x2, y2 = (torch.as_tensor(v, dtype=torch.float32) for v in box)
    width, height = image_size[0], image_size[1]
    box_width = x2 - x1
    box_height = y2 - y1
    new_x1 = int(max(x1 - expand_factor * box_width, 0))
    new_y1 = int(max(y1 - expand_factor * box_height, 0))
    new_x2 = int(min(x2 + expand_factor * box_width, width))
    new_y2 = int(min(y2 + expand_factor * box_height, height))
    return np.array([new_x1, new_y1, new_x2, new_y2])


def crop_boxes(image_bgr: np.ndarray, boxes: Sequence, expand_factor: float = 0.2) -> list:
    """BGR uint8 HWC frame + boxes (x1,y1,x2,y2) -> list of RGB PIL crops (reference :670-675, :694-697)."""
    from PIL import Image
    rgb = np.ascontiguousarray(image_bgr[..., ::-1])                     # cv2.COLOR_BGR2RGB
    pil = Image.fromarray(rgb)
    return [pil.crop(tuple(int(v) for v in expand_box(b, expand_factor, rgb.shape))) for b in boxes]


def record_name(episode: int, step: int) -> str:
    """File name of a pseudo-label record (reference :835-843)."""
    return f"episode_{episode}_step_{step}.npz"


def save_record(output_path: str, info: str, instances, image) -> str:
    """Write one pseudo-label record the way the reference does (:833-842): `np.savez_compressed(<output_path>/<info>.npz,
    {'instances': instances, 'image': image})` - a single pickled dict under `arr_0` (`np.load(f, allow_pickle=True)
    ['arr_0'].item()` gives it back).  `instances` is whatever carries `.captions` / `.embeddings` downstream (detectron2
    `Instances` in the reference; any picklable object here).  Returns the file name."""
    import os
    filename = os.path.join(output_path, f"{info}.npz")
    np.savez_compressed(filename, {"instances": instances, "image": image})
    return filename


class BatchedBoxCaptioner:
    """`captioner` is the plugin (`Captioner` with `caption_batch`) or any callable list[PIL] -> list[str];
    `encoder` (optional): an object with `encode(list[str], convert_to_tensor=True)` (SentenceEncoder, or the reference's
    SentenceTransformer) - called once for the whole batch - or a plain callable caption -> vector."""

    def __init__(self, captioner, encoder: Optional[Callable[[str], torch.Tensor]] = None, expand_factor: float = 0.2,
                 device_resize: Optional[bool] = None):
        self.captioner = captioner
        self.encoder = encoder
        self.expand_factor = expand_factor
        # Crop + bicubic resize on the device (bit-exact with the PIL path, preprocess.crop_resize_u8) when the captioner's
        # processor is a plain square resize (BLIP / BLIP-2 plugins expose `direct_resize_size`) or the shorter-side resize + centre crop of CoCa
        # (`shorter_side_resize_size`); None = use it if possible.
        size = getattr(captioner, "direct_resize_size", None)
        self._center_crop = False
        if size is None and getattr(captioner, "shorter_side_resize_size", None) is not None:   # CoCa's transform
            size, self._center_crop = captioner.shorter_side_resize_size, True
        can = size is not None and torch.cuda.is_available()
        if device_resize and not can:
            raise ValueError("device_resize needs a GPU and a captioner with `direct_resize_size`")
        self.device_resize = can if device_resize is None else bool(device_resize)
        self._size = int(size) if size is not None else None

    def _caption(self, crops) -> List[str]:
        if len(crops) == 0:
            return []
        fn = getattr(self.captioner, "caption_batch", None)
        return list(fn(crops)) if fn is not None else list(self.captioner(crops))

    def predict_captions(self, boxes_per_frame: Sequence[Sequence], frames_bgr: Sequence[np.ndarray]):
        """One captioner call for the whole dataloader batch.  Returns per frame
        {"captions": [str], "embeddings": tensor [n, d] | tensor([])} in box order (reference :664-688)."""
        crops, owner = [], []
        on_device = self.device_resize
        if on_device:
            # every box of every frame in one packed upload and two launches (preprocess.crop_resize_u8_frames): only the boxes'
            # in-frame pixels travel - a 1280 x 1280 frame with three boxes sends ~0.5 MB, not 4.9
            from .preprocess import crop_resize_u8_frames
            rects = [[expand_box(b, self.expand_factor, img.shape) for b in boxes] for boxes, img in zip(boxes_per_frame, frames_bgr)]
            if any(r[2] <= r[0] or r[3] <= r[1] for rs in rects for r in rs):      # an empty rectangle somewhere: the PIL path (raises as the reference)
                return BatchedBoxCaptioner(self.captioner, self.encoder, self.expand_factor, device_resize=False) \
                    .predict_captions(boxes_per_frame, frames_bgr)
            dev = getattr(self.captioner, "crop_device", None) or getattr(self.captioner, "device", "cuda:0")
            for fi, rs in enumerate(rects):
                owner += [fi] * len(rs)
            if owner:
                crops = crop_resize_u8_frames([np.asarray(f) for f in frames_bgr], rects, self._size, bgr=True, device=dev, center_crop=self._center_crop)
        else:
            for fi, (boxes, img) in enumerate(zip(boxes_per_frame, frames_bgr)):
                if len(boxes) == 0:
                    continue
                crops.extend(crop_boxes(img, boxes, self.expand_factor))
                owner += [fi] * len(boxes)
        captions = self._caption(crops)
        out = [{"captions": [], "embeddings": torch.tensor([])} for _ in frames_bgr]
        for fi, cap in zip(owner, captions):
            out[fi]["captions"].append(cap)
        if self.encoder is not None and captions:
            enc = getattr(self.encoder, "encode", None)
            if enc is not None:                         # SentenceEncoder / SentenceTransformer: one batched call
                emb = torch.as_tensor(enc(list(captions), convert_to_tensor=True)).cpu()
            else:                                       # plain callable caption -> vector
                emb = torch.stack([torch.as_tensor(self.encoder(c)).cpu() for c in captions])
            for fi in range(len(out)):
                idx = [i for i, o in enumerate(owner) if o == fi]
                if idx:
                    out[fi]["embeddings"] = emb[idx]
        return out

    def predict_caption(self, boxes: Sequence, image_bgr: np.ndarray):
        """Single-frame form (reference :690-711)."""
        return self.predict_captions([boxes], [image_bgr])[0]
