#!/usr/bin/env python
"""A list of PIL crops of different sizes (what `generate_batch` / `caption_batch` receive) -> uint8 [n, S, S, 3]: host PIL bicubic per
image against one cap_crop_resize_u8 call per image (upload + table kernel + resize kernel), bit-identity checked.
    python tools/pil_list_bench.py"""
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.preprocess import crop_resize_u8, resize_u8_list  # noqa: E402

rng = np.random.default_rng(0)
S, n = 224, 256
ims = [Image.fromarray(rng.integers(0, 256, size=(int(rng.integers(40, 400)), int(rng.integers(40, 400)), 3), dtype=np.uint8), "RGB") for _ in range(n)]
t0 = time.perf_counter()
ref = [np.asarray(im.convert("RGB").resize((S, S), resample=Image.BICUBIC)) for im in ims]
t_pil = time.perf_counter() - t0


def device_path():
    outs = []
    for im in ims:
        a = np.asarray(im.convert("RGB"))
        outs.append(crop_resize_u8(a, [(0, 0, a.shape[1], a.shape[0])], S))
    return torch.cat(outs)


out = device_path(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out = device_path()
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / 3
same = all(np.array_equal(out[i].cpu().numpy(), ref[i]) for i in range(n))


def list_path():
    return resize_u8_list([np.asarray(im.convert("RGB")) for im in ims], S)


out2 = list_path(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out2 = list_path()
torch.cuda.synchronize()
t_list = (time.perf_counter() - t0) / 3
same = same and all(np.array_equal(out2[i].cpu().numpy(), ref[i]) for i in range(n))
print(f"{n} PIL crops (40-400 px) -> {S}x{S}: host PIL {t_pil * 1e3:.1f} ms ({t_pil / n * 1e3:.2f} ms each, one core); one device call per image "
      f"{t_dev * 1e3:.1f} ms ({t_dev / n * 1e6:.0f} us each); the whole list in one packed upload + two launches (resize_u8_list) "
      f"{t_list * 1e3:.1f} ms ({t_list / n * 1e6:.0f} us each); bit-identical: {same}")
