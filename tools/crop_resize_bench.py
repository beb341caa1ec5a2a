"""Crop + bicubic resize of object boxes: host PIL (what the reference does per crop) against cap_crop_resize_u8."""
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.preprocess import crop_resize_u8  # noqa: E402

rng = np.random.default_rng(0)
H, W, S, n = 480, 640, int(os.environ.get("S", 224)), 64
frame = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
rects = []
for _ in range(n):
    w, h = int(rng.integers(40, 400)), int(rng.integers(40, 400))
    x, y = int(rng.integers(0, W - w)), int(rng.integers(0, H - h))
    rects.append((x, y, x + w, y + h))
pil = Image.fromarray(frame)
t0 = time.perf_counter()
ref = [np.asarray(pil.crop(r).resize((S, S), resample=Image.BICUBIC)) for r in rects]
t_pil = time.perf_counter() - t0
fd = torch.from_numpy(frame).cuda()
out = crop_resize_u8(fd, rects, S)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = crop_resize_u8(fd, rects, S)
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = crop_resize_u8(fd, rects, S)
torch.cuda.synchronize()
same = all(np.array_equal(out[i].cpu().numpy(), ref[i]) for i in range(n))
print(f"{n} crops of a {W}x{H} frame -> {S}x{S}: PIL {t_pil * 1e3:.1f} ms ({t_pil / n * 1e3:.2f} ms/crop, one core), "
      f"device path {t_all * 1e3:.2f} ms per call incl. the table kernel ({t_all / n * 1e6:.0f} us/crop), bit-identical: {same}")
