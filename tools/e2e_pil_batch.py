#!/usr/bin/env python
"""End to end through the plugin's batch entry point, from PIL crops to strings: `BLIP.generate_batch(list of PIL)` with
`captioner.streams: 3` - where the time goes (preprocess / generate / detokenise).    python tools/e2e_pil_batch.py [n]"""
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.captioner.utils.utils import Configuration  # noqa: E402
from embodied_captioning_amd.captioner.utils.utils_captioner import select_captioner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
ims = [Image.fromarray(rng.integers(0, 256, size=(int(rng.integers(40, 400)), int(rng.integers(40, 400)), 3), dtype=np.uint8), "RGB") for _ in range(n)]
for dr in (True, False):
    m = select_captioner(Configuration(arch_name="blip", model_name="procedural-blip:0:9.0", height=224, width=224, batch_size=256, streams=3,
                                       device_resize=dr).captioner).eval()
    m.generate_batch(ims[:512])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); px = m.preprocess(ims); torch.cuda.synchronize(); t1 = time.perf_counter()
    out = m.generate_batch(ims); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"device_resize={dr}: {n} PIL crops -> captions {1e3 * (t2 - t1):.1f} ms end to end = {n / (t2 - t1):.0f} captions/s "
          f"(preprocess alone {1e3 * (t1 - t0):.1f} ms)", flush=True)
    del m
