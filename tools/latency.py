#!/usr/bin/env python
"""Caption latency at the batch sizes the reference's callers use (one crop per call; a handful of boxes per frame)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd.config import BlipArch  # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402

arch = BlipArch()
eng = CaptionerEngine(arch, "bf16", 64, 3, 20)
eng.load_state_dict(procedural_blip_state_dict(arch, 0, eos_boost=9.0))
for B, beams in [(1, 1), (8, 1), (64, 1), (1, 3), (8, 3)]:
    px = synthetic_pixels(B, arch.image_size, seed=0).cuda()
    for _ in range(3):
        eng.generate(px, num_beams=beams, max_length=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        eng.generate(px, num_beams=beams, max_length=20)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"batch {B:3d} beams {beams}: {ms:7.2f} ms per call, {B / ms * 1e3:8.1f} captions/s", flush=True)
