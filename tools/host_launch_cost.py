"""Host time of one cap_generate call (asynchronous: launches only) against its GPU time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.config import BlipArch  # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402

B, L = int(os.environ.get("B", 256)), 20
arch = BlipArch()
eng = CaptionerEngine(arch, dtype="bf16", max_batch=B, max_beams=1, max_len=L)
eng.load_state_dict(procedural_blip_state_dict(arch, 0, eos_boost=9.0))
px = synthetic_pixels(B, arch.image_size, seed=3).cuda()
for _ in range(2):
    eng.generate(px, max_length=L)
torch.cuda.synchronize()
host, total = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.generate(px, max_length=L)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print(f"B={B}: host side of cap_generate {min(host):.2f} ms (launches only), until the GPU is done {min(total):.2f} ms", flush=True)
