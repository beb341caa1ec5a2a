#!/usr/bin/env python
"""Race screen + timing for the fused split-K consumer (GemmParams::ln_counter; CAP_FUSE_LN=1, off by default): a stale read
of another block's partial slab would show up as run-to-run differences.  Repeats full generates and compares tokens AND
every step's logits bitwise with the first run.  Run once with CAP_FUSE_LN=1 and once without for the A/B."""
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
sd = procedural_blip_state_dict(arch, 0, eos_boost=9.0)
reps = int(os.environ.get("REPS", 40))
for B in (256, 8, 64):
    eng = CaptionerEngine(arch, "bf16", B, 1, 20)
    eng.load_state_dict(sd)
    px = synthetic_pixels(B, arch.image_size, seed=0).cuda()
    ref = eng.generate(px, max_length=20, output_logits=True)
    rs, rl = ref["sequences"].clone(), ref["logits"].clone()
    bad = 0
    for i in range(reps):
        o = eng.generate(px, max_length=20, output_logits=True)
        if not torch.equal(o["sequences"], rs) or not torch.equal(o["logits"], rl):
            bad += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.generate(px, max_length=20)
    torch.cuda.synchronize()
    print(f"B={B}: {bad} of {reps} repeats differ from the first run; {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per generate "
          f"(CAP_FUSE_LN={os.environ.get('CAP_FUSE_LN', '0')})", flush=True)
    eng.close()
