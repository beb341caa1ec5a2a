#!/usr/bin/env python
"""Where the joules of a 256-frame batch go (GPU box): socket power from the GPU's hwmon node while (a) only the image tower
runs, (b) whole generates run one at a time, (c) whole generates overlap on three streams.  Uses bench.PowerSampler.
    python tools/energy_split.py [dtype]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import PowerSampler  # noqa: E402
from embodied_captioning_amd.config import BlipArch  # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine, EnginePool  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32s"
arch = BlipArch()
sd = procedural_blip_state_dict(arch, 0, eos_boost=9.0)
B, L = 256, 20
px = synthetic_pixels(B, arch.image_size, seed=0).cuda()
eng = CaptionerEngine(arch, dtype=dtype, max_batch=B, max_beams=1, max_len=L)
eng.load_state_dict(sd)
pool = EnginePool(arch, n=3, dtype=dtype, max_batch=B, max_beams=1, max_len=L, weights_of=eng)


def measure(name, fn, n):
    fn(3)
    torch.cuda.synchronize()
    with PowerSampler(0) as ps:
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    r = ps.result(B * n, dt)
    print(f"{dtype} {name:28s}: {1e3 * dt / n:7.2f} ms per batch, {r['watts_mean']:7.1f} W mean -> {r['watts_mean'] * dt / n:6.1f} J per batch", flush=True)


measure("image tower only", lambda n: [eng.encode(px) for _ in range(n)], 120)
measure("generate, one stream", lambda n: [eng.generate(px, max_length=L) for _ in range(n)], 60)
measure("generate, three streams", lambda n: pool.generate_many([px] * n, threads=True, max_length=L), 90)
