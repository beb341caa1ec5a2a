#!/usr/bin/env python
"""Pass plans of the pool's dynamic batching for the bench's 20 timed steps, same process, interleaved (the headline's workload:
BLIP-base, 256 frames per step, f32s, three engines): captions/s per plan.    python tools/plan_experiment.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.config import BlipArch  # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine, EnginePool  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402

B, L, STEPS = 256, 20, 20
arch = BlipArch()
sd = procedural_blip_state_dict(arch, 0, eos_boost=9.0)
one = CaptionerEngine(arch, dtype="f32s", max_batch=B, max_beams=1, max_len=L)
one.load_state_dict(sd)
px = synthetic_pixels(B, arch.image_size, seed=0).cuda()
pool = EnginePool(arch, n=3, dtype="f32s", max_batch=1024, max_beams=1, max_len=L, weights_of=one)
PLANS = {"4,4,3,3,3,3 (product)": [4, 4, 3, 3, 3, 3], "3,3,3,3,4,4": [3, 3, 3, 3, 4, 4], "4,4,4,3,3,2": [4, 4, 4, 3, 3, 2], "4,3,3,4,3,3": [4, 3, 3, 4, 3, 3],
         "4,4,4,4,2,2": [4, 4, 4, 4, 2, 2], "3,4,4,3,3,3": [3, 4, 4, 3, 3, 3]}
orig = EnginePool.coalesce_plan


def with_plan(sizes):
    def plan(rows, n_engines, max_rows):
        if len(rows) != STEPS:
            return orig(rows, n_engines, max_rows)
        out, i = [], 0
        for s in sizes:
            out.append(list(range(i, i + s)))
            i += s
        return out
    return staticmethod(plan)


def run():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pool.generate_many([px] * STEPS, threads=True, coalesce_rows=1024, num_beams=1, max_length=L)
    torch.cuda.synchronize()
    return B * STEPS / (time.perf_counter() - t0)


pool.generate_many([px] * 12, threads=True, coalesce_rows=1024, num_beams=1, max_length=L)
res = {k: [] for k in PLANS}
for rep in range(3):
    for name, sizes in PLANS.items():
        EnginePool.coalesce_plan = with_plan(sizes)
        run()                                            # once untimed: allocator caches for this plan's merged inputs
        res[name].append(run())
for name, v in res.items():
    print(f"{name:24s} " + "  ".join(f"{x:7.0f}" for x in v) + f"   median {sorted(v)[1]:7.0f}", flush=True)
