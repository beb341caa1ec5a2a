#!/usr/bin/env python
"""Determinism soak of the engine pool's dynamic batching on the CoCa (beam 5) and BLIP-2 (f32s / int8) wrappers' engines: N batches
through a 3-stream pool with merged passes, every output bit-equal to the single-engine result.    N=300 python tools/soak_families.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.config import Blip2Arch, CocaArch  # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine, EnginePool  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip2_state_dict, procedural_coca_state_dict, synthetic_pixels  # noqa: E402

N = int(os.environ.get("N", 300))
bad = 0
for name, arch, sd, kw, gen_kw, B, coal in (
        ("coca beam 5 bf16", CocaArch.tiny(), procedural_coca_state_dict(CocaArch.tiny(), 1, eos_boost=2.0), dict(dtype="bf16", max_beams=5, max_len=CocaArch.tiny().seq_len),
         dict(num_beams=5, max_length=CocaArch.tiny().seq_len), 8, 32),
        ("blip2 f32s", Blip2Arch.small(), procedural_blip2_state_dict(Blip2Arch.small(), 2, eos_boost=0.3), dict(dtype="f32s", max_beams=1, max_len=20),
         dict(max_length=20), 8, 32),
        ("blip2 int8", Blip2Arch.small(), procedural_blip2_state_dict(Blip2Arch.small(), 2, eos_boost=0.3), dict(dtype="bf16", max_beams=1, max_len=20, weight_int8=True),
         dict(max_length=20), 8, 32)):
    one = CaptionerEngine(arch, max_batch=B, **kw)
    one.load_state_dict(sd)
    frames = [synthetic_pixels(B, arch.image_size, seed=3, first=i * B).cuda() for i in range(5)]
    want = [{k: v.clone() for k, v in one.generate(f, **gen_kw).items()} for f in frames]
    pool = EnginePool(arch, n=3, max_batch=coal, weights_of=one, **kw)
    outs = pool.generate_many([frames[i % 5] for i in range(N)], threads=True, coalesce_rows=coal, **gen_kw)
    torch.cuda.synchronize()
    assert isinstance(pool.last_coalesce, list) and max(len(g) for g in pool.last_coalesce) >= 3
    b0 = bad
    for i, o in enumerate(outs):
        w = want[i % 5]
        if not all(torch.equal(o[k], w[k]) for k in ("sequences", "lengths") if k in w) or ("sequences_scores" in w and not torch.equal(o["sequences_scores"], w["sequences_scores"])):
            bad += 1
    print(f"{name}: {N} batches in {len(pool.last_coalesce)} merged passes, mismatching {bad - b0}", flush=True)
    pool.close(); one.close()
print("SOAK OK" if bad == 0 else f"SOAK FAILED: {bad}")
sys.exit(0 if bad == 0 else 1)
