"""Determinism soak: N batches through the 3-stream pool (threads on and off, early-exit polling on and off, and - the headline's
mode - merged into passes of up to 1024 rows with the compacted greedy loop); every output must equal the single-engine result of
the same frames bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.config import BlipArch  # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine, EnginePool  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402

B, L, N = 256, 20, int(os.environ.get("N", 120))
DTYPE = os.environ.get("DTYPE", "f32s")            # the default mode; DTYPE=bf16 for the bf16 kernels
arch = BlipArch()
sd = procedural_blip_state_dict(arch, 0, eos_boost=9.0)
one = CaptionerEngine(arch, dtype=DTYPE, max_batch=B, max_beams=1, max_len=L)
one.load_state_dict(sd)
frames = [synthetic_pixels(B, arch.image_size, seed=5, first=i * B).cuda() for i in range(4)]
want = [{k: v.clone() for k, v in one.generate(f, max_length=L).items()} for f in frames]
pool = EnginePool(arch, n=3, dtype=DTYPE, max_batch=B, max_beams=1, max_len=L, weights_of=one)
bad = 0
for threads, poll in ((False, 0), (True, 0), (True, 4)):
    pool.set_early_exit(poll)
    outs = pool.generate_many([frames[i % 4] for i in range(N)], threads=threads, max_length=L)
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        w = want[i % 4]
        if not (torch.equal(o["sequences"], w["sequences"]) and torch.equal(o["lengths"], w["lengths"])):
            bad += 1
    print(f"threads={threads} poll={poll}: {N} batches, mismatching so far {bad}", flush=True)
pool.close()
# the headline's mode: arenas of 1024 rows, dynamic batching (passes of 4 / 3 batches), row compaction on (the default)
big = EnginePool(arch, n=3, dtype=DTYPE, max_batch=1024, max_beams=1, max_len=L, weights_of=one)
for threads in (True, False):
    outs = big.generate_many([frames[i % 4] for i in range(N)], threads=threads, coalesce_rows=1024, max_length=L)
    torch.cuda.synchronize()
    plan = big.last_coalesce
    assert isinstance(plan, list) and max(len(g) for g in plan) >= 3, plan
    for i, o in enumerate(outs):
        w = want[i % 4]
        if not (torch.equal(o["sequences"], w["sequences"]) and torch.equal(o["lengths"], w["lengths"])):
            bad += 1
    print(f"merged passes (coalesce_rows=1024, {len(plan)} passes) threads={threads}: {N} batches, mismatching so far {bad}", flush=True)
big.close()
print("SOAK OK" if bad == 0 else f"SOAK FAILED: {bad}")
sys.exit(0 if bad == 0 else 1)
