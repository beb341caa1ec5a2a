#!/usr/bin/env python
"""Per-tag kernel times of one 256-frame generate on each decode path (GPU box only): the one-launch-per-operation batch
kernels against the fused variants of decode_tile.hip.
    python tools/bench_tile_decode.py [--batch 256] [--dtype f32s] [--paths batch,tile]
HIP-event brackets per launch (engine.profile): 1-3 us more per launch than a kernel trace shows, alike on every path."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd.config import BlipArch                                    # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine                             # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dtype", default="f32s")
ap.add_argument("--paths", default="batch,tile")
ap.add_argument("--max-length", type=int, default=20)
a = ap.parse_args()
arch = BlipArch()
sd = procedural_blip_state_dict(arch, 0, eos_boost=9.0)
px = synthetic_pixels(a.batch, arch.image_size, seed=0).cuda()
L = a.max_length
eng = CaptionerEngine(arch, dtype=a.dtype, max_batch=a.batch, max_beams=1, max_len=L)
eng.load_state_dict(sd)
ref = None
for path in a.paths.split(","):
    eng.set_decode_path(path)
    for _ in range(2):
        out = eng.generate(px, max_length=L)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        out = eng.generate(px, max_length=L)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    if ref is None:
        ref = out["sequences"].clone()
    same = bool(torch.equal(ref, out["sequences"]))
    eng.profile(True)
    eng.generate(px, max_length=L)
    rep = eng.profile_report()
    eng.profile(False)
    dec = {k: v for k, v in rep.items() if k.startswith("dec_") or k == "greedy_select"}
    tot = sum(v["ms"] for v in dec.values())
    print(f"== {path}: generate {min(ts):.2f} ms (min of 5), decode-side tagged kernels {tot:.2f} ms, tokens equal to the first path: {same}")
    for k, v in sorted(dec.items(), key=lambda kv: -kv[1]["ms"]):
        print(f"   {k:16s} {v['launches']:5d} x {1e3 * v['ms'] / v['launches']:7.2f} us = {v['ms']:7.3f} ms")
eng.close()
