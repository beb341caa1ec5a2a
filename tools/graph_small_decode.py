#!/usr/bin/env python
"""Does replaying one cap_generate as a captured HIP graph shorten the small-batch call?  (B = 1, 8; f32s; early exit off.)

    python tools/graph_small_decode.py
"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.config import BlipArch                                    # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine                             # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels   # noqa: E402


def med(fn, n=7):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return round(statistics.median(ts), 3)


def main():
    arch, L = BlipArch(), 20
    sd = procedural_blip_state_dict(arch, 0, eos_boost=9.0)
    for B in (1, 8):
        px = synthetic_pixels(B, arch.image_size, seed=0).cuda()
        eng = CaptionerEngine(arch, dtype="f32s", max_batch=B, max_beams=1, max_len=L)
        eng.load_state_dict(sd)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                ref = eng.generate(px, max_length=L)
            torch.cuda.synchronize()
            eager = med(lambda: eng.generate(px, max_length=L))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = eng.generate(px, max_length=L)
            g.replay()
            torch.cuda.synchronize()
            same = bool(torch.equal(out["sequences"], ref["sequences"]))
            graph = med(g.replay)
        print({"B": B, "eager_ms": eager, "graph_replay_ms": graph, "same_tokens": same})
        eng.close()


if __name__ == "__main__":
    main()
