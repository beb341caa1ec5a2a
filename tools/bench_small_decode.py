#!/usr/bin/env python
"""Per-kernel timing of the small-batch decode path (csrc/decode_small.hip) next to the batch path at the same row count.

    python tools/bench_small_decode.py [--dtype f32s] [--batches 1,8,16] [--reps 5]

For every batch size: wall time of one cap_generate (median), then the library's own per-tag HIP-event report for the decode
tags (launches, average microseconds per launch) on both paths.  BLIP-base shapes, procedural weights, max_length 20.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.config import BlipArch                                    # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine                             # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32s")
    ap.add_argument("--batches", default="1,8,16")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--model", default="blip", choices=["blip", "coca"], help="coca: ViT-L/14, 12 + 12 text layers, seq_len 30 (parity unpinned)")
    ap.add_argument("--stamps", action="store_true", help="experiments build (python -m embodied_captioning_amd.build --experiments): "
                    "cycle stamps of workgroup 0 of the last cross / GEMM launch")
    a = ap.parse_args()
    if a.model == "coca":
        from embodied_captioning_amd.config import CocaArch
        from embodied_captioning_amd.weights import procedural_coca_state_dict
        arch = CocaArch()
        L = arch.seq_len
        sd = procedural_coca_state_dict(arch, 0)
    else:
        arch, L = BlipArch(), 20
        sd = procedural_blip_state_dict(arch, 0, eos_boost=9.0)
    out = {}
    for B in [int(x) for x in a.batches.split(",")]:
        px = synthetic_pixels(B, arch.image_size, seed=0).cuda()
        eng = CaptionerEngine(arch, dtype=a.dtype, max_batch=B, max_beams=1, max_len=L)
        eng.load_state_dict(sd)
        rec = {}
        for path in ("small", "batch"):
            eng.set_decode_path(path)
            for _ in range(2):
                eng.generate(px, max_length=L)
            torch.cuda.synchronize()
            ts = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                eng.generate(px, max_length=L)
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - t0))
            eng.profile(True)
            eng.generate(px, max_length=L)
            rep = eng.profile_report()
            eng.profile(False)
            tags = {t: {"n": r["launches"], "us": round(1e3 * r["ms"] / r["launches"], 2), "ms": round(r["ms"], 3)}
                    for t, r in sorted(rep.items()) if t.startswith(("dec_", "coca_")) and "gemm_crosskv" not in t or t == "greedy_select"}
            if a.stamps and path == "small":
                import ctypes as C
                buf = (C.c_ulonglong * 64)()
                fn = eng.lib.cap_debug_small_stamps
                fn.argtypes = [C.c_void_p]
                fn(buf)
                for k, name in ((0, "cross"), (1, "gemm")):
                    cyc = [buf[k * 32 + 2 * i] for i in range(16)]
                    wal = [buf[k * 32 + 2 * i + 1] for i in range(16)]
                    tags.setdefault("_stamps", {})[name] = {"cycles_from_0": [int(c - cyc[0]) if c else None for c in cyc[:14]],
                                                            "wall_ns_from_0": [int(w - wal[0]) * 10 if w else None for w in wal[:14]]}
            rec[path] = {"generate_ms": round(statistics.median(ts), 3), "decode_kernel_ms": round(sum(v["ms"] for k, v in tags.items() if k != "_stamps"), 3),
                         "tags": tags}
        out[str(B)] = rec
        eng.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
