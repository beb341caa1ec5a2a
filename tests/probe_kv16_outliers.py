"""Probe (GPU): how far do the logits of the split mode move when the cross-attention K/V cache is KV16 instead of fp32 rows, as a
function of the OUTLIER FACTOR of two dimensions per key / value head (tests/_families.cross_kv_outliers) and of the number of
image tokens (197 at 224 px, 577 at 384 px)?  Decides the threshold of the load-time guard (weights.cross_kv_head_spread).

    python tests/probe_kv16_outliers.py > profiles/r04_kv16_outlier_probe.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _families import cross_kv_outliers                                                   # noqa: E402
from embodied_captioning_amd.config import BlipArch                                       # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine                                # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402


def main():
    L, B = 20, 4
    print("# max |logit(KV16 cache) - logit(fp32 rows)| over the live steps of 4 frames, split mode, BLIP-base, procedural weights")
    print("# image px | tokens | outlier factor | max logit difference | tokens identical")
    for size in (224, 384):
        arch = BlipArch()
        arch.image_size = size
        sd0 = procedural_blip_state_dict(arch, 0, eos_boost=5.0)
        px = synthetic_pixels(B, size, seed=31).cuda()
        for factor in (1, 4, 8, 12, 16, 30):
            sd = cross_kv_outliers(sd0, arch, 8, factor=float(factor)) if factor > 1 else sd0
            outs = []
            for cc in ("auto", "fp32"):
                eng = CaptionerEngine(arch, dtype="f32s", max_batch=B, max_beams=1, max_len=L, cross_cache=cc)
                eng._skip_kv16_guard = True
                eng.load_state_dict(sd)
                outs.append(eng.generate(px, max_length=L, output_logits=True))
                eng.close()
            a, b = outs
            seq = a["sequences"].cpu()
            err = 0.0
            for r in range(B):
                row = seq[r, 1:].tolist()
                n = row.index(arch.eos) + 1 if arch.eos in row else L - 1
                err = max(err, float((a["logits"][:n, r] - b["logits"][:n, r]).abs().max()))
            print(f"{size} | {arch.n_tokens} | {factor} | {err:.2e} | {bool(torch.equal(a['sequences'], b['sequences']))}", flush=True)


if __name__ == "__main__":
    main()
