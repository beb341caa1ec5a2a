#!/usr/bin/env python
"""Exploration companion of tests/test_envelope_gpu.py: runs every weight family of tests/_families.py through the split mode
and the CPU oracle and prints what held (greedy rows identical, smallest oracle top-2 margin, worst top-8 logit error,
beam-3 scores, clamp count).    python tests/explore_parity_envelope.py [n_frames] [family ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _families import FAMILIES, ILL_CONDITIONED, beyond_fp16  # noqa: E402
from embodied_captioning_amd.config import BlipArch  # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402
from oracle import blip_ref as R  # noqa: E402  (checker)

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12
names = [x for x in sys.argv[1:] if not x.isdigit()] or list(FAMILIES) + list(ILL_CONDITIONED) + ["beyond_fp16"]
arch, L = BlipArch(), 20
torch.set_num_threads(min(16, os.cpu_count() or 1))
base = procedural_blip_state_dict(arch, 0, eos_boost=5.0)
for name in names:
    sd = beyond_fp16(base, arch) if name == "beyond_fp16" else {**FAMILIES, **ILL_CONDITIONED}[name](base, arch)
    px = synthetic_pixels(n, arch.image_size, seed=21)
    t0 = time.time()
    ref = R.greedy_generate(sd, arch, px, L)
    rseq = np.full((n, L), arch.pad, dtype=np.int64)
    rseq[:, : ref["sequences"].shape[1]] = ref["sequences"].numpy()
    lg = torch.stack(ref["logits"], 0)                      # [steps, n, V]
    t2 = torch.topk(lg, 2, dim=-1).values
    margin = float((t2[..., 0] - t2[..., 1]).min())
    for dtype in ("f32s", "f32"):
        eng = CaptionerEngine(arch, dtype=dtype, max_batch=n, max_beams=3, max_len=L)
        eng.load_state_dict(sd)
        eng.saturations(reset=True)
        out = eng.generate(px.cuda(), max_length=L, output_logits=True)
        seq = out["sequences"].cpu().numpy()
        same = (seq == rseq).all(axis=1)
        top = torch.topk(lg, 8, dim=-1)
        ours = torch.gather(out["logits"][: lg.shape[0]].cpu(), 2, top.indices)
        # compare logits only while a row is still on the oracle's path and not finished
        steps = lg.shape[0]
        alive = np.ones((steps, n), dtype=bool)
        for s in range(steps):
            for r in range(n):
                if s + 1 >= (rseq[r] != arch.pad).sum() or not (seq[r, : s + 1] == rseq[r, : s + 1]).all():
                    alive[s, r] = False
        err = float((ours - top.values).abs()[torch.from_numpy(alive)].max()) if alive.any() else float("nan")
        emb_err = float((eng.encode(px.cuda()).cpu() - ref["image_embeds"]).abs().max())
        sat = eng.saturations(reset=True)
        eng.close()
        print(f"{name:24s} {dtype:5s}: greedy {int(same.sum())}/{n} rows identical, min oracle margin {margin:.2e}, top-8 logit err {err:.2e}, "
              f"image_embeds err {emb_err:.2e} (max |embed| {float(ref['image_embeds'].abs().max()):.1f}), clamped {sat}; {time.time() - t0:.0f}s", flush=True)
