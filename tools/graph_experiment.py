#!/usr/bin/env python
"""Experiment: capture one cap_generate (with its internal fork/join decode streams) into a HIP graph and replay it.
    CAP_DECODE_SLICES=2 python tools/graph_experiment.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd.config import BlipArch
from embodied_captioning_amd.engine import CaptionerEngine
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels

arch = BlipArch()
sd = procedural_blip_state_dict(arch, 0, eos_boost=9.0)
B, L = int(os.environ.get("GE_BATCH", 256)), 20
px = synthetic_pixels(B, arch.image_size, seed=0).cuda()
eng = CaptionerEngine(arch, dtype="bf16", max_batch=B, max_beams=1, max_len=L)
eng.load_state_dict(sd)
for _ in range(2):
    out = eng.generate(px, max_length=L)
torch.cuda.synchronize()
ref = out["sequences"].clone()
t0 = time.perf_counter()
for _ in range(5):
    out = eng.generate(px, max_length=L)
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 5 * 1e3, flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        gout = eng.generate(px, max_length=L)
torch.cuda.synchronize()
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) / 5 * 1e3, "same tokens:", bool((gout["sequences"] == ref).all()), flush=True)
