#!/usr/bin/env python
"""Where does a frame's result depend on the batch it rides in?  Bitwise comparisons: the GEMM tile shapes against each
other on one problem, then encoder outputs and per-step logits of frames 0..7 alone vs inside a batch of 256."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd import _native  # noqa: E402
from embodied_captioning_amd.config import BlipArch  # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402

lib = _native.load_library()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
M, N, K = 1576, 2304, 768
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda")
for f32 in (0, 1):
    outs = {}
    for tile in (1, 2, 3, 4, 5):
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
        rc = lib.cap_op_gemm(1, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(0),
                             C.c_void_p(out.data_ptr()), M, N, K, 0, f32, tile, s)
        assert rc == 0
        torch.cuda.synchronize()
        outs[tile] = out.float().cpu()
    for t in (2, 3, 4, 5):
        d = (outs[t] != outs[1]).sum().item()
        print(f"gemm f32out={f32}: tile{t} vs tile1: {d} differing elements, max |d| {(outs[t] - outs[1]).abs().max().item():.3g}")

arch = BlipArch()
sd = procedural_blip_state_dict(arch, seed=0, eos_boost=9.0)
eng = CaptionerEngine(arch, "bf16", 256, 1, 20)
eng.load_state_dict(sd)
px = synthetic_pixels(256, arch.image_size, seed=0).cuda()
e256 = eng.encode(px)[:8].cpu()
e8 = eng.encode(px[:8]).cpu()
print("encoder: differing elements", (e256 != e8).sum().item(), "max |d|", (e256 - e8).abs().max().item())
g256 = eng.generate(px, max_length=20, output_logits=True)
g8 = eng.generate(px[:8], max_length=20, output_logits=True)
l256 = g256["logits"][:, :8].cpu()
l8 = g8["logits"].cpu()
for t in range(l8.shape[0]):
    print(f"step {t}: differing logits {(l256[t] != l8[t]).sum().item()}  max |d| {(l256[t] - l8[t]).abs().max().item():.3g}")
    if (l256[t] != l8[t]).any():
        break
