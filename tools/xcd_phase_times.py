#!/usr/bin/env python
"""Where does a launch of the persistent decode-step kernel spend its time?  CAP_XCD_DBG=1 makes one workgroup stamp the
100 MHz clock after every XCD-local barrier; this prints the mean duration of every phase kind over the layers of the last
decode step of a batch-256 generate.   CAP_XCD_DBG=1 python tools/xcd_phase_times.py [f32s|bf16]"""
import ctypes as C
import os
import sys

os.environ["CAP_XCD_DBG"] = "1"
os.environ["CAP_DECODE_XCD"] = "1"
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd.config import BlipArch  # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32s"
arch = BlipArch()
eng = CaptionerEngine(arch, dtype=dtype, max_batch=256, max_beams=1, max_len=20)
eng.load_state_dict(procedural_blip_state_dict(arch, 0, eos_boost=9.0))
px = synthetic_pixels(256, 224, seed=0).cuda()
for _ in range(2):
    eng.generate(px, max_length=20)
torch.cuda.synchronize()
buf = (C.c_longlong * 4096)()
n = eng.lib.cap_debug_xcd_times(eng._h, buf, 4096)
t = np.array(buf[:n], dtype=np.int64)
nb = 1 + 11 * arch.t_layers
t = t[: nb + 1]
d = np.diff(t) / 100.0          # us
names = ["qkv gemm", "self attn", "so gemm", "LN", "cq gemm", "cross attn", "co gemm", "LN", "f1 gemm", "f2 gemm", "LN"]
print(f"{dtype}: launch {d.sum():.1f} us; embed {d[0]:.1f} us")
per = d[1:].reshape(arch.t_layers, 11)
for i, nm in enumerate(names):
    print(f"  {nm:10s} mean {per[:, i].mean():7.2f} us   min {per[:, i].min():7.2f}  max {per[:, i].max():7.2f}")
print(f"  layer total mean {per.sum(1).mean():.1f} us")
# inside the GEMM phases (workgroup 0 of XCD 0, wave 0): barrier exit -> loads + MFMAs done -> reduced, stored, acknowledged
full = np.array(buf[:4096], dtype=np.int64)
for i, nm in enumerate(names):
    if "gemm" not in nm:
        continue
    rows = []
    for layer in range(arch.t_layers):
        k = 1 + layer * 11 + i           # index of the timestamp taken at the barrier BEFORE this phase
        a, b, e = full[2048 + (k + 1) * 2], full[2048 + (k + 1) * 2 + 1], t[k + 1]
        rows.append(((a - t[k]) / 100.0, (b - a) / 100.0, (e - b) / 100.0))
    r = np.array(rows).mean(0)
    print(f"  {nm:10s} loads+mfma {r[0]:6.2f} us | reduce+store+ack {r[1]:6.2f} us | barrier {r[2]:6.2f} us")
