#!/usr/bin/env python
"""Run one GEMM shape/tile a few times (for rocprofv3 --pmc passes).  python tools/gemm_one.py M N K tile [bf16|f32] [reps]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd import _native  # noqa: E402

lib = _native.load_library()
M, N, K, tile = (int(x) for x in sys.argv[1:5])
dt = sys.argv[5] if len(sys.argv) > 5 else "bf16"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
tag, tdt = (1, torch.bfloat16) if dt == "bf16" else (0, torch.float32)
A = torch.randn(M, K, device="cuda").to(tdt)
W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(tdt)
bias = torch.randn(N, device="cuda")
out = torch.zeros(M, N, device="cuda", dtype=tdt)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    rc = lib.cap_op_gemm(tag, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(0),
                         C.c_void_p(out.data_ptr()), M, N, K, 0, 0, tile, s)
    assert rc == 0
torch.cuda.synchronize()
print("done")
