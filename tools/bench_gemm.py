#!/usr/bin/env python
"""Micro-benchmark of the GEMM kernel over the captioner's shapes (GPU box only): TFLOP/s per tile config.
    python tools/bench_gemm.py [bf16|f32]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd import _native  # noqa: E402

lib = _native.load_library()
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
tag, tdt = (1, torch.bfloat16) if dt == "bf16" else (0, torch.float32)
SHAPES = [("qkv", 50432, 2304, 768, 0, 0), ("proj", 50432, 768, 768, 0, 1), ("fc1", 50432, 3072, 768, 1, 0),
          ("fc2", 50432, 768, 3072, 0, 1), ("vocab", 256, 30524, 768, 0, 1), ("dec768", 256, 768, 768, 0, 0),
          ("dec_f1", 256, 3072, 768, 1, 0), ("dec_qkv", 256, 2304, 768, 0, 0)]
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, M, N, K, gelu, f32out in SHAPES:
    A = torch.randn(M, K, device="cuda").to(tdt)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(tdt)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32out else tdt)
    resid = None   # the branch GEMMs write a delta; the residual add lives in the add+LayerNorm kernel
    line = f"{name:8s} M={M} N={N} K={K}:"
    for tile in (1, 2, 3, 4, 16):
        if (tile == 2 and M > 1000) or (tile in (4, 16) and M < 1000):
            continue
        def run():
            rc = lib.cap_op_gemm(tag, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(bias.data_ptr()),
                                 C.c_void_p(resid.data_ptr() if resid is not None else 0), C.c_void_p(out.data_ptr()),
                                 M, N, K, gelu, f32out, tile, s)
            assert rc == 0, lib.cap_last_error()
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        line += f"  tile{tile}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF"
    print(line, flush=True)
