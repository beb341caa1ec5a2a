#!/usr/bin/env python
"""Steady-state check of the GEMM main loops on square problems.  python tools/bench_gemm_sq.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd import _native  # noqa: E402

lib = _native.load_library()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (50432, 2304, 3072), (50432, 2304, 768), (50432, 2304, 256)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    line = f"M={M} N={N} K={K}:"
    for tile in (3, 5):
        def run():
            rc = lib.cap_op_gemm(1, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(0), C.c_void_p(0),
                                 C.c_void_p(out.data_ptr()), M, N, K, 0, 0, tile, s)
            assert rc == 0, lib.cap_last_error()
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        line += f"  tile{tile}: {us:9.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF"
    print(line, flush=True)
