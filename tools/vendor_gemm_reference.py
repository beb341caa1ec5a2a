#!/usr/bin/env python
"""CONTEXT ONLY (nothing in the product calls a vendor GEMM): what torch.mm - hipBLASLt / rocBLAS, AMD's own tuned kernels - reaches
on this board on the ViT-B/16 Linear shapes of a 256-frame batch, fp16 and bf16 operands (ONE product per MAC), random normal data.
The split mode's encoder GEMM executes three fp16 products per MAC: its `executed_tflops` (bench line) is the figure to set beside
these.    python tools/vendor_gemm_reference.py [frames]"""
import sys
import time

import torch

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = frames * 197
shapes = {"qkv": (2304, 768), "proj": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}
print(f"M = {M} rows ({frames} frames); TFLOP/s as 2MNK / t, median of 5 x 20 launches")
for dt in (torch.float16, torch.bfloat16):
    for name, (N, K) in shapes.items():
        a = torch.randn(M, K, device="cuda", dtype=dt)
        w = (torch.randn(N, K, device="cuda") * 0.03).to(dt)
        for _ in range(5):
            c = torch.mm(a, w.t())
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                c = torch.mm(a, w.t())
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        t = sorted(ts)[2]
        print(f"{str(dt).split('.')[-1]:9s} {name:5s} N={N:5d} K={K:5d}  {t * 1e3:8.1f} us  {2.0 * M * N * K / (t * 1e-3) / 1e12:7.1f} TFLOP/s", flush=True)
    # sustained: the four in rotation for ~2 s (the clock settles under load)
    ops = [(torch.randn(M, K, device="cuda", dtype=dt), (torch.randn(N, K, device="cuda") * 0.03).to(dt)) for N, K in shapes.values()]
    fl = sum(2.0 * M * N * K for N, K in shapes.values())
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 2.0:
        for a, w in ops:
            torch.mm(a, w.t())
        n += 1
        if n % 20 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dtm = time.perf_counter() - t0
    print(f"{str(dt).split('.')[-1]:9s} the four in rotation for {dtm:.1f} s: {fl * n / dtm / 1e12:7.1f} TFLOP/s sustained", flush=True)
