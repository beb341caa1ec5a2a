#!/usr/bin/env python
"""Race screen for the LDS-DMA GEMM kernels (raw s_barrier + counted vmcnt): every repeat of every shape must reproduce, bit
for bit, the result of the register-staged generic kernel (tile 1), which shares their accumulation order.  An LDS read that
overtakes its DMA would show up as a rare wrong tile.  python tools/gemm_race_screen.py [repeats]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd import _native  # noqa: E402

lib = _native.load_library()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(1)
bad_total = 0
for (M, N, K, f32, gelu) in [(50432, 2304, 768, 0, 0), (50432, 768, 768, 1, 0), (50432, 3072, 768, 0, 1), (50432, 768, 3072, 1, 0),
                             (1576, 2304, 768, 0, 0), (73856, 1024, 1024, 1, 0), (9999, 516, 192, 0, 1), (4096, 4096, 4096, 0, 0)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    dt = torch.float32 if f32 else torch.bfloat16

    def run(tile, out):
        rc = lib.cap_op_gemm(1, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(0),
                             C.c_void_p(out.data_ptr()), M, N, K, gelu, f32, tile, s)
        assert rc == 0, lib.cap_last_error()
    ref = torch.empty(M, N, device="cuda", dtype=dt)
    run(1, ref)
    for tile in (12, 14):
        bad = 0
        out = torch.empty(M, N, device="cuda", dtype=dt)
        for _ in range(reps):
            out.fill_(float("nan"))
            run(tile, out)
            if not torch.equal(out, ref):
                bad += 1
        bad_total += bad
        print(f"M={M} N={N} K={K} f32out={f32} gelu={gelu} tile{tile}: {bad} of {reps} repeats differ from the generic kernel", flush=True)
print("RACE SCREEN", "CLEAN" if bad_total == 0 else f"FAILED ({bad_total})")
