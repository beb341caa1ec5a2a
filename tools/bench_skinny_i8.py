#!/usr/bin/env python
"""The int8 weight-stream GEMM (and the bf16 one) per launch by row count, OPT-2.7b shapes, launches replayed from a captured graph
over rotating weight copies (cold weights: 8 copies x 26-52 MB).    python tools/bench_skinny_i8.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd import _native as N  # noqa: E402

lib = N.load_library()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
NB = 8
print("shape           rows   int8 us   bf16 us   (finished form: bias + ReLU -> bf16; partial form for K = 10240)")
for name, Nn, K, fin in (("qkv", 7680, 2560, 1), ("o", 2560, 2560, 0), ("fc1", 10240, 2560, 1), ("fc2", 2560, 10240, 0)):
    ws = [(torch.randn(Nn, K, device="cuda") * 0.02) for _ in range(2)]
    packs = []
    for i in range(NB):
        pk = torch.empty(Nn * K, dtype=torch.uint8, device="cuda"); sc = torch.empty(Nn, device="cuda")
        lib.cap_op_quant_i8_pack(p(ws[i % 2]), p(pk), p(sc), Nn, K, st())
        packs.append((pk, sc))
    wb = [ws[i % 2].bfloat16() for i in range(NB)]
    bias = torch.randn(Nn, device="cuda")
    S8 = lib.cap_op_gemm_skinny_i8_slices(Nn, K, fin)
    Sb = lib.cap_op_gemm_skinny_slices(Nn, K, fin)
    for M in (1, 16, 32, 64, 264, 1056):
        a = torch.randn(M, K, device="cuda").bfloat16()
        out = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
        part = torch.empty(max(S8, Sb, 1), M, Nn, device="cuda")
        res = []
        for kind in ("i8", "bf16"):
            def run(i):
                if kind == "i8":
                    lib.cap_op_gemm_skinny_i8(p(a), p(packs[i][0]), p(packs[i][1]), p(bias) if fin else None, 2 if fin else 0, p(out) if fin else None,
                                              None if fin else p(part), M, Nn, K, st())
                else:
                    lib.cap_op_gemm_skinny(p(a), p(wb[i]), p(bias) if fin else None, 2 if fin else 0, p(out) if fin else None, None if fin else p(part), M, Nn, K, st())
            for i in range(NB):
                run(i)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for k in range(64):
                    run(k % NB)
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                gr.replay()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 192 * 1e3)
        print(f"{name:4s} {Nn:5d}x{K:5d} {M:5d} {res[0]:9.1f} {res[1]:9.1f}", flush=True)
