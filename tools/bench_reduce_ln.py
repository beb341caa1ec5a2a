#!/usr/bin/env python
"""Split-K consumer (sum of slices + bias + residual, LayerNorm): block-per-row against wave-per-row kernel by row count and form
(BLIP: post-LN, fp32 row + operand row out; CoCa: pre-LN, the sum back into the residual stream in place + operand row), us per
launch, cold-ish (rotating buffers), HIP events over interleaved launches.
    python tools/bench_reduce_ln.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd import _native as N  # noqa: E402

lib = N.load_library()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
D, S, NB = 768, 4, 6
print("rows form dtype  block-per-row  wave-per-row (us per launch)")
for M in (256, 512, 640, 768, 896, 1024, 1280):
    for form in ("blip", "coca"):
        for tag, tdt, name in ((2, torch.float32, "g8"), (1, torch.bfloat16, "bf16")):
            parts = [torch.randn(S, M, D, device="cuda") for _ in range(NB)]
            xs = [torch.randn(M, D, device="cuda") for _ in range(NB)]
            outs = [torch.empty(M, D, dtype=tdt, device="cuda") for _ in range(NB)]
            outf = [torch.empty(M, D, device="cuda") for _ in range(NB)]
            bias, g, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda"), torch.randn(D, device="cuda")

            def run(rb, i):
                if form == "blip":
                    lib.cap_op_reduce_layernorm(tag, p(parts[i]), S, p(bias), p(xs[i]), p(g), p(b), C.c_float(1e-12), p(outs[i]), p(outf[i]), None, M, D, rb, st())
                else:
                    lib.cap_op_reduce_layernorm(tag, p(parts[i]), S, p(bias), p(xs[i]), p(g), p(b), C.c_float(1e-5), p(outs[i]), None, p(xs[i]), M, D, rb, st())
            res = {}
            for rb in (1, 0):
                for i in range(NB):
                    run(rb, i)
                torch.cuda.synchronize()
                # the launches replayed from a captured graph: a ctypes call costs ~7 us of host time, more than the kernel
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for k in range(120):
                        run(rb, k % NB)
                gr.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    gr.replay()
                e1.record()
                torch.cuda.synchronize()
                res[rb] = e0.elapsed_time(e1) / 600 * 1e3
            print(f"{M:5d} {form} {name:5s} {res[1]:8.2f} {res[0]:8.2f}", flush=True)
