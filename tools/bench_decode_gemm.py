#!/usr/bin/env python
"""Micro-benchmark of the decode-step GEMMs (GPU box only): the six projections of a BLIP text-decoder layer at R rows, each
against weights that are COLD - the launches cycle over enough distinct weight buffers (> 256 MiB, the Infinity Cache) that a
weight is never resident when its turn comes, as in a real generate (400 MB of decoder weights per step).
    python tools/bench_decode_gemm.py [--rows 256] [--dtype f32s|bf16] [--tiles 2]
Prints per shape: us per launch (HIP events over the loop), algorithmic bytes (W + A + output / slabs) and GB/s."""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=256)
ap.add_argument("--dtype", default="f32s")
ap.add_argument("--tiles", default="2,6", help="tile ids to compare (2 = register-staged 64x64, 6 = rows kernel)")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
lib = _native.load_library()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
DT = {"f32": 0, "bf16": 1, "f32s": 2}[a.dtype]
esz = 2 if a.dtype == "bf16" else 4
tdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32


def operand(x, weight):
    if a.dtype == "f32":
        return x
    if a.dtype == "bf16":
        return x.to(torch.bfloat16)
    d = torch.empty_like(x)
    if weight:
        assert lib.cap_op_convert_weight(DT, C.c_void_p(x.data_ptr()), C.c_void_p(d.data_ptr()), x.shape[0], x.shape[1], s) == 0
    else:
        assert lib.cap_op_convert(DT, C.c_void_p(x.data_ptr()), C.c_void_p(d.data_ptr()), x.numel(), s) == 0
    return d


R = a.rows
# name, N, K, {tile: split-K slices} (0 = finished GEMM with bias + GELU -> operand type); the slices are what the captioner's
# plan picks for that kernel (captioner.hip decode_splitk)
SHAPES = [("qkv", 2304, 768, {2: 4, 6: 1}), ("so/cq/co", 768, 768, {2: 4, 6: 4 if a.dtype != "bf16" else 3}), ("f1", 3072, 768, {2: 0, 6: 0}),
          ("f2", 768, 3072, {2: 4, 6: 4})]
for name, N, K, plan in SHAPES:
    S = max(plan.values())
    nbuf = max(2, int(420e6 // (N * K * esz)))
    Ws = [operand(torch.randn(N, K, device="cuda") / K ** 0.5, True) for _ in range(nbuf)]
    A = operand(torch.randn(R, K, device="cuda"), False)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(max(S, 1) * R * N, device="cuda", dtype=torch.float32)
    line = f"{name:9s} R={R} N={N} K={K}:"
    for tile in [int(t) for t in a.tiles.split(",")]:
        S = plan.get(tile, plan[2])
        alg = N * K * esz + R * K * esz + (S * R * N * 4 if S else R * N * esz)
        def run(W):
            if S:
                rc = lib.cap_op_gemm_partial(DT, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()), R, N, K, S, tile, s)
            else:
                rc = lib.cap_op_gemm(DT, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(0),
                                     C.c_void_p(out.data_ptr()), R, N, K, 1, 0, tile, s)
            assert rc == 0, lib.cap_last_error()
        for W in Ws[:8]:
            run(W)
        best = 1e9
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for W in Ws:
                run(W)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / len(Ws))
        line += f"  tile{tile} S={S}: {best:6.2f} us = {alg / best / 1e3:5.0f} GB/s of {alg / 1e6:5.2f} MB;"
    print(line, flush=True)
    del Ws
