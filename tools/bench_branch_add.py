#!/usr/bin/env python
"""The ViT branch pair (proj or fc2 GEMM + the LayerNorm after it), two ways (GPU box only):
  old: GEMM -> delta (fp32);  add+LayerNorm kernel reads delta and X, writes X and the G8 / bf16 rows
  new: GEMM adds into X in place (gemm_pp.hip's residual epilogue);  LayerNorm reads X, writes the G8 / bf16 rows
Checks that X and the LayerNorm output have the same bits both ways, then times each kernel (HIP events, interleaved).
    python tools/bench_branch_add.py [--bf16]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd import _native  # noqa: E402

lib = _native.load_library()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
BF16 = "--bf16" in sys.argv
DT = 1 if BF16 else 2


def p(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def operand(x, w=False):
    if BF16:
        return x.to(torch.bfloat16)
    d = torch.empty_like(x)
    if w:
        assert lib.cap_op_convert_weight(DT, p(x), p(d), x.shape[0], x.shape[1], s) == 0
    else:
        assert lib.cap_op_convert(DT, p(x), p(d), x.numel(), s) == 0
    return d


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


M, D = 50432, 768
for name, K in (("proj", 768), ("fc2", 3072)):
    A = operand(torch.randn(M, K, device="cuda"))
    W = operand(torch.randn(D, K, device="cuda") / K ** 0.5, True)
    bias = torch.randn(D, device="cuda")
    g = torch.rand(D, device="cuda") + 0.5
    b = torch.randn(D, device="cuda")
    X0 = torch.randn(M, D, device="cuda")
    delta = torch.empty(M, D, device="cuda")
    esz = 2 if BF16 else 4
    ln_a = torch.empty(M * D * esz, device="cuda", dtype=torch.uint8)
    ln_b = torch.empty_like(ln_a)

    def gemm_old():
        assert lib.cap_op_gemm(DT, p(A), p(W), p(bias), p(None), p(delta), M, D, K, 0, 1, 0, s) == 0, lib.cap_last_error()

    def gemm_new(X):
        assert lib.cap_op_gemm(DT, p(A), p(W), p(bias), p(X), p(X), M, D, K, 0, 1, 0, s) == 0, lib.cap_last_error()

    def ln_old(X):
        assert lib.cap_op_reduce_layernorm(DT, p(delta), 1, p(None), p(X), p(g), p(b), 1e-5, p(ln_a), p(None), p(X), M, D, 0, s) == 0

    def ln_new(X):
        assert lib.cap_op_layernorm(DT, p(X), p(g), p(b), 1e-5, p(ln_b), p(None), M, D, s) == 0

    Xa, Xb = X0.clone(), X0.clone()
    gemm_old(); ln_old(Xa)
    gemm_new(Xb); ln_new(Xb)
    torch.cuda.synchronize()
    print(f"{name}: X identical {torch.equal(Xa.view(torch.int32), Xb.view(torch.int32))}  LayerNorm rows identical {torch.equal(ln_a, ln_b)}")
    X = X0.clone()
    for rep in range(3):
        X.zero_()                                          # (keeps the in-place accumulation finite)
        t = [timed(gemm_old), timed(lambda: ln_old(X)), timed(lambda: gemm_new(X)), timed(lambda: ln_new(X))]
        print(f"   old: GEMM {t[0]:6.1f} + add-LN {t[1]:6.1f} = {t[0] + t[1]:6.1f} us    in place: GEMM {t[2]:6.1f} + LN {t[3]:6.1f} = {t[2] + t[3]:6.1f} us", flush=True)
