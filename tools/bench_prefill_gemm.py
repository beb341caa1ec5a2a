#!/usr/bin/env python
"""OPT prefill projections with the residual operand (GPU box): 1056 rows (32 frames x 33 positions) against [2560, K] weights,
fp32 output added in place - which tile shape should launch_gemm's auto rule pick?   python tools/bench_prefill_gemm.py [--bf16]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd import _native
lib = _native.load_library()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
BF16 = "--bf16" in sys.argv
DT = 1 if BF16 else 2
p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)


def operand(x, w=False):
    if BF16:
        return x.to(torch.bfloat16)
    d = torch.empty_like(x)
    if w:
        assert lib.cap_op_convert_weight(DT, p(x), p(d), x.shape[0], x.shape[1], s) == 0
    else:
        assert lib.cap_op_convert(DT, p(x), p(d), x.numel(), s) == 0
    return d


for M, N, K in ((1056, 2560, 2560), (1056, 2560, 10240), (4224, 2560, 2560), (4224, 2560, 10240), (1024, 768, 768), (1024, 768, 3072),
                (1056, 2560, 768), (2048, 1024, 1024), (1576, 2304, 768), (1576, 3072, 768), (1576, 768, 3072), (788, 3072, 768)):
    A = operand(torch.randn(M, K, device="cuda"))
    W = operand(torch.randn(N, K, device="cuda") / K ** 0.5, True)
    bias = torch.randn(N, device="cuda")
    X0 = torch.randn(M, N, device="cuda")
    line, ref = f"M={M} N={N} K={K}:", None
    for tile in (0, 1, 2, 3, 4):
        X = X0.clone()
        rc = lib.cap_op_gemm(DT, p(A), p(W), p(bias), p(X), p(X), M, N, K, 0, 1, tile, s)
        if rc != 0:
            line += f"  t{tile}: refused"
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = X.clone()
        same = torch.equal(ref.view(torch.int32), X.view(torch.int32))
        X.zero_()
        for _ in range(3):
            lib.cap_op_gemm(DT, p(A), p(W), p(bias), p(X), p(X), M, N, K, 0, 1, tile, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.cap_op_gemm(DT, p(A), p(W), p(bias), p(X), p(X), M, N, K, 0, 1, tile, s)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        line += f"  t{tile}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF{'' if same else ' DIFF'}"
    print(line, flush=True)
