#!/usr/bin/env python
"""The vocabulary GEMM of a decode step (GPU box): 256 rows x 30524 words x 768, fp32 logits, cold weights (the 94 MB table is
evicted between calls by a 512 MB copy) - per tile shape.   python tools/bench_vocab_gemm.py [--bf16]"""
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
    f = lib.cap_op_convert_weight if w else None
    if w:
        assert lib.cap_op_convert_weight(DT, p(x), p(d), x.shape[0], x.shape[1], s) == 0
    else:
        assert lib.cap_op_convert(DT, p(x), p(d), x.numel(), s) == 0
    return d


M, N, K = 256, 30524, 768
A = operand(torch.randn(M, K, device="cuda"))
W = operand(torch.randn(N, K, device="cuda") / K ** 0.5, True)
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
junk = torch.empty(128 << 20, device="cuda")
ref = None
for tile in (0, 1, 2, 6, 3):
    if lib.cap_op_gemm(DT, p(A), p(W), p(bias), p(None), p(out), M, N, K, 0, 1, tile, s) != 0:
        print(f"tile {tile}: refused")
        continue
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    same = torch.equal(ref.view(torch.int32), out.view(torch.int32))
    tot = 0.0
    for _ in range(10):
        junk.add_(1.0)                                   # 1 GB of traffic: the table leaves the caches
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.cap_op_gemm(DT, p(A), p(W), p(bias), p(None), p(out), M, N, K, 0, 1, tile, s)
        e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    us = tot / 10
    by = N * K * (2 if BF16 else 4) + M * N * 4 + M * K * (2 if BF16 else 4)
    print(f"tile {tile}: {us:6.1f} us  {by / us / 1e3:6.0f} GB/s of {by / 1e6:.0f} MB  identical {same}", flush=True)
