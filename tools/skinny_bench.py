"""Time the weight-streaming decode GEMM (cap_op_gemm_skinny) against the tiled split-K path at the OPT-2.7b shapes.
Weights rotate over enough copies to exceed the 256 MB MALL, as consecutive layers do in the real decode step."""
import ctypes as C
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd import _native  # noqa: E402

lib = _native.load_library()
p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("SK_M", 32))
shapes = [("qkv", 7680, 2560), ("o", 2560, 2560), ("f1", 10240, 2560), ("f2", 2560, 10240)]
for name, N, K in shapes:
    copies = max(2, int(600e6 // (N * K * 2)))
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(copies)]
    A = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    part = torch.zeros(16, M, N, device="cuda")
    fin = lib.cap_op_gemm_skinny_slices(N, K, 1) == 1 and name in ("qkv", "f1")
    S = lib.cap_op_gemm_skinny_slices(N, K, 1 if fin else 0)

    def skinny(i):
        lib.cap_op_gemm_skinny(p(A), p(Ws[i % copies]), p(bias) if fin else None, 0, p(out) if fin else None, None if fin else p(part), M, N, K, st())

    def tiled(i):
        lib.cap_op_gemm(1, p(A), p(Ws[i % copies]), p(bias), None, p(out), M, N, K, 0, 0, 2, st())

    res = {}
    for tag, fn in (("skinny", skinny), ("tiled64", tiled)):
        for i in range(10):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        res[tag] = us
    gb = N * K * 2 / 1e9
    print(f"{name:4s} N={N:6d} K={K:6d} S={S}  skinny {res['skinny']:7.2f} us ({gb / res['skinny'] * 1e3:5.2f} TB/s)   tiled64 "
          f"{res['tiled64']:7.2f} us ({gb / res['tiled64'] * 1e3:5.2f} TB/s)", flush=True)
