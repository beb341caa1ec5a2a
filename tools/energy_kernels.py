#!/usr/bin/env python
"""Socket power while ONE kernel of the split-mode path runs in a loop (GPU box; bench.PowerSampler): what a millisecond of
each kernel costs in joules.    python tools/energy_kernels.py"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import PowerSampler  # noqa: E402
from embodied_captioning_amd import _native  # noqa: E402

lib = _native.load_library()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SPLIT, F32 = 2, 0
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def g8(x, w=False):
    d = torch.empty_like(x)
    if w:
        assert lib.cap_op_convert_weight(SPLIT, P(x), P(d), x.shape[0], x.shape[1], s) == 0
    else:
        assert lib.cap_op_convert(SPLIT, P(x), P(d), x.numel(), s) == 0
    return d


def loop(name, fn, seconds=2.5):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(); torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-5)
    n = max(10, int(seconds / one))
    with PowerSampler(0) as ps:
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    r = ps.result(n, dt)
    print(f"{name:44s}: {1e6 * dt / n:9.1f} us per launch, {r['watts_mean']:7.1f} W -> {r['watts_mean'] * dt / n * 1e3:8.3f} mJ per launch", flush=True)


# idle-but-busy reference: an empty-ish kernel chain
x = torch.zeros(1024, device="cuda")
loop("tiny torch kernel chain (launch-bound)", lambda: x.add_(1.0))
# encoder GEMM (qkv shape)
M, N, K = 50432, 2304, 768
A = g8(torch.randn(M, K, device="cuda")); W = g8(torch.randn(N, K, device="cuda") / K ** 0.5, True)
out = torch.zeros(M, N, device="cuda")
loop("encoder GEMM qkv (split, 256x256)", lambda: lib.cap_op_gemm(SPLIT, P(A), P(W), None, None, P(out), M, N, K, 0, 0, 3, s))
del out
# decode GEMM 256 x 768 x 768
Md = 256
Ad = g8(torch.randn(Md, K, device="cuda")); Wd = g8(torch.randn(768, K, device="cuda") / K ** 0.5, True)
od = torch.zeros(Md, 768, device="cuda")
loop("decode GEMM 256x768x768 (split, 64x64)", lambda: lib.cap_op_gemm(SPLIT, P(Ad), P(Wd), None, None, P(od), Md, 768, K, 0, 1, 0, s))
# cross attention: 256 rows x 12 heads over 197 fp32 keys
H, NT = 12, 197
q = torch.randn(256, H * 64, device="cuda")
Kc = torch.randn(256, H, NT, 64, device="cuda"); Vc = torch.randn(256, H, NT, 64, device="cuda")
oc = torch.zeros(256, H * 64, device="cuda")
loop("cross-attention 256 rows x 197 fp32 keys", lambda: lib.cap_op_decode_attention(F32, P(q), P(Kc), P(Vc), None, 0, 1, NT, NT, P(oc), 256, H, 0, s))
# bandwidth reference: device copy of 620 MB
a = torch.empty(155 * 1024 * 1024 // 4, device="cuda"); b = torch.empty_like(a)
loop("torch copy 155 MB -> 155 MB", lambda: b.copy_(a))
