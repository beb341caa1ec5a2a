#!/usr/bin/env python
"""A/B of one encoder GEMM shape between two builds of the library (GPU box only): the in-tree one and
embodied_captioning_amd/lib/libcaptioner_old.so (a copy of an earlier build), interleaved, outputs compared bit for bit.
    python tools/ab_gemm_libs.py [--bf16]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd import _native
new = _native.load_library()
OLD = os.path.join(ROOT, "embodied_captioning_amd", "lib", "libcaptioner_old.so")
if not os.path.exists(OLD):
    sys.exit(f"{OLD} is missing: build the earlier revision and copy its libcaptioner_hip.so there first")
old = C.CDLL(OLD)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
BF16 = "--bf16" in sys.argv
DT = 1 if BF16 else 2


def operand(x, w=False):
    if BF16:
        return x.to(torch.bfloat16)
    d = torch.empty_like(x)
    if w:
        assert new.cap_op_convert_weight(DT, C.c_void_p(x.data_ptr()), C.c_void_p(d.data_ptr()), x.shape[0], x.shape[1], s) == 0
    else:
        assert new.cap_op_convert(DT, C.c_void_p(x.data_ptr()), C.c_void_p(d.data_ptr()), x.numel(), s) == 0
    return d


for lib in (old, new):
    lib.cap_op_gemm.restype = C.c_int
    lib.cap_op_gemm.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p]
for name, M, N, K, gelu, f32out in (("qkv", 50432, 2304, 768, 0, 0), ("fc1", 50432, 3072, 768, 1, 0), ("proj", 50432, 768, 768, 0, 1), ("fc2", 50432, 768, 3072, 0, 1),
                                   ("qkv1024", 201728, 2304, 768, 0, 0), ("fc1_1024", 201728, 3072, 768, 1, 0)):
    A = operand(torch.randn(M, K, device="cuda"))
    W = operand(torch.randn(N, K, device="cuda") / K ** 0.5, True)
    bias = torch.randn(N, device="cuda")
    outs = []
    for lib in (old, new):
        o = torch.zeros(M, N, device="cuda")
        assert lib.cap_op_gemm(DT, A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, o.data_ptr(), M, N, K, gelu, f32out, 0, s) == 0
        torch.cuda.synchronize()
        outs.append(o)
    line = f"{name}: identical {torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))} "
    for rep in range(4):
        for tag, lib, o in (("old", old, outs[0]), ("new", new, outs[1])):
            for _ in range(3):
                lib.cap_op_gemm(DT, A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, o.data_ptr(), M, N, K, gelu, f32out, 0, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                lib.cap_op_gemm(DT, A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, o.data_ptr(), M, N, K, gelu, f32out, 0, s)
            e1.record(); torch.cuda.synchronize()
            line += f" {tag} {e0.elapsed_time(e1) * 1e3 / 30:.1f}"
        line += " |"
    print(line, flush=True)
