#!/usr/bin/env python
"""A/B of the split-mode ViT attention kernel between two builds of the library (GPU box only): the in-tree one and
embodied_captioning_amd/lib/libcaptioner_old.so (a copy of an earlier build), interleaved so both see the same clocks; outputs compared
bit for bit.    python tools/ab_vit_attention.py"""
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
B, N, H = 256, 197, 12
qkv = torch.randn(B * N, 3 * H * 64, device="cuda")
g = torch.empty_like(qkv)
assert new.cap_op_convert(2, C.c_void_p(qkv.data_ptr()), C.c_void_p(g.data_ptr()), qkv.numel(), s) == 0
outs = {}
for name, lib in (("old", old), ("new", new)):
    f = lib.cap_op_vit_attention
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    ctx = torch.zeros(B * N, H * 64, device="cuda")
    assert f(2, g.data_ptr(), ctx.data_ptr(), B, N, H, 3, s) == 0
    torch.cuda.synchronize()
    outs[name] = ctx
print("identical:", torch.equal(outs["old"].view(torch.int32), outs["new"].view(torch.int32)))
for rep in range(4):
    for name, lib in (("old", old), ("new", new)):
        f = lib.cap_op_vit_attention
        ctx = outs[name]
        for _ in range(3): f(2, g.data_ptr(), ctx.data_ptr(), B, N, H, 3, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f(2, g.data_ptr(), ctx.data_ptr(), B, N, H, 3, s)
        e1.record(); torch.cuda.synchronize()
        print("%s %.1f us" % (name, e0.elapsed_time(e1) * 1e3 / 20), end="   ")
    print()
