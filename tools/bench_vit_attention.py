#!/usr/bin/env python
"""Micro-benchmark of the split-mode ViT attention kernel (GPU box only): 256 images x 12 heads x 197 tokens, G8 q|k|v in, G8 context
out.    python tools/bench_vit_attention.py"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.getcwd())
from embodied_captioning_amd import _native
lib = _native.load_library(); s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, N, H = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 197, 12
qkv = torch.randn(B * N, 3 * H * 64, device='cuda')
g = torch.empty_like(qkv); assert lib.cap_op_convert(2, C.c_void_p(qkv.data_ptr()), C.c_void_p(g.data_ptr()), qkv.numel(), s) == 0
ctx = torch.zeros(B * N, H * 64, device='cuda')
# impl 3 = what the encoder runs (more units than CUs: the persistent kernel), impl 5 = the one-workgroup-per-unit kernel; interleaved
for impl in (5, 3):
    for _ in range(5): assert lib.cap_op_vit_attention(2, C.c_void_p(g.data_ptr()), C.c_void_p(ctx.data_ptr()), B, N, H, impl, s) == 0
for rep in range(4):
    for impl, name in ((5, 'per-unit'), (3, 'persistent')):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 20
        for _ in range(n): lib.cap_op_vit_attention(2, C.c_void_p(g.data_ptr()), C.c_void_p(ctx.data_ptr()), B, N, H, impl, s)
        e1.record(); torch.cuda.synchronize()
        print('%-10s us per launch %.1f  checksum %.6f' % (name, e0.elapsed_time(e1) * 1e3 / n, ctx.double().abs().sum().item()))
