#!/usr/bin/env python
"""A/B + in-kernel cycle accounting of the 256x256 LDS-DMA GEMMs.
tile 3 = shipped default, 5 = first-generation kernel (LDS-transposed epilogue), 10/11/12 = second generation with the
compiler / iglp_opt(0) / iglp_opt(1) schedule, 14/15 = four half-slab stages with cross-barrier fragment prefetch, 9 and 13 = instrumented builds of generation one and two: per wave, shader
cycles in total / waiting for its own DMA / waiting at the slab barrier / in the epilogue, and the shader clock derived
from the 100 MHz wall counter.  python tools/gemm_cycles.py [--check]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd import _native  # noqa: E402

lib = _native.load_library()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def gemm(A, W, bias, out, tile, gelu=0, dbg=None):
    M, K = A.shape
    N = W.shape[0]
    rc = lib.cap_op_gemm(1, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()),
                         C.c_void_p(bias.data_ptr() if bias is not None else 0),
                         C.c_void_p(dbg.data_ptr() if dbg is not None else 0), C.c_void_p(out.data_ptr()), M, N, K, gelu,
                         int(out.dtype == torch.float32), tile, s)
    assert rc == 0, lib.cap_last_error()


def check():
    torch.manual_seed(0)
    for (M, N, K, gelu, f32) in [(1576, 2304, 768, 0, 0), (1576, 3072, 768, 1, 0), (1000, 768, 3072, 0, 1),
                                 (50432, 768, 768, 0, 1), (256, 256, 128, 0, 0), (777, 516, 192, 1, 0)]:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        ref = A.float() @ W.float().t() + b
        if gelu:
            ref = torch.nn.functional.gelu(ref)
        for tile in (10, 11, 12, 14, 15):
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
            gemm(A, W, b, out, tile, gelu)
            torch.cuda.synchronize()
            err = (out.float() - ref).abs().max().item()
            tol = 2e-3 if f32 else 0.05
            print(f"check M={M} N={N} K={K} gelu={gelu} f32={f32} tile{tile}: max err {err:.4g}", "ok" if err < tol else "FAIL",
                  flush=True)
            assert err < tol


if "--check" in sys.argv:
    check()

for (M, N, K) in [(8192, 8192, 8192), (50432, 2304, 768), (50432, 768, 3072), (50432, 3072, 768), (50432, 768, 768)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    dbg = torch.zeros(256 * 8 * 6, device="cuda", dtype=torch.int64)
    line = f"M={M} N={N} K={K}:"
    for tile in (5, 12, 14, 5, 12, 14):
        for _ in range(3):
            gemm(A, W, None, out, tile)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gemm(A, W, None, out, tile)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        line += f"  tile{tile}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF"
    print(line, flush=True)
    for tile in (9, 13):
        dbg.zero_()
        for _ in range(3):
            gemm(A, W, None, out, tile, dbg=dbg)
        torch.cuda.synchronize()
        d = dbg.view(256, 8, 6).double().cpu()
        d = d[d[:, 0, 0] > 0]
        cyc, wall, dma, bar, epi, tc = [d[:, :, i] for i in range(6)]
        mhz = (cyc / (wall / 100.0)).mean().item()
        nslab = (tc * (K // 64)).mean().item()
        print(f"   tile{tile}: clock {mhz:7.1f} MHz  tiles/block {tc.mean().item():.2f}  per slab: total {cyc.mean().item() / nslab:.0f}"
              f"  dma-wait {dma.mean().item() / nslab:.0f}  barrier-wait {bar.mean().item() / nslab:.0f}"
              f"  | epilogue/tile {(epi / tc).mean().item():.0f} cycles", flush=True)
