#!/usr/bin/env python
"""Micro-benchmark of the split-fp16 (G8) GEMM over the captioner's shapes (GPU box only).
    python tools/bench_gemm_split.py            # encoder shapes (tile 3 = gemm_pp.hip), decode shapes x tiles
    python tools/bench_gemm_split.py --cycles   # in-kernel cycle accounting of gemm_big2_kernel<g8_t> (experiments build; the
                                                # kernel gemm_pp.hip replaced - tools/bench_gemm_pp.py --cycles shows both)
    python tools/bench_gemm_split.py --cus      # the same kernel on 256 / 64 workgroups: clock and cycles per stage
Prints 2MNK/t (the Linear layer's rate) and 3x that (MFMA flops executed) per variant."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from embodied_captioning_amd import _native  # noqa: E402

lib = _native.load_library()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SPLIT = 2


def g8(x, scale=1.0):
    d = torch.empty_like(x)
    if scale == 1.0:
        assert lib.cap_op_convert(SPLIT, C.c_void_p(x.data_ptr()), C.c_void_p(d.data_ptr()), x.numel(), s) == 0
    else:
        assert lib.cap_op_convert_weight(SPLIT, C.c_void_p(x.data_ptr()), C.c_void_p(d.data_ptr()), x.shape[0], x.shape[1], s) == 0
    return d


SHAPES = [("qkv", 50432, 2304, 768, 0, 0, (3,)), ("proj", 50432, 768, 768, 0, 1, (3,)),
          ("fc1", 50432, 3072, 768, 1, 0, (3,)), ("fc2", 50432, 768, 3072, 0, 1, (3,)),
          ("crosskv", 50432, 18432, 768, 0, 1, (3,)), ("sq8k", 8192, 8192, 8192, 0, 1, (3,)),
          ("vocab", 256, 30524, 768, 0, 1, (1, 2, 3)), ("dec768", 256, 768, 768, 0, 1, (1, 2)),
          ("dec_f1", 256, 3072, 768, 1, 0, (1, 2)), ("dec_qkv", 256, 2304, 768, 0, 1, (1, 2))]
for name, M, N, K, gelu, f32out, tiles in ([] if "--cycles" in sys.argv or "--cus" in sys.argv else SHAPES):
    A = g8(torch.randn(M, K, device="cuda"))
    W = g8(torch.randn(N, K, device="cuda") / K ** 0.5, 4096.0)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    line = f"{name:8s} M={M} N={N} K={K}:"
    for rep in range(2):
        for tile in tiles:
            def run():
                rc = lib.cap_op_gemm(SPLIT, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(bias.data_ptr()),
                                     C.c_void_p(0), C.c_void_p(out.data_ptr()), M, N, K, gelu, f32out, tile, s)
                assert rc == 0, lib.cap_last_error()
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            tf = 2.0 * M * N * K / us / 1e6
            line += f"  tile{tile}: {us:8.1f} us {tf:6.1f} TF ({3 * tf:6.0f} exec)"
    print(line, flush=True)

if "--cus" in sys.argv:
    # the instrumented kernel on 256 / 128 / 64 / 32 workgroups (experiments build, CAP_EXP_CUS): does the clock rise when fewer
    # CUs draw power?  M is scaled with the CU count so that every workgroup walks the same number of tiles
    PT = int(os.environ.get("PROF_TILE", "13"))
    for cus in (256, 64):
        os.environ["CAP_EXP_CUS"] = str(cus)
        for name, M0, N, K in [("qkv", 50432, 2304, 768), ("k3072", 50432, 768, 3072)]:
            M = (M0 * cus // 256 + 255) // 256 * 256
            A = g8(torch.randn(M, K, device="cuda"))
            W = g8(torch.randn(N, K, device="cuda") / K ** 0.5, 4096.0)
            out = torch.zeros(M, N, device="cuda", dtype=torch.float32)
            dbg = torch.zeros(256 * 8 * 6, device="cuda", dtype=torch.int64)
            def run():
                rc = lib.cap_op_gemm(SPLIT, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(0), C.c_void_p(dbg.data_ptr()),
                                     C.c_void_p(out.data_ptr()), M, N, K, 0, 0, PT, s)
                assert rc == 0, lib.cap_last_error()
            for _ in range(30):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(60):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 60
            d = dbg.view(256, 8, 6).double().cpu()
            d = d[d[:, 0, 0] > 0]
            cyc, wall, dma, bar, epi, tc = [d[:, :, i] for i in range(6)]
            mhz = (cyc / (wall / 100.0)).mean().item()
            nst = (tc * (K // 32)).mean().item()
            tf = 2.0 * M * N * K / us / 1e6
            print(f"cus {cus:3d} {name:6s} M={M}: {us:7.1f} us {tf:6.1f} TF = {tf / cus:.3f} TF/CU  clock {mhz:7.1f} MHz  per stage "
                  f"{(cyc.mean().item() - epi.mean().item()) / nst:.0f} cycles (dma-wait {dma.mean().item() / nst:.0f}, barrier-wait "
                  f"{bar.mean().item() / nst:.0f})  blocks {d.shape[0]}", flush=True)
    sys.exit(0)

if "--cycles" in sys.argv:
    # in-kernel cycle accounting of the shipped schedule (tile 13 = instrumented build, G8 output): per wave, shader cycles in
    # total / waiting for its own DMA (vmcnt) / waiting at the stage barrier / in the epilogue; a stage = 32 k values (64 KiB)
    for name, M, N, K in [("sq8k", 8192, 8192, 8192), ("qkv", 50432, 2304, 768), ("fc1", 50432, 3072, 768), ("k3072", 50432, 768, 3072)]:
        A = g8(torch.randn(M, K, device="cuda"))
        W = g8(torch.randn(N, K, device="cuda") / K ** 0.5, 4096.0)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32)          # G8 output: 4 bytes per element
        dbg = torch.zeros(256 * 8 * 6, device="cuda", dtype=torch.int64)
        for _ in range(3):
            rc = lib.cap_op_gemm(SPLIT, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(0), C.c_void_p(dbg.data_ptr()),
                                 C.c_void_p(out.data_ptr()), M, N, K, 0, 0, 13, s)
            assert rc == 0, lib.cap_last_error()
        torch.cuda.synchronize()
        d = dbg.view(256, 8, 6).double().cpu()
        d = d[d[:, 0, 0] > 0]
        cyc, wall, dma, bar, epi, tc = [d[:, :, i] for i in range(6)]
        mhz = (cyc / (wall / 100.0)).mean().item()
        nst = (tc * (K // 32)).mean().item()
        print(f"cycles {name:6s} M={M} N={N} K={K}: clock {mhz:7.1f} MHz  tiles/block {tc.mean().item():.2f}  per stage: total "
              f"{(cyc.mean().item() - epi.mean().item()) / nst:.0f} (MFMA issue alone: 3072)  dma-wait {dma.mean().item() / nst:.0f}  barrier-wait "
              f"{bar.mean().item() / nst:.0f}  | epilogue/tile {(epi / tc).mean().item():.0f} cycles", flush=True)
