#!/usr/bin/env python
"""A/B of the skewed-group split GEMM (gemm_pp.hip, tile 20 = what tile 3 selects) against gemm_big2_kernel (tile 10) on the encoder shapes (GPU box).
    python tools/bench_gemm_pp.py            # bit-identity + interleaved timing
    python tools/bench_gemm_pp.py --cycles   # in-kernel cycle stamps of both (experiments build), on 256 and 64 workgroups"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd import _native  # noqa: E402

lib = _native.load_library()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
BF16 = "--bf16" in sys.argv                   # bf16 operands: tile 20 against gemm_big3_kernel (tile 14; what K <= 1024 used) / gemm_big2 (tile 12)
SPLIT = 1 if BF16 else 2


def g8(x, w=False):
    if BF16:
        return x.to(torch.bfloat16)
    d = torch.empty_like(x)
    if w:
        assert lib.cap_op_convert_weight(SPLIT, C.c_void_p(x.data_ptr()), C.c_void_p(d.data_ptr()), x.shape[0], x.shape[1], s) == 0
    else:
        assert lib.cap_op_convert(SPLIT, C.c_void_p(x.data_ptr()), C.c_void_p(d.data_ptr()), x.numel(), s) == 0
    return d


def gemm(A, W, bias, out, M, N, K, gelu, f32out, tile, aux=None):
    rc = lib.cap_op_gemm(SPLIT, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(bias.data_ptr() if bias is not None else 0),
                         C.c_void_p(aux.data_ptr() if aux is not None else 0), C.c_void_p(out.data_ptr()), M, N, K, gelu, f32out, tile, s)
    assert rc == 0, lib.cap_last_error()


SHAPES = [("qkv", 50432, 2304, 768, 0, 0), ("proj", 50432, 768, 768, 0, 1), ("fc1", 50432, 3072, 768, 1, 0),
          ("fc2", 50432, 768, 3072, 0, 1), ("edge", 1000, 776, 96, 0, 1), ("edge2", 3333, 520, 64, 1, 0)]
if "--cycles" in sys.argv:
    for cus in (256, 64):
        os.environ["CAP_EXP_CUS"] = str(cus)
        for name, M0, N, K in [("qkv", 50432, 2304, 768), ("k3072", 50432, 768, 3072)]:
            M = (M0 * cus // 256 + 255) // 256 * 256
            A = g8(torch.randn(M, K, device="cuda"))
            W = g8(torch.randn(N, K, device="cuda") / K ** 0.5, True)
            out = torch.zeros(M, N, device="cuda", dtype=torch.float32)
            for tile, f32o in ((13, 0), (21, 0), (21, 1)):
                st = 8 if tile == 21 else 6
                dbg = torch.zeros(256 * 8 * st, device="cuda", dtype=torch.int64)
                for _ in range(30):
                    gemm(A, W, None, out, M, N, K, 0, f32o, tile, dbg)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(60):
                    gemm(A, W, None, out, M, N, K, 0, f32o, tile, dbg)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 60
                d = dbg.view(256, 8, st).double().cpu()
                d = d[d[:, 0, 0] > 0]
                cyc, wall, dma, bar, epi, tc = [d[:, :, i] for i in range(6)]
                mhz = (cyc / (wall / 100.0)).mean().item()
                nst = (tc * (K // 32)).mean().item()
                tf = 2.0 * M * N * K / us / 1e6
                extra = f" issue {d[:, :, 6].mean().item() / nst:.0f}" if st == 8 else ""
                print(f"cus {cus:3d} {name:6s} tile {tile} f32out {f32o}: {us:7.1f} us {tf:6.1f} TF  clock {mhz:7.1f} MHz  per stage "
                      f"{(cyc.mean().item() - epi.mean().item()) / nst:.0f} cycles (vm-wait {dma.mean().item() / nst:.0f} barrier "
                      f"{bar.mean().item() / nst:.0f}{extra})  epilogue/tile {(epi / tc).mean().item():.0f}", flush=True)
                if st == 8:
                    g0 = d[:, :4, :].mean(dim=(0, 1)); g1 = d[:, 4:, :].mean(dim=(0, 1))
                    print(f"      group 0 / 1: vm-wait {g0[2] / nst:.0f} / {g1[2] / nst:.0f}  barrier {g0[3] / nst:.0f} / {g1[3] / nst:.0f}  "
                          f"issue {g0[6] / nst:.0f} / {g1[6] / nst:.0f}  epilogue/tile {g0[4] / g0[5]:.0f} / {g1[4] / g1[5]:.0f} of which waiting for DMA {g0[7] / g0[5]:.0f} / {g1[7] / g1[5]:.0f}", flush=True)
    sys.exit(0)

if BF16:
    SHAPES = [x for x in SHAPES if x[3] % 64 == 0 and x[3] >= 128] + [("edgeb", 3333, 520, 192, 1, 0), ("edgef", 1000, 776, 128, 0, 1)]
for name, M, N, K, gelu, f32out in SHAPES:
    A = g8(torch.randn(M, K, device="cuda"))
    W = g8(torch.randn(N, K, device="cuda") / K ** 0.5, True)
    bias = torch.randn(N, device="cuda")
    outs = {}
    REF = (14 if K <= 1024 else 12) if BF16 else 10
    for tile in (REF, 20):
        out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32)
        gemm(A, W, bias, out, M, N, K, gelu, f32out, tile)
        torch.cuda.synchronize()
        outs[tile] = out
    same = torch.equal(outs[REF].view(torch.int32), outs[20].view(torch.int32))
    line = f"{name:6s} M={M} N={N} K={K}: identical={same}"
    if not same:
        d = (outs[REF] - outs[20]).abs()
        line += f" maxdiff={d.max().item():.3e} nan20={torch.isnan(outs[20]).sum().item()} nanref={torch.isnan(outs[REF]).sum().item()}"
    out = outs[REF]
    for rep in range(3):
        for tile in (REF, 20):
            for _ in range(3):
                gemm(A, W, bias, out, M, N, K, gelu, f32out, tile)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 30
            e0.record()
            for _ in range(n):
                gemm(A, W, bias, out, M, N, K, gelu, f32out, tile)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            line += f"  t{tile}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF"
    print(line, flush=True)
