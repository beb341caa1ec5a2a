"""GPU: each hand-written kernel, called through the C ABI, against a plain PyTorch fp32/fp64 reference of the same op."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from embodied_captioning_amd import _native
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _native.load_library()


def _p(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(lib, rc):
    assert rc == 0, lib.cap_last_error().decode()


DT = {"f32": (0, torch.float32), "bf16": (1, torch.bfloat16)}


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("tile", [1, 2, 3, 4, 6])
@pytest.mark.parametrize("shape", [(256, 256, 128), (197, 768, 768), (300, 200, 192), (33, 7632, 64), (1, 64, 64), (520, 516, 3072),
                                   (70000, 768, 128)])
def test_gemm_bias_against_fp64(lib, dtype, tile, shape):
    M, N, K = shape
    tag, tdt = DT[dtype]
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    Ad, Wd = A.to(tdt).cuda(), W.to(tdt).cuda()
    bd = bias.cuda()
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    _check(lib, lib.cap_op_gemm(tag, _p(Ad), _p(Wd), _p(bd), _p(None), _p(out), M, N, K, 0, 1, tile, _stream()))
    torch.cuda.synchronize()
    ref = Ad.double().cpu() @ Wd.double().cpu().T + bias.double()
    # operands are identical (already rounded to the compute dtype); only the fp32 accumulation order differs
    assert torch.isfinite(out).all()
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 2e-4 * math.sqrt(K / 64), err


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gemm_epilogues_gelu_residual_and_typed_output(lib, dtype):
    M, N, K = 200, 320, 256
    tag, tdt = DT[dtype]
    g = torch.Generator().manual_seed(11)
    A = torch.randn(M, K, generator=g).to(tdt)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(tdt)
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ref = A.double() @ W.double().T + bias.double()
    # GELU -> compute-dtype output
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()      # keep the device tensors alive across the async launches
    out_t = torch.zeros(M, N, dtype=tdt, device="cuda")
    _check(lib, lib.cap_op_gemm(tag, _p(Ad), _p(Wd), _p(bd), _p(None), _p(out_t), M, N, K, 1, 0, 0, _stream()))
    want = torch.nn.functional.gelu(ref)
    tol = 1e-4 if dtype == "f32" else 2e-2
    assert (out_t.float().cpu().double() - want).abs().max().item() < tol
    # in-place residual, fp32 output (C aliases resid)
    x = resid.clone().cuda()
    _check(lib, lib.cap_op_gemm(tag, _p(Ad), _p(Wd), _p(bd), _p(x), _p(x), M, N, K, 0, 1, 0, _stream()))
    assert (x.cpu().double() - (ref + resid.double())).abs().max().item() < 1e-4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("tile", [3, 4, 5])                # 3: gemm_pp.hip (bf16) / gemm_big_kernel (fp32); 5: first-generation LDS-DMA kernel
@pytest.mark.parametrize("N", [4, 8, 68, 200, 260, 516])
def test_gemm_bias_at_edge_widths_on_the_lds_dma_tiles(lib, dtype, tile, N):
    """Round-1 fault fence ("The 22:40 GEMM fault"): the 256x256 LDS-DMA kernels fetch the bias of a tile by
    16-byte DMA from bias + min(col, N - 4).  Widths that are not multiples of the 256-wide tile, down to N = 4, with the
    bias vector placed at the END of its allocation (any read past it leaves the buffer), edge rows (M = 197 / 300)."""
    tag, tdt = DT[dtype]
    for M, K in ((197, 768), (300, 192)):
        g = torch.Generator().manual_seed(N * 31 + M)
        A = torch.randn(M, K, generator=g).to(tdt)
        W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(tdt)
        pool = torch.zeros(4096, dtype=torch.float32, device="cuda")
        bias = pool[4096 - N:]                                    # 16-byte aligned: N % 4 == 0
        bias.copy_(torch.randn(N, generator=g))
        Ad, Wd = A.cuda(), W.cuda()
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        _check(lib, lib.cap_op_gemm(tag, _p(Ad), _p(Wd), _p(bias), _p(None), _p(out), M, N, K, 0, 1, tile, _stream()))
        torch.cuda.synchronize()
        ref = A.double() @ W.double().T + bias.cpu().double()
        assert (out.cpu().double() - ref).abs().max().item() < 2e-4 * math.sqrt(K / 64)


def test_gemm_rejects_widths_and_pointers_the_dma_path_cannot_take(lib):
    a = torch.zeros(64, 64, device="cuda")
    b = torch.zeros(80, device="cuda")
    # N % 4 != 0: the clamped bias DMA address would be 4-byte aligned only
    rc = lib.cap_op_gemm(0, _p(a), _p(a), _p(b), _p(None), _p(a), 64, 62, 64, 0, 1, 3, _stream())
    assert rc != 0 and b"multiples of 4" in lib.cap_last_error()
    # a bias vector that is not 16-byte aligned
    rc = lib.cap_op_gemm(0, _p(a), _p(a), _p(b[1:]), _p(None), _p(a), 64, 64, 64, 0, 1, 3, _stream())
    assert rc != 0 and b"16-byte aligned" in lib.cap_last_error()


def test_gemm_rejects_bad_shapes(lib):
    a = torch.zeros(64, 48, device="cuda")
    rc = lib.cap_op_gemm(0, _p(a), _p(a), _p(None), _p(None), _p(a), 64, 64, 48, 0, 1, 0, _stream())
    assert rc != 0 and b"multiple" in lib.cap_last_error()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("D", [128, 768, 1024])
def test_layernorm(lib, dtype, D):
    tag, tdt = DT[dtype]
    M = 131
    g = torch.Generator().manual_seed(D)
    x = torch.randn(M, D, generator=g) * 3 + 1
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    for eps in (1e-5, 1e-12):
        xd, gd, bd = x.cuda(), gamma.cuda(), beta.cuda()
        out_t = torch.zeros(M, D, dtype=tdt, device="cuda")
        out_f = torch.zeros(M, D, dtype=torch.float32, device="cuda")
        _check(lib, lib.cap_op_layernorm(tag, _p(xd), _p(gd), _p(bd), C.c_float(eps), _p(out_t),
                                         _p(out_f), M, D, _stream()))
        ref = torch.nn.functional.layer_norm(x, (D,), gamma, beta, eps)
        assert (out_f.cpu() - ref).abs().max().item() < 2e-5
        assert (out_t.float().cpu() - ref).abs().max().item() < (2e-5 if dtype == "f32" else 5e-2)


def _attn_ref(qkv, B, N, H):
    D = H * 64
    x = qkv.double().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (x[0] @ x[1].transpose(-1, -2)) * 0.125
    return (torch.softmax(s, -1) @ x[2]).permute(0, 2, 1, 3).reshape(B * N, D)


@pytest.mark.parametrize("dtype,impl", [("f32", 1), ("bf16", 1), ("bf16", 2)])
@pytest.mark.parametrize("N", [17, 197, 257, 577])
def test_vit_attention(lib, dtype, impl, N):
    tag, tdt = DT[dtype]
    B, H = 3, 4
    g = torch.Generator().manual_seed(N)
    qkv = (torch.randn(B * N, 3 * H * 64, generator=g) * 1.5).to(tdt)
    qd = qkv.cuda()
    ctx = torch.full((B * N, H * 64), float("nan"), dtype=tdt, device="cuda")
    _check(lib, lib.cap_op_vit_attention(tag, _p(qd), _p(ctx), B, N, H, impl, _stream()))
    torch.cuda.synchronize()
    ref = _attn_ref(qkv, B, N, H)
    err = (ctx.float().cpu().double() - ref).abs().max().item()
    assert err < (1e-5 if dtype == "f32" else 3e-2), err


def test_vit_attention_mfma_matches_scalar_kernel(lib):
    """Same bf16 inputs through both kernels: they differ only by the bf16 rounding of P."""
    B, H, N = 5, 12, 197
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g).to(torch.bfloat16).cuda()
    a = torch.zeros(B * N, H * 64, dtype=torch.bfloat16, device="cuda")
    b = torch.zeros_like(a)
    _check(lib, lib.cap_op_vit_attention(1, _p(qkv), _p(a), B, N, H, 1, _stream()))
    _check(lib, lib.cap_op_vit_attention(1, _p(qkv), _p(b), B, N, H, 2, _stream()))
    assert (a.float() - b.float()).abs().max().item() < 2e-2


@pytest.mark.parametrize("M,N,K", [(32, 7680, 2560), (2, 2560, 2560), (40, 2560, 10240), (7, 96, 512), (32, 64, 256),
                                   (5, 10240, 2560), (128, 10240, 2560), (97, 2560, 10240), (70, 160, 512)])
def test_gemm_skinny(lib, M, N, K):
    """Weight-streaming decode GEMM: finished output (bias + ReLU) and split-K slice sums against float64; a row's result
    does not depend on the rows it rides with (bitwise)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().T
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    S = lib.cap_op_gemm_skinny_slices(N, K, 0)
    assert S >= 1
    part = torch.full((S, M, N), float("nan"), device="cuda")
    assert lib.cap_op_gemm_skinny(_p(Ad), _p(Wd), None, 0, None, _p(part), M, N, K, _stream()) == S, lib.cap_last_error()
    torch.cuda.synchronize()
    assert (part.sum(0).cpu().double() - ref).abs().max().item() < 1e-3 * (K / 256) ** 0.5
    if lib.cap_op_gemm_skinny_slices(N, K, 1) == 1:      # one workgroup can hold the whole K: the finished form
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        assert lib.cap_op_gemm_skinny(_p(Ad), _p(Wd), _p(bd), 2, _p(out), None, M, N, K, _stream()) == 1
        want = torch.relu(ref + bias.double())
        assert (out.float().cpu().double() - want).abs().max().item() < 2e-2 * max(1.0, want.abs().max().item())
        one = torch.full((1, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        assert lib.cap_op_gemm_skinny(_p(Ad[M - 1:]), _p(Wd), _p(bd), 2, _p(one), None, 1, N, K, _stream()) == 1
        assert torch.equal(one[0], out[M - 1])
    else:
        assert K > 2560


@pytest.mark.parametrize("M,D,S,row_block", [(32, 2560, 4, 1), (32, 2560, 2, 1), (5, 3072, 1, 1), (7, 1028, 3, 1), (64, 768, 8, 1),
                                              (32, 2560, 4, 0), (300, 768, 4, 0)])
def test_reduce_layernorm(lib, M, D, S, row_block):
    """Split-K consumer (sum of slices + bias + residual, LayerNorm) in its three kernels, residual stream updated in place."""
    g = torch.Generator().manual_seed(M * D + S)
    part = torch.randn(S, M, D, generator=g)
    bias, resid = torch.randn(D, generator=g), torch.randn(M, D, generator=g)
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
    y = part.double().sum(0) + bias.double() + resid.double()
    ref = torch.nn.functional.layer_norm(y, (D,), gamma.double(), beta.double(), 1e-5)
    x, pd, bd, gd, btd = resid.cuda(), part.cuda(), bias.cuda(), gamma.cuda(), beta.cuda()
    out_t = torch.full((M, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    out_f = torch.full((M, D), float("nan"), device="cuda")
    _check(lib, lib.cap_op_reduce_layernorm(1, _p(pd), S, _p(bd), _p(x), _p(gd), _p(btd), C.c_float(1e-5), _p(out_t), _p(out_f),
                                            _p(x), M, D, row_block, _stream()))
    torch.cuda.synchronize()
    assert (x.cpu().double() - y).abs().max().item() < 1e-5
    assert (out_f.cpu().double() - ref).abs().max().item() < 3e-5
    assert (out_t.float().cpu().double() - ref).abs().max().item() < 5e-2


def test_gemm_skinny_rejects_bad_shapes(lib):
    x = torch.zeros(64, 512, dtype=torch.bfloat16, device="cuda")
    f = torch.zeros(16 * 64 * 64, device="cuda")
    assert lib.cap_op_gemm_skinny(_p(x), _p(x), None, 0, _p(x), None, 4, 48, 512, _stream()) == -1     # N % 32
    assert lib.cap_op_gemm_skinny(_p(x), _p(x), None, 0, None, _p(f), 4, 64, 320, _stream()) == -1     # K % 256
    assert lib.cap_op_gemm_skinny(_p(x), _p(x), None, 0, _p(x), None, 4, 64, 10240, _stream()) == -1   # finished form: K > 2560
    assert b"gemm_skinny" in lib.cap_last_error()
    assert lib.cap_op_gemm_skinny_slices(64, 320, 0) == 0


@pytest.mark.parametrize("hd", [72, 88, 128])
def test_vit_attention_wide_heads(lib, hd):
    """Heads wider than 64 at 257 tokens (BLIP-2's ViT-g/14 is 88): the two-half MFMA kernel against a float64 reference and
    against the scalar kernel on the same bf16 inputs."""
    B, H, N = 3, 5, 257
    g = torch.Generator().manual_seed(hd)
    qkv = (torch.randn(B * N, 3 * H * hd, generator=g) * 1.2).to(torch.bfloat16)
    qd = qkv.cuda()
    a = torch.full((B * N, H * hd), float("nan"), dtype=torch.bfloat16, device="cuda")
    b = torch.full_like(a, float("nan"))
    _check(lib, lib.cap_op_vit_attention_hd(1, _p(qd), _p(a), B, N, H, hd, 0, _stream()))
    _check(lib, lib.cap_op_vit_attention_hd(1, _p(qd), _p(b), B, N, H, hd, 1, _stream()))
    torch.cuda.synchronize()
    x = qkv.double().view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    sc = (x[0] @ x[1].transpose(-1, -2)) / hd ** 0.5
    ref = (torch.softmax(sc, -1) @ x[2]).permute(0, 2, 1, 3).reshape(B * N, H * hd)
    assert (a.float().cpu().double() - ref).abs().max().item() < 3e-2
    assert (a.float() - b.float()).abs().max().item() < 2e-2


@pytest.mark.parametrize("N,hd", [(33, 80), (64, 128), (17, 72), (33, 64)])
def test_causal_prefill_attention(lib, N, hd):
    """Decoder prefill over fused q|k|v rows (OPT: 33 positions, 80-wide heads): MFMA kernel and scalar kernel against float64."""
    B, H = 3, 4
    g = torch.Generator().manual_seed(N * hd)
    qkv = (torch.randn(B * N, 3 * H * hd, generator=g) * 1.2).to(torch.bfloat16)
    qd = qkv.cuda()
    x = qkv.double().view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    sc = (x[0] @ x[1].transpose(-1, -2)) / hd ** 0.5 + torch.full((N, N), float("-inf"), dtype=torch.float64).triu(1)
    ref = (torch.softmax(sc, -1) @ x[2]).permute(0, 2, 1, 3).reshape(B * N, H * hd)
    for impl in (8, 9):
        out = torch.full((B * N, H * hd), float("nan"), dtype=torch.bfloat16, device="cuda")
        _check(lib, lib.cap_op_vit_attention_hd(1, _p(qd), _p(out), B, N, H, hd, impl, _stream()))
        torch.cuda.synchronize()
        assert (out.float().cpu().double() - ref).abs().max().item() < 3e-2, impl


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("n_keys,beams", [(1, 1), (7, 1), (20, 3), (33, 2), (197, 3), (255, 1), (197, 5), (255, 2), (577, 4),
                                          (197, 6)])
def test_decode_attention_with_ancestry_and_shared_kv(lib, dtype, n_keys, beams, impl):
    tag, tdt = DT[dtype]
    Bimg, H, kv_ld = 4, 3, max(n_keys, 20)
    R = Bimg * beams
    g = torch.Generator().manual_seed(n_keys * 10 + beams)
    q = torch.randn(R, H * 64, generator=g).to(tdt)
    shared = n_keys >= 197                                 # cross-attention: rows of one image share K/V
    rows_kv = Bimg if shared else R
    K = torch.randn(rows_kv, H, kv_ld, 64, generator=g).to(tdt)
    V = torch.randn(rows_kv, H, kv_ld, 64, generator=g).to(tdt)
    anc = None
    if not shared:
        anc = torch.randint(0, R, (R, kv_ld), generator=g, dtype=torch.int32)
    qd, Kd, Vd = q.cuda(), K.cuda(), V.cuda()
    ad = anc.cuda() if anc is not None else None
    out = torch.zeros(R, H * 64, dtype=tdt, device="cuda")
    _check(lib, lib.cap_op_decode_attention(tag, _p(qd), _p(Kd), _p(Vd), _p(ad),
                                            kv_ld, beams if shared else 1, kv_ld, n_keys, _p(out), R, H, impl, _stream()))
    ref = torch.zeros(R, H * 64, dtype=torch.float64)
    for r in range(R):
        for h in range(H):
            src = [int(anc[r, j]) if anc is not None else r // beams for j in range(n_keys)]
            k = torch.stack([K[src[j], h, j] for j in range(n_keys)]).double()
            v = torch.stack([V[src[j], h, j] for j in range(n_keys)]).double()
            p = torch.softmax((k @ q[r, h * 64:(h + 1) * 64].double()) * 0.125, 0)
            ref[r, h * 64:(h + 1) * 64] = p @ v
    err = (out.float().cpu().double() - ref).abs().max().item()
    assert err < (1e-5 if dtype == "f32" else 2e-2), err
    if shared and beams > 1 and impl == 0:
        # the kernel that serves all beams of an image from one pass over the shared block (2-5 beams) gives every row the bits
        # of the one-wave-per-row kernel (here: each row with its own copy of the block)
        Kr, Vr = Kd.repeat_interleave(beams, 0).contiguous(), Vd.repeat_interleave(beams, 0).contiguous()
        out1 = torch.zeros_like(out)
        _check(lib, lib.cap_op_decode_attention(tag, _p(qd), _p(Kr), _p(Vr), None, kv_ld, 1, kv_ld, n_keys, _p(out1), R, H, 0,
                                                _stream()))
        assert torch.equal(out, out1)


@pytest.mark.parametrize("V,K", [(30524, 3), (49408, 5), (49408, 2), (1000, 8), (64, 3), (30524, 7)])
@pytest.mark.parametrize("kind", ["random", "quantised", "constant", "ramp"])
def test_beam_candidate_selection_exact_with_ties(lib, V, K, kind):
    """2K best (raw logit + running score, token) per row, ties to the lower token id: the threshold selection, its fallback
    (more than 1024 elements at the bound: constant / coarsely quantised rows) and both vocabulary paths (row staged in LDS /
    streamed from global memory) against a host sort.  Raw-score mode: one fp32 add per element, so the comparison is exact."""
    B = 3
    R, C = B * K, 2 * K
    ld = (V + 3) // 4 * 4 + 8
    g = torch.Generator().manual_seed(V + K)
    if kind == "random":
        x = torch.randn(R, ld, generator=g)
    elif kind == "quantised":
        x = torch.randint(0, 4, (R, ld), generator=g).float()
    elif kind == "constant":
        x = torch.full((R, ld), 1.5)
    else:
        x = torch.arange(ld).float().repeat(R, 1) * 0.25           # the best are the last columns of the row
    masked = 7 if V > 7 else -1
    xd = x.cuda()
    val = torch.zeros(R, C, device="cuda")
    idx = torch.zeros(R, C, dtype=torch.int32, device="cuda")
    _check(lib, lib.cap_op_beam_candidates(_p(xd), ld, V, B, K, 1, masked, _p(val), _p(idx), _stream()))
    run = np.where(np.arange(R) % K == 0, np.float32(0), np.float32(-1e9)).astype(np.float32)
    sc = x[:, :V].numpy() + run[:, None]
    if masked >= 0:
        sc[:, masked] = -np.inf
    for r in range(R):
        order = np.lexsort((np.arange(V), -sc[r].astype(np.float64)))[:C]
        assert np.array_equal(idx[r].cpu().numpy(), order), (r, idx[r].cpu().numpy(), order)
        assert np.array_equal(val[r].cpu().numpy(), sc[r][order])


@pytest.mark.parametrize("V,K", [(30524, 3), (49408, 3), (512, 2)])
def test_beam_candidate_selection_log_softmax_scores(lib, V, K):
    """HF-v5 scoring (log-softmax + running score) through both vocabulary paths (row staged in LDS / streamed from global
    memory): tokens equal torch's top-2K of the fp32 log-softmax, scores within 1e-5."""
    B = 2
    R, C = B * K, 2 * K
    ld = (V + 3) // 4 * 4
    g = torch.Generator().manual_seed(V * 7 + K)
    x = torch.randn(R, ld, generator=g) * 3.0
    xd = x.cuda()
    val = torch.zeros(R, C, device="cuda")
    idx = torch.zeros(R, C, dtype=torch.int32, device="cuda")
    _check(lib, lib.cap_op_beam_candidates(_p(xd), ld, V, B, K, 0, -1, _p(val), _p(idx), _stream()))
    run = torch.where(torch.arange(R) % K == 0, 0.0, -1.0e9).double()
    sc = torch.log_softmax(x[:, :V].double(), -1) + run[:, None]
    for r in range(0, R, K):                       # rows with running score 0: fp32 resolves the scores
        top = torch.topk(sc[r], C)
        assert torch.equal(idx[r].cpu().long(), top.indices), r
        assert (val[r].cpu().double() - top.values).abs().max().item() < 1e-5


def test_lds_dma_gemm_kernels_match_generic_kernel_bitwise(lib):
    """The persistent LDS-DMA kernels (raw barriers, counted vmcnt) share the generic kernel's accumulation order: any repeat
    that differs from it bit for bit would be an LDS read overtaking its DMA (tools/gemm_race_screen.py is the long form)."""
    torch.manual_seed(1)
    for (M, N, K, f32, gelu) in [(12608, 2304, 768, 0, 0), (12608, 768, 3072, 1, 0), (3000, 516, 192, 0, 1)]:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        dt = torch.float32 if f32 else torch.bfloat16

        def run(tile, out):
            _check(lib, lib.cap_op_gemm(1, _p(A), _p(W), _p(b), None, _p(out), M, N, K, gelu, f32, tile, _stream()))
        ref = torch.empty(M, N, device="cuda", dtype=dt)
        run(1, ref)
        for tile in (3, 5):                                       # gemm_pp.hip, the first-generation kernel
            for _ in range(10):
                out = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
                run(tile, out)
                assert torch.equal(out, ref), (M, N, K, tile)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("tile", [2, 3, 4, 5, 6])
def test_gemm_partial_row_tile_writes_nothing_outside_c(lib, dtype, tile):
    """Round 1's unexplained GEMM fault showed a wrong result on the LDS-DMA tile at M = 197 (one partial row tile) right
    before the abort - the signature of an out-of-bounds WRITE.  C sits between two canary bands here (and the bias vector at
    the very end of its allocation); after the launch the bands must be untouched and C must be right."""
    M, N, K = 197, 768, 768
    tag, tdt = DT[dtype]
    if tile == 6 and dtype == "f32":
        pytest.skip("the rows kernel is a bf16 / split kernel")
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).to(tdt).cuda()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(tdt).cuda()
    pad = 64 * N
    buf = torch.full((pad + M * N + pad,), 12345.0, dtype=torch.float32, device="cuda")
    Cv = buf[pad: pad + M * N].view(M, N)
    bias_store = torch.zeros(4096 + N, device="cuda")
    bias = bias_store[4096:]
    bias.copy_(torch.randn(N, generator=g))
    _check(lib, lib.cap_op_gemm(tag, _p(A), _p(W), _p(bias), _p(None), _p(Cv), M, N, K, 0, 1, tile, _stream()))
    torch.cuda.synchronize()
    assert (buf[:pad] == 12345.0).all() and (buf[pad + M * N:] == 12345.0).all()
    ref = A.double().cpu() @ W.double().cpu().T + bias.double().cpu()
    assert (Cv.cpu().double() - ref).abs().max().item() < 2e-4 * math.sqrt(K / 64)

@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("geom", [(3, 197, 12, 2, 768), (2, 50, 2, 3, 128), (1, 197, 12, 1, 768)])
def test_cross_kv_gemm_scatters_into_the_cache_layout(lib, dtype, geom):
    """The cross-K/V GEMM (EPI_CROSSKV; bf16: gemm_pp.hip for >= 128 tiles, the register-staged tiles below) writes
    [layer][k | v][image][head][token][64]: against the per-layer Linear calls it replaces (HF:modeling_blip_text.py:161-175)."""
    n_img, tokens, heads, layers, K = geom
    tag, tdt = DT[dtype]
    M, N = n_img * tokens, layers * 2 * heads * 64
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).to(tdt)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(tdt)
    bias = torch.randn(N, generator=g)
    cache = torch.full((layers * 2 * n_img * heads * tokens, 64), float("nan"), dtype=tdt, device="cuda")
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()               # (named: a temporary would be freed before the kernel runs)
    _check(lib, lib.cap_op_gemm_crosskv(tag, _p(Ad), _p(Wd), _p(bd), _p(cache), n_img, tokens, heads, layers, K, 0, _stream()))
    torch.cuda.synchronize()
    ref = (A.double() @ W.double().T + bias.double()).view(n_img, tokens, layers, 2, heads, 64).permute(2, 3, 0, 4, 1, 5)
    got = cache.cpu().double().view(layers, 2, n_img, heads, tokens, 64)
    tol = 2e-4 * math.sqrt(K / 64) if dtype == "f32" else 0.03
    assert torch.isfinite(got).all() and (got - ref).abs().max().item() < tol


@pytest.mark.parametrize("S,with_resid", [(1, True), (2, True), (4, True), (4, False), (3, True)])
@pytest.mark.parametrize("D", [768, 128, 1024])
def test_reduce_layernorm_kernels_agree_bit_for_bit(lib, D, S, with_resid):
    """The decoder's split-K consumer runs as one block per row up to a few hundred rows and as one wave per row at the engine
    pool's merged passes (~1 000 rows): a frame's LayerNorm row must not depend on which - same fp32 row, same G8 / bf16 operand
    row, bit for bit."""
    M = 333
    g = torch.Generator().manual_seed(D + S)
    part = (torch.randn(S, M, D, generator=g) * 3).cuda()
    bias, gamma, beta = torch.randn(D, generator=g).cuda(), torch.randn(D, generator=g).cuda(), torch.randn(D, generator=g).cuda()
    resid = (torch.randn(M, D, generator=g) * 2).cuda() if with_resid else None
    for tag, tdt in ((2, torch.float32), (1, torch.bfloat16)):          # split mode (G8 container: 4 bytes per element), bf16
        outs = []
        for row_block in (1, 0):
            out_t = torch.zeros(M, D, dtype=tdt, device="cuda")
            out_f = torch.zeros(M, D, device="cuda")
            _check(lib, lib.cap_op_reduce_layernorm(tag, _p(part), S, _p(bias), _p(resid), _p(gamma), _p(beta), C.c_float(1e-12), _p(out_t),
                                                    _p(out_f), _p(None), M, D, row_block, _stream()))
            torch.cuda.synchronize()
            outs.append((out_t, out_f))
        assert torch.equal(outs[0][1], outs[1][1])
        assert torch.equal(outs[0][0].view(torch.int32 if tdt == torch.float32 else torch.int16), outs[1][0].view(torch.int32 if tdt == torch.float32 else torch.int16))
        if with_resid:
            # the pre-LN form the CoCa decoder calls: the sum goes back into the residual stream IN PLACE (y_out aliases resid),
            # no fp32 LayerNorm row - same sum and same operand row from both kernels
            ys = []
            for row_block in (1, 0):
                x = resid.clone()
                out_t = torch.zeros(M, D, dtype=tdt, device="cuda")
                _check(lib, lib.cap_op_reduce_layernorm(tag, _p(part), S, _p(bias), _p(x), _p(gamma), _p(beta), C.c_float(1e-5), _p(out_t),
                                                        _p(None), _p(x), M, D, row_block, _stream()))
                torch.cuda.synchronize()
                ys.append((x, out_t))
            assert torch.equal(ys[0][0], ys[1][0])
            assert torch.equal(ys[0][1].view(torch.int32 if tdt == torch.float32 else torch.int16), ys[1][1].view(torch.int32 if tdt == torch.float32 else torch.int16))
            want = part.sum(0) + bias + resid
            assert (ys[0][0] - want).abs().max().item() < 1e-4
