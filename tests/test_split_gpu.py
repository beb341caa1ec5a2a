"""GPU: the split-fp16 mode (CAP_F32_SPLIT, dtype "f32s") kernel by kernel, through the C ABI.

Every GEMM operand is G8 (two fp16 halves per fp32 value, tests/_util.g8_encode); a product is hi.hi + hi.lo + lo.hi on
the fp16 MFMA pipe with fp32 accumulation.  The claims checked here: (1) the GEMM is as accurate as the fp32-MFMA GEMM
against an fp64 reference of the ORIGINAL fp32 operands (tolerance 1e-5 on O(1) outputs - bf16 misses it by 3 orders);
(2) every tile shape gives the same bits (batch invariance); (3) the kernels that feed a GEMM write correct G8."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from _util import G8_WSCALE, g8_decode, g8_encode

SPLIT = 2
gpu = pytest.mark.gpu


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


def _g8(x, scale=1.0):
    return torch.from_numpy(g8_encode(x.numpy(), scale)).cuda()


def test_g8_container_round_trips_on_the_host():
    x = (np.random.default_rng(0).standard_normal((5, 64)) * np.logspace(-6, 3, 64)).astype(np.float32)
    back = g8_decode(g8_encode(x))
    # 2^-22 relative while the lo half is a normal fp16 number; below that the absolute error is half an fp16 subnormal step
    assert np.all(np.abs(back - x) <= np.maximum(np.abs(x) * 2.0 ** -22, 2.0 ** -25))
    wb = g8_decode(g8_encode(x * 1e-3, G8_WSCALE), G8_WSCALE)         # weights travel scaled by 4096
    assert np.all(np.abs(wb - x * 1e-3) <= np.maximum(np.abs(x * 1e-3) * 2.0 ** -22, 2.0 ** -25 / G8_WSCALE))


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 6])
@pytest.mark.parametrize("shape", [(256, 256, 128), (197, 768, 768), (300, 200, 192), (33, 7632, 64), (1, 64, 64), (520, 516, 3072),
                                   (2000, 2304, 768), (70000, 768, 96)])
@gpu
def test_split_gemm_is_fp32_grade(lib, tile, shape):
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    Ad, Wd, bd = _g8(A), _g8(W, G8_WSCALE), bias.cuda()
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    _check(lib, lib.cap_op_gemm(SPLIT, _p(Ad), _p(Wd), _p(bd), _p(None), _p(out), M, N, K, 0, 1, tile, _stream()))
    torch.cuda.synchronize()
    ref = A.double() @ W.double().T + bias.double()          # the ORIGINAL fp32 operands, not their split images
    assert torch.isfinite(out).all()
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 1e-5 * math.sqrt(max(K, 256) / 256), err
    # the exact-product fp32 MFMA kernel on the same operands is no closer than a factor of a few
    o32 = torch.empty_like(out)
    _check(lib, lib.cap_op_gemm(0, _p(A.cuda()), _p(W.cuda()), _p(bd), _p(None), _p(o32), M, N, K, 0, 1, 0, _stream()))
    torch.cuda.synchronize()
    e32 = (o32.cpu().double() - ref).abs().max().item()
    assert err < 8 * e32 + 2e-6, (err, e32)


@gpu
@pytest.mark.parametrize("shape", [(520, 768, 768), (66000, 768, 192), (17000, 3072, 64)])
def test_split_gemm_tiles_are_bit_identical(lib, shape):
    """Batch invariance: the tile shape follows the row count, so every tile shape must produce the same bits.  The two large
    shapes have a last round of 256 x 256 tiles that is less than half full on 256 CUs (774 = 3 x 256 + 6 and 804 = 3 x 256 +
    36 tiles): the LDS-DMA kernel runs those tiles as 128-row halves in a second launch (launch_big2) - same bits."""
    M, N, K = shape
    g = torch.Generator().manual_seed(3)
    Ad, Wd = _g8(torch.randn(M, K, generator=g)), _g8(torch.randn(N, K, generator=g) / math.sqrt(K), G8_WSCALE)
    bd = torch.randn(N, generator=g).cuda()
    outs = []
    for tile in (1, 2, 3, 4):
        o = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        _check(lib, lib.cap_op_gemm(SPLIT, _p(Ad), _p(Wd), _p(bd), _p(None), _p(o), M, N, K, 0, 1, tile, _stream()))
        outs.append(o)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


@gpu
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("shape", [(52000, 776, 64), (1000, 520, 96), (66000, 768, 192), (300, 2304, 768), (1500, 4360, 64), (9500, 2040, 64)])
def test_split_gemm_g8_output_is_bit_identical_across_kernels(lib, shape, act):
    """The 256x256 kernel with the skewed wave groups (gemm_pp.hip: what tile 3 selects for G8 operands; its epilogue builds the
    [8 hi | 8 lo] row image in LDS strips and stores whole 128-byte lines) against the register-staged tiles: G8 output with
    bias and no activation / GELU / ReLU, ragged M and N edges, a half-tile tail launch ((66000, 768): 774 tiles), clamped groups
    counted alike; (1500, 4360): more than 16 column tiles, numbered in bands of four tile rows (the last band has two);
    (9500, 2040): 304 tiles with ragged edges both ways, the 48 of the partial round as half tiles.
    (gemm_big2_kernel<g8_t>, the kernel it replaced, exists in experiments builds only: tools/bench_gemm_pp.py
    checks bit-identity against it there.)"""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K + act)
    Ad, Wd = _g8(torch.randn(M, K, generator=g)), _g8(torch.randn(N, K, generator=g) / math.sqrt(K), G8_WSCALE)
    bd = (torch.randn(N, generator=g) * (3e4 if act == 0 else 1.0)).cuda()        # act 0: some outputs beyond +-65000
    outs, sats = [], []
    for tile in (3, 4, 1, 2):
        o = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")          # G8 container, 4 bytes per element
        lib.cap_g8_saturations(1)
        _check(lib, lib.cap_op_gemm(SPLIT, _p(Ad), _p(Wd), _p(bd), _p(None), _p(o), M, N, K, act, 0, tile, _stream()))
        torch.cuda.synchronize()
        outs.append(o.view(torch.int32))
        sats.append(lib.cap_g8_saturations(1))
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    assert sats[0] == sats[1] == sats[2] == sats[3], sats              # stored groups only
    if act == 0:
        assert sats[0] > 0


@gpu
@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("shape", [(52000, 776, 64), (1000, 520, 128), (66000, 768, 192), (300, 2304, 768), (50432, 768, 768), (9500, 2040, 128)])
def test_branch_gemm_adds_into_the_residual_stream_in_place(lib, shape, bf16):
    """The ViT branch GEMMs (proj, fc2) add their output to the residual stream in place: C = (acc + bias) + C with C aliasing the
    residual operand (gemm_pp.hip's residual epilogue: a lane loads the 16 bytes it is about to store, one piece ahead).  Against
    the same kernel without the operand followed by one fp32 add (what the add+LayerNorm kernel used to do): the same bits, on
    interior and ragged edge tiles, with the half-tile tail launch ((66000, 768): 774 tiles; (9500, 2040): 304 tiles, ragged both ways) and on the register-staged tile 4;
    and a second residual buffer (not aliased) is left untouched."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    if bf16:
        dt, Ad, Wd = 1, A.to(torch.bfloat16).cuda(), W.to(torch.bfloat16).cuda()
    else:
        dt, Ad, Wd = SPLIT, _g8(A), _g8(W, G8_WSCALE)
    bd = torch.randn(N, generator=g).cuda()
    X0 = torch.randn(M, N, generator=g).cuda()
    plain = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    _check(lib, lib.cap_op_gemm(dt, _p(Ad), _p(Wd), _p(bd), _p(None), _p(plain), M, N, K, 0, 1, 3, _stream()))
    want = (plain + X0).view(torch.int32)
    for tile in (3, 4):
        X = X0.clone()
        _check(lib, lib.cap_op_gemm(dt, _p(Ad), _p(Wd), _p(bd), _p(X), _p(X), M, N, K, 0, 1, tile, _stream()))
        torch.cuda.synchronize()
        assert torch.equal(X.view(torch.int32), want), tile
    R, out = X0.clone(), torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    _check(lib, lib.cap_op_gemm(dt, _p(Ad), _p(Wd), _p(bd), _p(R), _p(out), M, N, K, 0, 1, 3, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), want) and torch.equal(R, X0)


@gpu
@pytest.mark.parametrize("tile", [0, 2, 3, 6])
def test_split_gemm_gelu_into_g8_output(lib, tile):
    M, N, K = 600, 3072, 256
    g = torch.Generator().manual_seed(11)
    A, W, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    Ad, Wd, bd = _g8(A), _g8(W, G8_WSCALE), bias.cuda()
    out = torch.zeros(M, N, dtype=torch.float32, device="cuda")          # G8 container
    _check(lib, lib.cap_op_gemm(SPLIT, _p(Ad), _p(Wd), _p(bd), _p(None), _p(out), M, N, K, 1, 0, tile, _stream()))
    torch.cuda.synchronize()
    want = torch.nn.functional.gelu(A.double() @ W.double().T + bias.double())
    got = torch.from_numpy(g8_decode(out.cpu().numpy())).double()
    assert (got - want).abs().max().item() < 1e-5


@gpu
@pytest.mark.parametrize("dtype", ["f32s", "bf16"])
@pytest.mark.parametrize("shape", [(256, 768, 768, 4), (256, 2304, 768, 1), (256, 768, 3072, 4), (37, 768, 768, 3), (640, 768, 768, 2),
                                   (5, 200, 384, 1), (256, 768, 768, 6), (32, 2560, 2560, 4), (20, 768, 3072, 2), (16, 768, 768, 4)])
def test_decode_rows_kernel_split_k_slices_and_row_invariance(lib, dtype, shape):
    """The decode GEMM kernel (tile 6: 64x64 tile, the block's K range split over its four waves, partial tiles summed in wave
    order): split-K slices sum to the fp64 product of the original operands; and a row's sums do not depend on how many rows
    the launch has or where the row sits - the property the captioner's batch invariance rests on."""
    M, N, K, S = shape
    slab = 32 if dtype == "f32s" else 64
    if K % (slab * S):
        pytest.skip("K does not divide into whole slabs per slice")
    tag = SPLIT if dtype == "f32s" else 1
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + S)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    if dtype == "f32s":
        Ad, Wd = _g8(A), _g8(W, G8_WSCALE)
        scale, tol = 1.0, 1e-5 * math.sqrt(max(K, 256) / 256)          # the slabs come out already divided by the weight scale
        ref = A.double() @ W.double().T
    else:
        Ad, Wd = A.to(torch.bfloat16).cuda(), W.to(torch.bfloat16).cuda()
        scale, tol = 1.0, 2e-4 * math.sqrt(K / 64)
        ref = Ad.double().cpu() @ Wd.double().cpu().T
    part = torch.full((S, M, N), float("nan"), dtype=torch.float32, device="cuda")
    _check(lib, lib.cap_op_gemm_partial(tag, _p(Ad), _p(Wd), _p(part), M, N, K, S, 6, _stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(part).all()
    got = part.double().sum(0).cpu() * scale
    assert (got - ref).abs().max().item() < tol
    # the same rows alone (first 3, and 3 from the middle): bit-identical partial sums
    for r0 in (0, M // 2):
        n = min(3, M - r0)
        sub = torch.full((S, n, N), float("nan"), dtype=torch.float32, device="cuda")
        Asub = Ad[r0:r0 + n].contiguous()
        _check(lib, lib.cap_op_gemm_partial(tag, _p(Asub), _p(Wd), _p(sub), n, N, K, S, 6, _stream()))
        torch.cuda.synchronize()
        assert torch.equal(sub, part[:, r0:r0 + n])


@gpu
def test_split_gemm_rejects_rows_that_break_groups(lib):
    """A G8 row is whole groups of 8 elements: a G8 output 68 columns wide cannot exist (an fp32 output can)."""
    a = torch.zeros(68, 96, device="cuda")
    c = torch.zeros(68, 68, device="cuda")
    assert lib.cap_op_gemm(SPLIT, _p(a), _p(a), _p(None), _p(None), _p(c), 68, 68, 96, 0, 1, 0, _stream()) == 0
    rc = lib.cap_op_gemm(SPLIT, _p(a), _p(a), _p(None), _p(None), _p(c), 68, 68, 96, 0, 0, 0, _stream())
    assert rc != 0 and b"multiples of 8" in lib.cap_last_error()
    torch.cuda.synchronize()


@gpu
def test_device_converts_equal_the_host_encoding(lib):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(37, 200, generator=g) * torch.logspace(-5, 2, 200)
    d = torch.zeros(37, 200, dtype=torch.float32, device="cuda")
    _check(lib, lib.cap_op_convert(SPLIT, _p(x.cuda()), _p(d), x.numel(), _stream()))
    w = torch.zeros(37, 200, dtype=torch.float32, device="cuda")
    _check(lib, lib.cap_op_convert_weight(SPLIT, _p(x.cuda()), _p(w), 37, 200, _stream()))
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint32), g8_encode(x.numpy()).view(np.uint32))
    assert np.array_equal(w.cpu().numpy().view(np.uint32), g8_encode(x.numpy(), G8_WSCALE).view(np.uint32))


@gpu
@pytest.mark.parametrize("D", [128, 768, 1024])
def test_layernorm_writes_g8(lib, D):
    M = 131
    g = torch.Generator().manual_seed(D)
    x = torch.randn(M, D, generator=g) * 3 + 1
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    xd, gd, bd = x.cuda(), gamma.cuda(), beta.cuda()
    out_t = torch.zeros(M, D, dtype=torch.float32, device="cuda")
    out_f = torch.zeros(M, D, dtype=torch.float32, device="cuda")
    _check(lib, lib.cap_op_layernorm(SPLIT, _p(xd), _p(gd), _p(bd), C.c_float(1e-5), _p(out_t), _p(out_f), M, D, _stream()))
    torch.cuda.synchronize()
    # the G8 image holds exactly the fp32 output to 2^-23
    got = g8_decode(out_t.cpu().numpy())
    f = out_f.cpu().numpy()
    assert np.max(np.abs(got - f)) <= np.max(np.abs(f)) * 2.0 ** -21
    assert (out_f.cpu() - torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-5)).abs().max().item() < 2e-5


def _attn_ref(qkv, B, N, H):
    D = H * 64
    x = qkv.double().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (x[0] @ x[1].transpose(-1, -2)) * 0.125
    return (torch.softmax(s, -1) @ x[2]).permute(0, 2, 1, 3).reshape(B * N, D)


@gpu
@pytest.mark.parametrize("N", [17, 197, 257, 577])
@pytest.mark.parametrize("mode", ["f32", "split"])
def test_vit_attention_fp32_mfma(lib, N, mode):
    """fp32 q|k|v in; dtype 0: fp32 context (the fp32 MFMA kernel for 1 / 7 / 9 key blocks, the scalar one otherwise),
    dtype 2: G8 context."""
    B, H = 3, 4
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g) * 1.5
    ctx = torch.full((B * N, H * 64), float("nan"), dtype=torch.float32, device="cuda")
    qd = qkv.cuda()
    _check(lib, lib.cap_op_vit_attention(SPLIT if mode == "split" else 0, _p(qd), _p(ctx), B, N, H, 0, _stream()))
    torch.cuda.synchronize()
    got = ctx.cpu().numpy()
    if mode == "split":
        got = g8_decode(got)
    err = np.abs(got.astype(np.float64) - _attn_ref(qkv, B, N, H).numpy()).max()
    assert err < 1e-5, err


@gpu
def test_decode_attention_writes_g8(lib):
    R, H, n_keys, kv_ld = 24, 12, 197, 197
    g = torch.Generator().manual_seed(9)
    q = torch.randn(R, H * 64, generator=g)
    k = torch.randn(R, H, kv_ld, 64, generator=g)
    v = torch.randn(R, H, kv_ld, 64, generator=g)
    out = torch.zeros(R, H * 64, dtype=torch.float32, device="cuda")
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    _check(lib, lib.cap_op_decode_attention(SPLIT, _p(qd), _p(kd), _p(vd), _p(None), 0, 1, kv_ld, n_keys, _p(out), R, H, 0, _stream()))
    torch.cuda.synchronize()
    s = torch.einsum("rhd,rhkd->rhk", q.view(R, H, 64).double(), k.double()) * 0.125
    want = torch.einsum("rhk,rhkd->rhd", torch.softmax(s, -1), v.double()).reshape(R, H * 64)
    got = torch.from_numpy(g8_decode(out.cpu().numpy())).double()
    assert (got - want).abs().max().item() < 1e-5


@gpu
@pytest.mark.parametrize("N", [5, 17, 64, 65, 197, 224, 225, 257, 577])
def test_vit_attention_split_mfma_on_g8_qkv(lib, N):
    """The split-fp16 MFMA attention kernel (impl 3): G8 q|k|v as the split mode's qkv GEMM writes them, G8 context out;
    both products as hi.lo + lo.hi + hi.hi on the fp16 pipe - fp32-grade against an fp64 reference of the fp32 values."""
    B, H = 3, 4
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g) * 1.5
    qd = _g8(qkv)
    ctx = torch.full((B * N, H * 64), float("nan"), dtype=torch.float32, device="cuda")
    _check(lib, lib.cap_op_vit_attention(SPLIT, _p(qd), _p(ctx), B, N, H, 3, _stream()))
    torch.cuda.synchronize()
    got = g8_decode(ctx.cpu().numpy())
    assert np.isfinite(got).all()
    err = np.abs(got.astype(np.float64) - _attn_ref(qkv, B, N, H).numpy()).max()
    assert err < 1e-5, err


@gpu
@pytest.mark.parametrize("N,B,H", [(197, 30, 12), (197, 23, 12), (224, 25, 12), (193, 300, 1)])
def test_vit_attention_persistent_matches_per_unit_kernel(lib, N, B, H):
    """193..224 tokens with more (image, head) units than CUs: the persistent kernel (one workgroup per CU walks the units, the
    next unit's K / V / q loads under the current unit's arithmetic) against the one-workgroup-per-unit kernel (impl 5): the same
    bits - unit counts that are and are not multiples of the CU count."""
    g = torch.Generator().manual_seed(N + B)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g) * 1.5
    qd = _g8(qkv)
    outs = []
    for impl in (5, 3):                          # per unit; persistent
        ctx = torch.full((B * N, H * 64), float("nan"), dtype=torch.float32, device="cuda")
        _check(lib, lib.cap_op_vit_attention(SPLIT, _p(qd), _p(ctx), B, N, H, impl, _stream()))
        torch.cuda.synchronize()
        outs.append(ctx)
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    got = g8_decode(outs[1].cpu().numpy())
    assert np.isfinite(got).all()
    err = np.abs(got.astype(np.float64) - _attn_ref(qkv, B, N, H).numpy()).max()
    assert err < 1e-5, err


@gpu
def test_split_mode_refuses_out_of_range_weights_instead_of_clipping_them():
    """Weights travel as fp16 halves of 4096 w: |w| > 15.87 does not fit.  cap_load_weight names the tensor and fails; nothing
    is clipped silently (the mode was the plugin default before it could say so)."""
    from embodied_captioning_amd._native import CaptionerHipError
    from embodied_captioning_amd.config import BlipArch
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import procedural_blip_state_dict
    arch = BlipArch.tiny()
    sd = dict(procedural_blip_state_dict(arch, 3))
    name = "vision_model.encoder.layers.1.mlp.fc1.weight"
    for bad in (20.0, -16.0, float("nan"), float("inf")):
        eng = CaptionerEngine(arch, dtype="f32s", max_batch=2, max_beams=1, max_len=8)
        w = sd[name].clone()
        w[3, 5] = bad
        with pytest.raises(CaptionerHipError, match="fc1.weight has max"):
            eng.load_state_dict({**sd, name: w})
        eng.close()
    for dtype in ("f32s", "f32", "bf16"):                  # 15.8 is inside; the other modes take anything
        eng = CaptionerEngine(arch, dtype=dtype, max_batch=2, max_beams=1, max_len=8)
        w = sd[name].clone()
        w[3, 5] = 15.8 if dtype == "f32s" else 20.0
        eng.load_state_dict({**sd, name: w})
        eng.close()


@gpu
def test_split_mode_counts_the_activations_it_clamps(lib):
    """Activations beyond +-65000 at a GEMM input are clamped to fp16's range - and counted (cap_g8_saturations), so leaving
    the envelope in which the mode is fp32-grade is visible.  A LayerNorm with a huge gamma and a GEMM with a G8 output whose
    results pass 65000 both bump the counter; values inside the range never do; a whole golden generate stays at zero."""
    assert lib.cap_g8_saturations(1) >= 0
    M, D = 8, 256
    x = torch.randn(M, D, generator=torch.Generator().manual_seed(0)).cuda()
    out = torch.empty(M, D, dtype=torch.float32, device="cuda")
    ones, zeros = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    _check(lib, lib.cap_op_layernorm(SPLIT, _p(x), _p(ones), _p(zeros), C.c_float(1e-5), _p(out), None, M, D, _stream()))
    assert lib.cap_g8_saturations(0) == 0
    big = ones * 1e5                                           # |LayerNorm(x)| * 1e5 passes 65000 wherever |x_hat| > 0.65
    _check(lib, lib.cap_op_layernorm(SPLIT, _p(x), _p(big), _p(zeros), C.c_float(1e-5), _p(out), None, M, D, _stream()))
    n = lib.cap_g8_saturations(0)
    assert 0 < n <= M * D // 4
    got = g8_decode(out.cpu().numpy())
    assert np.abs(got).max() <= 65000.0 + 64                   # clamped to fp16's range (65000 rounds to a 32-step grid)
    assert lib.cap_g8_saturations(1) == n and lib.cap_g8_saturations(0) == 0       # reset
    # GEMM with a G8 output: A = 300 everywhere, W = 1 -> C = 300 K = 76 800 > 65000
    Mg, Ng, K = 64, 64, 256
    A = _g8(torch.full((Mg, K), 300.0))
    W = _g8(torch.ones(Ng, K), G8_WSCALE)
    Cg = torch.empty(Mg, Ng, dtype=torch.float32, device="cuda")
    _check(lib, lib.cap_op_gemm(SPLIT, _p(A), _p(W), None, None, _p(Cg), Mg, Ng, K, 0, 0, 2, _stream()))
    assert lib.cap_g8_saturations(1) == Mg * Ng // 4
    _check(lib, lib.cap_op_gemm(SPLIT, _p(A), _p(W), None, None, _p(Cg), Mg, Ng, K, 0, 1, 2, _stream()))   # fp32 output: no clamp
    assert lib.cap_g8_saturations(1) == 0 and abs(float(Cg[0, 0]) - 76800.0) < 1e-2
    # a whole generate on the golden's weights stays inside the envelope
    from _util import golden_inputs
    from embodied_captioning_amd.engine import CaptionerEngine
    g, meta, arch, sd, px = golden_inputs("blip_tiny")
    eng = CaptionerEngine(arch, dtype="f32s", max_batch=meta["batch"], max_beams=1, max_len=meta["max_length"])
    eng.load_state_dict(sd)
    eng.generate(px.cuda(), max_length=meta["max_length"])
    assert eng.saturations(reset=True) == 0
    eng.close()


def _kv16_ref(x):
    """numpy restatement of the KV16 quantisation (common.h): per 64-wide row, scale = max|x| / 32767, q = rint(x * (32767 / max|x|))."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 64)
    am = np.abs(x).max(axis=1)
    inv = np.where(am > 0, np.float32(32767.0) / np.where(am > 0, am, 1).astype(np.float32), np.float32(0)).astype(np.float32)
    q = np.rint(x * inv[:, None]).astype(np.int16)
    return q, (am * np.float32(1.0 / 32767.0)).astype(np.float32)


def _kv16_unpack(raw, rows):
    """bytes of a KV16 block -> (int16 [rows, 64], fp32 scales [rows]); groups of 32 rows = 32 x 128 B then 32 scales."""
    g = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 4224)
    q = g[:, :4096].reshape(-1, 128).view(np.int16)[:rows]
    sc = g[:, 4096:].reshape(-1, 128).view(np.float32).reshape(-1)[:rows]
    return q.copy(), sc.copy()


@gpu
@pytest.mark.parametrize("dtype", [0, SPLIT])
@pytest.mark.parametrize("n_keys,beams", [(197, 1), (197, 3), (255, 5), (577, 1), (40, 2)])
def test_kv16_cross_attention_cache(lib, dtype, n_keys, beams):
    """The split mode's cross-attention K/V cache is KV16: 64 int16 and one fp32 scale per head row, rows in groups of 32.  (1) the
    pack kernel writes exactly the numpy restatement's integers and scales; (2) the decode attention kernels on a KV16 cache -
    one row per block (greedy) and the shared-block kernel (2-5 beams) - give the fp64 attention over the DEQUANTISED values to
    fp32 rounding; (3) against the unquantised cache the context moves by ~2^-15 of the row's largest element."""
    Bimg, H, kv_ld = 3, 2, n_keys + 3
    R = Bimg * beams
    g = torch.Generator().manual_seed(n_keys + beams)
    q = torch.randn(R, H * 64, generator=g)
    K = torch.randn(Bimg, H, kv_ld, 64, generator=g)
    V = torch.randn(Bimg, H, kv_ld, 64, generator=g) * torch.logspace(-1, 1, 64)          # two decades inside a row
    V[0, 0, 5] = 0.0                                                                       # an all-zero row quantises to zeros
    rows = Bimg * H * kv_ld
    nbytes = (rows + 31) // 32 * 4224

    def pack(x):
        d = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        _check(lib, lib.cap_op_pack_kv16(_p(x.cuda().contiguous()), _p(d), rows, _stream()))
        return d

    Kp, Vp = pack(K), pack(V)
    torch.cuda.synchronize()
    # (1) integers and scales
    deq = {}
    for name, x, pk in (("k", K, Kp), ("v", V, Vp)):
        qi, sc = _kv16_unpack(pk.cpu().numpy().tobytes(), rows)
        qr, sr = _kv16_ref(x.numpy())
        assert np.array_equal(qi, qr) and np.array_equal(sc, sr)
        assert np.abs(qi).max() <= 32767
        deq[name] = torch.from_numpy(qi.astype(np.float64) * sc.astype(np.float64)[:, None]).view(Bimg, H, kv_ld, 64)
    # (2) fp64 attention over the dequantised values
    qd = q.cuda()
    out16 = torch.full((R, H * 64), float("nan"), dtype=torch.float32, device="cuda")
    _check(lib, lib.cap_op_decode_attention(dtype, _p(qd), _p(Kp), _p(Vp), None, 0, beams, kv_ld, n_keys, _p(out16), R, H, 16, _stream()))
    torch.cuda.synchronize()
    got = torch.from_numpy(g8_decode(out16.cpu().numpy())).double() if dtype == SPLIT else out16.cpu().double()

    def ref(Kx, Vx):
        qh = q.double().view(R, H, 64)
        img = torch.arange(R) // beams
        s = torch.einsum("rhd,rhkd->rhk", qh, Kx.double()[img][:, :, :n_keys]) / 8.0
        return torch.einsum("rhk,rhkd->rhd", torch.softmax(s, -1), Vx.double()[img][:, :, :n_keys]).reshape(R, H * 64)

    want = ref(deq["k"], deq["v"])
    assert (got - want).abs().max().item() < 2e-6 * max(1.0, want.abs().max().item())
    # (3) distance to the unquantised cache: each V element is off by at most half a step of its row, 2^-16 of the row maximum
    exact = ref(K, V)
    assert (got - exact).abs().max().item() < 3e-4 * max(1.0, exact.abs().max().item())
    # a KV16 cache with an ancestry table or a short history is refused
    anc = torch.zeros(R, kv_ld, dtype=torch.int32, device="cuda")
    assert lib.cap_op_decode_attention(dtype, _p(qd), _p(Kp), _p(Vp), _p(anc), kv_ld, 1, kv_ld, n_keys, _p(out16), R, H, 16, _stream()) != 0
    assert b"KV16" in lib.cap_last_error()


@gpu
@pytest.mark.parametrize("geom", [(3, 197, 12, 2, 768), (2, 50, 2, 3, 128), (5, 255, 12, 1, 768), (1, 197, 12, 1, 768), (9, 33, 1, 2, 64)])
def test_cross_kv_gemm_writes_the_kv16_cache_of_its_fp32_output(lib, geom):
    """The cross-K/V GEMM's KV16 epilogue (gemm_pp.hip: row maximum, scale, int16, 128-byte rows through the strips, groups of 32
    rows) against the same GEMM with fp32 rows packed by the stand-alone kernel: identical integers and scales in every (layer,
    k | v) block - ragged last group, rows of two images in one 16-row block, one image (72 tiles), one head."""
    n_img, tokens, heads, layers, K = geom
    M, N = n_img * tokens, layers * 2 * heads * 64
    g = torch.Generator().manual_seed(M + N)
    A = _g8(torch.randn(M, K, generator=g))
    W = _g8(torch.randn(N, K, generator=g) / math.sqrt(K), G8_WSCALE)
    bias = torch.randn(N, generator=g).cuda()
    rows = n_img * heads * tokens
    blk = (rows + 31) // 32 * 4224
    c16 = torch.zeros(layers * 2 * blk, dtype=torch.uint8, device="cuda")
    c32 = torch.full((layers * 2 * rows, 64), float("nan"), dtype=torch.float32, device="cuda")
    _check(lib, lib.cap_op_gemm_crosskv(SPLIT, _p(A), _p(W), _p(bias), _p(c16), n_img, tokens, heads, layers, K, 1, _stream()))
    _check(lib, lib.cap_op_gemm_crosskv(SPLIT, _p(A), _p(W), _p(bias), _p(c32), n_img, tokens, heads, layers, K, 0, _stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(c32).all()
    ref = c32.cpu().numpy().reshape(layers * 2, rows, 64)
    raw = c16.cpu().numpy().reshape(layers * 2, blk)
    for b in range(layers * 2):
        qi, sc = _kv16_unpack(raw[b].tobytes(), rows)
        qr, sr = _kv16_ref(ref[b])
        assert np.array_equal(sc, sr), b
        assert np.array_equal(qi, qr), b


@gpu
@pytest.mark.parametrize("dtype", [0, SPLIT])
@pytest.mark.parametrize("N,hd", [(257, 88), (33, 80), (100, 96), (197, 40), (300, 72), (31, 8)])
def test_wide_head_fp32_mfma_attention_fp32_and_g8_context(lib, dtype, N, hd):
    """Heads that are not 64 wide in the fp32 / split modes (BLIP-2's ViT-g/14: 88 at 257 tokens): the exact-product fp32 MFMA
    kernel with the keys walked in chunks of 96 and an online softmax, fp32 q | k | v in, context out as fp32 (CAP_F32) or G8
    (CAP_F32_SPLIT) - against a float64 reference and against the VALU kernel (impl 1) on the same inputs."""
    B, H = 2, 3
    g = torch.Generator().manual_seed(N * 7 + hd)
    qkv = torch.randn(B * N, 3 * H * hd, generator=g) * 1.5
    qd = qkv.cuda()
    a = torch.full((B * N, H * hd), float("nan"), dtype=torch.float32, device="cuda")
    b = torch.full_like(a, float("nan"))
    _check(lib, lib.cap_op_vit_attention_hd(dtype, _p(qd), _p(a), B, N, H, hd, 0, _stream()))
    _check(lib, lib.cap_op_vit_attention_hd(dtype, _p(qd), _p(b), B, N, H, hd, 1, _stream()))
    torch.cuda.synchronize()
    x = qkv.double().view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    sc = (x[0] @ x[1].transpose(-1, -2)) / hd ** 0.5
    ref = (torch.softmax(sc, -1) @ x[2]).permute(0, 2, 1, 3).reshape(B * N, H * hd)
    dec = (lambda t: torch.from_numpy(g8_decode(t.cpu().numpy()))) if dtype == SPLIT else (lambda t: t.cpu())
    assert torch.isfinite(dec(a)).all()
    assert (dec(a).double() - ref).abs().max().item() < 2e-5
    assert (dec(a) - dec(b)).abs().max().item() < 2e-5
