"""GPU: BLIP-2 `load_in_8bit` (reference captioner/models/blip2/blip2.py:19-22) - int8 Linear weights as bitsandbytes stores them,
bf16 activations.  PARITY UNPINNED against bitsandbytes itself (absent here): these tests hold the HIP path to the restatement
(oracle/blip2_ref.py::quantize_int8_rowwise / int8_state_dict) - the quantiser bit for bit, the weight-streaming GEMM against
the integer arithmetic in fp32, and a whole generate against the restatement computing with the dequantised weights."""
import ctypes as C

import numpy as np
import pytest
import torch

from _util import token_parity

pytestmark = pytest.mark.gpu


def _lib():
    from embodied_captioning_amd import _native as N
    return N, N.load_library()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pack(lib, N, w):
    rows, cols = w.shape
    packed = torch.empty(rows * cols, dtype=torch.uint8, device="cuda")
    scale = torch.empty(rows, dtype=torch.float32, device="cuda")
    N.check(lib.cap_op_quant_i8_pack(C.c_void_p(w.data_ptr()), C.c_void_p(packed.data_ptr()), C.c_void_p(scale.data_ptr()), rows, cols, _stream()),
            "cap_op_quant_i8_pack")
    return packed, scale


def _unpack(packed, rows, cols):
    """fragment order -> [rows, cols] int8: block (t, s) of 1 KiB, lane r + 16 g owns 16 bytes = k-step 0 (8) then k-step 1 (8)"""
    b = packed.view(torch.int8).cpu().view(rows // 16, cols // 64, 4, 16, 2, 8)           # t, s, g, r, ks, e
    return b.permute(0, 3, 1, 4, 2, 5).reshape(rows, cols)                               # (t, r), (s, ks, g, e)


@pytest.mark.parametrize("rows,cols", [(32, 64), (96, 256), (2560, 2560), (256, 10240)])
def test_quantiser_is_the_restatement_bit_for_bit(rows, cols):
    from oracle import blip2_ref as R
    N, lib = _lib()
    g = torch.Generator().manual_seed(rows + cols)
    w = torch.randn(rows, cols, generator=g) * 0.03
    w[3] = 0.0                                                             # an all-zero row quantises to zeros with scale 0
    w[5, 7] = 0.9                                                          # an outlier sets its row's scale
    w[6, :4] = torch.tensor([0.5, 1.5, 2.5, -0.5]) * (w[6].abs().max() / 127.0)          # ties: round-half-even
    packed, scale = _pack(lib, N, w.cuda())
    q, sc = R.quantize_int8_rowwise(w)
    assert torch.equal(scale.cpu(), sc)
    assert torch.equal(_unpack(packed, rows, cols), q)


SHAPES = [(7680, 2560), (2560, 2560), (10240, 2560), (2560, 10240), (768, 256), (256, 512)]


@pytest.mark.parametrize("N_,K", SHAPES)
@pytest.mark.parametrize("M", [1, 5, 16, 17, 33, 70, 1056])      # 33 .. : the kernel that walks the row groups (a prompt pass)
def test_int8_weight_stream_gemm_against_integer_arithmetic(N_, K, M):
    Nn, lib = _lib()
    g = torch.Generator().manual_seed(N_ * 7 + K + M)
    w = (torch.randn(N_, K, generator=g) * 0.02).cuda()
    a = (torch.randn(M, K, generator=g)).cuda().bfloat16()
    bias = (torch.randn(N_, generator=g) * 0.1).cuda()
    packed, scale = _pack(lib, Nn, w)
    q = _unpack(packed, N_, K).cuda().float()
    exact = (a.double() @ q.double().T) * scale.double()[None, :]                       # integers x bf16 values: exact in fp64
    # finished form: bias + ReLU -> bf16
    S1 = lib.cap_op_gemm_skinny_i8_slices(N_, K, 1)
    if S1 >= 1:
        out = torch.empty(M, N_, dtype=torch.bfloat16, device="cuda")
        rc = lib.cap_op_gemm_skinny_i8(C.c_void_p(a.data_ptr()), C.c_void_p(packed.data_ptr()), C.c_void_p(scale.data_ptr()), C.c_void_p(bias.data_ptr()),
                                       2, C.c_void_p(out.data_ptr()), None, M, N_, K, _stream())
        assert rc == 1, Nn.last_error()
        want = torch.relu(exact + bias.double())
        err = (out.double() - want).abs()
        assert (err <= want.abs() * 2.0 ** -8 + 1e-3).all(), err.max().item()           # one bf16 rounding of an fp32-grade sum
    # slice sums for the reduce + LayerNorm consumer
    S = lib.cap_op_gemm_skinny_i8_slices(N_, K, 0)
    assert S >= 1
    part = torch.full((S, M, N_), float("nan"), dtype=torch.float32, device="cuda")
    rc = lib.cap_op_gemm_skinny_i8(C.c_void_p(a.data_ptr()), C.c_void_p(packed.data_ptr()), C.c_void_p(scale.data_ptr()), None, 0, None,
                                   C.c_void_p(part.data_ptr()), M, N_, K, _stream())
    assert rc == S, Nn.last_error()
    got = part.double().sum(0)
    mag = (a.double().abs() @ q.double().abs().T) * scale.double()[None, :]
    assert ((got - exact).abs() <= mag * 1e-6 + 1e-6).all()                              # fp32 accumulation of exact products
    # a row's sums do not depend on the row count (one 16-row tile or 32-row groups)
    if M > 1:
        p1 = torch.empty((S, 1, N_), dtype=torch.float32, device="cuda")
        a1 = a[M - 1:].contiguous()
        assert lib.cap_op_gemm_skinny_i8(C.c_void_p(a1.data_ptr()), C.c_void_p(packed.data_ptr()), C.c_void_p(scale.data_ptr()), None, 0, None,
                                         C.c_void_p(p1.data_ptr()), 1, N_, K, _stream()) == S
        assert torch.equal(p1[:, 0], part[:, M - 1])


def _small_arch():
    from embodied_captioning_amd.config import Blip2Arch
    # OPT widths the int8 weight stream takes (multiples of 256), everything else fixture-sized
    return Blip2Arch(image_size=28, patch_size=14, v_hidden=192, v_layers=2, v_heads=8, v_mlp=256, q_hidden=128, q_layers=2, q_heads=2,
                     q_ffn=256, num_query_tokens=8, t_hidden=256, t_layers=3, t_heads=4, t_ffn=512, vocab=512, max_pos=64, eos=3,
                     image_token=511, max_new_tokens=8)


@pytest.mark.parametrize("B", [1, 3, 20])
def test_int8_generate_against_restatement_with_dequantised_weights(B):
    """bf16 activations + int8 weights against the fp32 restatement computing with the SAME quantised weights: the bf16 mode's bar
    (tests/test_blip2_gpu.py: logits 0.1, tokens equal wherever the restatement's margin exceeds the tolerance).  B = 1: the one-tile
    decode kernels and a 9-row prompt; 20: 32-row groups in the steps, 180 rows in the prompt."""
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import procedural_blip2_state_dict, synthetic_pixels
    from oracle import blip2_ref as R
    a = _small_arch()
    sd = procedural_blip2_state_dict(a, 5, eos_boost=0.3)
    px = synthetic_pixels(B, a.image_size, seed=5)
    sdq = R.int8_state_dict(sd)
    changed = [k for k in sd if not torch.equal(sd[k], sdq[k])]
    assert len(changed) == 6 * a.t_layers + 4 * a.v_layers + 1                            # q k v o fc1 fc2 | qkv proj fc1 fc2 | language_projection
    ref = R.greedy_generate(sdq, a, px)
    eng = CaptionerEngine(a, dtype="bf16", max_batch=B, max_beams=1, max_len=a.max_new_tokens, weight_int8=True)
    eng.load_state_dict(sd)
    out = eng.generate(px.cuda(), max_length=a.max_new_tokens, output_logits=True)
    lg = out["logits"].cpu()
    rl = torch.stack(ref["logits"], 0)
    assert (lg[0] - rl[0]).abs().max().item() < 0.1
    new = ref["sequences"][:, a.num_query_tokens + 1:].numpy()
    want = np.full((B, a.max_new_tokens), a.pad, dtype=np.int64)
    want[:, : new.shape[1]] = new
    seq = out["sequences"].cpu().numpy()
    margin = np.zeros((a.max_new_tokens, B), dtype=np.float32)                             # top-1 / top-2 gap of the restatement per step
    for t, l in enumerate(ref["logits"]):
        top2 = torch.topk(l, 2, dim=-1).values
        margin[t] = (top2[:, 0] - top2[:, 1]).numpy()
    full = np.concatenate([np.full((B, 1), a.bos), seq], 1)
    exact, diverged, bad = token_parity(full, np.concatenate([np.full((B, 1), a.bos), want], 1), margin, 0.05)
    assert bad is None, bad
    # and the int8 path is not the bf16 path on other weights: the unquantised restatement differs by more than the int8 one
    ref_fp = R.greedy_generate(sd, a, px)
    assert (lg[0] - ref_fp["logits"][0]).abs().max().item() > (lg[0] - rl[0]).abs().max().item()
    eng.close()


def test_int8_same_captions_alone_and_in_a_batch():
    """Batch invariance of the mode.  The decode steps' slice plan and every row's sums depend on (N, K) only.  The prompt pass runs
    on the weight-streaming kernels up to 4 crops per call and on the tiled GEMM beyond (csrc/captioner.hip::kI8SkinnyPromptCrops):
    inside each range a crop's tokens AND logits are the same bits whatever batch it is in; across the two ranges the prompt's sums
    are formed in a different order - equal to bf16 rounding noise (tokens equal wherever the margin allows, logits within 0.05)."""
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import procedural_blip2_state_dict, synthetic_pixels
    a = _small_arch()
    sd = procedural_blip2_state_dict(a, 6, eos_boost=0.3)
    px = synthetic_pixels(20, a.image_size, seed=6).cuda()
    eng = CaptionerEngine(a, dtype="bf16", max_batch=20, max_beams=1, max_len=a.max_new_tokens, weight_int8=True)
    eng.load_state_dict(sd)

    def gen(rows):
        o = eng.generate(rows, max_length=a.max_new_tokens, output_logits=True)
        return o["sequences"].clone(), o["lengths"].clone(), o["logits"].clone()

    def same(x, i, y, j):
        n = int(x[1][i])
        return torch.equal(x[0][i], y[0][j]) and int(y[1][j]) == n and torch.equal(x[2][:n, i], y[2][:n, j])
    four = gen(px[:4])                                   # weight-stream prompt pass
    for b in (0, 3):
        assert same(gen(px[b:b + 1]), 0, four, b)
    twenty = gen(px)                                     # tiled prompt pass
    five = gen(px[7:12])
    for r in range(5):
        assert same(five, r, twenty, 7 + r)
    one = gen(px[2:3])                                   # across the ranges
    n = int(one[1][0])
    assert (one[2][:n, 0] - twenty[2][:n, 2]).abs().max().item() < 0.05
    top2 = torch.topk(one[2][:n, 0], 2, dim=-1).values
    for t in range(n):
        if (top2[t, 0] - top2[t, 1]).item() > 0.1:
            assert int(one[0][0, t]) == int(twenty[0][2, t])
        else:
            break
    eng.close()


def test_int8_production_width_against_restatement():
    """OPT-2.7b's own widths (2560 / 10240, vocabulary 50272; ViT-g 1408, two layers per tower - the geometry of
    tests/golden/blip2_width.npz) in the reference's load mode, against the restatement with the quantised weights."""
    from test_blip2_cpu import load_blip2
    from embodied_captioning_amd.engine import CaptionerEngine
    from oracle import blip2_ref as R
    g, meta, a, sd, px = load_blip2("blip2_width")
    B = meta["batch"]
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = R.greedy_generate(R.int8_state_dict(sd), a, px)
    eng = CaptionerEngine(a, dtype="bf16", max_batch=B, max_beams=1, max_len=a.max_new_tokens, weight_int8=True)
    eng.load_state_dict(sd)
    out = eng.generate(px.cuda(), max_length=a.max_new_tokens, output_logits=True)
    lg = out["logits"].cpu()
    assert (lg[0] - ref["logits"][0]).abs().max().item() < 0.15
    top = ref["logits"][0].argmax(-1)
    gap = torch.topk(ref["logits"][0], 2, dim=-1).values
    for b in range(B):
        if (gap[b, 0] - gap[b, 1]).item() > 0.15:
            assert int(out["sequences"][b, 0]) == int(top[b])
    eng.close()


def test_wrapper_load_in_8bit_through_the_factory():
    """`captioner.load_in_8bit: true` (what the reference's BLIP-2 wrapper hard-codes, blip2.py:19-22) through select_captioner."""
    from PIL import Image
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    from embodied_captioning_amd.captioner.utils.utils_captioner import select_captioner
    rng = np.random.default_rng(0)
    im = Image.fromarray(rng.integers(0, 256, size=(50, 41, 3), dtype=np.uint8), "RGB")
    cfg = Configuration(arch_name="blip2", model_name="procedural-blip2-small:11:0.5", height=224, width=224, load_in_8bit=True,
                        torch_dtype="float16").captioner
    model = select_captioner(cfg).eval()
    assert model.engine.weight_int8 and model.engine.dtype == "bf16"
    out = model(im)
    assert isinstance(out["text"], str) and 1 <= len(out["logits"]) <= model.arch.max_new_tokens
    assert out["logits"][0].shape == (1, model.arch.vocab)
    assert torch.isfinite(model.compute_perplexity())
    # a geometry the int8 weight stream does not take is refused at creation with the reason
    cfg = Configuration(arch_name="blip2", model_name="procedural-blip2-tiny:11:0.5", height=224, width=224, load_in_8bit=True).captioner
    with pytest.raises(Exception, match="weight_int8"):
        select_captioner(cfg)


@pytest.mark.parametrize("mode", ["f32s", "bf16", "int8"])
def test_blip2_wrapper_streams_same_captions(mode):
    """`captioner.streams: 3` on the BLIP-2 wrapper (plain and `load_in_8bit`): generate_batch over the engine pool returns the one-engine
    wrapper's sequences, lengths and texts (every micro-batch is computed by exactly the kernels of a single engine)."""
    from PIL import Image
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    from embodied_captioning_amd.captioner.utils.utils_captioner import select_captioner
    rng = np.random.default_rng(9)
    ims = [Image.fromarray(rng.integers(0, 256, size=(44 + i, 60, 3), dtype=np.uint8), "RGB") for i in range(11)]
    kw = dict(arch_name="blip2", model_name="procedural-blip2-small:4:0.4", height=224, width=224, batch_size=3)
    q8 = mode == "int8"
    kw.update(dict(load_in_8bit=True) if q8 else dict(dtype=mode))
    one = select_captioner(Configuration(**kw).captioner).eval()
    many = select_captioner(Configuration(streams=3, **kw).captioner).eval()
    # dynamic batching: on by default (4 micro-batches per pass) - except load_in_8bit with micro-batches of up to 4 crops, where a
    # merged pass would move the prompt pass to the other kernel range
    assert many.pool is not None and len(many.pool) == 3 and many.coalesce_rows == (0 if q8 else 12) and many.pool.engines[0].weight_int8 == q8
    a, b = one.generate_batch(ims), many.generate_batch(ims)
    if not q8:
        assert isinstance(many.pool.last_coalesce, list) and any(len(g) > 1 for g in many.pool.last_coalesce)
    assert torch.equal(a["sequences"], b["sequences"]) and torch.equal(a["lengths"], b["lengths"]) and a["texts"] == b["texts"]
    if q8:                                   # micro-batches of 5: merged or not, every prompt pass is in the tiled-GEMM range - same bits
        kw5 = dict(kw, batch_size=5)
        one5 = select_captioner(Configuration(**kw5).captioner).eval()
        many5 = select_captioner(Configuration(streams=3, **kw5).captioner).eval()
        assert many5.coalesce_rows == 20
        a5, b5 = one5.generate_batch(ims + ims), many5.generate_batch(ims + ims)
        assert isinstance(many5.pool.last_coalesce, list) and any(len(g) > 1 for g in many5.pool.last_coalesce)
        assert torch.equal(a5["sequences"], b5["sequences"]) and torch.equal(a5["lengths"], b5["lengths"])
