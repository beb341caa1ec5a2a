"""GPU: the small-batch decode path (csrc/decode_small.hip; images x beams <= 16 rows - the reference calls its captioner with
ONE crop per call: captioner/models/coca/coca.py:27-33, blip2/blip2.py:24-29, agents/goal_exploration/goal_exploration.py:95-105;
BASELINE config 1 is 8 crops) against the batch path, the HF goldens and itself.

The bar: the fused kernels form the same sums in the same order as the batch kernels, so tokens AND per-step logits have the
same bits on either path (`torch.equal`, no tolerance), a frame decodes to the same bits alone and in a batch of 8, and config
1's batch (tests/golden/blip_base.npz IS that batch) is token-identical to HF through this path."""
import numpy as np
import pytest
import torch

from _util import golden_inputs, pad_to

pytestmark = pytest.mark.gpu


def _engine(arch, dtype, batch, beams, max_len, path="auto"):
    from embodied_captioning_amd.engine import CaptionerEngine
    eng = CaptionerEngine(arch, dtype=dtype, max_batch=batch, max_beams=beams, max_len=max_len)
    eng.set_decode_path(path)
    return eng


def _live_mask(seq, arch, steps):
    """[steps, B] True where row b's logits of step t are still looked at (the row had not ended before step t)."""
    B, L = seq.shape
    live = np.ones((steps, B), dtype=bool)
    for b in range(B):
        row = list(seq[b, 1:])
        if arch.eos in row:
            live[row.index(arch.eos) + 1:, b] = False
    return live


@pytest.mark.parametrize("dtype", ["f32s", "bf16"])
@pytest.mark.parametrize("name", ["blip_tiny", "blip_tiny_eos", "blip_base"])
def test_small_path_has_the_bits_of_the_batch_path(name, dtype):
    g, meta, arch, sd, px = golden_inputs(name)
    B, L = meta["batch"], meta["max_length"]
    assert B <= 16
    outs = {}
    for path in ("small", "batch"):
        eng = _engine(arch, dtype, B, 1, L, path)
        eng.load_state_dict(sd)
        outs[path] = eng.generate(px.cuda(), num_beams=1, max_length=L, output_logits=True)
        assert eng.last_decode_path == path
        if dtype == "f32s":
            assert eng.saturations() == 0
        eng.close()
    a, b = outs["small"], outs["batch"]
    assert torch.equal(a["sequences"], b["sequences"])
    assert torch.equal(a["lengths"], b["lengths"])
    live = torch.from_numpy(_live_mask(a["sequences"].cpu().numpy(), arch, L - 1))
    la, lb = a["logits"].cpu(), b["logits"].cpu()
    assert torch.equal(la[live], lb[live]), float((la[live] - lb[live]).abs().max())


@pytest.mark.parametrize("dtype", ["f32s", "bf16"])
@pytest.mark.parametrize("name", ["blip_tiny", "blip_base"])
def test_small_path_beams_have_the_bits_of_the_batch_path(name, dtype):
    g, meta, arch, sd, px = golden_inputs(name)
    L, K = meta["max_length"], meta["beams"]
    B = min(meta["batch"], 16 // K)
    outs = {}
    for path in ("small", "batch"):
        eng = _engine(arch, dtype, B, K, L, path)
        eng.load_state_dict(sd)
        outs[path] = eng.generate(px[:B].cuda(), num_beams=K, max_length=L)
        assert eng.last_decode_path == path
        eng.close()
    a, b = outs["small"], outs["batch"]
    assert torch.equal(a["sequences"], b["sequences"])
    assert torch.equal(a["sequences_scores"], b["sequences_scores"])


@pytest.mark.parametrize("dtype", ["f32s", "f32"])
def test_config1_batch_of_8_is_token_identical_to_hf(dtype):
    """BASELINE config 1: 8 frames, greedy, max_length 20 - tests/golden/blip_base.npz is that batch through the real HF loop.
    "f32s" takes the small-batch path by row count; the exact-product "f32" mode has no fused kernels and stays on the batch
    path (same tokens)."""
    g, meta, arch, sd, px = golden_inputs("blip_base")
    B, L = meta["batch"], meta["max_length"]
    assert B == 8
    eng = _engine(arch, dtype, B, 1, L)
    eng.load_state_dict(sd)
    out = eng.generate(px.cuda(), num_beams=1, max_length=L, output_logits=True)
    assert eng.last_decode_path == ("small" if dtype == "f32s" else "batch")
    seq = out["sequences"].cpu().numpy()
    ref = pad_to(g["greedy_sequences"], L, arch.pad)
    assert np.array_equal(seq, ref), (seq, ref)
    T = g["greedy_top8_ids"].shape[0]
    top = torch.topk(out["logits"].cpu()[:T], 8, dim=-1)
    live = _live_mask(ref, arch, T)
    assert np.array_equal(top.indices.numpy()[live], g["greedy_top8_ids"][live])
    np.testing.assert_allclose(top.values.numpy()[live], g["greedy_top8_vals"][live], rtol=0, atol=1e-3)
    eng.close()


@pytest.mark.parametrize("dtype", ["f32s", "bf16"])
def test_one_frame_alone_equals_its_row_in_a_batch_of_8_and_of_64(dtype):
    """Batch invariance across the path switch: frame 0 alone (small path), inside config 1's 8 (small path) and inside 64 frames
    (batch path) - same tokens, same logits bits while the row is open."""
    g, meta, arch, sd, px = golden_inputs("blip_base64")
    L = meta["max_length"]
    res = {}
    for B in (1, 8, 16, 64):
        eng = _engine(arch, dtype, B, 1, L)
        eng.load_state_dict(sd)
        out = eng.generate(px[:B].cuda(), num_beams=1, max_length=L, output_logits=True)
        assert eng.last_decode_path == ("small" if B <= 16 else "batch")
        res[B] = (out["sequences"][0].cpu(), out["logits"][:, 0].cpu())
        eng.close()
    n = _live_mask(res[1][0][None].numpy(), arch, L - 1)[:, 0]
    for B in (8, 16, 64):
        assert torch.equal(res[B][0], res[1][0]), B
        assert torch.equal(res[B][1][n], res[1][1][n]), B


@pytest.mark.parametrize("rows", [1, 2, 3, 5, 13, 16])
def test_ragged_row_counts(rows):
    """Row counts that do not fill the 16-row MFMA block or the waves of the LayerNorm prologue: row r of a batch of `rows` equals
    row r of the 16-row batch (tokens and live logits)."""
    g, meta, arch, sd, px = golden_inputs("blip_tiny")
    from embodied_captioning_amd.weights import synthetic_pixels
    px = synthetic_pixels(16, arch.image_size, seed=meta["seed"])
    L = meta["max_length"]
    ref = None
    for B in (16, rows):
        eng = _engine(arch, "f32s", 16, 1, L, "small")
        eng.load_state_dict(sd)
        out = eng.generate(px[:B].cuda(), num_beams=1, max_length=L, output_logits=True)
        cur = (out["sequences"].cpu(), out["logits"].cpu())
        eng.close()
        if ref is None:
            ref = cur
    live = torch.from_numpy(_live_mask(cur[0].numpy(), arch, L - 1))
    assert torch.equal(cur[0], ref[0][:rows])
    assert torch.equal(cur[1][live], ref[1][:, :rows][live])


def test_six_launches_per_layer_step_and_early_exit():
    g, meta, arch, sd, px = golden_inputs("blip_base")
    B, L = meta["batch"], meta["max_length"]
    eng = _engine(arch, "f32s", B, 1, L)
    eng.load_state_dict(sd)
    eng.profile(True)
    full = eng.generate(px.cuda(), num_beams=1, max_length=L)
    rep = eng.profile_report()
    eng.profile(False)
    per = sum(r["launches"] for t, r in rep.items() if t.startswith("dec_small_") and t not in ("dec_small_tr", "dec_small_vocab"))
    assert per == 6 * arch.t_layers * (L - 1), rep
    assert not any(t.startswith("dec_gemm_") or t in ("dec_reduce_ln", "dec_reduce_ln_wave", "dec_self_attn", "dec_cross_attn") for t in rep), rep
    eng.set_early_exit(2)
    early = eng.generate(px.cuda(), num_beams=1, max_length=L)
    assert torch.equal(early["sequences"], full["sequences"])
    eng.close()


def test_forcing_the_small_path_beyond_its_row_limit_fails_by_name():
    from embodied_captioning_amd._native import CaptionerHipError
    g, meta, arch, sd, px = golden_inputs("blip_tiny")
    from embodied_captioning_amd.weights import synthetic_pixels
    px = synthetic_pixels(17, arch.image_size, seed=1)
    eng = _engine(arch, "f32s", 17, 1, meta["max_length"], "small")
    eng.load_state_dict(sd)
    with pytest.raises(CaptionerHipError, match="small-batch decode path was forced"):
        eng.generate(px.cuda(), num_beams=1, max_length=meta["max_length"])
    eng.set_decode_path("auto")
    eng.generate(px.cuda(), num_beams=1, max_length=meta["max_length"])
    assert eng.last_decode_path == "batch"
    eng.close()


# ---- CoCa (pre-LN decoder, 36 blocks: the reference's production captioner, called with ONE crop - coca.py:27-33).  Parity of
# the CoCa path against open_clip is unpinned; what is held here needs no oracle: the fused kernels against the batch kernels.

def _coca_outs(arch, sd, px, dtype, beams, groups=None):
    from embodied_captioning_amd.engine import CaptionerEngine
    outs = {}
    for path in ("small", "batch"):
        eng = CaptionerEngine(arch, dtype=dtype, max_batch=px.shape[0], max_beams=beams, max_len=arch.seq_len)
        eng.set_decode_path(path)
        eng.load_state_dict(sd)
        outs[path] = eng.generate(px.cuda(), num_beams=beams, max_length=arch.seq_len, output_logits=beams == 1 and groups is None,
                                  num_beam_groups=groups)
        assert eng.last_decode_path == path
        eng.close()
    return outs["small"], outs["batch"]


@pytest.mark.parametrize("dtype", ["f32s", "bf16"])
@pytest.mark.parametrize("boost", [0.0, 4.0])
def test_coca_tiny_small_path_has_the_bits_of_the_batch_path_unpinned(boost, dtype):
    from embodied_captioning_amd.config import CocaArch
    from embodied_captioning_amd.weights import procedural_coca_state_dict, synthetic_pixels
    a = CocaArch.tiny()
    sd = procedural_coca_state_dict(a, 1, eos_boost=boost)
    px = synthetic_pixels(5, a.image_size, seed=1)
    s, b = _coca_outs(a, sd, px, dtype, 1)
    assert torch.equal(s["sequences"], b["sequences"]) and torch.equal(s["lengths"], b["lengths"])
    seq = s["sequences"].cpu().numpy()
    steps = s["logits"].shape[0]
    live = np.ones((steps, seq.shape[0]), dtype=bool)
    for r in range(seq.shape[0]):
        row = list(seq[r, 1:])
        ends = [i for i, tk in enumerate(row) if tk in (a.eos, a.pad)]
        if ends:
            live[ends[0] + 1:, r] = False
    live = torch.from_numpy(live)
    assert torch.equal(s["logits"].cpu()[live], b["logits"].cpu()[live])
    # beams (the reference's _generate_beamsearch) and beam groups
    for beams, groups in ((3, None), (5, None), (6, 3)):
        s, b = _coca_outs(a, sd, px[: 16 // beams], dtype, beams, groups)
        assert torch.equal(s["sequences"], b["sequences"]) and torch.equal(s["sequences_scores"], b["sequences_scores"]), (beams, groups)


@pytest.mark.parametrize("dtype", ["f32s", "bf16"])
def test_coca_vit_l14_small_path_has_the_bits_of_the_batch_path_unpinned(dtype):
    """Production geometry (ViT-L/14, 12 + 12 text layers, vocabulary 49408, 255 cross-attention keys in the KV16 cache): one
    crop and two crops, greedy, every step's live logits."""
    from embodied_captioning_amd.config import CocaArch
    from embodied_captioning_amd.weights import procedural_coca_state_dict, synthetic_pixels
    a = CocaArch()
    sd = procedural_coca_state_dict(a, 2, eos_boost=3.0)
    px = synthetic_pixels(2, a.image_size, seed=5)
    s, b = _coca_outs(a, sd, px, dtype, 1)
    assert torch.equal(s["sequences"], b["sequences"])
    seq = s["sequences"].cpu().numpy()
    for r in range(2):
        row = list(seq[r, 1:])
        ends = [i for i, tk in enumerate(row) if tk in (a.eos, a.pad)]
        n = ends[0] + 1 if ends else len(row)
        assert torch.equal(s["logits"][:n, r], b["logits"][:n, r]), r
    s1, _ = _coca_outs(a, sd, px[:1], dtype, 1)
    assert torch.equal(s1["sequences"][0], s["sequences"][0])


@pytest.mark.parametrize("dtype,cross_cache", [("f32s", "auto"), ("f32s", "fp32"), ("bf16", "auto")])
def test_577_token_checkpoints_small_path_has_the_bits_of_the_batch_path(dtype, cross_cache):
    """BLIP at 384 px (what the published checkpoints ship): the (row, head) K/V block no longer fits the cross kernel's LDS copy
    (2 x 80 KB of KV16 groups), so the block is read from global memory - same arithmetic, same bits; also with fp32 rows."""
    from embodied_captioning_amd.config import BlipArch
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels
    arch = BlipArch()
    arch.image_size = 384
    sd = procedural_blip_state_dict(arch, 0, eos_boost=6.0)
    px = synthetic_pixels(3, arch.image_size, seed=41)
    L = 20
    outs = {}
    for path in ("small", "batch"):
        eng = CaptionerEngine(arch, dtype=dtype, max_batch=3, max_beams=1, max_len=L, cross_cache=cross_cache)
        eng.set_decode_path(path)
        eng.load_state_dict(sd)
        outs[path] = eng.generate(px.cuda(), max_length=L, output_logits=True)
        assert eng.last_decode_path == path
        eng.close()
    a, b = outs["small"], outs["batch"]
    assert torch.equal(a["sequences"], b["sequences"])
    live = torch.from_numpy(_live_mask(a["sequences"].cpu().numpy(), arch, L - 1))
    assert torch.equal(a["logits"].cpu()[live], b["logits"].cpu()[live])


def test_engine_pool_and_a_large_arena_take_the_small_path_for_small_calls():
    """An engine sized for 256 frames that is handed 3, and the engines of a stream pool: the row count of the CALL selects the path."""
    from embodied_captioning_amd.engine import CaptionerEngine, EnginePool
    g, meta, arch, sd, px = golden_inputs("blip_base")
    L = meta["max_length"]
    ref = pad_to(g["greedy_sequences"], L, arch.pad)
    eng = CaptionerEngine(arch, dtype="f32s", max_batch=64, max_beams=1, max_len=L)
    eng.load_state_dict(sd)
    out = eng.generate(px[:3].cuda(), max_length=L)
    assert eng.last_decode_path == "small" and np.array_equal(out["sequences"].cpu().numpy(), ref[:3])
    pool = EnginePool(arch, n=2, dtype="f32s", max_batch=8, max_beams=1, max_len=L, weights_of=eng)
    outs = pool.generate_many([px[i:i + 4].cuda() for i in (0, 4)], threads=True, max_length=L)
    assert np.array_equal(torch.cat([o["sequences"] for o in outs]).cpu().numpy(), ref)
    assert all(e.last_decode_path == "small" for e in pool.engines)
    pool.close()
    eng.close()


@pytest.mark.parametrize("dtype", ["f32s", "bf16"])
def test_automatic_selection_crosses_position_32_with_the_bits_of_the_batch_path(dtype):
    """The fused kernels take at most 32 cached positions: with max_length 40 the automatic selection runs them for positions 1..32
    and the batch kernels from 33 on, in the middle of ONE generate.  Both read and write the same self-attention cache and
    LayerNorm rows, so tokens and live logits equal the batch path's; forcing the small path for such a call fails at entry."""
    from embodied_captioning_amd._native import CaptionerHipError
    from embodied_captioning_amd.config import BlipArch
    from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels
    arch = BlipArch.tiny()
    sd = procedural_blip_state_dict(arch, 5, eos_boost=-6.0)          # EOS pushed down: the captions run the whole length
    B, L = 5, 40
    px = synthetic_pixels(B, arch.image_size, seed=5).cuda()
    outs = {}
    for path in ("auto", "batch"):
        eng = _engine(arch, dtype, B, 1, L, path)
        eng.load_state_dict(sd)
        outs[path] = eng.generate(px, num_beams=1, max_length=L, output_logits=True)
        assert eng.last_decode_path == "batch"                        # what the LAST step ran on, either way
        eng.close()
    a, b = outs["auto"], outs["batch"]
    assert int(a["lengths"].max()) > 34                               # the switch happened inside live captions
    assert torch.equal(a["sequences"], b["sequences"]) and torch.equal(a["lengths"], b["lengths"])
    live = torch.from_numpy(_live_mask(a["sequences"].cpu().numpy(), arch, L - 1))
    assert torch.equal(a["logits"].cpu()[live], b["logits"].cpu()[live])
    eng = _engine(arch, dtype, B, 1, L, "small")
    eng.load_state_dict(sd)
    with pytest.raises(CaptionerHipError, match="at most 32"):
        eng.generate(px, num_beams=1, max_length=L)
    assert eng.last_decode_steps == 0 or eng.last_decode_path in ("none", "small", "batch")
    eng.close()
