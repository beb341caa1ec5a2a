"""GPU: the batch path's fused cross block (csrc/decode_tile.hip: split-K consumer + LayerNorm, query projection and
cross-attention of a decoder layer in ONE launch per 16-row tile and head - 9 launches per layer-step instead of 11) against the
one-launch-per-operation batch kernels and the HF goldens.

The bar is the small-batch path's: the fused kernel forms the same sums in the same order (it calls the same functions), so tokens
AND per-step logits of every live row have the same bits on either path - `torch.equal`, no tolerance."""
import numpy as np
import pytest
import torch

from _util import golden_inputs

pytestmark = pytest.mark.gpu


def _engine(arch, dtype, batch, max_len, path, **kw):
    from embodied_captioning_amd.engine import CaptionerEngine
    eng = CaptionerEngine(arch, dtype=dtype, max_batch=batch, max_beams=1, max_len=max_len, **kw)
    eng.set_decode_path(path)
    return eng


def _live_mask(seq, arch, steps):
    B, L = seq.shape
    live = np.ones((steps, B), dtype=bool)
    for b in range(B):
        row = list(seq[b, 1:])
        if arch.eos in row:
            live[row.index(arch.eos) + 1:, b] = False
    return live


def _both(arch, sd, px, dtype, L, **kw):
    outs = {}
    for path in ("tile", "batch"):
        eng = _engine(arch, dtype, px.shape[0], L, path, **kw)
        eng.load_state_dict(sd)
        outs[path] = eng.generate(px.cuda(), num_beams=1, max_length=L, output_logits=True)
        assert eng.last_decode_path == path
        if dtype == "f32s":
            assert eng.saturations() == 0
        eng.close()
    return outs["tile"], outs["batch"]


def _same_bits(a, b, arch, L):
    assert torch.equal(a["sequences"], b["sequences"])
    assert torch.equal(a["lengths"], b["lengths"])
    live = torch.from_numpy(_live_mask(a["sequences"].cpu().numpy(), arch, L - 1))
    la, lb = a["logits"].cpu(), b["logits"].cpu()
    assert torch.equal(la[live], lb[live]), float((la[live] - lb[live]).abs().max())


@pytest.mark.parametrize("dtype", ["f32s", "bf16"])
def test_fused_cross_block_has_the_bits_of_the_batch_path_blip_base_64_rows(dtype):
    """BLIP-base, 64 frames of the golden batch (4 row tiles x 12 heads; KV16 and bf16 caches), captions that end at different
    steps (rows of ended captions are skipped by both paths)."""
    g, meta, arch, sd, px = golden_inputs("blip_base64")
    L = meta["max_length"]
    a, b = _both(arch, sd, px, dtype, L)
    _same_bits(a, b, arch, L)
    if dtype == "f32s":
        assert np.array_equal(a["sequences"].cpu().numpy(), g["greedy_sequences"])        # and HF's tokens


@pytest.mark.parametrize("dtype", ["f32s", "bf16"])
@pytest.mark.parametrize("rows", [17, 20, 33])
def test_fused_cross_block_ragged_tiles_and_short_image_towers(dtype, rows):
    """The fixture-sized tower (5 image tokens: the one-round-trip attention unit; fp32 / bf16 rows) with row counts that leave a
    partial last tile (17 = 16 + 1, 20, 33 = 2 x 16 + 1)."""
    from embodied_captioning_amd.config import BlipArch
    from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels
    arch = BlipArch.tiny()
    sd = procedural_blip_state_dict(arch, 3, eos_boost=2.0)
    px = synthetic_pixels(rows, arch.image_size, seed=3)
    L = 12
    a, b = _both(arch, sd, px, dtype, L)
    _same_bits(a, b, arch, L)


def test_fused_cross_block_whole_headline_batch():
    """256 frames = the bench's batch on the fused kernels: the tokens are HF's on all 256 rows, the bits those of the
    one-launch-per-operation kernels (which the automatic selection keeps above 16 rows: the two measure level)."""
    g, meta, arch, sd, px = golden_inputs("blip_base256")
    L = meta["max_length"]
    eng = _engine(arch, "f32s", 256, L, "tile")
    eng.load_state_dict(sd)
    out = eng.generate(px.cuda(), num_beams=1, max_length=L, output_logits=True)
    assert eng.last_decode_path == "tile"
    assert np.array_equal(out["sequences"].cpu().numpy(), g["greedy_sequences"])
    eng.set_decode_path("auto")
    ref = eng.generate(px.cuda(), num_beams=1, max_length=L, output_logits=True)
    assert eng.last_decode_path == "batch"
    _same_bits(out, ref, arch, L)
    # beams are not greedy: the selection keeps the batch kernels, forcing the fused ones fails by name
    eng.close()
    from embodied_captioning_amd._native import CaptionerHipError
    from embodied_captioning_amd.engine import CaptionerEngine
    eng = CaptionerEngine(arch, dtype="f32s", max_batch=32, max_beams=3, max_len=L)
    eng.load_state_dict(sd)
    eng.generate(px[:32].cuda(), num_beams=3, max_length=L)
    assert eng.last_decode_path == "batch"
    eng.set_decode_path("tile")
    with pytest.raises(CaptionerHipError, match="fused batch decode path"):
        eng.generate(px[:32].cuda(), num_beams=3, max_length=L)
    eng.close()
    # fp32 cross-attention rows at 197 keys (cross_cache="fp32"): chunks of 56 keys do not fit the fused kernel's registers
    eng = CaptionerEngine(arch, dtype="f32s", max_batch=32, max_beams=1, max_len=L, cross_cache="fp32")
    eng.load_state_dict(sd)
    eng.set_decode_path("tile")
    with pytest.raises(CaptionerHipError, match="fused batch decode path"):
        eng.generate(px[:32].cuda(), num_beams=1, max_length=L)
    eng.close()
