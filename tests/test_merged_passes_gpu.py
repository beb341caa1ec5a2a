"""GPU: the passes the bench's headline runs on - 768 / 1024-row `cap_generate` calls on `max_batch = 1024` arenas, produced by
`EnginePool.generate_many(coalesce_rows=1024)` - against the HF golden (tests/golden/blip_base256.npz, the real
`BlipForConditionalGeneration` greedy loop) and against the unmerged call, bit for bit; and the same mode reached through the
reference-shaped plugin API (`BLIP.generate_batch`, `Captioner.caption_batch`, `BatchedBoxCaptioner`) via `shim.install()`.

A frame decodes to the same bits alone, in its 256-frame batch and in a merged pass, so every comparison
here is `torch.equal` / `np.array_equal` - no tolerance."""
import types

import numpy as np
import pytest
import torch

from _util import golden_inputs

pytestmark = pytest.mark.gpu


def _tags_of_one_generate(eng, px, L):
    eng.profile(True)
    eng.generate(px, num_beams=1, max_length=L)
    rep = eng.profile_report()
    eng.profile(False)
    return rep


def test_merged_1024_row_passes_are_the_golden_and_the_unmerged_bits():
    """One EnginePool(max_batch=1024): 4 x the 256 golden frames merged into passes of up to 1024 rows, and a ragged 256 + 512 +
    256 - every batch token-identical to the HF golden AND equal (sequences, lengths) to the call with every batch its own pass;
    the large-pass split-K consumer (wave per row: from 512 rows on in the compacted loop, 832 otherwise) is the kernel a 1024-row pass runs."""
    from embodied_captioning_amd.engine import EnginePool
    g, meta, arch, sd, px = golden_inputs("blip_base256")
    B, L = meta["batch"], meta["max_length"]
    ref = torch.from_numpy(np.asarray(g["greedy_sequences"])).int()
    pool = EnginePool(arch, n=3, dtype="f32s", max_batch=1024, max_beams=1, max_len=L)
    pool.load_state_dict(sd)
    pxd = px.cuda()
    # (a) four whole batches.  Three engines: the plan never has fewer passes than engines -> 512 + 256 + 256 rows
    batches = [pxd] * 4
    plain = pool.generate_many(batches, threads=True, num_beams=1, max_length=L)
    assert pool.last_coalesce is None
    merged = pool.generate_many(batches, threads=True, coalesce_rows=1024, num_beams=1, max_length=L)
    plan = pool.last_coalesce
    assert isinstance(plan, list) and max(sum(B for _ in gp) for gp in plan) >= 512, plan
    for a, b in zip(plain, merged):
        assert torch.equal(a["sequences"], b["sequences"]) and torch.equal(a["lengths"], b["lengths"])
        assert torch.equal(b["sequences"].cpu(), ref)
    # (b) eight batches: passes of 1024 / 768 rows (what the bench's 20 steps run: 4, 4, 3, 3, 3, 3 batches)
    merged8 = pool.generate_many([pxd] * 8, threads=True, coalesce_rows=1024, num_beams=1, max_length=L)
    rows = sorted(len(gp) * B for gp in pool.last_coalesce)
    assert rows[-1] >= 768 and sum(rows) == 8 * B, rows
    for b in merged8:
        assert torch.equal(b["sequences"].cpu(), ref) and torch.equal(b["lengths"], plain[0]["lengths"])
    # (c) ragged: 256 + 512 + 256 (the 512 is two golden batches back to back)
    ragged = [pxd, torch.cat([pxd, pxd]), pxd, pxd[:96], pxd[96:]]
    plain_r = pool.generate_many(ragged, threads=True, num_beams=1, max_length=L)
    merged_r = pool.generate_many(ragged, threads=True, coalesce_rows=1024, num_beams=1, max_length=L)
    assert isinstance(pool.last_coalesce, list) and any(len(gp) > 1 for gp in pool.last_coalesce)
    want = [ref, torch.cat([ref, ref]), ref, ref[:96], ref[96:]]
    for a, b, w in zip(plain_r, merged_r, want):
        assert torch.equal(a["sequences"], b["sequences"]) and torch.equal(a["lengths"], b["lengths"])
        assert torch.equal(b["sequences"].cpu(), w)
    # (d) which split-K consumer a 1024-row pass runs, and a 256-row one
    big = torch.cat([pxd] * 4)
    with torch.cuda.stream(pool.streams[0]):
        rep_big = _tags_of_one_generate(pool.engines[0], big, L)
        rep_small = _tags_of_one_generate(pool.engines[0], pxd, L)
    torch.cuda.synchronize()
    assert "dec_reduce_ln_wave" in rep_big and "dec_reduce_ln" not in rep_big, sorted(rep_big)
    assert "dec_reduce_ln" in rep_small and "dec_reduce_ln_wave" not in rep_small, sorted(rep_small)
    # an output key the splitter does not know is an error, never a guess by shape
    eng0 = pool.engines[0]
    orig = eng0.generate
    eng0.generate = lambda *a, **k: dict(orig(*a, **k), mystery=torch.zeros(3))
    try:
        with pytest.raises(Exception, match="mystery"):
            pool.generate_many([pxd[:8], pxd[8:16], pxd[16:24], pxd[24:32], pxd[32:40], pxd[40:48]], coalesce_rows=16, max_length=L)
    finally:
        del eng0.generate
    torch.cuda.synchronize()
    pool.close()


@pytest.mark.parametrize("cross_cache", ["auto", "fp32"])
def test_one_1024_row_generate_matches_the_golden(cross_cache):
    """A single `cap_generate` of 1024 rows (4 x the golden batch) on a max_batch = 1024 arena - KV16 cross cache (7.7 GB per
    engine: byte offsets beyond 2^32) and fp32 rows: every row token-identical to the HF golden, lengths those of a 256-row call."""
    from embodied_captioning_amd.engine import CaptionerEngine
    g, meta, arch, sd, px = golden_inputs("blip_base256")
    L = meta["max_length"]
    ref = np.asarray(g["greedy_sequences"])
    kw = {"cross_cache": "fp32"} if cross_cache == "fp32" else {}
    eng = CaptionerEngine(arch, dtype="f32s", max_batch=1024, max_beams=1, max_len=L, **kw)
    eng.load_state_dict(sd)
    assert eng.cross_cache_kind == ("fp32" if cross_cache == "fp32" else "kv16")
    pxd = px.cuda()
    small = eng.generate(pxd, num_beams=1, max_length=L)
    big = eng.generate(torch.cat([pxd] * 4), num_beams=1, max_length=L)
    seq = big["sequences"].cpu().numpy().reshape(4, 256, L)
    for k in range(4):
        same = (seq[k] == ref).all(axis=1)
        assert same.all(), (k, int(same.sum()), np.nonzero(~same)[0][:8])
    assert torch.equal(big["lengths"].view(4, 256), small["lengths"][None].expand(4, -1))
    assert torch.equal(big["sequences"].view(4, 256, L)[3], small["sequences"])
    assert eng.saturations() == 0
    eng.close()


def test_tiny_arch_rows_alone_and_inside_an_860_row_pass_same_tokens_and_logits():
    """Across the consumer kernels' row-count threshold on the fixture-sized architecture: the same 24 frames generated alone (block
    per row) and inside a pass of 860 rows (wave per row: per-step logits switch the compaction off, so from 832 rows on) - tokens AND
    per-step logits `torch.equal`."""
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import synthetic_pixels
    g, meta, arch, sd, px = golden_inputs("blip_tiny")
    L = meta["max_length"]
    eng = CaptionerEngine(arch, dtype="f32s", max_batch=860, max_beams=1, max_len=L)
    eng.load_state_dict(sd)
    mine = synthetic_pixels(24, arch.image_size, seed=5).cuda()
    filler = synthetic_pixels(836, arch.image_size, seed=77).cuda()
    alone = eng.generate(mine, num_beams=1, max_length=L, output_logits=True)
    rep = _tags_of_one_generate(eng, torch.cat([filler[:300], mine, filler[300:]]), L)
    assert "dec_reduce_ln_wave" in rep and "dec_reduce_ln" not in rep, sorted(rep)
    inside = eng.generate(torch.cat([filler[:300], mine, filler[300:]]), num_beams=1, max_length=L, output_logits=True)
    assert torch.equal(alone["sequences"], inside["sequences"][300:324])
    assert torch.equal(alone["lengths"], inside["lengths"][300:324])
    # logits of the steps in which the row was still open (a finished row's inputs are pad tokens on both sides too, but only the
    # open steps are part of the contract)
    lens = alone["lengths"].cpu().numpy()
    la, li = alone["logits"], inside["logits"][:, 300:324]
    for r in range(24):
        n = int(lens[r]) - 1
        assert torch.equal(la[:n, r], li[:n, r]), r
    eng.close()


def _box_inputs():
    rng = np.random.default_rng(3)
    frames = [rng.integers(0, 256, size=(120, 160, 3), dtype=np.uint8) for _ in range(4)]
    boxes = [[(10, 20, 60, 90), (100, 5, 158, 60), (30, 30, 90, 100)], [(0, 0, 160, 120)], [],
             [(40, 40, 94, 97), (70, 30, 130, 110), (5, 5, 50, 60), (20, 60, 150, 118)]]
    return boxes, frames


def test_plugin_batch_entry_points_reach_the_merged_mode_through_the_shim():
    """The reference-side callers' path (`detector/pseudolabeler.py:664-711`, `scripts/run_pseudolabeler.py:77-107`): with
    `captioner.streams > 1` the wrapper's pool merges micro-batches (cfg key `captioner.coalesce_rows`, on by default) - imported
    through the reference's module paths (shim.install()); captions equal those of the single-engine plugin, frame by frame."""
    import embodied_captioning_amd.shim as shim
    shim.install()
    from experimenting_env.utils.predictor_utils import Captioner
    from embodied_captioning_amd.pseudolabeler import BatchedBoxCaptioner

    def plugin(**kw):
        cap_cfg = types.SimpleNamespace(arch_name="blip", model_name="procedural-tiny:4:2.0", checkpoint_name=None, height=224, width=224,
                                        dtype="f32s", max_length=12, batch_size=4, **kw)
        return Captioner(types.SimpleNamespace(captioner=cap_cfg)).to("cuda:0").eval()

    one = plugin()
    pooled = plugin(streams=3)                         # coalesce_rows default: 4 x batch_size = 16 rows per pass
    assert pooled.model.coalesce_rows == 16 and pooled.model.pool.engines[0].max_batch == 16
    off = plugin(streams=3, coalesce_rows=0)
    assert off.model.coalesce_rows == 0 and off.model.pool.engines[0].max_batch == 4
    rng = np.random.default_rng(11)
    crops = torch.from_numpy(rng.integers(0, 256, size=(45, 32, 32, 3), dtype=np.uint8))     # the tiny architecture's 32 x 32 frames
    want = one.caption_batch(crops)
    got = pooled.caption_batch(crops)
    assert isinstance(pooled.model.pool.last_coalesce, list) and any(len(gp) > 1 for gp in pooled.model.pool.last_coalesce)
    assert got == want
    assert off.caption_batch(crops) == want and off.model.pool.last_coalesce is None
    full = pooled.model.generate_batch(crops)
    ref = one.model.generate_batch(crops)
    assert torch.equal(full["sequences"], ref["sequences"]) and torch.equal(full["lengths"], ref["lengths"])
    # the box driver on top of it
    boxes, frames = _box_inputs()
    a = BatchedBoxCaptioner(one).predict_captions(boxes, frames)
    b = BatchedBoxCaptioner(pooled).predict_captions(boxes, frames)
    assert [x["captions"] for x in a] == [x["captions"] for x in b]
    assert [len(x["captions"]) for x in b] == [3, 1, 0, 4]


@pytest.mark.parametrize("dtype,name", [("f32s", "blip_base256"), ("bf16", "blip_base64"), ("f32", "blip_base64"), ("f32s", "blip_tiny_eos")])
def test_compacted_decode_loop_gives_the_uncompacted_bits(dtype, name):
    """`cap_set_row_compaction`: the greedy loop that packs the open captions' rows to the front after every token selection and
    runs every kernel of the next step on those rows only - against the loop that keeps every row in place: sequences and lengths
    `torch.equal`, at one batch and at 4 batches in one pass (1 024 rows for the 256-frame golden), rows in a shuffled order too
    (captions end at different steps: the compact position of a row changes from step to step)."""
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import synthetic_pixels
    g, meta, arch, sd, px = golden_inputs(name)
    L, B = meta["max_length"], meta["batch"]
    if B < 17:                                           # the fixture-sized goldens hold 8 frames: more of them, other seeds
        px = torch.cat([px, synthetic_pixels(40, arch.image_size, seed=123)])
    eng = CaptionerEngine(arch, dtype=dtype, max_batch=4 * px.shape[0], max_beams=1, max_len=L)
    eng.load_state_dict(sd)
    pxd = px.cuda()
    perm = torch.randperm(4 * px.shape[0], generator=torch.Generator().manual_seed(5)).cuda()
    for frames in (pxd, torch.cat([pxd] * 4), torch.cat([pxd] * 4)[perm]):
        eng.set_row_compaction(True)
        on = eng.generate(frames, num_beams=1, max_length=L)
        assert eng.last_row_compaction is True and eng.last_decode_path == "batch"
        eng.set_row_compaction(False)
        off = eng.generate(frames, num_beams=1, max_length=L)
        assert eng.last_row_compaction is False
        assert torch.equal(on["sequences"], off["sequences"]) and torch.equal(on["lengths"], off["lengths"])
    if dtype != "bf16" and B >= 17:                      # and the HF golden itself (bf16 is not a parity mode)
        ref = torch.from_numpy(np.asarray(g["greedy_sequences"])).int()
        assert torch.equal(on["sequences"].cpu(), torch.cat([ref] * 4)[perm.cpu()])
    # asking for per-step logits keeps every row in place (their rows are the batch's rows)
    eng.set_row_compaction(True)
    eng.generate(pxd, num_beams=1, max_length=L, output_logits=True)
    assert eng.last_row_compaction is False
    eng.close()


def test_compacted_loop_with_early_exit_and_every_caption_ended():
    """Captions that all end within a few steps (a large EOS offset): the compacted loop runs its remaining steps on zero rows
    - every kernel returns at its first instruction - and the early-exit poll still leaves the loop; same tokens as uncompacted."""
    from embodied_captioning_amd.config import BlipArch
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels
    arch = BlipArch.tiny()
    sd = procedural_blip_state_dict(arch, 3, eos_boost=30.0)
    px = synthetic_pixels(200, arch.image_size, seed=9).cuda()
    eng = CaptionerEngine(arch, dtype="f32s", max_batch=200, max_beams=1, max_len=20)
    eng.load_state_dict(sd)
    on = eng.generate(px, num_beams=1, max_length=20)
    assert eng.last_row_compaction and eng.last_decode_steps == 19 and int(on["lengths"].max()) <= 6
    eng.set_early_exit(2)
    polled = eng.generate(px, num_beams=1, max_length=20)
    assert eng.last_decode_steps < 19
    eng.set_early_exit(0)
    eng.set_row_compaction(False)
    off = eng.generate(px, num_beams=1, max_length=20)
    for o in (polled, off):
        assert torch.equal(on["sequences"], o["sequences"]) and torch.equal(on["lengths"], o["lengths"])
    eng.close()
