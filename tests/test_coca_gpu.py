"""GPU: CoCa through the C ABI against the CPU restatement (oracle/coca_ref.py - unpinned, see its header)."""
import numpy as np
import pytest
import torch

from _util import token_parity

pytestmark = pytest.mark.gpu


def _setup(arch, seed, boost, batch, dtype):
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import procedural_coca_state_dict, synthetic_pixels
    sd = procedural_coca_state_dict(arch, seed, eos_boost=boost)
    px = synthetic_pixels(batch, arch.image_size, seed=seed)
    eng = CaptionerEngine(arch, dtype=dtype, max_batch=batch, max_beams=1, max_len=arch.seq_len)
    eng.load_state_dict(sd)
    return sd, px, eng


def _padded(text, L, pad):
    out = np.full((text.shape[0], L), pad, dtype=np.int64)
    out[:, : text.shape[1]] = text.numpy()
    return out


@pytest.mark.parametrize("dtype", ["f32", "f32s"])
@pytest.mark.parametrize("boost", [0.0, 4.0])
def test_coca_tiny_fp32_matches_restatement(boost, dtype):
    """fp32 and the split mode (fp32-grade products on the fp16 MFMA pipe) hold the same bar against the restatement."""
    from embodied_captioning_amd.config import CocaArch
    from oracle import coca_ref as R
    a = CocaArch.tiny()
    sd, px, eng = _setup(a, 1, boost, 4, dtype)
    pooled, embs = R.encode_image(sd, a, px)
    tok = eng.encode(px.cuda()).cpu()                       # [B, Q, E]: row 0 pooled token (before visual.proj), rows 1.. image_embs
    assert (tok[:, 1:] - embs).abs().max().item() < 2e-4
    assert (tok[:, 0] @ sd["visual.proj"] - pooled).abs().max().item() < 2e-4
    ref = R.generate_top1(sd, a, px, image_embs=embs)
    out = eng.generate(px.cuda(), max_length=a.seq_len, output_logits=True)
    want = _padded(ref["text"], a.seq_len, a.pad)
    assert np.array_equal(out["sequences"].cpu().numpy(), want)
    # step-0 logits (every row active): raw logits except the MinLength-masked EOS column
    l0 = out["logits"][0].cpu()
    r0 = ref["logits"][0]
    fin = torch.isfinite(r0)
    assert (l0[fin] - r0[fin]).abs().max().item() < 1e-3
    eng.close()


def test_coca_tiny_bf16_within_tolerance():
    from embodied_captioning_amd.config import CocaArch
    from oracle import coca_ref as R
    a = CocaArch.tiny()
    sd, px, eng = _setup(a, 1, 4.0, 4, "bf16")
    _, embs = R.encode_image(sd, a, px)
    tok = eng.encode(px.cuda()).cpu()
    assert (tok[:, 1:] - embs).abs().max().item() < 0.15
    ref = R.generate_top1(sd, a, px, image_embs=embs)
    out = eng.generate(px.cuda(), max_length=a.seq_len)
    # margins of the active rows, scattered back to [steps, B]
    B = 4
    margins = np.full((a.seq_len - 1, B), 1e9, dtype=np.float32)
    active = np.ones(B, dtype=bool)
    text = ref["text"]
    for t, lg in enumerate(ref["logits"]):
        t2 = torch.topk(lg, 2, dim=-1).values
        margins[t, active] = (t2[:, 0] - t2[:, 1]).numpy()
        if t + 1 < text.shape[1]:
            active = active & ~np.isin(text[:, t + 1].numpy(), [a.eos, a.pad])
    exact, diverged, bad = token_parity(out["sequences"].cpu().numpy(), _padded(text, a.seq_len, a.pad), margins, 0.3)
    assert bad is None, bad
    eng.close()


def test_coca_vit_l14_full_size_bf16_encoder_and_first_tokens():
    """Real coca_ViT-L-14 shapes (24x1024 ViT, 257 tokens, 96-wide pooler heads, 49408 vocab), batch 2."""
    from embodied_captioning_amd.config import CocaArch
    from oracle import coca_ref as R
    a = CocaArch()
    sd, px, eng = _setup(a, 0, 0.0, 2, "bf16")
    _, embs = R.encode_image(sd, a, px)
    tok = eng.encode(px.cuda()).cpu()
    err = (tok[:, 1:] - embs).abs().max().item()
    assert err < 0.25, err
    out = eng.generate(px.cuda(), max_length=a.seq_len, output_logits=True)
    st = R.CocaState(a.t_layers + a.mm_layers)
    r0 = R.step(sd, a, torch.full((2,), a.sot, dtype=torch.int64), embs, st)
    top = torch.topk(r0, 8, dim=-1)
    ours = torch.gather(out["logits"][0].cpu(), 1, top.indices)
    assert (ours - top.values).abs().max().item() < 0.35
    seq = out["sequences"].cpu()
    assert (seq[:, 0] == a.sot).all() and (seq[:, -1] == a.eos).all()
    eng.close()


def test_coca_vit_l14_full_size_split_mode_unpinned_encoder_and_tokens():
    """Real coca_ViT-L-14 shapes in the split mode, batch 2: pooled tokens within fp32 tolerance of the restatement (bf16: 0.25),
    step-0 top-8 logits within 1e-3, greedy tokens over the first steps identical, nothing clamped."""
    import dataclasses
    from embodied_captioning_amd.config import CocaArch
    from oracle import coca_ref as R
    a = dataclasses.replace(CocaArch(), seq_len=8, min_seq_len=3)
    sd, px, eng = _setup(a, 0, 2.0, 2, "f32s")
    _, embs = R.encode_image(sd, a, px)
    tok = eng.encode(px.cuda()).cpu()
    err = (tok[:, 1:] - embs).abs().max().item()
    assert err < 2e-3, err
    out = eng.generate(px.cuda(), max_length=a.seq_len, output_logits=True)
    ref = R.generate_top1(sd, a, px, image_embs=tok[:, 1:].contiguous())
    assert np.array_equal(out["sequences"].cpu().numpy(), _padded(ref["text"], a.seq_len, a.pad))
    r0 = ref["logits"][0]
    top = torch.topk(torch.where(torch.isfinite(r0), r0, torch.full_like(r0, -1e30)), 8, dim=-1)
    ours = torch.gather(out["logits"][0].cpu(), 1, top.indices)
    assert (ours - top.values).abs().max().item() < 1e-3
    assert eng.saturations(reset=True) == 0
    eng.close()


def test_coca_vit_l14_336_encoder_bf16():
    """SURVEY.md 8(d) config 5's input size: 336x336 -> 577 tokens (the online-softmax ViT attention kernel), with the
    position table of a 224-pixel checkpoint resized at load the way open_clip's force_image_size does."""
    import dataclasses
    from embodied_captioning_amd.coca_weights import coca_library_state_dict
    from embodied_captioning_amd.config import CocaArch
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import procedural_coca_state_dict, synthetic_pixels
    from oracle import coca_ref as R
    a224 = CocaArch()
    a = dataclasses.replace(a224, image_size=336)
    assert a.n_tokens == 577
    sd224 = procedural_coca_state_dict(a224, 0, eos_boost=0.0)
    sd = coca_library_state_dict(sd224, a)                 # resizes visual.positional_embedding 257 -> 577 rows
    assert sd["visual.positional_embedding"].shape == (577, a.v_hidden)
    assert torch.equal(sd["visual.positional_embedding"][0], sd224["visual.positional_embedding"][0].float())
    px = synthetic_pixels(2, 336, seed=0)
    eng = CaptionerEngine(a, dtype="bf16", max_batch=2, max_beams=1, max_len=a.seq_len)
    eng.load_state_dict(sd)
    _, embs = R.encode_image(sd, a, px)
    tok = eng.encode(px.cuda()).cpu()
    err = (tok[:, 1:] - embs).abs().max().item()
    assert err < 0.25, err
    seq = eng.generate(px.cuda(), max_length=a.seq_len)["sequences"].cpu()
    assert (seq[:, 0] == a.sot).all() and (seq[:, -1] == a.eos).all()
    eng.close()


def test_coca_wrapper_dict_api():
    from PIL import Image
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    from embodied_captioning_amd.captioner.utils.utils_captioner import select_captioner
    rng = np.random.default_rng(0)
    im = Image.fromarray(rng.integers(0, 256, size=(40, 52, 3), dtype=np.uint8), "RGB")
    cfg = Configuration(arch_name="coca", model_name="procedural-coca-tiny:1:4.0", height=224, width=224, dtype="f32").captioner
    model = select_captioner(cfg).eval()
    out = model(im)
    assert isinstance(out["text"], str) and len(out["logits"]) >= 1
    assert out["logits"][0].shape == (1, model.arch.vocab)
    assert torch.isinf(out["logits"][0][0, model.arch.eos])        # MinLength mask visible in the recorded logits
    ppl = model.compute_perplexity()
    assert torch.isfinite(ppl)
    # options of the reference's generate() that the engine was BUILT with (coca_model.py:209-223): accepted at the value in force,
    # refused by name at any other - never dropped
    a = model.arch
    ids = model.generate(im, min_seq_len=a.min_seq_len, eos_token_id=a.eos, pad_token_id=a.pad, sot_token_id=a.sot, max_seq_len=77)
    assert ids.shape == (1, a.seq_len)
    for opts, word in (({"min_seq_len": a.min_seq_len + 2}, "min_seq_len"), ({"eos_token_id": 7}, "eos_token_id=7"),
                       ({"fixed_output_length": True}, "fixed_output_length"), ({"max_seq_len": 4}, "max_seq_len=4")):
        with pytest.raises(ValueError, match=word):
            model.generate(im, **opts)


def _beam_expected(ref, L, pad):
    """The reference returns [B, <= seq_len] (EOS appended where it fits, pad after); the ABI returns [B, max_len] + lengths."""
    seq = ref["sequences"]
    want = _padded(seq, L, pad)
    lens = []
    for r in seq.tolist():
        n = len(r)
        while n > 1 and r[n - 1] == pad:
            n -= 1
        lens.append(n)
    return want, np.array(lens)


@pytest.mark.parametrize("dtype", ["f32", "f32s"])
@pytest.mark.parametrize("boost,K", [(0.0, 5), (4.0, 5), (2.0, 3), (2.0, 1 + 5)])
def test_coca_beam_search_unpinned_tiny_fp32_matches_restatement(boost, K, dtype):
    """Config 5's decode: the reference's `_generate_beamsearch` (coca_model.py:335-482, raw-logit scores, HF's legacy
    BeamSearchScorer, one beam group) on the device against its restatement oracle/coca_ref.generate_beamsearch - UNPINNED:
    neither open_clip nor the scorer (gone from transformers 5) can be run here."""
    from embodied_captioning_amd.config import CocaArch
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import procedural_coca_state_dict, synthetic_pixels
    from oracle import coca_ref as R
    a = CocaArch.tiny()
    B = 5
    sd = procedural_coca_state_dict(a, 3, eos_boost=boost)
    px = synthetic_pixels(B, a.image_size, seed=3)
    eng = CaptionerEngine(a, dtype=dtype, max_batch=B, max_beams=K, max_len=a.seq_len)
    eng.load_state_dict(sd)
    _, embs = R.encode_image(sd, a, px)
    ref = R.generate_beamsearch(sd, a, px, num_beams=K, image_embs=embs)
    out = eng.generate(px.cuda(), num_beams=K, max_length=a.seq_len, length_penalty=1.0)
    want, lens = _beam_expected(ref, a.seq_len, a.pad)
    assert np.array_equal(out["sequences"].cpu().numpy(), want), (out["sequences"].cpu().numpy(), want)
    assert np.array_equal(out["lengths"].cpu().numpy(), lens)
    np.testing.assert_allclose(out["sequences_scores"].cpu().numpy(), ref["scores"].numpy(), rtol=0, atol=1e-3)
    # greedy on the same handle still is the reference's top-k(1) loop
    g = R.generate_top1(sd, a, px, image_embs=embs)
    assert np.array_equal(eng.generate(px.cuda(), max_length=a.seq_len)["sequences"].cpu().numpy(), _padded(g["text"], a.seq_len, a.pad))
    eng.close()


def test_coca_beam_search_unpinned_vit_l14_336_first_steps():
    """The production geometry of config 5 (ViT-L/14 at 336x336, 577 image tokens, 12 + 12 text layers, vocabulary 49408,
    beam 5): fp32 beams identical to the restatement over the first steps (seq_len 7: the restatement recomputes the whole
    prefix on the host, as the reference does), bf16 beams well-formed."""
    import dataclasses
    from embodied_captioning_amd.config import CocaArch
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import procedural_coca_state_dict, synthetic_pixels
    from oracle import coca_ref as R
    a = dataclasses.replace(CocaArch(), image_size=336)
    sd = procedural_coca_state_dict(a, 0, eos_boost=3.0)
    B, K, L, MIN = 2, 5, 7, 3
    px = synthetic_pixels(B, 336, seed=1)
    eng = CaptionerEngine(dataclasses.replace(a, min_seq_len=MIN), dtype="f32", max_batch=B, max_beams=K, max_len=L)
    eng.load_state_dict(sd)
    tok = eng.encode(px.cuda()).cpu()
    ref = R.generate_beamsearch(sd, a, px, num_beams=K, seq_len=L, min_seq_len=MIN, image_embs=tok[:, 1:].contiguous())
    out = eng.generate(px.cuda(), num_beams=K, max_length=L, length_penalty=1.0)
    want, lens = _beam_expected(ref, L, a.pad)
    assert np.array_equal(out["sequences"].cpu().numpy(), want), (out["sequences"].cpu().numpy(), want)
    np.testing.assert_allclose(out["sequences_scores"].cpu().numpy(), ref["scores"].numpy(), rtol=0, atol=2e-3)
    eng.close()
    eng = CaptionerEngine(a, dtype="bf16", max_batch=B, max_beams=K, max_len=a.seq_len)
    eng.load_state_dict(sd)
    o = eng.generate(px.cuda(), num_beams=K, max_length=a.seq_len, length_penalty=1.0)
    seq, ln = o["sequences"].cpu().numpy(), o["lengths"].cpu().numpy()
    assert (seq[:, 0] == a.sot).all() and (ln >= a.min_seq_len).all() and (ln <= a.seq_len).all()
    for r, n in zip(seq, ln):
        assert (r[n:] == a.pad).all() and (n == a.seq_len or r[n - 1] == a.eos)
    eng.close()


@pytest.mark.parametrize("boost,K,G", [(2.0, 6, 3), (0.0, 4, 2), (4.0, 3, 3), (1.0, 6, 1), (3.0, 8, 4)])
def test_coca_beam_groups_unpinned_tiny_fp32_matches_the_literal_group_loop(boost, K, G):
    """cap_generate_groups (the reference's `generate()` defaults are 6 beams in 3 groups, coca_model.py:218-219) against the
    LITERAL restatement of the reference's group loop (oracle/coca_ref.generate_beamsearch_groups): sequences, lengths, scores.
    Group size one (K == G) is a 1-beam BEAM search - not the greedy loop, whose forced EOS it lacks."""
    from embodied_captioning_amd._native import CaptionerHipError
    from embodied_captioning_amd.config import CocaArch
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import procedural_coca_state_dict, synthetic_pixels
    from oracle import coca_ref as R
    a = CocaArch.tiny()
    B = 5
    sd = procedural_coca_state_dict(a, 3, eos_boost=boost)
    px = synthetic_pixels(B, a.image_size, seed=3)
    eng = CaptionerEngine(a, dtype="f32", max_batch=B, max_beams=K, max_len=a.seq_len)
    eng.load_state_dict(sd)
    _, embs = R.encode_image(sd, a, px)
    ref = R.generate_beamsearch_groups(sd, a, px, num_beams=K, num_beam_groups=G, image_embs=embs)
    out = eng.generate(px.cuda(), num_beams=K, num_beam_groups=G, max_length=a.seq_len, length_penalty=1.0)
    want, lens = _beam_expected(ref, a.seq_len, a.pad)
    assert np.array_equal(out["sequences"].cpu().numpy(), want), (out["sequences"].cpu().numpy(), want)
    assert np.array_equal(out["lengths"].cpu().numpy(), lens)
    np.testing.assert_allclose(out["sequences_scores"].cpu().numpy(), ref["scores"].numpy(), rtol=0, atol=1e-3)
    with pytest.raises(CaptionerHipError, match="multiple of num_beam_groups"):
        eng.generate(px.cuda(), num_beams=K, num_beam_groups=K + 1 if K > 1 else 2, max_length=a.seq_len)
    eng.close()


def test_coca_wrapper_beam_groups_and_bpe_text(tmp_path):
    """The plugin with the model's own generate() defaults (6 beams, 3 groups) and a CLIP BPE vocabulary next to nothing but a
    `tokenizer_dir`: captions come back as TEXT without open_clip (reference coca.py:30 `open_clip.decode`)."""
    from PIL import Image
    from embodied_captioning_amd.captioner.clip_bpe import vocab_from_merges
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    from embodied_captioning_amd.captioner.utils.utils_captioner import select_captioner
    # a toy vocabulary for the tiny arch (512 ids, sot / eos = 510 / 511) in the HF layout: byte symbols, then the two specials
    import json
    sym = vocab_from_merges([])[:510] + ["<start_of_text>", "<end_of_text>"]
    (tmp_path / "vocab.json").write_text(json.dumps({s_: i for i, s_ in enumerate(sym)}), encoding="utf-8")
    rng = np.random.default_rng(0)
    ims = [Image.fromarray(rng.integers(0, 256, size=(40, 52, 3), dtype=np.uint8), "RGB") for _ in range(3)]
    cfg = Configuration(arch_name="coca", model_name="procedural-coca-tiny:1:4.0", height=224, width=224, dtype="f32", num_beams=6,
                        num_beam_groups=3, tokenizer_dir=str(tmp_path)).captioner
    model = select_captioner(cfg).eval()
    if model.tokenizer is not None:
        pytest.skip("open_clip is installed: the wrapper uses it")
    out = model.generate_batch(ims)
    a = model.arch
    for text, row, n in zip(out["texts"], out["sequences"].tolist(), out["lengths"].tolist()):
        body = [t for t in row[:n] if t not in (a.sot, a.eos)]
        assert text == model.bpe.decode(body)            # cut at <end_of_text>, <start_of_text> dropped (coca.py:30)
        assert "<start_of_text>" not in text and "<end_of_text>" not in text and len(text) >= len(body)


@pytest.mark.parametrize("dtype", ["bf16", "f32s"])
def test_coca_pool_dynamic_batching_returns_the_unmerged_bits(dtype):
    """The engine pool's dynamic batching on CoCa (config 5's mode in bench.py --model coca --coalesce-rows N): consecutive image batches
    merged into larger passes and split back - every image's beams are its own and every kernel's sums are batch-independent, so
    sequences, lengths and beam scores are those of the unmerged call bit for bit; greedy (top-k 1) too."""
    from embodied_captioning_amd.config import CocaArch
    from embodied_captioning_amd.engine import EnginePool
    from embodied_captioning_amd.weights import procedural_coca_state_dict, synthetic_pixels
    a = CocaArch.tiny()
    sd = procedural_coca_state_dict(a, 7, eos_boost=2.0)
    px = synthetic_pixels(48, a.image_size, seed=7).cuda()
    for K in (5, 1):
        pool = EnginePool(a, n=3, dtype=dtype, max_batch=24, max_beams=K, max_len=a.seq_len)
        pool.load_state_dict(sd)
        for sizes in ([8] * 6, [4, 12, 8, 2, 6, 16]):
            cuts = np.cumsum([0] + sizes)
            batches = [px[i:j] for i, j in zip(cuts[:-1], cuts[1:])]
            plain = pool.generate_many(batches, threads=True, num_beams=K, max_length=a.seq_len)
            merged = pool.generate_many(batches, threads=True, coalesce_rows=24, num_beams=K, max_length=a.seq_len)
            assert isinstance(pool.last_coalesce, list) and any(len(gp) > 1 for gp in pool.last_coalesce), pool.last_coalesce
            for x, y in zip(plain, merged):
                assert torch.equal(x["sequences"], y["sequences"]) and torch.equal(x["lengths"], y["lengths"])
                if K > 1:
                    assert torch.equal(x["sequences_scores"], y["sequences_scores"])
        pool.close()


def test_coca_wrapper_streams_and_dynamic_batching_same_captions():
    """`captioner.streams: 3` (+ the default `coalesce_rows`) on the CoCa wrapper: generate_batch over an engine pool with merged
    passes returns what the one-engine wrapper returns - sequences, lengths, beam scores, texts."""
    from PIL import Image
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    from embodied_captioning_amd.captioner.utils.utils_captioner import select_captioner
    rng = np.random.default_rng(5)
    ims = [Image.fromarray(rng.integers(0, 256, size=(40 + i % 19, 52, 3), dtype=np.uint8), "RGB") for i in range(14)]
    kw = dict(arch_name="coca", model_name="procedural-coca-tiny:2:3.0", height=224, width=224, dtype="f32s", num_beams=3, batch_size=4)
    one = select_captioner(Configuration(**kw).captioner).eval()
    many = select_captioner(Configuration(streams=3, **kw).captioner).eval()
    assert many.pool is not None and many.coalesce_rows == 16
    a, b = one.generate_batch(ims), many.generate_batch(ims)
    assert isinstance(many.pool.last_coalesce, list) and any(len(g) > 1 for g in many.pool.last_coalesce)
    assert torch.equal(a["sequences"], b["sequences"]) and torch.equal(a["lengths"], b["lengths"]) and a["texts"] == b["texts"]
    assert torch.equal(a["scores"], b["scores"])
    # a list longer than one round of passes (3 engines x 16 images): preprocessed round by round by a helper thread
    long = [ims[i % 14] for i in range(61)]
    a, b = one.generate_batch(long), many.generate_batch(long)
    assert torch.equal(a["sequences"], b["sequences"]) and torch.equal(a["scores"], b["scores"]) and a["texts"] == b["texts"]
