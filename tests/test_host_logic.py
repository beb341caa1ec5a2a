"""CPU: host-side logic of the boundary - plugin API mirror, import shim, shard math, consensus grouping."""
import json
import os

import pytest
import torch

from embodied_captioning_amd import distributed as D


def test_perplexity_kats_through_product_class(golden_dir):
    from embodied_captioning_amd.captioner.captioning_predictor import CaptioningPredictor
    m = CaptioningPredictor()
    kats = [k for k in json.load(open(os.path.join(golden_dir, "perplexity_kat.json"))) if k["target_is_argmax"]]
    assert len(kats) == 3
    for k in kats:
        x = torch.tensor(k["input"])
        ppl = m.compute_perplexity(x.permute(1, 0, 2))             # exactly how the reference's KATs call it
        assert ppl.dtype == torch.float64
        assert torch.isclose(ppl, torch.tensor(k["expected"], dtype=torch.float64), rtol=1e-3)
        m.outputs["logits"] = [x[i] for i in range(x.shape[0])]    # list-of-steps form stored by forward()
        assert torch.isclose(m.compute_perplexity(), ppl, rtol=1e-6)


def test_shim_resolves_reference_import_paths():
    import embodied_captioning_amd.shim as shim
    shim.install()
    from experimenting_env.captioner.utils.utils import Configuration
    from experimenting_env.captioner.utils.utils_captioner import select_captioner
    from experimenting_env.utils.predictor_utils import Captioner
    cfg = Configuration(arch_name="blip", model_name="procedural-tiny:1", height=224, width=224)
    assert cfg.captioner.arch_name == "blip" and cfg.captioner.checkpoint_name is None
    assert callable(select_captioner) and Captioner is not None


def test_select_captioner_rejects_unknown_arch():
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    from embodied_captioning_amd.captioner.utils.utils_captioner import select_captioner
    with pytest.raises(AssertionError):
        select_captioner(Configuration(arch_name="florence", model_name="x").captioner)


def test_missing_checkpoint_raises_runtime_error_like_the_reference_factory():
    # reference factory.py:309-314 raises RuntimeError for an unknown pretrained tag
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    from embodied_captioning_amd.captioner.utils.utils_captioner import select_captioner
    with pytest.raises(RuntimeError):
        select_captioner(Configuration(arch_name="blip", model_name="/nonexistent/blip").captioner)


def test_shard_range_covers_every_frame_once():
    for n, w in ((50000, 8), (17, 4), (3, 8), (0, 2), (256, 1)):
        seen = []
        per0 = None
        for r in range(w):
            first, last, per = D.shard_range(n, r, w)
            per0 = per if per0 is None else per0
            assert per == per0 and last - first <= per
            seen += list(range(first, last))
        assert seen == list(range(n))


def test_consensus_grouping_matches_reference_semantics():
    keys = [(0, 1), (0, 1), (0, 2), (0, 1), (1, 1), (0, 2)]
    caps = ["a wooden chair", "a wooden chair", "a man on a sofa", "a brown chair", "a lamp", "a red sofa"]
    g = D.group_captions(keys, caps)
    assert g == {(0, 1): ["a wooden chair", "a wooden chair", "a brown chair"], (1, 1): ["a lamp"], (0, 2): ["a red sofa"]}
    f = D.captions_frequency(g)
    assert f[(0, 1)] == [[2, "a wooden chair"], [1, "a brown chair"]]
    assert D.consensus_caption(f[(0, 1)]) == "a wooden chair"
    assert not D.filter_caption("A Person standing") and D.filter_caption("a white table")


def test_state_dict_file_readers(tmp_path):
    from embodied_captioning_amd.weights import load_state_dict_file
    sd = {"module.a.weight": torch.randn(3, 2), "b": torch.arange(4.0)}
    p1 = tmp_path / "w.pt"
    torch.save({"model": sd}, p1)
    got = load_state_dict_file(str(p1))
    assert set(got) == {"a.weight", "b"} and torch.equal(got["b"], sd["b"])
    from safetensors.torch import save_file
    p2 = tmp_path / "w.safetensors"
    save_file({"a.weight": sd["module.a.weight"].contiguous()}, str(p2))
    assert torch.equal(load_state_dict_file(str(p2))["a.weight"], sd["module.a.weight"])


def test_expand_box_known_answers_from_the_reference_arithmetic():
    # cases worked out by hand from pseudolabeler.py:629-643
    from embodied_captioning_amd.pseudolabeler import expand_box, record_name
    assert expand_box((100, 200, 300, 400), 0.2, (1280, 1280, 3)).tolist() == [60, 160, 340, 440]
    assert expand_box((0, 0, 1280, 1280), 0.2, (1280, 1280, 3)).tolist() == [0, 0, 1280, 1280]
    assert expand_box((10.7, 5.2, 20.9, 15.5), 0.2, (480, 640, 3)).tolist() == [8, 3, 22, 17]     # int() truncation
    assert expand_box((600, 10, 700, 400), 0.2, (480, 640, 3)).tolist() == [580, 0, 480, 478]      # clamp quirk: x <- shape[0]
    assert record_name(3, 17) == "episode_3_step_17.npz"


def test_batched_box_captioner_keeps_frame_and_box_order():
    import numpy as np
    from embodied_captioning_amd.pseudolabeler import BatchedBoxCaptioner, crop_boxes
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8) for _ in range(3)]
    boxes = [[(4, 4, 20, 20), (30, 30, 60, 60)], [], [(0, 0, 64, 64)]]
    calls = []

    def fake(crops):                                   # caption = crop size + mean red value (after the BGR->RGB swap)
        calls.append(len(crops))
        return [f"{c.size[0]}x{c.size[1]}:{int(np.asarray(c)[..., 0].mean())}" for c in crops]

    out = BatchedBoxCaptioner(fake, encoder=lambda s: torch.full((4,), float(len(s)))).predict_captions(boxes, frames)
    assert calls == [3]                                # one call for the whole batch
    assert [len(o["captions"]) for o in out] == [2, 0, 1]
    c0 = crop_boxes(frames[0], boxes[0])
    assert out[0]["captions"][0].startswith(f"{c0[0].size[0]}x{c0[0].size[1]}")
    red = int(frames[2][..., 2].mean())                # channel 2 of BGR is red
    assert out[2]["captions"][0] == f"64x64:{red}"
    assert out[1]["embeddings"].numel() == 0 and out[0]["embeddings"].shape == (2, 4)


def test_save_record_round_trip(tmp_path):
    """Reference record format (pseudolabeler.py:833-842): one pickled dict {'instances', 'image'} under arr_0."""
    import numpy as np
    from embodied_captioning_amd.pseudolabeler import record_name, save_record
    inst = {"captions": ["a chair", "a lamp"], "embeddings": np.ones((2, 4), dtype=np.float32)}
    img = np.arange(24, dtype=np.uint8).reshape(2, 4, 3)
    f = save_record(str(tmp_path), record_name(3, 17)[:-4], inst, img)
    assert f.endswith("episode_3_step_17.npz")
    back = np.load(f, allow_pickle=True)["arr_0"].item()
    assert back["instances"]["captions"] == inst["captions"] and np.array_equal(back["image"], img)


def _toy_bpe(tmp_path):
    """A small CLIP-style BPE: merge rules in SimpleTokenizer's file format (header line, "left right" pairs)."""
    merges = [("t", "h"), ("th", "e</w>"), ("c", "a"), ("ca", "t</w>"), ("s", "a"), ("sa", "t</w>"), ("o", "n</w>"),
              ("m", "a"), ("ma", "t</w>"), ("Ã", "©</w>"), ("c", "a"), ("f", "Ã©</w>")]
    merges = list(dict.fromkeys(merges))
    text = "#version: toy\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n"
    return merges, text


def test_clip_bpe_decoder_rebuilds_the_simpletokenizer_vocabulary_and_decodes(tmp_path):
    """captioner/clip_bpe.py stands in for `open_clip.decode` (reference coca.py:30).  The vocabulary order is SimpleTokenizer's
    (256 byte symbols, the same + </w>, one symbol per merge, <start_of_text>, <end_of_text>); decoding maps symbols back to
    bytes, UTF-8 with errors=replace, </w> -> space."""
    import gzip
    from embodied_captioning_amd.captioner.clip_bpe import ClipBpeDecoder, bytes_to_unicode, vocab_from_merges
    b2u = bytes_to_unicode()
    assert len(b2u) == 256 and len(set(b2u.values())) == 256 and b2u[ord("a")] == "a" and b2u[ord(" ")] == chr(256 + 32)
    merges, text = _toy_bpe(tmp_path)
    with gzip.open(tmp_path / "bpe_simple_vocab_16e6.txt.gz", "wt", encoding="utf-8") as f:
        f.write(text)
    dec = ClipBpeDecoder.find(str(tmp_path))
    vocab = vocab_from_merges(merges)
    assert len(vocab) == 512 + len(merges) + 2 and vocab[-2:] == ["<start_of_text>", "<end_of_text>"]
    sid = {s: i for i, s in enumerate(vocab)}
    ids = [sid["<start_of_text>"], sid["the</w>"], sid["cat</w>"], sid["sat</w>"], sid["on</w>"], sid["the</w>"], sid["mat</w>"],
           sid["c"], sid["a"], sid["fÃ©</w>"], sid["<end_of_text>"], 0, 0]
    assert dec.decode(ids[:-2]) == "<start_of_text>the cat sat on the mat café <end_of_text>"
    assert dec.caption(ids) == "the cat sat on the mat café "                  # coca.py:30: cut at EOT, drop SOT (pad = '!' is cut off)
    assert dec.decode([sid["Ã"]]) == "�"                                    # half a UTF-8 sequence: errors="replace"
    # the same vocabulary as HF-style vocab.json
    import json
    d2 = tmp_path / "hf"
    d2.mkdir()
    (d2 / "vocab.json").write_text(json.dumps(sid), encoding="utf-8")
    assert ClipBpeDecoder.find(None, str(d2 / "model.safetensors")).decode(ids[:-2]) == dec.decode(ids[:-2])
    (tmp_path / "empty").mkdir()
    assert ClipBpeDecoder.find(str(tmp_path / "empty"), str(tmp_path / "empty" / "weights.pt")) is None


def test_clip_bpe_decoder_agrees_with_the_hf_clip_tokenizer_on_its_own_files(tmp_path):
    """Third-party pin of the decode algorithm: transformers' CLIPTokenizer built from the same vocab.json + merges.txt decodes
    the same ids to the same text (it strips the ends and spells the specials <|startoftext|> / <|endoftext|>)."""
    import json
    transformers = pytest.importorskip("transformers")
    from embodied_captioning_amd.captioner.clip_bpe import ClipBpeDecoder, vocab_from_merges
    merges, text = _toy_bpe(tmp_path)
    (tmp_path / "merges.txt").write_text(text, encoding="utf-8")
    vocab = vocab_from_merges(merges, ("<|startoftext|>", "<|endoftext|>"))
    sid = {s: i for i, s in enumerate(vocab)}
    (tmp_path / "vocab.json").write_text(json.dumps(sid), encoding="utf-8")
    try:
        tok = transformers.CLIPTokenizer(vocab=sid, merges=[tuple(m) for m in merges])         # transformers 5.x signature
        assert len(tok.get_vocab()) == len(sid)
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"CLIPTokenizer cannot be built from a vocabulary in this transformers version: {e!r}")
    ours = ClipBpeDecoder.from_vocab_json(str(tmp_path / "vocab.json"))
    ours_m = ClipBpeDecoder.from_merges_file(str(tmp_path / "merges.txt"))
    words = ["the</w>", "cat</w>", "sat</w>", "on</w>", "mat</w>", "c", "a", "fÃ©</w>", "t", "h", "e</w>"]
    import numpy as np
    rng = np.random.default_rng(0)
    for _ in range(20):
        ids = [sid[words[i]] for i in rng.integers(0, len(words), size=9)]
        want = tok.decode(ids, skip_special_tokens=False, clean_up_tokenization_spaces=False)
        assert ours.decode(ids).strip() == want.strip()
        assert ours_m.decode(ids) == ours.decode(ids)
    full = [sid["<|startoftext|>"], sid["the</w>"], sid["cat</w>"], sid["<|endoftext|>"]]
    assert ours.caption(full) == "the cat "


def test_kv16_guard_sees_outliers_through_the_post_layernorm_gamma():
    """weights.cross_kv_head_spread sizes a K/V dimension on the projection's INPUT distribution: BLIP's cross-attention reads
    post_layernorm(x) = xhat * gamma + beta, so a large gamma channel that one head dimension weights heavily is an outlier
    dimension of that head although the projection's rows alone look level."""
    from embodied_captioning_amd.config import BlipArch
    from embodied_captioning_amd.weights import KV16_MAX_HEAD_SPREAD, cross_kv_head_spread, procedural_blip_state_dict
    sd = procedural_blip_state_dict(BlipArch.tiny(), 0)
    assert cross_kv_head_spread(sd) < 1.5
    key = next(k for k in sd if k.endswith("crossattention.self.key.weight"))
    w = sd[key].clone(); w[5, 3] *= 30
    g = sd["vision_model.post_layernorm.weight"].clone(); g[3] *= 200
    assert cross_kv_head_spread({**sd, key: w}) < 3                                      # the rows alone: level
    assert cross_kv_head_spread({**sd, key: w, "vision_model.post_layernorm.weight": g}) > KV16_MAX_HEAD_SPREAD


def test_engine_pool_coalesce_plan():
    """EnginePool's dynamic batching plan: consecutive batches merged up to a row limit, never fewer passes than engines, the pass
    count a multiple of the engine count where the batches allow it, batches spread evenly and kept in order."""
    from embodied_captioning_amd.engine import EnginePool
    P = EnginePool.coalesce_plan
    assert [len(g) for g in P([256] * 20, 3, 1024)] == [4, 4, 3, 3, 3, 3]
    assert [len(g) for g in P([256] * 12, 3, 1024)] == [4, 4, 4]
    assert [len(g) for g in P([256] * 5, 3, 1024)] == [2, 2, 1]
    assert P([256] * 2, 3, 1024) == [[0], [1]]                       # two engines in parallel beat one merged pass
    assert P([256] * 4, 3, 256) == [[0], [1], [2], [3]] and P([300, 10], 2, 256) == [[0], [1]]
    for rows, n, cap in (([64] * 40, 3, 1024), ([100, 256, 30, 7, 256], 2, 600), ([256] * 7, 3, 512)):
        plan = P(rows, n, cap)
        assert [j for g in plan for j in g] == list(range(len(rows)))
        assert all(sum(rows[j] for j in g) <= max(cap, max(rows)) for g in plan)
    assert P([], 3, 1024) == []
