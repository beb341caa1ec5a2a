"""GPU: the reference-shaped plugin API (BLIP wrapper, Captioner) end to end on PIL input, against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pil(seed, size=(48, 40)):
    from PIL import Image
    rng = np.random.default_rng(seed)
    return Image.fromarray(rng.integers(0, 256, size=(size[1], size[0], 3), dtype=np.uint8), "RGB")


def test_blip_wrapper_forward_matches_oracle_on_pil_crop():
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    from embodied_captioning_amd.captioner.utils.utils_captioner import select_captioner
    from embodied_captioning_amd.config import BlipArch
    from embodied_captioning_amd.engine import OPENAI_CLIP_MEAN, OPENAI_CLIP_STD
    from embodied_captioning_amd.weights import procedural_blip_state_dict
    from oracle import blip_ref as R
    from PIL import Image
    cfg = Configuration(arch_name="blip", model_name="procedural-tiny:4:2.0", height=224, width=224, dtype="f32",
                        max_length=12).captioner
    model = select_captioner(cfg).eval()
    im = _pil(1)
    out = model(im)
    assert set(out) == {"text", "logits"} and isinstance(out["text"], str)
    # oracle on the same preprocessing (bicubic resize to the model size, /255, CLIP mean/std)
    arch = BlipArch.tiny()
    sd = procedural_blip_state_dict(arch, 4, eos_boost=2.0)
    u8 = torch.from_numpy(np.asarray(im.resize((arch.image_size,) * 2, resample=Image.BICUBIC)))
    px = (u8.permute(2, 0, 1)[None].float() / 255.0 - torch.tensor(OPENAI_CLIP_MEAN).view(1, 3, 1, 1)) / \
        torch.tensor(OPENAI_CLIP_STD).view(1, 3, 1, 1)
    ref = R.greedy_generate(sd, arch, px, 12)
    ids = [int(t) for t in ref["sequences"][0].tolist() if t not in (arch.bos, arch.eos, arch.pad)]
    assert out["text"] == " ".join(str(i) for i in ids)
    n_steps = ref["sequences"].shape[1] - 1
    assert len(out["logits"]) == n_steps and out["logits"][0].shape == (1, arch.vocab)
    for t in range(n_steps):
        assert (out["logits"][t].cpu() - ref["logits"][t]).abs().max().item() < 1e-3
    # perplexity through the base-class API equals the oracle's on the same logits
    ppl = model.compute_perplexity()
    assert torch.isclose(ppl, R.compute_perplexity(ref["logits"]), rtol=1e-4)
    # outputs are fresh objects per call (callers keep references across calls)
    out2 = model(_pil(2))
    assert out2 is not out and out2["logits"] is not out["logits"]


def test_captioner_plugin_returns_str_and_batches():
    import types
    from embodied_captioning_amd.utils.predictor_utils import Captioner
    cap_cfg = types.SimpleNamespace(arch_name="blip", model_name="procedural-tiny:4:2.0", checkpoint_name=None,
                                    height=224, width=224, dtype="bf16", max_length=12, batch_size=4)
    cfg = types.SimpleNamespace(captioner=cap_cfg)
    cap = Captioner(cfg).to("cuda:0").eval()
    s = cap(_pil(3))
    assert isinstance(s, str)
    many = cap.caption_batch([_pil(i) for i in range(3, 9)])      # 6 crops, micro-batches of 4
    assert len(many) == 6 and many[0] == s


def test_box_driver_device_resize_equals_pil_path():
    """The batched box driver with crop + resize on the device (bit-exact Pillow bicubic) returns the captions of the host
    PIL path: same crop rectangles (expand_box), same BGR->RGB swap, same pixels, hence the same tokens."""
    import types
    from embodied_captioning_amd.pseudolabeler import BatchedBoxCaptioner
    from embodied_captioning_amd.utils.predictor_utils import Captioner
    cap_cfg = types.SimpleNamespace(arch_name="blip", model_name="procedural-tiny:4:2.0", checkpoint_name=None,
                                    height=224, width=224, dtype="f32", max_length=12, batch_size=4)
    model = Captioner(types.SimpleNamespace(captioner=cap_cfg)).to("cuda:0").eval()
    assert model.direct_resize_size == model.model.arch.image_size
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, size=(120, 160, 3), dtype=np.uint8) for _ in range(3)]
    boxes = [[(10, 20, 60, 90), (100, 5, 158, 60)], [], [(0, 0, 160, 120), (40, 40, 44, 47), (70, 30, 130, 110)]]
    dev = BatchedBoxCaptioner(model, device_resize=True).predict_captions(boxes, frames)
    pil = BatchedBoxCaptioner(model, device_resize=False).predict_captions(boxes, frames)
    assert [d["captions"] for d in dev] == [p["captions"] for p in pil]
    assert [len(d["captions"]) for d in dev] == [2, 0, 3]
    assert BatchedBoxCaptioner(model).device_resize is True            # the default picks the device path for this plugin


def test_plugin_streams_option_gives_the_same_captions():
    """cfg.streams = 3: the micro-batches of one caption_batch call rotate over three engines / streams (a host thread each,
    early-exit polling on): the captions are those of the single-engine plugin."""
    import types
    from embodied_captioning_amd.utils.predictor_utils import Captioner
    crops = [_pil(i) for i in range(20, 31)]                      # 11 crops, micro-batches of 2 -> 6 batches over 3 engines

    def build(streams):
        cap_cfg = types.SimpleNamespace(arch_name="blip", model_name="procedural-tiny:4:2.0", checkpoint_name=None,
                                        height=224, width=224, dtype="f32", max_length=12, batch_size=2, streams=streams)
        return Captioner(types.SimpleNamespace(captioner=cap_cfg)).to("cuda:0").eval()
    one, three = build(1), build(3)
    assert three.model.pool is not None and len(three.model.pool) == 3 and one.model.pool is None
    assert three.caption_batch(crops) == one.caption_batch(crops)


def test_captioner_load_checkpoint_reaches_engine_and_pool(tmp_path):
    """`Captioner(load_checkpoint=True, checkpoint_path=...)` - reference predictor_utils.py:182-185 loads
    `checkpoint['model']` into the WRAPPER, so its keys carry a `model.` prefix (DDP: `module.model.`).  The weights must
    reach the engine and every replica of the stream pool; a dict with no tensor of the architecture must raise, not log
    success over the base weights."""
    import types
    from embodied_captioning_amd._native import CaptionerHipError
    from embodied_captioning_amd.config import BlipArch
    from embodied_captioning_amd.utils.predictor_utils import Captioner
    from embodied_captioning_amd.weights import procedural_blip_state_dict
    arch = BlipArch.tiny()
    new = procedural_blip_state_dict(arch, 11, eos_boost=2.0)
    path = str(tmp_path / "finetuned.pt")
    torch.save({"model": {"module.model." + k: v for k, v in new.items()}}, path)

    def plugin(model_name, **kw):
        cap_cfg = types.SimpleNamespace(arch_name="blip", model_name=model_name, checkpoint_name=None, height=224, width=224,
                                        dtype="f32s", max_length=12, batch_size=2, streams=2)
        return Captioner(types.SimpleNamespace(captioner=cap_cfg), **kw)

    crops = [_pil(i) for i in range(20, 26)]
    want = plugin("procedural-tiny:11:2.0").caption_batch(crops)            # built directly from the new weights
    base = plugin("procedural-tiny:4:2.0")
    assert base.caption_batch(crops) != want
    cap = plugin("procedural-tiny:4:2.0", load_checkpoint=True, checkpoint_path=path)
    assert cap.caption_batch(crops) == want                                 # 3 micro-batches over the 2 pool engines
    assert [cap(c) for c in crops[:2]] == want[:2]                          # forward() runs on the wrapper's own engine
    bad = str(tmp_path / "other.pt")
    torch.save({"model": {"backbone.layer.weight": torch.zeros(3, 3)}}, bad)
    with pytest.raises(CaptionerHipError):
        plugin("procedural-tiny:4:2.0", load_checkpoint=True, checkpoint_path=bad)


def test_generate_logits_tail_is_zero_after_early_exit():
    """With early exit the steps after the last executed one are never written: callers must see zeros, not stale memory."""
    from embodied_captioning_amd.config import BlipArch
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels
    arch = BlipArch.tiny()
    eng = CaptionerEngine(arch, dtype="f32s", max_batch=3, max_beams=1, max_len=20)
    eng.load_state_dict(procedural_blip_state_dict(arch, seed=5, eos_boost=12.0))
    eng.set_early_exit(2)
    out = eng.generate(synthetic_pixels(3, arch.image_size, seed=9).cuda(), max_length=20, output_logits=True)
    n = eng.last_decode_steps
    assert n < 19 and float(out["logits"][n:].abs().max()) == 0.0 and float(out["logits"][0].abs().max()) > 0.0
    eng.close()


def test_generate_batch_long_pil_list_is_preprocessed_in_rounds_same_captions():
    """A PIL list longer than one round of passes (engines x pass size) goes through `generate_batch` in rounds - the next round's crops are
    preprocessed by a helper thread while the current round generates; captions are those of the one-engine wrapper."""
    import numpy as np
    import torch
    from PIL import Image
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    from embodied_captioning_amd.captioner.utils.utils_captioner import select_captioner
    rng = np.random.default_rng(21)
    ims = [Image.fromarray(rng.integers(0, 256, size=(30 + i % 17, 40 + i % 11, 3), dtype=np.uint8), "RGB") for i in range(83)]
    kw = dict(arch_name="blip", model_name="procedural-blip-tiny:3:2.0", height=224, width=224, dtype="f32s", batch_size=4)
    one = select_captioner(Configuration(**kw).captioner).eval()
    many = select_captioner(Configuration(streams=2, **kw).captioner).eval()
    assert many.pool is not None and len(ims) > len(many.pool) * max(many.batch_size, many.coalesce_rows)       # 83 > 2 x 16: three rounds
    a, b = one.generate_batch(ims), many.generate_batch(ims)
    assert torch.equal(a["sequences"], b["sequences"]) and torch.equal(a["lengths"], b["lengths"]) and a["texts"] == b["texts"]
