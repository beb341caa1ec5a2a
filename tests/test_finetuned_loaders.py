"""The fine-tuned-checkpoint loaders of the boundary.  Reference: scripts/evaluate_finetuned_model.py:139-148 (CoCa: `checkpoint[
'state_dict']` with `module.` stripped; BLIP-2: `PeftModel.from_pretrained(model, ckpt_path)` on an fp16 base model) and
captioner/models/blip2/blip2.py:19-22 (`load_in_8bit=True, torch_dtype=torch.float16`).  CPU tests: the LoRA merge against the
formula PEFT's adapted Linear computes, the rejections by name; GPU test: an fp16-stored checkpoint enters the split format exactly."""
import json
import os

import numpy as np
import pytest
import torch

from embodied_captioning_amd.weights import is_peft_adapter, merge_peft_lora


def _write_adapter(path, tensors, **cfg):
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    base = {"peft_type": "LORA", "r": 4, "lora_alpha": 8, "target_modules": ["q_proj", "v_proj"], "bias": "none", "fan_in_fan_out": False}
    base.update(cfg)
    json.dump(base, open(os.path.join(path, "adapter_config.json"), "w"))
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(path, "adapter_model.safetensors"))


def _base(seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for i in range(2):
        for m in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[f"language_model.model.decoder.layers.{i}.self_attn.{m}.weight"] = torch.randn(24, 24, generator=g) * 0.1
            sd[f"language_model.model.decoder.layers.{i}.self_attn.{m}.bias"] = torch.randn(24, generator=g) * 0.1
    sd["language_model.lm_head.weight"] = torch.randn(50, 24, generator=g)
    return sd


def _adapter(sd, r=4, seed=1, mods=("q_proj", "v_proj"), name_infix=""):
    g = torch.Generator().manual_seed(seed)
    ad = {}
    for k in sd:
        if k.endswith(".weight") and any(f".{m}." in k for m in mods):
            mod = k[: -len(".weight")]
            ad[f"base_model.model.{mod}.lora_A{name_infix}.weight"] = torch.randn(r, sd[k].shape[1], generator=g) * 0.3
            ad[f"base_model.model.{mod}.lora_B{name_infix}.weight"] = torch.randn(sd[k].shape[0], r, generator=g) * 0.3
    return ad


@pytest.mark.parametrize("rslora", [False, True])
@pytest.mark.parametrize("infix", ["", ".default"])
def test_lora_merge_is_the_adapted_linear(tmp_path, rslora, infix):
    """W' = W + scaling * B A with scaling = alpha / r (alpha / sqrt(r) under rsLoRA): the merged Linear gives what PEFT's adapted
    forward - base(x) + scaling * B(A(x)) - gives, in float64 to 1e-12, and untouched modules keep their bits."""
    sd = _base()
    ad = _adapter(sd, name_infix=infix)
    _write_adapter(str(tmp_path / "a"), ad, r=4, lora_alpha=8, use_rslora=rslora)
    assert is_peft_adapter(str(tmp_path / "a")) and not is_peft_adapter(str(tmp_path))
    out, rep = merge_peft_lora(sd, str(tmp_path / "a"))
    assert rep["merged"] == 4 and rep["replaced"] == 0
    s = 8 / np.sqrt(4) if rslora else 8 / 4
    x = torch.randn(5, 24, generator=torch.Generator().manual_seed(3)).double()
    for k, w in sd.items():
        if k.endswith("q_proj.weight") or k.endswith("v_proj.weight"):
            mod = k[: -len(".weight")]
            A = ad[f"base_model.model.{mod}.lora_A{infix}.weight"].double()
            B = ad[f"base_model.model.{mod}.lora_B{infix}.weight"].double()
            want = x @ w.double().T + s * ((x @ A.T) @ B.T)
            got = x @ out[k].double().T
            assert (got - want).abs().max().item() < 1e-5          # the merge itself is fp32
            assert torch.equal(out[k], w + (B.float() @ A.float()) * s)
            assert abs(rep["scaling"][mod] - s) < 1e-12
        else:
            assert torch.equal(out[k], w), k


def test_lora_merge_patterns_transposition_and_saved_modules(tmp_path):
    sd = _base()
    ad = _adapter(sd, r=2, mods=("q_proj",))
    # fan_in_fan_out adapters hold the factors of W^T
    ad_t = {k: v for k, v in ad.items()}
    head = torch.randn(50, 24, generator=torch.Generator().manual_seed(9))
    ad_t["base_model.model.language_model.lm_head.modules_to_save.default.weight"] = head
    _write_adapter(str(tmp_path / "a"), ad_t, r=8, lora_alpha=16, rank_pattern={"layers.0.self_attn.q_proj": 2, "q_proj": 2},
                   alpha_pattern={"layers.1.self_attn.q_proj": 6}, modules_to_save=["lm_head"])
    out, rep = merge_peft_lora(sd, str(tmp_path / "a"))
    assert rep == {"merged": 2, "replaced": 1, "scaling": rep["scaling"]}
    assert abs(rep["scaling"]["language_model.model.decoder.layers.0.self_attn.q_proj"] - 16 / 2) < 1e-12
    assert abs(rep["scaling"]["language_model.model.decoder.layers.1.self_attn.q_proj"] - 6 / 2) < 1e-12
    assert torch.equal(out["language_model.lm_head.weight"], head)
    _write_adapter(str(tmp_path / "t"), ad, r=2, lora_alpha=2, fan_in_fan_out=True)
    out_t, _ = merge_peft_lora(sd, str(tmp_path / "t"))
    k = "language_model.model.decoder.layers.0.self_attn.q_proj.weight"
    A, B = ad["base_model.model." + k[:-7] + ".lora_A.weight"], ad["base_model.model." + k[:-7] + ".lora_B.weight"]
    assert torch.equal(out_t[k], sd[k] + (B @ A).t())


@pytest.mark.parametrize("cfg,tensor,msg", [
    ({"peft_type": "IA3"}, None, "IA3"),
    ({"use_dora": True}, None, "use_dora"),
    ({"bias": "lora_only"}, None, "bias='lora_only'"),
    ({}, "base_model.model.language_model.model.decoder.embed_tokens.lora_embedding_A", "lora_embedding"),
    ({}, "base_model.model.not.a.module.lora_A.weight", "no weight"),
])
def test_lora_merge_rejects_by_name(tmp_path, cfg, tensor, msg):
    sd = _base()
    ad = _adapter(sd)
    if tensor:
        ad[tensor] = torch.zeros(4, 24)
        if tensor.endswith("lora_A.weight"):
            ad[tensor.replace("lora_A", "lora_B")] = torch.zeros(24, 4)
    _write_adapter(str(tmp_path / "a"), ad, **cfg)
    with pytest.raises(RuntimeError, match=msg.replace("'", ".")):
        merge_peft_lora(sd, str(tmp_path / "a"))
    with pytest.raises(RuntimeError, match="adapter_config.json"):
        merge_peft_lora(sd, str(tmp_path / "nothing-here"))


def test_generation_options_are_rejected_by_name():
    """coca_model.py:205-224 / :236-241 / :266-275: temperature, top_p, top_k > 1, repetition_penalty are not implemented -
    a caller that asks for them gets a ValueError citing the option, never a silently different decode."""
    from embodied_captioning_amd.captioner.generation_options import reject_unsupported_generation_options as check
    from embodied_captioning_amd.captioner.models.coca.coca import CoCa
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    check({"generation_type": "top_k", "top_k": 1, "top_p": 0.1, "temperature": 1.0, "repetition_penalty": 1.0})   # the reference's own defaults
    check({"generation_type": "beam_search", "num_beams": 6, "num_beam_groups": 3})
    for opts, word in (({"top_k": 5}, "top_k=5"), ({"generation_type": "top_p", "top_p": 0.1}, "top_p"), ({"temperature": 0.7}, "temperature=0.7"),
                       ({"repetition_penalty": 1.3}, "repetition_penalty=1.3"), ({"do_sample": True}, "do_sample"),
                       ({"generation_type": "contrastive"}, "generation_type"), ({"stopping_criteria": [object()]}, "stopping_criteria"),
                       ({"text": torch.zeros(1, 3)}, "text=")):
        with pytest.raises(ValueError, match=word):
            check(opts, "test")
    # through the wrapper's configuration: raised before anything touches a GPU
    for key, val, word in (("top_k", 5, "top_k=5"), ("temperature", 0.5, "temperature=0.5"), ("repetition_penalty", 2.0, "repetition_penalty=2.0"),
                           ("generation_type", "top_p", "generation_type")):
        cfg = Configuration(arch_name="coca", model_name="procedural-coca-tiny:1", height=224, width=224, **{key: val}).captioner
        with pytest.raises(ValueError, match=word):
            CoCa(cfg)


def test_blip2_rejects_4bit_loading_and_contradicting_dtype_by_name():
    """blip2.py:19-22 asks for bitsandbytes int8 weights: `load_in_8bit` is the int8-weight mode with bf16 activations (tests/
    test_blip2_int8_gpu.py); NF4 and a dtype that contradicts it raise instead of changing the arithmetic silently."""
    from embodied_captioning_amd.captioner.models.blip2.blip2 import BLIP2
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    cfg = Configuration(arch_name="blip2", model_name="procedural-blip2-tiny:1", height=224, width=224, load_in_4bit=True).captioner
    with pytest.raises(ValueError, match="load_in_4bit"):
        BLIP2(cfg)
    cfg = Configuration(arch_name="blip2", model_name="procedural-blip2-tiny:1", height=224, width=224, load_in_8bit=True, dtype="f32s").captioner
    with pytest.raises(ValueError, match="load_in_8bit"):
        BLIP2(cfg)
    cfg = Configuration(arch_name="blip2", model_name="procedural-blip2-tiny:1", height=224, width=224, torch_dtype="int8").captioner
    with pytest.raises(ValueError, match="torch_dtype"):
        BLIP2(cfg)


@pytest.mark.gpu
def test_fp16_stored_checkpoint_enters_the_split_format_exactly_and_lora_adapter_loads(tmp_path):
    """An fp16-stored checkpoint (the reference loads BLIP-2 with torch_dtype=float16): every weight is an fp16 value x, the split
    format stores hi = 4096 x (exact: a power of two) and lo = 0 - nothing is rounded at load; and the wrapper with
    `checkpoint_name` = a PEFT directory decodes with the merged weights."""
    import ctypes as C
    from _util import g8_decode
    from embodied_captioning_amd import _native as N
    from embodied_captioning_amd.captioner.models.blip2.blip2 import BLIP2
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    from embodied_captioning_amd.config import Blip2Arch
    from embodied_captioning_amd.engine import CaptionerEngine
    from embodied_captioning_amd.weights import procedural_blip2_state_dict, synthetic_pixels
    lib = N.load_library()
    x16 = (torch.randn(64, 256, generator=torch.Generator().manual_seed(0)) * 0.05).half()
    x16[0, :8] = torch.tensor([6e-8, -6e-8, 1e-5, 15.8, -15.8, 0.0, 3e-3, -7e-4]).half()       # subnormal, tiny, the range's edge
    src = x16.float().cuda()
    dst = torch.zeros(64, 256, dtype=torch.float32, device="cuda")
    N.check(lib.cap_op_convert_weight(N.CAP_F32_SPLIT, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), 64, 256,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "cap_op_convert_weight")
    torch.cuda.synchronize()
    raw = dst.cpu().numpy()
    halves = raw.view(np.float16).reshape(64, -1, 2, 8)
    assert not halves[:, :, 1, :].any()                                   # every lo half is zero
    assert np.array_equal(g8_decode(raw, 4096.0), x16.float().numpy())      # and hi / 4096 is the fp16 value
    # the same captions from the fp16-stored dict and from its fp32 copy
    a = Blip2Arch.tiny()
    sd = {k: v.half() for k, v in procedural_blip2_state_dict(a, 3, eos_boost=0.5).items()}
    px = synthetic_pixels(2, a.image_size, seed=3).cuda()
    outs = []
    for d in (sd, {k: v.float() for k, v in sd.items()}):
        eng = CaptionerEngine(a, dtype="f32s", max_batch=2, max_beams=1, max_len=a.max_new_tokens)
        eng.load_state_dict(d)
        outs.append(eng.generate(px, max_length=a.max_new_tokens, output_logits=True))
        eng.close()
    assert torch.equal(outs[0]["sequences"], outs[1]["sequences"]) and torch.equal(outs[0]["logits"], outs[1]["logits"])
    # PEFT adapter through the wrapper: equals an engine loaded with the merged dict, differs from the base model
    base = procedural_blip2_state_dict(a, 3, eos_boost=0.5)
    g = torch.Generator().manual_seed(5)
    ad = {}
    for k, v in base.items():
        if k.startswith("language_model.") and (k.endswith("q_proj.weight") or k.endswith("v_proj.weight")):
            ad[f"base_model.model.{k[:-7]}.lora_A.weight"] = torch.randn(4, v.shape[1], generator=g) * 0.2
            ad[f"base_model.model.{k[:-7]}.lora_B.weight"] = torch.randn(v.shape[0], 4, generator=g) * 0.2
    assert ad
    _write_adapter(str(tmp_path / "lora"), ad, r=4, lora_alpha=16)
    merged, rep = merge_peft_lora(base, str(tmp_path / "lora"))
    eng = CaptionerEngine(a, dtype="f32s", max_batch=2, max_beams=1, max_len=a.max_new_tokens)
    eng.load_state_dict(merged)
    want = eng.generate(px, max_length=a.max_new_tokens, output_logits=True)
    eng.close()
    cfg = Configuration(arch_name="blip2", model_name="procedural-blip2-tiny:3:0.5", checkpoint_name=str(tmp_path / "lora"),
                        height=224, width=224, dtype="f32s", batch_size=2).captioner
    model = BLIP2(cfg)
    assert model.adapter_report["merged"] == rep["merged"] > 0
    got = model.engine.generate(px, max_length=a.max_new_tokens, output_logits=True)
    assert torch.equal(got["logits"], want["logits"])
    assert not torch.equal(got["logits"][0], outs[1]["logits"][0])
    model.engine.close()
