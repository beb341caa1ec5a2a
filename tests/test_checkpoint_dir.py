"""The HuggingFace checkpoint-directory contract of the boundary (reference: blip2.py:19-22 `from_pretrained(cfg.model_name)`):
config.json + model.safetensors + tokenizer files.  No real checkpoint exists offline, so the directory is written here
from the seeded tiny weights with a synthetic WordPiece vocabulary."""
import json
import os

import numpy as np
import pytest
import torch

from embodied_captioning_amd.config import BlipArch
from embodied_captioning_amd.weights import (BLIP_TIED, load_hf_blip_checkpoint, procedural_blip_state_dict,
                                             synthetic_frames_u8)


def write_hf_dir(path, arch: BlipArch, sd, with_tokenizer=True):
    from safetensors.torch import save_file
    cfg = {"model_type": "blip",
           "vision_config": {"hidden_size": arch.v_hidden, "intermediate_size": arch.v_mlp, "num_hidden_layers": arch.v_layers,
                             "num_attention_heads": arch.v_heads, "image_size": arch.image_size, "patch_size": arch.patch_size,
                             "layer_norm_eps": arch.v_eps},
           "text_config": {"vocab_size": arch.vocab, "hidden_size": arch.t_hidden, "intermediate_size": arch.t_ffn,
                           "num_hidden_layers": arch.t_layers, "num_attention_heads": arch.t_heads,
                           "max_position_embeddings": arch.max_pos, "layer_norm_eps": arch.t_eps, "bos_token_id": arch.bos,
                           "sep_token_id": arch.eos, "pad_token_id": arch.pad}}
    os.makedirs(path, exist_ok=True)
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    # safetensors refuses shared storage: tied heads are left out, exactly like HF's own saves
    save_file({k: v.contiguous() for k, v in sd.items() if k not in BLIP_TIED}, os.path.join(path, "model.safetensors"))
    if with_tokenizer:
        vocab = ["[PAD]"] + [f"w{i}" for i in range(1, arch.vocab)]
        vocab[arch.eos] = "[SEP]"
        vocab[arch.bos] = "[DEC]"
        vocab[100 % arch.vocab] = "[UNK]"
        open(os.path.join(path, "vocab.txt"), "w").write("\n".join(vocab) + "\n")
        json.dump({"tokenizer_class": "BertTokenizer", "do_lower_case": True, "pad_token": "[PAD]", "sep_token": "[SEP]",
                   "unk_token": "[UNK]", "cls_token": "[DEC]", "mask_token": "[UNK]"},
                  open(os.path.join(path, "tokenizer_config.json"), "w"))


def test_hf_directory_round_trip(tmp_path):
    arch = BlipArch.tiny()
    sd = procedural_blip_state_dict(arch, 4, eos_boost=2.0)
    write_hf_dir(str(tmp_path / "ckpt"), arch, sd, with_tokenizer=False)
    a2, sd2 = load_hf_blip_checkpoint(str(tmp_path / "ckpt"))
    for f in ("image_size", "patch_size", "v_hidden", "v_layers", "v_heads", "v_mlp", "t_hidden", "t_layers", "t_heads",
              "t_ffn", "vocab", "max_pos", "bos", "eos", "pad"):
        assert getattr(a2, f) == getattr(arch, f), f
    assert set(sd2) == set(sd)                                         # tied heads restored
    for k in sd:
        assert torch.equal(sd2[k], sd[k]), k
    with pytest.raises(RuntimeError):
        os.remove(tmp_path / "ckpt" / "model.safetensors")
        load_hf_blip_checkpoint(str(tmp_path / "ckpt"))


@pytest.mark.gpu
def test_wrapper_from_hf_directory_with_tokenizer(tmp_path):
    """`arch_name: blip, model_name: <dir>`: weights, geometry and detokenisation all come from the directory; the text is
    the tokenizer's decode of the oracle's greedy tokens."""
    from PIL import Image
    from embodied_captioning_amd.captioner.utils.utils import Configuration
    from embodied_captioning_amd.captioner.utils.utils_captioner import select_captioner
    from oracle import blip_ref as R
    arch = BlipArch.tiny()
    sd = procedural_blip_state_dict(arch, 4, eos_boost=2.0)
    d = str(tmp_path / "ckpt")
    write_hf_dir(d, arch, sd)
    cfg = Configuration(arch_name="blip", model_name=d, height=224, width=224, dtype="f32", max_length=12).captioner
    model = select_captioner(cfg).eval()
    assert model.tokenizer is not None and model.arch.image_size == arch.image_size
    frame = synthetic_frames_u8(1, arch.image_size, arch.image_size, seed=9)[0].numpy()
    out = model(Image.fromarray(frame, "RGB"))
    px = ((torch.from_numpy(frame).float() / 255.0 - torch.tensor([0.48145466, 0.4578275, 0.40821073])) /
          torch.tensor([0.26862954, 0.26130258, 0.27577711])).permute(2, 0, 1)[None]
    ref = R.greedy_generate(sd, arch, px, 12)["sequences"][0].tolist()
    words = [f"w{t}" for t in ref[1:] if t not in (arch.eos, arch.pad)]
    assert out["text"] == " ".join(words), (out["text"], words)
    assert len(out["logits"]) >= 1 and out["logits"][0].shape == (1, arch.vocab)


def test_blip2_hf_directory_round_trip_with_shards(tmp_path):
    """`Salesforce/blip2-opt-2.7b` layout: config.json (model_type blip-2), generation_config.json with the EOS id generate()
    uses, weights sharded over model-0000x-of-0000y.safetensors, tied lm_head absent from the files."""
    from safetensors.torch import save_file
    from embodied_captioning_amd.captioner.models.blip2.blip2 import load_hf_blip2_checkpoint
    from embodied_captioning_amd.config import Blip2Arch
    from embodied_captioning_amd.weights import procedural_blip2_state_dict
    a = Blip2Arch.tiny()
    sd = procedural_blip2_state_dict(a, 2)
    d = tmp_path / "blip2"
    os.makedirs(d)
    cfg = {"model_type": "blip-2", "num_query_tokens": a.num_query_tokens, "image_token_index": a.image_token,
           "vision_config": {"hidden_size": a.v_hidden, "intermediate_size": a.v_mlp, "num_hidden_layers": a.v_layers,
                             "num_attention_heads": a.v_heads, "image_size": a.image_size, "patch_size": a.patch_size, "layer_norm_eps": a.v_eps},
           "qformer_config": {"hidden_size": a.q_hidden, "num_hidden_layers": a.q_layers, "num_attention_heads": a.q_heads,
                              "intermediate_size": a.q_ffn, "cross_attention_frequency": a.q_cross_freq, "layer_norm_eps": a.q_eps},
           "text_config": {"model_type": "opt", "hidden_size": a.t_hidden, "num_hidden_layers": a.t_layers, "num_attention_heads": a.t_heads,
                           "ffn_dim": a.t_ffn, "vocab_size": a.vocab, "max_position_embeddings": a.max_pos, "word_embed_proj_dim": a.t_hidden,
                           "do_layer_norm_before": True, "bos_token_id": 2, "eos_token_id": 2, "pad_token_id": 1}}
    json.dump(cfg, open(d / "config.json", "w"))
    json.dump({"bos_token_id": 2, "eos_token_id": a.eos, "pad_token_id": 1}, open(d / "generation_config.json", "w"))
    keys = [k for k in sd if k != "language_model.lm_head.weight"]
    half = len(keys) // 2
    save_file({k: sd[k].contiguous() for k in keys[:half]}, str(d / "model-00001-of-00002.safetensors"))
    save_file({k: sd[k].contiguous() for k in keys[half:]}, str(d / "model-00002-of-00002.safetensors"))
    a2, sd2 = load_hf_blip2_checkpoint(str(d))
    for f in ("image_size", "patch_size", "v_hidden", "v_layers", "v_heads", "v_mlp", "q_hidden", "q_layers", "q_heads", "q_ffn",
              "q_cross_freq", "num_query_tokens", "t_hidden", "t_layers", "t_heads", "t_ffn", "vocab", "max_pos", "bos", "eos", "pad",
              "image_token"):
        assert getattr(a2, f) == getattr(a, f), f
    assert set(sd2) == set(sd)
    assert all(torch.equal(sd2[k], sd[k]) for k in sd)
    cfg["text_config"]["model_type"] = "t5"
    json.dump(cfg, open(d / "config.json", "w"))
    with pytest.raises(RuntimeError):
        load_hf_blip2_checkpoint(str(d))
