#!/usr/bin/env python
"""Golden vectors for the BLIP-2 OPT path: the REAL HuggingFace Blip2ForConditionalGeneration (imported only in the build
container) on the seeded weights / frames of embodied_captioning_amd.weights.  Writes tests/golden/blip2_tiny.npz.

    python tools/make_goldens_blip2.py
"""
import dataclasses
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd.config import Blip2Arch  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip2_state_dict, synthetic_pixels  # noqa: E402


def build_hf(a: Blip2Arch, sd):
    from transformers import Blip2Config, Blip2ForConditionalGeneration, Blip2QFormerConfig, Blip2VisionConfig, OPTConfig
    v = Blip2VisionConfig(hidden_size=a.v_hidden, intermediate_size=a.v_mlp, num_hidden_layers=a.v_layers,
                          num_attention_heads=a.v_heads, image_size=a.image_size, patch_size=a.patch_size, layer_norm_eps=a.v_eps)
    q = Blip2QFormerConfig(hidden_size=a.q_hidden, num_hidden_layers=a.q_layers, num_attention_heads=a.q_heads,
                           intermediate_size=a.q_ffn, encoder_hidden_size=a.v_hidden, cross_attention_frequency=a.q_cross_freq,
                           layer_norm_eps=a.q_eps, vocab_size=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    t = OPTConfig(vocab_size=a.vocab, hidden_size=a.t_hidden, num_hidden_layers=a.t_layers, ffn_dim=a.t_ffn,
                  num_attention_heads=a.t_heads, max_position_embeddings=a.max_pos, word_embed_proj_dim=a.t_hidden,
                  bos_token_id=a.bos, eos_token_id=a.eos, pad_token_id=a.pad, dropout=0.0)
    c = Blip2Config(vision_config=v.to_dict(), qformer_config=q.to_dict(), text_config=t.to_dict(),
                    num_query_tokens=a.num_query_tokens, image_token_index=a.image_token)
    m = Blip2ForConditionalGeneration(c).eval()
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert not res.missing_keys, res.missing_keys
    return m


def run(a: Blip2Arch, seed: int, batch: int, eos_boost: float):
    sd = procedural_blip2_state_dict(a, seed, eos_boost=eos_boost)
    m = build_hf(a, sd)
    px = synthetic_pixels(batch, a.image_size, seed=seed)
    with torch.no_grad():
        out = m.generate(pixel_values=px, output_logits=True, return_dict_in_generate=True)
        emb = m.vision_model(px).last_hidden_state
        qo = m.qformer(query_embeds=m.query_tokens.expand(batch, -1, -1), encoder_hidden_states=emb).last_hidden_state
    logits = torch.stack(list(out.logits), 0)                               # [T, B, V]
    top = torch.topk(logits, 2, dim=-1).values
    print("sequences", out.sequences[:, a.num_query_tokens:].tolist())
    return {"sequences": out.sequences.numpy().astype(np.int32), "logits": logits.numpy(), "margin": (top[..., 0] - top[..., 1]).numpy(),
            "image_embeds": emb.numpy(), "query_output": qo.numpy(),
            "meta": np.array(json.dumps(dict(seed=seed, batch=batch, eos_boost=eos_boost, arch=dataclasses.asdict(a),
                                             transformers=__import__("transformers").__version__)))}


def main():
    gold = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(gold, "blip2_tiny.npz"), **run(Blip2Arch.tiny(), seed=11, batch=4, eos_boost=0.5))
    print("done")


if __name__ == "__main__":
    main()
