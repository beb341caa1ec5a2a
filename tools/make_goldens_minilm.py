#!/usr/bin/env python
"""Golden vectors for the caption-embedding step: the REAL HuggingFace BertModel (transformers, imported only in the
build container) on the seeded weights / token batches of embodied_captioning_amd.weights, followed by
sentence-transformers' Pooling(mean) + Normalize formulas.  Writes tests/golden/minilm_{tiny,base}.npz.

    python tools/make_goldens_minilm.py
"""
import dataclasses
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd.config import MiniLMArch  # noqa: E402
from embodied_captioning_amd.weights import procedural_minilm_state_dict, synthetic_token_batch  # noqa: E402


def run(arch: MiniLMArch, seed: int, batch: int, L: int):
    from transformers import BertConfig, BertModel
    sd = procedural_minilm_state_dict(arch, seed)
    ids, lens = synthetic_token_batch(arch, batch, L, seed)
    cfg = BertConfig(vocab_size=arch.vocab, hidden_size=arch.hidden, num_hidden_layers=arch.layers,
                     num_attention_heads=arch.heads, intermediate_size=arch.ffn, max_position_embeddings=arch.max_pos,
                     layer_norm_eps=arch.eps, hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    mask = (torch.arange(L)[None, :] < lens.long()[:, None]).long()
    with torch.no_grad():
        h = model(input_ids=ids.long(), attention_mask=mask).last_hidden_state
    m = mask[:, :, None].float()
    emb = torch.nn.functional.normalize((h * m).sum(1) / m.sum(1).clamp(min=1e-9), p=2, dim=1)
    meta = {"arch": dataclasses.asdict(arch), "seed": seed, "batch": batch, "L": L,
            "transformers": __import__("transformers").__version__}
    return {"ids": ids.numpy(), "lens": lens.numpy(), "embeddings": emb.numpy(),
            "hidden_first_row": h[0, : int(lens[0])].numpy(), "meta": json.dumps(meta)}


def main():
    gold = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(gold, "minilm_tiny.npz"), **run(MiniLMArch.tiny(), seed=5, batch=6, L=12))
    np.savez_compressed(os.path.join(gold, "minilm_base.npz"), **run(MiniLMArch(), seed=0, batch=16, L=24))
    print("done")


if __name__ == "__main__":
    main()
