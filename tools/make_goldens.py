#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REAL third-party implementation (HF transformers 5.15.0
`BlipForConditionalGeneration`, CPU fp32) that the reference's BLIP-family wrappers delegate to
(experimenting_env/captioner/models/blip2/blip2.py:19-28).  Run in the build container only:

    python tools/make_goldens.py            # writes tests/golden/blip_tiny.npz, blip_tiny_eos.npz, blip_base.npz, perplexity_kat.json

The fixtures hold inputs' seeds and expected outputs (data only).  Weights are not stored: they are
re-drawn from `embodied_captioning_amd.weights.procedural_blip_state_dict(arch, seed)`.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from embodied_captioning_amd.config import BlipArch                      # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402


def build_hf(arch: BlipArch, sd):
    from transformers import BlipConfig, BlipForConditionalGeneration
    cfg = BlipConfig(
        vision_config=dict(hidden_size=arch.v_hidden, intermediate_size=arch.v_mlp, num_hidden_layers=arch.v_layers,
                           num_attention_heads=arch.v_heads, image_size=arch.image_size, patch_size=arch.patch_size,
                           layer_norm_eps=arch.v_eps),
        text_config=dict(vocab_size=arch.vocab, hidden_size=arch.t_hidden, encoder_hidden_size=arch.v_hidden,
                         intermediate_size=arch.t_ffn, num_hidden_layers=arch.t_layers,
                         num_attention_heads=arch.t_heads, max_position_embeddings=arch.max_pos,
                         layer_norm_eps=arch.t_eps, bos_token_id=arch.bos, sep_token_id=arch.eos,
                         pad_token_id=arch.pad),
    )
    model = BlipForConditionalGeneration(cfg).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m for m in missing), missing
    return model


def run(arch: BlipArch, seed: int, batch: int, max_length: int, beams: int, full: bool, eos_boost: float):
    sd = procedural_blip_state_dict(arch, seed, eos_boost=eos_boost)
    model = build_hf(arch, sd)
    pixels = synthetic_pixels(batch, arch.image_size, seed=seed)
    out = {}
    with torch.no_grad():
        t0 = time.time()
        vis = model.vision_model(pixel_values=pixels, output_hidden_states=True)
        embeds = vis[0]
        g = model.generate(pixel_values=pixels, max_length=max_length, num_beams=1, do_sample=False,
                           output_logits=True, return_dict_in_generate=True)
        t1 = time.time()
        b = model.generate(pixel_values=pixels, max_length=max_length, num_beams=beams, do_sample=False,
                           length_penalty=1.0, early_stopping=False,
                           output_scores=True, return_dict_in_generate=True)
        t2 = time.time()
    print(f"  HF encoder+greedy {t1 - t0:.2f}s, beam-{beams} {t2 - t1:.2f}s, batch {batch}")
    logits = torch.stack(list(g.logits), dim=0)                        # [T, B, V]
    top = torch.topk(logits, k=8, dim=-1)
    out["greedy_sequences"] = g.sequences.numpy().astype(np.int32)
    out["greedy_top8_ids"] = top.indices.numpy().astype(np.int32)
    out["greedy_top8_vals"] = top.values.numpy()
    out["greedy_margin"] = (top.values[..., 0] - top.values[..., 1]).numpy()
    out["greedy_logsumexp"] = torch.logsumexp(logits, dim=-1).numpy()
    out["beam_sequences"] = b.sequences.numpy().astype(np.int32)
    out["beam_scores"] = b.sequences_scores.numpy()
    # encoder: strided sample + per-token L2 norms (full tensor only for the tiny config)
    flat = embeds.reshape(batch, -1)
    out["embeds_sample_stride"] = np.int32(97)
    out["embeds_sample"] = flat[:, ::97].numpy()
    out["embeds_token_norm"] = embeds.norm(dim=-1).numpy()
    hs = vis.hidden_states
    out["hidden_norms"] = np.stack([h.norm(dim=-1).mean(dim=-1).numpy() for h in hs], 0)   # [L+1, B]
    if full:
        out["embeds_full"] = embeds.numpy()
        out["greedy_logits_full"] = logits.numpy()
    out["meta"] = np.array(json.dumps(dict(seed=seed, eos_boost=eos_boost, batch=batch, max_length=max_length, beams=beams,
                                           arch=arch.__dict__, transformers="5.15.0",
                                           torch=torch.__version__)))
    print("  greedy min margin %.3e  median %.3e ; beam scores %s" % (
        out["greedy_margin"].min(), np.median(out["greedy_margin"]), np.round(out["beam_scores"], 3)))
    print("  greedy lens:", [(r != 0).sum() for r in out["greedy_sequences"]], "beam lens:", [(r != 0).sum() for r in out["beam_sequences"]])
    return out


def run_greedy_only(arch: BlipArch, seed: int, batch: int, max_length: int, eos_boost: float):
    """Wider greedy fixture (same weights and frames as blip_base, more rows): ids + per-step top-2 margins only."""
    sd = procedural_blip_state_dict(arch, seed, eos_boost=eos_boost)
    model = build_hf(arch, sd)
    pixels = synthetic_pixels(batch, arch.image_size, seed=seed)
    seqs, margins = [], []
    with torch.no_grad():
        for i in range(0, batch, 16):
            g = model.generate(pixel_values=pixels[i:i + 16], max_length=max_length, num_beams=1, do_sample=False,
                               output_logits=True, return_dict_in_generate=True)
            seq = torch.full((g.sequences.shape[0], max_length), arch.pad, dtype=torch.long)
            seq[:, : g.sequences.shape[1]] = g.sequences
            top = torch.topk(torch.stack(list(g.logits), 0), k=2, dim=-1).values          # [T, b, 2]
            m = torch.full((max_length - 1, seq.shape[0]), 1e9)
            m[: top.shape[0]] = top[..., 0] - top[..., 1]
            seqs.append(seq); margins.append(m)
    return {"greedy_sequences": torch.cat(seqs).numpy().astype(np.int32), "greedy_margin": torch.cat(margins, 1).numpy(),
            "meta": np.array(json.dumps(dict(seed=seed, eos_boost=eos_boost, batch=batch, max_length=max_length, beams=1,
                                             arch=arch.__dict__, transformers="5.15.0", torch=torch.__version__)))}


def run_beam_only(arch: BlipArch, seed: int, batch: int, max_length: int, beams: int, eos_boost: float):
    """Config 3's batch (64 frames, beam 3) through the real HF beam search: sequences + sequences_scores only."""
    sd = procedural_blip_state_dict(arch, seed, eos_boost=eos_boost)
    model = build_hf(arch, sd)
    pixels = synthetic_pixels(batch, arch.image_size, seed=seed)
    seqs, scores = [], []
    with torch.no_grad():
        for i in range(0, batch, 16):
            b = model.generate(pixel_values=pixels[i:i + 16], max_length=max_length, num_beams=beams, do_sample=False,
                               length_penalty=1.0, early_stopping=False, output_scores=True, return_dict_in_generate=True)
            seq = torch.full((b.sequences.shape[0], max_length), arch.pad, dtype=torch.long)
            seq[:, : b.sequences.shape[1]] = b.sequences
            seqs.append(seq); scores.append(b.sequences_scores)
    return {"beam_sequences": torch.cat(seqs).numpy().astype(np.int32), "beam_scores": torch.cat(scores).numpy(),
            "meta": np.array(json.dumps(dict(seed=seed, eos_boost=eos_boost, batch=batch, max_length=max_length, beams=beams,
                                             arch=arch.__dict__, transformers="5.15.0", torch=torch.__version__)))}


def main():
    torch.manual_seed(0)
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    if "--beam64-only" in sys.argv:
        print("blip_base64_beam3")
        np.savez_compressed(os.path.join(gold, "blip_base64_beam3.npz"), **run_beam_only(BlipArch(), seed=0, batch=64,
                                                                                           max_length=20, beams=3, eos_boost=9.0))
        return
    if "--wide256-only" in sys.argv:
        # 256 frames = one whole headline batch through the real HF greedy loop (rows 0..63 are blip_base64's)
        print("blip_base256")
        np.savez_compressed(os.path.join(gold, "blip_base256.npz"), **run_greedy_only(BlipArch(), seed=0, batch=256,
                                                                                         max_length=20, eos_boost=9.0))
        return
    if "--wide-only" in sys.argv:
        print("blip_base64")
        np.savez_compressed(os.path.join(gold, "blip_base64.npz"), **run_greedy_only(BlipArch(), seed=0, batch=64,
                                                                                        max_length=20, eos_boost=9.0))
        return
    print("blip_tiny")
    np.savez_compressed(os.path.join(gold, "blip_tiny.npz"), **run(BlipArch.tiny(), seed=3, batch=4,
                                                                     max_length=12, beams=3, full=True, eos_boost=2.0))
    print("blip_tiny_eos")
    np.savez_compressed(os.path.join(gold, "blip_tiny_eos.npz"), **run(BlipArch.tiny(), seed=3, batch=4,
                                                                         max_length=12, beams=3, full=True, eos_boost=2.5))
    print("blip_base")
    np.savez_compressed(os.path.join(gold, "blip_base.npz"), **run(BlipArch(), seed=0, batch=8,
                                                                     max_length=20, beams=3, full=False, eos_boost=9.0))
    write_perplexity_kats(gold)
    print("blip_base64")
    np.savez_compressed(os.path.join(gold, "blip_base64.npz"), **run_greedy_only(BlipArch(), seed=0, batch=64,
                                                                                    max_length=20, eos_boost=9.0))
    print("done")


def write_perplexity_kats(gold):
    """The three known-answer tests of the reference (captioning_predictor.py:66-98): inputs AND targets are the literals
    there; expected = what the reference compares against, torcheval's Perplexity(input, target) = exp(mean cross-entropy of
    the target tokens) - computed here from that definition (F.cross_entropy on the literal targets, float64), NOT from the
    max-softmax formula under test.  The definition itself is anchored on the value torcheval's documentation publishes for
    its first example (targets [[2], [1]] -> 2.7593), kept as a fourth entry."""
    cases = [([[[0.3659, 0.7025, 0.3104]], [[0.0097, 0.6577, 0.1947]], [[0.5659, 0.0025, 0.0104]], [[0.9097, 0.0577, 0.7947]]],
              [[1], [1], [0], [0]]),
             ([[[0.5659, 0.0025, 0.0104]], [[0.9097, 0.0577, 0.7947]]], [[0], [0]]),
             ([[[0.3659, 0.7025, 0.3104]], [[0.0097, 0.6577, 0.1947]]], [[1], [1]])]
    kats = []
    for inp, tgt in cases:
        x = torch.tensor(inp, dtype=torch.float64)                      # [n, 1, V]
        t = torch.tensor(tgt)
        ce = torch.nn.functional.cross_entropy(x.reshape(-1, x.shape[-1]), t.reshape(-1), reduction="mean")
        kats.append({"input": inp, "target": tgt, "expected": float(torch.exp(ce)), "target_is_argmax": True})
    doc_in, doc_t = [[[0.3659, 0.7025, 0.3104]], [[0.0097, 0.6577, 0.1947]]], [[2], [1]]
    x = torch.tensor(doc_in, dtype=torch.float64)
    ce = torch.nn.functional.cross_entropy(x.reshape(-1, 3), torch.tensor(doc_t).reshape(-1), reduction="mean")
    assert abs(float(torch.exp(ce)) - 2.7593) < 1e-4
    kats.append({"input": doc_in, "target": doc_t, "expected": float(torch.exp(ce)), "published": 2.7593,
                 "target_is_argmax": False})
    with open(os.path.join(gold, "perplexity_kat.json"), "w") as f:
        json.dump(kats, f, indent=1)


if __name__ == "__main__":
    if "--only-perplexity" in sys.argv:
        write_perplexity_kats(os.path.join(ROOT, "tests", "golden"))
    else:
        main()
