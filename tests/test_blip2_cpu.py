"""CPU: the BLIP-2 OPT restatement (oracle/blip2_ref.py) against vectors captured from the real HF
Blip2ForConditionalGeneration (tools/make_goldens_blip2.py)."""
import json
import os

import numpy as np
import torch

from embodied_captioning_amd.config import Blip2Arch
from embodied_captioning_amd.weights import procedural_blip2_state_dict, synthetic_pixels
from oracle import blip2_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_blip2(name="blip2_tiny"):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    arch = Blip2Arch(**meta["arch"])
    sd = procedural_blip2_state_dict(arch, meta["seed"], eos_boost=meta["eos_boost"])
    px = synthetic_pixels(meta["batch"], arch.image_size, seed=meta["seed"])
    return g, meta, arch, sd, px


def test_restatement_matches_hf_golden():
    g, meta, a, sd, px = load_blip2()
    out = R.greedy_generate(sd, a, px)
    assert np.abs(out["image_embeds"].numpy() - g["image_embeds"]).max() < 2e-5
    assert np.abs(out["query_output"].numpy() - g["query_output"]).max() < 2e-5
    seq = out["sequences"].numpy()
    assert np.array_equal(seq, g["sequences"][:, : seq.shape[1]]) and (g["sequences"][:, seq.shape[1]:] == a.pad).all()
    lg = torch.stack(out["logits"], 0).numpy()
    # rows that finished feed pad tokens in HF and are not stepped here: compare the steps every row was still running
    ref = g["logits"][: lg.shape[0]]
    new = g["sequences"][:, a.num_query_tokens + 1:]
    for b in range(meta["batch"]):
        n = int(np.argmax(new[b] == a.eos)) + 1 if (new[b] == a.eos).any() else lg.shape[0]
        assert np.abs(lg[:n, b] - ref[:n, b]).max() < 5e-5
    # the fixture exercises both endings
    assert (new == a.eos).any() and not (new == a.eos).any(axis=1).all()


def test_restatement_matches_hf_golden_at_production_width():
    """The production GEOMETRY of `Salesforce/blip2-opt-2.7b` (reference captioner/models/blip2/blip2.py:19-28): ViT-g/14 1408 wide
    / 16 heads of 88 / 257 tokens, Q-Former 768 / 32 queries, OPT 2560 wide / 32 heads of 80 / FFN 10240, the real 50272-token
    vocabulary - two layers per tower.  The golden is HF Blip2ForConditionalGeneration itself on these weights
    (tools/make_goldens_blip2.py --width); it pins the restatement at the widths the tiny golden cannot reach."""
    g, meta, a, sd, px = load_blip2("blip2_width")
    assert (a.v_hidden, a.v_heads, a.q_hidden, a.t_hidden, a.t_heads, a.t_ffn, a.vocab, a.n_tokens) == (1408, 16, 768, 2560, 32, 10240, 50272, 257)
    torch.set_num_threads(min(8, torch.get_num_threads()))
    out = R.greedy_generate(sd, a, px)
    emb = out["image_embeds"]
    assert np.abs(emb[:, :, :16].numpy() - g["image_embeds_head"]).max() < 1e-4
    assert np.abs(emb.norm(dim=-1).numpy() - g["image_embeds_norm"]).max() < 1e-3 * float(g["image_embeds_norm"].max())
    assert np.abs(out["query_output"].numpy() - g["query_output"]).max() < 5e-5
    seq = out["sequences"].numpy()
    assert np.array_equal(seq, g["sequences"][:, : seq.shape[1]]) and (g["sequences"][:, seq.shape[1]:] == a.pad).all()
    lg = torch.stack(out["logits"], 0)                                     # [steps, B, V]
    assert lg.shape[0] == g["top8_ids"].shape[0]
    got = torch.gather(lg, 2, torch.from_numpy(g["top8_ids"]).long()).numpy()
    assert np.abs(got - g["top8_values"]).max() < 1e-4
    assert np.array_equal(torch.topk(lg, 8, dim=-1).indices.numpy(), g["top8_ids"])


def test_int8_quantiser_host_form_is_the_restatement_and_names_match():
    """`load_in_8bit`: the product's host quantiser (weights.py: tensors stored dequantised) and name policy against the restatement's
    (oracle/blip2_ref.py; PARITY UNPINNED against bitsandbytes, absent here)."""
    from embodied_captioning_amd.weights import blip2_int8_host_names, int8_roundtrip, quantize_int8_rowwise
    g, meta, a, sd, px = load_blip2()
    w = sd["vision_model.encoder.layers.0.mlp.fc1.weight"].clone()
    w[2] = 0.0
    q, s = quantize_int8_rowwise(w)
    q2, s2 = R.quantize_int8_rowwise(w)
    assert torch.equal(q, q2) and torch.equal(s, s2) and int(q.abs().max()) == 127 and (q[2] == 0).all() and s[2] == 0
    assert (int8_roundtrip(w) - w).abs().max() <= s.max() * 0.5 * (1 + 1e-6)
    host = set(blip2_int8_host_names(sd))
    lm = "language_model.model.decoder.layers."
    assert host | {k for k in R.int8_linear_names(sd) if k.startswith(lm)} == set(R.int8_linear_names(sd))
    assert not any(k.startswith(("qformer.", "language_model.lm_head", lm)) for k in host)
    assert len(host) == 4 * a.v_layers + 1


def test_int8_quantiser_properties():
    """Size-independent properties of the row-wise int8 quantiser (product's host form = the restatement): |q| <= 127 with the row's
    largest element at +-127, |w - q s| <= s / 2, zero rows -> zeros, and quantising the dequantised weights gives the same bytes back."""
    from hypothesis import given, settings, strategies as st
    from embodied_captioning_amd.weights import quantize_int8_rowwise

    @settings(max_examples=40, deadline=None)
    @given(st.integers(1, 24), st.integers(1, 96), st.integers(0, 2 ** 31 - 1), st.sampled_from([1e-6, 1e-3, 0.03, 1.0, 40.0]))
    def prop(rows, cols, seed, scale):
        g = torch.Generator().manual_seed(seed)
        w = torch.randn(rows, cols, generator=g) * scale
        if rows > 2:
            w[1] = 0.0
        q, s = quantize_int8_rowwise(w)
        q2, s2 = R.quantize_int8_rowwise(w)
        assert torch.equal(q, q2) and torch.equal(s, s2)
        assert int(q.abs().max()) <= 127
        nz = w.abs().amax(1) > 0
        assert (q[nz].abs().amax(1) == 127).all() and (q[~nz] == 0).all() and (s[~nz] == 0).all()
        assert ((w - q.float() * s[:, None]).abs() <= s[:, None] * 0.5 * (1 + 1e-5) + 1e-30).all()
        q3, _ = quantize_int8_rowwise(q.float() * s[:, None])
        assert torch.equal(q3, q)
    prop()
