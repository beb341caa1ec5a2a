"""CPU: CoCa oracle restatement (parity UNPINNED - open_clip is not available): self-consistency, and its building blocks
against the torch.nn modules open_clip composes."""
import pytest
import torch

from embodied_captioning_amd.config import CocaArch
from embodied_captioning_amd.coca_weights import derive_coca_tensors
from embodied_captioning_amd.weights import procedural_coca_state_dict, synthetic_pixels
from oracle import coca_ref as R


def test_kv_cached_step_equals_full_prefix_recompute():
    """The reference re-runs both towers on the whole prefix every step (coca_model.py:294-303); the cached step must
    give the same tokens and the same processed logits."""
    a = CocaArch.tiny()
    for boost in (0.0, 4.0):
        sd = procedural_coca_state_dict(a, 1, eos_boost=boost)
        px = synthetic_pixels(4, a.image_size, seed=1)
        g = R.generate_top1(sd, a, px)
        f = R.generate_top1(sd, a, px, use_cache=False)
        assert torch.equal(g["text"], f["text"])
        for x, y in zip(g["logits"], f["logits"]):
            fin = torch.isfinite(x)
            assert torch.equal(fin, torch.isfinite(y)) and (x[fin] - y[fin]).abs().max() < 1e-4


def test_decode_loop_semantics():
    a = CocaArch.tiny()
    sd = procedural_coca_state_dict(a, 1, eos_boost=4.0)
    px = synthetic_pixels(4, a.image_size, seed=1)
    g = R.generate_top1(sd, a, px)
    text = g["text"]
    assert (text[:, 0] == a.sot).all() and text.shape[1] <= a.seq_len
    for row in text.tolist():
        if a.eos in row:
            i = row.index(a.eos)
            assert i >= a.min_seq_len                      # MinLength: EOS cannot be emitted before min_seq_len tokens
            assert all(t == a.pad for t in row[i + 1:])    # rows whose last token is EOS/pad emit pad
    full = R.generate_top1(procedural_coca_state_dict(a, 1), a, px)["text"]
    assert full.shape[1] == a.seq_len and (full[:, -1] == a.eos).all()   # forced EOS at cur_len + 1 == seq_len
    assert torch.isinf(g["logits"][0][:, a.eos]).all() and torch.isfinite(g["logits"][a.min_seq_len][:, a.eos]).all()


def test_folded_cross_kv_equals_ln_then_projection():
    import torch.nn.functional as F
    a = CocaArch.tiny()
    sd = procedural_coca_state_dict(a, 2)
    d = derive_coca_tensors(sd, a)
    E = a.embed_dim
    x = torch.randn(7, E)
    xh = F.layer_norm(x, (E,), None, None, a.eps)
    for i in range(a.mm_layers):
        c = f"text_decoder.cross_attn.{i}."
        ref = F.linear(F.layer_norm(x, (E,), sd[c + "ln_1_kv.weight"], sd[c + "ln_1_kv.bias"], a.eps),
                       sd[c + "attn.in_proj_weight"][E:], sd[c + "attn.in_proj_bias"][E:])
        got = F.linear(xh, d["derived.cross_kv.weight"][2 * E * i:2 * E * (i + 1)], d["derived.cross_kv.bias"][2 * E * i:2 * E * (i + 1)])
        assert (ref - got).abs().max() < 1e-5
    assert d["derived.vocab.weight"].shape == (a.vocab, E) and d["derived.pool_q"].shape == (a.pool_queries, E)


def test_pos_embed_resize_matches_open_clip_recipe():
    """Class row kept, grid rows bicubic-antialias interpolated (open_clip resize_pos_embed); identity at equal size."""
    import dataclasses
    import torch.nn.functional as F
    from embodied_captioning_amd.coca_weights import resize_visual_pos_embed
    from embodied_captioning_amd.config import CocaArch
    a = CocaArch.tiny()                                     # 28 px / patch 14 -> 2x2 grid
    b = dataclasses.replace(a, image_size=56)               # 4x4 grid
    pos = torch.randn(a.n_tokens, a.v_hidden, generator=torch.Generator().manual_seed(0))
    assert resize_visual_pos_embed(pos, a) is pos
    out = resize_visual_pos_embed(pos, b)
    assert out.shape == (17, a.v_hidden) and torch.equal(out[0], pos[0])
    want = F.interpolate(pos[1:].reshape(1, 2, 2, -1).permute(0, 3, 1, 2), size=(4, 4), mode="bicubic", antialias=True,
                         align_corners=False).permute(0, 2, 3, 1).reshape(16, -1)
    assert torch.allclose(out[1:], want)
    with pytest.raises(ValueError):
        resize_visual_pos_embed(torch.zeros(7, a.v_hidden), b)


def _mha_state(mod, prefix):
    return {prefix + "." + k: v.detach() for k, v in mod.state_dict().items()}


def test_attention_and_block_equal_the_torch_modules_open_clip_composes():
    """open_clip is not installed here, so CoCa's towers cannot be run - but they are compositions of torch.nn modules
    (`nn.MultiheadAttention`, `nn.LayerNorm`, Linear-GELU-Linear) whose real implementations ARE here.  This pins the
    restatement's attention (packed in_proj, causal mask; separate q/k/v projections with kdim/vdim != embed_dim as the
    attentional pooler uses) and its pre-LN residual block (self and cross form) to those modules."""
    import torch.nn as nn
    torch.manual_seed(0)
    E, H, B, T, S, Ck = 48, 4, 3, 7, 11, 80
    x, ctx = torch.randn(B, T, E), torch.randn(B, S, Ck)
    # 1. packed self-attention with a causal mask
    m = nn.MultiheadAttention(E, H, batch_first=True).eval()
    nn.init.normal_(m.in_proj_bias, std=0.2); nn.init.normal_(m.out_proj.bias, std=0.2)
    mask = torch.full((T, T), float("-inf")).triu(1)
    want = m(x, x, x, need_weights=False, attn_mask=mask)[0]
    got = R._mha(x, x, x, _mha_state(m, "a"), "a", H, causal=True)
    assert (got - want).abs().max().item() < 2e-6
    # 2. separate projections, keys/values from a wider context (AttentionalPooler: kdim = vdim = context_dim)
    p = nn.MultiheadAttention(E, H, kdim=Ck, vdim=Ck, batch_first=True).eval()
    nn.init.normal_(p.in_proj_bias, std=0.2)
    q = torch.randn(B, 5, E)
    want = p(q, ctx, ctx, need_weights=False)[0]
    sd = _mha_state(p, "a")
    got = R._mha(q, ctx, ctx, sd, "a", H, wq=sd["a.q_proj_weight"], wk=sd["a.k_proj_weight"], wv=sd["a.v_proj_weight"])
    assert (got - want).abs().max().item() < 2e-6

    # 3. the residual block as open_clip writes it: x + attn(ln_1(x)[, ln_1_kv(kv)]) ; x + mlp(ln_2(x))
    class Block(nn.Module):
        def __init__(self, cross):
            super().__init__()
            self.ln_1, self.ln_2 = nn.LayerNorm(E), nn.LayerNorm(E)
            self.attn = nn.MultiheadAttention(E, H, batch_first=True)
            self.mlp = nn.Sequential()
            self.mlp.add_module("c_fc", nn.Linear(E, 4 * E)); self.mlp.add_module("gelu", nn.GELU())
            self.mlp.add_module("c_proj", nn.Linear(4 * E, E))
            if cross:
                self.ln_1_kv = nn.LayerNorm(E)

        def forward(self, x, kv=None, mask=None):
            k = self.ln_1_kv(kv) if kv is not None else None
            h = self.ln_1(x)
            x = x + self.attn(h, k if k is not None else h, k if k is not None else h, need_weights=False, attn_mask=mask)[0]
            return x + self.mlp(self.ln_2(x))

    for cross in (False, True):
        blk = Block(cross).eval()
        for prm in blk.parameters():
            if prm.dim() == 1:
                nn.init.normal_(prm, std=0.3)
        bsd = {"b." + k: v.detach() for k, v in blk.state_dict().items()}
        kv = torch.randn(B, S, E) if cross else None
        with torch.no_grad():
            want = blk(x, kv, None if cross else mask)
        got = R._block(x, bsd, "b", H, 1e-5, causal=not cross, kv=kv)
        assert (got - want).abs().max().item() < 5e-6


def test_beam_search_restatement_unpinned_properties():
    """oracle/coca_ref.generate_beamsearch (coca_model.py:335-482 + HF's legacy BeamSearchScorer, restated): with one beam it
    walks the arg-max path of the top-k(1) loop; with more beams the returned hypothesis scores at least as well under the
    scorer's own measure (sum of raw logits / length incl. start token and EOS); MinLength holds; rows are EOS-terminated
    and padded."""
    from embodied_captioning_amd.config import CocaArch
    from embodied_captioning_amd.weights import procedural_coca_state_dict, synthetic_pixels
    from oracle import coca_ref as R
    a = CocaArch.tiny()
    sd = procedural_coca_state_dict(a, 3, eos_boost=2.0)
    px = synthetic_pixels(4, a.image_size, seed=3)
    g = R.generate_top1(sd, a, px)
    b1 = R.generate_beamsearch(sd, a, px, num_beams=1, image_embs=g["image_embs"])
    n = min(g["text"].shape[1], b1["sequences"].shape[1])
    assert torch.equal(g["text"][:, :n], b1["sequences"][:, :n])
    b5 = R.generate_beamsearch(sd, a, px, num_beams=5, image_embs=g["image_embs"])
    assert (b5["scores"] >= b1["scores"] - 1e-5).all()
    for row in b5["sequences"].tolist():
        assert row[0] == a.sot
        body = [t for t in row if t != a.pad]
        assert len(body) >= a.min_seq_len and (body[-1] == a.eos or len(body) == a.seq_len)
        assert a.eos not in body[:-1]


@pytest.mark.parametrize("seed,boost,K,G", [(3, 2.0, 6, 3), (3, 0.0, 4, 2), (5, 4.0, 3, 3), (5, 1.0, 6, 1)])
def test_beam_groups_literal_loop_unpinned_equals_one_search_of_a_group(seed, boost, K, G):
    """The reference's `_generate_beamsearch` with beam GROUPS (coca_model.py:403-466; generate() defaults 6 beams / 3 groups),
    restated literally, against ONE search of K / G beams: without a diversity processor (:236-241) the groups never see each
    other, start from the same scores and rank the same candidates, so every group ends with the same best hypothesis and
    `finalize` returns it - the equivalence cap_generate_groups relies on.  (Both sides are restatements: unpinned.)"""
    from embodied_captioning_amd.config import CocaArch
    from embodied_captioning_amd.weights import procedural_coca_state_dict, synthetic_pixels
    from oracle import coca_ref as R
    a = CocaArch.tiny()
    sd = procedural_coca_state_dict(a, seed, eos_boost=boost)
    px = synthetic_pixels(3, a.image_size, seed=seed)
    _, embs = R.encode_image(sd, a, px)
    lit = R.generate_beamsearch_groups(sd, a, px, num_beams=K, num_beam_groups=G, image_embs=embs)
    one = R.generate_beamsearch(sd, a, px, num_beams=K // G, image_embs=embs)
    assert torch.equal(lit["sequences"], one["sequences"])
    assert (lit["scores"] - one["scores"]).abs().max().item() < 1e-6
    for per_image in lit["group_best"]:
        for score, hyp in per_image[1:]:
            assert torch.equal(hyp, per_image[0][1]) and abs(score - per_image[0][0]) < 1e-6
