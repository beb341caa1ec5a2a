"""CPU restatement of the CoCa captioning path.  TEST INFRASTRUCTURE - NOT PRODUCT CODE.  **PARITY UNPINNED.**

The reference's CoCa (``experimenting_env/captioner/models/coca/coca_model.py``) builds its towers from `open_clip`
(``open_clip.transformer.{VisionTransformer, TextTransformer, MultimodalTransformer}``, unpinned in
``requirements.txt:15``), which is not installed in the build container and has no wheel offline; the reference's own
`generate` additionally asserts on transformers >= 5 (``coca_model.py:20-46,227``: `BeamSearchScorer` is gone).  So no
golden vector can be produced here.  This file restates, from the published open_clip (v2.2x) algorithm:

  * VisionTransformer(attentional_pool=True, output_tokens=True): conv patch-embed (no bias) + cls + abs-pos, ln_pre,
    pre-LN blocks (nn.MultiheadAttention in_proj, exact GELU MLP), AttentionalPooler (256 learned queries, 8 heads,
    ln_q / ln_k, k/v from the 1024-wide tokens), ln_post; pooled = token 0, image_embs = tokens 1..255
  * TextTransformer(embed_cls=True): token + abs-pos embeddings, causal pre-LN blocks; with a cls embedding only the
    pooled vector goes through ln_final - the per-token outputs that feed the decoder do not
  * MultimodalTransformer: per layer a causal self-attention block then a cross-attention block
    (ln_1 on the text, ln_1_kv on the image tokens), ln_final, @ text_projection [width, vocab]

and the decode loop that IS in the reference tree (``coca_model.py:205-333``, generation_type='top_k', top_k=1 as
``captioner/models/coca/coca.py:29`` calls it): MinLength(min_seq_len, eos), forced EOS at cur_len + 1 == seq_len,
rows whose last token is EOS/pad emit pad, per-step logits of the active rows.

What the tests can establish for CoCa is therefore self-consistency (KV-cached step == full-prefix recompute, the
reference's way), HIP-vs-this-restatement parity, and identity of the building blocks (`_mha`, `_block`) with the torch.nn
modules open_clip composes (tests/test_coca_cpu.py) - not identity of the whole composition with open_clip's code.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _ln(x, sd, p, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _mha(q_in, k_in, v_in, sd, p, heads, causal=False, wq=None, wk=None, wv=None):
    """nn.MultiheadAttention forward (batch_first).  Packed in_proj unless separate q/k/v weights are given."""
    E = q_in.shape[-1]
    b = sd[p + ".in_proj_bias"]
    if wq is None:
        w = sd[p + ".in_proj_weight"]
        wq, wk, wv = w[:E], w[E:2 * E], w[2 * E:]
    q = F.linear(q_in, wq, b[:E]); k = F.linear(k_in, wk, b[E:2 * E]); v = F.linear(v_in, wv, b[2 * E:])
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    hd = E // heads
    q = q.view(B, Tq, heads, hd).transpose(1, 2); k = k.view(B, Tk, heads, hd).transpose(1, 2)
    v = v.view(B, Tk, heads, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
    if causal:
        s = s + torch.full((Tq, Tk), float("-inf")).triu(1 + Tk - Tq)
    o = torch.matmul(torch.softmax(s, dim=-1), v).transpose(1, 2).reshape(B, Tq, E)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _block(x, sd, p, heads, eps, causal=False, kv: Optional[Tensor] = None):
    """ResidualAttentionBlock (pre-LN).  kv given -> cross-attention block with ln_1_kv."""
    h = _ln(x, sd, p + ".ln_1", eps)
    if kv is None:
        x = x + _mha(h, h, h, sd, p + ".attn", heads, causal=causal)
    else:
        c = _ln(kv, sd, p + ".ln_1_kv", eps)
        x = x + _mha(h, c, c, sd, p + ".attn", heads)
    h = _ln(x, sd, p + ".ln_2", eps)
    h = F.gelu(F.linear(h, sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"]))
    return x + F.linear(h, sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])


@torch.no_grad()
def encode_image(sd: Dict[str, Tensor], a, pixels: Tensor):
    """-> (pooled [B, embed_dim] un-normalised, image_embs [B, n_queries-1, embed_dim])."""
    v = "visual."
    x = F.conv2d(pixels, sd[v + "conv1.weight"], None, stride=a.patch_size).flatten(2).transpose(1, 2)
    cls = sd[v + "class_embedding"].view(1, 1, -1).expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[v + "positional_embedding"]
    x = _ln(x, sd, v + "ln_pre", a.eps)
    for i in range(a.v_layers):
        x = _block(x, sd, f"{v}transformer.resblocks.{i}", a.v_heads, a.eps)
    # attentional pooler
    p = v + "attn_pool"
    ctx = _ln(x, sd, p + ".ln_k", a.eps)
    q = _ln(sd[p + ".query"], sd, p + ".ln_q", a.eps).unsqueeze(0).expand(x.shape[0], -1, -1)
    x = _mha(q, ctx, ctx, sd, p + ".attn", a.pool_heads, wq=sd[p + ".attn.q_proj_weight"],
             wk=sd[p + ".attn.k_proj_weight"], wv=sd[p + ".attn.v_proj_weight"])
    x = _ln(x, sd, v + "ln_post", a.eps)
    pooled, tokens = x[:, 0], x[:, 1:]
    return pooled @ sd[v + "proj"], tokens


@torch.no_grad()
def text_tokens_full(sd, a, text: Tensor) -> Tensor:
    """TextTransformer(embed_cls=True) per-token outputs [B, T, W] (no ln_final on this branch).  The appended cls token
    sits after the text and is causally invisible to it, so it is simply not materialised here."""
    T = text.shape[1]
    x = sd["text.token_embedding.weight"][text] + sd["text.positional_embedding"][:T]
    for i in range(a.t_layers):
        x = _block(x, sd, f"text.transformer.resblocks.{i}", a.t_heads, a.eps, causal=True)
    return x


@torch.no_grad()
def decoder_logits_full(sd, a, image_embs: Tensor, token_embs: Tensor) -> Tensor:
    x = token_embs
    for i in range(a.mm_layers):
        x = _block(x, sd, f"text_decoder.resblocks.{i}", a.t_heads, a.eps, causal=True)
        x = _block(x, sd, f"text_decoder.cross_attn.{i}", a.t_heads, a.eps, kv=image_embs)
    x = _ln(x, sd, "text_decoder.ln_final", a.eps)
    return x @ sd["text_decoder.text_projection"]


@torch.no_grad()
def last_logits_full(sd, a, image_embs, text):
    """What the reference computes every step: whole prefix through both towers, last position's logits."""
    return decoder_logits_full(sd, a, image_embs, text_tokens_full(sd, a, text))[:, -1]


# ---------------------------------------------------------------------------------------------- KV-cached step
class CocaState:
    def __init__(self, n):
        self.k: List[Optional[Tensor]] = [None] * n
        self.v: List[Optional[Tensor]] = [None] * n
        self.length = 0


def _cached_block(x, sd, p, heads, eps, st: CocaState, idx: int):
    """Causal self-attention block on ONE new position with a growing K/V cache."""
    E = x.shape[-1]; hd = E // heads
    h = _ln(x, sd, p + ".ln_1", eps)
    w, b = sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"]
    qkv = F.linear(h, w, b)
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    B = x.shape[0]
    q = q.view(B, 1, heads, hd).transpose(1, 2); k = k.view(B, 1, heads, hd).transpose(1, 2)
    v = v.view(B, 1, heads, hd).transpose(1, 2)
    st.k[idx] = k if st.k[idx] is None else torch.cat([st.k[idx], k], dim=2)
    st.v[idx] = v if st.v[idx] is None else torch.cat([st.v[idx], v], dim=2)
    s = torch.matmul(q, st.k[idx].transpose(-1, -2)) / math.sqrt(hd)
    o = torch.matmul(torch.softmax(s, -1), st.v[idx]).transpose(1, 2).reshape(B, 1, E)
    x = x + F.linear(o, sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"])
    h = _ln(x, sd, p + ".ln_2", eps)
    h = F.gelu(F.linear(h, sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"]))
    return x + F.linear(h, sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])


@torch.no_grad()
def step(sd, a, tokens: Tensor, image_embs: Tensor, st: CocaState) -> Tensor:
    """tokens int64 [B] = newest token; -> logits [B, V] of the next position."""
    t = st.length
    x = (sd["text.token_embedding.weight"][tokens] + sd["text.positional_embedding"][t]).unsqueeze(1)
    for i in range(a.t_layers):
        x = _cached_block(x, sd, f"text.transformer.resblocks.{i}", a.t_heads, a.eps, st, i)
    for i in range(a.mm_layers):
        x = _cached_block(x, sd, f"text_decoder.resblocks.{i}", a.t_heads, a.eps, st, a.t_layers + i)
        x = _block(x, sd, f"text_decoder.cross_attn.{i}", a.t_heads, a.eps, kv=image_embs)
    st.length += 1
    x = _ln(x[:, 0], sd, "text_decoder.ln_final", a.eps)
    return x @ sd["text_decoder.text_projection"]


@torch.no_grad()
def generate_top1(sd, a, pixels: Tensor, seq_len: Optional[int] = None, min_seq_len: Optional[int] = None,
                  image_embs: Optional[Tensor] = None, use_cache: bool = True):
    """Reference loop coca_model.py:278-327 with generation_type='top_k', top_k=1, temperature 1 (argmax; ties aside).
    Returns {"text": int64 [B, L], "logits": list of [n_active, V] (MinLength-processed), "image_embs"}."""
    seq_len = seq_len or a.seq_len
    min_seq_len = a.min_seq_len if min_seq_len is None else min_seq_len
    if image_embs is None:
        _, image_embs = encode_image(sd, a, pixels)
    B = image_embs.shape[0]
    text = torch.full((B, 1), a.sot, dtype=torch.int64)
    st = CocaState(a.t_layers + a.mm_layers)
    out_logits = []
    while True:
        cur_len = text.shape[1]
        logits = step(sd, a, text[:, -1], image_embs, st) if use_cache else last_logits_full(sd, a, image_embs, text)
        mask = (text[:, -1] == a.eos) | (text[:, -1] == a.pad)
        sample = torch.full((B, 1), a.pad, dtype=torch.int64)
        if mask.all():
            break
        lg = logits[~mask].clone()
        if cur_len < min_seq_len:
            lg[:, a.eos] = float("-inf")                     # MinLengthLogitsProcessor
        out_logits.append(lg)
        if cur_len + 1 == seq_len:
            sample[~mask, 0] = a.eos
        else:
            sample[~mask, 0] = lg.argmax(dim=-1)
        text = torch.cat([text, sample], dim=-1)
        if text.shape[1] >= seq_len:                          # MaxLengthCriteria(seq_len)
            break
    return {"text": text, "logits": out_logits, "image_embs": image_embs}


# ------------------------------------------------------------------------------------------------------------------
# Beam search: the reference's `_generate_beamsearch` (coca_model.py:335-482) with HF's legacy `BeamSearchScorer`
# (coca_model.py:20-46 imports it, :353-358 builds it with the scorer's defaults: length_penalty 1.0,
# do_early_stopping False, num_beam_hyps_to_keep 1).  **PARITY UNPINNED**, doubly: `BeamSearchScorer` left transformers in
# 5.x, so the scorer below is restated from the published 4.4x source (transformers/generation/beam_search.py:
# `BeamSearchScorer.process` / `.finalize`, `BeamHypotheses.add` / `.is_done`), not imported.
# Things that are specific to the reference's loop and easy to get wrong:
#   * the scores are RAW logits (+ MinLength's -inf on EOS), not log-probabilities: coca_model.py:418-425 adds
#     `beam_scores` to `next_token_logits` after the processors, there is no log_softmax;
#   * `decoder_prompt_len` is never passed, so the start-of-text token counts in every length-penalty denominator;
#   * no forced EOS at seq_len (that is the sampling loop's, :317-318): MaxLengthCriteria(seq_len) just ends the loop and
#     `finalize` adds the open beams;
#   * num_beam_groups must divide num_beams (:370); the reference's defaults 6 / 3 do, config 5's beam=5 runs with one group.
class _BeamHyps:
    def __init__(self, num_beams, length_penalty=1.0):
        self.k, self.lp, self.beams, self.worst = num_beams, length_penalty, [], 1e9

    def add(self, hyp, sum_logprobs, generated_len):
        score = sum_logprobs / (generated_len ** self.lp)
        if len(self.beams) < self.k or score > self.worst:
            self.beams.append((score, hyp))
            if len(self.beams) > self.k:
                order = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[order[0][1]]
                self.worst = order[1][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs, cur_len):             # early_stopping False: the heuristic
        if len(self.beams) < self.k:
            return False
        return self.worst >= best_sum_logprobs / cur_len ** self.lp


@torch.no_grad()
def generate_beamsearch(sd, a, pixels: Tensor, num_beams: int = 5, seq_len: Optional[int] = None,
                        min_seq_len: Optional[int] = None, image_embs: Optional[Tensor] = None):
    """coca_model.py:335-482 with num_beam_groups = 1.  Returns {"sequences": int64 [B, <= seq_len] (EOS appended where it
    fits, pad after), "scores": the winning hypotheses' length-normalised scores, "image_embs"}."""
    seq_len = seq_len or a.seq_len
    min_seq_len = a.min_seq_len if min_seq_len is None else min_seq_len
    assert seq_len > min_seq_len
    if image_embs is None:
        _, image_embs = encode_image(sd, a, pixels)
    B, K = image_embs.shape[0], num_beams
    embs = torch.repeat_interleave(image_embs, K, dim=0)                      # :350 (the reference re-encodes K copies)
    input_ids = torch.full((B * K, 1), a.sot, dtype=torch.int64)
    hyps = [_BeamHyps(K) for _ in range(B)]
    done = [False] * B
    beam_scores = torch.full((B, K), -1e9, dtype=torch.float32)
    beam_scores[:, 0] = 0                                                     # :383 with num_sub_beams == num_beams
    beam_scores = beam_scores.view(B * K)
    while True:
        logits = last_logits_full(sd, a, embs, input_ids)                     # whole prefix, as the reference does (:394-401)
        V = logits.shape[-1]
        if input_ids.shape[1] < min_seq_len:
            logits = logits.clone()
            logits[:, a.eos] = float("-inf")                                  # MinLengthLogitsProcessor
        scores = (logits + beam_scores[:, None]).view(B, K * V)               # :418-425
        top_v, top_i = torch.topk(scores, 2 * K, dim=1, largest=True, sorted=True)
        next_idx, next_tok = top_i // V, top_i % V
        # ---- BeamSearchScorer.process
        cur_len = input_ids.shape[-1] + 1
        nb_scores = torch.zeros(B, K); nb_tok = torch.full((B, K), a.pad, dtype=torch.int64); nb_idx = torch.zeros(B, K, dtype=torch.int64)
        for b in range(B):
            if done[b]:
                continue                                                      # zeros / pad / index 0
            slot = 0
            for rank in range(2 * K):
                tok, sc, bi = int(next_tok[b, rank]), float(top_v[b, rank]), b * K + int(next_idx[b, rank])
                if tok == a.eos:
                    if rank >= K:
                        continue
                    hyps[b].add(input_ids[bi].clone(), sc, generated_len=cur_len)
                else:
                    nb_scores[b, slot], nb_tok[b, slot], nb_idx[b, slot] = sc, tok, bi
                    slot += 1
                if slot == K:
                    break
            done[b] = done[b] or hyps[b].is_done(float(top_v[b].max()), cur_len)
        beam_scores = nb_scores.view(B * K)
        beam_idx = nb_idx.view(B * K)
        input_ids = torch.cat([input_ids[beam_idx], nb_tok.view(B * K, 1)], dim=-1)          # :455-457, :468
        if all(done) or input_ids.shape[1] >= seq_len:                        # :472
            break
    # ---- BeamSearchScorer.finalize(max_length = seq_len)
    for b in range(B):
        if done[b]:
            continue
        for k in range(K):
            row = b * K + k
            hyps[b].add(input_ids[row], float(beam_scores[row]), generated_len=input_ids.shape[-1])
    best, best_scores = [], []
    for b in range(B):
        s, h = sorted(hyps[b].beams, key=lambda x: x[0])[-1]
        best.append(h); best_scores.append(s)
    lens = [len(h) for h in best]
    width = min(max(lens) + 1, seq_len)
    out = torch.full((B, width), a.pad, dtype=torch.int64)
    for b, h in enumerate(best):
        out[b, : lens[b]] = h
        if lens[b] < width:
            out[b, lens[b]] = a.eos
    return {"sequences": out, "scores": torch.tensor(best_scores), "image_embs": image_embs}


@torch.no_grad()
def generate_beamsearch_groups(sd, a, pixels: Tensor, num_beams: int = 6, num_beam_groups: int = 3, seq_len: Optional[int] = None,
                               min_seq_len: Optional[int] = None, image_embs: Optional[Tensor] = None):
    """coca_model.py:335-482 LITERALLY, with its beam groups (the `generate()` defaults are num_beams = 6, num_beam_groups = 3,
    :218-219): per step ONE forward over all B * num_beams rows (:394-401), then group by group (:403-466) the rows of that
    group (:409-415), MinLength on their logits, + the group's running scores, top 2 * group_size over group_size * V (:429-432),
    `BeamSearchScorer.process(..., group_index=g)` (a `BeamHypotheses` per (image, group), :440-449), and the group's rows of
    `input_ids` reordered in place (:455-457).  `finalize` (:472-481) adds the open beams of every (image, group) that is not
    done and returns the best hypothesis over ALL groups of an image (the legacy scorer's `num_beam_hyps_to_keep = 1`).
    No diversity processor is attached (:236-241), so nothing couples the groups.  **PARITY UNPINNED** as everything CoCa here.
    Returns {"sequences", "scores", "group_sequences": per (image, group) best hypothesis - for the test of the equivalence
    with ONE search of num_beams / num_beam_groups beams that the product relies on}."""
    seq_len = seq_len or a.seq_len
    min_seq_len = a.min_seq_len if min_seq_len is None else min_seq_len
    assert seq_len > min_seq_len and num_beams % num_beam_groups == 0
    if image_embs is None:
        _, image_embs = encode_image(sd, a, pixels)
    B, K, G = image_embs.shape[0], num_beams, num_beam_groups
    g = K // G                                                                 # num_sub_beams
    embs = torch.repeat_interleave(image_embs, K, dim=0)
    input_ids = torch.full((B * K, 1), a.sot, dtype=torch.int64)
    hyps = [_BeamHyps(g) for _ in range(B * G)]                                # index = batch_idx * G + group (scorer's batch_group_idx)
    done = [False] * (B * G)
    beam_scores = torch.full((B, K), -1e9, dtype=torch.float32)
    beam_scores[:, ::g] = 0                                                    # :383
    beam_scores = beam_scores.view(B * K)
    while True:
        current_tokens = torch.zeros(B * K, dtype=torch.int64)
        logits_all = last_logits_full(sd, a, embs, input_ids)
        V = logits_all.shape[-1]
        cur_len = input_ids.shape[-1] + 1
        for gi in range(G):
            rows = [b * K + i for b in range(B) for i in range(gi * g, (gi + 1) * g)]          # batch_group_indices
            rows_t = torch.tensor(rows)
            group_ids = input_ids[rows_t]
            logits = logits_all[rows_t].clone()
            if group_ids.shape[1] < min_seq_len:
                logits[:, a.eos] = float("-inf")
            scores = (logits + beam_scores[rows_t][:, None]).view(B, g * V)
            top_v, top_i = torch.topk(scores, 2 * g, dim=1, largest=True, sorted=True)
            next_idx, next_tok = top_i // V, top_i % V
            nb_scores = torch.zeros(B, g); nb_tok = torch.full((B, g), a.pad, dtype=torch.int64); nb_idx = torch.zeros(B, g, dtype=torch.int64)
            for b in range(B):
                bg = b * G + gi
                if done[bg]:
                    continue
                slot = 0
                for rank in range(2 * g):
                    tok, sc, bi = int(next_tok[b, rank]), float(top_v[b, rank]), b * g + int(next_idx[b, rank])   # index into the GROUP's rows
                    if tok == a.eos:
                        if rank >= g:
                            continue
                        hyps[bg].add(group_ids[bi].clone(), sc, generated_len=cur_len)
                    else:
                        nb_scores[b, slot], nb_tok[b, slot], nb_idx[b, slot] = sc, tok, bi
                        slot += 1
                    if slot == g:
                        break
                done[bg] = done[bg] or hyps[bg].is_done(float(top_v[b].max()), cur_len)
            beam_scores[rows_t] = nb_scores.view(B * g)
            beam_idx = nb_idx.view(B * g)
            input_ids[rows_t] = group_ids[beam_idx]                            # :455
            current_tokens[rows_t] = nb_tok.view(B * g)
        input_ids = torch.cat([input_ids, current_tokens.unsqueeze(-1)], dim=-1)
        if all(done) or input_ids.shape[1] >= seq_len:
            break
    for bg in range(B * G):                                                    # finalize: open beams of unfinished (image, group)s
        if done[bg]:
            continue
        for i in range(g):
            row = bg * g + i                                                   # = batch_idx * K + group * g + i
            hyps[bg].add(input_ids[row], float(beam_scores[row]), generated_len=input_ids.shape[-1])
    best, best_scores, per_group = [], [], []
    for b in range(B):
        cands = [bm for gi in range(G) for bm in hyps[b * G + gi].beams]
        s, h = sorted(cands, key=lambda x: x[0])[-1]
        best.append(h); best_scores.append(s)
        per_group.append([sorted(hyps[b * G + gi].beams, key=lambda x: x[0])[-1] for gi in range(G)])
    lens = [len(h) for h in best]
    width = min(max(lens) + 1, seq_len)
    out = torch.full((B, width), a.pad, dtype=torch.int64)
    for b, h in enumerate(best):
        out[b, : lens[b]] = h
        if lens[b] < width:
            out[b, lens[b]] = a.eos
    return {"sequences": out, "scores": torch.tensor(best_scores), "group_best": per_group, "image_embs": image_embs}
