"""CPU oracle for the captioner forward path.  TEST INFRASTRUCTURE - NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product package (``embodied_captioning_amd``) never does and fails loudly when its HIP
library is missing.

What it is: a plain PyTorch-CPU fp32 restatement (written for this repo; no ``transformers`` import) of
the arithmetic the reference's captioner wrappers delegate to a third-party dependency that is NOT in
``/root/reference``: HuggingFace ``transformers`` (unpinned by the reference - ``requirements.txt`` does not
list it; reference call sites ``experimenting_env/captioner/models/blip2/blip2.py:8,19-28``,
``captioner/models/coca/coca_model.py:21-32``).  The version restated is transformers==5.15.0
(``HF:`` below = site-packages/transformers/):

  * vision tower      HF:models/blip/modeling_blip.py:231-243 (embeddings), :309-344 (attention),
                      :356-360 (MLP), :373-392 (pre-LN block), :473-498 (post_layernorm)
  * text decoder      HF:models/blip/modeling_blip_text.py:62-89 (embeddings), :130-198 (self/cross attention
                      with KV cache), :209-213/:262-269 (post-LN outputs), :287-312 (layer), :388-419 (LM head)
  * generate          HF:models/blip/modeling_blip.py:858-932 (prompt = [bos], eos = sep_token_id)
  * greedy            HF:generation/utils.py:2876-2941
  * beam search       HF:generation/utils.py:3010-3204 (helpers), :3316-3523 (loop)
  * perplexity        reference experimenting_env/captioner/captioning_predictor.py:34-47

Pinning: the reference itself holds no golden vector for this path except three perplexity KATs
(captioning_predictor.py:66-98; carried in tests/golden/perplexity_kat.json).  The restatement is pinned
against outputs of the real HF implementation run in the build container: tools/make_goldens.py imports
transformers 5.15.0, loads the same procedural weights, and commits image_embeds / per-step logits /
greedy ids / beam ids+scores under tests/golden/; tests/test_oracle_golden.py checks this file against
them on CPU.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# vision tower
# ----------------------------------------------------------------------------------------------

def vision_embeddings(sd: Dict[str, Tensor], arch, pixels: Tensor) -> Tensor:
    """Conv2d(k=s=patch) + cls + abs-pos.  HF:modeling_blip.py:231-243."""
    p = "vision_model.embeddings."
    x = F.conv2d(pixels, sd[p + "patch_embedding.weight"], sd[p + "patch_embedding.bias"], stride=arch.patch_size)
    x = x.flatten(2).transpose(1, 2)                                  # [B, P, D]
    cls = sd[p + "class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1)
    return x + sd[p + "position_embedding"][:, : x.shape[1], :]


def vision_layer(sd: Dict[str, Tensor], arch, i: int, x: Tensor) -> Tensor:
    """Pre-LN block.  HF:modeling_blip.py:373-392, attention :309-344, MLP :356-360."""
    p = f"vision_model.encoder.layers.{i}."
    B, N, D = x.shape
    H, hd = arch.v_heads, arch.v_hidden // arch.v_heads
    h = F.layer_norm(x, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], arch.v_eps)
    qkv = F.linear(h, sd[p + "self_attn.qkv.weight"], sd[p + "self_attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
    a = torch.softmax(s, dim=-1)
    ctx = torch.matmul(a, v).permute(0, 2, 1, 3).reshape(B, N, D)
    x = x + F.linear(ctx, sd[p + "self_attn.projection.weight"], sd[p + "self_attn.projection.bias"])
    h = F.layer_norm(x, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], arch.v_eps)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def encode_image(sd: Dict[str, Tensor], arch, pixels: Tensor, return_hidden: bool = False):
    """pixels fp32 [B,3,H,W] (already normalised) -> image_embeds [B, 1+P, D].  HF:modeling_blip.py:473-498."""
    x = vision_embeddings(sd, arch, pixels)
    hidden = [x]
    for i in range(arch.v_layers):
        x = vision_layer(sd, arch, i, x)
        hidden.append(x)
    out = F.layer_norm(x, (arch.v_hidden,), sd["vision_model.post_layernorm.weight"],
                       sd["vision_model.post_layernorm.bias"], arch.v_eps)
    return (out, hidden) if return_hidden else out


# ----------------------------------------------------------------------------------------------
# text decoder with KV cache
# ----------------------------------------------------------------------------------------------

class DecoderState:
    """Per-row self-attention K/V (grown one position per step) and per-row cross-attention K/V."""

    def __init__(self, n_layers: int):
        self.self_k: List[Optional[Tensor]] = [None] * n_layers
        self.self_v: List[Optional[Tensor]] = [None] * n_layers
        self.cross_k: List[Optional[Tensor]] = [None] * n_layers
        self.cross_v: List[Optional[Tensor]] = [None] * n_layers
        self.length = 0

    def reorder(self, idx: Tensor) -> None:
        """Beam reorder of the self-attention cache (HF:generation/utils.py:3479-3485)."""
        for i in range(len(self.self_k)):
            self.self_k[i] = self.self_k[i].index_select(0, idx)
            self.self_v[i] = self.self_v[i].index_select(0, idx)


def _heads(x: Tensor, H: int) -> Tensor:
    R, T, D = x.shape
    return x.view(R, T, H, D // H).transpose(1, 2)


def cross_kv(sd: Dict[str, Tensor], arch, image_embeds: Tensor, state: DecoderState, repeat: int = 1) -> None:
    """K,V = Linear(image_embeds), once per image per layer (HF:modeling_blip_text.py:161-175).
    `repeat` expands rows per beam the way HF expands encoder_hidden_states."""
    enc = image_embeds.repeat_interleave(repeat, dim=0) if repeat > 1 else image_embeds
    for i in range(arch.t_layers):
        p = f"text_decoder.bert.encoder.layer.{i}.crossattention.self."
        state.cross_k[i] = _heads(F.linear(enc, sd[p + "key.weight"], sd[p + "key.bias"]), arch.t_heads)
        state.cross_v[i] = _heads(F.linear(enc, sd[p + "value.weight"], sd[p + "value.bias"]), arch.t_heads)


def _attend(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    # q [R,H,1,hd]; single new query so the causal mask is all-visible over the cached prefix
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    a = torch.softmax(s, dim=-1)
    ctx = torch.matmul(a, v)                                            # [R,H,1,hd]
    return ctx.permute(0, 2, 1, 3).reshape(q.shape[0], 1, -1)


def decoder_step(sd: Dict[str, Tensor], arch, tokens: Tensor, state: DecoderState) -> Tensor:
    """One cached decode step.  tokens int64 [R] (the newest token of every row) -> logits fp32 [R, V]."""
    tb = "text_decoder.bert."
    T, H = arch.t_hidden, arch.t_heads
    pos = state.length
    x = sd[tb + "embeddings.word_embeddings.weight"][tokens] + sd[tb + "embeddings.position_embeddings.weight"][pos]
    x = F.layer_norm(x, (T,), sd[tb + "embeddings.LayerNorm.weight"], sd[tb + "embeddings.LayerNorm.bias"], arch.t_eps)
    x = x.unsqueeze(1)                                                 # [R,1,T]
    for i in range(arch.t_layers):
        p = f"{tb}encoder.layer.{i}."
        a = p + "attention."
        q = _heads(F.linear(x, sd[a + "self.query.weight"], sd[a + "self.query.bias"]), H)
        k = _heads(F.linear(x, sd[a + "self.key.weight"], sd[a + "self.key.bias"]), H)
        v = _heads(F.linear(x, sd[a + "self.value.weight"], sd[a + "self.value.bias"]), H)
        if state.self_k[i] is None:
            state.self_k[i], state.self_v[i] = k, v
        else:
            state.self_k[i] = torch.cat([state.self_k[i], k], dim=2)
            state.self_v[i] = torch.cat([state.self_v[i], v], dim=2)
        ctx = _attend(q, state.self_k[i], state.self_v[i])
        x = F.layer_norm(F.linear(ctx, sd[a + "output.dense.weight"], sd[a + "output.dense.bias"]) + x, (T,),
                         sd[a + "output.LayerNorm.weight"], sd[a + "output.LayerNorm.bias"], arch.t_eps)
        c = p + "crossattention."
        q = _heads(F.linear(x, sd[c + "self.query.weight"], sd[c + "self.query.bias"]), H)
        ctx = _attend(q, state.cross_k[i], state.cross_v[i])
        x = F.layer_norm(F.linear(ctx, sd[c + "output.dense.weight"], sd[c + "output.dense.bias"]) + x, (T,),
                         sd[c + "output.LayerNorm.weight"], sd[c + "output.LayerNorm.bias"], arch.t_eps)
        h = F.gelu(F.linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        x = F.layer_norm(F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"]) + x, (T,),
                         sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], arch.t_eps)
    state.length += 1
    cp = "text_decoder.cls.predictions."
    h = F.gelu(F.linear(x[:, 0], sd[cp + "transform.dense.weight"], sd[cp + "transform.dense.bias"]))
    h = F.layer_norm(h, (T,), sd[cp + "transform.LayerNorm.weight"], sd[cp + "transform.LayerNorm.bias"], arch.t_eps)
    # decoder weight is tied to the word embeddings, decoder bias to cls.predictions.bias
    return F.linear(h, sd[tb + "embeddings.word_embeddings.weight"], sd[cp + "bias"])


# ----------------------------------------------------------------------------------------------
# generation
# ----------------------------------------------------------------------------------------------

@torch.no_grad()
def greedy_generate(sd, arch, pixels: Tensor, max_length: int = 20, image_embeds: Optional[Tensor] = None):
    """HF greedy (`_sample`, do_sample=False): argmax, pad after EOS, stop when every row is finished or
    max_length is reached.  Returns dict(sequences int64 [B, <=max_length] incl. BOS, logits list of [B,V],
    image_embeds)."""
    if image_embeds is None:
        image_embeds = encode_image(sd, arch, pixels)
    B = image_embeds.shape[0]
    state = DecoderState(arch.t_layers)
    cross_kv(sd, arch, image_embeds, state)
    seq = torch.full((B, 1), arch.bos, dtype=torch.int64)
    unfinished = torch.ones(B, dtype=torch.int64)
    step_logits: List[Tensor] = []
    while True:
        logits = decoder_step(sd, arch, seq[:, -1], state)
        step_logits.append(logits)
        nxt = torch.argmax(logits, dim=-1)
        nxt = nxt * unfinished + arch.pad * (1 - unfinished)
        seq = torch.cat([seq, nxt[:, None]], dim=-1)
        unfinished = unfinished & (nxt != arch.eos).long()
        if seq.shape[1] >= max_length:
            unfinished = torch.zeros_like(unfinished)
        if unfinished.max() == 0:
            break
    return {"sequences": seq, "logits": step_logits, "image_embeds": image_embeds}


def _gather_beams(t: Tensor, idx: Tensor) -> Tensor:
    while idx.dim() < t.dim():
        idx = idx.unsqueeze(-1)
    return torch.gather(t, 1, idx.expand(-1, -1, *t.shape[2:]))


@torch.no_grad()
def beam_search_generate(sd, arch, pixels: Tensor, num_beams: int = 3, max_length: int = 20,
                         length_penalty: float = 1.0, early_stopping: bool = False,
                         image_embeds: Optional[Tensor] = None):
    """HF v5 beam search (`_beam_search`): 2*num_beams candidates per item, -1e9 masking, finished pool with
    length penalty `(cur_len+1-prompt_len)**lp`, early-stop heuristic on `cur_len-prompt_len`.
    Returns dict(sequences [B, L], sequences_scores [B])."""
    if image_embeds is None:
        image_embeds = encode_image(sd, arch, pixels)
    B, K, V, P = image_embeds.shape[0], num_beams, arch.vocab, 1
    state = DecoderState(arch.t_layers)
    cross_kv(sd, arch, image_embeds, state, repeat=K)
    keep = 2 * K
    # HF: `output_fill_value = pad_token_id or eos_token_id[0]` (utils.py:3324) - with BLIP's pad id 0 the
    # `or` falls through, so beam outputs are padded with the EOS id (102), not 0.
    fill = arch.pad or arch.eos
    running_seq = torch.full((B, K, max_length), fill, dtype=torch.int64)
    running_seq[:, :, 0] = arch.bos
    sequences = running_seq.clone()
    running_scores = torch.zeros((B, K), dtype=torch.float32)
    running_scores[:, 1:] = -1e9
    beam_scores = torch.full((B, K), -1e9, dtype=torch.float32)
    finished = torch.zeros((B, K), dtype=torch.bool)
    heuristic_open = torch.ones((B, 1), dtype=torch.bool)
    top_mask = torch.cat([torch.ones(K, dtype=torch.bool), torch.zeros(keep - K, dtype=torch.bool)])
    gen_len = torch.zeros((B, K), dtype=torch.int64)          # generated length of each finished hypothesis
    run_len_dummy = None
    cur_len = 1
    while True:
        logits = decoder_step(sd, arch, running_seq[:, :, cur_len - 1].reshape(-1), state)
        logp = F.log_softmax(logits.float(), dim=-1).view(B, K, V) + running_scores[:, :, None]
        topk_lp, topk_idx = torch.topk(logp.view(B, K * V), k=keep)
        src_beam = topk_idx // V
        tok = topk_idx % V
        cand_seq = _gather_beams(running_seq, src_beam)
        cand_seq[:, :, cur_len] = tok
        hits = (tok == arch.eos) | (cur_len + 1 >= max_length)
        # running beams for the next iteration
        run_lp = topk_lp + hits.float() * -1.0e9
        nxt = torch.topk(run_lp, k=K)[1]
        running_seq = _gather_beams(cand_seq, nxt)
        running_scores = _gather_beams(run_lp, nxt)
        beam_idx = _gather_beams(src_beam, nxt)
        # finished pool
        just_done = hits & top_mask[None, :]
        fin_lp = topk_lp / ((cur_len + 1 - P) ** length_penalty)
        full = torch.all(finished, dim=-1, keepdim=True) & (early_stopping is True)
        fin_lp = fin_lp + full.float() * -1.0e9
        fin_lp = fin_lp + (~heuristic_open).float() * -1.0e9
        fin_lp = fin_lp + (~just_done).float() * -1.0e9
        m_seq = torch.cat([sequences, cand_seq], dim=1)
        m_sc = torch.cat([beam_scores, fin_lp], dim=1)
        m_fin = torch.cat([finished, just_done], dim=1)
        m_len = torch.cat([gen_len, torch.full((B, keep), cur_len + 1 - P, dtype=torch.int64)], dim=1)
        sel = torch.topk(m_sc, k=K)[1]
        sequences = _gather_beams(m_seq, sel)
        beam_scores = _gather_beams(m_sc, sel)
        finished = _gather_beams(m_fin, sel)
        gen_len = _gather_beams(m_len, sel)
        # cache reorder
        state.reorder((beam_idx + torch.arange(B)[:, None] * K).reshape(-1))
        cur_len += 1
        best_running = running_scores[:, :1] / ((cur_len - P) ** length_penalty)
        worst_fin = torch.where(finished, torch.min(beam_scores, dim=1, keepdim=True)[0],
                                torch.tensor(-1.0e9))
        heuristic_open = heuristic_open & torch.any(best_running > worst_fin, dim=-1, keepdim=True)
        go = bool(torch.any(heuristic_open)) and not (bool(torch.all(finished)) and early_stopping is True) \
            and not bool(torch.all(hits))
        if not go:
            break
    out_len = P + int(gen_len[:, 0].max())
    return {"sequences": sequences[:, 0, :out_len], "sequences_scores": beam_scores[:, 0],
            "image_embeds": image_embeds}


@torch.no_grad()
def score_sequences(sd, arch, image_embeds: Tensor, sequences: Tensor, lengths: Tensor, length_penalty: float = 1.0):
    """Teacher-forced HF beam score of given hypotheses: sum of log-probs of the generated tokens (after BOS, up to
    and including EOS or the last token) divided by n_generated ** length_penalty.  Lets a reduced-precision beam
    search be judged on the sequence it actually returned, even when a near-tie made it leave the oracle's path."""
    B, L = sequences.shape
    state = DecoderState(arch.t_layers)
    cross_kv(sd, arch, image_embeds, state)
    total = torch.zeros(B, dtype=torch.float64)
    for t in range(int(lengths.max()) - 1):
        logits = decoder_step(sd, arch, sequences[:, t].long(), state)
        lp = F.log_softmax(logits.float(), dim=-1)
        nxt = sequences[:, t + 1].long().clamp(min=0)
        live = (t + 1) < lengths
        total += torch.where(live, lp.gather(1, nxt[:, None])[:, 0].double(), torch.zeros(B, dtype=torch.float64))
    return (total / (lengths - 1).double() ** length_penalty).float()


# ----------------------------------------------------------------------------------------------
# perplexity (reference experimenting_env/captioner/captioning_predictor.py:34-47)
# ----------------------------------------------------------------------------------------------

def compute_perplexity(logits) -> Tensor:
    """logits: tensor [n, T, V] or a list of T tensors [n, V].  exp(-sum(log max softmax) / T) as float64."""
    if not torch.is_tensor(logits):
        logits = torch.stack([l.float() for l in logits], dim=1)
    probs = torch.softmax(logits, dim=-1).max(dim=-1).values
    return torch.exp(-probs.log().sum() / probs.shape[1]).double()
