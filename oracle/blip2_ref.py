"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke, bench.py's cpu_baseline): CPU restatement of the BLIP-2 OPT
captioner behind the reference's `captioner/models/blip2/blip2.py:24-29`
(`Blip2ForConditionalGeneration.generate(pixel_values, output_logits=True, return_dict_in_generate=True)`).

The arithmetic lives in transformers (not in /root/reference; installed here: 5.15.0).  Restated, fp32 throughout:
  HF:models/blip_2/modeling_blip_2.py   Blip2VisionModel (same block structure as BLIP's ViT, eps 1e-6) - reused from
                                        oracle/blip_ref.py; Blip2QFormerModel (layernorm on the query tokens, per layer:
                                        self-attention among the queries, cross-attention to the image tokens on layers
                                        i % cross_attention_frequency == 0, query FFN; post-LN, eps 1e-12);
                                        language_projection; `generate`: inputs_embeds = [projected queries ; embed(bos)],
                                        input_ids = [image_token] * n_queries + [bos].
  HF:models/opt/modeling_opt.py         OPTDecoder with do_layer_norm_before (pre-LN), ReLU FFN, learned positions with
                                        offset 2, final_layer_norm, tied lm_head without bias.
  HF:generation/utils.py                greedy `_sample`; with no length argument HF uses max_length = prompt + 20.
Pinned by tests/golden/blip2_tiny.npz (tools/make_goldens_blip2.py runs the real HF model on seeded weights).

`load_in_8bit` (reference blip2.py:19-22; bitsandbytes 0.4x `Linear8bitLt`, NOT in /root/reference and not installed here:
PARITY UNPINNED for this part): `quantize_int8_rowwise` restates the published vector-wise quantisation of a weight
(`int8_vectorwise_quant`: CB = round(W * 127 / absmax(row)), SCB = absmax(row); dequantised W' = CB * SCB / 127) and
`int8_state_dict` applies it to the modules transformers converts (`replace_with_bnb_linear`: every nn.Linear outside
`_keep_in_fp32_modules = ["query_tokens", "qformer"]` and lm_head).  LLM.int8's activation half (per-token int8 rows, columns
with a value beyond 6.0 in fp16) is not restated: activations stay floating point, as in the product.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

from . import blip_ref


def quantize_int8_rowwise(w: torch.Tensor):
    """-> (q int8 [out, in], scale fp32 [out]) with w ~ q * scale; fp32 arithmetic, round-half-even."""
    w = w.detach().float()
    amax = w.abs().amax(dim=1, keepdim=True)
    inv = torch.where(amax > 0, torch.full_like(amax, 127.0) / amax, torch.zeros_like(amax))   # (a true division: `127.0 / t` is t.reciprocal() * 127 in torch)
    return torch.round(w * inv).to(torch.int8), (amax / 127.0).squeeze(1)


def int8_linear_names(sd) -> List[str]:
    lm = "language_model.model.decoder.layers."
    out = []
    for k, v in sd.items():
        if not k.endswith(".weight") or v.dim() != 2:
            continue
        if k.startswith(lm) and any(t in k for t in ("q_proj", "k_proj", "v_proj", "out_proj", "fc1", "fc2")):
            out.append(k)
        elif k.startswith("vision_model.encoder.layers.") and (".self_attn." in k or ".mlp." in k):
            out.append(k)
        elif k == "language_projection.weight":
            out.append(k)
    return out


def int8_state_dict(sd) -> Dict[str, torch.Tensor]:
    """The state dict a `load_in_8bit` model computes with: converted Linears replaced by their dequantised int8 weights."""
    sd = dict(sd)
    for k in int8_linear_names(sd):
        q, sc = quantize_int8_rowwise(sd[k])
        sd[k] = q.float() * sc[:, None]
    return sd


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"].float(), sd[name + ".bias"].float() if name + ".bias" in sd else None)


def _mha(q, k, v, heads, mask=None):
    B, Lq, D = q.shape
    hd = D // heads
    q = q.view(B, Lq, heads, hd).transpose(1, 2)
    k = k.view(B, -1, heads, hd).transpose(1, 2)
    v = v.view(B, -1, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if mask is not None:
        s = s + mask
    return (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, Lq, D)


def qformer(sd: Dict[str, torch.Tensor], a, image_embeds: torch.Tensor) -> torch.Tensor:
    """image_embeds [B, N, v_hidden] -> query outputs [B, num_query_tokens, q_hidden]."""
    B = image_embeds.shape[0]
    Q = a.q_hidden
    x = sd["query_tokens"].float().expand(B, -1, -1)
    x = F.layer_norm(x, (Q,), sd["qformer.layernorm.weight"].float(), sd["qformer.layernorm.bias"].float(), a.q_eps)
    for i in range(a.q_layers):
        p = f"qformer.encoder.layer.{i}."
        blocks = [("attention", x)] + ([("crossattention", image_embeds)] if i % a.q_cross_freq == 0 else [])
        for blk, kv_src in blocks:
            src = x if blk == "attention" else kv_src
            ctx = _mha(_lin(sd, p + blk + ".attention.query", x), _lin(sd, p + blk + ".attention.key", src),
                       _lin(sd, p + blk + ".attention.value", src), a.q_heads)
            x = F.layer_norm(_lin(sd, p + blk + ".output.dense", ctx) + x, (Q,), sd[p + blk + ".output.LayerNorm.weight"].float(),
                             sd[p + blk + ".output.LayerNorm.bias"].float(), a.q_eps)
        h = F.gelu(_lin(sd, p + "intermediate_query.dense", x))
        x = F.layer_norm(_lin(sd, p + "output_query.dense", h) + x, (Q,), sd[p + "output_query.LayerNorm.weight"].float(),
                         sd[p + "output_query.LayerNorm.bias"].float(), a.q_eps)
    return x


def language_inputs(sd, a, pixels: torch.Tensor):
    """-> (image_embeds, query_output, projected queries [B, n_q, t_hidden])."""
    emb = blip_ref.encode_image(sd, a, pixels)
    qo = qformer(sd, a, emb)
    return emb, qo, _lin(sd, "language_projection", qo)


class OptState:
    def __init__(self, n_layers):
        self.k: List = [None] * n_layers
        self.v: List = [None] * n_layers
        self.length = 0


def opt_forward(sd, a, x: torch.Tensor, st: OptState) -> torch.Tensor:
    """x [B, L, T] = embeddings of L new positions (token or projected-query embeddings, WITHOUT positions) appended to the
    cached prefix; returns the logits of the last position [B, vocab]."""
    lm = "language_model.model.decoder."
    B, L, T = x.shape
    pos = torch.arange(st.length, st.length + L) + 2                                  # OPTLearnedPositionalEmbedding offset
    x = x + sd[lm + "embed_positions.weight"].float()[pos]
    past = st.length
    causal = torch.full((L, past + L), 0.0)
    causal.masked_fill_(torch.arange(past + L)[None, :] > (past + torch.arange(L))[:, None], torch.finfo(torch.float32).min)
    for i in range(a.t_layers):
        p = f"{lm}layers.{i}."
        h = F.layer_norm(x, (T,), sd[p + "self_attn_layer_norm.weight"].float(), sd[p + "self_attn_layer_norm.bias"].float(), a.t_eps)
        q, k, v = _lin(sd, p + "self_attn.q_proj", h), _lin(sd, p + "self_attn.k_proj", h), _lin(sd, p + "self_attn.v_proj", h)
        st.k[i] = k if st.k[i] is None else torch.cat([st.k[i], k], 1)
        st.v[i] = v if st.v[i] is None else torch.cat([st.v[i], v], 1)
        x = x + _lin(sd, p + "self_attn.out_proj", _mha(q, st.k[i], st.v[i], a.t_heads, causal))
        h = F.layer_norm(x, (T,), sd[p + "final_layer_norm.weight"].float(), sd[p + "final_layer_norm.bias"].float(), a.t_eps)
        x = x + _lin(sd, p + "fc2", F.relu(_lin(sd, p + "fc1", h)))
    st.length += L
    x = F.layer_norm(x[:, -1], (T,), sd[lm + "final_layer_norm.weight"].float(), sd[lm + "final_layer_norm.bias"].float(), a.t_eps)
    return F.linear(x, sd["language_model.lm_head.weight"].float())


@torch.no_grad()
def greedy_generate(sd, a, pixels: torch.Tensor, max_new_tokens: int | None = None):
    """-> {"sequences": [B, n_q + 1 + n_new] (image tokens, bos, generated; pad after eos), "logits": list of [B, vocab],
    "query_output", "image_embeds"} - HF `generate` with output_logits, greedy."""
    n_new = max_new_tokens or a.max_new_tokens
    emb, qo, proj = language_inputs(sd, a, pixels)
    B = pixels.shape[0]
    tok = sd["language_model.model.decoder.embed_tokens.weight"].float()
    st = OptState(a.t_layers)
    logits = opt_forward(sd, a, torch.cat([proj, tok[torch.full((B, 1), a.bos)]], 1), st)
    seq = torch.cat([torch.full((B, a.num_query_tokens), a.image_token), torch.full((B, 1), a.bos)], 1)
    unfinished = torch.ones(B, dtype=torch.bool)
    steps = []
    for t in range(n_new):
        steps.append(logits)
        nxt = torch.where(unfinished, logits.argmax(-1), torch.full((B,), a.pad))
        seq = torch.cat([seq, nxt[:, None]], 1)
        unfinished = unfinished & (nxt != a.eos)
        if not unfinished.any() or t + 1 == n_new:
            break
        logits = opt_forward(sd, a, tok[nxt][:, None], st)
    return {"sequences": seq, "logits": steps, "query_output": qo, "image_embeds": emb, "language_inputs": proj}
