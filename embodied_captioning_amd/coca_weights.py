"""Host-side, one-off preparation of an open_clip CoCa state dict for libcaptioner_hip.so.

The library takes the checkpoint's own tensors by their open_clip names plus a few `derived.*` tensors computed here
once at load time (plain fp32 torch on the host; nothing here runs per caption):

  derived.pool_q            ln_q(query) @ Wq^T + bq                 the pooler's queries are parameters, so their
                                                                    projection is a constant [n_queries, embed_dim]
  derived.pool_kv.{weight,bias}   [Wk; Wv] of the pooler fused to one [2E, 1024] projection
  derived.cross_q.{i}.{weight,bias}   query rows of cross-attention block i's packed in_proj
  derived.cross_kv.{weight,bias}  k|v rows of every cross-attention block with that block's ln_1_kv folded in:
                                  W' = W diag(gamma), b' = b + W beta  - all layers then share one affine-free
                                  LayerNorm of the image tokens and ONE GEMM produces every layer's cross K/V
  derived.vocab.weight      text_decoder.text_projection^T  ([vocab, width], the GEMM's K-contiguous layout)
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .config import CocaArch


def derive_coca_tensors(sd: Dict[str, torch.Tensor], a: CocaArch) -> Dict[str, torch.Tensor]:
    E = a.embed_dim
    out: Dict[str, torch.Tensor] = {}
    p = "visual.attn_pool."
    bias = sd[p + "attn.in_proj_bias"].float()
    q = F.layer_norm(sd[p + "query"].float(), (E,), sd[p + "ln_q.weight"].float(), sd[p + "ln_q.bias"].float(), a.eps)
    out["derived.pool_q"] = F.linear(q, sd[p + "attn.q_proj_weight"].float(), bias[:E]).contiguous()
    out["derived.pool_kv.weight"] = torch.cat([sd[p + "attn.k_proj_weight"].float(), sd[p + "attn.v_proj_weight"].float()], 0)
    out["derived.pool_kv.bias"] = bias[E:].contiguous()
    ws, bs = [], []
    for i in range(a.mm_layers):
        c = f"text_decoder.cross_attn.{i}."
        w, b = sd[c + "attn.in_proj_weight"].float(), sd[c + "attn.in_proj_bias"].float()
        out[f"derived.cross_q.{i}.weight"] = w[:E].contiguous()
        out[f"derived.cross_q.{i}.bias"] = b[:E].contiguous()
        g, beta = sd[c + "ln_1_kv.weight"].float(), sd[c + "ln_1_kv.bias"].float()
        wkv = w[E:]                                                    # [2E, E]: k rows then v rows
        ws.append(wkv * g[None, :])
        bs.append(b[E:] + wkv @ beta)
    out["derived.cross_kv.weight"] = torch.cat(ws, 0).contiguous()
    out["derived.cross_kv.bias"] = torch.cat(bs, 0).contiguous()
    out["derived.vocab.weight"] = sd["text_decoder.text_projection"].float().t().contiguous()
    return out


def resize_visual_pos_embed(pos: torch.Tensor, a: CocaArch) -> torch.Tensor:
    """Position table of a checkpoint trained at another resolution -> this arch's token count, the way open_clip's
    `resize_pos_embed` does when `force_image_size` differs from the pretrained size (reference factory.py:243-245,
    334-337): the class row is kept, the g0 x g0 grid rows are interpolated to g x g (bicubic, antialias,
    align_corners=False)."""
    n, d = pos.shape
    if n == a.n_tokens:
        return pos
    g0 = int(round((n - 1) ** 0.5))
    if g0 * g0 + 1 != n:
        raise ValueError(f"visual.positional_embedding has {n} rows: not a class row + square grid")
    grid = pos[1:].float().reshape(1, g0, g0, d).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(a.grid, a.grid), mode="bicubic", antialias=True, align_corners=False)
    grid = grid.permute(0, 2, 3, 1).reshape(a.grid * a.grid, d)
    return torch.cat([pos[:1].float(), grid], 0).contiguous()


def coca_library_state_dict(sd: Dict[str, torch.Tensor], a: CocaArch) -> Dict[str, torch.Tensor]:
    """Everything `CaptionerEngine.load_state_dict` should stream in: the checkpoint + the derived tensors."""
    full = dict(sd)
    full["visual.positional_embedding"] = resize_visual_pos_embed(sd["visual.positional_embedding"], a)
    full.update(derive_coca_tensors(sd, a))
    return full
