"""Weight sources for the captioner: procedural (seeded) state dicts and checkpoint readers.

The state-dict key set is exactly what ``BlipForConditionalGeneration.state_dict()`` has
(SURVEY.md §8c lists all 473 tensors) - that key set is the HuggingFace-checkpoint loading
contract of the boundary (reference precedent: ``captioner/models/blip2/blip2.py:19-22``
``from_pretrained``; ``utils/predictor_utils.py:182-185`` ``torch.load(path)['model']``).

No checkpoint exists offline, so tests/bench use ``procedural_blip_state_dict``: every tensor is
drawn from numpy's PCG64 keyed by (seed, crc32(name)), so the values are identical on every
machine, in any generation order, and for the HF oracle run that produced tests/golden/.
"""
from __future__ import annotations

import os
import zlib
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch

from .config import Blip2Arch, BlipArch, CocaArch, MiniLMArch


def blip_param_specs(a: BlipArch) -> List[Tuple[str, Tuple[int, ...], str, float]]:
    """(name, shape, kind, scale) for every learnable tensor, in HF state-dict naming."""
    D, M, L = a.v_hidden, a.v_mlp, a.v_layers
    P = a.patch_size
    s: List[Tuple[str, Tuple[int, ...], str, float]] = []
    vm = "vision_model."
    s.append((vm + "embeddings.class_embedding", (1, 1, D), "normal", 0.5))
    s.append((vm + "embeddings.position_embedding", (1, a.n_tokens, D), "normal", 0.5))
    s.append((vm + "embeddings.patch_embedding.weight", (D, 3, P, P), "normal", 1.0 / np.sqrt(3 * P * P)))
    s.append((vm + "embeddings.patch_embedding.bias", (D,), "normal", 0.02))
    for i in range(L):
        p = f"{vm}encoder.layers.{i}."
        s.append((p + "self_attn.qkv.weight", (3 * D, D), "normal", 1.0 / np.sqrt(D)))
        s.append((p + "self_attn.qkv.bias", (3 * D,), "normal", 0.02))
        s.append((p + "self_attn.projection.weight", (D, D), "normal", 1.0 / np.sqrt(D)))
        s.append((p + "self_attn.projection.bias", (D,), "normal", 0.02))
        s.append((p + "layer_norm1.weight", (D,), "gamma", 0.1))
        s.append((p + "layer_norm1.bias", (D,), "normal", 0.05))
        s.append((p + "mlp.fc1.weight", (M, D), "normal", 1.0 / np.sqrt(D)))
        s.append((p + "mlp.fc1.bias", (M,), "normal", 0.02))
        s.append((p + "mlp.fc2.weight", (D, M), "normal", 1.0 / np.sqrt(M)))
        s.append((p + "mlp.fc2.bias", (D,), "normal", 0.02))
        s.append((p + "layer_norm2.weight", (D,), "gamma", 0.1))
        s.append((p + "layer_norm2.bias", (D,), "normal", 0.05))
    s.append((vm + "post_layernorm.weight", (D,), "gamma", 0.1))
    s.append((vm + "post_layernorm.bias", (D,), "normal", 0.05))

    T, F, V = a.t_hidden, a.t_ffn, a.vocab
    tb = "text_decoder.bert."
    s.append((tb + "embeddings.word_embeddings.weight", (V, T), "normal", 0.08))
    s.append((tb + "embeddings.position_embeddings.weight", (a.max_pos, T), "normal", 0.08))
    s.append((tb + "embeddings.LayerNorm.weight", (T,), "gamma", 0.1))
    s.append((tb + "embeddings.LayerNorm.bias", (T,), "normal", 0.05))
    for i in range(a.t_layers):
        p = f"{tb}encoder.layer.{i}."
        for blk, kin in (("attention", T), ("crossattention", D)):
            for nm, fan in (("query", T), ("key", kin), ("value", kin)):
                s.append((f"{p}{blk}.self.{nm}.weight", (T, fan), "normal", 1.0 / np.sqrt(fan)))
                s.append((f"{p}{blk}.self.{nm}.bias", (T,), "normal", 0.02))
            # small output gains keep the token/position signal alive through 12 post-LN blocks, so greedy
            # captions vary step to step and with the image (a gain of 1 collapses to one repeated token)
            gain = 0.5 if blk == "attention" else 0.15
            s.append((f"{p}{blk}.output.dense.weight", (T, T), "normal", gain / np.sqrt(T)))
            s.append((f"{p}{blk}.output.dense.bias", (T,), "normal", 0.02))
            s.append((f"{p}{blk}.output.LayerNorm.weight", (T,), "gamma", 0.1))
            s.append((f"{p}{blk}.output.LayerNorm.bias", (T,), "normal", 0.05))
        s.append((p + "intermediate.dense.weight", (F, T), "normal", 1.0 / np.sqrt(T)))
        s.append((p + "intermediate.dense.bias", (F,), "normal", 0.02))
        s.append((p + "output.dense.weight", (T, F), "normal", 0.3 / np.sqrt(F)))
        s.append((p + "output.dense.bias", (T,), "normal", 0.02))
        s.append((p + "output.LayerNorm.weight", (T,), "gamma", 0.1))
        s.append((p + "output.LayerNorm.bias", (T,), "normal", 0.05))
    cp = "text_decoder.cls.predictions."
    s.append((cp + "transform.dense.weight", (T, T), "normal", 1.0 / np.sqrt(T)))
    s.append((cp + "transform.dense.bias", (T,), "normal", 0.02))
    s.append((cp + "transform.LayerNorm.weight", (T,), "gamma", 0.1))
    s.append((cp + "transform.LayerNorm.bias", (T,), "normal", 0.05))
    s.append((cp + "bias", (V,), "normal", 0.1))
    return s


# HF ties these at load (HF:modeling_blip_text.py:594-597); a checkpoint may or may not carry them.
BLIP_TIED = {
    "text_decoder.cls.predictions.decoder.weight": "text_decoder.bert.embeddings.word_embeddings.weight",
    "text_decoder.cls.predictions.decoder.bias": "text_decoder.cls.predictions.bias",
}


def _draw(seed: int, name: str, shape, kind: str, scale: float) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    x = rng.standard_normal(size=shape, dtype=np.float32)
    if kind == "gamma":
        return (1.0 + scale * x).astype(np.float32)
    return (scale * x).astype(np.float32)


def procedural_blip_state_dict(arch: BlipArch, seed: int = 0, eos_boost: float = 0.0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 state dict with HF key names (tied entries included, sharing storage).
    `eos_boost` is added to the LM-head bias of the EOS token so that some captions end early."""
    sd: Dict[str, torch.Tensor] = {}
    for name, shape, kind, scale in blip_param_specs(arch):
        sd[name] = torch.from_numpy(_draw(seed, name, shape, kind, scale))
    if eos_boost:
        sd["text_decoder.cls.predictions.bias"][arch.eos] += eos_boost
    for dst, src in BLIP_TIED.items():
        sd[dst] = sd[src]
    return sd


def coca_param_specs(a: CocaArch) -> List[Tuple[str, Tuple[int, ...], str, float]]:
    """open_clip CoCa state-dict names/shapes (`visual.*`, `text.*`, `text_decoder.*`; SURVEY.md §5 checkpoint contract)."""
    s: List[Tuple[str, Tuple[int, ...], str, float]] = []

    def block(p, W, F, out_gain=1.0, cross=False):
        s.append((p + ".ln_1.weight", (W,), "gamma", 0.1)); s.append((p + ".ln_1.bias", (W,), "normal", 0.05))
        if cross:
            s.append((p + ".ln_1_kv.weight", (W,), "gamma", 0.1)); s.append((p + ".ln_1_kv.bias", (W,), "normal", 0.05))
        s.append((p + ".attn.in_proj_weight", (3 * W, W), "normal", 1.0 / np.sqrt(W)))
        s.append((p + ".attn.in_proj_bias", (3 * W,), "normal", 0.02))
        s.append((p + ".attn.out_proj.weight", (W, W), "normal", out_gain / np.sqrt(W)))
        s.append((p + ".attn.out_proj.bias", (W,), "normal", 0.02))
        s.append((p + ".ln_2.weight", (W,), "gamma", 0.1)); s.append((p + ".ln_2.bias", (W,), "normal", 0.05))
        s.append((p + ".mlp.c_fc.weight", (F, W), "normal", 1.0 / np.sqrt(W)))
        s.append((p + ".mlp.c_fc.bias", (F,), "normal", 0.02))
        s.append((p + ".mlp.c_proj.weight", (W, F), "normal", out_gain / np.sqrt(F)))
        s.append((p + ".mlp.c_proj.bias", (W,), "normal", 0.02))

    D, E, T = a.v_hidden, a.embed_dim, a.t_hidden
    P = a.patch_size
    s.append(("visual.class_embedding", (D,), "normal", 0.5))
    s.append(("visual.positional_embedding", (a.n_tokens, D), "normal", 0.5))
    s.append(("visual.conv1.weight", (D, 3, P, P), "normal", 1.0 / np.sqrt(3 * P * P)))
    s.append(("visual.ln_pre.weight", (D,), "gamma", 0.1)); s.append(("visual.ln_pre.bias", (D,), "normal", 0.05))
    for i in range(a.v_layers):
        block(f"visual.transformer.resblocks.{i}", D, a.v_mlp)
    s.append(("visual.attn_pool.query", (a.pool_queries, E), "normal", 0.5))
    s.append(("visual.attn_pool.attn.q_proj_weight", (E, E), "normal", 1.0 / np.sqrt(E)))
    s.append(("visual.attn_pool.attn.k_proj_weight", (E, D), "normal", 1.0 / np.sqrt(D)))
    s.append(("visual.attn_pool.attn.v_proj_weight", (E, D), "normal", 1.0 / np.sqrt(D)))
    s.append(("visual.attn_pool.attn.in_proj_bias", (3 * E,), "normal", 0.02))
    s.append(("visual.attn_pool.attn.out_proj.weight", (E, E), "normal", 1.0 / np.sqrt(E)))
    s.append(("visual.attn_pool.attn.out_proj.bias", (E,), "normal", 0.02))
    for n, w in (("ln_q", E), ("ln_k", D)):
        s.append((f"visual.attn_pool.{n}.weight", (w,), "gamma", 0.1)); s.append((f"visual.attn_pool.{n}.bias", (w,), "normal", 0.05))
    s.append(("visual.ln_post.weight", (E,), "gamma", 0.1)); s.append(("visual.ln_post.bias", (E,), "normal", 0.05))
    s.append(("visual.proj", (E, E), "normal", 1.0 / np.sqrt(E)))
    s.append(("text.token_embedding.weight", (a.vocab, T), "normal", 0.5))
    s.append(("text.positional_embedding", (a.context_length + 1, T), "normal", 0.5))
    s.append(("text.cls_emb", (T,), "normal", 0.5))
    for i in range(a.t_layers):
        block(f"text.transformer.resblocks.{i}", T, a.t_ffn, out_gain=0.4)
    s.append(("text.ln_final.weight", (T,), "gamma", 0.1)); s.append(("text.ln_final.bias", (T,), "normal", 0.05))
    s.append(("text.text_projection", (T, E), "normal", 1.0 / np.sqrt(T)))
    for i in range(a.mm_layers):
        block(f"text_decoder.resblocks.{i}", T, a.t_ffn, out_gain=0.4)
        block(f"text_decoder.cross_attn.{i}", T, a.t_ffn, out_gain=0.4, cross=True)
    s.append(("text_decoder.ln_final.weight", (T,), "gamma", 0.1)); s.append(("text_decoder.ln_final.bias", (T,), "normal", 0.05))
    s.append(("text_decoder.text_projection", (T, a.vocab), "normal", 0.08))
    return s


def procedural_coca_state_dict(arch: CocaArch, seed: int = 0, eos_boost: float = 0.0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 CoCa state dict (open_clip key names).  `eos_boost` shifts the EOS logit through ln_final's bias
    direction (the head has no bias vector of its own)."""
    sd: Dict[str, torch.Tensor] = {}
    for name, shape, kind, scale in coca_param_specs(arch):
        sd[name] = torch.from_numpy(_draw(seed, name, shape, kind, scale))
    if eos_boost:
        beta = sd["text_decoder.ln_final.bias"]
        sd["text_decoder.text_projection"][:, arch.eos] += eos_boost * beta / float(beta @ beta)
    sd["logit_scale"] = torch.tensor(float(np.log(1 / 0.07)))
    return sd


def minilm_param_specs(a: MiniLMArch) -> List[Tuple[str, Tuple[int, ...], str, float]]:
    """HF BertModel state-dict names (what sentence-transformers saves for all-MiniLM-L6-v2; pooler absent/unused)."""
    T, F = a.hidden, a.ffn
    s: List[Tuple[str, Tuple[int, ...], str, float]] = [
        ("embeddings.word_embeddings.weight", (a.vocab, T), "normal", 0.5),
        ("embeddings.position_embeddings.weight", (a.max_pos, T), "normal", 0.2),
        ("embeddings.token_type_embeddings.weight", (2, T), "normal", 0.1),
        ("embeddings.LayerNorm.weight", (T,), "gamma", 0.1),
        ("embeddings.LayerNorm.bias", (T,), "normal", 0.05),
    ]
    for i in range(a.layers):
        p = f"encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            s.append((p + f"attention.self.{nm}.weight", (T, T), "normal", 1.5 / np.sqrt(T)))
            s.append((p + f"attention.self.{nm}.bias", (T,), "normal", 0.02))
        s.append((p + "attention.output.dense.weight", (T, T), "normal", 1.0 / np.sqrt(T)))
        s.append((p + "attention.output.dense.bias", (T,), "normal", 0.02))
        s.append((p + "attention.output.LayerNorm.weight", (T,), "gamma", 0.1))
        s.append((p + "attention.output.LayerNorm.bias", (T,), "normal", 0.05))
        s.append((p + "intermediate.dense.weight", (F, T), "normal", 1.0 / np.sqrt(T)))
        s.append((p + "intermediate.dense.bias", (F,), "normal", 0.02))
        s.append((p + "output.dense.weight", (T, F), "normal", 1.0 / np.sqrt(F)))
        s.append((p + "output.dense.bias", (T,), "normal", 0.02))
        s.append((p + "output.LayerNorm.weight", (T,), "gamma", 0.1))
        s.append((p + "output.LayerNorm.bias", (T,), "normal", 0.05))
    return s


def procedural_minilm_state_dict(arch: MiniLMArch, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {name: torch.from_numpy(_draw(seed, name, shape, kind, scale)) for name, shape, kind, scale in minilm_param_specs(arch)}


def blip2_param_specs(a: Blip2Arch) -> List[Tuple[str, Tuple[int, ...], str, float]]:
    """`Blip2ForConditionalGeneration.state_dict()` names (transformers 5.x: one fused `qkv` Linear with bias in the ViT)."""
    s: List[Tuple[str, Tuple[int, ...], str, float]] = []
    D, M, P = a.v_hidden, a.v_mlp, a.patch_size
    vm = "vision_model."
    s.append((vm + "embeddings.class_embedding", (1, 1, D), "normal", 0.5))
    s.append((vm + "embeddings.position_embedding", (1, a.n_tokens, D), "normal", 0.5))
    s.append((vm + "embeddings.patch_embedding.weight", (D, 3, P, P), "normal", 1.0 / np.sqrt(3 * P * P)))
    s.append((vm + "embeddings.patch_embedding.bias", (D,), "normal", 0.02))
    for i in range(a.v_layers):
        p = f"{vm}encoder.layers.{i}."
        s += [(p + "self_attn.qkv.weight", (3 * D, D), "normal", 1.0 / np.sqrt(D)), (p + "self_attn.qkv.bias", (3 * D,), "normal", 0.02),
              (p + "self_attn.projection.weight", (D, D), "normal", 0.5 / np.sqrt(D)), (p + "self_attn.projection.bias", (D,), "normal", 0.02),
              (p + "layer_norm1.weight", (D,), "gamma", 0.1), (p + "layer_norm1.bias", (D,), "normal", 0.05),
              (p + "mlp.fc1.weight", (M, D), "normal", 1.0 / np.sqrt(D)), (p + "mlp.fc1.bias", (M,), "normal", 0.02),
              (p + "mlp.fc2.weight", (D, M), "normal", 0.5 / np.sqrt(M)), (p + "mlp.fc2.bias", (D,), "normal", 0.02),
              (p + "layer_norm2.weight", (D,), "gamma", 0.1), (p + "layer_norm2.bias", (D,), "normal", 0.05)]
    s += [(vm + "post_layernorm.weight", (D,), "gamma", 0.1), (vm + "post_layernorm.bias", (D,), "normal", 0.05)]
    Q, F = a.q_hidden, a.q_ffn
    s.append(("query_tokens", (1, a.num_query_tokens, Q), "normal", 0.5))
    s += [("qformer.layernorm.weight", (Q,), "gamma", 0.1), ("qformer.layernorm.bias", (Q,), "normal", 0.05)]
    for i in range(a.q_layers):
        p = f"qformer.encoder.layer.{i}."
        blocks = [("attention", Q)] + ([("crossattention", D)] if i % a.q_cross_freq == 0 else [])
        for blk, kin in blocks:
            for nm, fan in (("query", Q), ("key", kin), ("value", kin)):
                s.append((p + f"{blk}.attention.{nm}.weight", (Q, fan), "normal", 1.2 / np.sqrt(fan)))
                s.append((p + f"{blk}.attention.{nm}.bias", (Q,), "normal", 0.02))
            s += [(p + f"{blk}.output.dense.weight", (Q, Q), "normal", 1.0 / np.sqrt(Q)), (p + f"{blk}.output.dense.bias", (Q,), "normal", 0.02),
                  (p + f"{blk}.output.LayerNorm.weight", (Q,), "gamma", 0.1), (p + f"{blk}.output.LayerNorm.bias", (Q,), "normal", 0.05)]
        s += [(p + "intermediate_query.dense.weight", (F, Q), "normal", 1.0 / np.sqrt(Q)), (p + "intermediate_query.dense.bias", (F,), "normal", 0.02),
              (p + "output_query.dense.weight", (Q, F), "normal", 1.0 / np.sqrt(F)), (p + "output_query.dense.bias", (Q,), "normal", 0.02),
              (p + "output_query.LayerNorm.weight", (Q,), "gamma", 0.1), (p + "output_query.LayerNorm.bias", (Q,), "normal", 0.05)]
    T, G = a.t_hidden, a.t_ffn
    s += [("language_projection.weight", (T, Q), "normal", 1.0 / np.sqrt(Q)), ("language_projection.bias", (T,), "normal", 0.02)]
    lm = "language_model.model.decoder."
    s += [(lm + "embed_tokens.weight", (a.vocab, T), "normal", 0.08), (lm + "embed_positions.weight", (a.max_pos + 2, T), "normal", 0.25)]
    for i in range(a.t_layers):
        p = f"{lm}layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj"):
            s += [(p + f"self_attn.{nm}.weight", (T, T), "normal", 1.2 / np.sqrt(T)), (p + f"self_attn.{nm}.bias", (T,), "normal", 0.02)]
        s += [(p + "self_attn.out_proj.weight", (T, T), "normal", 1.5 / np.sqrt(T)), (p + "self_attn.out_proj.bias", (T,), "normal", 0.02),
              (p + "self_attn_layer_norm.weight", (T,), "gamma", 0.1), (p + "self_attn_layer_norm.bias", (T,), "normal", 0.05),
              (p + "fc1.weight", (G, T), "normal", 1.0 / np.sqrt(T)), (p + "fc1.bias", (G,), "normal", 0.02),
              (p + "fc2.weight", (T, G), "normal", 1.5 / np.sqrt(G)), (p + "fc2.bias", (T,), "normal", 0.02),
              (p + "final_layer_norm.weight", (T,), "gamma", 0.1), (p + "final_layer_norm.bias", (T,), "normal", 0.05)]
    s += [(lm + "final_layer_norm.weight", (T,), "gamma", 0.1), (lm + "final_layer_norm.bias", (T,), "normal", 0.05)]
    return s


def quantize_int8_rowwise(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """bitsandbytes' vector-wise int8 quantisation of a Linear weight [out, in] (what `load_in_8bit=True` stores in a
    `Linear8bitLt`: `CB` int8 and the rows' absmax `SCB`): q = rint(w * (127 / absmax(row))) in fp32 (round-half-even), scale =
    absmax / 127; w ~ q * scale.  The library does this on the device for the tensors it keeps as bytes (csrc/gemm_skinny.hip,
    bit for bit this arithmetic); this host form is for the tensors that are stored dequantised."""
    w = w.detach().float()
    amax = w.abs().amax(dim=1, keepdim=True)
    inv = torch.where(amax > 0, torch.full_like(amax, 127.0) / amax, torch.zeros_like(amax))   # (a true division: `127.0 / t` is t.reciprocal() * 127 in torch)
    q = torch.round(w * inv).to(torch.int8)
    return q, (amax / 127.0).squeeze(1)


def int8_roundtrip(w: torch.Tensor) -> torch.Tensor:
    q, s = quantize_int8_rowwise(w)
    return q.float() * s[:, None]


def blip2_int8_host_names(sd: Dict[str, torch.Tensor]) -> List[str]:
    """`load_in_8bit` on BLIP-2 (transformers `replace_with_bnb_linear`): every nn.Linear outside `_keep_in_fp32_modules`
    (= the Q-Former and the query tokens) and outside lm_head becomes int8.  The OPT decoder layers' Linears are kept as bytes by
    the library (CapConfig.weight_int8); the names returned here - the vision tower's Linears and language_projection - are
    the rest: stored at the compute dtype, so they pass through the quantiser on the host (weights only, same values)."""
    return [k for k in sd if k.endswith(".weight") and sd[k].dim() == 2 and
            ((k.startswith("vision_model.encoder.layers.") and (".self_attn." in k or ".mlp." in k)) or k == "language_projection.weight")]


def procedural_blip2_state_dict(arch: Blip2Arch, seed: int = 0, eos_boost: float = 0.0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 state dict; `language_model.lm_head.weight` is tied to the token table (shares storage).  The LM head has
    no bias: `eos_boost` adds boost * beta / |beta|^2 (beta = bias of the final LayerNorm) to the EOS row of the tied table,
    which raises the EOS logit by about `eos_boost` at every step so that some captions end early."""
    sd = {name: torch.from_numpy(_draw(seed, name, shape, kind, scale)) for name, shape, kind, scale in blip2_param_specs(arch)}
    if eos_boost:
        beta = sd["language_model.model.decoder.final_layer_norm.bias"]
        sd["language_model.model.decoder.embed_tokens.weight"][arch.eos] += eos_boost * beta / float(beta.pow(2).sum())
    sd["language_model.lm_head.weight"] = sd["language_model.model.decoder.embed_tokens.weight"]
    return sd


def synthetic_token_batch(arch: MiniLMArch, batch: int, max_len: int, seed: int = 0):
    """Seeded WordPiece-like id rows: [CLS] w.. [SEP] then pad; lengths 3..max_len (ragged).  -> (ids int32 [B, L], lens int32 [B])"""
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    lens = rng.integers(3, max_len + 1, size=batch)
    lens[0] = max_len
    ids = np.full((batch, max_len), arch.pad, dtype=np.int32)
    for b in range(batch):
        n = int(lens[b])
        ids[b, 0] = arch.cls
        ids[b, 1:n - 1] = rng.integers(max(arch.sep, arch.cls) + 1, arch.vocab, size=n - 2)
        ids[b, n - 1] = arch.sep
    return torch.from_numpy(ids), torch.from_numpy(lens.astype(np.int32))


def synthetic_pixels(batch: int, image_size: int, seed: int = 0, first: int = 0) -> torch.Tensor:
    """Normalised fp32 NCHW frames, one PCG64 stream per frame index (frame i is the same in any batch).

    Stands in for `BlipImageProcessor` output on 224x224 crops (synthetic
    frames are generated from seed = frame index so shards need no shared storage).
    """
    out = np.empty((batch, 3, image_size, image_size), dtype=np.float32)
    for i in range(batch):
        rng = np.random.Generator(np.random.PCG64([seed, 0x1A6E, first + i]))
        out[i] = rng.standard_normal(size=(3, image_size, image_size), dtype=np.float32)
    return torch.from_numpy(out)


def synthetic_frames_u8(batch: int, height: int, width: int, seed: int = 0, first: int = 0) -> torch.Tensor:
    """uint8 HWC RGB frames (what the callers crop out of habitat observations)."""
    out = np.empty((batch, height, width, 3), dtype=np.uint8)
    for i in range(batch):
        rng = np.random.Generator(np.random.PCG64([seed, 0xF8A3E, first + i]))
        out[i] = rng.integers(0, 256, size=(height, width, 3), dtype=np.uint8)
    return torch.from_numpy(out)


# ----------------------------------------------------------------------------------------------
# checkpoint readers (HF directory / safetensors / torch pickles / {'model': sd} wrappers)
# ----------------------------------------------------------------------------------------------

def _strip_prefixes(sd: Dict[str, torch.Tensor], prefixes: Iterable[str] = ("module.",)):
    """DDP-saved checkpoints carry a 'module.' prefix (reference: factory.py:139-142)."""
    out = {}
    for k, v in sd.items():
        for p in prefixes:
            if k.startswith(p):
                k = k[len(p):]
        out[k] = v
    return out


def strip_wrapper_prefixes(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Keys of a checkpoint saved from the reference's WRAPPER module (`torch.save({'model': captioner.state_dict()})`,
    loaded back by predictor_utils.py:182-185 into the wrapper): the wrapped network is the attribute `model`, DDP adds
    `module.` - both are dropped, repeatedly, from the front of every key."""
    out = {}
    for k, v in sd.items():
        while k.startswith("module.") or k.startswith("model."):
            k = k.split(".", 1)[1]
        out[k] = v
    return out


def load_state_dict_file(path: str) -> Dict[str, torch.Tensor]:
    """Read one weight file. Accepts .safetensors, or a torch pickle that is a state dict or wraps one
    under 'model' / 'state_dict' (reference: predictor_utils.py:182-185, evaluate_finetuned_model.py:139-146,
    factory.py:131-143)."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        obj = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):
            obj = obj["model"]
        elif isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
            obj = obj["state_dict"]
        sd = obj
    return _strip_prefixes(sd, ("module.",))


# Largest spread of the 64 dimensions of one cross-attention key / value head that the KV16 cache (int16 + ONE scale per head row)
# takes: measured on the GPU with two dimensions per head `factor` times the others (tests/probe_kv16_outliers.py,
# profiles/r04_kv16_outlier_probe.txt), the logits move by 1.5e-4 at factor 8, 3.0e-4 at 12, 5.4e-4 at 16 and 2.3e-3 at 30 against
# the same engine with fp32 rows - the parity bar is 1e-3.  Beyond this spread the cache must keep fp32 rows (cross_cache="fp32").
KV16_MAX_HEAD_SPREAD = 12.0


def cross_kv_head_spread(sd: Dict[str, torch.Tensor]) -> float:
    """max over the key / value heads of every cross-attention layer of (largest / median) magnitude of the head's 64 output
    dimensions, a dimension's magnitude being sqrt(|W row|^2 + b^2) - what its values are sized like on LayerNorm'ed inputs (the
    LayerNorm's own gamma / beta folded into W / b).  A weight-side PROXY: it cannot see outliers that only the activations carry.
    BLIP (`...crossattention.self.{key,value}.*` with `vision_model.post_layernorm` folded in), CoCa (`text_decoder.cross_attn.<i>.attn.in_proj_*`, k and v rows, with
    `ln_1_kv` folded in as the library does) or the library's own `derived.cross_kv.*`.  1.0 when the dict has no such tensors."""
    mags = []
    if "derived.cross_kv.weight" in sd:
        w, b = sd["derived.cross_kv.weight"].float(), sd["derived.cross_kv.bias"].float()
        mags.append((w.pow(2).sum(1) + b.pow(2)).sqrt())
    else:
        # BLIP: the projections read the image tower's post_layernorm output x = xhat * gamma + beta (xhat unit-scale), so what sizes
        # a dimension is the row of W * gamma and the bias b + W . beta - a checkpoint whose final LayerNorm carries a few large
        # gamma channels (the usual place of a ViT's massive activations) shows up here, not in W alone
        pg, pb = sd.get("vision_model.post_layernorm.weight"), sd.get("vision_model.post_layernorm.bias")
        for k, w in sd.items():
            if k.endswith(".weight") and (".crossattention.self.key." in k or ".crossattention.self.value." in k):
                b = sd.get(k[:-6] + "bias")
                w = w.float()
                b = b.float() if b is not None else w.new_zeros(w.shape[0])
                if pg is not None and pg.numel() == w.shape[1]:
                    if pb is not None:
                        b = b + w @ pb.float()
                    w = w * pg.float()[None, :]
                mags.append((w.pow(2).sum(1) + b.pow(2)).sqrt())
            elif k.startswith("text_decoder.cross_attn.") and k.endswith(".attn.in_proj_weight"):
                pre = k[: -len("attn.in_proj_weight")]
                w, b = w.float(), sd[pre + "attn.in_proj_bias"].float()
                E = w.shape[1]
                g, beta = sd.get(pre + "ln_1_kv.weight"), sd.get(pre + "ln_1_kv.bias")
                wkv, bkv = w[E:], b[E:]
                if g is not None:
                    bkv = bkv + wkv @ beta.float()
                    wkv = wkv * g.float()[None, :]
                mags.append((wkv.pow(2).sum(1) + bkv.pow(2)).sqrt())
    spread = 1.0
    for m in mags:
        h = m.reshape(-1, 64)
        med = h.median(dim=1).values.clamp_min(1e-30)
        spread = max(spread, float((h.max(dim=1).values / med).max()))
    return spread


def _peft_pattern_value(pattern: dict, module: str, default):
    """PEFT's rank_pattern / alpha_pattern: keys are module-name suffixes or regular expressions (peft.utils.get_pattern_key)."""
    import re
    for key, val in (pattern or {}).items():
        if module == key or module.endswith("." + key) or re.fullmatch(key, module) or re.search(rf"(^|\.){key}$", module):
            return val
    return default


def merge_peft_lora(sd: Dict[str, torch.Tensor], adapter: str, strict: bool = True) -> Tuple[Dict[str, torch.Tensor], Dict[str, object]]:
    """Fold a PEFT LoRA adapter into a base state dict - what `PeftModel.from_pretrained(model, ckpt_path)` + the adapted
    forward compute (reference: scripts/evaluate_finetuned_model.py:147-148, the fine-tuned BLIP-2 of the paper):

        W' = W + scaling * (lora_B @ lora_A),   scaling = lora_alpha / r   (lora_alpha / sqrt(r) with use_rslora)

    in fp32, per adapted module (transposed when the config says fan_in_fan_out).  `adapter`: a PEFT output directory
    (adapter_config.json + adapter_model.safetensors | adapter_model.bin) or the weight file itself with the config beside it.
    Keys: `base_model.model.<module path>.lora_A[.<adapter name>].weight` / `lora_B...`; `modules_to_save` copies
    (`<module>.modules_to_save[.<name>].<param>`) replace the base tensor.  Anything the merge does not implement - DoRA
    magnitude vectors, non-LoRA PEFT types, LoRA on embeddings, trained LoRA biases - is rejected by name.
    Returns (new state dict, {"merged": n, "replaced": n, "scaling": {module: s}})."""
    import json
    cfg_dir = adapter if os.path.isdir(adapter) else os.path.dirname(adapter)
    cfg_path = os.path.join(cfg_dir, "adapter_config.json")
    if not os.path.exists(cfg_path):
        raise RuntimeError(f"PEFT adapter: no adapter_config.json under {cfg_dir}")
    cfg = json.load(open(cfg_path))
    ptype = str(cfg.get("peft_type", "LORA")).upper()
    if ptype != "LORA":
        raise RuntimeError(f"PEFT adapter of type {ptype} is not supported (LoRA only)")
    if cfg.get("use_dora"):
        raise RuntimeError("PEFT adapter: use_dora (weight-decomposed LoRA) is not supported")
    if str(cfg.get("bias", "none")) != "none":
        raise RuntimeError(f"PEFT adapter: bias='{cfg.get('bias')}' (trained biases stored with the adapter) is not supported")
    if os.path.isdir(adapter):
        for fn in ("adapter_model.safetensors", "adapter_model.bin"):
            if os.path.exists(os.path.join(adapter, fn)):
                ad = load_state_dict_file(os.path.join(adapter, fn))
                break
        else:
            raise RuntimeError(f"PEFT adapter: no adapter_model.safetensors / adapter_model.bin under {adapter}")
    else:
        ad = load_state_dict_file(adapter)
    r0, a0 = int(cfg.get("r", 8)), float(cfg.get("lora_alpha", 8))
    out = dict(sd)
    report = {"merged": 0, "replaced": 0, "scaling": {}}
    import re
    pat = re.compile(r"^(?:base_model\.model\.)?(?P<mod>.+?)\.lora_(?P<ab>[AB])(?:\.[^.]+)?\.weight$")
    pairs: Dict[str, Dict[str, torch.Tensor]] = {}
    for k, v in ad.items():
        m = pat.match(k)
        if m:
            pairs.setdefault(m.group("mod"), {})[m.group("ab")] = v
            continue
        ms = re.match(r"^(?:base_model\.model\.)?(?P<mod>.+?)\.modules_to_save(?:\.[^.]+)?\.(?P<par>weight|bias)$", k)
        if ms:
            key = f"{ms.group('mod')}.{ms.group('par')}"
            if key not in out and strict:
                raise RuntimeError(f"PEFT adapter: modules_to_save tensor {k} has no counterpart {key} in the base checkpoint")
            out[key] = v.float()
            report["replaced"] += 1
            continue
        if "lora_embedding" in k or "lora_magnitude" in k:
            raise RuntimeError(f"PEFT adapter: tensor {k} (LoRA on embeddings / DoRA) is not supported")
        if strict:
            raise RuntimeError(f"PEFT adapter: tensor {k} is not a LoRA A/B matrix nor a modules_to_save copy")
    for mod, ab in pairs.items():
        if "A" not in ab or "B" not in ab:
            raise RuntimeError(f"PEFT adapter: module {mod} has only one of lora_A / lora_B")
        key = mod + ".weight"
        if key not in out:
            raise RuntimeError(f"PEFT adapter: adapted module {mod} has no weight {key} in the base checkpoint")
        A, B = ab["A"].float(), ab["B"].float()            # [r, in], [out, r]
        r = int(_peft_pattern_value(cfg.get("rank_pattern"), mod, r0))
        if A.shape[0] != r or B.shape[1] != r:
            r = A.shape[0]                                   # the tensors are what was trained; the config's pattern did not name them
        alpha = float(_peft_pattern_value(cfg.get("alpha_pattern"), mod, a0))
        scaling = alpha / (r ** 0.5) if cfg.get("use_rslora") else alpha / r
        delta = (B @ A) * scaling
        if cfg.get("fan_in_fan_out"):
            delta = delta.t()
        W = out[key].float()
        if delta.shape != W.shape:
            raise RuntimeError(f"PEFT adapter: {mod}: B @ A is {tuple(delta.shape)}, the base weight {tuple(W.shape)}")
        out[key] = W + delta
        report["merged"] += 1
        report["scaling"][mod] = scaling
    if not report["merged"] and not report["replaced"]:
        raise RuntimeError(f"PEFT adapter {adapter}: no LoRA matrices found")
    return out, report


def is_peft_adapter(path: str) -> bool:
    return bool(path) and os.path.exists(os.path.join(path if os.path.isdir(path) else os.path.dirname(path), "adapter_config.json"))


def load_hf_blip_checkpoint(model_dir: str) -> Tuple[BlipArch, Dict[str, torch.Tensor]]:
    """HF ``from_pretrained`` directory layout: config.json + model.safetensors | pytorch_model.bin."""
    arch = BlipArch.from_hf_config(model_dir)
    for fn in ("model.safetensors", "pytorch_model.bin"):
        p = os.path.join(model_dir, fn)
        if os.path.exists(p):
            sd = load_state_dict_file(p)
            break
    else:
        raise RuntimeError(f"no model.safetensors / pytorch_model.bin under {model_dir}")
    for dst, src in BLIP_TIED.items():
        if dst not in sd and src in sd:
            sd[dst] = sd[src]
    # position table tells the image size the checkpoint was trained at
    pe = sd["vision_model.embeddings.position_embedding"]
    g = int(round((pe.shape[1] - 1) ** 0.5))
    arch.image_size = g * arch.patch_size
    return arch, sd


def resolve_hf_dir(model_name: str) -> str | None:
    """Map a hub id ('Salesforce/blip-image-captioning-base') or a path to a local directory, offline."""
    if os.path.isdir(model_name):
        return model_name
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(model_name, local_files_only=True)
    except Exception:
        return None
