"""Architecture constants of the captioner forward path.

BLIP-base values follow the HF config defaults the checkpoint's ``config.json``
overrides at load time (HF:models/blip/configuration_blip.py:52-107; SURVEY.md §8
"Model constants").  CoCa values follow the reference's
``experimenting_env/captioner/models/coca/model_configs/coca_ViT-L-14.json:1-30``.
"""
from __future__ import annotations

import dataclasses
import json
import os


@dataclasses.dataclass
class BlipArch:
    # vision tower (ViT-B/16)
    image_size: int = 224
    patch_size: int = 16
    v_hidden: int = 768
    v_layers: int = 12
    v_heads: int = 12
    v_mlp: int = 3072
    v_eps: float = 1e-5
    # text decoder (BERT-style, post-LN, cross-attention to image tokens)
    t_hidden: int = 768
    t_layers: int = 12
    t_heads: int = 12
    t_ffn: int = 3072
    vocab: int = 30524
    max_pos: int = 512
    t_eps: float = 1e-12
    bos: int = 30522
    eos: int = 102      # sep_token_id: what BlipForConditionalGeneration.generate passes as eos
    pad: int = 0

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def n_patches(self) -> int:
        return self.grid * self.grid

    @property
    def n_tokens(self) -> int:
        return self.n_patches + 1

    @property
    def v_head_dim(self) -> int:
        return self.v_hidden // self.v_heads

    @property
    def t_head_dim(self) -> int:
        return self.t_hidden // self.t_heads

    def encoder_flops_per_image(self) -> float:
        """2*MAC of patch-embed + 12x(QKV, QK^T, PV, proj, FC1, FC2); BASELINE.md §4."""
        n, d, m = self.n_tokens, self.v_hidden, self.v_mlp
        per_layer = 2.0 * n * d * 3 * d + 2 * 2.0 * n * n * d + 2.0 * n * d * d + 2 * 2.0 * n * d * m
        patch = 2.0 * self.n_patches * (3 * self.patch_size ** 2) * d
        return per_layer * self.v_layers + patch

    def cross_kv_flops_per_image(self) -> float:
        return self.t_layers * 2 * 2.0 * self.n_tokens * self.v_hidden * self.t_hidden

    def decode_flops_per_token(self) -> float:
        d, f, v = self.t_hidden, self.t_ffn, self.vocab
        per_layer = 2.0 * d * d * 4 + 2.0 * d * d * 2 + 2 * 2.0 * d * f   # self q,k,v,o + cross q,o + ffn
        return per_layer * self.t_layers + 2.0 * d * d + 2.0 * d * v

    @staticmethod
    def tiny() -> "BlipArch":
        """Fixture-sized config (tests/golden/blip_tiny*.npz): hidden 128 (2 heads of 64), 2 layers, vocab 512,
        32x32 image with 8x8 patches (17 tokens)."""
        return BlipArch(image_size=32, patch_size=8, v_hidden=128, v_layers=2, v_heads=2, v_mlp=256,
                        t_hidden=128, t_layers=2, t_heads=2, t_ffn=256, vocab=512, max_pos=40,
                        bos=510, eos=102, pad=0)

    @staticmethod
    def from_hf_config(path_or_dict) -> "BlipArch":
        """Read a HF ``config.json`` (checkpoint directory or parsed dict)."""
        if isinstance(path_or_dict, (str, os.PathLike)):
            p = path_or_dict
            if os.path.isdir(p):
                p = os.path.join(p, "config.json")
            with open(p) as f:
                cfg = json.load(f)
        else:
            cfg = dict(path_or_dict)
        v = cfg.get("vision_config", {})
        t = cfg.get("text_config", {})
        a = BlipArch()
        a.image_size = v.get("image_size", 384)
        a.patch_size = v.get("patch_size", 16)
        a.v_hidden = v.get("hidden_size", 768)
        a.v_layers = v.get("num_hidden_layers", 12)
        a.v_heads = v.get("num_attention_heads", 12)
        a.v_mlp = v.get("intermediate_size", 3072)
        a.v_eps = v.get("layer_norm_eps", 1e-5)
        a.t_hidden = t.get("hidden_size", 768)
        a.t_layers = t.get("num_hidden_layers", 12)
        a.t_heads = t.get("num_attention_heads", 8)
        a.t_ffn = t.get("intermediate_size", 3072)
        a.vocab = t.get("vocab_size", 30524)
        a.max_pos = t.get("max_position_embeddings", 512)
        a.t_eps = t.get("layer_norm_eps", 1e-12)
        a.bos = t.get("bos_token_id", 30522)
        a.eos = t.get("sep_token_id", 102)
        a.pad = t.get("pad_token_id", 0)
        return a


@dataclasses.dataclass
class CocaArch:
    """CoCa ViT-L/14 - reference ``experimenting_env/captioner/models/coca/model_configs/coca_ViT-L-14.json:1-30``
    (+ open_clip defaults: 256 pooler queries, mlp_ratio 4, LayerNorm eps 1e-5, exact GELU)."""
    image_size: int = 224
    patch_size: int = 14
    v_hidden: int = 1024
    v_layers: int = 24
    v_heads: int = 16
    v_mlp: int = 4096
    embed_dim: int = 768          # pooler output width = text width
    pool_queries: int = 256
    pool_heads: int = 8
    t_hidden: int = 768
    t_layers: int = 12            # unimodal text tower
    mm_layers: int = 12           # multimodal decoder (each layer = causal self-attn block + cross-attn block)
    t_heads: int = 12
    t_ffn: int = 3072
    vocab: int = 49408
    context_length: int = 76
    eps: float = 1e-5
    sot: int = 49406
    eos: int = 49407
    pad: int = 0
    seq_len: int = 30             # coca_model.py:209 generate defaults
    min_seq_len: int = 5

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def n_tokens(self) -> int:
        return self.grid * self.grid + 1

    @property
    def n_image_embs(self) -> int:
        return self.pool_queries - 1

    @staticmethod
    def tiny() -> "CocaArch":
        return CocaArch(image_size=28, patch_size=14, v_hidden=128, v_layers=2, v_heads=2, v_mlp=256, embed_dim=128,
                        pool_queries=8, pool_heads=2, t_hidden=128, t_layers=2, mm_layers=2, t_heads=2, t_ffn=256,
                        vocab=512, context_length=20, sot=510, eos=511, pad=0, seq_len=12, min_seq_len=3)


@dataclasses.dataclass
class MiniLMArch:
    """sentence-transformers/all-MiniLM-L6-v2 (the caption embedder of the reference: goal_exploration.py:57,
    pseudolabeler.py:568): HF BertModel 6 x 384, 12 heads of 32, FFN 1536, post-LN eps 1e-12, WordPiece vocab 30522,
    then mean pooling over the attention mask and L2 normalisation -> 384-d."""
    hidden: int = 384
    layers: int = 6
    heads: int = 12
    ffn: int = 1536
    vocab: int = 30522
    max_pos: int = 512
    eps: float = 1e-12
    cls: int = 101
    sep: int = 102
    pad: int = 0
    max_seq_length: int = 256      # sentence-transformers truncation for this model

    @staticmethod
    def tiny() -> "MiniLMArch":
        return MiniLMArch(hidden=64, layers=2, heads=2, ffn=128, vocab=300, max_pos=40, cls=1, sep=2, pad=0,
                          max_seq_length=32)


@dataclasses.dataclass
class Blip2Arch:
    """BLIP-2 OPT (reference `captioner/models/blip2/blip2.py:19-22`: `Salesforce/blip2-opt-2.7b`) - HF defaults of
    `HF:models/blip_2/configuration_blip_2.py` for the vision tower (ViT-g/14: 39 x 1408, 16 heads of 88) and the
    Q-Former (12 x 768, 32 queries, cross-attention every 2nd layer), `facebook/opt-2.7b` for the language model
    (32 x 2560, 32 heads of 80, FFN 10240, ReLU, pre-LN, learned positions with offset 2, tied LM head)."""
    image_size: int = 224
    patch_size: int = 14
    v_hidden: int = 1408
    v_layers: int = 39
    v_heads: int = 16
    v_mlp: int = 6144
    v_eps: float = 1e-6
    q_hidden: int = 768
    q_layers: int = 12
    q_heads: int = 12
    q_ffn: int = 3072
    q_cross_freq: int = 2
    q_eps: float = 1e-12
    num_query_tokens: int = 32
    t_hidden: int = 2560
    t_layers: int = 32
    t_heads: int = 32
    t_ffn: int = 10240
    vocab: int = 50272
    max_pos: int = 2048
    t_eps: float = 1e-5
    bos: int = 2
    eos: int = 50118          # generation_config of blip2-opt-2.7b ("\n"); OPT's own eos is 2
    pad: int = 1
    image_token: int = 50265
    max_new_tokens: int = 20  # HF generate default when the caller gives no length (blip2.py:26 gives none)

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2

    @property
    def n_tokens(self) -> int:
        return self.n_patches + 1

    @staticmethod
    def tiny() -> "Blip2Arch":
        # head dims of the real model's kinds: vision 24 (not a power of two, like 88), q-former 64, language model 16
        return Blip2Arch(image_size=28, patch_size=14, v_hidden=192, v_layers=2, v_heads=8, v_mlp=256, q_hidden=128, q_layers=2,
                         q_heads=2, q_ffn=256, num_query_tokens=8, t_hidden=64, t_layers=2, t_heads=4, t_ffn=128, vocab=512,
                         max_pos=64, eos=3, image_token=511)

    @staticmethod
    def small() -> "Blip2Arch":
        # the fixture geometry with OPT widths the int8 weight stream takes (`load_in_8bit`: multiples of 256)
        return Blip2Arch(image_size=28, patch_size=14, v_hidden=192, v_layers=2, v_heads=8, v_mlp=256, q_hidden=128, q_layers=2,
                         q_heads=2, q_ffn=256, num_query_tokens=8, t_hidden=256, t_layers=2, t_heads=4, t_ffn=512, vocab=512,
                         max_pos=64, eos=3, image_token=511)
