"""Caption embedder behind the call the reference makes after every caption:

    self.encoder = SentenceTransformer("all-MiniLM-L6-v2").to(device)      # goal_exploration.py:57, pseudolabeler.py:568
    emb = self.encoder.encode(caption)                                     # goal_exploration.py:102  -> numpy [384]
    emb = self.encoder.encode(caption, convert_to_tensor=True)             # pseudolabeler.py:677     -> tensor [384]

`SentenceEncoder(name)` keeps that surface (`encode(str | list[str], convert_to_tensor=False, batch_size=32,
normalize_embeddings=...)`, `.to(device)`, `get_sentence_embedding_dimension()`); the BertModel + mean pooling + L2
normalisation run in libcaptioner_hip.so (`TextEncoderEngine`).  Tokenisation (WordPiece, lower-casing, truncation at
max_seq_length 256) is host work done with the checkpoint's own `tokenizer.json` through the `tokenizers` package.

name: a local sentence-transformers directory / cached hub id, or `procedural-minilm[-tiny][:seed]` (seeded weights, ids
passed directly through `encode_ids` - no vocabulary exists offline).
"""
from __future__ import annotations

import glob
import json
import os
from typing import List, Sequence, Union

import numpy as np
import torch

from ..config import MiniLMArch
from ..engine import TextEncoderEngine
from ..weights import load_state_dict_file, procedural_minilm_state_dict


def _resolve_dir(name: str) -> str | None:
    if os.path.isdir(name):
        return name
    cache = os.environ.get("HF_HOME", os.path.expanduser("~/.cache/huggingface"))
    for org in ("sentence-transformers--", ""):
        hits = sorted(glob.glob(os.path.join(cache, "hub", f"models--{org}{name.replace('/', '--')}", "snapshots", "*")))
        if hits:
            return hits[-1]
    return None


class SentenceEncoder:
    def __init__(self, model_name_or_path: str = "all-MiniLM-L6-v2", device: str = "cuda:0", dtype: str = "bf16",
                 batch_size: int = 64, max_tokens: int | None = None):
        self.tokenizer = None
        self._device = torch.device(device)
        if model_name_or_path.startswith("procedural-minilm"):
            parts = model_name_or_path.split(":")
            self.arch = MiniLMArch.tiny() if parts[0].endswith("-tiny") else MiniLMArch()
            sd = procedural_minilm_state_dict(self.arch, int(parts[1]) if len(parts) > 1 else 0)
        else:
            d = _resolve_dir(model_name_or_path)
            if d is None:
                raise RuntimeError(f"sentence encoder checkpoint {model_name_or_path!r} not found locally (no network)")
            cfg = json.load(open(os.path.join(d, "config.json")))
            self.arch = MiniLMArch(hidden=cfg["hidden_size"], layers=cfg["num_hidden_layers"], heads=cfg["num_attention_heads"],
                                   ffn=cfg["intermediate_size"], vocab=cfg["vocab_size"],
                                   max_pos=cfg["max_position_embeddings"], eps=cfg.get("layer_norm_eps", 1e-12))
            st_cfg = os.path.join(d, "sentence_bert_config.json")
            if os.path.exists(st_cfg):
                self.arch.max_seq_length = int(json.load(open(st_cfg)).get("max_seq_length", 256))
            w = [p for p in (os.path.join(d, "model.safetensors"), os.path.join(d, "pytorch_model.bin")) if os.path.exists(p)]
            if not w:
                raise RuntimeError(f"no model.safetensors / pytorch_model.bin under {d}")
            sd = load_state_dict_file(w[0])
            from tokenizers import Tokenizer
            self.tokenizer = Tokenizer.from_file(os.path.join(d, "tokenizer.json"))
            self.tokenizer.no_padding()
            self.tokenizer.enable_truncation(self.arch.max_seq_length)
        self.batch_size = batch_size
        self.max_tokens = min(max_tokens or self.arch.max_seq_length, self.arch.max_pos)
        self.engine = TextEncoderEngine(self.arch, dtype=dtype, max_batch=batch_size, max_len=self.max_tokens, device=self._device)
        self.engine.load_state_dict(sd)

    # --- SentenceTransformer surface used by the reference
    def to(self, *args, **kwargs):
        return self

    def eval(self):
        return self

    @property
    def device(self):
        return self._device

    def get_sentence_embedding_dimension(self) -> int:
        return self.arch.hidden

    def tokenize_ids(self, sentences: Sequence[str]) -> List[List[int]]:
        if self.tokenizer is None:
            raise RuntimeError("this encoder was built from procedural weights: no vocabulary, pass token ids to encode_ids()")
        return [e.ids for e in self.tokenizer.encode_batch(list(sentences))]

    @torch.no_grad()
    def encode_ids(self, rows: Sequence[Sequence[int]]) -> torch.Tensor:
        """Ragged WordPiece id rows (incl. [CLS]/[SEP]) -> fp32 [n, hidden] on the device, input order kept."""
        out = torch.empty((len(rows), self.arch.hidden), dtype=torch.float32, device=self._device)
        order = np.argsort([-len(r) for r in rows], kind="stable")           # longest first, like sentence-transformers
        for i in range(0, len(rows), self.batch_size):
            idx = order[i:i + self.batch_size]
            L = min(max(len(rows[j]) for j in idx), self.max_tokens)
            ids = np.full((len(idx), L), self.arch.pad, dtype=np.int32)
            lens = np.zeros(len(idx), dtype=np.int32)
            for k, j in enumerate(idx):
                r = list(rows[j])[:L]
                if not r:
                    raise ValueError("empty token row")
                ids[k, : len(r)] = r
                lens[k] = len(r)
            out[torch.as_tensor(idx, device=self._device)] = self.engine.embed(torch.from_numpy(ids), torch.from_numpy(lens))
        return out

    @torch.no_grad()
    def encode_caption_tokens(self, sequences: torch.Tensor, lengths: torch.Tensor, eos: int) -> torch.Tensor:
        """Embeddings straight from the captioner's token table (`generate` output: sequences [n, L] incl. the decoder
        BOS, lengths [n]) without detokenising: row = [CLS] + caption tokens + [SEP].  Valid when captioner and embedder
        share a WordPiece vocabulary - BLIP and all-MiniLM-L6-v2 both use bert-base-uncased's (BLIP adds two ids at the
        end, its BOS among them, which never appears inside a caption) - and equal to decode -> encode whenever WordPiece
        round-trips the caption text."""
        seq, lens = sequences.cpu().tolist(), lengths.cpu().tolist()
        rows = []
        for r, n in zip(seq, lens):
            body = [t for t in r[1:n] if t != eos]
            rows.append([self.arch.cls] + body + [self.arch.sep])
        return self.encode_ids(rows)

    def encode(self, sentences: Union[str, Sequence[str]], batch_size: int | None = None, convert_to_tensor: bool = False,
               convert_to_numpy: bool = True, normalize_embeddings: bool = False, **_):
        """all-MiniLM-L6-v2 ends in a Normalize module, so its embeddings are unit-norm whatever `normalize_embeddings` says."""
        single = isinstance(sentences, str)
        emb = self.encode_ids(self.tokenize_ids([sentences] if single else sentences))
        if single:
            emb = emb[0]
        if convert_to_tensor:
            return emb
        return emb.cpu().numpy() if convert_to_numpy else list(emb.cpu())
