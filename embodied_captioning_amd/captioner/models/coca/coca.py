"""CoCa captioner wrapper with the reference's wrapper shape (``captioner/models/coca/coca.py:19-33``):
``CoCa(cfg)`` with ``cfg.model_name`` / ``cfg.checkpoint_name``; ``forward(PIL.Image) -> {"text", "logits"}`` where
``logits`` is the list of per-step MinLength-processed logits of the still-active rows (``coca_model.py:312-313``).

Arithmetic runs in libcaptioner_hip.so (`CaptionerEngine` with a `CocaArch`); the decode loop is the reference's
``generate(generation_type='top_k')`` with top_k = 1 (coca.py:29), KV-cached instead of re-running the prefix.

model_name: ``coca_ViT-L-14`` (the reference's model_configs json) with ``checkpoint_name`` = path of an open_clip
state-dict file (``.pt`` / ``.bin`` / ``.safetensors``; pretrained *tags* need the network and raise like
``factory.py:309-314``), or ``procedural-coca:<seed>[:<eos_boost>]`` / ``procedural-coca-tiny:...`` (seeded weights).
"""
from __future__ import annotations

import dataclasses
import logging
import os
from typing import List, Sequence

import numpy as np
import torch

from ...captioning_predictor import CaptioningPredictor
from ....config import CocaArch
from ....engine import CaptionerEngine
from ....weights import load_state_dict_file, procedural_coca_state_dict

logger = logging.getLogger(__name__)


class CoCa(CaptioningPredictor):
    def __init__(self, cfg=None):
        super().__init__(cfg)
        name = cfg.model_name or "coca_ViT-L-14"
        self.batch_size = int(getattr(cfg, "batch_size", 8) or 8)
        # num_beams 1 (default) = the wrapper's call in the reference, generate(generation_type="top_k", top_k=1) (coca.py:29);
        # > 1 = the model's `_generate_beamsearch` with one beam group (coca_model.py:335-482; SURVEY config 5 asks beam 5)
        self.num_beams = int(getattr(cfg, "num_beams", 1) or 1)
        # optional: the model's beam GROUPS (its generate() defaults are 6 beams in 3 groups, coca_model.py:218-219)
        g = getattr(cfg, "num_beam_groups", None)
        self.num_beam_groups = int(g) if g else None
        if self.num_beam_groups and self.num_beams % self.num_beam_groups:
            raise ValueError(f"num_beams ({self.num_beams}) must be a multiple of num_beam_groups ({self.num_beam_groups})")
        # the other generation options of the reference's model (coca_model.py:205-224): off, or rejected by name
        from ...generation_options import reject_unsupported_generation_options
        self.generation_options = {k: getattr(cfg, k) for k in ("generation_type", "top_k", "top_p", "temperature", "repetition_penalty")
                                   if getattr(cfg, k, None) is not None}
        reject_unsupported_generation_options(self.generation_options, "CoCa(cfg)")
        if self.generation_options.get("generation_type") == "beam_search" and self.num_beams < 2:
            raise ValueError("CoCa(cfg): generation_type='beam_search' needs num_beams >= 2 (the reference's default is 6 in 3 groups)")
        dtype = getattr(cfg, "dtype", None) or "f32s"      # fp32-grade default (token-identical to the fp32 restatement); "bf16" is ~2x faster
        self._device = torch.device(getattr(cfg, "device", "cuda:0") or "cuda:0")
        self.tokenizer = None
        # optional config key `image_size` = open_clip's force_image_size (factory.py:243-245): 224 (pretrained) or e.g. 336;
        # the checkpoint's position table is resized at load (coca_weights.resize_visual_pos_embed)
        image_size = int(getattr(cfg, "image_size", 0) or 0)
        if name.startswith("procedural-coca"):
            parts = name.split(":")
            seed = int(parts[1]) if len(parts) > 1 else 0
            boost = float(parts[2]) if len(parts) > 2 else 0.0
            self.arch = CocaArch.tiny() if parts[0] == "procedural-coca-tiny" else CocaArch()
            if image_size:
                self.arch = dataclasses.replace(self.arch, image_size=image_size)
            sd = procedural_coca_state_dict(self.arch, seed, eos_boost=boost)
        else:
            if name != "coca_ViT-L-14":
                raise RuntimeError(f"Model config for {name} not found.")               # factory.py:231-233
            ck = getattr(cfg, "checkpoint_name", None)
            if not ck or not os.path.exists(ck):
                raise RuntimeError(f"Pretrained weights ({ck}) not found for model {name}.")   # factory.py:309-314
            self.arch = CocaArch(image_size=image_size) if image_size else CocaArch()
            sd = load_state_dict_file(ck)
        # detokeniser (coca.py:30 `open_clip.decode`): open_clip itself when it is installed, else the CLIP BPE vocabulary read
        # from files next to the checkpoint (cfg.tokenizer_dir / the checkpoint's directory: vocab.json, merges.txt or
        # bpe_simple_vocab_16e6.txt[.gz]) through captioner/clip_bpe.py; with neither, ids are returned as text
        self.bpe = None
        try:
            import open_clip
            self.tokenizer = open_clip
        except Exception:  # noqa: BLE001
            from ...clip_bpe import ClipBpeDecoder
            self.bpe = ClipBpeDecoder.find(getattr(cfg, "tokenizer_dir", None), getattr(cfg, "checkpoint_name", None))
            if self.bpe is None and not name.startswith("procedural-coca"):
                logger.warning("no CLIP BPE vocabulary (vocab.json / merges.txt / bpe_simple_vocab_16e6.txt.gz) next to the "
                               "checkpoint and open_clip is not installed: captions are returned as space-separated token ids")
        self.engine = CaptionerEngine(self.arch, dtype=dtype, max_batch=self.batch_size, max_beams=self.num_beams,
                                      max_len=self.arch.seq_len, device=self._device,
                                      cross_cache=self._cross_cache_for(cfg, sd, dtype))
        # HF generate stops once every caption has its EOS; look every few steps (cfg early_exit_poll, 0 = never)
        poll = getattr(cfg, "early_exit_poll", None)
        self.engine.set_early_exit(4 if poll is None else int(poll))
        self.engine.load_state_dict(sd)
        self.strict_range = bool(getattr(cfg, "strict_range", False))
        self.device_resize = getattr(cfg, "device_resize", None) is not False
        # cfg.streams > 1 (as for BLIP): the micro-batches of one generate_batch call rotate over that many engines / HIP streams on
        # ONE copy of the weights (engine.EnginePool), and cfg.coalesce_rows (IMAGES per pass; None = 4 micro-batches, at most 512
        # images) lets the pool merge consecutive micro-batches into larger passes and split the outputs back - every image's
        # beams are its own and the kernels' sums do not depend on the batch, so sequences, lengths and beam scores are those of
        # the unmerged call (tests/test_coca_gpu.py).  Config 5 (ViT-L/14 at 336, beam 5, 128 images per micro-batch, bf16): 1 068
        # captions/s on three streams, 1 147 with passes of 512 images.
        self.pool = None
        n_streams = int(getattr(cfg, "streams", 1) or 1)
        cr = getattr(cfg, "coalesce_rows", None)
        self.coalesce_rows = 0
        if n_streams > 1:
            self.coalesce_rows = min(4 * self.batch_size, 512) if cr is None else max(0, int(cr))
            if self.coalesce_rows <= self.batch_size:
                self.coalesce_rows = 0
            from ....engine import EnginePool
            self.pool = EnginePool(self.arch, n=n_streams, device=self._device, dtype=dtype, max_batch=max(self.batch_size, self.coalesce_rows),
                                   max_beams=self.num_beams, max_len=self.arch.seq_len, weights_of=self.engine, cross_cache=self.engine.cross_cache)
            self.pool.set_early_exit(4 if poll is None else int(poll))
        elif cr:
            logger.warning("captioner.coalesce_rows is the engine pool's dynamic batching: it needs captioner.streams > 1 - ignored")

    @property
    def device(self):
        return self._device

    def to(self, *args, **kwargs):
        return self

    @property
    def shorter_side_resize_size(self) -> int:
        """Side of the transform's shorter-side resize + centre crop: callers that hold uint8 frames may do it on the device
        (preprocess.crop_resize_u8(..., center_crop=True)) and pass uint8 [n, S, S, 3]."""
        return self.arch.image_size

    def preprocess(self, images) -> torch.Tensor:
        """open_clip `image_transform(is_train=False)`: bicubic resize of the shorter side to the model size, centre
        crop, RGB; rescale + OPENAI mean/std are fused into the patch-gather kernel (uint8 in)."""
        from PIL import Image
        S = self.arch.image_size
        if isinstance(images, torch.Tensor):
            return images if images.dim() == 4 else images[None]
        if isinstance(images, Image.Image):
            images = [images]
        if getattr(self, "device_resize", True):      # shorter-side bicubic resize + centre crop on the device, bit-exact with Pillow
            from ....preprocess import resize_u8_list
            return resize_u8_list([np.asarray(im.convert("RGB")) for im in images], S, device=self._device, center_crop=True)
        frames = []
        from ....preprocess import shorter_side_geometry
        for im in images:
            im = im.convert("RGB")
            nw, nh, left, top = shorter_side_geometry(im.size[0], im.size[1], S)
            im = im.resize((nw, nh), resample=Image.BICUBIC)
            frames.append(np.asarray(im.crop((left, top, left + S, top + S))))
        return torch.from_numpy(np.stack(frames))

    def decode(self, ids: Sequence[int]) -> str:
        ids = [int(i) for i in ids]
        if self.tokenizer is not None:
            text = self.tokenizer.decode(torch.tensor(ids))
            return text.split("<end_of_text>")[0].replace("<start_of_text>", "")       # coca.py:30
        if self.bpe is not None:
            return self.bpe.caption(ids)
        a = self.arch
        return " ".join(str(i) for i in ids if i not in (a.sot, a.eos, a.pad))

    @torch.no_grad()
    def generate(self, image, **options) -> torch.Tensor:
        """The reference model's `generate(image, ...)` entry (coca_model.py:205-224) on preprocessed frames: token ids [B, seq_len].
        generation_type "top_k" with top_k=1 (the reference wrapper's call, coca.py:29) or "beam_search" with num_beams /
        num_beam_groups; every other option must be at its neutral value or the call raises ValueError naming it."""
        from ...generation_options import reject_unsupported_generation_options
        opts = dict(options)
        gt = opts.pop("generation_type", "top_k" if self.num_beams == 1 else "beam_search")
        beams = int(opts.pop("num_beams", self.num_beams if gt == "beam_search" else 1))
        groups = opts.pop("num_beam_groups", self.num_beam_groups if gt == "beam_search" else None)
        seq_len = int(opts.pop("seq_len", self.arch.seq_len))
        reject_unsupported_generation_options({"generation_type": gt, **opts}, "CoCa.generate")
        extra = sorted(set(opts) - {"top_k", "top_p", "temperature", "repetition_penalty", "stopping_criteria", "text", "max_seq_len",
                                     "pad_token_id", "eos_token_id", "sot_token_id", "min_seq_len", "fixed_output_length"})
        if extra:
            raise TypeError(f"CoCa.generate() got unexpected keyword argument(s) {extra}")
        # options the engine was BUILT with (cap_create: arch.min_seq_len, the token ids, the 77-token context of the reference's
        # window `text[:, -max_seq_len:]`, coca_model.py:295): accepted at the value in force, refused - by name - at any other
        a = self.arch
        built = {"min_seq_len": a.min_seq_len, "eos_token_id": a.eos, "pad_token_id": a.pad, "sot_token_id": a.sot,
                 "fixed_output_length": False}
        differ = [f"{k}={opts[k]!r} (this captioner was built with {v!r})" for k, v in built.items() if k in opts and opts[k] is not None and opts[k] != v]
        if "max_seq_len" in opts and opts["max_seq_len"] is not None and int(opts["max_seq_len"]) < seq_len:
            differ.append(f"max_seq_len={opts['max_seq_len']!r} (a context window shorter than seq_len={seq_len} is not implemented)")
        if differ:
            raise ValueError("CoCa.generate: " + "; ".join(differ) + " - set it in the captioner's configuration instead of per call")
        if gt == "top_k":
            beams, groups = 1, None
        if beams > self.engine.max_beams or seq_len > self.engine.max_len:
            raise ValueError(f"CoCa.generate: num_beams={beams} / seq_len={seq_len} exceed what this captioner was built for "
                             f"(cfg.num_beams={self.engine.max_beams}, seq_len={self.engine.max_len})")
        px = self.preprocess(image)
        out = self.engine.generate(px.to(self._device), num_beams=beams, max_length=seq_len, num_beam_groups=groups)
        return out["sequences"]

    @torch.no_grad()
    def generate_batch(self, images) -> dict:
        kw = dict(num_beams=self.num_beams, max_length=self.arch.seq_len, num_beam_groups=self.num_beam_groups)
        rnd = len(self.pool) * max(self.batch_size, self.coalesce_rows) if self.pool is not None else 0
        if self.pool is not None and isinstance(images, (list, tuple)) and len(images) > rnd:
            # a long list of PIL crops: in rounds of one pass per engine, the next round preprocessed by a helper thread meanwhile
            from concurrent.futures import ThreadPoolExecutor
            groups = [images[i:i + rnd] for i in range(0, len(images), rnd)]
            outs = []
            with ThreadPoolExecutor(max_workers=1) as ex:
                nxt = ex.submit(self.preprocess, groups[0])
                for g in range(len(groups)):
                    px = nxt.result()
                    if g + 1 < len(groups):
                        nxt = ex.submit(self.preprocess, groups[g + 1])
                    chunks = [px[i:i + self.batch_size].to(self._device) for i in range(0, px.shape[0], self.batch_size)]
                    outs += self.pool.generate_many(chunks, threads=True, coalesce_rows=self.coalesce_rows, **kw)
        else:
            px = self.preprocess(images)
            chunks = [px[i:i + self.batch_size].to(self._device) for i in range(0, px.shape[0], self.batch_size)]
            if self.pool is not None and len(chunks) > 1:
                outs = self.pool.generate_many(chunks, threads=True, coalesce_rows=self.coalesce_rows, **kw)
            else:
                outs = [self.engine.generate(c, **kw) for c in chunks]
        seq, ln = torch.cat([o["sequences"] for o in outs]).cpu(), torch.cat([o["lengths"] for o in outs]).cpu()
        self._range_tick()
        res = {"texts": [self.decode(r[:n]) for r, n in zip(seq.tolist(), ln.tolist())], "sequences": seq, "lengths": ln}
        if all("sequences_scores" in o for o in outs):
            res["scores"] = torch.cat([o["sequences_scores"] for o in outs]).cpu()
        return res

    @torch.no_grad()
    def forward(self, inputs):
        a = self.arch
        px = self.preprocess(inputs)[:1]
        out = self.engine.generate(px.to(self._device), max_length=a.seq_len, output_logits=True)
        n = int(out["lengths"][0])
        ids = out["sequences"][0, :n].tolist()
        self._range_tick()
        steps: List[torch.Tensor] = []
        for t in range(n - 1):                       # one entry per generated token, like the reference's loop
            lg = out["logits"][t].clone()
            if t + 1 < a.min_seq_len:
                lg[:, a.eos] = float("-inf")         # MinLengthLogitsProcessor applied before the logits are recorded
            steps.append(lg)
        self.outputs = {"text": self.decode(ids), "logits": steps}
        return self.outputs
