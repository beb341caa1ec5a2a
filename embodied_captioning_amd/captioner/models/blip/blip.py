"""BLIP captioner wrapper with the reference's wrapper shape (``captioner/models/blip2/blip2.py:16-29``,
``captioner/models/coca/coca.py:19-33``): ``BLIP(cfg)`` with ``cfg.model_name`` / ``cfg.checkpoint_name``;
``forward(PIL.Image) -> {"text": str, "logits": tuple of per-step fp32 [n_rows, vocab]}`` also kept in ``self.outputs``.

All arithmetic runs in libcaptioner_hip.so (``CaptionerEngine``); this file only resolves the checkpoint, resizes /
uploads frames and detokenises.

model_name forms:
  * a local HF directory or a hub id present in the offline HF cache (``config.json`` + ``model.safetensors`` /
    ``pytorch_model.bin`` + tokenizer files) - e.g. ``Salesforce/blip-image-captioning-base``;
  * ``procedural:<seed>[:<eos_boost>]`` - seeded random weights of BLIP-base shape at 224x224 (what tests/bench use:
    no checkpoint exists offline); ``procedural-tiny:<seed>`` for the fixture-sized architecture.
checkpoint_name (optional): a ``torch.save({'model': state_dict})`` / state-dict / safetensors file that overrides the
weights (reference: ``utils/predictor_utils.py:182-185``).
"""
from __future__ import annotations

import logging
from typing import List, Sequence

import numpy as np
import torch

from ...captioning_predictor import CaptioningPredictor
from ....config import BlipArch
from ....engine import CaptionerEngine
from ....weights import (load_hf_blip_checkpoint, load_state_dict_file, procedural_blip_state_dict, resolve_hf_dir,
                         BLIP_TIED)

logger = logging.getLogger(__name__)


class BLIP(CaptioningPredictor):
    def __init__(self, cfg=None):
        super().__init__(cfg)
        name = cfg.model_name or "Salesforce/blip-image-captioning-base"
        self.num_beams = int(getattr(cfg, "num_beams", 1) or 1)
        self.max_length = int(getattr(cfg, "max_length", 20) or 20)
        self.batch_size = int(getattr(cfg, "batch_size", 8) or 8)
        # default arithmetic: split-fp16 GEMMs ("f32s") - token-identical to the reference's fp32 CPU path; "bf16" is ~2x
        # faster and leaves the reference's token path at near-ties
        dtype = getattr(cfg, "dtype", None) or "f32s"
        self._device = torch.device(getattr(cfg, "device", "cuda:0") or "cuda:0")
        self.tokenizer = None
        if name.startswith("procedural"):
            parts = name.split(":")
            seed = int(parts[1]) if len(parts) > 1 else 0
            boost = float(parts[2]) if len(parts) > 2 else 0.0
            self.arch = BlipArch.tiny() if parts[0] == "procedural-tiny" else BlipArch()
            sd = procedural_blip_state_dict(self.arch, seed, eos_boost=boost)
        else:
            model_dir = resolve_hf_dir(name)
            if model_dir is None:
                raise RuntimeError(f"Pretrained BLIP checkpoint '{name}' not found locally (offline); pass a directory "
                                   f"with config.json + model.safetensors, or 'procedural:<seed>'")
            self.arch, sd = load_hf_blip_checkpoint(model_dir)
            try:
                from transformers import AutoTokenizer
                self.tokenizer = AutoTokenizer.from_pretrained(model_dir)
            except Exception as e:  # noqa: BLE001
                logger.warning("no tokenizer under %s (%s): captions are returned as space-separated token ids", model_dir, e)
        if getattr(cfg, "checkpoint_name", None):
            over = load_state_dict_file(cfg.checkpoint_name)
            for dst, src in BLIP_TIED.items():
                if dst not in over and src in over:
                    over[dst] = over[src]
            sd.update(over)
            logger.info("Captioner model checkpoint loaded successfully from %s", cfg.checkpoint_name)
        self.engine = CaptionerEngine(self.arch, dtype=dtype, max_batch=self.batch_size, max_beams=self.num_beams,
                                      max_len=self.max_length, device=self._device,
                                      cross_cache=self._cross_cache_for(cfg, sd, dtype))
        # HF generate stops once every caption has its EOS; look every few steps (cfg early_exit_poll, 0 = never)
        poll = getattr(cfg, "early_exit_poll", None)
        poll = 4 if poll is None else int(poll)
        self.engine.set_early_exit(poll)
        self.engine.load_state_dict(sd)
        self.strict_range = bool(getattr(cfg, "strict_range", False))
        self.device_resize = getattr(cfg, "device_resize", None) is not False
        # cfg.streams > 1: micro-batches of one generate_batch call rotate over that many engines / HIP streams and overlap
        # (engine.EnginePool; same captions; one arena per engine, ONE copy of the weights: the pool's engines attach to
        # this engine's weight store)
        self.pool = None
        n_streams = int(getattr(cfg, "streams", 1) or 1)
        # cfg.coalesce_rows (with streams > 1): dynamic batching - the pool merges consecutive micro-batches of one
        # generate_batch call into passes of at most that many rows (EnginePool.generate_many(coalesce_rows=); a frame decodes
        # to the same bits alone, in its micro-batch and in a merged pass, so the captions are those of the unmerged call and
        # the decode chain's per-launch costs are paid once per pass).  None = on: four micro-batches, at most 1024 rows (the
        # arenas of the pool's engines are sized for it); 0 = every micro-batch its own pass.
        cr = getattr(cfg, "coalesce_rows", None)
        self.coalesce_rows = 0
        if n_streams > 1:
            self.coalesce_rows = min(4 * self.batch_size, 1024) if cr is None else max(0, int(cr))
            if self.coalesce_rows <= self.batch_size:
                self.coalesce_rows = 0
            from ....engine import EnginePool
            self.pool = EnginePool(self.arch, n=n_streams, device=self._device, dtype=dtype,
                                   max_batch=max(self.batch_size, self.coalesce_rows),
                                   max_beams=getattr(self, "num_beams", 1), max_len=self.engine.max_len, weights_of=self.engine,
                                   cross_cache=self.engine.cross_cache)
            self.pool.set_early_exit(poll)
        elif cr:
            logger.warning("captioner.coalesce_rows is the engine pool's dynamic batching: it needs captioner.streams > 1 - ignored")

    # nn.Module surface the callers use; weights live in the engine, so .to() only re-targets host-side tensors
    @property
    def device(self):
        return self._device

    @property
    def direct_resize_size(self) -> int:
        """Side of the square the processor resizes every image to, with no aspect handling (HF BlipImageProcessor): callers
        that hold uint8 frames may resize crops on the device (preprocess.crop_resize_u8) and pass uint8 [n, S, S, 3]."""
        return self.arch.image_size

    def to(self, *args, **kwargs):
        return self

    # ------------------------------------------------------------------------------------------ preprocessing
    def preprocess(self, images) -> torch.Tensor:
        """PIL image(s) / uint8 [B,H,W,3] / float [B,3,S,S] -> what the engine takes.  Resize = bicubic to the model's
        square input (HF `BlipImageProcessor`: HF:models/blip/image_processing_pil_blip.py:22-31); rescale and
        OPENAI-CLIP normalisation are fused into the patch-gather kernel for uint8 input."""
        S = self.arch.image_size
        if isinstance(images, torch.Tensor):
            t = images
            if t.dtype == torch.uint8:
                if t.dim() == 3:
                    t = t[None]
                if t.shape[1] != S or t.shape[2] != S:         # Pillow's bicubic, bit-exact, on the device
                    from ....preprocess import crop_resize_u8
                    H, W = int(t.shape[1]), int(t.shape[2])
                    t = torch.cat([crop_resize_u8(f, [(0, 0, W, H)], S, device=self._device) for f in t])
                return t
            return t if t.dim() == 4 else t[None]
        from PIL import Image
        if isinstance(images, Image.Image):
            images = [images]
        if getattr(self, "device_resize", True):
            # Pillow's bicubic resize on the device, bit-exact (csrc/preprocess.hip; tests/test_preprocess_gpu.py): one small upload and
            # two launches per image - 256 crops of 40-400 px: 26 ms against 130 ms of host PIL on one core (tools/pil_list_bench.py)
            from ....preprocess import resize_u8_list
            return resize_u8_list([np.asarray(im.convert("RGB")) for im in images], S, device=self._device)
        frames = [np.asarray(im.convert("RGB").resize((S, S), resample=Image.BICUBIC)) for im in images]
        return torch.from_numpy(np.stack(frames))

    def decode(self, ids: Sequence[int]) -> str:
        ids = [int(i) for i in ids]
        if self.tokenizer is not None:
            return self.tokenizer.decode(ids, skip_special_tokens=True).strip()
        a = self.arch
        return " ".join(str(i) for i in ids if i not in (a.bos, a.eos, a.pad))

    # ------------------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def generate_batch(self, images, output_logits: bool = False) -> dict:
        """Batched extension: any number of frames -> {"texts": [str], "sequences": int32 [N, L], "lengths", "scores"}."""
        texts: List[str] = []
        seqs, lens, scores, logits = [], [], [], []
        pool = getattr(self, "pool", None)
        # a long list of PIL crops on a pool: in rounds of one pass per engine, the next round's crops are preprocessed (host
        # `Image.convert` + pack, upload, device resize) by a helper thread while the current round generates
        rnd = len(pool) * max(self.batch_size, self.coalesce_rows) if pool is not None else 0
        if (pool is not None and not output_logits and not isinstance(images, torch.Tensor) and isinstance(images, (list, tuple))
                and len(images) > rnd):
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
                    outs += self.pool.generate_many(chunks, threads=True, coalesce_rows=self.coalesce_rows, num_beams=self.num_beams,
                                                    max_length=self.max_length)
        else:
            px = self.preprocess(images)
            chunks = [px[i:i + self.batch_size].to(self._device) for i in range(0, px.shape[0], self.batch_size)]
            if pool is not None and len(chunks) > 1 and not output_logits:
                outs = self.pool.generate_many(chunks, threads=True, coalesce_rows=self.coalesce_rows, num_beams=self.num_beams,
                                               max_length=self.max_length)
            else:
                outs = [self.engine.generate(c, num_beams=self.num_beams, max_length=self.max_length, output_logits=output_logits)
                        for c in chunks]
        for out in outs:
            seqs.append(out["sequences"]); lens.append(out["lengths"])
            if "sequences_scores" in out:
                scores.append(out["sequences_scores"])
            if output_logits:
                logits.append(out["logits"])
        seq = torch.cat(seqs).cpu()
        ln = torch.cat(lens).cpu()
        self._range_tick()
        for r, n in zip(seq.tolist(), ln.tolist()):
            texts.append(self.decode(r[:n]))
        res = {"texts": texts, "sequences": seq, "lengths": ln}
        if scores:
            res["scores"] = torch.cat(scores).cpu()
        if output_logits:
            res["logits"] = logits
        return res

    @torch.no_grad()
    def forward(self, inputs):
        px = self.preprocess(inputs)[:1]
        out = self.engine.generate(px.to(self._device), num_beams=self.num_beams, max_length=self.max_length,
                                   output_logits=True)
        n = int(out["lengths"][0])
        ids = out["sequences"][0, :n].tolist()
        self._range_tick()
        # new objects every call: callers keep references to outputs["logits"] across calls
        # (reference generate_pseudo_caption_from_file.py:152)
        self.outputs = {"text": self.decode(ids),
                        "logits": tuple(out["logits"][t] for t in range(max(n - 1, 1)))}
        return self.outputs
