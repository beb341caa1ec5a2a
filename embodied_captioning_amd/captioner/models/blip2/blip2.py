"""`BLIP2(cfg)` entry of the plugin factory - reference ``captioner/models/blip2/blip2.py:16-29``:
`Blip2ForConditionalGeneration.from_pretrained(cfg.model_name)` + `generate(**inputs, output_logits=True,
return_dict_in_generate=True)` + `processor.batch_decode(..., skip_special_tokens=True)[0].strip()`.

Arithmetic: libcaptioner_hip.so (`CaptionerEngine` with a `Blip2Arch`: ViT-g/14, Q-Former, OPT decoder with a K/V cache).
``model_name``:
  * a local HF directory / cached hub id (``Salesforce/blip2-opt-2.7b``): config.json, model.safetensors or the sharded
    ``model-0000x-of-0000y.safetensors`` + index, tokenizer files;
  * ``procedural-blip2[-tiny|-small][:seed[:eos_boost]]`` - seeded weights of the published / the fixture geometry (no vocabulary:
    captions come back as space-separated token ids);
  * a BLIP(-base) captioning checkpoint selected with ``arch_name: blip2`` still runs through the BLIP path (compatibility
    with configs written before this class existed).
The reference loads 8-bit weights / fp16 activations (``blip2.py:19-22``: ``load_in_8bit=True, torch_dtype=float16``).  Policy
here: the checkpoint's tensors are taken at the precision they are STORED in - an fp16-stored
checkpoint enters fp32 / the split format exactly (every fp16 value is a hi half with a zero lo half) - and the arithmetic is
``dtype``: "f32s" by default (fp32-grade split-fp16 GEMMs, token-identical to HF's fp32 CPU model on the committed golden,
tests/test_blip2_gpu.py - the parity clause's mode), "bf16" (halves the decoder's weight stream, about 2x the captions/s,
near-tie token flips), "f32" (exact).  ``load_in_8bit: true`` (the reference's own mode) selects bf16 activations with the
weight half of bitsandbytes' LLM.int8: every Linear HF would convert is row-quantised to int8 exactly as bitsandbytes stores it
(q = rint(w * 127 / absmax(row)), scale = absmax / 127), the OPT decoder layers' weights stay bytes in HBM and the decode GEMMs
stream them (half the bytes of a bf16 step); the activation half (int8 rows with fp16 outlier columns) is not restated -
activations stay bf16, which is closer to the fp16 model than LLM.int8 itself.  Parity of this mode is UNPINNED (bitsandbytes is
absent here): tests establish HIP = the restatement with the quantised weights.  ``load_in_4bit`` is rejected by name.
``checkpoint_name`` (optional): a PEFT LoRA adapter directory (``adapter_config.json`` + ``adapter_model.safetensors``) - what
the reference's fine-tuned BLIP-2 is (``scripts/evaluate_finetuned_model.py:147-148``: ``PeftModel.from_pretrained``) - merged
into the base weights at load (weights.merge_peft_lora), or a state-dict file that overrides tensors of the base checkpoint.
"""
from __future__ import annotations

import glob
import json
import logging
import os
from typing import List, Sequence

import torch

from ...captioning_predictor import CaptioningPredictor
from ....config import Blip2Arch
from ....engine import CaptionerEngine
from ....weights import is_peft_adapter, load_state_dict_file, merge_peft_lora, procedural_blip2_state_dict, resolve_hf_dir
from ..blip.blip import BLIP

logger = logging.getLogger(__name__)


def blip2_arch_from_hf_config(cfg: dict) -> Blip2Arch:
    v, q, t = cfg.get("vision_config", {}), cfg.get("qformer_config", {}), cfg.get("text_config", {})
    if t.get("model_type", "opt") != "opt":
        raise RuntimeError(f"BLIP-2 language model '{t.get('model_type')}' is not supported (OPT only)")
    if t.get("word_embed_proj_dim", t.get("hidden_size", 768)) != t.get("hidden_size", 768) or not t.get("do_layer_norm_before", True):
        raise RuntimeError("OPT variants with project_in/out or post-LayerNorm (opt-350m) are not supported")
    a = Blip2Arch()
    a.image_size, a.patch_size = v.get("image_size", 224), v.get("patch_size", 14)
    a.v_hidden, a.v_layers, a.v_heads = v.get("hidden_size", 1408), v.get("num_hidden_layers", 39), v.get("num_attention_heads", 16)
    a.v_mlp, a.v_eps = v.get("intermediate_size", 6144), v.get("layer_norm_eps", 1e-6)
    a.q_hidden, a.q_layers, a.q_heads = q.get("hidden_size", 768), q.get("num_hidden_layers", 12), q.get("num_attention_heads", 12)
    a.q_ffn, a.q_cross_freq, a.q_eps = q.get("intermediate_size", 3072), q.get("cross_attention_frequency", 2), q.get("layer_norm_eps", 1e-12)
    a.num_query_tokens = cfg.get("num_query_tokens", 32)
    a.t_hidden, a.t_layers, a.t_heads = t.get("hidden_size", 768), t.get("num_hidden_layers", 12), t.get("num_attention_heads", 12)
    a.t_ffn, a.vocab, a.max_pos = t.get("ffn_dim", 3072), t.get("vocab_size", 50272), t.get("max_position_embeddings", 2048)
    a.bos, a.pad = t.get("bos_token_id", 2), t.get("pad_token_id", 1)
    a.eos = t.get("eos_token_id", 2)
    a.image_token = cfg.get("image_token_index") or cfg.get("image_token_id") or a.vocab - 1
    return a


def load_hf_blip2_checkpoint(model_dir: str):
    """config.json (+ generation_config.json for the EOS id generate() uses) + single-file or sharded safetensors."""
    cfg = json.load(open(os.path.join(model_dir, "config.json")))
    arch = blip2_arch_from_hf_config(cfg)
    gen = os.path.join(model_dir, "generation_config.json")
    if os.path.exists(gen):
        g = json.load(open(gen))
        eos = g.get("eos_token_id", arch.eos)
        arch.eos = eos[0] if isinstance(eos, list) else eos
        arch.bos, arch.pad = g.get("bos_token_id", arch.bos), g.get("pad_token_id", arch.pad)
    files = sorted(glob.glob(os.path.join(model_dir, "model*.safetensors"))) or sorted(glob.glob(os.path.join(model_dir, "pytorch_model*.bin")))
    if not files:
        raise RuntimeError(f"no model*.safetensors / pytorch_model*.bin under {model_dir}")
    sd = {}
    for f in files:
        sd.update(load_state_dict_file(f))
    if "language_model.lm_head.weight" not in sd:
        sd["language_model.lm_head.weight"] = sd["language_model.model.decoder.embed_tokens.weight"]
    return arch, sd


class BLIP2(BLIP):
    def __init__(self, cfg=None):
        name = cfg.model_name or "Salesforce/blip2-opt-2.7b"
        model_dir = None if name.startswith("procedural") else resolve_hf_dir(name)
        is_blip2 = name.startswith("procedural-blip2")
        if model_dir is not None:
            try:
                is_blip2 = json.load(open(os.path.join(model_dir, "config.json"))).get("model_type") == "blip-2"
            except OSError:
                pass
        if not is_blip2:
            if "blip2" in name.lower() and model_dir is None:
                raise RuntimeError(f"Pretrained BLIP-2 checkpoint '{name}' not found locally (offline)")
            super().__init__(cfg)                       # a BLIP captioning checkpoint under arch_name blip2
            return
        CaptioningPredictor.__init__(self, cfg)
        if getattr(cfg, "load_in_4bit", None):
            raise ValueError("BLIP2(cfg): load_in_4bit (bitsandbytes NF4) is not implemented by the MI355X captioner library - "
                             "use load_in_8bit (the reference's own mode, blip2.py:19-22) or drop the key")
        # the reference's load mode (blip2.py:19-22: load_in_8bit=True, torch_dtype=float16): int8 Linear weights as bitsandbytes
        # stores them, half-precision activations - here bf16; any other cfg.dtype with it is a contradiction
        self.load_in_8bit = bool(getattr(cfg, "load_in_8bit", None))
        if self.load_in_8bit and (getattr(cfg, "dtype", None) or "bf16") != "bf16":
            raise ValueError(f"BLIP2(cfg): load_in_8bit runs with half-precision activations (dtype 'bf16'); got dtype={cfg.dtype!r}")
        td = getattr(cfg, "torch_dtype", None)
        if td is not None and str(td).replace("torch.", "") not in ("float16", "half", "bfloat16", "float32", "float"):
            raise ValueError(f"BLIP2(cfg): torch_dtype={td!r} is not a floating type a checkpoint is stored in")
        self.num_beams = 1
        self.batch_size = int(getattr(cfg, "batch_size", 8) or 8)
        dtype = "bf16" if self.load_in_8bit else getattr(cfg, "dtype", None) or "f32s"   # parity-grade default: tokens identical to HF fp32 on the golden
        self._device = torch.device(getattr(cfg, "device", "cuda:0") or "cuda:0")
        self.tokenizer = None
        if model_dir is None:
            parts = name.split(":")
            self.arch = {"procedural-blip2-tiny": Blip2Arch.tiny, "procedural-blip2-small": Blip2Arch.small}.get(parts[0], Blip2Arch)()
            sd = procedural_blip2_state_dict(self.arch, int(parts[1]) if len(parts) > 1 else 0,
                                             eos_boost=float(parts[2]) if len(parts) > 2 else 0.0)
        else:
            self.arch, sd = load_hf_blip2_checkpoint(model_dir)
            try:
                from transformers import AutoTokenizer
                self.tokenizer = AutoTokenizer.from_pretrained(model_dir)
            except Exception as e:  # noqa: BLE001
                logger.warning("no tokenizer under %s (%s): captions are returned as space-separated token ids", model_dir, e)
        ck = getattr(cfg, "checkpoint_name", None)
        self.adapter_report = None
        if ck:
            if is_peft_adapter(ck):
                sd, self.adapter_report = merge_peft_lora(sd, ck)
                logger.info("PEFT LoRA adapter %s merged into %d modules", ck, self.adapter_report["merged"])
            else:
                sd.update(load_state_dict_file(ck))
                logger.info("Captioner model checkpoint loaded successfully from %s", ck)
        # the reference passes no length: HF then generates 20 new tokens; the optional config key `max_new_tokens` (HF's
        # name) overrides that.  The plugin's `max_length` key is BLIP's / CoCa's TOTAL length and is not read here.
        self.max_length = int(getattr(cfg, "max_new_tokens", 0) or self.arch.max_new_tokens)
        self.engine = CaptionerEngine(self.arch, dtype=dtype, max_batch=self.batch_size, max_beams=1, max_len=self.max_length,
                                      device=self._device, cross_cache=getattr(cfg, "cross_cache", None) or "auto",
                                      weight_int8=self.load_in_8bit)
        # HF generate stops once every caption has its EOS; look every few steps (cfg early_exit_poll, 0 = never)
        poll = getattr(cfg, "early_exit_poll", None)
        self.engine.set_early_exit(4 if poll is None else int(poll))
        self.engine.load_state_dict(sd)
        self.strict_range = bool(getattr(cfg, "strict_range", False))
        self.device_resize = getattr(cfg, "device_resize", None) is not False
        # cfg.streams > 1 (as for BLIP / CoCa): the micro-batches of one generate_batch call rotate over that many engines / HIP
        # streams on ONE copy of the weights (engine.EnginePool) - the decoder's weight-streaming launches of independent batches fill
        # each other's gaps (OPT-2.7b geometry, 32 frames per micro-batch, bf16: 374 captions/s on one stream, 516 on three).  The pool's
        # dynamic batching (cfg.coalesce_rows: crops per merged pass; None = 4 micro-batches, at most 64 crops) pays doubly here: a
        # decode step streams the weights once per PASS whatever its rows.  A crop's bits do not depend on the batch it is in - except
        # in the int8 mode across 4 crops per pass, where the prompt pass changes kernels (captioner.hip::kI8SkinnyPromptCrops: same
        # captions to bf16 rounding noise, not the same bits): with load_in_8bit and micro-batches of up to 4 crops the default is
        # therefore OFF (set coalesce_rows to opt in).
        self.pool = None
        n_streams = int(getattr(cfg, "streams", 1) or 1)
        cr = getattr(cfg, "coalesce_rows", None)
        self.coalesce_rows = 0
        if n_streams > 1:
            if cr is None:
                self.coalesce_rows = 0 if (self.load_in_8bit and self.batch_size <= 4) else min(4 * self.batch_size, 64)
            else:
                self.coalesce_rows = max(0, int(cr))
            if self.coalesce_rows <= self.batch_size:
                self.coalesce_rows = 0
            from ....engine import EnginePool
            self.pool = EnginePool(self.arch, n=n_streams, device=self._device, dtype=dtype, max_batch=max(self.batch_size, self.coalesce_rows),
                                   max_beams=1, max_len=self.max_length, weights_of=self.engine, cross_cache=self.engine.cross_cache,
                                   weight_int8=self.load_in_8bit)
            self.pool.set_early_exit(4 if poll is None else int(poll))
        elif cr:
            logger.warning("captioner.coalesce_rows is the engine pool's dynamic batching: it needs captioner.streams > 1 - ignored")

    def decode(self, ids: Sequence[int]) -> str:
        if not hasattr(self.arch, "num_query_tokens"):
            return super().decode(ids)
        ids = [int(i) for i in ids]
        if self.tokenizer is not None:
            return self.tokenizer.decode(ids, skip_special_tokens=True).strip()
        a = self.arch
        return " ".join(str(i) for i in ids if i not in (a.bos, a.eos, a.pad))

    @torch.no_grad()
    def forward(self, inputs):
        if not hasattr(self.arch, "num_query_tokens"):
            return super().forward(inputs)
        px = self.preprocess(inputs)[:1]
        out = self.engine.generate(px.to(self._device), max_length=self.max_length, output_logits=True)
        n = int(out["lengths"][0])
        self._range_tick()
        # HF's `logits` tuple has one entry per generated token; new objects every call (callers keep references)
        self.outputs = {"text": self.decode(out["sequences"][0, :n].tolist()), "logits": tuple(out["logits"][t] for t in range(n))}
        return self.outputs
