"""Host orchestration over the C ABI: builds a handle, streams checkpoint tensors in, exposes encode / generate on
torch CUDA tensors.  PyTorch here is plumbing only (device buffers, streams); all arithmetic runs in
libcaptioner_hip.so.  Mirrors what the reference's wrappers call on their torch models:
`model.generate(...)` (captioner/models/blip2/blip2.py:26, coca/coca.py:29) and `_encode_image` (coca_model.py:152-155).
"""
from __future__ import annotations

import ctypes as C
import json
import logging
from typing import Dict, List, Optional, Sequence

import torch

from . import _native as N
from .config import Blip2Arch, BlipArch, CocaArch, MiniLMArch

logger = logging.getLogger(__name__)

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

# "f32s" = CAP_F32_SPLIT: fp32 values carried into every GEMM as two fp16 halves, three fp16 MFMAs per product (fp32-grade
# products, token-identical to the fp32 reference on the goldens, several times faster than "f32"); BLIP, BLIP-2 and CoCa.
_DTYPES = {"f32": N.CAP_F32, "fp32": N.CAP_F32, "float32": N.CAP_F32, "bf16": N.CAP_BF16, "bfloat16": N.CAP_BF16,
           "f32s": N.CAP_F32_SPLIT, "split": N.CAP_F32_SPLIT, "f32_split": N.CAP_F32_SPLIT}


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class CaptionerEngine:
    """One handle = one model replica on one GPU, bound to torch's current stream of `device` at each call."""

    def __init__(self, arch: BlipArch, dtype: str = "bf16", max_batch: int = 8, max_beams: int = 1,
                 max_len: int = 20, device: str | torch.device = "cuda:0", share_weights_with: "CaptionerEngine | None" = None,
                 cross_cache: str = "auto", weight_int8: bool = False):
        """weight_int8 (BLIP-2, dtype "bf16" only): the reference's `load_in_8bit=True` (blip2.py:19-22) - the OPT decoder layers' Linear
        weights are kept as row-quantised int8 + fp32 row scales (bitsandbytes' storage) and streamed as bytes by the decode GEMMs,
        the vision tower's Linears and language_projection pass through the same quantiser at load; activations stay bf16.
        cross_cache: "auto" = the mode's own cross-attention K/V cache ("f32s": KV16 - int16 + one scale per 64-wide head row,
        the decode side's HBM stream at half the bytes; "bf16": bf16 rows; "f32": fp32 rows); "fp32" = fp32 rows in "f32s" too
        (`cross_cache_kind` tells what the handle uses).
        share_weights_with: an engine of the same model / dtype / GPU whose (read-only) weights this one uses instead of
        holding a copy - it gets its own arena only (cap_create_shared); load_state_dict through either is seen by both."""
        if not torch.cuda.is_available():
            raise N.CaptionerHipError("CaptionerEngine needs a GPU (torch.cuda.is_available() is False); "
                                      "there is no CPU fallback in the product path")
        self.lib = N.load_library()
        self.arch = arch
        self.dtype = dtype
        self.device = torch.device(device)
        self.max_batch, self.max_beams, self.max_len = max_batch, max_beams, max_len
        cfg = N.CapConfig()
        cfg.struct_size = C.sizeof(N.CapConfig)
        cfg.compute_dtype = _DTYPES[dtype]
        cfg.image_size, cfg.patch_size = arch.image_size, arch.patch_size
        self.is_coca = isinstance(arch, CocaArch)
        self.is_blip2 = isinstance(arch, Blip2Arch)
        if self.is_blip2:
            cfg.arch = 3
            cfg.v_hidden, cfg.v_layers, cfg.v_heads, cfg.v_mlp, cfg.v_eps = (arch.v_hidden, arch.v_layers, arch.v_heads,
                                                                             arch.v_mlp, arch.v_eps)
            cfg.q_hidden, cfg.q_layers, cfg.q_heads, cfg.q_ffn = arch.q_hidden, arch.q_layers, arch.q_heads, arch.q_ffn
            cfg.q_cross_freq, cfg.num_query_tokens, cfg.q_eps = arch.q_cross_freq, arch.num_query_tokens, arch.q_eps
            cfg.t_hidden, cfg.t_layers, cfg.t_heads, cfg.t_ffn = arch.t_hidden, arch.t_layers, arch.t_heads, arch.t_ffn
            cfg.vocab, cfg.max_pos, cfg.t_eps = arch.vocab, arch.max_pos, arch.t_eps
            cfg.bos, cfg.eos, cfg.pad = arch.bos, arch.eos, arch.pad
        elif self.is_coca:
            cfg.arch = 1
            cfg.v_hidden, cfg.v_layers, cfg.v_heads, cfg.v_mlp, cfg.v_eps = (arch.v_hidden, arch.v_layers, arch.v_heads,
                                                                             arch.v_mlp, arch.eps)
            cfg.t_hidden, cfg.t_layers, cfg.t_heads, cfg.t_ffn = arch.t_hidden, arch.t_layers, arch.t_heads, arch.t_ffn
            cfg.vocab, cfg.max_pos, cfg.t_eps = arch.vocab, arch.context_length + 1, arch.eps
            cfg.bos, cfg.eos, cfg.pad = arch.sot, arch.eos, arch.pad
            cfg.embed_dim, cfg.pool_queries, cfg.pool_heads = arch.embed_dim, arch.pool_queries, arch.pool_heads
            cfg.mm_layers, cfg.min_len = arch.mm_layers, arch.min_seq_len
        else:
            cfg.arch = 0
            cfg.v_hidden, cfg.v_layers, cfg.v_heads, cfg.v_mlp, cfg.v_eps = (arch.v_hidden, arch.v_layers, arch.v_heads,
                                                                             arch.v_mlp, arch.v_eps)
            cfg.t_hidden, cfg.t_layers, cfg.t_heads, cfg.t_ffn = arch.t_hidden, arch.t_layers, arch.t_heads, arch.t_ffn
            cfg.vocab, cfg.max_pos, cfg.t_eps = arch.vocab, arch.max_pos, arch.t_eps
            cfg.bos, cfg.eos, cfg.pad = arch.bos, arch.eos, arch.pad
        cfg.max_batch, cfg.max_beams, cfg.max_len = max_batch, max_beams, max_len
        if cross_cache not in ("auto", "fp32"):
            raise ValueError(f"cross_cache must be 'auto' or 'fp32', got {cross_cache!r}")
        cfg.cross_kv_fp32 = int(cross_cache == "fp32")
        self.cross_cache = cross_cache
        self.weight_int8 = bool(weight_int8)
        if self.weight_int8 and not (self.is_blip2 and dtype == "bf16"):
            raise ValueError("weight_int8 (load_in_8bit) is built for BLIP-2 with dtype 'bf16'")
        cfg.weight_int8 = int(self.weight_int8)
        for i in range(3):
            cfg.pix_mean[i] = OPENAI_CLIP_MEAN[i]
            cfg.pix_std[i] = OPENAI_CLIP_STD[i]
        self._h = C.c_void_p()
        self.shares_weights = share_weights_with is not None
        with torch.cuda.device(self.device):
            if share_weights_with is not None:
                N.check(self.lib.cap_create_shared(C.byref(cfg), share_weights_with._h, C.byref(self._h)), "cap_create_shared")
            else:
                N.check(self.lib.cap_create(C.byref(cfg), C.byref(self._h)), "cap_create")

    # ------------------------------------------------------------------------------------------ lifetime
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            with torch.cuda.device(self.device):          # cap_destroy synchronises and frees on the CURRENT device
                self.lib.cap_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    @property
    def device_bytes(self) -> int:
        return int(self.lib.cap_device_bytes(self._h))

    def saturations(self, reset: bool = False) -> int:
        """Split mode ("f32s") only has a finite range: values beyond +-65000 at a GEMM input are clamped - and counted.
        Returns the count on this engine's GPU since the last reset (0 = every value was inside the fp32-grade envelope);
        synchronises the device.  Weights outside the mode's range never get this far: load_state_dict raises."""
        with torch.cuda.device(self.device):
            n = int(self.lib.cap_g8_saturations(int(reset)))
        if n < 0:
            raise N.CaptionerHipError(f"cap_g8_saturations: {N.last_error()}")
        return n

    def set_early_exit(self, poll_steps: int) -> None:
        """Leave the decode loop once every caption is finished, as HF generate does; the device state is looked at every
        `poll_steps` steps (one stream synchronisation each).  0 = never (default): no host sync inside generate."""
        N.check(self.lib.cap_set_early_exit(self._h, int(poll_steps)), "cap_set_early_exit")

    @property
    def last_decode_steps(self) -> int:
        return int(self.lib.cap_last_decode_steps(self._h))

    @property
    def cross_cache_kind(self) -> str:
        """Layout of this handle's cross-attention K/V cache: "fp32", "bf16" or "kv16"."""
        return {0: "fp32", 1: "bf16", 2: "kv16"}[int(self.lib.cap_cross_cache_kind(self._h))]

    DECODE_PATHS = {"auto": 0, "batch": 1, "small": 2}

    def set_decode_path(self, path: str) -> None:
        """Kernels of the decode steps (BLIP, "f32s" / "bf16"): "auto" = the fused small-batch kernels for images x beams <= 16
        rows (6 launches per layer-step instead of 11; same bits), the batch kernels above; "batch" / "small" force one
        (forcing "small" makes generate fail for calls those kernels do not take)."""
        N.check(self.lib.cap_set_decode_path(self._h, self.DECODE_PATHS[path]), "cap_set_decode_path")

    def set_row_compaction(self, on: bool) -> None:
        """Greedy BLIP decode on the batch kernels: work on the rows of the captions still open only (default on; same tokens
        and lengths either way - off is for A/B runs and the equality tests)."""
        N.check(self.lib.cap_set_row_compaction(self._h, int(bool(on))), "cap_set_row_compaction")

    @property
    def last_row_compaction(self) -> bool:
        return int(self.lib.cap_last_row_compaction(self._h)) == 1

    @property
    def last_decode_path(self) -> str:
        return {0: "none", 1: "batch", 2: "small"}[int(self.lib.cap_last_decode_path(self._h))]

    # ------------------------------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> Dict[str, object]:
        """HF BLIP key names or an open_clip CoCa state dict (derived tensors are added here).
        Tensors the architecture does not store (tied heads, buffers) are skipped and reported; with strict=True missing
        ones raise.  A dict in which NOTHING matches always raises - it is a wrong checkpoint or wrong key prefixes, never a
        successful load.  Returns {"matched": n, "unknown": [names]}."""
        try:
            if self.is_coca and "derived.pool_q" not in sd:
                from .coca_weights import coca_library_state_dict
                sd = coca_library_state_dict(sd, self.arch)
            if getattr(self, "is_blip2", False) and "derived.qformer_x0" not in sd:
                # the Q-Former's input = LayerNorm(query_tokens) is a constant of the checkpoint: computed once here
                sd = dict(sd)
                q = sd["query_tokens"].float()[0]
                sd["derived.qformer_x0"] = torch.nn.functional.layer_norm(
                    q, (q.shape[-1],), sd["qformer.layernorm.weight"].float(), sd["qformer.layernorm.bias"].float(), self.arch.q_eps)
            if getattr(self, "weight_int8", False):
                from .weights import blip2_int8_host_names, int8_roundtrip
                sd = dict(sd)
                for k in blip2_int8_host_names(sd):
                    sd[k] = int8_roundtrip(sd[k])
        except KeyError as e:
            raise N.CaptionerHipError(f"state dict lacks {e}: this architecture derives tensors from the checkpoint at load "
                                      f"and needs the complete dict (keys with a 'model.' / 'module.' prefix? see "
                                      f"weights.strip_wrapper_prefixes)") from e
        # the KV16 cross-attention cache has ONE scale per 64-wide head row: a checkpoint whose key / value heads carry outlier
        # dimensions is refused here (measured bound: weights.KV16_MAX_HEAD_SPREAD), never served with coarser logits
        if not getattr(self, "_skip_kv16_guard", False) and hasattr(self.lib, "cap_cross_cache_kind") \
                and not isinstance(self, TextEncoderEngine) and self.cross_cache_kind == "kv16":
            from .weights import KV16_MAX_HEAD_SPREAD, cross_kv_head_spread
            spread = cross_kv_head_spread(sd)
            if spread > KV16_MAX_HEAD_SPREAD:
                raise N.CaptionerHipError(
                    f"the cross-attention key / value heads of this checkpoint have dimensions {spread:.1f}x their head's median "
                    f"magnitude; the split mode's KV16 cache (one scale per 64-wide head row) holds the 1e-3 logit bar up to "
                    f"{KV16_MAX_HEAD_SPREAD:.0f}x - create the engine with cross_cache='fp32' (captioner.cross_cache: fp32; the "
                    f"wrappers choose it themselves), or use dtype 'f32' / 'bf16'")
        matched, unknown = 0, []
        with torch.cuda.device(self.device):
            s = _stream_ptr(self.device)
            for name, t in sd.items():
                t = t.detach()
                if t.dtype != torch.float32:
                    t = t.float()
                t = t.contiguous()
                shape = (C.c_int64 * max(t.dim(), 1))(*(t.shape if t.dim() else (1,)))
                rc = self.lib.cap_load_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), int(t.is_cuda),
                                              max(t.dim(), 1), shape, C.c_void_p(s))
                if rc < 0:
                    raise N.CaptionerHipError(f"cap_load_weight({name}): {N.last_error()}")
                if rc == 0:
                    matched += 1
                else:
                    unknown.append(name)
            if len(sd) and not matched:
                raise N.CaptionerHipError(f"none of the {len(sd)} tensors is one this architecture stores "
                                          f"(first keys: {list(sd)[:3]}): wrong checkpoint or key prefix")
            missing = self.lib.cap_finalize_weights(self._h)
            if missing and strict:
                raise N.CaptionerHipError(f"checkpoint incomplete: {N.last_error()}")
        return {"matched": matched, "unknown": unknown}

    # ------------------------------------------------------------------------------------------ forward
    def _pixels(self, pixels: torch.Tensor):
        if pixels.device != self.device:
            pixels = pixels.to(self.device, non_blocking=True)
        a = self.arch
        if pixels.dtype == torch.uint8:
            if pixels.dim() != 4 or pixels.shape[1:] != (a.image_size, a.image_size, 3):
                raise ValueError(f"uint8 frames must be [B,{a.image_size},{a.image_size},3], got {tuple(pixels.shape)}")
            fmt = N.CAP_PIX_U8_NHWC
        else:
            if pixels.dim() != 4 or pixels.shape[1:] != (3, a.image_size, a.image_size):
                raise ValueError(f"float frames must be [B,3,{a.image_size},{a.image_size}], got {tuple(pixels.shape)}")
            pixels = pixels.float()
            fmt = N.CAP_PIX_F32_NCHW
        return pixels.contiguous(), fmt

    def encode(self, pixels: torch.Tensor) -> torch.Tensor:
        pixels, fmt = self._pixels(pixels)
        B = pixels.shape[0]
        shape = (B, self.arch.pool_queries, self.arch.embed_dim) if self.is_coca else (B, self.arch.n_tokens, self.arch.v_hidden)   # BLIP / BLIP-2: ViT image_embeds
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            N.check(self.lib.cap_encode(self._h, C.c_void_p(pixels.data_ptr()), fmt, B, C.c_void_p(out.data_ptr()),
                                        C.c_void_p(_stream_ptr(self.device))), "cap_encode")
        return out

    def generate(self, pixels: torch.Tensor, num_beams: int = 1, max_length: Optional[int] = None,
                 length_penalty: float = 1.0, output_logits: bool = False, num_beam_groups: Optional[int] = None,
                 **sampling_options) -> Dict[str, torch.Tensor]:
        """Returns device tensors: sequences int32 [B, max_length] (incl. BOS), lengths int32 [B],
        sequences_scores fp32 [B] (beams only), logits fp32 [max_length-1, B*num_beams, vocab] (optional).
        BLIP-2: max_length counts NEW tokens (HF max_new_tokens); sequences are those new tokens only (no image
        placeholders / BOS), logits [max_length, B, vocab].
        num_beam_groups (CoCa): the reference's `_generate_beamsearch` with that many beam groups (coca_model.py:335-482;
        its `generate()` defaults are 6 beams in 3 groups) - cap_generate_groups; no per-step logits in that mode.
        sampling_options: anything else a caller of the reference's / HF's `generate` may pass (top_p, top_k, temperature,
        repetition_penalty, do_sample, ...): accepted at their neutral values, rejected BY NAME otherwise - never ignored."""
        if sampling_options:
            from .captioner.generation_options import reject_unsupported_generation_options, _NEUTRAL
            known = set(_NEUTRAL) | {"generation_type", "top_k", "top_p"}
            unknown = sorted(set(sampling_options) - known)
            if unknown:
                raise TypeError(f"generate() got unexpected keyword argument(s) {unknown}")
            reject_unsupported_generation_options(sampling_options, "CaptionerEngine.generate")
        pixels, fmt = self._pixels(pixels)
        B = pixels.shape[0]
        L = max_length or self.max_len
        ids = torch.empty((B, L), dtype=torch.int32, device=self.device)
        lens = torch.empty((B,), dtype=torch.int32, device=self.device)
        scores = torch.zeros((B,), dtype=torch.float32, device=self.device)
        logits = None
        if output_logits:
            steps = L if getattr(self, "is_blip2", False) else L - 1
            # zeros, not empty: with early exit the steps after the last executed one are never written (callers see 0, not
            # stale memory); `last_decode_steps` tells how many steps ran
            logits = torch.zeros((steps, B * num_beams, self.arch.vocab), dtype=torch.float32, device=self.device)
        if num_beam_groups is not None:
            if output_logits:
                raise ValueError("per-step logits are not recorded by the group beam search")
            with torch.cuda.device(self.device):
                N.check(self.lib.cap_generate_groups(self._h, C.c_void_p(pixels.data_ptr()), fmt, B, num_beams, int(num_beam_groups), L,
                                                     C.c_float(length_penalty), C.c_void_p(ids.data_ptr()), C.c_void_p(lens.data_ptr()),
                                                     C.c_void_p(scores.data_ptr()), C.c_void_p(_stream_ptr(self.device))), "cap_generate_groups")
            return {"sequences": ids, "lengths": lens, "sequences_scores": scores}
        with torch.cuda.device(self.device):
            N.check(self.lib.cap_generate(self._h, C.c_void_p(pixels.data_ptr()), fmt, B, num_beams, L,
                                          C.c_float(length_penalty), C.c_void_p(ids.data_ptr()),
                                          C.c_void_p(lens.data_ptr()), C.c_void_p(scores.data_ptr()),
                                          C.c_void_p(logits.data_ptr() if logits is not None else 0),
                                          C.c_void_p(_stream_ptr(self.device))), "cap_generate")
        out = {"sequences": ids, "lengths": lens}
        if num_beams > 1:
            out["sequences_scores"] = scores
        if logits is not None:
            out["logits"] = logits
        return out

    # ------------------------------------------------------------------------------------------ profiling
    def profile(self, on: bool) -> None:
        N.check(self.lib.cap_profile_enable(self._h, int(on)), "cap_profile_enable")

    def profile_report(self) -> dict:
        buf = C.create_string_buffer(1 << 16)
        N.check(self.lib.cap_profile_report(self._h, buf, len(buf)), "cap_profile_report")
        return json.loads(buf.value.decode())


class EnginePool:
    """Several CaptionerEngines on their own streams: consecutive batches overlap.  Every engine has its own arena (activations,
    K/V caches); the weights exist ONCE - engines 1.. are created on engine 0's weight store (cap_create_shared).

    One `cap_generate` is a chain of ~2 800 dependent kernels; between two dependent kernels of one HIP queue the GPU
    idles for the dispatch hand-over, and the decode kernels' small grids leave CUs free.  Independent batches do not depend
    on each other, so kernels of another queue run in those gaps AND next to them: in the pooled rocprofv3 trace kernels of
    different streams do co-run ("Overlapping whole batches"), and a pooled step is SHORTER than the sum
    of one batch's kernel durations.  What bounds the pool is the MFMA-bound image side, which two batches cannot run faster
    than one after the other.  Every batch is computed by exactly the kernels of a single engine: results are the same bits.

        pool = EnginePool(arch, n=3, dtype="bf16", max_batch=256)
        pool.load_state_dict(sd)
        outs = pool.generate_many(batches)            # or: out = pool.submit(px) ... pool.join()
    """

    def __init__(self, arch, n: int = 2, device: str | torch.device = "cuda:0", engine_cls=None, weights_of=None, **engine_kw):
        """weights_of: an existing engine whose weight store ALL n engines of the pool attach to (nothing to load then)."""
        if n < 1:
            raise ValueError("EnginePool needs at least one engine")
        self.device = torch.device(device)
        engine_cls = engine_cls or CaptionerEngine           # TextEncoderEngine: submit(ids, lens, method="embed")
        first = engine_cls(arch, device=device, share_weights_with=weights_of, **engine_kw)
        self.engines = [first] + [engine_cls(arch, device=device, share_weights_with=first, **engine_kw) for _ in range(n - 1)]
        with torch.cuda.device(self.device):
            self.streams = [torch.cuda.Stream(self.device) for _ in range(n)]
        self.arch, self._next = arch, 0
        self.last_coalesce = None

    def __len__(self) -> int:
        return len(self.engines)

    def load_state_dict(self, sd, strict: bool = True):
        return self.engines[0].load_state_dict(sd, strict=strict)      # one weight store behind every engine of the pool

    def set_early_exit(self, poll_steps: int) -> None:
        for e in self.engines:
            e.set_early_exit(poll_steps)

    def set_decode_path(self, path: str) -> None:
        for e in self.engines:
            e.set_decode_path(path)

    def set_row_compaction(self, on: bool) -> None:
        for e in self.engines:
            e.set_row_compaction(on)

    def close(self) -> None:
        for e in self.engines:
            e.close()

    @property
    def device_bytes(self) -> int:
        return sum(e.device_bytes for e in self.engines)

    def run(self, n: int, *inputs: torch.Tensor, **kw):
        """The same batch n times, rotating over the engines (benchmarks); returns the last output after join()."""
        out = None
        for _ in range(n):
            out = self.submit(*inputs, **kw)
        self.join()
        return out

    def submit(self, *inputs: torch.Tensor, then=None, method: str = "generate", **kw):
        """Start one batch on the next engine / stream - `engine.<method>(*inputs, **kw)` - and return its output (or
        `then(out)`, run on that stream) at once; the tensors are valid for the caller's stream after `join()`.  The inputs
        may come from the caller's stream."""
        i, self._next = self._next, (self._next + 1) % len(self.engines)
        s = self.streams[i]
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            out = getattr(self.engines[i], method)(*inputs, **kw)
            for t in inputs:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(s)
            return then(out) if then is not None else out

    def join(self) -> None:
        """Make the caller's stream wait for everything submitted so far (device-side dependency, no host sync)."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    # outputs of `generate` whose leading dimension is the batch's rows (what a merged pass is split back by)
    _PER_ROW_OUTPUTS = ("sequences", "lengths", "sequences_scores")

    @staticmethod
    def coalesce_plan(rows: Sequence[int], n_engines: int, max_rows: int) -> List[List[int]]:
        """Dynamic batching plan: consecutive batches (their row counts in `rows`) merged into passes of at most `max_rows` rows -
        as few passes as the rows allow, but never fewer than engines (concurrency comes first) and rounded up to a multiple of
        the engine count (every engine runs the same number of passes); the rows spread over the passes as evenly as the order
        permits.  -> lists of batch indices, in order.  A batch larger than `max_rows` is a pass of its own."""
        nb = len(rows)
        if nb == 0:
            return []
        if max_rows <= 0:
            return [[i] for i in range(nb)]
        total = sum(rows)
        passes = max(-(-total // max_rows), min(nb, n_engines))
        if passes % n_engines:
            passes = min(nb, (passes // n_engines + 1) * n_engines)
        plan: List[List[int]] = []
        cur: List[int] = []
        cur_rows, left = 0, total
        for i, r in enumerate(rows):
            groups_left = max(1, passes - len(plan))
            # close the open pass when the next batch would overflow it, or when it already holds its share of what is left
            if cur and (cur_rows + r > max_rows or cur_rows >= (left + cur_rows) / groups_left):
                plan.append(cur)
                cur, cur_rows = [], 0
            cur.append(i)
            cur_rows += r
            left -= r
        plan.append(cur)
        return plan

    def generate_many(self, batches, threads: bool = False, coalesce_rows: int = 0, **generate_kw):
        """All batches, in order.  threads=True: one host thread per engine (batch j goes to engine j % n) - needed when
        the engines poll for early exit (cap_set_early_exit synchronises its stream: from a single host thread that would
        stall the launches of the other streams; ctypes releases the GIL during cap_generate, so the threads do overlap).
        coalesce_rows > 0: dynamic batching - consecutive batches of the same frame shape are concatenated into passes of at most
        that many rows (`coalesce_plan`; the engines must have been built with max_batch >= coalesce_rows) and the outputs split
        back per batch.  A frame decodes to the same bits alone, in its own batch and in a merged pass (batch invariance,
        DESIGN.md section 2), so the results are those of the uncoalesced call; what changes is that the decode chain's fixed
        costs are paid once per pass: 256-frame batches on 3 engines 6 010 captions/s, merged to 1024 rows 6 600 (round 6)."""
        batches = list(batches)
        self.last_coalesce = None            # what the last call did with coalesce_rows: the plan, or why it was not applied
        if coalesce_rows and len(batches) > 1:
            same = all(b.shape[1:] == batches[0].shape[1:] and b.dtype == batches[0].dtype and b.device == batches[0].device for b in batches)
            cap_rows = min(coalesce_rows, min(e.max_batch for e in self.engines))
            if generate_kw.get("output_logits"):
                self.last_coalesce = "not applied: per-step logits are recorded per pass"
            elif not same:
                self.last_coalesce = "not applied: the batches differ in frame shape, dtype or device"
            else:
                plan = self.coalesce_plan([int(b.shape[0]) for b in batches], len(self.engines), cap_rows)
                if any(len(g) > 1 for g in plan):
                    merged = [batches[g[0]] if len(g) == 1 else torch.cat([batches[j] for j in g], dim=0) for g in plan]
                    outs_m = self.generate_many(merged, threads=threads, **generate_kw)
                    self.last_coalesce = plan            # (the inner call cleared it)
                    outs: list = [None] * len(batches)
                    for g, om in zip(plan, outs_m):
                        unknown = sorted(set(om) - set(self._PER_ROW_OUTPUTS))
                        if unknown:              # a new output key must say here whether it is per row - never guessed from its shape
                            raise N.CaptionerHipError(f"generate_many(coalesce_rows=): output(s) {unknown} are not in the list of per-row "
                                                      f"outputs {self._PER_ROW_OUTPUTS}; cannot split a merged pass")
                        r0 = 0
                        for j in g:
                            n_j = int(batches[j].shape[0])
                            outs[j] = {k: v[r0:r0 + n_j] for k, v in om.items()}
                            r0 += n_j
                    return outs
                self.last_coalesce = f"not applied: {len(batches)} batches on {len(self.engines)} engines leave nothing to merge within {cap_rows} rows"
            logger.debug("generate_many(coalesce_rows=%d) %s", coalesce_rows, self.last_coalesce)
        if not threads or len(self.engines) == 1 or len(batches) <= 1:
            outs = [self.submit(b, **generate_kw) for b in batches]
            self.join()
            return outs
        import threading
        n = len(self.engines)
        outs: list = [None] * len(batches)
        errors: list = []
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)

        def work(i):
            try:
                with torch.cuda.device(self.device), torch.cuda.stream(self.streams[i]):
                    for j in range(i, len(batches), n):
                        outs[j] = self.engines[i].generate(batches[j], **generate_kw)
                        if batches[j].is_cuda:
                            batches[j].record_stream(self.streams[i])
            except Exception as e:  # noqa: BLE001 - re-raised on the caller's thread
                errors.append(e)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(min(n, len(batches)))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errors:
            raise errors[0]
        self.join()
        self._next = len(batches) % n
        return outs


class TextEncoderEngine:
    """Sentence encoder replica (CAP_ARCH_MINILM handle): WordPiece ids + lengths -> L2-normalised mean-pooled embeddings.
    Replaces `SentenceTransformer("all-MiniLM-L6-v2").encode(...)` (reference goal_exploration.py:57,102;
    pseudolabeler.py:568,677); tokenisation stays on the host (captioner/sentence_encoder.py)."""

    def __init__(self, arch: MiniLMArch, dtype: str = "bf16", max_batch: int = 64, max_len: int = 32,
                 device: str | torch.device = "cuda:0", share_weights_with: "TextEncoderEngine | None" = None):
        if not torch.cuda.is_available():
            raise N.CaptionerHipError("TextEncoderEngine needs a GPU; there is no CPU fallback in the product path")
        self.lib = N.load_library()
        self.arch, self.dtype, self.device = arch, dtype, torch.device(device)
        self.max_batch, self.max_len = max_batch, max_len
        cfg = N.CapConfig()
        cfg.struct_size = C.sizeof(N.CapConfig)
        cfg.arch = 2
        cfg.compute_dtype = _DTYPES[dtype]
        cfg.t_hidden, cfg.t_layers, cfg.t_heads, cfg.t_ffn = arch.hidden, arch.layers, arch.heads, arch.ffn
        cfg.vocab, cfg.max_pos, cfg.t_eps = arch.vocab, arch.max_pos, arch.eps
        cfg.max_batch, cfg.max_beams, cfg.max_len = max_batch, 1, max_len
        self._h = C.c_void_p()
        self.shares_weights = share_weights_with is not None
        with torch.cuda.device(self.device):
            if share_weights_with is not None:
                N.check(self.lib.cap_create_shared(C.byref(cfg), share_weights_with._h, C.byref(self._h)), "cap_create_shared")
            else:
                N.check(self.lib.cap_create(C.byref(cfg), C.byref(self._h)), "cap_create")

    close = CaptionerEngine.close
    __del__ = CaptionerEngine.__del__
    device_bytes = CaptionerEngine.device_bytes
    profile = CaptionerEngine.profile
    profile_report = CaptionerEngine.profile_report

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        """HF BertModel names; a leading `0.auto_model.` / `bert.` prefix (sentence-transformers / BertFor* saves) is dropped."""
        clean = {}
        for k, v in sd.items():
            for pre in ("0.auto_model.", "auto_model.", "bert."):
                if k.startswith(pre):
                    k = k[len(pre):]
            clean[k] = v
        self.is_coca = False
        CaptionerEngine.load_state_dict(self, clean, strict)

    def embed(self, ids: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
        """ids int [B, L] (pad anywhere after `lens[b]` tokens), lens int [B] -> fp32 [B, hidden] on the device."""
        if ids.dim() != 2 or lens.shape != (ids.shape[0],):
            raise ValueError(f"ids must be [B, L] and lens [B], got {tuple(ids.shape)} / {tuple(lens.shape)}")
        B, L = ids.shape
        if not ids.is_cuda and not lens.is_cuda:      # host tensors: validate here for free; device tensors are clamped by the
            if int(lens.max()) > L or int(lens.min()) < 1:                      # kernels (no host synchronisation per call)
                raise ValueError("lens must be within 1..L")
            if int(ids.max()) >= self.arch.vocab or int(ids.min()) < 0:
                raise ValueError("token id outside the vocabulary")
        ids = ids.to(self.device, torch.int32).contiguous()
        lens = lens.to(self.device, torch.int32).contiguous()
        out = torch.empty((B, self.arch.hidden), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            N.check(self.lib.cap_embed_text(self._h, C.c_void_p(ids.data_ptr()), C.c_void_p(lens.data_ptr()), B, L,
                                            C.c_void_p(out.data_ptr()), C.c_void_p(_stream_ptr(self.device))), "cap_embed_text")
        return out
