"""Base class of every captioner - mirror of the reference's
``experimenting_env/captioner/captioning_predictor.py:8-53`` (same attribute names, same `compute_perplexity`
arithmetic, same `print_caption`).  The reference derives from ``pl.LightningModule``; Lightning is used when it is
installed, otherwise ``torch.nn.Module`` provides the surface the callers rely on (`.to`, `.eval`, `__call__`)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

try:  # pragma: no cover - not installed in the build container
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    _Base = torch.nn.Module


class CaptioningPredictor(_Base):
    def __init__(self, cfg=None):
        super().__init__()
        self.perplexity = 0.0
        self.outputs = {}
        if cfg is not None:
            self.input_height, self.input_width = cfg.height, cfg.width

    def load_checkpoint_state_dict(self, sd, strict: bool = False):
        """What `Captioner(load_checkpoint=True)` does with `checkpoint['model']` (reference predictor_utils.py:182-185:
        `self.model.load_state_dict(checkpoint['model'], strict=False)` on the wrapper): wrapper / DDP key prefixes are
        dropped, tied heads filled in, and the tensors go to the weight store that the engine and every replica of its
        stream pool share.  Raises when no tensor of the dict belongs to the architecture."""
        from ..weights import BLIP_TIED, strip_wrapper_prefixes
        sd = strip_wrapper_prefixes(dict(sd))
        for dst, src in BLIP_TIED.items():
            if dst not in sd and src in sd:
                sd[dst] = sd[src]
        # the pool of a wrapper (cfg.streams > 1) is attached to the engine's weight store: one load serves every replica
        return self.engine.load_state_dict(sd, strict=strict)

    # ---- range of the split mode ("f32s", the wrappers' default): a GEMM input beyond +-65000 is clamped AND counted by the
    # library (cap_g8_saturations).  The wrappers look at the counter after their FIRST call and every `range_check_every`-th
    # call after it (each look synchronises the device), and `check_range()` can be called at the end of a job; a non-zero
    # count is logged as an error naming the remedy, or raised when cfg.strict_range is set.  The counter is per process and
    # GPU, not per handle: engines of one pool share it, which is the conservative side.
    range_check_every = 256

    def check_range(self, reset: bool = False) -> int:
        eng = getattr(self, "engine", None)
        if eng is None or getattr(eng, "dtype", None) not in ("f32s", "split", "f32_split"):
            return 0
        n = eng.saturations(reset=reset)
        if n:
            msg = (f"{n} GEMM-input values left the range of dtype 'f32s' (|x| > 65000) and were clamped: captions of this "
                   f"checkpoint are NOT fp32-grade in this mode - use dtype 'f32' (exact) or 'bf16' (INTEGRATION.md 6a)")
            if getattr(self, "strict_range", False):
                raise RuntimeError(msg)
            import logging
            logging.getLogger(__name__).error(msg)
        return n

    def _range_tick(self) -> None:
        k = getattr(self, "_range_calls", 0)
        self._range_calls = k + 1
        if k == 0 or (k + 1) % self.range_check_every == 0:
            self.check_range()

    @staticmethod
    def _cross_cache_for(cfg, sd, dtype) -> str:
        """The engine's cross-attention cache layout: cfg.cross_cache when given; otherwise "auto" (split mode: KV16) unless the
        checkpoint's key / value heads have outlier dimensions beyond what KV16 holds the parity bar for - then fp32 rows."""
        asked = getattr(cfg, "cross_cache", None)
        if asked:
            return asked
        if dtype in ("f32s", "split", "f32_split"):
            from ..weights import KV16_MAX_HEAD_SPREAD, cross_kv_head_spread
            sp = cross_kv_head_spread(sd)
            if sp > KV16_MAX_HEAD_SPREAD:
                import logging
                logging.getLogger(__name__).warning(
                    "cross-attention K/V heads of this checkpoint spread %.1fx around their median: keeping fp32 rows in the cross "
                    "cache instead of KV16 (1.9x the decode side's K/V bytes)", sp)
                return "fp32"
        return "auto"

    def pre_process_input(self, inputs):
        pass

    def forward(self, inputs):
        pass

    def training_step(self, batch, batch_idx):
        pass

    def backward(self):
        pass

    def configure_optimizers(self):
        pass

    def return_probabilities(self):
        pass

    def compute_perplexity(self, logits=None):
        """exp(-sum(log max softmax) / T) as float64 (reference :34-47).  `logits` [n, T, V]; default: the per-step
        logits of the last `forward` (`self.outputs["logits"]`, a sequence of T tensors [n, V])."""
        if logits is None:
            logits = torch.stack([l.float().cpu() for l in self.outputs["logits"]], dim=1)
        probs = F.softmax(logits, dim=-1)
        probs = torch.max(probs, dim=-1).values
        sum_log_probs = -probs.log().sum()
        num_tokens = probs.shape[1]
        self.perplexity = torch.exp(sum_log_probs / num_tokens).double()
        return self.perplexity

    def post_process_output(self, outputs):
        pass

    def print_caption(self):
        print(self.outputs["text"])
