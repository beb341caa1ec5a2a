"""Generation options the reference's models accept and this library does NOT implement are rejected BY NAME, never ignored.

Reference surface: ``coca_model.py:205-224`` (``CoCa.generate``: temperature, generation_type, top_p, top_k, repetition_penalty,
stopping_criteria, text prompt, ...; logit processors at ``:236-241``, warpers at ``:266-275``, the sampling loop at ``:304-320``)
and HF ``GenerationMixin.generate`` behind ``blip2.py:26``.  What the library runs: greedy decoding (CoCa: ``generation_type=
"top_k"`` with ``top_k=1`` - what the reference's wrapper calls, ``coca.py:29``; MinLength and forced EOS included) and beam search
(HF v5 semantics for BLIP / BLIP-2, the reference's ``_generate_beamsearch`` incl. beam groups for CoCa).  Everything below is
deterministic arg-max / beam arithmetic; sampling, temperature, nucleus / top-k filtering and repetition penalties are not built.
"""
from __future__ import annotations

from typing import Any, Mapping

# option -> the one value that means "off" (the reference's / HF's default)
_NEUTRAL = {
    "temperature": 1.0,
    "repetition_penalty": 1.0,
    "do_sample": False,
    "no_repeat_ngram_size": 0,
    "encoder_no_repeat_ngram_size": 0,
    "diversity_penalty": 0.0,
    "penalty_alpha": None,
    "typical_p": 1.0,
    "epsilon_cutoff": 0.0,
    "eta_cutoff": 0.0,
    "min_p": None,
    "stopping_criteria": None,
    "logits_processor": None,
    "bad_words_ids": None,
    "force_words_ids": None,
    "constraints": None,
    "prefix_allowed_tokens_fn": None,
    "text": None,                      # coca_model.py:207 - a text prompt to continue
}
GENERATION_TYPES = ("top_k", "beam_search")


def reject_unsupported_generation_options(opts: Mapping[str, Any], where: str = "generate") -> None:
    """Raise ValueError naming every option of `opts` that asks for something the library does not compute.  Options set to their
    neutral value (temperature 1, repetition_penalty 1, top_k 1 with generation_type "top_k", ...) pass; unknown keys are not
    judged here (callers keep their own signatures)."""
    bad = []
    gt = opts.get("generation_type")
    if gt is not None and gt not in GENERATION_TYPES:
        bad.append(f"generation_type={gt!r} (implemented: 'top_k' with top_k=1 = greedy, 'beam_search'; 'top_p' is nucleus sampling)")
    k = opts.get("top_k")
    if k is not None and int(k) > 1:
        bad.append(f"top_k={k} (only top_k=1, i.e. arg-max, is implemented: larger values sample from the k best tokens)")
    p = opts.get("top_p")
    # the reference's signature default is top_p=0.1 and it is only read when generation_type == "top_p" (coca_model.py:266-267)
    if p is not None and gt == "top_p":
        bad.append(f"top_p={p} (nucleus sampling)")
    if p is not None and gt is None and float(p) < 1.0 and opts.get("do_sample"):
        bad.append(f"top_p={p} (nucleus sampling)")
    for name, neutral in _NEUTRAL.items():
        if name in opts and opts[name] is not None and opts[name] != neutral:
            bad.append(f"{name}={opts[name]!r}")
    if bad:
        raise ValueError(f"{where}: not implemented by the MI355X captioner library (greedy / beam search only) - " + "; ".join(bad))
