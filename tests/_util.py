"""Shared helpers for the GPU parity tests (never imported by the product package)."""
import json
import os

import numpy as np
import torch

from embodied_captioning_amd.config import BlipArch
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    arch = BlipArch(**meta["arch"])
    return g, meta, arch


_SD_CACHE = {}


def golden_inputs(name):
    """(golden, meta, arch, state_dict, pixels) - the state dict is cached per fixture."""
    g, meta, arch = load_golden(name)
    key = (name, meta["seed"], meta["eos_boost"])
    if key not in _SD_CACHE:
        _SD_CACHE[key] = procedural_blip_state_dict(arch, meta["seed"], eos_boost=meta["eos_boost"])
    px = synthetic_pixels(meta["batch"], arch.image_size, seed=meta["seed"])
    return g, meta, arch, _SD_CACHE[key], px


def pad_to(seq, L, fill):
    """HF crops generated sequences at the longest row; our ABI returns [B, max_len]."""
    seq = np.asarray(seq)
    if seq.shape[1] == L:
        return seq
    out = np.full((seq.shape[0], L), fill, dtype=seq.dtype)
    out[:, : seq.shape[1]] = seq
    return out


def token_parity(ours, ref, margins, tau):
    """Greedy-token comparison that knows about near-ties.

    ours/ref: int [B, L] incl. BOS.  margins: fp32 [L-1, B] oracle top1-top2 logit gap at each step.
    A row may leave the oracle's path only at a step whose oracle margin is below `tau`; after that step
    the row is on a different (equally valid) prefix and is not compared further.
    Returns (n_exact_rows, n_diverged_rows, first_bad) - first_bad is None when every divergence was at a near-tie.
    """
    B, L = ref.shape
    exact = diverged = 0
    first_bad = None
    for b in range(B):
        row_ok = True
        for s in range(1, L):
            if ours[b, s] != ref[b, s]:
                row_ok = False
                m = margins[s - 1, b] if s - 1 < margins.shape[0] else 0.0
                if not (m < tau) and first_bad is None:
                    first_bad = (b, s, int(ours[b, s]), int(ref[b, s]), float(m))
                break
        exact += row_ok
        diverged += (not row_ok)
    return exact, diverged, first_bad


# ---- G8: the GEMM-operand layout of the split-fp16 mode (embodied_captioning_amd/csrc/common.h) ----------------------
G8_WSCALE = 4096.0


def g8_encode(x, scale=1.0):
    """fp32 [..., K] (K % 8 == 0) -> float32-typed container of the same shape whose bytes are the G8 image: every 8
    consecutive elements of a row = [8 fp16 hi | 8 fp16 lo], hi = rn16(s x), lo = rn16(s x - hi)."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32)) * np.float32(scale)
    x = np.clip(x, -65000.0, 65000.0).astype(np.float32)     # G8_AMAX: the device clamps to fp16's range the same way
    assert x.shape[-1] % 8 == 0
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    g = np.stack([hi.reshape(*x.shape[:-1], -1, 8), lo.reshape(*x.shape[:-1], -1, 8)], axis=-2)   # [..., K/8, 2, 8]
    return np.ascontiguousarray(g).view(np.float32).reshape(x.shape)


def g8_decode(c, scale=1.0):
    """Inverse of g8_encode: container (float32-typed, [..., K]) -> the fp32 values hi + lo (divided by scale)."""
    c = np.ascontiguousarray(np.asarray(c, dtype=np.float32))
    h = c.view(np.float16).reshape(*c.shape[:-1], -1, 2, 8).astype(np.float32)
    return ((h[..., 0, :] + h[..., 1, :]) / np.float32(scale)).reshape(c.shape)
