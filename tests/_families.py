"""Weight families for the parity envelope of the split mode (tests/test_envelope_gpu.py, tests/explore_parity_envelope.py): transforms of
the procedural BLIP state dict that make it look like a TRAINED checkpoint in the ways that matter to a two-halves fp16
representation - heavy-tailed weights, LayerNorm gains spread over orders of magnitude, a few massive residual channels, rows
that LayerNorm squeezes far below 1, and activations beyond fp16's range.  Every transform is deterministic."""
import numpy as np
import torch


def _rng(seed, tag):
    import zlib
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(tag.encode())]))


def _is_linear_weight(k, v):
    return v.dim() == 2 and k.endswith("weight") and "embeddings" not in k


def heavy_tails(sd, seed=0, sigma=0.6):
    """Log-normal magnitudes on every Linear weight (kurtosis ~ 20 instead of 3), rescaled to the original RMS per tensor."""
    out = dict(sd)
    for k, v in sd.items():
        if _is_linear_weight(k, v):
            m = torch.from_numpy(_rng(seed, k).lognormal(0.0, sigma, size=tuple(v.shape)).astype(np.float32))
            w = v * m
            out[k] = w * (v.pow(2).mean().sqrt() / w.pow(2).mean().sqrt())
    return out


def gamma_spread(sd, seed=0, lo=0.05, hi=20.0, unit_rms=True):
    """LayerNorm gains log-uniform over [lo, hi] per channel - a factor hi / lo = 400 between channels of one GEMM input row
    (vision and text towers; the LM head's transform LayerNorm is left alone so that the logits keep their scale).
    unit_rms: the gains of a LayerNorm are rescaled to RMS 1, so the network keeps its overall gain.  Without that (gains up to
    20 in 49 LayerNorms) the network is ill-conditioned: two CORRECT fp32 implementations that differ only in summation order
    diverge from each other (measured: the exact-product fp32 kernels leave the CPU oracle as far as the split mode does)."""
    out = dict(sd)
    for k, v in sd.items():
        if v.dim() == 1 and ("layer_norm" in k or "LayerNorm" in k or "layernorm" in k) and k.endswith("weight") and "predictions" not in k:
            g = np.exp(_rng(seed, k).uniform(np.log(lo), np.log(hi), size=v.shape[0])).astype(np.float32)
            if unit_rms:
                g = g / np.sqrt(np.mean(g * g))
            out[k] = torch.from_numpy(g) * torch.sign(v)
    return out


def massive_channels(sd, arch, seed=0, scale=300.0, n=3):
    """A few residual-stream channels carry values hundreds of times the rest (the 'massive activations' of trained ViTs / LMs):
    the bias of an early fc2 (vision) and of an early FFN output (text) gets +-scale on n channels.  LayerNorm then squeezes every
    OTHER channel of those rows by ~1/scale * sqrt(width / n): G8 lo halves of the squeezed values go subnormal."""
    out = dict(sd)
    r = _rng(seed, "massive")
    for key, width in (("vision_model.encoder.layers.1.mlp.fc2.bias", arch.v_hidden),):
        b = sd[key].clone()
        ch = r.choice(width, size=n, replace=False)
        b[torch.from_numpy(ch)] += torch.from_numpy((scale * r.choice([-1.0, 1.0], size=n)).astype(np.float32))
        out[key] = b
    return out


def small_gains(sd, seed=0, gain=2e-3):
    """Every LayerNorm of the vision tower has gain ~ 2e-3 and bias 0: all LayerNorm outputs are below 1e-2 in magnitude, so the
    lo half of every GEMM input of the tower is an fp16 SUBNORMAL (hi + lo keeps ~15-18 bits instead of 22)."""
    out = dict(sd)
    for k, v in sd.items():
        if k.startswith("vision_model.encoder") and "layer_norm" in k:
            out[k] = torch.full_like(v, gain) if k.endswith("weight") else torch.zeros_like(v)
    return out


def beyond_fp16(sd, arch, value=1.0e5):
    """One fc1 unit of the vision tower is pushed to a pre-GELU value of 1e5 on every token: GELU keeps it, the G8 store of the
    fc2 GEMM's input must clamp it to 65 000 - the case the mode cannot represent and has to REPORT."""
    out = dict(sd)
    b = sd["vision_model.encoder.layers.2.mlp.fc1.bias"].clone()
    b[7] = value
    out["vision_model.encoder.layers.2.mlp.fc1.bias"] = b
    return out


def cross_kv_outliers(sd, arch, seed=0, factor=10.0, n=2):
    """n of the 64 dimensions of every head of every cross-attention key and value projection are `factor` times the others:
    the head rows of the KV16 cache (64 int16 with ONE scale) then quantise their ordinary dimensions `factor` times more
    coarsely.  Measured against the same engine with fp32 rows (profiles/r04_kv16_outlier_probe.txt): logits move by 1.5e-4 at
    factor 8, 3e-4 at 12, 2.3e-3 at 30 - so the library refuses KV16 beyond a spread of 12 (weights.KV16_MAX_HEAD_SPREAD);
    the family sits inside that guard, test_cross_cache_gpu.py exercises the outside."""
    out = dict(sd)
    r = _rng(seed, "kvout")
    for k, v in sd.items():
        if ".crossattention.self.key." in k or ".crossattention.self.value." in k:
            m = torch.ones(v.shape[0])
            for h in range(v.shape[0] // 64):
                m[h * 64 + torch.from_numpy(r.choice(64, size=n, replace=False))] = factor
            out[k] = v * (m[:, None] if v.dim() == 2 else m)
    return out


ILL_CONDITIONED = {"gamma_spread_raw": lambda sd, arch: gamma_spread(sd, 2, unit_rms=False)}

FAMILIES = {
    "heavy_tails": lambda sd, arch: heavy_tails(sd, 1),
    "gamma_spread": lambda sd, arch: gamma_spread(sd, 2),                         # 0.05 .. 20 (x400), unit RMS
    "gamma_spread_wide": lambda sd, arch: gamma_spread(sd, 3, 0.02, 50.0),          # x2500
    "massive_channels": lambda sd, arch: massive_channels(sd, arch, 4),
    "massive_channels_1000": lambda sd, arch: massive_channels(sd, arch, 5, scale=1000.0),
    "small_gains": lambda sd, arch: small_gains(sd, 6),
    "heavy_tails+massive": lambda sd, arch: massive_channels(heavy_tails(sd, 7), arch, 7),
    "cross_kv_outliers": lambda sd, arch: cross_kv_outliers(sd, arch, 8),          # the block-scaled KV16 cache's hard case (x10)
}
