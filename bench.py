#!/usr/bin/env python
"""Captioner throughput bench (BASELINE.json metric: captions/sec, 224x224, beam=1, + greedy token parity).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W         # starts its own N ranks (a torch.distributed.run CHILD process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W            # the driver's form: WORLD_SIZE is set, nothing is spawned

One step = the hot path (ViT encoder -> cross-K/V -> 19 greedy decode steps) over one batch of 256 synthetic 224x224 frames
per GPU that are already resident in HBM, followed by the RCCL all-gather of the caption records (ids int32 [256,20] + lengths)
that feeds the consensus step.  Frames and weights are synthetic/procedural (no dataset or checkpoint exists offline).  Rank 0
prints ONE JSON line.

`value` is measured the way the product's batch entry points run (`BLIP.generate_batch` / `Captioner.caption_batch` /
`pseudolabeler.BatchedBoxCaptioner` with `captioner.streams: 3`): the K timed steps' batches go to an EnginePool of --streams
engines whose dynamic batching (--coalesce-rows, default 1024) merges CONSECUTIVE STEPS' batches into passes of up to 1024 rows -
20 steps run as passes of 4, 4, 3, 3, 3, 3 batches - and splits the captions back per step.  A frame decodes to the same bits
alone, in its 256-frame batch and in a merged pass (tests/test_merged_passes_gpu.py), so the captions are those of the unmerged
steps; what a merged pass changes is the latency of a batch (its pass finishes as a whole) and the number of decode chains.
`config.workload` / `config.pass_rows` name the pass size; the figure with every 256-frame batch as its own pass
(--coalesce-rows 0; rounds 1-4's headline, BASELINE's "batch=256" read strictly) is on the same line as `pool_uncoalesced` and in
`config.also`; one batch at a time on one stream is `single_stream`.

The headline (`value`, `dtype`, `roofline`, `kernels`, `parity`) is the mode that holds the metric's parity clause: "f32s" =
CAP_F32_SPLIT, fp32 values carried into every GEMM as two fp16 halves with three fp16 MFMAs per product (DESIGN.md section
2): greedy tokens identical to the fp32 reference on all 256 golden rows (one whole batch).  The faster bf16 mode, which is NOT token-identical,
is reported under the extra key `bf16`; the exact-product fp32-MFMA mode under `f32_exact`.

    python bench.py --gpus N --strong --frames 50000      # strong scaling (SURVEY config 4): a FIXED total of frames,
                                                          # contiguous shards, micro-batches of --batch, one all-gather

The K timed steps rotate over --streams engines (default 3), each with its own arena and HIP stream: batches are
independent, a single generate leaves most of the GPU idle (launch-bound decode chain), and kernels of different streams
overlap here - every step is still one whole batch and all K finish inside the timed region; --streams 1 times them one
after the other.  The roofline / per-kernel pass, the encoder-only, fp32 and CPU legs run one engine on one stream.

Context keys of a full run, measured live and labelled as such: `roofline.vendor_fp16_gemm_tflops_same_shapes` (torch.mm / hipBLASLt
on the encoder GEMM's four shapes with plain fp16 operands - nothing in the product calls a vendor GEMM) beside the kernel's
`executed_tflops`; `cpu_baseline.runs.hf_transformers_8_frames` (HF's own BlipForConditionalGeneration.generate, what the
reference's wrappers call) beside the CPU port - `cpu_baseline.value` is the CPU's best figure, `kind` says whose.

Other lines: --beams 3 --batch 64 (config 3), --model coca --image-size 336 --beams 5 --batch 128 (config 5 on one GPU),
--model blip2 [--load-in-8bit] [--batch 1] (the reference's production captioner), --model minilm; --gpus 2 --share-gpu
rehearses the N > 1 path with real engines on a one-GPU box (all ranks on cuda:0 over gloo; not a scaling figure).
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from embodied_captioning_amd.config import BlipArch                                   # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine                            # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "f32s": 2500.0}          # dense MFMA peaks, MI355X_MICROARCH.md (f32s runs on the fp16 pipe)
MFMA_PER_PRODUCT = {"bf16": 1, "f32": 1, "f32s": 3}                     # f32s: hi.hi + hi.lo + lo.hi
ENC_GEMM_TAGS = ("gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--max-length", type=int, default=20)
    ap.add_argument("--dtype", default=None, choices=["bf16", "f32", "f32s"],
                    help="f32s (token-identical to the fp32 reference; default for blip and blip2) | bf16 (fastest, near-tie token flips; "
                         "default for coca - SURVEY config 5 names bf16 - and minilm) | f32")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --frames in total, sharded contiguously over the "
                    "ranks (distributed.caption_shard: micro-batches of --batch on the stream pool, ONE caption all-gather at "
                    "the end); value = frames / wall time, scaling = strong")
    ap.add_argument("--frames", type=int, default=50000, help="--strong: total frames of the job (SURVEY config 4: 50000)")
    ap.add_argument("--no-extra-modes", action="store_true", help="skip the bf16 and exact-fp32 legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strict", action="store_true", help="(same as --no-extra-modes)")
    ap.add_argument("--cpu-sample", type=int, default=64, help="captions timed on the host CPU oracle")
    ap.add_argument("--model", default="blip", choices=["blip", "coca", "minilm", "blip2"],
                    help="blip = BASELINE.json metric workload (default); coca = extra line for config 5's model "
                         "(CoCa ViT-L/14, reference top-k(1) loop, seq_len 30)")
    ap.add_argument("--load-in-8bit", action="store_true", help="blip2: the reference's load mode (blip2.py:19-22) - int8 Linear weights as "
                    "bitsandbytes stores them, bf16 activations (forces --dtype bf16)")
    ap.add_argument("--beams", type=int, default=1, help="> 1: extra line for SURVEY config 3 (HF beam search; use --batch 64)")
    ap.add_argument("--streams", type=int, default=3, help="blip: engines (own arena + HIP stream each) the timed steps rotate "
                    "over, so that consecutive batches overlap; 1 = one engine, one stream (the profiling passes always use one)")
    ap.add_argument("--coalesce-rows", type=int, default=None, help="blip / coca: dynamic batching of the engine pool - consecutive steps' batches "
                    "are merged into passes of at most this many rows (images, for coca) (EnginePool.generate_many(coalesce_rows=)); a frame "
                    "has the same bits alone, in its batch and in a merged pass; 0 = every batch its own pass (the `pool_uncoalesced` key).  "
                    "Default: 1024 (blip), 4 batches up to 512 images (coca: the wrapper's default)")
    ap.add_argument("--early-exit", type=int, default=0, help="poll the device every N decode steps and leave the loop when "
                    "every caption is finished (HF's stopping rule; 0 = never, the default: no host sync in generate)")
    ap.add_argument("--eos-boost", type=float, default=9.0, help="blip: EOS logit offset of the procedural weights (9 = the "
                    "golden's weights; larger values end every caption early)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling job (--frames in total over all ranks) that the "
                    "default line reports beside the weak-scaling value")
    ap.add_argument("--stub-engine", action="store_true", help="TEST ONLY (tests/test_distributed_cpu.py): the launch / rendezvous / "
                    "timed-region / gather / strong-scaling plumbing on CPU ranks over gloo with a fake captioner whose ids are a "
                    "function of the frame index; prints a line marked \"stub\": true and never touches a GPU")
    ap.add_argument("--share-gpu", action="store_true", help="REHEARSAL ONLY (a one-GPU box): every rank of --gpus N on cuda:0, process "
                    "group over gloo - the N > 1 code path (shards, per-step gather, timed region, strong-scaling job) with the real "
                    "engines, everything but RCCL; the line says so and its value is not a scaling figure")
    ap.add_argument("--decode-path", default="auto", choices=["auto", "batch", "small"],
                    help="blip: decode kernels of the timed steps (engine.set_decode_path; A/B of the batch path's kernel sets)")
    ap.add_argument("--row-compaction", default="on", choices=["on", "off"],
                    help="blip: the greedy decode loop on the open captions' rows only (engine.set_row_compaction; same tokens; off = A/B)")
    ap.add_argument("--lite", action="store_true", help="timed steps only (profiler counter passes): no roofline / "
                    "encoder-only / parity / fp32 / CPU legs")
    ap.add_argument("--no-latency", action="store_true", help="skip the small-batch `latency` block (profiler passes: its B = 1 / 8 / 64 "
                    "launches would be averaged into the headline workload's kernel durations)")
    ap.add_argument("--latency-only", action="store_true", help="print only the small-batch `latency` block (B = 1, 8, 64)")
    ap.add_argument("--image-size", type=int, default=224, help="coca: 224 or 336 (SURVEY config 5); blip: 224 (the "
                    "BASELINE config) or 384 (what the published BLIP checkpoints ship - extra line, no golden)")
    return ap.parse_args()


class PowerSampler:
    """Socket power of THIS process's GPU from the amdgpu hwmon node (power1_input, microwatts), sampled every 50 ms by a
    host thread while the timed steps run - a reading, never part of the metric.  The card is found through the device's PCI
    bus id; any failure (no sysfs access, no node) leaves `result()` as None."""

    def __init__(self, device_index: int):
        import glob
        import threading
        self.path, self.samples, self._stop, self._t = None, [], threading.Event(), None
        try:
            bus = torch.cuda.get_device_properties(device_index).pci_bus_id
            want = f"{bus:02x}:" if isinstance(bus, int) else str(bus).lower()
            dom = getattr(torch.cuda.get_device_properties(device_index), "pci_domain_id", 0)
            dev = getattr(torch.cuda.get_device_properties(device_index), "pci_device_id", 0)
            addr = f"{dom:04x}:{bus:02x}:{dev:02x}" if isinstance(bus, int) else want
            for card in glob.glob("/sys/class/drm/card*/device"):
                if os.path.realpath(card).lower().rsplit("/", 1)[-1].startswith(addr):
                    nodes = glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_input"))
                    if nodes:
                        self.path = nodes[0]
        except Exception:  # noqa: BLE001
            self.path = None
        self._threading = threading

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append(int(open(self.path).read()) * 1e-6)
            except Exception:  # noqa: BLE001
                return
            self._stop.wait(0.05)

    def __enter__(self):
        if self.path:
            self._t = self._threading.Thread(target=self._run, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._t:
            self._t.join()

    def result(self, captions: float, seconds: float):
        if len(self.samples) < 4:
            return None
        w = float(np.mean(self.samples[1:]))
        return {"watts_mean": round(w, 1), "watts_max": round(float(np.max(self.samples)), 1), "samples": len(self.samples),
                "joules_per_caption": round(w * seconds / captions, 4),
                "source": "amdgpu hwmon power1_input of this GPU, 50 ms samples over the timed region (informational)"}


class _NoPower:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        pass

    def result(self, captions, seconds):
        return None


def timed_steps(eng, px, L, steps, warmup, world, gather, beams=1, dev="cuda", coalesce=0):
    """`eng`: a CaptionerEngine, or an EnginePool - consecutive steps then run on the pool's engines / streams and overlap
    (each step is still one whole batch through encoder + decode + gather; all of them finish inside the timed region)."""
    pool = eng if hasattr(eng, "submit") else None

    def run(n):
        if pool is None:
            for _ in range(n):
                out = eng.generate(px, num_beams=beams, max_length=L)
                res = gather(out["sequences"], out["lengths"])
            return res
        # the batches overlap on the pool's streams; the caption all-gathers (one per step, as before) are issued in step
        # order on the caller's stream once the batches are joined - every rank issues its collectives in the same order
        outs = pool.generate_many([px] * n, threads=True, coalesce_rows=coalesce, num_beams=beams, max_length=L)   # a host thread per engine
        for out in outs:
            res = gather(out["sequences"], out["lengths"])
        return res
    from embodied_captioning_amd.distributed import timed_region
    # every engine of a pool runs once untimed - with dynamic batching at the timed passes' size (their merged input buffers come
    # out of the allocator's cache in the timed region)
    run(max(warmup, len(pool) * max(1, coalesce // max(1, int(px.shape[0])))) if pool is not None else warmup)
    on_gpu = dev != "cpu"
    with (PowerSampler(torch.cuda.current_device()) if on_gpu else _NoPower()) as ps:
        # barrier + synchronise, the K steps, synchronise + barrier, MAX over ranks (distributed.timed_region)
        dt, res = timed_region(lambda: run(steps), world, torch.device("cuda", torch.cuda.current_device()) if on_gpu else None)
    timed_steps.power = ps
    return dt, res


def pooled(a, arch, sd, **kw):
    """EnginePool of --streams engines for the timed steps of the auxiliary lines (None for --streams 1)."""
    if a.streams <= 1:
        return None
    from embodied_captioning_amd.engine import EnginePool
    pool = EnginePool(arch, n=a.streams, dtype=a.dtype, **kw)
    pool.load_state_dict(sd)
    return pool


KERNEL_NAME = {"bf16": "gemm_pp_kernel<bf16> 256x256 LDS-DMA, wave groups half a stage apart, 16x16x32 bf16 MFMA (ViT qkv/proj/fc1/fc2)",
               "f32": "gemm_big_kernel 256x256 LDS-DMA, 32x32x2 fp32 MFMA (ViT qkv/proj/fc1/fc2 launches)",
               "f32s": "gemm_pp_kernel<g8_t> 256x256 LDS-DMA, wave groups half a stage apart, split fp16: 3 x 16x16x32 f16 MFMA per product "
                       "(ViT qkv/proj/fc1/fc2)"}


def encoder_only(eng, px, arch, steps=5):
    """image tower only on the same batch; 2*MAC flops of the GEMMs + attention per image."""
    for _ in range(2):
        eng.encode(px)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.encode(px)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fl = arch.encoder_flops_per_image() * px.shape[0]
    return {"images_per_s": round(px.shape[0] / dt, 1), "ms_per_batch": round(1e3 * dt, 3),
            "tflops": round(fl / dt / 1e12, 1), "gflop_per_image": round(arch.encoder_flops_per_image() / 1e9, 3)}


DEC_GEMM_TAGS = ("dec_gemm_qkv", "dec_gemm_so", "dec_gemm_cq", "dec_gemm_co", "dec_gemm_f1", "dec_gemm_f2")
HBM_PEAK_GBPS = 8000.0                                                 # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK = 157.3                                                 # the pipe the reference's fp32 arithmetic would need


POWER_LIMITED_MFMA = {"f32s": 1960.0}     # v_mfma_f32_16x16x32_f16, tools/mfma_power_probe.sh (measured, one box)


VENDOR_CONTEXT = True        # main() clears it for --no-extra-modes / --lite runs (profiler passes: only the product's kernels in the trace)


def vendor_gemm_context(batch, n_tokens, D, Mv, seconds=1.0):
    """CONTEXT ONLY - nothing in the product calls a vendor GEMM: what torch.mm (hipBLASLt / rocBLAS, AMD's tuned kernels) sustains
    on this board on the same four ViT Linear shapes with plain fp16 operands (ONE MFMA product per MAC, no bias / GELU / split
    epilogue), random normal data, the four in rotation for ~1 s.  The split kernel's `executed_tflops` is the figure beside it."""
    if not VENDOR_CONTEXT:
        return None
    try:
        M = batch * n_tokens
        shapes = [(3 * D, D), (D, D), (Mv, D), (D, Mv)]
        ops = [(torch.randn(M, K, device="cuda", dtype=torch.float16), (torch.randn(N, K, device="cuda") * 0.03).half()) for N, K in shapes]
        fl = sum(2.0 * M * N * K for N, K in shapes)
        for a, w in ops:
            torch.mm(a, w.t())
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < seconds:
            for a, w in ops:
                torch.mm(a, w.t())
            n += 1
            if n % 20 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return round(fl * n / (time.perf_counter() - t0) / 1e12, 1)
    except Exception as e:  # noqa: BLE001 - context only
        log(f"vendor GEMM context skipped: {e!r}")
        return None


def roofline_pass(eng, px, L, dtype, arch, batch):
    """Per-kernel HIP-event timing (events recorded on the launch stream inside the library).

    roofline.achieved / frac follow SURVEY.md 8(d): ALGORITHMIC flops of the dominant kernel's launches = 2 M N K of the Linear
    layers, divided by the launch duration and by the dense peak of the MFMA pipe the kernel runs on.  What the split kernel
    executes (three fp16 products per MAC) and how busy the pipe was are separate keys (executed_tflops / frac_executed /
    mfma_busy_pmc), as is the ratio to the fp32 MFMA pipe the same arithmetic would otherwise need (frac_vs_fp32_mfma)."""
    eng.profile(True)
    reps = 2
    for _ in range(reps):
        gen = eng.generate(px, num_beams=1, max_length=L)
    rep = eng.profile_report()
    eng.profile(False)
    # rows a decode step still reads: a caption of n tokens (BOS .. EOS) is open for its first n - 1 steps; the attention
    # kernels skip a row from the step after its EOS on
    live_row_steps = float(torch.clamp(gen["lengths"].float() - 1, max=L - 1).sum())
    kernels = {}
    for tag, r in rep.items():
        kernels[tag] = {"launches_per_step": r["launches"] // reps, "ms_per_step": r["ms"] / reps,
                        "avg_us": 1e3 * r["ms"] / max(r["launches"], 1),
                        "tflops": (r["flops"] / (r["ms"] * 1e-3) / 1e12) if r["flops"] else None,
                        "gbps": r["bytes"] / (r["ms"] * 1e-3) / 1e9}
    fl = sum(rep[t]["flops"] for t in ENC_GEMM_TAGS if t in rep)
    ms = sum(rep[t]["ms"] for t in ENC_GEMM_TAGS if t in rep)
    n = sum(rep[t]["launches"] for t in ENC_GEMM_TAGS if t in rep)
    k = MFMA_PER_PRODUCT[dtype]
    alg = fl / (ms * 1e-3) / 1e12                    # 2 M N K / t
    peak = PEAK_TFLOPS[dtype]
    def _avg_us(tags):
        ms_ = sum(rep[t]["ms"] for t in tags if t in rep); n_ = sum(rep[t]["launches"] for t in tags if t in rep)
        return 1e3 * ms_ / n_ if n_ else None
    pm = pmc_summary(dtype, {"enc_gemm": 1e3 * ms / n if n else None, "cross_attention": _avg_us(("dec_cross_attn",)),
                             "decode_gemm": _avg_us(DEC_GEMM_TAGS)})
    roof = {"bound": "mfma", "kernel": KERNEL_NAME[dtype], "achieved": round(alg, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(alg / peak, 4), "traffic": pm["enc_gemm"].get("traffic"),
            "algorithmic_flops_per_launch": fl / n, "avg_launch_us": round(1e3 * ms / n, 2), "launches_per_step": n // reps,
            "algorithmic_bytes_per_launch": round(sum(rep[t]["bytes"] for t in ENC_GEMM_TAGS if t in rep) / n),
            "mfma_products_per_mac": k, "executed_tflops": round(k * alg, 2), "frac_executed": round(k * alg / peak, 4),
            "mfma_busy_pmc": pm["enc_gemm"].get("mfma_busy"), "frac_vs_fp32_mfma": round(alg / FP32_MFMA_PEAK, 3),
            # what a register-only loop of the same MFMA instruction sustains on this board with RANDOM fp16 operands: the socket
            # power limit (~1.3 kW) holds the clock at 2.0 GHz (profiles/r03_mfma_power.txt; 2 390 TFLOP/s with one constant pair)
            "power_limited_mfma_tflops": POWER_LIMITED_MFMA.get(dtype), "frac_executed_vs_power_limited":
            round(k * alg / POWER_LIMITED_MFMA[dtype], 4) if POWER_LIMITED_MFMA.get(dtype) else None,
            # context, measured in this run: AMD's own tuned GEMM (torch.mm -> hipBLASLt) on the same four shapes with plain fp16
            # operands, one product per MAC and no epilogue work - what the vendor's kernel makes of this board on these shapes
            "vendor_fp16_gemm_tflops_same_shapes": vendor_gemm_context(batch, arch.n_tokens, arch.v_hidden, arch.v_mlp) if dtype == "f32s" else None,
            "traffic_source": pm["source"], "pmc_stale": pm["pmc_stale"], "pmc_stale_why": pm["pmc_stale_why"],
            "note": "achieved = 2MNK of the ViT Linear layers per launch / HIP-event launch time (the proj / fc2 launches also add their "
                    "output into the fp32 residual stream in place: +155 MB read each, counted in algorithmic_bytes_per_launch); frac = achieved / dense peak of "
                    "the MFMA pipe the kernel runs on (MI355X_MICROARCH.md).  executed_tflops counts the MFMA products the "
                    "algorithm spends per MAC (split mode: hi.hi + hi.lo + lo.hi = 3); mfma_busy_pmc = SQ_VALU_MFMA_BUSY_CYCLES / "
                    "(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) of the same launches in the committed counter pass; frac_vs_fp32_mfma = "
                    "achieved / 157.3 TFLOP/s, the pipe exact fp32 products need (the f32_exact leg runs there)"}
    # decode side: HBM-bound kernels as bytes/s against the HBM peak.  cross-attention = the image's fp32 K/V blocks of the rows
    # still open; decode GEMMs = weights + activations + split-K slabs per launch (what the ProfScope of each launch counts)
    dec = {}
    if "dec_cross_attn" in rep:
        r = rep["dec_cross_attn"]
        all_rows = r["bytes"] / r["launches"]                              # K and V blocks of every row of the batch
        per_row = all_rows / batch
        layers = arch.t_layers
        live = per_row * live_row_steps * layers * reps                    # what the launches of the pass actually had to read
        g = live / (r["ms"] * 1e-3) / 1e9
        dec["cross_attention"] = {"bound": "hbm", "achieved": round(g, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": round(g / HBM_PEAK_GBPS, 4), "traffic": pm["cross_attention"].get("traffic"),
                                  "algorithmic_bytes_per_launch": round(live / r["launches"]),
                                  "bytes_per_launch_if_every_row_were_open": round(all_rows),
                                  "live_row_steps": int(live_row_steps), "row_steps": batch * (L - 1),
                                  "avg_launch_us": round(1e3 * r["ms"] / r["launches"], 2), "launches_per_step": r["launches"] // reps,
                                  "ms_per_step": round(r["ms"] / reps, 3),
                                  "note": "algorithmic bytes = KV16 K and V blocks (132 bytes per 64-wide head row) of the rows still OPEN at each "
                                          "step (a caption of n tokens is read for n - 1 steps; ended rows are skipped by the kernel), "
                                          "counted from this run's caption lengths; traffic = the counter file's bytes per launch"}
    tags = [t for t in DEC_GEMM_TAGS if t in rep]
    if tags:
        by = sum(rep[t]["bytes"] for t in tags); ms_d = sum(rep[t]["ms"] for t in tags); nl = sum(rep[t]["launches"] for t in tags)
        g = by / (ms_d * 1e-3) / 1e9
        dec["gemm"] = {"bound": "hbm", "achieved": round(g, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(g / HBM_PEAK_GBPS, 4),
                       "traffic": pm["decode_gemm"].get("traffic"), "algorithmic_bytes_per_launch": round(by / nl),
                       "avg_launch_us": round(1e3 * ms_d / nl, 2), "launches_per_step": nl // reps, "ms_per_step": round(ms_d / reps, 3),
                       "kernel": pm["decode_gemm"].get("kernel")}
    dec_ms = sum(r["ms"] for t, r in rep.items() if t.startswith("dec_") or t in ("greedy_select", "beam_step")) / reps
    dec["ms_per_step_all_decode_kernels"] = round(dec_ms, 3)
    total_ms = sum(r["ms"] for r in rep.values()) / reps
    return roof, kernels, total_ms, dec


def host_cores() -> int:
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return max(1, n)


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


PMC_FILES = {"f32s": "r06_bench_pmc.json", "bf16": "r06_bench_bf16_pmc.json"}


def _kernel_in_build(name: str, lib_bytes: bytes) -> bool:
    """Is a kernel of this (demangled) name compiled into the library of THIS run?  Kernels are templates in an anonymous
    namespace: their mangled symbols carry <length><identifier>."""
    m = re.match(r"(?:void )?\d*([A-Za-z_][A-Za-z0-9_]*?)(?=I[A-Z0-9]|<|\(|$)", name)     # (bf16 names reach rocprofv3 half-mangled: 14gemm_pp_kernelIDF16b...)
    return bool(m) and f"{len(m.group(1))}{m.group(1)}".encode() in lib_bytes


def pmc_summary(dtype, live_avg_us=None):
    """Counter figures of the committed rocprofv3 --pmc passes of THIS command (tools/profile_round.sh -> profiles/<file>) -
    NOT measured by this process: rocprofv3 cannot run inside the bench.  Per kernel class, weighted by launches:
    traffic = FETCH_SIZE x 2 (the gfx950 correction) + WRITE_SIZE in bytes per launch; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES /
    (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).
    The file is a record of an earlier run, so it can go stale: `pmc_stale` is true - and `pmc_stale_why` says what - when
    the file is missing, a kernel class matches no entry of it, a matched kernel's name is not compiled into the library this
    run loaded, or (live_avg_us given: class -> this run's HIP-event microseconds per launch) the file's average launch duration
    of a class differs from this run's by more than a third."""
    out = {"enc_gemm": {}, "cross_attention": {}, "decode_gemm": {}, "source": None, "pmc_stale": False, "pmc_stale_why": []}
    fn = PMC_FILES.get(dtype)
    if fn is None:
        out["pmc_stale_why"].append(f"no counter pass is kept for mode {dtype}")
        return out
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", fn)))
        out["source"] = f"profiles/{fn} (committed rocprofv3 --pmc passes of this command, not this run)"
    except Exception as e:  # noqa: BLE001
        out["pmc_stale"] = True
        out["pmc_stale_why"].append(f"no counter file for mode {dtype} ({fn}): {e!r}")
        return out
    try:
        from embodied_captioning_amd.build import LIB
        lib_bytes = open(LIB, "rb").read()
    except Exception:  # noqa: BLE001
        lib_bytes = None
    # (the demangler of rocprofv3 does not know __bf16: its kernels arrive as `14gemm_pp_kernelIDF16bLb0ELi0E...` or as
    # `gemm_pp_kernel<bool _Accum, bool, E, 0, ...>`; template arguments: operand type, OUT_F32, EPI = 0 (plain store), ...)
    enc = (r"gemm_pp_kernel<g8_t, (true|false), 0," if dtype == "f32s"
           else r"gemm_pp_kernel(IDF16bLb[01]ELi0E|<bool _Accum, bool, E, 0,|<__bf16, (true|false), 0,)")
    classes = {"enc_gemm": enc, "cross_attention": r"decode_attention_(online|shared)_kernel", "decode_gemm": r"gemm_rows_kernel"}
    for cls, pat in classes.items():
        n = b = busy = act = us = nl = 0.0
        names = []
        for k, v in d.items():
            if re.search(pat, k) and "hbm_read_bytes_corrected" in v:
                w = v["launches_per_pass"]
                # a Linear layer whose last tile round is cut runs as TWO launches (256-row tiles, then the 128-row halves of the
                # tail round): bytes and cycles of both count, the layer counts once
                n += 0 if (cls == "enc_gemm" and re.search(r", 128(, (true|false))?>", k)) else w
                b += w * (v["hbm_read_bytes_corrected"] + v.get("hbm_write_bytes", 0.0))
                busy += w * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
                act += w * v.get("GRBM_GUI_ACTIVE", 0.0)
                us += w * v.get("avg_us", 0.0); nl += w
                names.append(k.split("(")[0][:80])
                if lib_bytes is not None and not _kernel_in_build(k, lib_bytes):
                    out["pmc_stale"] = True
                    out["pmc_stale_why"].append(f"{cls}: kernel {k.split('(')[0][:60]} of the counter file is not in this build")
        if not n:
            out["pmc_stale"] = True
            out["pmc_stale_why"].append(f"{cls}: no kernel of the counter file matches /{pat}/")
            continue
        out[cls] = {"traffic": round(b / n), "mfma_busy": round(busy / (act / 8 * 1024), 4) if act else None,
                    "kernel": "; ".join(sorted(set(names)))[:240], "pmc_avg_launch_us": round(us / n, 2)}
        if live_avg_us and live_avg_us.get(cls):
            r = out[cls]["pmc_avg_launch_us"] / live_avg_us[cls]
            out[cls]["pmc_over_live_launch_time"] = round(r, 3)
            if not (0.67 <= r <= 1.5):
                out["pmc_stale"] = True
                out["pmc_stale_why"].append(f"{cls}: {out[cls]['pmc_avg_launch_us']} us per launch in the counter file, "
                                            f"{round(live_avg_us[cls], 2)} us in this run")
    return out


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:  # noqa: BLE001
        pass
    return "unknown"


def cpu_baseline(sd, arch, L, sample):
    """The CPU oracle (PyTorch-CPU restatement of the reference path, parity-locked to HF by tests/golden) timed on this host's
    cores, per SURVEY.md 8(d): config 1's 8 frames and a `sample`-frame (64) batch, greedy max_length L, fp32; one warm-up, then
    the MEDIAN of 5 runs each; torch threads = the cores this process may use.  `value` is the 64-frame figure (the larger
    batch is the faster of the two per caption)."""
    import statistics
    from oracle import blip_ref as R
    torch.set_num_threads(host_cores())
    runs, out = {}, None
    for name, n in (("config1_8_frames", 8), (f"batch_{sample}_frames", sample)):
        px = synthetic_pixels(n, arch.image_size, seed=0)
        R.greedy_generate(sd, arch, px, L)                       # the warm-up run (thread pool, allocator, page-in)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            out = R.greedy_generate(sd, arch, px, L)
            ts.append(time.perf_counter() - t0)
        med = statistics.median(ts)
        runs[name] = {"frames": n, "captions_per_s": round(n / med, 3), "median_s": round(med, 3), "runs_s": [round(t, 3) for t in ts]}
    big, small = runs[f"batch_{sample}_frames"], runs["config1_8_frames"]
    best = max(big, small, key=lambda r: r["captions_per_s"])       # the CPU's better figure is the baseline (small batches fit its caches)
    # beside the port: the third-party implementation the reference's BLIP-family wrappers delegate to - HF transformers'
    # BlipForConditionalGeneration.generate itself (what tests/golden was captured from) - on config 1's 8 frames, same threads
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        from make_goldens import build_hf
        import transformers
        model = build_hf(arch, sd)
        px = synthetic_pixels(8, arch.image_size, seed=0)
        ts = []
        with torch.no_grad():
            model.generate(pixel_values=px, max_length=L, num_beams=1, do_sample=False)
            for _ in range(3):
                t0 = time.perf_counter()
                model.generate(pixel_values=px, max_length=L, num_beams=1, do_sample=False)
                ts.append(time.perf_counter() - t0)
        med = statistics.median(ts)
        runs["hf_transformers_8_frames"] = {"frames": 8, "captions_per_s": round(8 / med, 3), "median_s": round(med, 3), "runs_s": [round(t, 3) for t in ts],
                                            "what": f"transformers {transformers.__version__} BlipForConditionalGeneration.generate (greedy, max_length {L}, "
                                                    f"fp32, same weights and frames): the implementation the reference's wrappers call, for context beside the port"}
        del model
    except Exception as e:  # noqa: BLE001 - context only
        runs["hf_transformers_8_frames"] = {"skipped": repr(e)}
    hf = runs.get("hf_transformers_8_frames", {})
    kind, how = "port", "oracle/blip_ref.py"
    if hf.get("captions_per_s", 0) > best["captions_per_s"]:          # the CPU's best figure is the baseline, whoever produced it
        best, kind, how = hf, "reference", "HF transformers BlipForConditionalGeneration.generate (what the reference's wrappers call)"
    return {"value": best["captions_per_s"], "unit": "captions/s", "cores": host_cores(), "torch_threads": torch.get_num_threads(),
            "os_cpu_count": os.cpu_count(), "cpu_model": cpu_model(), "kind": kind, "baseline_batch": best["frames"],
            "sample": f"config 1 (8 frames) and {sample} frames 224x224, encoder + greedy max_length={L}, fp32, oracle/blip_ref.py on "
                      f"{torch.get_num_threads()} threads: 1 warm-up + median of 5 runs each, and HF transformers' own generate on the 8 "
                      f"frames (median of 3); value = the best of the three: {how}, {best['frames']} frames, median {best['median_s']} s",
            "runs": runs}, out


def main_coca(a):
    """Extra (non-headline) measurement: CoCa ViT-L/14 at --image-size, batch --batch (default 128 here), top-k(1)."""
    from embodied_captioning_amd.config import CocaArch
    from embodied_captioning_amd.weights import procedural_coca_state_dict
    torch.cuda.set_device(0)
    import dataclasses
    arch = dataclasses.replace(CocaArch(), image_size=a.image_size)
    B = a.batch if a.batch != 256 else 128
    sd = procedural_coca_state_dict(arch, 0)
    px = synthetic_pixels(B, arch.image_size, seed=0).cuda()
    K = a.beams
    eng = CaptionerEngine(arch, dtype=a.dtype, max_batch=B, max_beams=K, max_len=arch.seq_len)
    eng.load_state_dict(sd)
    pool = None
    # dynamic batching of the pool (as for BLIP): --coalesce-rows counts IMAGES per merged pass here (each image is K beam rows)
    cr = min(4 * B, 512) if a.coalesce_rows is None else a.coalesce_rows
    coal = cr if (a.streams > 1 and cr > B) else 0
    if a.streams > 1:
        from embodied_captioning_amd.engine import EnginePool
        pool = EnginePool(arch, n=a.streams, dtype=a.dtype, max_batch=max(B, coal), max_beams=K, max_len=arch.seq_len, weights_of=eng)
    dt, _ = timed_steps(pool or eng, px, arch.seq_len, a.steps, a.warmup, 1, lambda i, l: (i, l), K, coalesce=coal)
    if pool is not None:
        pool.close()
    eng.profile(True)
    eng.generate(px, num_beams=K, max_length=arch.seq_len)
    rep = eng.profile_report()
    eng.profile(False)
    tags = [t for t in ENC_GEMM_TAGS if t in rep]
    fl = sum(rep[t]["flops"] for t in tags); ms = sum(rep[t]["ms"] for t in tags)
    S = a.image_size
    how = "top_k=1" if K == 1 else f"beam={K} (reference _generate_beamsearch, one beam group)"
    line = {"metric": f"captions/sec (CoCa ViT-L/14 {S}x{S}, {how}, seq_len=30)", "value": round(B * a.steps / dt, 2),
            "unit": "captions/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "dtype": a.dtype, "data": "synthetic frames, procedural weights",
            "config": {"workload": f"CoCa ViT-L/14 encoder + attentional pooler + 29 KV-cached decode steps x {K} beam(s), {B} frames "
                                   "(SURVEY config 5's model and decode on ONE GPU; parity of this path is unpinned - DESIGN.md section 2)",
                       "streams": a.streams, "beams": K, "coalesce_images": coal},
            "roofline": {"bound": "mfma", "kernel": "gemm_pp_kernel (ViT-L qkv/proj/fc1/fc2)",
                         "achieved": round(fl / (ms * 1e-3) / 1e12, 2), "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s",
                         "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_TFLOPS[a.dtype], 4), "traffic": None},
            "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:12]}}
    print(json.dumps(line))


def main_minilm(a):
    """Extra (non-headline) measurement: the caption-embedding step (all-MiniLM-L6-v2 shapes) on one batch of --batch
    caption-sized token rows (3..24 tokens), with the CPU restatement timed beside it."""
    if a.dtype == "f32s":      # the sentence encoder has no split mode
        a.dtype = "bf16"
    from embodied_captioning_amd.config import MiniLMArch
    from embodied_captioning_amd.engine import TextEncoderEngine
    from embodied_captioning_amd.weights import procedural_minilm_state_dict, synthetic_token_batch
    torch.cuda.set_device(0)
    arch, B, L = MiniLMArch(), a.batch, 24
    sd = procedural_minilm_state_dict(arch, 0)
    ids, lens = synthetic_token_batch(arch, B, L, 0)
    eng = TextEncoderEngine(arch, dtype=a.dtype, max_batch=B, max_len=L)
    eng.load_state_dict(sd)
    idd, lnd = ids.cuda(), lens.cuda()
    for _ in range(a.warmup):
        eng.embed(idd, lnd)
    pool = None
    if a.streams > 1:
        from embodied_captioning_amd.engine import EnginePool, TextEncoderEngine
        pool = EnginePool(arch, n=a.streams, engine_cls=TextEncoderEngine, dtype=a.dtype, max_batch=B, max_len=L)
        pool.load_state_dict(sd)
        pool.run(max(a.warmup, a.streams), idd, lnd, method="embed")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if pool is not None:
        out = pool.run(a.steps, idd, lnd, method="embed")
    else:
        for _ in range(a.steps):
            out = eng.embed(idd, lnd)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if pool is not None:
        pool.close()
    eng.profile(True)
    eng.embed(idd, lnd)
    rep = eng.profile_report()
    eng.profile(False)
    line = {"metric": "caption embeddings/sec (all-MiniLM-L6-v2 shapes, 3..24 tokens)", "value": round(B * a.steps / dt, 1),
            "unit": "sentences/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "dtype": a.dtype, "data": "synthetic token rows, procedural weights",
            "config": {"workload": f"6-layer BERT encoder + mean pool + L2 norm, {B} sentences x {L} padded tokens", "streams": a.streams},
            "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])}}
    if not a.no_cpu_baseline:
        from oracle import minilm_ref as R
        torch.set_num_threads(host_cores())
        R.encode_tokens(sd, arch, ids[:32], lens[:32])
        t0 = time.perf_counter()
        ref = R.encode_tokens(sd, arch, ids, lens)
        cdt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(B / cdt, 1), "unit": "sentences/s", "cores": host_cores(), "kind": "port",
                                "sample": f"{B} sentences, oracle/minilm_ref.py fp32, {cdt:.2f}s wall"}
        line["max_abs_diff_vs_oracle"] = float((out.cpu() - ref).abs().max())
    print(json.dumps(line))


def main_blip2(a):
    """Extra (non-headline) measurement: BLIP-2 OPT-2.7b geometry (the reference's production captioner, blip2.py:19-22),
    batch --batch (default 32 here), greedy, 20 new tokens; seeded weights (3.7 B parameters are drawn on the host first)."""
    from embodied_captioning_amd.config import Blip2Arch
    from embodied_captioning_amd.weights import procedural_blip2_state_dict
    torch.cuda.set_device(0)
    arch = Blip2Arch()
    B = a.batch if a.batch != 256 else 32
    log("drawing 3.7 B seeded parameters on the host ...")
    t0 = time.perf_counter()
    sd = procedural_blip2_state_dict(arch, 0, eos_boost=0.0)
    log(f"... {time.perf_counter() - t0:.0f}s; loading")
    px = synthetic_pixels(B, arch.image_size, seed=0).cuda()
    q8 = bool(getattr(a, "load_in_8bit", False))
    if q8:
        a.dtype = "bf16"
    eng = CaptionerEngine(arch, dtype=a.dtype, max_batch=B, max_beams=1, max_len=arch.max_new_tokens, weight_int8=q8)
    eng.load_state_dict(sd)
    log(f"weights loaded ({eng.device_bytes / 2**30:.1f} GiB on device); timing")
    pool = None
    if a.streams > 1:                       # engines on their own streams over THIS engine's weights
        from embodied_captioning_amd.engine import EnginePool
        pool = EnginePool(arch, n=a.streams, dtype=a.dtype, max_batch=B, max_beams=1, max_len=arch.max_new_tokens, weights_of=eng, weight_int8=q8)
    # the wrapper's dynamic batching default: 4 micro-batches per pass, at most 64 crops (off for int8 micro-batches of <= 4 crops)
    cr = (0 if (q8 and B <= 4) else min(4 * B, 64)) if a.coalesce_rows in (None, 1024) else a.coalesce_rows
    coal = cr if (pool is not None and cr > B) else 0
    if coal:
        pool.close()
        pool = EnginePool(arch, n=a.streams, dtype=a.dtype, max_batch=coal, max_beams=1, max_len=arch.max_new_tokens, weights_of=eng, weight_int8=q8)
    dt, (ids, lens) = timed_steps(pool or eng, px, arch.max_new_tokens, a.steps, a.warmup, 1, lambda i, l: (i, l), coalesce=coal)
    if pool is not None:
        pool.close()
    eng.profile(True)
    eng.generate(px, max_length=arch.max_new_tokens)
    rep = eng.profile_report()
    eng.profile(False)
    tags = [t for t in ENC_GEMM_TAGS if t in rep]
    fl = sum(rep[t]["flops"] for t in tags); ms = sum(rep[t]["ms"] for t in tags)
    line = {"metric": "captions/sec (BLIP-2 OPT-2.7b geometry, 224x224, greedy, 20 new tokens)", "value": round(B * a.steps / dt, 2),
            "unit": "captions/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "dtype": a.dtype + ("+int8w" if q8 else ""), "data": "synthetic frames, procedural weights",
            "config": {"workload": f"ViT-g/14 + Q-Former + OPT-2.7b prefill (33 positions) + 19 cached decode steps, {B} frames", "streams": a.streams, "coalesce_crops": coal,
                       "load_in_8bit": q8, "device_GiB": round(eng.device_bytes / 2**30, 2)},
            "roofline": {"bound": "mfma", "kernel": "gemm_pp_kernel (ViT-g qkv/proj/fc1/fc2)",
                         "achieved": round(fl / (ms * 1e-3) / 1e12, 2), "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s",
                         "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_TFLOPS[a.dtype], 4), "traffic": None},
            "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:16]}}
    if not a.no_cpu_baseline:
        from oracle import blip2_ref as R
        torch.set_num_threads(host_cores())
        n = min(2, B)
        t0 = time.perf_counter()
        ref = R.greedy_generate(R.int8_state_dict(sd) if q8 else sd, arch, px[:n].cpu())
        cdt = time.perf_counter() - t0
        new = ref["sequences"][:, arch.num_query_tokens + 1:]
        ours = ids[:n, : new.shape[1]].cpu()
        t2 = torch.topk(torch.stack(ref["logits"], 0), 2, dim=-1).values
        line["cpu_baseline"] = {"value": round(n / cdt, 3), "unit": "captions/s", "cores": host_cores(), "kind": "port",
                                "sample": f"{n} captions, oracle/blip2_ref.py fp32, {cdt:.1f}s wall"}
        line["parity"] = {"rows": n, "tokens_equal": int((ours == new).sum()), "tokens": int(new.numel()),
                          "min_margin_of_oracle_path": float((t2[..., 0] - t2[..., 1]).min())}
    print(json.dumps(line))


def extra_mode(arch, sd, px, L, B, dev, dtype, streams, ref_ids, golden, cross_cache="auto", coalesce=0):
    """One of the non-headline arithmetic modes on the same workload: pooled timed steps, the encoder-GEMM roofline of one
    engine, token agreement with the headline run and with the HF golden.  cross_cache="fp32": the split mode with fp32
    cross-attention K/V rows instead of KV16 (the encoder is the headline's: no roofline pass)."""
    from embodied_captioning_amd.engine import EnginePool
    kw = {"cross_cache": cross_cache} if cross_cache != "auto" else {}
    eng = CaptionerEngine(arch, dtype=dtype, max_batch=B, max_beams=1, max_len=L, device=dev, **kw)
    eng.load_state_dict(sd)
    coalesce = coalesce if streams > 1 else 0
    runner = EnginePool(arch, n=streams, device=dev, dtype=dtype, max_batch=max(B, coalesce), max_beams=1, max_len=L, weights_of=eng, **kw) if streams > 1 else eng
    steps = 12 if dtype != "f32" else 2
    dt, (ids, _) = timed_steps(runner, px, L, steps, 3 if dtype != "f32" else 1, 1, lambda i, l: (i, l), coalesce=coalesce)
    if streams > 1:
        runner.close()
    out = {"value": round(B * steps / dt, 2), "unit": "captions/s", "ms_per_step": round(1e3 * dt / steps, 3), "streams": streams,
           "coalesce_rows": coalesce, "cross_cache": eng.cross_cache_kind}
    if cross_cache == "auto":
        out["roofline"] = roofline_pass(eng, px, L, dtype, arch, B)[0]
    out["rows_identical_to_headline"] = round(float((ids == ref_ids).all(dim=1).float().mean().item()), 4)
    if golden is not None:
        out["parity"] = golden_parity(ids, golden, arch, L, B, 0.3 if dtype == "bf16" else 0.0)
    eng.close()
    return out


def golden_parity(ids, g, arch, L, B, tau):
    """Greedy tokens against the committed HF greedy loop (tests/golden/blip_base256.npz: the whole 256-frame batch, same
    seeds).  tau = 0:
    every differing row counts as a mismatch (the bar of the fp32-grade modes); bf16 is judged with the near-tie rule."""
    from tests._util import pad_to, token_parity
    ref = pad_to(g["greedy_sequences"], L, arch.pad)
    n = min(ref.shape[0], B)
    exact, div, bad = token_parity(ids[:n].cpu().numpy(), ref[:n], g["greedy_margin"][:, :n], tau)
    return {"rows": int(n), "token_identical_rows": int(exact), "diverged_at_near_tie": int(div) if tau > 0 else 0,
            "mismatched_rows": int(div) if tau == 0 else (0 if bad is None else 1), "near_tie_margin": tau,
            "reference": "HF transformers 5.15 BlipForConditionalGeneration greedy, fp32 CPU (tests/golden/blip_base256.npz)"}


LAYER_STEP_TAGS = ("dec_gemm_qkv", "dec_self_attn", "dec_gemm_so", "dec_reduce_ln", "dec_reduce_ln_wave", "dec_gemm_cq", "dec_cross_attn", "dec_gemm_co",
                   "dec_gemm_f1", "dec_gemm_f2", "dec_small_qkv", "dec_small_so", "dec_small_cross", "dec_small_co", "dec_small_f1",
                   "dec_small_f2")


def latency_block(arch, sd, dev, dtype, L, batches=(1, 8, 64), reps=7):
    """The reference's real call pattern: ONE crop per call (coca.py:27-33, blip2.py:24-29, goal_exploration.py:95-105,
    pseudolabeler.py:673-676); BASELINE config 1 is 8 crops.  Per batch size: median wall time of one `cap_generate` on one
    stream (frames resident, host synchronised after each call), with HF's stopping rule off (all L-1 steps) and on (the plugin's
    poll every 4 steps), and how many kernel launches one decoder layer-step takes (per-tag launch counts of the library's own
    profile pass over the tags of one text layer / (layers x steps)).  `forward_pil_ms`: `BLIP.forward(PIL.Image)` end to end -
    resize on the host, upload, generate with per-step logits, detokenise - at batch 1."""
    import statistics
    out = {"unit": "ms", "max_length": L, "dtype": dtype, "reps": reps, "batches": {}}
    for B in batches:
        eng = CaptionerEngine(arch, dtype=dtype, max_batch=B, max_beams=1, max_len=L, device=dev)
        eng.load_state_dict(sd)
        px = synthetic_pixels(B, arch.image_size, seed=0).to(dev)
        rec = {}
        for key, poll in (("generate_ms", 0), ("generate_early_exit_ms", 4)):
            eng.set_early_exit(poll)
            for _ in range(2):
                eng.generate(px, max_length=L)
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                o = eng.generate(px, max_length=L)
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - t0))
            rec[key] = round(statistics.median(ts), 3)
            rec[key.replace("_ms", "_steps")] = eng.last_decode_steps
        eng.set_early_exit(0)
        rec["mean_caption_tokens"] = round(float(o["lengths"].float().mean()), 2)
        eng.profile(True)
        eng.generate(px, max_length=L)
        rep = eng.profile_report()
        eng.profile(False)
        n = sum(r["launches"] for t, r in rep.items() if t in LAYER_STEP_TAGS)
        rec["launches_per_layer_step"] = round(n / (arch.t_layers * (L - 1)), 2)
        rec["tagged_launches_per_generate"] = sum(r["launches"] for r in rep.values())
        rec["decode_kernel_ms"] = round(sum(r["ms"] for t, r in rep.items() if t.startswith("dec_") or t == "greedy_select"), 3)
        rec["image_side_kernel_ms"] = round(sum(r["ms"] for t, r in rep.items() if not (t.startswith("dec_") or t == "greedy_select")), 3)
        rec["ms_per_caption"] = round(rec["generate_ms"] / B, 3)
        out["batches"][str(B)] = rec
        eng.close()
    try:
        from PIL import Image
        from embodied_captioning_amd.captioner.models.blip.blip import BLIP
        from embodied_captioning_amd.captioner.utils.utils import CaptionerField
        model = BLIP(CaptionerField(arch_name="blip", model_name="procedural:0:9", height=arch.image_size, width=arch.image_size,
                                    dtype=dtype, batch_size=1, max_length=L, device=str(dev)))
        rng = np.random.default_rng(0)
        im = Image.fromarray(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8))
        for _ in range(2):
            model.forward(im)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            model.forward(im)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        out["forward_pil_ms"] = round(statistics.median(ts), 3)
        out["forward_pil_note"] = ("BLIP.forward(PIL 640x480): upload + Pillow-exact bicubic resize on the device + cap_generate with per-step logits "
                                   "(early-exit poll 4, the plugin default) + detokenise")
        model.engine.close()
    except Exception as e:  # noqa: BLE001
        out["forward_pil_error"] = repr(e)
    return out


def main_strong(a, arch, sd, dev, rank, world):
    """Strong scaling (north_star: >= 6x at 8 GPUs; SURVEY config 4): --frames in total, contiguous shards, micro-batches of
    --batch rotating over the stream pool, ONE caption all-gather at the end, consensus grouping on rank 0's table."""
    from embodied_captioning_amd.distributed import shard_range, strong_scaling_job
    from embodied_captioning_amd.engine import EnginePool
    L, B = a.max_length, a.batch
    pool = EnginePool(arch, n=max(a.streams, 1), device=dev, dtype=a.dtype, max_batch=B, max_beams=1, max_len=L)
    pool.load_state_dict(sd)
    gen = torch.Generator(device=dev)

    def frames_of(first, count):       # raw RGB frames made on the device from the first frame index
        gen.manual_seed(1_000_003 * first + 17)
        return torch.randint(0, 256, (count, arch.image_size, arch.image_size, 3), dtype=torch.uint8, device=dev, generator=gen)

    pool.generate_many([frames_of(0, B)] * len(pool), threads=True, max_length=L)      # warm-up: every engine once
    job = strong_scaling_job(lambda f: pool.submit(f, max_length=L), frames_of, a.frames, B, L, arch.pad, join=pool.join,
                             keys_of=lambda i: (i // 500, (i // 10) % 50), device=dev,
                             range_check=(lambda: pool.engines[0].saturations()) if a.dtype == "f32s" else None)
    dt = job["seconds"]
    if rank == 0:
        first, last, per = shard_range(a.frames, 0, world)
        S = arch.image_size
        print(json.dumps({"metric": f"captions/sec ({S}x{S}, beam=1)", "value": round(a.frames / dt, 2), "unit": "captions/s",
                          "n_gpus": world, "steps": (per + B - 1) // B, "warmup": len(pool), "ms_per_step": round(1e3 * dt / ((per + B - 1) // B), 3),
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": a.dtype,
                          "data": "synthetic frames made on the device (seed = first frame index of the micro-batch), procedural weights",
                          "config": {"workload": f"BLIP-base encoder + greedy decode over {a.frames} frames in total, contiguous shards of "
                                                 f"{per}, micro-batches of {B} on {len(pool)} streams, one caption all-gather, consensus grouping",
                                     "global_batch": world * B, "parallelism": f"dp{world}", "streams": len(pool), "frames": a.frames},
                          "job_s": round(dt, 3), "grouping_s": round(job["grouping_s"], 3), "objects": job["objects"],
                          "range_clamps": job.get("range_clamps"), "mean_caption_tokens": round(job["mean_caption_tokens"], 2)}))
    pool.close()


def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` outside a launcher: start the N ranks as ONE child process - `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` - wait for it and
    return its exit code.  This process has made no GPU call (parse() and imports only) and makes none: it never replaces
    itself with another program, the ranks are children."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log(f"--gpus {a.gpus} without WORLD_SIZE: starting the ranks as a child process: {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on these hosts
    return subprocess.run(cmd, env=env).returncode


class StubEngine:
    """--stub-engine: a captioner whose ids depend on the frame index only (CPU; plumbing tests)."""

    def __init__(self, L):
        self.L = L
        self.last_decode_steps = L - 1

    def generate(self, frames, num_beams=1, max_length=None):
        f = frames.to(torch.int32).reshape(-1)
        ids = torch.stack([(f * 7 + j) % 1000 for j in range(self.L)], dim=1).int()
        return {"sequences": ids, "lengths": (f % (self.L - 1) + 2).int()}

    def saturations(self):
        return 0


def strong_figure(a, arch, runner, dev, world, L, B, stub=False):
    """The strong-scaling job beside the weak-scaling value (north_star's >= 6x at 8 GPUs is a STRONG-scaling target; SURVEY config
    4): --frames in total over all ranks, contiguous shards, micro-batches of --batch on the engines the timed steps used, ONE
    caption all-gather, consensus grouping on rank 0.  Every rank calls this (collectives inside); rank 0 gets the dict."""
    from embodied_captioning_amd.distributed import shard_range, strong_scaling_job
    pool = runner if hasattr(runner, "submit") else None
    if stub:
        def frames_of(first, count):
            return torch.arange(first, first + count, dtype=torch.int32)
    else:
        gen = torch.Generator(device=dev)

        def frames_of(first, count):       # raw RGB frames made on the device from the first frame index
            gen.manual_seed(1_000_003 * first + 17)
            return torch.randint(0, 256, (count, arch.image_size, arch.image_size, 3), dtype=torch.uint8, device=dev, generator=gen)
    submit = (lambda f: pool.submit(f, max_length=L)) if pool is not None else (lambda f: runner.generate(f, max_length=L))
    eng0 = pool.engines[0] if pool is not None else runner
    job = strong_scaling_job(submit, frames_of, a.frames, B, L, 0 if stub else arch.pad, join=pool.join if pool is not None else None,
                             keys_of=lambda i: (i // 500, (i // 10) % 50), device=None if stub else dev,
                             range_check=(lambda: eng0.saturations()) if (stub or a.dtype == "f32s") else None)
    if not job.get("objects"):
        return None
    first, last, per = shard_range(a.frames, 0, world)
    return {"frames": a.frames, "n_gpus": world, "captions_per_s": round(a.frames / job["seconds"], 2), "job_s": round(job["seconds"], 3),
            "frames_per_rank": per, "micro_batch": B, "grouping_s": round(job["grouping_s"], 3), "objects": job["objects"],
            "range_clamps": job.get("range_clamps"), "mean_caption_tokens": round(job["mean_caption_tokens"], 2),
            "scaling": "strong", "note": "a FIXED total of frames sharded contiguously over the ranks, one caption all-gather at the end, "
                                         "consensus grouping on rank 0; divide by the 1-GPU line's figure for the strong-scaling speed-up"}


def main_stub(a, dev, rank, world):
    """--stub-engine: the weak-scaling steps and the strong-scaling job of main() with StubEngine on CPU ranks (gloo)."""
    from embodied_captioning_amd.distributed import make_step_gather
    L, B = a.max_length, a.batch
    eng = StubEngine(L)
    px = torch.arange(rank * B, rank * B + B, dtype=torch.int32)
    gather = make_step_gather(world, B, L, "cpu")
    dt, (ids, lens) = timed_steps(eng, px, L, a.steps, a.warmup, world, gather, 1, dev="cpu")
    want = eng.generate(torch.arange(world * B, dtype=torch.int32))
    assert torch.equal(ids, want["sequences"]) and torch.equal(lens, want["lengths"]), "gathered records are not rank-major"
    strong = None if a.no_strong else strong_figure(a, None, eng, dev, world, L, B, stub=True)
    if rank == 0:
        print(json.dumps({"metric": "captions/sec (stub)", "value": round(world * B * a.steps / dt, 2), "unit": "captions/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True,
                          "scaling": "weak", "stub": True, "strong_scaling": strong}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def finish_line(line):
    """The figures that qualify the headline, in two places a truncating reader still sees: inside `config` (`also`) and as the
    LAST key of the line (`summary`) - numbers only; the keys they come from carry the details."""
    def pick(key, *path):
        v = line.get(key)
        for k in path:
            v = v.get(k) if isinstance(v, dict) else None
        return v
    par = line.get("parity") or {}
    also = {"value_batch256_passes": pick("pool_uncoalesced", "value"), "single_stream": pick("single_stream", "value"),
            "strong_scaling_captions_per_s": pick("strong_scaling", "captions_per_s"),
            "strong_scaling_at_micro_batch_256": pick("strong_scaling", "at_micro_batch_256", "captions_per_s"),
            "golden_rows_identical": f"{par.get('token_identical_rows')}/{par.get('rows')}" if par else None,
            "enc_gemm_frac": pick("roofline", "frac"), "enc_gemm_frac_executed": pick("roofline", "frac_executed"),
            "enc_gemm_executed_tflops": pick("roofline", "executed_tflops"), "vendor_fp16_gemm_tflops_same_shapes": pick("roofline", "vendor_fp16_gemm_tflops_same_shapes"),
            "decode_gemm_hbm_frac": pick("decode", "gemm", "frac"), "cross_attention_hbm_frac": pick("decode", "cross_attention", "frac"),
            "decode_kernels_ms_per_pass": pick("decode", "ms_per_step_all_decode_kernels"),
            "decode_ms_per_256_frames_at_pass_rows": pick("pass_rows_profile", "decode_ms_per_256_frames"),
            "image_side_ms_per_256_frames_at_pass_rows": pick("pass_rows_profile", "image_side_ms_per_256_frames"),
            "encoder_only_images_per_s": pick("encoder_only", "images_per_s"),
            "f32s_fp32kv": pick("f32s_fp32kv", "value"), "f32_exact": pick("f32_exact", "value"), "bf16_not_parity": pick("bf16", "value"),
            "cpu_baseline": pick("cpu_baseline", "value"),
            "cpu_hf_transformers_8_frames": pick("cpu_baseline", "runs", "hf_transformers_8_frames", "captions_per_s")}
    also = {k: v for k, v in also.items() if v is not None}
    line["config"]["also"] = also
    line["summary"] = dict(also, value=line["value"], ms_per_step=line["ms_per_step"], pass_rows=line["config"].get("pass_rows"))
    return line


def main():
    a = parse()
    if a.dtype is None:
        a.dtype = {"blip": "f32s", "blip2": "f32s", "coca": "bf16", "minilm": "bf16"}[a.model]
    if a.coalesce_rows is None and a.model != "coca":
        a.coalesce_rows, a.coalesce_defaulted = 1024, True
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if a.model != "blip":
            raise SystemExit(f"--gpus {a.gpus}: only the blip workload (BASELINE's metric) runs on several ranks")
        sys.exit(spawn_ranks(a))                 # before anything touches the GPU: the ranks are children of this process
    if a.model == "coca":
        return main_coca(a)
    if a.model == "blip2":
        return main_blip2(a)
    if a.model == "minilm":
        return main_minilm(a)
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does)")
    if a.share_gpu:
        local = 0
    if a.stub_engine:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.stub_engine or a.share_gpu:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)
        world = torch.distributed.get_world_size()      # n_gpus of the line = the ranks the process group actually has
        rank = torch.distributed.get_rank()
    if a.stub_engine:
        return main_stub(a, dev, rank, world)

    global VENDOR_CONTEXT
    VENDOR_CONTEXT = not (a.no_extra_modes or a.no_strict or a.lite)
    arch = BlipArch()
    arch.image_size = a.image_size
    L, B = a.max_length, a.batch
    sd = procedural_blip_state_dict(arch, 0, eos_boost=a.eos_boost)       # 9.0: same weights as tests/golden/blip_base.npz
    if a.strong:
        main_strong(a, arch, sd, dev, rank, world)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    if a.latency_only:
        print(json.dumps({"latency": latency_block(arch, sd, dev, a.dtype, L)}))
        return
    px = synthetic_pixels(B, arch.image_size, seed=0, first=rank * B).to(dev)

    from embodied_captioning_amd.distributed import make_step_gather
    gather = make_step_gather(world, B, L, dev)       # the per-step caption all-gather into buffers allocated once

    log(f"rank {rank}/{world}: weights + {B} frames ready, host cores usable: {host_cores()}")
    eng = CaptionerEngine(arch, dtype=a.dtype, max_batch=B, max_beams=a.beams, max_len=L, device=dev)
    eng.load_state_dict(sd)
    eng.set_early_exit(a.early_exit)
    eng.set_decode_path(a.decode_path)
    eng.set_row_compaction(a.row_compaction == "on")
    runner = eng
    # dynamic batching only where it is the plain greedy workload on a pool (beams / early exit / other sizes: extra lines as before)
    coal = a.coalesce_rows if (a.streams > 1 and not a.early_exit and a.coalesce_rows > B) else 0
    if a.beams > 1 and getattr(a, "coalesce_defaulted", False):
        coal = min(coal, 8 * B)                # (config 3's batch is 64 images: passes of up to 512 images x beams - 24 steps: 4 080 captions/s
                                               # against 3 930 with 256-image passes; --coalesce-rows overrides)
    from embodied_captioning_amd.engine import EnginePool
    if a.streams > 1:
        runner = EnginePool(arch, n=a.streams, device=dev, dtype=a.dtype, max_batch=max(B, coal), max_beams=a.beams, max_len=L, weights_of=eng)
        runner.set_early_exit(a.early_exit)
        runner.set_decode_path(a.decode_path)
        runner.set_row_compaction(a.row_compaction == "on")
    log(f"weights loaded once, {a.streams} engine(s) / stream(s) on them; timing ({a.dtype})")
    dt, (ids, lens) = timed_steps(runner, px, L, a.steps, a.warmup, world, gather, a.beams, coalesce=coal)
    uncoalesced = None
    if coal and a.streams > 1 and not a.lite:
        # the same steps with every batch as its own pass (rounds 1-4's headline), for the record
        udt, _ = timed_steps(runner, px, L, a.steps, a.streams, world, gather, a.beams)
        uncoalesced = {"value": round(world * B * a.steps / udt, 2), "unit": "captions/s", "ms_per_step": round(1e3 * udt / a.steps, 3),
                       "steps": a.steps, "streams": a.streams}
    decode_steps = (runner.engines[0] if a.streams > 1 else eng).last_decode_steps
    log(f"timed region: {dt:.3f}s for {a.steps} steps (max over ranks)")
    strong = None
    if not (a.no_strong or a.lite or a.beams > 1 or arch.image_size != 224 or a.early_exit):
        log(f"strong-scaling job: {a.frames} frames in total over {world} rank(s)")
        strong = strong_figure(a, arch, runner, dev, world, L, max(B, coal))      # micro-batches of the merged passes' size
        if coal > B:                             # and at SURVEY config 4's own micro-batch (256), every micro-batch its own pass
            s256 = strong_figure(a, arch, runner, dev, world, L, B)
            if strong is not None and s256 is not None:
                strong["at_micro_batch_256"] = {k: s256[k] for k in ("captions_per_s", "job_s", "micro_batch", "objects")}
    if a.streams > 1:
        runner.close()

    if rank == 0:
        value = world * B * a.steps / dt
        S = arch.image_size
        mode = {"f32s": "f32 carried as split fp16: every GEMM operand = two fp16 halves, 3 fp16 MFMA products per MAC, fp32 accumulate; "
                        "cross-attention K/V cache = KV16 (int16 + one fp32 scale per 64-wide head row: 15 value bits); LayerNorm / softmax / "
                        "residual stream / self-attention cache fp32.  The exact-product fp32 MFMA mode of the same run is the `f32_exact` key",
                "bf16": "bf16 operands and K/V caches, fp32 accumulate / LayerNorm / softmax / residual stream", "f32": "f32 (fp32 MFMA, exact products)"}[a.dtype]
        short = {"f32s": "f32s (fp32 as split fp16: 3 fp16 MFMA per MAC, fp32 accumulate; KV16 cross cache)", "bf16": "bf16", "f32": "f32"}[a.dtype]
        untimed = (max(a.warmup, a.streams * max(1, coal // B)) if a.streams > 1 else a.warmup)
        passes = EnginePool.coalesce_plan([B] * a.steps, a.streams, coal) if (coal and a.streams > 1) else None
        pass_rows = sorted({len(gp) * B for gp in passes}, reverse=True) if passes else [B]
        line = {"metric": f"captions/sec ({S}x{S}, beam={a.beams})", "value": round(value, 2), "unit": "captions/s",
                "n_gpus": world, "steps": a.steps, "warmup": untimed, "ms_per_step": round(1e3 * dt / a.steps, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": short,
                "data": "synthetic frames (PCG64, seed = frame index), procedural weights (no checkpoint offline)",
                "config": {"workload": f"BLIP-base ViT-B/16 encoder + {'greedy' if a.beams == 1 else f'beam-{a.beams}'} decode, {B} frames/GPU per step {S}x{S}, "
                                       f"max_length={L}, caption all-gather"
                                       + (f"; {a.steps} steps merged by the engine pool into {len(passes)} passes of {'/'.join(str(r) for r in pass_rows)} rows" if passes else ""),
                           "global_batch": world * B, "pass_rows": pass_rows,
                           "parallelism": f"dp{world}" + (" REHEARSAL: all ranks share cuda:0 over gloo (--share-gpu) - not a scaling figure" if a.share_gpu else ""), "streams": a.streams, "compute_mode": a.dtype, "compute_mode_note": mode,
                           "value_is": (f"consecutive batches overlapped on {a.streams} engines / HIP streams of one GPU (EnginePool)"
                                        + (f", the pool's dynamic batching merging consecutive steps' batches into passes of up to {coal} rows "
                                           f"(a frame has the same bits alone, in its batch and in a merged pass); every batch as its own pass is the "
                                           f"`pool_uncoalesced` key" if coal else "")
                                        + "; one batch at a time on one stream is the `single_stream` key") if a.streams > 1 else "one batch at a time on one stream",
                           "coalesce_rows": coal, "row_compaction": a.row_compaction,
                           "warmup_requested": a.warmup, "untimed_steps_run": untimed}}
        if uncoalesced:
            line["pool_uncoalesced"] = uncoalesced
        ln = lens[:B].float()
        line["caption_tokens"] = {"mean": round(float(ln.mean()), 2), "max": int(ln.max()), "of": L}
        if strong:
            line["strong_scaling"] = strong
        pw = getattr(timed_steps, "power", None)
        pw = pw.result(B * a.steps, dt) if pw is not None else None      # rank 0's GPU, its own captions
        if pw:
            line["power"] = pw
        if a.early_exit or a.eos_boost != 9.0:
            line["config"]["early_exit_poll"] = a.early_exit
            line["config"]["eos_boost"] = a.eos_boost
            line["decode_steps_run"] = decode_steps
        if a.beams > 1:                      # config 3 extra line: per-kernel profile of one beam generate, then stop
            eng.profile(True)
            eng.generate(px, num_beams=a.beams, max_length=L)
            rep = eng.profile_report()
            eng.profile(False)
            line["kernels_ms"] = {k: round(v["ms"], 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:14]}
        if a.lite or a.beams > 1 or S != 224:
            print(json.dumps(line))
            eng.close()
            if world > 1:                    # the other ranks are waiting at the closing barrier below
                torch.distributed.barrier()
                torch.distributed.destroy_process_group()
            return
        if a.streams > 1:                    # the timed steps ran on the pool's engines: this one has not touched its arena yet
            eng.generate(px, num_beams=1, max_length=L)
            torch.cuda.synchronize()
        if a.streams > 1:
            # one batch at a time on one stream: what a caller without the pool gets, and the basis of the per-kernel figures
            # below (kernel_ms_per_step sums ONE stream's kernel durations; the pooled ms_per_step is shorter because kernels
            # of different batches co-run)
            sdt, _ = timed_steps(eng, px, L, a.steps, 1, 1, lambda i, l: (i, l), a.beams)
            line["single_stream"] = {"value": round(B * a.steps / sdt, 2), "unit": "captions/s", "ms_per_step": round(1e3 * sdt / a.steps, 3),
                                     "steps": a.steps, "streams": 1}
        roof, kernels, kernel_ms, dec = roofline_pass(eng, px, L, a.dtype, arch, B)
        log(f"roofline pass done: {roof['achieved']} TFLOP/s as 2MNK/t on the encoder GEMMs ({roof['executed_tflops']} executed)")
        line["roofline"] = roof
        line["decode"] = dec
        line["kernels"] = {k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                           for k, v in kernels.items()}
        line["kernel_ms_per_step"] = round(kernel_ms, 3)           # sum of ONE stream's kernel durations (see single_stream)
        if coal > B and a.streams > 1:
            # the same per-kernel pass at the size of the merged passes the timed steps ran (one engine, one stream): where a
            # 1024-row pass goes.  The projections are no longer a weight stream at this size: their bytes per
            # second are reported for continuity, the bound is the CUs' L2 -> LDS fill rate
            k = coal // B
            big = CaptionerEngine(arch, dtype=a.dtype, max_batch=k * B, max_beams=1, max_len=L, device=dev, share_weights_with=eng)
            big.set_row_compaction(a.row_compaction == "on")
            pxk = torch.cat([px] * k)
            for _ in range(2):                       # a fresh 1024-row arena: its first passes touch new memory
                big.generate(pxk, num_beams=1, max_length=L)
            torch.cuda.synchronize()
            _, kern_k, total_k, dec_k = roofline_pass(big, pxk, L, a.dtype, arch, k * B)
            big.close()
            enc_ms = sum(v["ms_per_step"] for t, v in kern_k.items() if not (t.startswith("dec_") or t in ("greedy_select", "beam_step")))
            line["pass_rows_profile"] = {
                "rows": k * B, "kernel_ms_per_pass": round(total_k, 3), "image_side_ms_per_256_frames": round(enc_ms / k, 3),
                "decode_ms_per_256_frames": round(dec_k["ms_per_step_all_decode_kernels"] / k, 3),
                "decode_gemm": {x: dec_k["gemm"][x] for x in ("achieved", "frac", "avg_launch_us", "ms_per_step")},
                "cross_attention": {x: dec_k["cross_attention"][x] for x in ("achieved", "frac", "avg_launch_us", "ms_per_step")},
                "top_kernels_ms": {t: round(v["ms_per_step"], 3) for t, v in sorted(kern_k.items(), key=lambda kv: -kv[1]["ms_per_step"])[:12]}}
        line["encoder_only"] = encoder_only(eng, px, arch)
        eng.close()
        if not a.no_latency:
            log("latency block: B = 1, 8, 64")
            line["latency"] = latency_block(arch, sd, dev, a.dtype, L)
        golden = None
        try:
            from tests._util import load_golden
            golden, _, _ = load_golden("blip_base256")              # 256 rows of the real HF greedy loop, same seeds
            line["parity"] = golden_parity(ids, golden, arch, L, B, 0.3 if a.dtype == "bf16" else 0.0)
        except Exception as e:  # noqa: BLE001
            line["parity"] = {"error": repr(e)}
        if world == 1 and not (a.no_strict or a.no_extra_modes):
            for other, key in (("bf16", "bf16"), ("f32", "f32_exact"), ("f32s", "f32s")):
                if other == a.dtype:
                    continue
                if a.dtype != "f32s" and other == "f32":
                    continue
                log(f"extra mode: {other}")
                line[key] = extra_mode(arch, sd, px, L, B, dev, other, a.streams if other != "f32" else 1, ids, golden, coalesce=coal)
            if a.dtype == "f32s":
                # the headline's arithmetic with fp32 cross-attention K/V rows (CapConfig.cross_kv_fp32): what the split mode costs
                # when the checkpoint's K/V heads are refused for KV16 - and the line's figure without the 15-bit cache
                log("extra mode: f32s with fp32 cross-attention K/V rows")
                line["f32s_fp32kv"] = extra_mode(arch, sd, px, L, B, dev, "f32s", a.streams, ids, golden, cross_cache="fp32", coalesce=coal)
        if world == 1 and not a.no_cpu_baseline:
            log(f"cpu baseline: {a.cpu_sample} captions on {host_cores()} host threads")
            cb, _ = cpu_baseline(sd, arch, L, a.cpu_sample)
            line["cpu_baseline"] = cb
            line["vs_cpu_baseline"] = round(value / cb["value"], 1)       # context only: the roofline fractions judge the kernels
        print(json.dumps(finish_line(line)))
    else:
        eng.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
