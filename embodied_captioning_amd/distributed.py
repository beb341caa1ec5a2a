"""Multi-GPU data path of the captioner: frames shard across ranks, each rank runs a full replica, and ONE collective
(RCCL all-gather over xGMI; gloo in CPU tests) collects fixed-shape caption records for the consensus step.

Reference analogue: ``experimenting_env/utils/train_helpers.py:218-246`` (`collect_results_gpu`: two all-gathers of
pickled, zero-padded bytes; never called in the reference, which therefore never merges per-rank captions -
SURVEY.md F8).  Here a record is plain integers, so no size exchange and no pickle are needed:

    ids  int32 [n_pad, max_len]   token ids incl. BOS (rows past the shard's end: all `pad`, length 0)
    lens int32 [n_pad]            tokens per row (0 = padding row)

Consensus grouping mirrors ``experimenting_env/captioner/pseudocaptioner.py:125-177`` (`group_captions`,
`compute_captions_frequency`, banned-word filter :96-123) on the gathered table.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Contiguous, equal-size shards: returns (first, last_exclusive, padded_shard_size).  Every rank gets
    ceil(n/world) slots; the tail shard is padded with sentinel rows so the gather has a fixed shape."""
    per = (n_items + world - 1) // world
    first = min(rank * per, n_items)
    last = min(first + per, n_items)
    return first, last, per


def init_distributed(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment; one process per GPU, backend "nccl" (= RCCL on ROCm)."""
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, local, world


def gather_caption_records(ids: torch.Tensor, lens: torch.Tensor, n_pad: int, pad_id: int = 0):
    """All-gather one shard's records.  ids int32 [n_local, L], lens int32 [n_local] with n_local <= n_pad.
    Returns (ids_all [world*n_pad, L], lens_all [world*n_pad]) on every rank, rank-major (= global frame order
    for contiguous shards)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    n_local, L = ids.shape
    if n_local < n_pad:
        ids = torch.cat([ids, torch.full((n_pad - n_local, L), pad_id, dtype=ids.dtype, device=ids.device)])
        lens = torch.cat([lens, torch.zeros(n_pad - n_local, dtype=lens.dtype, device=lens.device)])
    if world == 1:
        return ids, lens
    ids = ids.contiguous()
    lens = lens.contiguous()
    ids_all = torch.empty((world * n_pad, L), dtype=ids.dtype, device=ids.device)
    lens_all = torch.empty((world * n_pad,), dtype=lens.dtype, device=lens.device)
    dist.all_gather_into_tensor(ids_all, ids)
    dist.all_gather_into_tensor(lens_all, lens)
    return ids_all, lens_all


def _sync(device) -> None:
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def make_step_gather(world: int, rows: int, max_len: int, device) -> Callable[[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]:
    """The per-step collective of the weak-scaling bench: every rank contributes `rows` caption records (ids int32 [rows,
    max_len], lens int32 [rows]) and receives all world * rows of them, rank-major, in buffers allocated ONCE here (no
    allocation inside the timed region).  world == 1: the records themselves.  Reference analogue: train_helpers.py:218-246."""
    if world <= 1:
        return lambda ids, lens: (ids, lens)
    ids_all = torch.empty((world * rows, max_len), dtype=torch.int32, device=device)
    len_all = torch.empty((world * rows,), dtype=torch.int32, device=device)

    def gather(ids: torch.Tensor, lens: torch.Tensor):
        if tuple(ids.shape) != (rows, max_len) or tuple(lens.shape) != (rows,):
            raise ValueError(f"step gather was built for [{rows}, {max_len}] records, got {tuple(ids.shape)} / {tuple(lens.shape)}")
        dist.all_gather_into_tensor(ids_all, ids.contiguous())
        dist.all_gather_into_tensor(len_all, lens.contiguous())
        return ids_all, len_all
    return gather


def timed_region(run: Callable[[], object], world: int, device=None):
    """The bench contract's bracket: barrier + device synchronise, `run()`, device synchronise + barrier, wall time; then the
    MAX over ranks (one all-reduce) - every rank returns (seconds of the slowest rank, run()'s result)."""
    import time
    if world > 1:
        dist.barrier()
    _sync(device)
    t0 = time.perf_counter()
    res = run()
    _sync(device)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, res


def strong_scaling_job(submit: Callable[[torch.Tensor], Dict[str, torch.Tensor]], frames_of: Callable[[int, int], torch.Tensor],
                       n_frames: int, micro_batch: int, max_len: int, pad_id: int = 0, join: Callable[[], None] | None = None,
                       keys_of: Callable[[int], Tuple[int, int]] | None = None, detokenise: Callable[[Sequence[int]], str] | None = None,
                       device=None, range_check: Callable[[], int] | None = None) -> Dict[str, object]:
    """SURVEY config 4 / north_star's strong-scaling shape as ONE function: a FIXED total of frames, contiguous shards
    (`caption_shard`), ONE caption all-gather at the end, then - on rank 0 - the consensus grouping of the gathered table
    (`group_captions` / `captions_frequency` under the key `keys_of(frame index)`).  Timed with `timed_region` (max over ranks).
    Returns {"seconds", "ids", "lens"} on every rank, plus {"grouping_s", "objects", "frequencies", "mean_caption_tokens"} on
    rank 0.  `range_check` (optional, e.g. engine.saturations): called after the job; a non-zero count is reported under
    "range_clamps" and warned about (split mode: values clamped to its range)."""
    import time
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    dt, (ids, lens) = timed_region(lambda: caption_shard(submit, frames_of, n_frames, micro_batch, max_len, pad_id, join=join), world, device)
    out: Dict[str, object] = {"seconds": dt, "ids": ids, "lens": lens}
    if range_check is not None:
        n = int(range_check())
        out["range_clamps"] = n
        if n:
            import warnings
            warnings.warn(f"strong_scaling_job: {n} GEMM-input values were clamped to the split mode's range (|x| > 65000)")
    if rank == 0:
        t1 = time.perf_counter()
        ids_h, lens_h = ids.cpu().numpy(), lens.cpu().numpy()
        detok = detokenise or (lambda row: " ".join(str(int(x)) for x in row))
        caps = [detok(row[1:max(int(n) - 1, 1)]) for row, n in zip(ids_h, lens_h)]
        keys = [keys_of(i) for i in range(n_frames)] if keys_of is not None else [(0, i) for i in range(n_frames)]
        freq = captions_frequency(group_captions(keys, caps, apply_filter=False))
        out.update(grouping_s=time.perf_counter() - t1, objects=len(freq), frequencies=freq,
                   mean_caption_tokens=float(lens_h.mean()) if len(lens_h) else 0.0)
    return out


def _fp_hash(fingerprint: str) -> str:
    import hashlib
    return hashlib.sha256(fingerprint.encode()).hexdigest()[:12]


def job_fingerprint(**parts) -> str:
    """A readable, order-independent identity of a captioning job for `caption_shard(fingerprint=...)`, e.g.
    job_fingerprint(weights="blip-base@sha256:...", dtype="f32s", beams=1, frames="synthetic:seed=17")."""
    return ";".join(f"{k}={parts[k]}" for k in sorted(parts))


def caption_shard(generate: Callable[[torch.Tensor], Dict[str, torch.Tensor]], frames_of: Callable[[int, int], torch.Tensor],
                  n_frames: int, micro_batch: int, max_len: int, pad_id: int = 0, join: Callable[[], None] | None = None,
                  resume_dir: str | None = None, record_every: int = 16, fingerprint: str | None = None):
    """Caption frames [first, last) of this rank in micro-batches and gather everything.
    `frames_of(first, count)` returns the device tensor of frames; `generate(frames)` returns {"sequences","lengths"}.
    `join`: called after the last micro-batch (of a record, see below) was issued - for a `generate` that only starts the work
    on another stream (engine.EnginePool.submit / .join: consecutive micro-batches overlap).
    `resume_dir`: failure handling for long jobs.  After every `record_every` micro-batches the finished caption records of
    that span are written to `resume_dir/records_<first>_<last>_L<max_len>.npz` (temporary name + rename: a file is whole
    or absent); a rerun with the same arguments loads the spans whose file exists instead of captioning them again, so a
    job killed at frame k restarts at the span that contains k.  The file name carries the span, which is a function of
    (n_frames, world, rank, micro_batch, record_every): runs with another sharding simply do not find each other's files.
    `fingerprint`: what produced the records - any string that identifies the job (`job_fingerprint(...)`: checkpoint, dtype,
    beams, frame source).  Its hash is part of the file name and the string itself is stored in the file: records written under
    another fingerprint are never loaded (a warning names them), and a file whose stored string differs from its name's hash
    is refused.  Without a fingerprint (None) every record of the span is trusted, as before.
    (The reference's driver has no resume: `detector/pseudolabeler.py:835-843` rewrites every episode_<e>_step_<s>.npz.)
    Returns (ids_all, lens_all) trimmed to n_frames rows, in global frame order."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    first, last, per = shard_range(n_frames, rank, world)
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    if resume_dir is not None:
        os.makedirs(resume_dir, exist_ok=True)
    span = micro_batch * max(int(record_every), 1) if resume_dir is not None else max(last - first, 1)
    ids_parts, len_parts = [], []
    for s0 in range(first, last, span):
        s1 = min(s0 + span, last)
        stem = f"records_{s0:010d}_{s1:010d}_L{max_len}"
        path = os.path.join(resume_dir, stem + (f"_{_fp_hash(fingerprint)}" if fingerprint is not None else "") + ".npz") \
            if resume_dir is not None else None
        if path is not None and fingerprint is not None and not os.path.exists(path):
            other = [f for f in os.listdir(resume_dir) if f.startswith(stem) and f.endswith(".npz") and ".tmp" not in f]
            if other:
                import warnings
                warnings.warn(f"caption_shard: {resume_dir} holds records of this span from another job ({other[0]}); "
                              f"they are ignored and the span is captioned again under fingerprint {fingerprint!r}")
        if path is not None and os.path.exists(path):
            import numpy as np
            with np.load(path) as rec:
                if fingerprint is not None and ("fingerprint" not in rec or str(rec["fingerprint"]) != fingerprint):
                    raise RuntimeError(f"caption_shard: {path} was not written by this job (stored fingerprint "
                                       f"{str(rec['fingerprint']) if 'fingerprint' in rec else None!r}, expected {fingerprint!r})")
                ids_parts.append(torch.from_numpy(rec["ids"]).to(dev))
                len_parts.append(torch.from_numpy(rec["lens"]).to(dev))
            continue
        span_ids, span_lens = [], []
        for i in range(s0, s1, micro_batch):
            n = min(micro_batch, s1 - i)
            out = generate(frames_of(i, n))
            span_ids.append(out["sequences"][:, :max_len])
            span_lens.append(out["lengths"])
        if join is not None:
            join()
        ids_s, lens_s = torch.cat(span_ids), torch.cat(span_lens)
        if path is not None:
            import numpy as np
            tmp = f"{path}.tmp{os.getpid()}.npz"
            extra = {"fingerprint": np.array(fingerprint)} if fingerprint is not None else {}
            np.savez(tmp, ids=ids_s.cpu().numpy(), lens=lens_s.cpu().numpy(), **extra)      # .cpu() waits for the span's work
            os.replace(tmp, path)
        ids_parts.append(ids_s.to(dev))
        len_parts.append(lens_s.to(dev))
    if ids_parts:
        ids = torch.cat(ids_parts)
        lens = torch.cat(len_parts)
    else:   # a rank beyond the end of a short job still takes part in the collective (and still joins its streams once)
        if join is not None:
            join()
        ids = torch.full((0, max_len), pad_id, dtype=torch.int32, device=dev)
        lens = torch.zeros((0,), dtype=torch.int32, device=dev)
    ids_all, lens_all = gather_caption_records(ids, lens, per, pad_id)
    keep = torch.cat([torch.arange(r * per, r * per + (shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0]))
                      for r in range(world)]).to(ids_all.device)
    return ids_all[keep], lens_all[keep]


# ------------------------------------------------------------------------------------------------------------------
# consensus grouping on the gathered table (host side)
# ------------------------------------------------------------------------------------------------------------------

BANNED_WORDS = [
    # Living beings
    "person", "man", "woman", "boy", "girl", "child", "children", "adult", "kid", "baby", "human", "people", "group",
    "crowd", "dog", "cat", "bird", "fish", "horse", "animal", "pet", "elephant", "lion", "tiger", "monkey", "mouse",
    "rabbit", "cow", "pig", "sheep", "deer", "bear", "chicken", "duck", "goat", "camel", "snake", "frog", "turtle",
    "whale", "dolphin", "insect", "bug", "spider",
    # Image quality or context
    "blurry", "picture", "image", "photo", "portrait", "painting", "drawing", "sketch", "screenshot", "artwork",
    "filter", "3d", "rendering",
    # Generic / non-descriptive terms
    "thing", "stuff", "object", "item", "something", "stuff", "device", "equipment", "material", "machine", "gadget",
    "unknown", "unidentified", "indistinguishable", "living room", "kitchen", "bedroom", "bathroom", "dining room",
    "living room", "room",
    # Non-indoor terms
    "car", "vehicle", "bike", "truck", "street", "road", "tree", "forest", "mountain", "park", "outdoor", "sky",
    "landscape", "scenery",
    # Action words
    "running", "jumping", "walking", "talking", "playing", "sitting", "standing", "moving", "holding", "eating",
    "drinking", "flying", "swimming", "driving",
]


def filter_caption(caption: str) -> bool:
    """True when the caption contains none of the banned substrings (pseudocaptioner.py:96-123)."""
    low = caption.lower()
    return not any(w.lower() in low for w in BANNED_WORDS)


def group_captions(keys: Sequence[Tuple[int, int]], captions: Sequence[str], apply_filter: bool = True):
    """(episode_id, object_id) -> list of captions, in input order (pseudocaptioner.py:125-154)."""
    grouped: Dict[Tuple[int, int], List[str]] = {}
    for key, cap in zip(keys, captions):
        if apply_filter and not filter_caption(cap):
            continue
        grouped.setdefault(tuple(key), []).append(cap)
    return grouped


def captions_frequency(grouped: Dict[Tuple[int, int], Iterable[str]]):
    """(episode_id, object_id) -> [[freq, caption], ...] in first-seen order (pseudocaptioner.py:156-177)."""
    out = {}
    for key, caps in grouped.items():
        freq: Dict[str, int] = {}
        for c in caps:
            freq[c] = freq.get(c, 0) + 1
        out[key] = [[n, c] for c, n in freq.items()]
    return out


def consensus_caption(freq_list):
    """Most frequent caption of an object; ties go to the first seen (a deterministic stand-in for the reference's LLM
    merge step, pseudocaptioner.py:364-447, which stays out of scope)."""
    best = max(freq_list, key=lambda fc: fc[0])
    return best[1]
