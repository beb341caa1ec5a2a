"""CPU, world_size 2 over gloo: the fixed-shape caption-record all-gather and the sharded caption driver."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from embodied_captioning_amd import distributed as D
    L = 6

    def frames_of(first, n):                      # a "frame" is just its global index here
        return torch.arange(first, first + n, dtype=torch.int32)

    def generate(frames):                         # fake captioner: ids depend only on the frame index
        ids = torch.stack([(frames * 7 + j) % 1000 for j in range(L)], dim=1).int()
        return {"sequences": ids, "lengths": (frames % L + 1).int()}

    # `generate` may only START the work (EnginePool.submit): the results are filled in by `join`, which caption_shard must
    # call once, after the last micro-batch and before it touches the outputs
    pending, joined = [], []

    def submit(frames):
        out = {"sequences": torch.zeros((frames.shape[0], L), dtype=torch.int32), "lengths": torch.zeros(frames.shape[0], dtype=torch.int32)}
        pending.append((frames, out))
        return out

    def join():
        joined.append(len(pending))
        for frames, out in pending:
            ref = generate(frames)
            out["sequences"].copy_(ref["sequences"]); out["lengths"].copy_(ref["lengths"])

    ids, lens = D.caption_shard(submit, frames_of, n_frames, micro_batch=4, max_len=L, join=join)
    assert len(joined) == 1 and joined[0] == len(pending)
    ids2, lens2 = D.caption_shard(generate, frames_of, n_frames, micro_batch=4, max_len=L)
    assert torch.equal(ids, ids2) and torch.equal(lens, lens2)
    q.put((rank, ids.clone(), lens.clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [10, 13, 1])
def test_sharded_caption_gather_world2(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames = torch.arange(n_frames, dtype=torch.int32)
    want_ids = torch.stack([(frames * 7 + j) % 1000 for j in range(6)], dim=1).int()
    want_len = (frames % 6 + 1).int()
    for _, ids, lens in res:                      # every rank holds the full table in global frame order
        assert torch.equal(ids, want_ids) and torch.equal(lens, want_len)


def test_caption_shard_resume_skips_finished_spans(tmp_path):
    """Failure handling of the shard driver: records of finished spans are files; a rerun captions
    only the spans whose file is missing and returns the same table."""
    from embodied_captioning_amd import distributed as D
    L, calls = 5, []

    def frames_of(first, n):
        return torch.arange(first, first + n, dtype=torch.int32)

    def generate(frames):
        calls.append(int(frames[0]))
        ids = torch.stack([(frames * 3 + j) % 97 for j in range(L)], dim=1).int()
        return {"sequences": ids, "lengths": (frames % L + 1).int()}

    d = str(tmp_path / "records")
    want = D.caption_shard(generate, frames_of, 23, micro_batch=4, max_len=L)
    calls.clear()
    a = D.caption_shard(generate, frames_of, 23, micro_batch=4, max_len=L, resume_dir=d, record_every=2)
    assert calls == [0, 4, 8, 12, 16, 20] and torch.equal(a[0], want[0]) and torch.equal(a[1], want[1])
    files = sorted(os.listdir(d))
    assert files == ["records_0000000000_0000000008_L5.npz", "records_0000000008_0000000016_L5.npz",
                     "records_0000000016_0000000023_L5.npz"]
    calls.clear()
    b = D.caption_shard(generate, frames_of, 23, micro_batch=4, max_len=L, resume_dir=d, record_every=2)
    assert calls == [] and torch.equal(b[0], want[0]) and torch.equal(b[1], want[1])
    os.remove(os.path.join(d, files[1]))                   # the job died while this span was in flight
    calls.clear()
    c = D.caption_shard(generate, frames_of, 23, micro_batch=4, max_len=L, resume_dir=d, record_every=2)
    assert calls == [8, 12] and torch.equal(c[0], want[0]) and torch.equal(c[1], want[1])
    assert not [f for f in os.listdir(d) if "tmp" in f]


def test_caption_shard_resume_ignores_records_of_another_job(tmp_path):
    """A rerun in the same directory with another checkpoint / dtype / beam count must not pick up the old spans: the job
    fingerprint is part of the record's name and content."""
    import numpy as np
    from embodied_captioning_amd import distributed as D
    L, calls = 4, []

    def frames_of(first, n):
        return torch.arange(first, first + n, dtype=torch.int32)

    def make(mult):
        def generate(frames):
            calls.append(int(frames[0]))
            return {"sequences": torch.stack([(frames * mult + j) % 89 for j in range(L)], dim=1).int(), "lengths": (frames % L + 1).int()}
        return generate

    d = str(tmp_path / "records")
    fa = D.job_fingerprint(weights="A", dtype="f32s", beams=1)
    fb = D.job_fingerprint(dtype="f32s", beams=1, weights="B")
    assert fa == D.job_fingerprint(beams=1, dtype="f32s", weights="A") and fa != fb
    a = D.caption_shard(make(3), frames_of, 10, micro_batch=4, max_len=L, resume_dir=d, record_every=1, fingerprint=fa)
    calls.clear()
    with pytest.warns(UserWarning, match="another job"):
        b = D.caption_shard(make(5), frames_of, 10, micro_batch=4, max_len=L, resume_dir=d, record_every=1, fingerprint=fb)
    assert calls == [0, 4, 8] and not torch.equal(a[0], b[0])            # job B captioned everything itself
    calls.clear()
    a2 = D.caption_shard(make(3), frames_of, 10, micro_batch=4, max_len=L, resume_dir=d, record_every=1, fingerprint=fa)
    assert calls == [] and torch.equal(a2[0], a[0])                      # job A still finds its own records
    # a file renamed into another job's slot is refused, not loaded
    name_a = [f for f in sorted(os.listdir(d)) if D._fp_hash(fa) in f][0]
    name_b = name_a.replace(D._fp_hash(fa), D._fp_hash(fb))
    os.replace(os.path.join(d, name_a), os.path.join(d, name_b))
    with pytest.raises(RuntimeError, match="not written by this job"):
        D.caption_shard(make(5), frames_of, 10, micro_batch=4, max_len=L, resume_dir=d, record_every=1, fingerprint=fb)
    assert np.load(os.path.join(d, name_b))["fingerprint"] == fa


def _bench_worker(rank, world, port, n_frames, q):
    """What bench.py's N > 1 branches run, on gloo with a fake captioner: the per-step gather into buffers allocated once, the
    timed bracket with its max-over-ranks all-reduce, and the strong-scaling job (shards, one gather, grouping on rank 0)."""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from embodied_captioning_amd import distributed as D
    B, L = 5, 6

    def generate(frames):
        ids = torch.stack([(frames * 7 + j) % 1000 for j in range(L)], dim=1).int()
        return {"sequences": ids, "lengths": (frames % (L - 1) + 2).int()}

    # weak scaling: every rank captions its own B frames per step, one all-gather per step
    gather = D.make_step_gather(world, B, L, "cpu")
    mine = torch.arange(rank * B, rank * B + B, dtype=torch.int32)

    def steps():
        for _ in range(3):
            out = generate(mine)
            res = gather(out["sequences"], out["lengths"])
        time.sleep(0.05 * (rank + 1))                     # ranks finish at different times: everyone must report the slowest
        return res
    dt, (ids_all, len_all) = D.timed_region(steps, world, None)
    want = generate(torch.arange(world * B, dtype=torch.int32))
    assert torch.equal(ids_all, want["sequences"]) and torch.equal(len_all, want["lengths"])
    assert dt >= 0.05 * world
    try:
        gather(want["sequences"], want["lengths"])           # a record block of another shape is refused, not mis-gathered
        raise AssertionError("shape check missing")
    except ValueError:
        pass
    # strong scaling: a fixed total of frames
    clamps = []
    job = D.strong_scaling_job(generate, lambda first, n: torch.arange(first, first + n, dtype=torch.int32), n_frames, micro_batch=4,
                               max_len=L, keys_of=lambda i: (i // 6, i % 3), range_check=lambda: clamps.append(1) or 0)
    assert clamps == [1] and job["range_clamps"] == 0
    q.put((rank, dt, job["seconds"], job["ids"].clone(), job["lens"].clone(), job.get("objects"), job.get("frequencies")))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_multi_rank_paths_world2():
    n_frames = 13
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames = torch.arange(n_frames, dtype=torch.int32)
    want_ids = torch.stack([(frames * 7 + j) % 1000 for j in range(6)], dim=1).int()
    want_len = (frames % 5 + 2).int()
    (r0, dt0, job0, ids0, lens0, objects, freq), (r1, dt1, job1, ids1, lens1, obj1, freq1) = res
    assert dt0 == dt1 and job0 == job1                       # the max over ranks, on every rank
    for ids, lens in ((ids0, lens0), (ids1, lens1)):
        assert torch.equal(ids, want_ids) and torch.equal(lens, want_len)
    assert obj1 is None and freq1 is None                    # grouping runs on rank 0 only
    keys = {(i // 6, i % 3) for i in range(n_frames)}
    assert objects == len(keys) and set(freq) == keys
    assert sum(n for fl in freq.values() for n, _ in fl) == n_frames


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: bench.py starts the two ranks itself (a
    torch.distributed.run child process at 127.0.0.1, spawned before any GPU call), they rendezvous, run the timed steps with the
    per-step caption all-gather and the strong-scaling job, and rank 0's line carries n_gpus = the size of the process group.
    --stub-engine: CPU ranks over gloo with a fake captioner (the GPU branch differs in the backend name and set_device only)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-engine", "--steps", "3", "--warmup", "1",
                        "--frames", "203", "--batch", "8", "--max-length", "6"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                        # rank 0 alone prints, once
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["stub"] is True and line["scaling"] == "weak" and line["value"] > 0
    st = line["strong_scaling"]
    assert st["frames"] == 203 and st["n_gpus"] == 2 and st["frames_per_rank"] == 102 and st["scaling"] == "strong"
    assert st["objects"] == len({(i // 500, (i // 10) % 50) for i in range(203)})
    # a launcher that set WORLD_SIZE to something else is an error, not a silent single-rank run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-engine"], env={**env, "WORLD_SIZE": "1"},
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr
