#!/usr/bin/env python
"""pseudo-label N synthetic frames (seed = frame index), sharded contiguously over the ranks,
per-GPU micro-batches (--micro-batch, default 1024; SURVEY config 4 names 256), greedy; ONE fixed-shape all-gather of the caption records at the end; consensus grouping
with the synthetic (episode, object) key = (i // 500, (i // 10) % 50).  One JSON line from rank 0.

    python tools/caption_frames.py --frames 5120                      # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/caption_frames.py --frames 50000
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd.config import BlipArch  # noqa: E402
from embodied_captioning_amd.distributed import (caption_shard, captions_frequency, consensus_caption, group_captions,  # noqa: E402
                                                  shard_range)
from embodied_captioning_amd.engine import CaptionerEngine  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=50000)
    ap.add_argument("--micro-batch", type=int, default=1024, help="frames per pass (round 5: 256 / 512 / 1024 frames per pass on 3 engines = "
                    "5 840 / 6 310 / 6 460 captions/s; SURVEY config 4 names 256)")
    ap.add_argument("--max-length", type=int, default=20)
    ap.add_argument("--dtype", default="f32s", help="f32s (token-identical to the fp32 reference, default) | bf16 | f32")
    ap.add_argument("--streams", type=int, default=3, help="engines / HIP streams the micro-batches rotate over (engine.EnginePool)")
    ap.add_argument("--boxes", type=int, default=0, help="> 0: every unit is an object crop - raw 512x512 BGR frames made on "
                    "the device, this many boxes per frame (host RNG, seed = frame index), the reference's expand_box, crop + "
                    "Pillow-exact bicubic resize on the device (preprocess.crop_resize_u8), then the captioner; --frames "
                    "counts crops")
    ap.add_argument("--resume", default=None, metavar="DIR", help="write the finished caption records of every --record-every "
                    "micro-batches to DIR and, on a rerun with the same arguments, skip the spans whose record exists "
                    "(distributed.caption_shard)")
    ap.add_argument("--record-every", type=int, default=16)
    a = ap.parse_args()
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=dev)
    arch = BlipArch()
    from embodied_captioning_amd.engine import EnginePool
    eng = EnginePool(arch, n=a.streams, device=dev, dtype=a.dtype, max_batch=a.micro_batch, max_beams=1, max_len=a.max_length)
    eng.load_state_dict(procedural_blip_state_dict(arch, 0, eos_boost=9.0))
    gen = torch.Generator(device=dev)

    def frames_of(first, count):
        # raw RGB frames made on the device, a function of the first frame index only (the same whatever the sharding
        # as long as micro-batches start at multiples of --micro-batch within a shard)
        gen.manual_seed(1_000_003 * first + 17)
        return torch.randint(0, 256, (count, arch.image_size, arch.image_size, 3), dtype=torch.uint8, device=dev, generator=gen)

    if a.boxes > 0:
        import numpy as np
        from embodied_captioning_amd.preprocess import crop_resize_u8
        from embodied_captioning_amd.pseudolabeler import expand_box
        if a.micro_batch % a.boxes:
            raise SystemExit("--micro-batch must be a multiple of --boxes")
        FH, FW = 512, 512          # square, as the habitat frames are: the reference's expand_box swaps the clamps of x and y

        def frames_of(first, count):      # noqa: F811 - `count` crops = count / boxes raw frames
            out = []
            for f in range(first // a.boxes, (first + count) // a.boxes):
                gen.manual_seed(1_000_003 * f + 17)
                frame = torch.randint(0, 256, (FH, FW, 3), dtype=torch.uint8, device=dev, generator=gen)
                rng = np.random.default_rng(f)
                rects = []
                for _ in range(a.boxes):
                    w, h = int(rng.integers(24, 400)), int(rng.integers(24, 400))
                    x, y = int(rng.integers(0, FW - w)), int(rng.integers(0, FH - h))
                    rects.append(expand_box((x, y, x + w, y + h), 0.2, (FH, FW, 3)))
                out.append(crop_resize_u8(frame, rects, arch.image_size, bgr=True, device=dev))
            return torch.cat(out)

    eng.generate_many([frames_of(0, a.micro_batch)] * a.streams, max_length=a.max_length)   # warm-up (allocations, code load)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first, last, _ = shard_range(a.frames, rank, world)
    ids, lens = caption_shard(lambda f: eng.submit(f, max_length=a.max_length), frames_of, a.frames, a.micro_batch,
                              a.max_length, arch.pad, join=eng.join, resume_dir=a.resume, record_every=a.record_every)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ids_h, lens_h = ids.cpu().numpy(), lens.cpu().numpy()
    captions = [" ".join(str(t) for t in row[1:n - 1]) for row, n in zip(ids_h, lens_h)]   # no vocabulary offline: id strings
    keys = [(i // 500, (i // 10) % 50) for i in range(a.frames)]
    freq = captions_frequency(group_captions(keys, captions, apply_filter=False))
    best = {k: consensus_caption(v) for k, v in freq.items()}
    t2 = time.perf_counter()
    t = torch.tensor([t1 - t0, t2 - t1], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    if rank == 0:
        what = "frames made on device" if a.boxes == 0 else f"512x512 frames made on device, {a.boxes} boxes each expanded / cropped / resized on device"
        print(json.dumps({"metric": f"captions/sec end to end (config 4: {what}, caption, all-gather)",
                          "value": round(a.frames / float(t[0]), 1), "unit": "captions/s", "n_gpus": world,
                          "frames": a.frames, "shard": [first, last], "caption_and_gather_s": round(float(t[0]), 3),
                          "grouping_s": round(float(t[1]), 3), "objects": len(best),
                          "mean_caption_tokens": round(float(lens_h.mean()), 2), "dtype": a.dtype, "streams": a.streams}))
    eng.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
