#!/usr/bin/env python
"""What a gloo all-gather of CUDA records costs when two ranks share one GPU (the --share-gpu rehearsal's per-step collective):
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/gloo_cuda_probe.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group("gloo")
w, r = dist.get_world_size(), dist.get_rank()
ids = torch.full((256, 20), r, dtype=torch.int32, device="cuda")
out = torch.empty((w * 256, 20), dtype=torch.int32, device="cuda")
for dev_name, a, b in (("cuda", ids, out), ("cpu", ids.cpu(), out.cpu())):
    for _ in range(2):
        dist.all_gather_into_tensor(b, a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        dist.all_gather_into_tensor(b, a)
    torch.cuda.synchronize()
    if r == 0:
        print(f"gloo all_gather_into_tensor of [256, 20] int32 on {dev_name}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per call", flush=True)
dist.destroy_process_group()
