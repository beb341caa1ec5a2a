#!/usr/bin/env python
"""RCCL smoke on ONE GPU (the box has one): the exact collectives bench.py / distributed.py issue at N > 1 - init with
device_id, all_gather_into_tensor of the int32 caption records, MAX all-reduce of the step time, barrier - on a world of one
rank, so that the API use and the library load are exercised on hardware before the driver's multi-GPU run.
    python tools/rccl_sanity.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
ids = torch.arange(256 * 20, dtype=torch.int32, device=dev).view(256, 20)
lens = torch.arange(256, dtype=torch.int32, device=dev)
ids_all = torch.empty_like(ids)
len_all = torch.empty_like(lens)
dist.all_gather_into_tensor(ids_all, ids)
dist.all_gather_into_tensor(len_all, lens)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(ids_all, ids) and torch.equal(len_all, lens) and float(t.item()) == 1.25
from embodied_captioning_amd.distributed import gather_caption_records
a, b = gather_caption_records(ids[:100], lens[:100], 128)
assert a.shape == (128, 20) and int(b[100:].sum()) == 0
dist.destroy_process_group()
print("rccl sanity ok: nccl backend =", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "?")
