#!/usr/bin/env python
"""Experiment: is the stream pool bound by the HOST issuing ~2 800 launches per generate?
  (a) host time to enqueue one generate (no sync), alone and with three host threads enqueueing at once;
  (b) the pool with every engine's generate captured into a HIP graph once and replayed (one hipGraphLaunch per batch).
    python tools/pool_graph_experiment.py [dtype] [streams]"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_captioning_amd.config import BlipArch
from embodied_captioning_amd.engine import CaptionerEngine
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32s"
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
arch = BlipArch()
sd = procedural_blip_state_dict(arch, 0, eos_boost=9.0)
B, L, STEPS = 256, 20, 12
px = [synthetic_pixels(B, arch.image_size, seed=0, first=i * B).cuda() for i in range(NS)]
engs = [CaptionerEngine(arch, dtype=dtype, max_batch=B, max_beams=1, max_len=L) for _ in range(1)]
engs[0].load_state_dict(sd)
engs += [CaptionerEngine(arch, dtype=dtype, max_batch=B, max_beams=1, max_len=L, share_weights_with=engs[0]) for _ in range(NS - 1)]
streams = [torch.cuda.Stream() for _ in range(NS)]
torch.cuda.synchronize()

# (a) host enqueue time
with torch.cuda.stream(streams[0]):
    for _ in range(2):
        ref = engs[0].generate(px[0], max_length=L)["sequences"].clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engs[0].generate(px[0], max_length=L)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"one thread: enqueue {1e3 * (t1 - t0):.1f} ms, until done {1e3 * (t2 - t0):.1f} ms", flush=True)
enq = [0.0] * NS


def work(i, n):
    with torch.cuda.stream(streams[i]):
        t = time.perf_counter()
        for _ in range(n):
            engs[i].generate(px[i], max_length=L)
        enq[i] = (time.perf_counter() - t) / n


for rep in range(2):
    ts = [threading.Thread(target=work, args=(i, 4)) for i in range(NS)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
print(f"{NS} threads x 4 generates: enqueue per generate {[round(1e3 * e, 1) for e in enq]} ms; host done after {1e3 * th:.1f} ms, "
      f"GPU after {1e3 * tt:.1f} ms = {1e3 * tt / (4 * NS):.1f} ms per batch", flush=True)

# (b) graphs
graphs, gouts = [], []
for i in range(NS):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(streams[i]):
        with torch.cuda.graph(g, stream=streams[i]):
            gouts.append(engs[i].generate(px[i], max_length=L))
    graphs.append(g)
torch.cuda.synchronize()
for _ in range(2):
    for i in range(NS):
        with torch.cuda.stream(streams[i]):
            graphs[i].replay()
torch.cuda.synchronize()
assert torch.equal(gouts[0]["sequences"], ref), "graph replay changed the captions"
t0 = time.perf_counter()
for k in range(STEPS):
    i = k % NS
    with torch.cuda.stream(streams[i]):
        graphs[i].replay()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print(f"graphs, {NS} streams: {STEPS} batches, host {1e3 * th:.1f} ms, GPU {1e3 * tt:.1f} ms = {1e3 * tt / STEPS:.2f} ms per batch = "
      f"{B * STEPS / tt:.0f} captions/s", flush=True)
