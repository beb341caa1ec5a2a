"""Do two independent cap_generate calls on two streams overlap on this GPU?  Two engines (own arenas, same weights), two
streams; 2N batches issued sequentially on one stream vs alternately on two (optionally from two host threads)."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.config import BlipArch  # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402

B, L, N = int(os.environ.get("B", 256)), 20, int(os.environ.get("N", 4))
NE = int(os.environ.get("ENGINES", 2))
arch = BlipArch()
sd = procedural_blip_state_dict(arch, 0, eos_boost=9.0)
engs = []
for i in range(NE):
    e = CaptionerEngine(arch, dtype="bf16", max_batch=B, max_beams=1, max_len=L)
    e.load_state_dict(sd)
    engs.append(e)
px = [synthetic_pixels(B, arch.image_size, seed=3, first=i * B).cuda() for i in range(NE)]
streams = [torch.cuda.Stream() for _ in range(NE)]
for e, p in zip(engs, px):
    e.generate(p, max_length=L)
torch.cuda.synchronize()

t0 = time.perf_counter()
for _ in range(N):
    for e, p in zip(engs, px):
        ref = e.generate(p, max_length=L)
torch.cuda.synchronize()
t_seq = time.perf_counter() - t0

t0 = time.perf_counter()
for _ in range(N):
    for e, p, s in zip(engs, px, streams):
        with torch.cuda.stream(s):
            out = e.generate(p, max_length=L)
torch.cuda.synchronize()
t_one_thread = time.perf_counter() - t0


def worker(e, p, s):
    with torch.cuda.stream(s):
        for _ in range(N):
            e.generate(p, max_length=L)


t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=a) for a in zip(engs, px, streams)]
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize()
t_threads = time.perf_counter() - t0
tot = NE * N * B
print(f"B={B} engines={NE}: sequential {t_seq / (NE * N) * 1e3:.2f} ms/batch ({tot / t_seq:.0f} captions/s) | {NE} streams, one host "
      f"thread {t_one_thread / (NE * N) * 1e3:.2f} ms/batch ({tot / t_one_thread:.0f}) | {NE} streams, {NE} host threads "
      f"{t_threads / (NE * N) * 1e3:.2f} ms/batch ({tot / t_threads:.0f})", flush=True)
