"""How do the image tower of one batch and the decode chain of another share the GPU?  Stream A runs encode() back to back,
stream B runs generate() back to back, each from its own host thread, for a fixed wall time; completions are compared with
the rates each reaches alone."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.config import BlipArch  # noqa: E402
from embodied_captioning_amd.engine import CaptionerEngine  # noqa: E402
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels  # noqa: E402

B, L, WALL = 256, 20, 1.0
arch = BlipArch()
sd = procedural_blip_state_dict(arch, 0, eos_boost=9.0)
engs = [CaptionerEngine(arch, dtype="bf16", max_batch=B, max_beams=1, max_len=L) for _ in range(2)]
for e in engs:
    e.load_state_dict(sd)
px = synthetic_pixels(B, arch.image_size, seed=3).cuda()
for e in engs:
    e.generate(px, max_length=L)
torch.cuda.synchronize()


def loop(fn, stream, out, key, stop):
    n = 0
    with torch.cuda.stream(stream):
        while not stop.is_set():
            fn()
            if n % 4 == 3:
                stream.synchronize()          # keep the host at most a few calls ahead
            n += 1
        stream.synchronize()
    out[key] = n


def run(jobs):
    stop, out = threading.Event(), {}
    ths = [threading.Thread(target=loop, args=(fn, torch.cuda.Stream(), out, k, stop)) for k, fn in jobs.items()]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    time.sleep(WALL)
    stop.set()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return {k: v / dt for k, v in out.items()}


enc = lambda: engs[0].encode(px)                                   # noqa: E731
gen = lambda: engs[1].generate(px, max_length=L)                   # noqa: E731
a = run({"encode": enc}); b = run({"generate": gen}); c = run({"encode": enc, "generate": gen})
print(f"alone: {a['encode']:.1f} encodes/s ({1e3 / a['encode']:.1f} ms), {b['generate']:.1f} generates/s ({1e3 / b['generate']:.1f} ms)")
print(f"together: {c['encode']:.1f} encodes/s ({c['encode'] / a['encode']:.2f} of alone), {c['generate']:.1f} generates/s "
      f"({c['generate'] / b['generate']:.2f} of alone)", flush=True)
