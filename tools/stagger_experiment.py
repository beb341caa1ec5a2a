#!/usr/bin/env python
"""Pool scheduling experiment (GPU box, experiments build): N engines on their own streams and host threads, each running its
batches back to back; engine i starts i x STAGGER ms late (a device-side sleep on its stream), so that in steady state one
engine encodes while the others decode.  With CAP_EXP_NONPERSIST=1 the encoder GEMMs run one tile per workgroup, so decode
kernels of the other streams find free CUs all the time.
    [CAP_EXP_NONPERSIST=1] python tools/stagger_experiment.py [--streams 3] [--stagger 15] [--batches 24]"""
import argparse, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_captioning_amd.config import BlipArch
from embodied_captioning_amd.engine import CaptionerEngine
from embodied_captioning_amd.weights import procedural_blip_state_dict, synthetic_pixels

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--stagger", type=float, default=0.0, help="ms between the first batches of consecutive engines")
ap.add_argument("--batches", type=int, default=24)
ap.add_argument("--prio", action="store_true", help="engines on alternating stream priorities (no effect expected: both phases share a stream)")
a = ap.parse_args()
B, L = 256, 20
arch = BlipArch()
sd = procedural_blip_state_dict(arch, 0, eos_boost=9.0)
first = CaptionerEngine(arch, dtype="f32s", max_batch=B, max_beams=1, max_len=L)
first.load_state_dict(sd)
engines = [first] + [CaptionerEngine(arch, dtype="f32s", max_batch=B, max_beams=1, max_len=L, share_weights_with=first) for _ in range(a.streams - 1)]
streams = [torch.cuda.Stream() for _ in engines]
px = synthetic_pixels(B, arch.image_size, seed=0).cuda()
for e, s in zip(engines, streams):
    with torch.cuda.stream(s):
        e.generate(px, max_length=L)
torch.cuda.synchronize()
clk = 2.4e9      # shader cycles per second at the top clock (the sleep is longer when the clock is lower)
per = a.batches // a.streams


def work(i):
    with torch.cuda.stream(streams[i]):
        if a.stagger > 0 and i > 0:
            torch.cuda._sleep(int(i * a.stagger * 1e-3 * clk))
        for _ in range(per):
            engines[i].generate(px, max_length=L)


for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(a.streams)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = per * a.streams
    print(f"streams {a.streams} stagger {a.stagger} ms nonpersist {bool(os.environ.get('CAP_EXP_NONPERSIST'))}: {1e3 * dt / n:.2f} ms per batch "
          f"({n * B / dt:.0f} captions/s, sleeps included)", flush=True)
