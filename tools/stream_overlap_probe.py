"""Do kernels of two HIP streams run at the same time on this GPU?  Each stream gets a chain of single-block spin kernels
(torch.cuda._sleep); if the queues are served concurrently two chains take as long as one."""
import time

import torch

cyc = 2_000_000
torch.cuda._sleep(cyc)
torch.cuda.synchronize()


def run(n_streams, per):
    ss = [torch.cuda.Stream() for _ in range(n_streams)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(per):
        for s in ss:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for n in (1, 2, 3, 4, 8):
    print(f"{n} stream(s) x 20 spin kernels: {run(n, 20):.1f} ms", flush=True)
