#!/usr/bin/env python3
"""development probe (GPU box): does a tensor that was just written stream back faster than a cold one?  (MI355X: 256 MB of
Infinity Cache in front of HBM.)  A copy kernel over tensors of several sizes, (a) the same pair of buffers every launch,
(b) cycling through > 2 GB of buffers."""
import torch

DEV = "cuda:0"


def run(nbytes, nbuf, reps=200):
    n = nbytes // 2
    xs = [torch.empty(n, dtype=torch.bfloat16, device=DEV).normal_() for _ in range(nbuf)]
    ys = [torch.empty_like(x) for x in xs]
    for i in range(nbuf):
        ys[i].copy_(xs[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        ys[i % nbuf].copy_(xs[i % nbuf])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    return us, 2 * nbytes / us / 1e6


if __name__ == "__main__":
    for mb in (4, 8, 17, 34, 67, 134, 268):
        nb = mb << 20
        warm = run(nb, 1)
        chain = run(nb, 2)
        cold = run(nb, max(2, (3 << 30) // (2 * nb)))
        print(f"{mb:4d} MB copy: same buffers {warm[0]:7.1f} us ({warm[1]:5.2f} TB/s)   two pairs {chain[0]:7.1f} us ({chain[1]:5.2f} TB/s)   "
              f"cycling 3 GB {cold[0]:7.1f} us ({cold[1]:5.2f} TB/s)")
