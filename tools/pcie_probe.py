#!/usr/bin/env python3
"""Host <-> device copy rates of this box: pageable and page-locked buffers, one direction and both at once
(two streams), as hipMemcpyAsync serves them (HSA_ENABLE_SDMA=0 switches the runtime to shader copies)."""
import os, sys, time
import torch
n = 393216000 // 4                       # the cfg4 batch: 393 MB each way
dev = torch.empty(n, dtype=torch.float32, device='cuda')
dev2 = torch.empty(n, dtype=torch.float32, device='cuda')
pageable = torch.empty(n, dtype=torch.float32).uniform_(-1, 1)
pinned = torch.empty(n, dtype=torch.float32).pin_memory()
pinned2 = torch.empty(n, dtype=torch.float32).pin_memory()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def both():
    with torch.cuda.stream(s1): dev.copy_(pinned, non_blocking=True)
    with torch.cuda.stream(s2): pinned2.copy_(dev2, non_blocking=True)

def split_h2d():
    h = n // 2
    with torch.cuda.stream(s1): dev[:h].copy_(pinned[:h], non_blocking=True)
    with torch.cuda.stream(s2): dev[h:].copy_(pinned[h:], non_blocking=True)

gb = n * 4 / 1e9
print('HSA_ENABLE_SDMA =', os.environ.get('HSA_ENABLE_SDMA', '(default)'))
print(f'H2D pageable        {gb / timed(lambda: dev.copy_(pageable)):6.1f} GB/s')
print(f'H2D pinned          {gb / timed(lambda: dev.copy_(pinned, non_blocking=True)):6.1f} GB/s')
print(f'D2H pinned          {gb / timed(lambda: pinned2.copy_(dev2, non_blocking=True)):6.1f} GB/s')
print(f'H2D pinned, 2 streams x half {gb / timed(split_h2d):6.1f} GB/s')
print(f'H2D + D2H pinned at once     {2 * gb / timed(both):6.1f} GB/s in + out')
