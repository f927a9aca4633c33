#!/usr/bin/env python3
"""One cfg2 signal (and a 60 s one) through the synchronous host API: ms per call for the piece count in
VND_HOST_TIME_PIECES (0 = the library's own choice; VND_HOST_TIME_CHUNKS=0 = one piece)."""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
rng = np.random.default_rng(3)
tag = f"pieces={os.environ.get('VND_HOST_TIME_PIECES', 'auto')} chunks={os.environ.get('VND_HOST_TIME_CHUNKS', '1')}"
for n in (480000, 2880000):
    x = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    xp = torch.from_numpy(x).pin_memory().numpy()
    for kind, a in (('pageable', x), ('pinned', xp)):
        for mode in (vnd.MODE_EXACT, vnd.MODE_FAST):
            vnd.convolve_velvet_noise(a, fir, mode=mode)
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 0.4:
                vnd.convolve_velvet_noise(a, fir, mode=mode); reps += 1
            dt = (time.perf_counter() - t0) / reps
            print(f'{tag:24s} n={n:8d} {kind:8s} mode={mode}  {dt * 1e3:7.3f} ms/call  {2 * x.nbytes / dt / 1e9:6.1f} GB/s in+out', flush=True)
