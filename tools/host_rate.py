#!/usr/bin/env python3
"""End-to-end rate of the synchronous host API (pageable NumPy in, NumPy out):
H2D + kernel + D2H over PCIe, for the record in DESIGN.md (never the bench value)."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import vndecorrelate_amd.decorrelation as vnd

fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
for batch in (1, 16, 64):
    x = np.random.default_rng(0).uniform(-1, 1, (batch, 480000, 2)).astype(np.float32)
    vnd.convolve_velvet_noise_batched(x, fir, mode=vnd.MODE_FAST)
    best = 1e9
    for _ in range(5):
        t = time.perf_counter(); y = vnd.convolve_velvet_noise_batched(x, fir, mode=vnd.MODE_FAST); best = min(best, time.perf_counter() - t)
    print(f'host API, {batch:3d} cfg2 signals: {best*1e3:8.2f} ms  {x.size/best/1e6:10.1f} Msamples/s  '
          f'{8*x.size/best/1e9:7.2f} GB/s over PCIe (in+out)')
x1 = np.ascontiguousarray(x[0])
t = time.perf_counter()
for _ in range(20): vnd.convolve_velvet_noise(x1, fir, mode=vnd.MODE_FAST)
print(f'single cfg2 signal, drop-in call: {(time.perf_counter()-t)/20*1e3:.3f} ms per call')
