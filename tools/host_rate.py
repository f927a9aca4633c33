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

# cfg1 (BASELINE configs[0], the reference's own CPU-runnable case): VelvetNoise.decorrelate on 5.7 s of
# 44.1 kHz stereo (viola.wav's shape), host to host, bit-identical mode, against the reference's
# procedure restated on the CPU (oracle, 1 core)
from oracle import vnd_oracle as O
xv = np.random.default_rng(1).uniform(-0.5, 0.5, (250774, 2)).astype(np.float32)
vn = vnd.VelvetNoise(sample_rate_hz=44100, seed=1)
want = O.decorrelate(xv.copy(), sample_rate_hz=44100, seed=1)
assert np.array_equal(vn.decorrelate(xv), want)
t = time.perf_counter()
for _ in range(20): vn.decorrelate(xv)
gpu = (time.perf_counter() - t) / 20
t = time.perf_counter()
for _ in range(3): O.decorrelate(xv.copy(), sample_rate_hz=44100, seed=1)
cpu = (time.perf_counter() - t) / 3
print(f'cfg1 VelvetNoise.decorrelate (250774 x 2, bit-identical): {gpu*1e3:.2f} ms host to host; '
      f'reference procedure on 1 host core {cpu*1e3:.1f} ms  ({cpu/gpu:.0f}x)')
