#!/usr/bin/env python3
"""One-off: a single stereo stream longer than 2^31 bytes (300 M frames, 2.4 GB each way) through the
host API, exact and fast, against the C oracle - exercises the 64-bit stream offsets and the clamped
buffer descriptors.  Not part of the test suite (it needs ~10 GB of host memory and a minute)."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import vndecorrelate_amd.decorrelation as vnd
from oracle import c_oracle, vnd_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
rng = np.random.default_rng(0)
x = np.empty((n, 2), np.float32)
step = 10_000_000
for i in range(0, n, step):
    x[i:i + step] = rng.uniform(-1, 1, (min(step, n - i), 2)).astype(np.float32)
print(f'signal {x.nbytes / 2**30:.2f} GiB', flush=True)
offs, idx, w = O.fir_to_taps(fir)
t = time.perf_counter(); want = c_oracle.convolve(x, offs, idx, w, threads=64); print(f'oracle {time.perf_counter() - t:.1f} s', flush=True)
t = time.perf_counter(); y = vnd.convolve_velvet_noise(x, fir, mode=vnd.MODE_EXACT); print(f'exact host call {time.perf_counter() - t:.2f} s', flush=True)
assert np.array_equal(y, want), 'exact mode differs'
y = vnd.convolve_velvet_noise(x, fir, mode=vnd.MODE_FAST)
err = 0.0
for i in range(0, n, step):
    err = max(err, float(np.max(np.abs(y[i:i + step].astype(np.float64) - want[i:i + step]))))
peak = float(np.max(np.abs(want[:step])))
print(f'exact: bit-identical over {n} frames; fast: max error {err:.2e} = {err / peak:.2e} of peak')
assert err <= 1e-6 * peak
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
t = time.perf_counter(); d = vn.decorrelate(x); t_dev = time.perf_counter() - t
vnd.set_device_epilogue(False)
t = time.perf_counter(); h = vn.decorrelate(x); t_host = time.perf_counter() - t
print(f'decorrelate: device epilogue {t_dev:.2f} s, NumPy epilogue {t_host:.2f} s, bit-identical: {bool(np.array_equal(d, h))}')
assert np.array_equal(d, h)
