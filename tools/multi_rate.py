#!/usr/bin/env python3
"""The one-process device pool against the plain call: cfg4's batch (1024 x 1 s stereo, 30 taps) host to host through
convolve_velvet_noise_batched(x, fir) and through (x, fir, devices='all') - on a one-GPU box the pool is [0]: what the
thread hop, the per-call table lookup and the shared result array cost.  On a node: the aggregate over its GPUs."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native, multi

fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
x = np.random.default_rng(0).uniform(-1, 1, (1024, 48000, 2)).astype(np.float32)
print('devices visible:', _native.device_count())
for mode, name in ((vnd.MODE_EXACT, 'exact'), (vnd.MODE_FAST, 'fast')):
    for label, call in (('plain', lambda: vnd.convolve_velvet_noise_batched(x, fir, mode=mode)),
                        ("devices='all'", lambda: vnd.convolve_velvet_noise_batched(x, fir, mode=mode, devices='all'))):
        y = call()
        best, reps, t0 = 1e9, 0, time.perf_counter()
        while time.perf_counter() - t0 < 1.5:
            t = time.perf_counter(); y = call(); best = min(best, time.perf_counter() - t); reps += 1
        print(f'{name:5s} {label:14s} best {best * 1e3:7.3f} ms of {reps}  {2 * x.nbytes / best / 1e9:6.1f} GB/s in + out')
    a = vnd.convolve_velvet_noise_batched(x, fir, mode=mode); b = vnd.convolve_velvet_noise_batched(x, fir, mode=mode, devices='all')
    print('      identical:', bool(np.array_equal(a, b)), ' blocks:', multi.pool_for('all').last_blocks, multi.pool_for('all').last_transport)
multi.close_pools()
