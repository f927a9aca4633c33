#!/usr/bin/env python3
"""Round 6: which kernels the drop-in calls actually run (under `rocprofv3 --kernel-trace --stats`, one scenario per invocation) - a check
that no default path falls to an older, slower form unnoticed (how the mono decorrelate stage was found on the pair-read form).
usage: rocprofv3 --kernel-trace --stats -d out -o p --output-format csv -- python3 tools/api_audit.py <scenario>
scenarios: dec_stereo_1, dec_mono_1, dec_stereo_128, dec_mono_128, dec_mono_512, conv_class_128, fn_batched_128, fn_mono_128, chain_1, c8_dec_16"""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import vndecorrelate_amd.decorrelation as vnd

which = sys.argv[1]
rng = np.random.default_rng(1)
n = 480000
f32 = lambda *shape: rng.uniform(-1, 1, shape).astype(np.float32)
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
if which == 'dec_stereo_1':
    x = f32(n, 2); run = lambda: vn.decorrelate(x)
elif which == 'dec_mono_1':
    x = f32(n); run = lambda: vn.decorrelate(x)
elif which == 'dec_stereo_128':
    x = f32(128, n, 2); run = lambda: vn.decorrelate_batched(x)
elif which == 'dec_mono_128':
    x = f32(128, n, 1); run = lambda: vn.decorrelate_batched(x)
elif which == 'dec_mono_512':
    x = f32(512, n // 4, 1); run = lambda: vn.decorrelate_batched(x)
elif which == 'conv_class_128':
    x = f32(128, n, 2); run = lambda: [vn.convolve(x[b]) for b in range(0, 128, 16)]
elif which == 'fn_batched_128':
    x = f32(128, n, 2); run = lambda: vnd.convolve_velvet_noise_batched(x, fir)
elif which == 'fn_mono_128':
    x = f32(128, n, 1); run = lambda: vnd.convolve_velvet_noise_batched(x, fir)
elif which == 'chain_1':
    x = f32(n, 2)
    chain = vnd.SignalChain(sample_rate_hz=48000, device_resident=True).velvet_noise(seed=1).haas_effect(delay_time_seconds=0.02, delayed_channel=1, mode='LR')
    run = lambda: chain(x)
elif which == 'c8_dec_16':
    v8 = vnd.VelvetNoise(sample_rate_hz=96000, num_outs=8, num_impulses=64, filtered_channels=tuple(range(8)), mode='LR', seed=1)
    x = f32(16, 960000, 8); run = lambda: v8.decorrelate_batched(x)
else:
    raise SystemExit(__doc__)
import time
run()
t0 = time.perf_counter()
for _ in range(5):
    run()
print(f'{which}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per call, host to host', flush=True)
