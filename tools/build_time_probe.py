#!/usr/bin/env python3
"""How long the first launch of a new tap table waits for hipRTC (code-object cache off): describe() builds the kernel the launch
would take, both modes, the four bench tables."""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
os.environ['VND_SPEC_CACHE_DIR'] = 'off'
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
KW = {'cfg2': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (128, 480000, 2)),
      'cfg3': (dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1), (24, 2880000, 2)),
      'cfg5': (dict(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1), (16, 960000, 8))}
for name, (kw, shape) in KW.items():
    a = function_path_arrays(vnd.generate_velvet_noise(**kw))
    for mode, label in ((2, 'fast'), (0, 'exact')):
        t = _native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)
        t0 = time.perf_counter()
        text = t.describe(*shape, mode)
        print(f'{name} {label}: {time.perf_counter() - t0:5.1f} s   {text[:150]}', flush=True)
        t.close()
cls = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)._device_table()
for mode, label in ((2, 'fast'), (0, 'exact')):
    t0 = time.perf_counter()
    text = cls.describe(128, 480000, 2, mode)
    print(f'class path {label}: {time.perf_counter() - t0:5.1f} s   {text[:150]}', flush=True)
