#!/usr/bin/env python3
"""Mono -> stereo on the cfg2 shape (pool of 128 x 10 s mono in, stereo out; 12 B per frame): the per-table kernels (one LDS
plane, VS_BC) against the generic fan-out kernels, fast and exact mode, function-path and class-path tables."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
tables = {'function path': _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight),
          'class path': vnd.VelvetNoise(sample_rate_hz=48000, seed=1)._device_table()}
pool, n = 128, 480000
x = torch.empty((pool, n, 1), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for rnd in range(2):
    for name, table in tables.items():
        for mode, mname in ((2, 'fast'), (0, 'exact')):
            for label, variant in (('generic', 1 << 25), ('per-table', 1 << 15)):
                ctx.set_variant(variant)
                def run(): table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 1, mode, st)
                for _ in range(20): run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(200): run()
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 200
                print(f'{name:14s} {mname:5s} {label:9s} {ms:.4f} ms  {12e-6 * pool * n / ms:5.0f} GB/s  {pool * n * 2 / ms / 1e6:6.0f} Gsamples/s  {table.describe(pool, n, 1, mode)[:44]}', flush=True)
ctx.set_variant(-1)
