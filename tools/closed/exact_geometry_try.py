#!/usr/bin/env python3
"""VND_MODE_EXACT, per-table kernel with the shifted plane copies: tile geometries on the cfg2 pool, function- and class-path tables."""
import os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
tables = {'function': _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight),
          'class': vnd.VelvetNoise(sample_rate_hz=48000, seed=1)._device_table()}
pool, n = 128, 480000
x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
configs = [dict(), dict(nt=256, rr=2), dict(nt=384, rr=2), dict(nt=320, rr=2), dict(nt=192, rr=2), dict(nt=512, rr=2), dict(nt=256, rr=3), dict(nt=192, rr=4), dict(nt=256, rr=2, la=4), dict(nt=256, rr=2, dd=1)]
for rnd in range(2):
    for name, table in tables.items():
        for c in configs:
            for k in ('VND_SPEC_NT', 'VND_SPEC_RR', 'VND_SPEC_DD', 'VND_SPEC_LA'):
                os.environ.pop(k, None)
            for k, v in c.items():
                os.environ['VND_SPEC_' + k.upper()] = str(v)
            ctx.set_variant(1 << 15)
            def run(): table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, 0, st)
            for _ in range(10): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(100): run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 100
            print(f'{name:9s} {str(c):34s} {ms:.4f} ms {983.04 / ms:6.0f} GB/s  {table.describe(pool, n, 2, 0)[38:120]}', flush=True)
ctx.set_variant(-1)
