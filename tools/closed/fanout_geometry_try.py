#!/usr/bin/env python3
"""Mono -> stereo, throughput mode, per-table kernel (VS_BC): tile geometries on the cfg2 shape (128 x 10 s mono in)."""
import os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
pool, n = 128, 480000
x = torch.empty((pool, n, 1), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
st = torch.cuda.current_stream().cuda_stream
configs = [dict(), dict(nt=256, rr=4), dict(nt=128, rr=4), dict(nt=256, rr=2), dict(nt=192, rr=4, dd=2), dict(nt=384, rr=2), dict(nt=192, rr=2), dict(nt=320, rr=4), dict(nt=384, rr=4)]
for rnd in range(2):
    for c in configs:
        for k in ('VND_SPEC_NT', 'VND_SPEC_RR', 'VND_SPEC_DD', 'VND_SPEC_LA'):
            os.environ.pop(k, None)
        for k, v in c.items():
            os.environ['VND_SPEC_' + k.upper()] = str(v)
        ctx.set_variant(-1)
        def run(): table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 1, 2, st)
        for _ in range(20): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(200): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
        print(f'{str(c):28s} {ms:.4f} ms  {pool * n * 2 / ms / 1e6:6.0f} Gsamples/s  {table.describe(pool, n, 1, 2)[30:150]}', flush=True)
