#!/usr/bin/env python3
"""Wider signals through the pair-read per-table kernel with and without the lane-quad exchange before the stores
(VS_QUAD_STORES: one store instruction writes four consecutive frames' pieces instead of every other frame's).
usage: c8_quad_try.py [channels=8]   (8: cfg5's table; 4 / 6: 48 kHz, 30 taps per channel)"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
os.environ['VND_TUNING'] = '1'
import numpy as np, torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if C == 8:
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1)
    pool, n = 16, 960000
else:
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=C, sample_rate_hz=48000, seed=1)
    pool, n = 64, 480000
arr = function_path_arrays(fir)
x = torch.empty((pool, n, C), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
ref = None
for quad in (0, 1, 0, 1):
    os.environ['VND_SPEC_QUAD_STORES'] = str(quad)
    table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
    for mode in (2, 0):
        ctx.set_variant(-1)
        table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, C, mode=mode, stream=st); torch.cuda.synchronize()
        if mode == 0:
            if ref is None: ref = y.clone()
            same = bool(torch.equal(y, ref))
        else:
            fast = y.clone()
        best = [table.time_device(x.data_ptr(), y.data_ptr(), pool, n, C, mode=mode, n_buffers=1, stride_elems=0, iters=40, stream=st) for _ in range(12)]
        print(f'quad={quad} mode={mode}: {np.mean(best[6:]):.4f} ms {8e-6*pool*n*C/np.mean(best[6:]):.0f} GB/s', table.describe(pool, n, C, mode)[:60], flush=True)
    print('   exact equal to first run:', same, ' fast vs exact of peak:', float((fast - ref).abs().max()) / float(ref.abs().max()), flush=True)
