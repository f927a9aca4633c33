#!/usr/bin/env python3
"""Mono in, stereo out (128 x 10 s): the pair-read per-table kernel (one staged plane, reads shared between the two channels'
taps at equal offsets) against the window forms - the plain one (one plane, a pass per channel) and the split one (round 4: 64-frame
runs, the waves split over the two OUTPUT channels, the mono input staged into both plane sets) - fast and exact mode."""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
os.environ.setdefault('VND_TUNING', '1')
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
for rep in range(2):
    for mode in (2, 0):
        for label, v, env in (('automatic', -1, {}), ('pair-read', 1 << 5, {}), ('window 32, a pass per channel', 3 << 5, {'VND_WIN_SPLIT_FANOUT': '0', 'VND_WIN_FANOUT_MERGED': '0'}),
                              ('window 32, merged reads', -1, {'VND_WIN_FANOUT_MERGED': '1'}),
                              ('window 64 split', -1, {'VND_WIN_SPLIT_FANOUT': '1', 'VND_WIN_FANOUT_MERGED': '0'}), ('generic', 1 << 25, {})):
            os.environ.pop('VND_WIN_SPLIT_FANOUT', None)
            os.environ.pop('VND_WIN_FANOUT_MERGED', None)
            os.environ.update(env)
            ctx.set_variant(v)
            for _ in range(30):
                table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 1, mode, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(300):
                table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 1, mode, st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 300
            print(f'mode {mode} {label:30s} {ms:.4f} ms  {pool * n * 2 / ms / 1e3:9.0f} output Msamples/s  {12e-6 * pool * n / ms:6.0f} GB/s (12 B/frame = {12e-9 * pool * n / ms / 8:.3f} of 8 TB/s)  {table.describe(pool, n, 1, mode)[:110]}', flush=True)
ctx.set_variant(-1)
