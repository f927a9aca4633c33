#!/usr/bin/env python3
"""Device-resident rate of the fan-out launches: mono -> stereo (cfg2 shape) and one stereo
signal through a bank of F filter pairs."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import concat_tap_arrays, function_path_arrays

ctx = _native.default_context()
stream = torch.cuda.current_stream().cuda_stream


def rate(table, x, y, batch, n, cin, mode, label, iters=300):
    for _ in range(3):
        ms = 0.0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(iters // 3):
            table.convolve_device(x.data_ptr(), y.data_ptr(), batch, n, cin, mode, stream)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            table.convolve_device(x.data_ptr(), y.data_ptr(), batch, n, cin, mode, stream)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
    out_samples = y.numel()
    moved = 4 * (x.numel() + y.numel())
    print(f'{label:44s} {table.describe(batch, n, cin, mode).split()[0]:20s} {ms:.4f} ms  '
          f'{out_samples / ms / 1e3:9.0f} Msamples/s  {moved / ms / 1e6:6.0f} GB/s moved')


fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
pool, n = 128, 480000
xs = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
xm = torch.empty((pool, n, 1), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
for mode, name in ((2, 'fast'), (0, 'exact')):
    rate(table, xs, y, pool, n, 2, mode, f'stereo -> stereo, pool 128, {name}')
    rate(table, xm, y, pool, n, 1, mode, f'mono -> stereo (fan-out), pool 128, {name}')
del xs, xm, y

# one 10 s stereo signal through F candidate filter pairs (the optimiser's grid: F = 400)
F = 400
firs = [vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=s)
        for s in range(F)]
bank = concat_tap_arrays([function_path_arrays(f) for f in firs])
tb = _native.TapTable.create(ctx, bank.tap_offsets, bank.tap_index, bank.tap_weight)
x1 = torch.empty((1, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
yb = torch.empty((1, n, 2 * F), dtype=torch.float32, device='cuda')
for mode, name in ((2, 'fast'), (0, 'exact')):
    rate(tb, x1, yb, 1, n, 2, mode, f'1 signal x {F} filter pairs (bank), {name}', iters=60)
