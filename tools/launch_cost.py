#!/usr/bin/env python3
"""Host cost of one vnd_convolve_f32_dev call through ctypes (the Python launch loop of bench.py's legs), and the
kernel-only time of the N = 8 shard of cfg4 through the C loop."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
stream = torch.cuda.current_stream().cuda_stream
x = torch.empty((128, 48000, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
xs = torch.empty((7, 128, 48000, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
ys = torch.empty_like(xs)
for shape, label in (((1, 64, 2), 'tiny launch (host-bound)'), ((128, 48000, 2), 'N=8 shard')):
    b, n, c = shape
    for _ in range(200): table.convolve_device(x.data_ptr(), y.data_ptr(), b, n, c, 2, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): table.convolve_device(x.data_ptr(), y.data_ptr(), b, n, c, 2, stream)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'{label:28s} issue {t_issue / 2000 * 1e6:6.2f} us per call, with the GPU {t_all / 2000 * 1e6:6.2f} us per call')
ms = table.time_device(xs.data_ptr(), ys.data_ptr(), 128, 48000, 2, mode=2, n_buffers=7, stride_elems=128 * 48000 * 2, iters=2000, stream=stream)
print(f'N=8 shard, C launch loop, 7 rotating buffers: {ms * 1e3:.2f} us per launch')
for v, label in ((2, 'pairs 2'), (3, 'pairs 3'), (8, 'pairs 8'), (4, 'pairs 4')):
    ctx.set_variant(v | (1 << 25))
    ms = table.time_device(xs.data_ptr(), ys.data_ptr(), 128, 48000, 2, mode=2, n_buffers=7, stride_elems=128 * 48000 * 2, iters=2000, stream=stream)
    print(f'  generic {label}: {ms * 1e3:.2f} us   {table.describe(128, 48000, 2, 2)[:90]}')
ctx.set_variant(-1)
