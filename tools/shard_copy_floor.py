#!/usr/bin/env python3
"""The floor under a small pass: a plain device copy of the N = 8 / 4 / 2 / 1 shards of cfg4 (128 ... 1024 x 1 s stereo,
rotating buffers so that nothing stays in the caches), next to the convolution of the same shard."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
n = 48000
stream = torch.cuda.current_stream().cuda_stream
for ranks in (8, 4, 2, 1):
    mine = 1024 // ranks
    buffers = max(1, int(np.ceil(600e6 / (mine * n * 2 * 4 * 2))))
    xs = [torch.empty((mine, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1) for _ in range(buffers)]
    ys = [torch.empty_like(xs[0]) for _ in range(buffers)]
    res = {}
    for label, fn in (('copy', lambda i: ys[i % buffers].copy_(xs[i % buffers])),
                      ('convolve', lambda i: table.convolve_device(xs[i % buffers].data_ptr(), ys[i % buffers].data_ptr(), mine, n, 2, 2, stream))):
        for i in range(200): fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(400): fn(i)
        e1.record(); torch.cuda.synchronize()
        res[label] = e0.elapsed_time(e1) / 400 * 1e3
    mb = mine * n * 2 * 4 * 2 / 1e6
    print(f'N={ranks}: {mine:4d} streams, {mb:6.1f} MB per pass: copy {res["copy"]:6.1f} us ({mb / res["copy"] / 1e3 * 1e3:5.2f} TB/s)   convolve {res["convolve"]:6.1f} us ({mb / res["convolve"]:5.2f} TB/s)   ratio {res["copy"] / res["convolve"]:.2f}', flush=True)
    del xs, ys
    torch.cuda.empty_cache()
