#!/usr/bin/env python3
"""What one rank of the cfg4 strong-scaling leg does at N = 1, 2, 4, 8 GPUs (1024 / N streams of 1 s stereo per
pass, rotating buffers), measured on ONE GPU: per-pass time through the Python launch loop bench.py uses, and
through the C loop (vnd_time_convolve) that shows the kernel alone."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
n = 48000
stream = torch.cuda.current_stream().cuda_stream
base = None
for ranks in (1, 2, 4, 8):
    mine = 1024 // ranks
    buffers = max(1, int(np.ceil(600e6 / (mine * n * 2 * 4 * 2))))
    xs = [torch.empty((mine, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1) for _ in range(buffers)]
    ys = [torch.empty_like(xs[0]) for _ in range(buffers)]
    for variant, label in ((-1, 'auto'), (1 << 25, 'generic'), ((1 << 23) | (4 << 20), 'spec forced, spans >= 4 tiles')):
        ctx.set_variant(variant)
        def step(i):
            table.convolve_device(xs[i % buffers].data_ptr(), ys[i % buffers].data_ptr(), mine, n, 2, 2, stream)
        for i in range(300): step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(400): step(i)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / 400
        if ranks == 1 and label == 'auto': base = per
        print(f'N={ranks} ({mine:4d} streams, {buffers} buffers) {label:32s} {per * 1e6:7.1f} us per pass   scaling vs N=1 auto: {base / per:4.2f}x   {table.describe(mine, n, 2, 2)[:60]}', flush=True)
    del xs, ys
    torch.cuda.empty_cache()
ctx.set_variant(-1)
