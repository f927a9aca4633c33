#!/usr/bin/env python3
"""The window kernel with the ring's length rounded up to a multiple of 16 entries (the wrap falls on a bank period) against
the exact-fit ring: cfg2, cfg3, cfg4, fast and exact.  A fresh table per setting."""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
os.environ['VND_TUNING'] = '1'
import numpy as np, torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
for which in ('cfg2', 'cfg3', 'cfg4'):
    kw = dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1); pool, n = 512, 480000
    if which == 'cfg3':
        kw.update(num_impulses=128, log_distribution_strength=0.0); pool, n = 24, 2880000
    if which == 'cfg4':
        pool, n = 1024, 48000
    arr = function_path_arrays(vnd.generate_velvet_noise(**kw))
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1); y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    ref = {}
    for rep in range(2):
        for align in (0, 1):
            os.environ['VND_WIN_ALIGN_RING'] = str(align)
            table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
            for mode in (2, 0):
                table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, stream=st); torch.cuda.synchronize()
                if mode not in ref: ref[mode] = y[:2].clone()
                same = bool(torch.equal(y[:2], ref[mode]))
                best = []
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.8:
                    best.append(table.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, n_buffers=1, stride_elems=0, iters=30, stream=st))
                tail = best[len(best) // 2:]
                print(f'{which} mode={mode} ring aligned={align}  {np.mean(tail):.4f} ms  {8e-6 * pool * n * 2 / np.mean(tail):6.0f} GB/s  same: {same}  {table.describe(pool, n, 2, mode)[52:150]}', flush=True)
            table.close()
    del x, y; torch.cuda.empty_cache()
