#!/usr/bin/env python3
"""One signal host to host: the staged path (vnd_convolve_f32_host: H2D + kernel + D2H) against the kernel working
directly on page-locked host memory (reads and writes cross PCIe inside the launch, both directions at once).
usage: zero_copy_try.py"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
os.environ.setdefault('VND_TUNING', '1')      # VND_HOST_DIRECT is read live
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays

ctx = _native.default_context()
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
st = torch.cuda.current_stream().cuda_stream
for batch, n in ((1, 480000), (1, 2880000), (8, 480000), (64, 48000), (1024, 48000)):
    x = torch.empty((batch, n, 2), dtype=torch.float32).uniform_(-1, 1).pin_memory()
    y = torch.empty_like(x).pin_memory()
    xd = x.cuda(); yd = torch.empty_like(xd)
    for mode in (2, 0):
        # kernel straight on the page-locked buffers
        def zero_copy():
            table.convolve_device(x.data_ptr(), y.data_ptr(), batch, n, 2, mode=mode, stream=st)
            torch.cuda.synchronize()
        def staged():
            os.environ['VND_HOST_DIRECT'] = '0'
            try:
                return table.convolve_host(x.numpy(), mode)
            finally:
                os.environ.pop('VND_HOST_DIRECT')
        def library():                                         # page-locked input, result from the page-locked pool: in place
            return table.convolve_host(x.numpy(), mode)
        def with_direct(k, arr):
            def fn():
                os.environ['VND_HOST_DIRECT'] = str(k)
                try:
                    return table.convolve_host(arr, mode)
                finally:
                    os.environ.pop('VND_HOST_DIRECT')
            return fn
        def read_host_write_device():
            table.convolve_device(x.data_ptr(), yd.data_ptr(), batch, n, 2, mode=mode, stream=st)
            torch.cuda.synchronize()
        def read_device_write_host():
            table.convolve_device(xd.data_ptr(), y.data_ptr(), batch, n, 2, mode=mode, stream=st)
            torch.cuda.synchronize()
        xp = x.numpy().copy()                                  # pageable
        def staged_pageable():
            return table.convolve_host(xp, mode)
        xp_t = torch.from_numpy(xp)
        def upload_then_write_host():
            xd.copy_(xp_t)
            table.convolve_device(xd.data_ptr(), y.data_ptr(), batch, n, 2, mode=mode, stream=st)
            torch.cuda.synchronize()
        want = staged()
        y.zero_(); zero_copy()
        same = bool(np.array_equal(y.numpy(), want))
        res = {}
        for name, fn in (('staged', staged), ('library call', library), ('zero-copy', zero_copy), ('host->device only', read_host_write_device), ('device->host only', read_device_write_host),
                         ('pageable in: staged', with_direct(0, xp)), ('pageable in: library', staged_pageable), ('pageable in: upload + kernel writes host', upload_then_write_host)):
            for _ in range(3): fn()
            ts = []
            for _ in range(30):
                t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            res[name] = 1e3 * float(np.median(ts))
        mb = x.numel() * 4 / 1e6
        print(f'batch={batch} n={n} mode={mode} ({mb:.1f} MB each way): ' + '  '.join(f'{k} {v:.3f} ms' for k, v in res.items()) +
              f'   zero-copy {2 * mb / res["zero-copy"]:.1f} GB/s in+out, identical to the staged result: {same}   {table.describe(batch, n, 2, mode)[:40]}', flush=True)
