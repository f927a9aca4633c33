#!/usr/bin/env python3
"""Window kernel, fast mode: what the store / refill phase costs and whether wave priorities help.
A fresh table per setting (a table keeps its built kernels; these switches are not part of a kernel's key).
  VND_WIN_DEBUG=1  no stores, 4: no barriers (wrong results on purpose)      VND_WIN_PRIO=k  s_setprio k between the tile's two barriers
usage: win_phase_try.py [cfg2|cfg3|cfg4] [seconds per setting]"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
os.environ.setdefault('VND_TUNING', '1')
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays

which = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
ctx = _native.default_context()
if which == 'cfg3':
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1)
    pool, n = 24, 2880000
elif which == 'cfg4':
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
    pool, n = 1024, 48000
else:
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
    pool, n = 512, 480000
arr = function_path_arrays(fir)
x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
ref = None
settings = [dict(), dict(VND_WIN_DEBUG=1), dict(VND_WIN_DEBUG=4), dict(VND_WIN_DEBUG=5), dict(VND_WIN_PRIO=0), dict()]
modes = [2, 0] if len(sys.argv) > 3 and sys.argv[3] == 'both' else [2]
for env in settings:
    for k in ('VND_WIN_DEBUG', 'VND_WIN_PRIO'):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
    for mode in modes:
        table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, stream=st); torch.cuda.synchronize()
        if mode == 2:
            if ref is None:
                ref = y[:2].clone()
            same = bool(torch.equal(y[:2], ref))
        t0 = time.perf_counter(); best = []
        while time.perf_counter() - t0 < seconds:
            best.append(table.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, n_buffers=1, stride_elems=0, iters=30, stream=st))
        tail = best[len(best) // 2:]
        print(f'{which} mode={mode} {env}: {np.mean(tail):.4f} ms {16e-6 * pool * n / np.mean(tail):.0f} GB/s  identical to the first run: {same}   {table.describe(pool, n, 2, mode)[:90]}', flush=True)
