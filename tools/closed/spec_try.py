#!/usr/bin/env python3
"""Specialised (hipRTC) fast kernel against the generic one on the cfg2 pool: parity and sustained rate.
usage: spec_try.py [seconds per variant]"""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import os
import numpy as np
os.environ.setdefault('VND_TUNING', '1')      # geometry variables are read live
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
which = sys.argv[2] if len(sys.argv) > 2 else 'cfg2'
ctx = _native.default_context()
if which == 'cfg3':
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000,
                                    log_distribution_strength=0.0, seed=1)
    pool, n = 24, 2880000
elif which == 'cfg4':
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
    pool, n = 1024, 48000
else:
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
    pool, n = 128, 480000
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
GENERIC = 1 << 25


def rate(variant, label):
    ctx.set_variant(variant)
    desc = table.describe(pool, n, 2, 2)
    t0 = time.perf_counter(); best = []
    while time.perf_counter() - t0 < seconds:
        best.append(table.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=2, n_buffers=1, stride_elems=0,
                                      iters=100, stream=stream))
    tail = best[len(best) // 2:]
    print(f'{label:28s} {np.mean(tail):.4f} ms/launch {8e-6 * pool * n * 2 / np.mean(tail):6.0f} GB/s  (min {min(best):.4f})  {desc}', flush=True)


# parity first: spec vs exact (oracle-identical) on a few streams, and the whole pool vs the generic fast kernel
ctx.set_variant(-1)
print(table.describe(pool, n, 2, 2), flush=True)
table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=2, stream=stream)
torch.cuda.synchronize()
y_spec = y.clone()
ctx.set_variant(GENERIC)
table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=2, stream=stream)
torch.cuda.synchronize()
peak = float(y.abs().max())
d = (y_spec - y).abs()
print(f'spec vs generic fast: max |diff| {float(d.max()):.3e} = {float(d.max()) / peak:.2e} of peak {peak:.3f}', flush=True)
ye = torch.empty((4, n, 2), dtype=torch.float32, device='cuda')
table.convolve_device(x[:4].contiguous().data_ptr(), ye.data_ptr(), 4, n, 2, mode=0, stream=stream)
torch.cuda.synchronize()
d = (y_spec[:4] - ye).abs()
print(f'spec vs exact (4 streams): {float(d.max()) / peak:.2e} of peak; worst frame {int(d.amax(dim=(0, 2)).argmax())}', flush=True)
bad = (d.amax(dim=2) > 2e-6 * peak).nonzero()
print('frames off by more than 2e-6 of peak:', bad[:10].tolist(), flush=True)

import os


def env_rate(label, **env):
    for k in ('VND_SPEC_NT', 'VND_SPEC_RR', 'VND_SPEC_DD', 'VND_SPEC_LA'):
        os.environ.pop(k, None)
    for k, val in env.items():
        os.environ['VND_SPEC_' + k.upper()] = str(val)
    # parity of this geometry first (a wrong ring layout must not hide behind a good rate)
    ctx.set_variant(-1)
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=2, stream=stream)
    torch.cuda.synchronize()
    err = float((y[:4] - ye).abs().max()) / peak
    rate(-1, f'{label} err={err:.1e}')


configs = [dict(), dict(nt=192, rr=4), dict(nt=192, rr=4, dd=2), dict(nt=128, rr=4, la=4), dict(nt=256, rr=2), dict(nt=256, rr=4), dict(nt=128, rr=4, dd=1)]
if len(sys.argv) > 3 and sys.argv[3] == 'wide':       # the round-2 closing sweep: every shape the library would consider, and read depths
    configs = [dict(), dict(nt=192, rr=4, la=12), dict(nt=192, rr=4, la=6), dict(nt=256, rr=4), dict(nt=256, rr=4, la=12), dict(nt=128, rr=4),
               dict(nt=128, rr=8), dict(nt=192, rr=2), dict(nt=256, rr=2, dd=2), dict(nt=384, rr=2), dict(nt=64, rr=8), dict(nt=192, rr=4, dd=2)]
for rep in range(2):
    rate(GENERIC, 'generic fast')
    for c in configs:
        env_rate('spec ' + ' '.join(f'{k}={v}' for k, v in c.items()), **c)
