#!/usr/bin/env python3
"""cfg5 (96 kHz, 8 channels, 64 taps): the generic fast kernel against the per-table kernel (forced: it is
off by default for more than two channels) in several geometries."""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
os.environ.setdefault('VND_TUNING', '1')      # geometry variables are read live
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1)
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
pool, n, C = 16, 960000, 8
x = torch.empty((pool, n, C), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
ctx.set_variant(1 << 25)
table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, C, mode=2, stream=stream); torch.cuda.synchronize()
ref = y.clone()
def rate(variant, label, **env):
    for k in ('VND_SPEC_NT', 'VND_SPEC_RR', 'VND_SPEC_DD', 'VND_SPEC_LA'):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ['VND_SPEC_' + k.upper()] = str(v)
    ctx.set_variant(variant)
    desc = table.describe(pool, n, C, 2)
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, C, mode=2, stream=stream); torch.cuda.synchronize()
    err = float((y - ref).abs().max()) / float(ref.abs().max())
    t0 = time.perf_counter(); best = []
    while time.perf_counter() - t0 < 1.0:
        best.append(table.time_device(x.data_ptr(), y.data_ptr(), pool, n, C, mode=2, n_buffers=1, stride_elems=0, iters=40, stream=stream))
    tail = best[len(best) // 2:]
    print(f'{label:26s} {np.mean(tail):.4f} ms/launch {8e-6 * pool * n * C / np.mean(tail):6.0f} GB/s  err {err:.1e}  {desc[:140]}', flush=True)
F = 1 << 23
for rep in range(2):
    rate(1 << 25, 'generic')
    rate(F, 'spec default')
    rate(F, 'spec nt=256 rr=4', nt=256, rr=4)
    rate(F, 'spec nt=256 rr=2', nt=256, rr=2)
    rate(F, 'spec nt=128 rr=4', nt=128, rr=4)
    rate(F, 'spec nt=192 rr=2', nt=192, rr=2)
    rate(F, 'spec nt=128 rr=2', nt=128, rr=2)
    rate(F, 'spec nt=64 rr=4', nt=64, rr=4)
    rate(F, 'spec nt=128 rr=4 dd=1', nt=128, rr=4, dd=1)
    rate(F, 'spec nt=128 rr=4 la=4', nt=128, rr=4, la=4)
    rate(F, 'spec nt=128 rr=4 la=12', nt=128, rr=4, la=12)
    rate(F, 'spec nt=192 rr=4', nt=192, rr=4)
