#!/usr/bin/env python3
"""VND_MODE_EXACT on the cfg2 pool: the per-table (hipRTC) kernel against the generic ordered kernel."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
import os
ctx = _native.default_context()
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
pool, n = 128, 480000
x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
def rate(variant, label, **env):
    for k in ('VND_SPEC_NT', 'VND_SPEC_RR', 'VND_SPEC_DD', 'VND_SPEC_LA'):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ['VND_SPEC_' + k.upper()] = str(v)
    ctx.set_variant(variant)
    desc = table.describe(pool, n, 2, 0)
    t0 = time.perf_counter(); best = []
    while time.perf_counter() - t0 < 1.0:
        best.append(table.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=0, n_buffers=1, stride_elems=0, iters=100, stream=stream))
    tail = best[len(best) // 2:]
    print(f'{label:26s} {np.mean(tail):.4f} ms/launch {983.04 / np.mean(tail):6.0f} GB/s  {desc[:150]}', flush=True)
for rep in range(2):
    rate(1 << 25, 'generic ordered')
    rate(1 << 15, 'spec exact')
    rate(1 << 15, 'spec exact la=2', la=2)
    rate(1 << 15, 'spec exact la=4', la=4)
    rate(1 << 15, 'spec exact nt=256 rr=4', nt=256, rr=4)
    rate(1 << 15, 'spec exact nt=128 rr=4', nt=128, rr=4)
    rate(1 << 15, 'spec exact nt=256 rr=2', nt=256, rr=2)
