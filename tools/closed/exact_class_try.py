#!/usr/bin/env python3
"""VND_MODE_EXACT on the CLASS-path table (VelvetNoise.convolve: +-1 weights, segment gains - what VelvetNoise.decorrelate
runs by default) on the cfg2 pool: the per-table (hipRTC) kernel against the generic ordered kernel."""
import pathlib, sys, time, os
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
ctx = _native.default_context()
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
table = vn._device_table()
pool, n = 128, 480000
x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
ref = None
def rate(variant, label, **env):
    global ref
    for k in ('VND_SPEC_NT', 'VND_SPEC_RR', 'VND_SPEC_DD', 'VND_SPEC_LA'):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ['VND_SPEC_' + k.upper()] = str(v)
    ctx.set_variant(variant)
    desc = table.describe(pool, n, 2, 0)
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=0, stream=stream); torch.cuda.synchronize()
    if ref is None: ref = y.clone()
    same = bool(torch.equal(y, ref))
    t0 = time.perf_counter(); best = []
    while time.perf_counter() - t0 < 1.0:
        best.append(table.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=0, n_buffers=1, stride_elems=0, iters=100, stream=stream))
    tail = best[len(best) // 2:]
    print(f'{label:26s} {np.mean(tail):.4f} ms/launch {983.04 / np.mean(tail):6.0f} GB/s  identical={same}  {desc[:110]}', flush=True)
for rep in range(2):
    rate(1 << 25, 'generic ordered')
    rate(1 << 15, 'spec exact')
    rate(1 << 15, 'spec exact nt=256 rr=4', nt=256, rr=4)
    rate(1 << 15, 'spec exact la=4', la=4)
