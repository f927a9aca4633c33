#!/usr/bin/env python3
"""Class-path table (VelvetNoise.convolve's segments), exact and fast: the plain window form against 64-frame split runs, a fresh
table per setting."""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
os.environ.update(VND_TUNING='1', VND_SPEC_VERBOSE='1')
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
ctx = _native.default_context()
st = torch.cuda.current_stream().cuda_stream
pool, n = 128, 480000
x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
SETTINGS = [('plain', dict(VND_WIN_SPLIT='0'), -1)] + [
    (f'split 64 late={late} la={la}', dict(VND_WIN_SPLIT='2', VND_SPEC_NT='256', VND_WIN_SPLIT_LATE=str(late), VND_SPEC_LA=str(la)), 4 << 5)
    for late, la in ((8, 3), (15, 2), (15, 3), (12, 2))]
for mode in (0, 2):
    ref = None
    for rep in range(2):
        for label, env, variant in SETTINGS:
            for k in ('VND_WIN_SPLIT', 'VND_SPEC_NT', 'VND_WIN_SPLIT_LATE', 'VND_SPEC_LA'):
                os.environ.pop(k, None)
            os.environ.update(env)
            t = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)._generate_device_table() if hasattr(vnd.VelvetNoise, '_generate_device_table') else None
            vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
            t = vn._device_table()
            ctx.set_variant(variant)
            desc = t.describe(pool, n, 2, mode)
            t.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, stream=st)
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            same = bool(torch.equal(y, ref)); err = float((y - ref).abs().max() / ref.abs().max())
            t0 = time.perf_counter(); best = []
            while time.perf_counter() - t0 < 0.8:
                best.append(t.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, n_buffers=1, stride_elems=0, iters=30, stream=st))
            tail = best[len(best) // 2:]
            print(f'class mode {mode} {label:24s} {np.mean(tail):.4f} ms (min {min(best):.4f})  identical {same} ({err:.1e})  {desc[:120]}', flush=True)
            ctx.set_variant(-1)
            del vn, t
