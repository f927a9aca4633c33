#!/usr/bin/env python3
"""Fast mode, 64-frame split runs with (almost) the whole refill loaded late, one fresh table per setting (a table remembers a
rejected build per geometry, and the late share is not part of the geometry key): cfg2 / class path / cfg3 against the plain form."""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
os.environ.update(VND_TUNING='1')
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
st = torch.cuda.current_stream().cuda_stream
KW = {'cfg2': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), 128, 480000),
      'cfg3': (dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1), 24, 2880000)}
SETTINGS = [('plain', dict(VND_WIN_SPLIT='0'), -1),
            ('split 64 late=15 la=2', dict(VND_WIN_SPLIT='2', VND_SPEC_NT='256', VND_WIN_SPLIT_LATE='15', VND_SPEC_LA='2'), 4 << 5),
            ('split 64 late=12 la=2', dict(VND_WIN_SPLIT='2', VND_SPEC_NT='256', VND_WIN_SPLIT_LATE='12', VND_SPEC_LA='2'), 4 << 5)]
for name, (kw, pool, n) in KW.items():
    a = function_path_arrays(vnd.generate_velvet_noise(**kw))
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty_like(x)
    ref = None
    for rep in range(2):
        for label, env, variant in SETTINGS:
            for k in ('VND_WIN_SPLIT', 'VND_SPEC_NT', 'VND_WIN_SPLIT_LATE', 'VND_SPEC_LA'):
                os.environ.pop(k, None)
            os.environ.update(env)
            t = _native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)
            ctx.set_variant(variant)
            desc = t.describe(pool, n, 2, 2)
            t.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=2, stream=st)
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            err = float((y - ref).abs().max() / ref.abs().max())
            t0 = time.perf_counter(); best = []
            while time.perf_counter() - t0 < 1.0:
                best.append(t.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=2, n_buffers=1, stride_elems=0, iters=30, stream=st))
            tail = best[len(best) // 2:]
            print(f'{name} {label:24s} {np.mean(tail):.4f} ms (min {min(best):.4f})  vs plain {err:.1e}  {desc[:150]}', flush=True)
            ctx.set_variant(-1)
            t.close()
