#!/usr/bin/env python3
"""Spans per resident workgroup slot ("rounds", variant bits 28-30; 0 = the plan's choice: the fewest spans that fill one round) on the
bench pools: do shorter spans - more units than workgroups, taken in turn - even out the workgroups' finishing times?
(cfg3's PMC pass: 1.52 waves resident per SIMD on average where 2 fit.)"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
st = torch.cuda.current_stream().cuda_stream
KW = {'cfg3': (dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1), (24, 2880000, 2)),
      'cfg2x2048': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (2048, 480000, 2)),
      'cfg5': (dict(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1), (16, 960000, 8)),
      'cfg4': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (1024, 48000, 2))}
only = [a for a in sys.argv[1:] if a in KW] or list(KW)
for name in only:
    kw, shape = KW[name]
    a = function_path_arrays(vnd.generate_velvet_noise(**kw))
    t = _native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)
    x = torch.empty(shape, dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty_like(x)
    for rep in range(2):
        for mode in (2, 0):
            for rounds in (0, 1, 2, 3, 4, 6):
                ctx.set_variant(rounds << 28 if rounds else -1)
                desc = t.describe(*shape, mode)
                iters = 10 if shape[0] >= 2048 else 30
                t0 = time.perf_counter(); best = []
                while time.perf_counter() - t0 < 0.7:
                    best.append(t.time_device(x.data_ptr(), y.data_ptr(), *shape, mode=mode, n_buffers=1, stride_elems=0, iters=iters, stream=st))
                tail = best[len(best) // 2:]
                units = desc[desc.index('workgroups='):desc.index(' threads=')]
                print(f'{name} mode {mode} rounds {rounds}: {np.mean(tail):.4f} ms (min {min(best):.4f})  {units}', flush=True)
    ctx.set_variant(-1)
    t.close()
    del x, y
    torch.cuda.empty_cache()
