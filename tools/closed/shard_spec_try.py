#!/usr/bin/env python3
"""The N = 8 shard of cfg4 (128 x 1 s stereo) through forced per-table kernels of several geometries (C launch loop,
7 rotating buffers), against the generic kernel."""
import os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
stream = torch.cuda.current_stream().cuda_stream
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nb = 7
xs = torch.empty((nb, streams, 48000, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
ys = torch.empty_like(xs)
def rate(label, variant, **env):
    for k in ('VND_SPEC_NT', 'VND_SPEC_RR', 'VND_SPEC_DD', 'VND_SPEC_LA'):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ['VND_SPEC_' + k.upper()] = str(v)
    ctx.set_variant(variant)
    desc = table.describe(streams, 48000, 2, 2)
    best = min(table.time_device(xs.data_ptr(), ys.data_ptr(), streams, 48000, 2, mode=2, n_buffers=nb, stride_elems=streams * 48000 * 2, iters=1000, stream=stream) for _ in range(3))
    print(f'{label:34s} {best * 1e3:6.2f} us   {desc[:150]}', flush=True)
F = 1 << 23
rate('generic', 1 << 25)
for ms in (1, 2, 3, 4, 6):
    rate(f'spec default geometry min_span={ms}', F | (ms << 20))
for nt, rr in ((128, 2), (128, 4), (192, 2), (256, 2), (256, 1), (128, 1)):
    for ms in (2, 4):
        rate(f'spec nt={nt} rr={rr} min_span={ms}', F | (ms << 20), nt=nt, rr=rr)
    rate(f'spec nt={nt} rr={rr} min_span=4 dd=3', F | (4 << 20), nt=nt, rr=rr, dd=3)
