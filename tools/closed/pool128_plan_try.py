#!/usr/bin/env python3
"""Launches of 128 ten-second streams (the f1 pool, the mono fan-out of the bench): the CU-chunk plan (a chunk of 30 tiles per CU as
17 + 13) against uniform spans (4 x 15 tiles) with and without pacing - stereo and mono input, fast and exact; interleaved repeats."""
import os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
os.environ['VND_TUNING'] = '1'
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
pool, n = 128, 480000
st = torch.cuda.current_stream().cuda_stream
settings = [dict(), dict(VND_WIN_CHUNKS='0'), dict(VND_WIN_CHUNKS='0', VND_WIN_PACE_MIN_TILES='8'), dict(VND_WIN_CHUNK_LEN0='15'), dict(VND_WIN_CHUNK_LEN0='16')]
if len(sys.argv) > 1:          # settings from the command line: K=V,K=V groups (an empty group "-" is the default)
    settings = [dict(kv.split('=', 1) for kv in g.split(',') if '=' in kv) for g in sys.argv[1:]]
names = sorted({k for g in settings for k in g} | {'VND_WIN_CHUNKS', 'VND_WIN_PACE_MIN_TILES', 'VND_WIN_CHUNK_LEN0'})
for cx in (1, 2):
    x = torch.empty((pool, n, cx), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
    for rep in range(2):
        for env in settings:
            for k in names:
                os.environ.pop(k, None)
            os.environ.update(env)
            table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
            out = []
            for mode in (2, 0):
                for _ in range(40):
                    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, cx, mode, st)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(300):
                    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, cx, mode, st)
                e1.record(); torch.cuda.synchronize()
                out.append(e0.elapsed_time(e1) / 300)
            d = table.describe(pool, n, cx, 2)
            print(f'{"mono  " if cx == 1 else "stereo"} {str(env):70s} fast {out[0]:.4f} ms  exact {out[1]:.4f} ms   {d[d.find("reads_ahead="):][:60]}', flush=True)
            table.close()
