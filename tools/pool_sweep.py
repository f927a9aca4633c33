#!/usr/bin/env python3
"""Round 6: the plain convolution over resident pools of 8 ... 2048 stereo 10 s signals (and 1 s signals), fast and exact: fraction of
8 TB/s by pool size and the launch the plan picks - a check for dips where a round of workgroups fills badly.  usage: pool_sweep.py [frames]"""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import bench
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
n = int(sys.argv[1]) if len(sys.argv) > 1 else 480000
ctx = _native.default_context()
a = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
t = _native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)
for pool in (8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048):
    if pool * n > 2048 * 480000:
        break
    buffers = max(1, int(600e6 // (pool * n * 16)))
    row = []
    for mode in (vnd.MODE_FAST, vnd.MODE_EXACT):
        t.prepare(pool, n, 2, mode)
        r = bench.device_rate(torch, t, (pool, n, 2), mode, buffers=min(buffers, 8), min_ms=20.0)
        d = r['launch']
        row.append(f"{r['frac_of_8TBs']:.3f} ({d[d.find('workgroups='):d.find(' threads=')][:70]})")
    print(f'{pool:5d} x {n}: fast {row[0]}   exact {row[1]}', flush=True)
    torch.cuda.empty_cache()
