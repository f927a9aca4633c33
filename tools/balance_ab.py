#!/usr/bin/env python3
"""Round 6: the window form's BALANCED cut (every workgroup a contiguous range of the pool's tiles: VND_WIN_BALANCE=1, taken where its cost
model says so) against uniform spans (=0), interleaved, over pool sizes that fill the one round of workgroups unevenly; fast and exact."""
import os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
os.environ['VND_TUNING'] = '1'
import torch
import bench
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
n = 480000
ctx = _native.default_context()
a = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
t = _native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)
for pool in ([int(v) for v in sys.argv[1:]] or [144, 160, 192, 224, 288, 320, 448, 576, 640]):
    for mode, name in ((vnd.MODE_FAST, 'fast'), (vnd.MODE_EXACT, 'exact')):
        row = []
        for r in range(2):
            for bal in ('0', '1'):
                os.environ['VND_WIN_BALANCE'] = bal
                t.prepare(pool, n, 2, mode)
                rec = bench.device_rate(torch, t, (pool, n, 2), mode, buffers=max(1, min(4, int(600e6 // (pool * n * 16)))), min_ms=20.0)
                d = rec['launch']
                row.append(f"{'bal' if 'balanced' in d else 'spans'} {rec['frac_of_8TBs']:.3f}")
        print(f'{pool:4d} {name:5s}: ' + '   '.join(row), flush=True)
