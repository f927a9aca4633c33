#!/usr/bin/env python3
"""cfg5 (96 kHz, 8 channels, 64 taps; pool of 16 ten-second signals) and cfg3 through bench.py's own timing loop, fast and exact, for the
settings given on the command line (KEY=VALUE ..., read live: VND_TUNING) - one fresh table per call.  usage: cfg5_try.py [cfg2|cfg3] KEY=VALUE ...   (cfg2: a pool of 512 signals)"""
import os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
os.environ['VND_TUNING'] = '1'
which = 'cfg3' if 'cfg3' in sys.argv[1:] else ('cfg2' if 'cfg2' in sys.argv[1:] else 'cfg5')
for kv in sys.argv[1:]:
    if '=' in kv:
        k, v = kv.split('=', 1); os.environ[k] = v
import torch
import bench
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
if which == 'cfg2':
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1); shape = (512, 480000, 2)
elif which == 'cfg5':
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1); shape = (16, 960000, 8)
else:
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1); shape = (24, 2880000, 2)
a = function_path_arrays(fir)
taps = (a.tap_offsets, a.tap_index, a.tap_weight)
t = _native.TapTable.create(ctx, *taps)
for mode, name in ((vnd.MODE_FAST, 'fast'), (vnd.MODE_EXACT, 'exact')):
    r = bench.device_rate(torch, t, shape, mode, taps=taps)
    print(f'{which} {name:5s} {r["kernel_ms"]:.4f} ms  {r["frac_of_8TBs"]:.4f} of 8 TB/s  parity {r["parity_vs_oracle_of_peak"]:.1e}  {r["launch"][r["launch"].find("frames_per_lane"):][:150]}', flush=True)
