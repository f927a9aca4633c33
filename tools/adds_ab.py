#!/usr/bin/env python3
"""Round 6: the fast mode's window forms with one v_pk_fma_f32 per tap (VND_WIN_ADDS=0) against the reference's class-path association
(VND_WIN_ADDS=1: v_pk_add_f32 inside a run of equal |w|, the gain ratio once per segment - vnd_win.hpp) on the bench's pools, interleaved
A B A B in ONE process (bench.py's own timing loop, board power and shader clock beside every leg, every stream of the fast output against
the exact kernel).  usage: adds_ab.py [cfg2 cfg3 cfg3k1 cfg5 cfg4 m2s ...] [repeats=N]"""
import os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
os.environ['VND_TUNING'] = '1'
import torch
import bench
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays

names = [a for a in sys.argv[1:] if '=' not in a] or ['cfg2', 'cfg3', 'cfg5', 'cfg4']
repeats = next((int(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('repeats=')), 2)
ctx = _native.default_context()
power = bench.PowerSampler(torch, 0)
CONFIGS = {
    'cfg2': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (1024, 480000, 2), 1),
    'cfg3': (dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1), (24, 2880000, 2), 1),
    'cfg3k1': (dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=1.0, seed=1), (24, 2880000, 2), 1),
    'cfg5': (dict(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1), (16, 960000, 8), 1),
    'cfg4': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (1024, 48000, 2), 2),
    'shard8': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (128, 48000, 2), 13),
}
for name in names:
    kw, shape, buffers = CONFIGS[name]
    a = function_path_arrays(vnd.generate_velvet_noise(**kw))
    taps = (a.tap_offsets, a.tap_index, a.tap_weight)
    t = _native.TapTable.create(ctx, *taps)
    for r in range(repeats):
        for adds in ('0', '1'):
            os.environ['VND_WIN_ADDS'] = adds
            torch.manual_seed(7000 + r)
            rec = bench.device_rate(torch, t, shape, vnd.MODE_FAST, buffers=buffers, taps=taps, exact_pool=(r == 0), power=power)
            b = rec.get('board') or {}
            print(f"{name:6s} [{r}] adds={adds}  {rec['kernel_ms']:.4f} ms  {rec['frac_of_8TBs']:.4f} of 8 TB/s  {b.get('power_W')} W {b.get('sclk_MHz')} MHz  "
                  f"parity {rec['parity_vs_oracle_of_peak']:.2e} worst-of-pool {rec.get('parity_max_over_pool', float('nan')):.2e}  "
                  f"{rec['launch'][rec['launch'].find('frames_per_lane'):][:40]} ...{rec['launch'][-40:]}", flush=True)
    t.close()
    torch.cuda.empty_cache()
power.close()
