#!/usr/bin/env python3
"""Round 6: a mono input fanned out through a function-path stereo table, VND_MODE_EXACT - the split form with the input staged into
both plane sets (VND_WIN_EXACT_MERGED=0) against the plain form with ONE read stream and shared products for both channels
(VND_WIN_EXACT_MERGED=1, win_taps_function_exact_merged), interleaved; 128 x 10 s mono signals, 12 algorithmic bytes per frame; the
first and last stream of every leg bit for bit against the C oracle.  usage: m2s_exact_ab.py [repeats=N]"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
os.environ['VND_TUNING'] = '1'
import numpy as np
import torch
import bench
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
from oracle import c_oracle

repeats = next((int(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('repeats=')), 2)
ctx = _native.default_context()
power = bench.PowerSampler(torch, 0)
n, pool = 480000, 128
for name, kw in (('cfg2 table', dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)),
                 ('128-tap table', dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1))):
    arr = function_path_arrays(vnd.generate_velvet_noise(**kw))
    taps = (arr.tap_offsets, arr.tap_index, arr.tap_weight)
    table = _native.TapTable.create(ctx, *taps)
    torch.manual_seed(11)
    x = torch.empty((pool, n, 1), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    for r in range(repeats):
        for merged in ('0', '1'):
            os.environ['VND_WIN_EXACT_MERGED'] = merged
            run = lambda k=0: table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 1, vnd.MODE_EXACT, st)
            run(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.12:
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(200):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 200
            board = bench.board_under(torch, power, run) or {}
            for b in (0, pool - 1):
                xs2 = np.ascontiguousarray(np.repeat(x[b].cpu().numpy(), 2, axis=1))
                assert np.array_equal(y[b].cpu().numpy(), c_oracle.convolve(xs2, *taps, threads=8)), f'stream {b} differs from the oracle'
            d = table.describe(pool, n, 1, vnd.MODE_EXACT)
            print(f"{name:14s} [{r}] merged={merged}  {ms:.4f} ms  {12e-9 * pool * n / ms / 8.0:.4f} of 8 TB/s (12 B/frame)  {board.get('power_W')} W {board.get('sclk_MHz')} MHz  "
                  f"bit-identical  {d[d.find('frames_per_lane'):][:34]} ... {d[d.find('threads='):]}", flush=True)
    table.close()
power.close()
