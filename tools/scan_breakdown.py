#!/usr/bin/env python3
"""Where the candidate scan's host-to-host time goes (400 candidates x 5.7 s stereo)."""
import contextlib, io, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import vndecorrelate_amd.decorrelation as vnd
import vndecorrelate_amd.optimization as opt
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import concat_tap_arrays
F, fs = 400, 44100
n = int(fs * 5.7)
sig = np.random.default_rng(0).uniform(-1, 1, (n, 2)).astype(np.float32)
cands = [vnd.VelvetNoise(sample_rate_hz=fs, duration_seconds=0.03, num_impulses=30, log_distribution_strength=k,
                         normalizer=None, filtered_channels=(0,), mode='LR', seed=1) for k in np.linspace(0, 1, F)]
KW = dict(angle_limit=float(np.pi / 4), lambda_mean=5.0, lambda_skew=2.0, lambda_correlation=15.0, lambda_penalty=1e3)
with contextlib.redirect_stdout(io.StringIO()):
    opt.grid_scan(sig, cands, **KW)
for mode in (0, 2):
    T = {}
    for rep in range(5):
        t = time.perf_counter(); arrays = concat_tap_arrays([d._tap_arrays() for d in cands]); T.setdefault('concat tables', []).append(time.perf_counter() - t)
        t = time.perf_counter(); table = _native.TapTable.create(_native.default_context(), arrays.tap_offsets, arrays.tap_index, arrays.tap_weight, **arrays.kwargs()); T.setdefault('TapTable.create', []).append(time.perf_counter() - t)
        t = time.perf_counter(); m = table.scan_host(sig, mode); T.setdefault('scan_host', []).append(time.perf_counter() - t)
        t = time.perf_counter(); table.close(); T.setdefault('close', []).append(time.perf_counter() - t)
        t = time.perf_counter(); s = [opt.score_from_moments(r, **KW) for r in m]; T.setdefault('scores', []).append(time.perf_counter() - t)
    print('mode', mode, {k: round(min(v) * 1e3, 3) for k, v in T.items()}, 'ms (min of 5)')
