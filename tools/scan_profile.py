#!/usr/bin/env python3
"""grid_scan host to host (400 candidates x 5.7 s stereo): wall time and where the host side goes (cProfile)."""
import contextlib, cProfile, io, pathlib, pstats, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import vndecorrelate_amd.decorrelation as vnd
import vndecorrelate_amd.optimization as opt
F, fs = 400, 44100
n = int(fs * 5.7)
sig = np.random.default_rng(0).uniform(-1, 1, (n, 2)).astype(np.float32)
KW = dict(angle_limit=float(np.pi / 4), lambda_mean=5.0, lambda_skew=2.0, lambda_correlation=15.0, lambda_penalty=1e3)
def make():
    return [vnd.VelvetNoise(sample_rate_hz=fs, duration_seconds=0.03, num_impulses=30, log_distribution_strength=k,
                            normalizer=None, filtered_channels=(0,), mode='LR', seed=1) for k in np.linspace(0, 1, F)]
with contextlib.redirect_stdout(io.StringIO()):
    opt.grid_scan(sig, make(), **KW)
    for label, fresh in (('candidates built before the clock starts', False), ('building the 400 candidates included', True)):
        best = 1e9
        for _ in range(7):
            cands = None if fresh else make()
            t = time.perf_counter()
            opt.grid_scan(sig, make() if fresh else cands, **KW)
            best = min(best, time.perf_counter() - t)
        print(f'grid_scan, {label}: {best * 1e3:.2f} ms', file=sys.stderr)
    pr = cProfile.Profile()
    cands = make()
    pr.enable(); opt.grid_scan(sig, cands, **KW); pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(18)
print(out.getvalue()[:3500], file=sys.stderr)
