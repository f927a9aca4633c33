#!/usr/bin/env python3
"""The optimiser's candidate scan: F velvet-noise candidates on one signal.
Device scan (one upload, one fan-out convolution, moments kernel) against the reference's
procedure restated on the CPU (oracle: decorrelate + polar objective per candidate)."""
import contextlib, io, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import vndecorrelate_amd.decorrelation as vnd
import vndecorrelate_amd.optimization as opt
from oracle import vnd_oracle as O

F = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 5.7
fs = 44100
n = int(fs * seconds)
sig = np.random.default_rng(0).uniform(-1, 1, (n, 2)).astype(np.float32)
KW = dict(angle_limit=float(np.pi / 4), lambda_mean=5.0, lambda_skew=2.0, lambda_correlation=15.0, lambda_penalty=1e3)
kappas = np.linspace(0.0, 1.0, F)

t0 = time.perf_counter()
cands = [vnd.VelvetNoise(sample_rate_hz=fs, duration_seconds=0.03, num_impulses=30, log_distribution_strength=k,
                         normalizer=None, filtered_channels=(0,), mode='LR', seed=1) for k in kappas]
t_build = time.perf_counter() - t0
for mode, name in ((vnd.MODE_EXACT, 'exact'), (vnd.MODE_FAST, 'fast')):
    vnd.set_default_mode(mode)
    with contextlib.redirect_stdout(io.StringIO()):
        opt.grid_scan(sig, cands[:4], **KW)                      # warm up
        t0 = time.perf_counter()
        scores = opt.grid_scan(sig, cands, **KW)
        dt = time.perf_counter() - t0
    by_mode = dict(globals().get('by_mode', {}), **{name: scores})
    print(f'device scan, {name:5s}: F={F} n={n}: {dt*1e3:8.1f} ms wall ({dt/F*1e3:.3f} ms/candidate), '
          f'{F*n*2/dt/1e6:.0f} Msamples/s scored; building the {F} candidates on the host took {t_build*1e3:.0f} ms')
vnd.set_default_mode(vnd.MODE_EXACT)

# the reference's procedure on the CPU (NumPy restatement, 1 core): a few candidates, extrapolated
t0 = time.perf_counter()
k_cpu = 3
for k in kappas[:k_cpu]:
    out = O.decorrelate(sig.copy(), sample_rate_hz=fs, duration_seconds=0.03, num_impulses=30,
                        log_distribution_strength=float(k), filtered_channels=(0,), mode='LR', normalize=False, seed=1)
    cpu_score = O.symmetry_aware_objective(out, **KW)
dt_cpu = (time.perf_counter() - t0) / k_cpu
print(f'CPU restatement of the reference loop: {dt_cpu*1e3:.1f} ms/candidate -> {dt_cpu*F:.1f} s for F={F} '
      f'({F*n*2/(dt_cpu*F)/1e6:.1f} Msamples/s scored)')
print('|device - CPU| on that candidate: exact %.2e, fast %.2e' % tuple(abs(by_mode[m][k_cpu - 1] - cpu_score) for m in ('exact', 'fast')))
