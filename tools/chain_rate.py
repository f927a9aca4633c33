#!/usr/bin/env python3
"""Stage-by-stage chain vs device-resident chain: velvet noise -> Haas, host to host."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import vndecorrelate_amd.decorrelation as vnd

fs, seconds = 48000, 10
x = np.random.default_rng(0).uniform(-1, 1, (fs * seconds, 2)).astype(np.float32)


def build(**kw):
    return (vnd.SignalChain(sample_rate_hz=fs, **kw).velvet_noise(seed=1)
            .haas_effect(delay_time_seconds=0.02, delayed_channel=1, mode='LR'))


for label, chain in (('stage by stage (host arrays between stages)', build()),
                     ('device resident', build(device_resident=True))):
    for mode, name in ((vnd.MODE_EXACT, 'exact'), (vnd.MODE_FAST, 'fast')):
        vnd.set_default_mode(mode)
        chain(x)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            y = chain(x)
        dt = (time.perf_counter() - t0) / reps
        print(f'{label:46s} {name:5s}: {dt*1e3:8.2f} ms per 10 s stereo signal  ({x.size/dt/1e6:8.1f} Msamples/s host to host)')
