#!/usr/bin/env python3
"""Board power and clocks while the convolve kernel runs back to back (diagnostic).

usage: power_probe.py [mode] [seconds]   (VND_AMD_LIBRARY picks a diagnostic build)
"""
import glob, pathlib, subprocess, sys, threading, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays

mode = sys.argv[1] if len(sys.argv) > 1 else '2'          # 0 | 1 | 2 = kernel arithmetic, 'copy' = plain device copy
copy_only = mode == 'copy'
mode = 2 if copy_only else int(mode)
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0


def read_sysfs():
    out = {}
    for path in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_*') + \
            glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/freq*_input') + \
            glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/temp*_input'):
        try:
            out[path.split('/device/')[0][-5:] + ':' + path.rsplit('/', 1)[1]] = int(open(path).read())
        except Exception:
            pass
    return out


samples, stop = [], False


def sampler():
    while not stop:
        t = time.perf_counter()
        try:
            smi = subprocess.run(['rocm-smi', '-P', '-c', '--csv'], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:
            smi = repr(e)
        samples.append((t, read_sysfs(), smi))
        time.sleep(0.4)


ctx = _native.default_context()
import os
CONFIGS = {     # VND_PROBE_CFG: the workload (default cfg2)
    'cfg2': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), 128, 480000, 2),
    'cfg3': (dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1), 24, 2880000, 2),
    'cfg5': (dict(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1), 16, 960000, 8),
}
kw, pool, n, chans = CONFIGS[os.environ.get('VND_PROBE_CFG', 'cfg2')]
fir = vnd.generate_velvet_noise(**kw)
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
x = torch.empty((pool, n, chans), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
print(table.describe(pool, n, chans, mode))
print('idle:', read_sysfs())
th = threading.Thread(target=sampler); th.start()
time.sleep(1.0)
t_start = time.perf_counter()
rates = []
while time.perf_counter() - t_start < seconds:
    if copy_only:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            y.copy_(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
    else:
        ms = table.time_device(x.data_ptr(), y.data_ptr(), pool, n, chans, mode=mode, n_buffers=1, stride_elems=0,
                               iters=200 if chans == 2 and n <= 480000 else 60, stream=stream)
    rates.append((time.perf_counter() - t_start, ms))
t_end = time.perf_counter()
time.sleep(1.0)
stop = True; th.join()
for t, ms in rates[::4]:
    print(f't={t:5.2f}s  {ms:.4f} ms/launch  {8e-6 * pool * n * chans / ms:.0f} GB/s')
for t, fs, smi in samples:
    tag = 'RUN ' if t_start <= t <= t_end else 'idle'
    print(tag, f'{t - t_start:6.2f}', fs)
    print('     ', ' | '.join(l for l in smi.strip().splitlines()[:3]))
