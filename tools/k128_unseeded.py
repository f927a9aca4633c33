#!/usr/bin/env python3
"""Round 6: how far VND_MODE_FAST lands from the reference (= VND_MODE_EXACT, bit for bit) on UNSEEDED pools - one measurement, not a
loop until it passes.  For each table: `pools` fresh pools of the bench's shape drawn from torch's non-deterministic seed, fast and
exact kernels on each, the distance of every stream's worst sample as a fraction of the POOL's output peak (the bench's and the tests'
figure) and of the stream's OWN peak.  Output: profiles-style JSON (gpurun_out/r06_k128_unseeded.json) with every pool's maximum, the
quantiles and the count above 1e-6.  usage: k128_unseeded.py [pools=64]"""
import json, os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays

pools = next((int(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('pools=')), 64)
ctx = _native.default_context()
stream = torch.cuda.current_stream().cuda_stream
TABLES = {
    'cfg3 (128 taps, kappa 0)': (dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1), (24, 2880000, 2), pools),
    'cfg3 kappa 1 (123 taps)': (dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=1.0, seed=1), (24, 2880000, 2), pools // 2),
    'cfg5 (8 channels, 64 taps)': (dict(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1), (16, 960000, 8), pools // 2),
    'cfg2 (30 taps), pools of 144 signals': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (144, 480000, 2), pools // 2),
}
out = {'what': __doc__.split('usage')[0].strip(), 'device': torch.cuda.get_device_name(0), 'tables': {}}
for name, (kw, shape, count) in TABLES.items():
    a = function_path_arrays(vnd.generate_velvet_noise(**kw))
    t = _native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)
    batch, n, c = shape
    x = torch.empty(shape, dtype=torch.float32, device='cuda')
    yf, ye = torch.empty_like(x), torch.empty_like(x)
    of_pool, of_own, seeds = [], [], []
    launch = None
    for k in range(count):
        seeds.append(torch.seed())                       # non-deterministic: a fresh draw every time, recorded
        x.uniform_(-1.0, 1.0)
        t.convolve_device(x.data_ptr(), ye.data_ptr(), batch, n, c, vnd.MODE_EXACT, stream)
        t.convolve_device(x.data_ptr(), yf.data_ptr(), batch, n, c, vnd.MODE_FAST, stream)
        torch.cuda.synchronize()
        launch = launch or t.describe(batch, n, c, vnd.MODE_FAST)
        err = (yf - ye).abs().amax(dim=(1, 2))
        of_pool.append(float((err / ye.abs().max()).max()))
        of_own.append(float((err / ye.abs().amax(dim=(1, 2))).max()))
    q = lambda v: {'min': min(v), 'median': float(np.median(v)), 'p90': float(np.quantile(v, 0.9)), 'max': max(v), 'above_1e-6': int(sum(e > 1e-6 for e in v)), 'of': len(v)}
    out['tables'][name] = {'pool_shape': list(shape), 'samples_per_pool': batch * n * c, 'launch': launch[:200],
                           'worst_stream_of_pool_peak': q(of_pool), 'worst_stream_of_its_own_peak': q(of_own),
                           'per_pool_of_pool_peak': [float(f'{e:.4g}') for e in of_pool], 'seeds': seeds}
    print(name, json.dumps(out['tables'][name]['worst_stream_of_pool_peak']), json.dumps(out['tables'][name]['worst_stream_of_its_own_peak']), flush=True)
    t.close()
    del x, yf, ye
    torch.cuda.empty_cache()
target = pathlib.Path(__file__).resolve().parents[1] / 'gpurun_out'
target.mkdir(exist_ok=True)
(target / 'r06_k128_unseeded.json').write_text(json.dumps(out, indent=1))
