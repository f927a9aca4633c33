#!/usr/bin/env python3
"""One of the secondary BASELINE configs (cfg3: 128 taps stereo 60 s; cfg5: 96 kHz, 8 channels, 64 taps) as a
plain launch loop for rocprofv3 (kernel trace or one --pmc pass per run):
    python3 tools/secondary_profile.py cfg3|cfg5 [launches]"""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays

CONFIGS = {
    'cfg2': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (128, 480000, 2)),
    'cfg3': (dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1), (24, 2880000, 2)),
    'cfg5': (dict(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1), (16, 960000, 8)),
}
name = sys.argv[1]
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
kw, (pool, n, c) = CONFIGS[name]
ctx = _native.default_context()
arr = function_path_arrays(vnd.generate_velvet_noise(**kw))
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
x = torch.empty((pool, n, c), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
print(table.describe(pool, n, c, vnd.MODE_FAST), flush=True)
for _ in range(launches):
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, c, vnd.MODE_FAST, st)
torch.cuda.synchronize()
print('taps per channel', (len(arr.tap_index)) // c, 'algorithmic bytes per launch', 8 * pool * n * c)
