#!/usr/bin/env python3
"""VND_MODE_EXACT, per-table kernel (default) against the generic ordered kernel on the other BASELINE shapes:
cfg3 (128 taps, 60 s stereo), cfg5 (96 kHz, 8 channels, 64 taps), cfg4 (1024 x 1 s stereo) and its N = 8 shard."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
CASES = {
    'cfg3': (dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1), (24, 2880000, 2)),
    'cfg5': (dict(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1), (16, 960000, 8)),
    'cfg4': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (1024, 48000, 2)),
    'cfg4 N=8 shard': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (128, 48000, 2)),
    'cfg2 x 32': (dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (32, 480000, 2)),
}
st = torch.cuda.current_stream().cuda_stream
for name, (kw, (pool, n, c)) in CASES.items():
    arr = function_path_arrays(vnd.generate_velvet_noise(**kw))
    table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
    x = torch.empty((pool, n, c), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty_like(x)
    res = {}
    for rnd in range(2):
        for label, variant in (('generic', 1 << 25), ('default', -1)):
            ctx.set_variant(variant)
            def run(): table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, c, 0, st)
            for _ in range(5): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(30): run()
            e1.record(); torch.cuda.synchronize()
            res[label] = min(res.get(label, 1e9), e0.elapsed_time(e1) / 30)
    ctx.set_variant(-1)
    print(f'{name:16s} generic {res["generic"]:.4f} ms   default {res["default"]:.4f} ms ({res["generic"] / res["default"]:.2f}x)   {table.describe(pool, n, c, 0)[:100]}', flush=True)
    table.close(); del x, y; torch.cuda.empty_cache()
