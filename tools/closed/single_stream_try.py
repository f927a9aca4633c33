#!/usr/bin/env python3
"""One long stream (cfg3's 60 s, 128 taps; and 10 s / 60 s with 30 taps): generic kernels against per-table kernels forced
with several minimum span lengths - where does the persistent kernel start to pay for little work?"""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
FORCE = 1 << 23
st = torch.cuda.current_stream().cuda_stream
for name, kw, (pool, n) in (('cfg3 one stream', dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1), (1, 2880000)),
                            ('60 s, 30 taps', dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (1, 2880000)),
                            ('10 s, 30 taps', dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (1, 480000)),
                            ('8 x 10 s, 30 taps', dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1), (8, 480000))):
    arr = function_path_arrays(vnd.generate_velvet_noise(**kw))
    table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
    xs = [torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1) for _ in range(12)]
    ys = [torch.empty_like(xs[0]) for _ in range(12)]
    for mode, mname in ((2, 'fast'), (0, 'exact')):
        # 'auto' takes the per-table kernel for launches this small only when its code object already exists (it never
        # triggers a build for them): build it once first, as an earlier large launch or an earlier run would have
        ctx.set_variant(FORCE)
        table.convolve_device(xs[0].data_ptr(), ys[0].data_ptr(), pool, n, 2, mode, st); torch.cuda.synchronize()
        out = []
        for label, variant in (('generic', 1 << 25), ('auto', -1), ('spec span>=8', FORCE | (8 % 8 << 20)), ('spec span>=4', FORCE | (4 << 20)), ('spec span>=2', FORCE | (2 << 20)), ('spec span>=1', FORCE | (1 << 20))):
            ctx.set_variant(variant)
            def run(i): table.convolve_device(xs[i % 12].data_ptr(), ys[i % 12].data_ptr(), pool, n, 2, mode, st)
            for i in range(24): run(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(120): run(i)
            e1.record(); torch.cuda.synchronize()
            out.append(f'{label} {e0.elapsed_time(e1) / 120 * 1e3:.1f}')
        print(f'{name:18s} {mname:5s} us per launch: ' + ' | '.join(out), flush=True)
    ctx.set_variant(-1)
    table.close(); del xs, ys; torch.cuda.empty_cache()
