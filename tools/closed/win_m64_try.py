#!/usr/bin/env python3
"""64-frame runs (one wave per SIMD, half the LDS reads per FMA of 32-frame runs on a dense table) with deeper read
pipelines - a lane alone on its SIMD has 512 registers: cfg3 and cfg2, fast and exact, against the 32-frame default."""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
os.environ['VND_TUNING'] = '1'
import numpy as np, torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
WIN = {32: 3 << 5, 64: 4 << 5}
for which in ('cfg3', 'cfg2'):
    kw = dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1); pool, n = 128, 480000
    if which == 'cfg3':
        kw.update(num_impulses=128, log_distribution_strength=0.0); pool, n = 24, 2880000
    arr = function_path_arrays(vnd.generate_velvet_noise(**kw))
    table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1); y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    ref = {}
    for rep in range(2):
        for M, nt, la in ((32, 256, 4), (64, 128, 4), (64, 128, 8), (64, 128, 12), (64, 128, 16), (64, 128, 24), (64, 256, 12), (64, 64, 12)):
            os.environ['VND_SPEC_LA'] = str(la); os.environ['VND_SPEC_NT'] = str(nt)
            for mode in (2, 0):
                ctx.set_variant(WIN[M])
                try:
                    desc = table.describe(pool, n, 2, mode)
                    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, stream=st); torch.cuda.synchronize()
                except Exception as e:
                    print(f'{which} M={M} nt={nt} LA={la} mode={mode}: {e!r}', flush=True); continue
                if mode not in ref: ref[mode] = y[:2].clone()
                same = bool(torch.equal(y[:2], ref[mode]))
                best = []
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.5:
                    best.append(table.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, n_buffers=1, stride_elems=0, iters=100 if which == 'cfg2' else 30, stream=st))
                tail = best[len(best) // 2:]
                print(f'{which} mode={mode} M={M} nt={nt} LA={la:2d}  {np.mean(tail):.4f} ms  {8e-6 * pool * n * 2 / np.mean(tail):6.0f} GB/s  same as the first: {same}  {desc[:100]}', flush=True)
    del x, y; table.close(); torch.cuda.empty_cache()
