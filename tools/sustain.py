#!/usr/bin/env python3
"""Does the kernel hold its rate?  Times consecutive blocks of launches."""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
variant = int(sys.argv[2]) if len(sys.argv) > 2 else -1
ctx = _native.default_context()
ctx.set_variant(variant)
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
pool, n = 128, 480000
x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
print(table.describe(pool, n, 2, mode))
for blk in range(12):
    ms = table.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, n_buffers=1, stride_elems=0,
                           iters=25, stream=stream)
    print(f'launches {blk*25:4d}-{blk*25+24:4d}: {ms:.4f} ms/launch  {983.04/ms:.0f} GB/s')
# python-loop launches, as bench.py does
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(100):
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode, stream)
t_issue = time.perf_counter() - t0
e1.record(); torch.cuda.synchronize()
print(f'python loop: {e0.elapsed_time(e1)/100:.4f} ms/launch (host issue {t_issue/100*1e3:.4f} ms/launch)')
