#!/usr/bin/env python3
"""cfg5 (96 kHz, 8 channels, 64 taps) as it is - interleaved frames, one launch - against the same arithmetic on
PLANAR channel pairs (four stereo launches, each with its pair's taps): what the 8-byte pieces of 32-byte frames cost."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1)
arr = function_path_arrays(fir)
full = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
pairs = []
for g in range(4):
    lo, mid, hi = arr.tap_offsets[2 * g], arr.tap_offsets[2 * g + 1], arr.tap_offsets[2 * g + 2]
    offs = np.array([0, mid - lo, hi - lo], np.int32)
    pairs.append(_native.TapTable.create(ctx, offs, arr.tap_index[lo:hi].copy(), arr.tap_weight[lo:hi].copy()))
pool, n = 16, 960000
st = torch.cuda.current_stream().cuda_stream
x8 = torch.empty((pool, n, 8), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y8 = torch.empty_like(x8)
x2 = [x8[:, :, 2 * g:2 * g + 2].contiguous() for g in range(4)]
y2 = [torch.empty_like(v) for v in x2]

def run8():
    full.convolve_device(x8.data_ptr(), y8.data_ptr(), pool, n, 8, 2, st)

def run2():
    for g in range(4):
        pairs[g].convolve_device(x2[g].data_ptr(), y2[g].data_ptr(), pool, n, 2, 2, st)

print(full.describe(pool, n, 8, 2)); print(pairs[0].describe(pool, n, 2, 2))
run8(); run2(); torch.cuda.synchronize()
err = max(float((y8[:, :, 2 * g:2 * g + 2] - y2[g]).abs().max()) for g in range(4))
print('max difference between the two forms', err)
def planar(variant):
    def fn():
        ctx.set_variant(variant); run2(); ctx.set_variant(-1)
    return fn

# round 3: the planar pairs take the WINDOW form by default; 1 << 5 turns it off (the pair-read kernel), 2 << 5 = 16 frames per lane
forms = (('interleaved, one launch', run8), ('planar pairs, window (default)', run2), ('planar pairs, pair-read', planar(1 << 5)),
         ('planar pairs, window 16', planar(2 << 5)))
ctx.set_variant(2 << 5); print(pairs[0].describe(pool, n, 2, 2)); ctx.set_variant(-1)
for rnd in range(3):
    for name, fn in forms:
        for _ in range(10): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(40): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        print(f'{name:32s} {ms:.4f} ms  {8 * pool * n * 8 / ms / 1e6:.0f} GB/s', flush=True)
