#!/usr/bin/env python3
"""Diagnostic: phase durations per workgroup from the -DVND_STAMPS build (s_memtime ticks)."""
import ctypes, os, pathlib, sys
root = pathlib.Path(__file__).resolve().parents[1]
os.environ['VND_AMD_LIBRARY'] = str(root / 'tools' / 'ablate' / 'libvnd_stamps.so')
sys.path.insert(0, str(root))
import numpy as np, torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
ctx.set_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
pool, n = 128, 480000
x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
for _ in range(300):
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, 2, st)
torch.cuda.synchronize()
lib = _native.load_library()
nblk = 30080
buf = np.zeros(nblk * 8, np.uint64)
rc = lib.vnd_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), nblk * 8)
assert rc == 0, rc
s = buf.reshape(nblk, 8).astype(np.int64)
names = ['prologue', 'staging (loads+LDS writes)', 'barrier after staging', 'tap loops', 'barrier before merge',
         'merge + store issue', 'stores drain']
print(table.describe(pool, n, 2, 2))
d = np.diff(s, axis=1)
ok = (d >= 0).all(axis=1) & (d.sum(axis=1) < 10_000_000)
d = d[ok]
print(f'{ok.sum()} workgroups; s_memtime ticks are 100 MHz constant-rate? -> reported as ticks')
for i, nm in enumerate(names):
    print(f'{nm:30s} median {np.median(d[:, i]):9.0f}  mean {d[:, i].mean():9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}')
tot = d.sum(axis=1)
print(f'{"lifetime":30s} median {np.median(tot):9.0f}  mean {tot.mean():9.0f}')
span = s[ok][:, 7].max() - s[ok][:, 0].min()
print(f'kernel span {span} ticks; sum of lifetimes / span = {tot.sum() / span:.1f} workgroups in flight')
