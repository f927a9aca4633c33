#!/usr/bin/env python3
"""In-kernel timeline of a window-form launch (split stereo form): every workgroup's wave 0 stamps the 100 MHz wall clock
at its phase boundaries (diagnosis build: VND_TUNING=1 VND_WIN_STAMPS=<workgroups>; include/vnd_amd_internal.h).
Prints, relative to the first workgroup's start: when workgroups start, when their rings are filled, and per tile the
end of the tap phase, of the output exchange and of the refill - mean, min, max over the workgroups - and the same
split by the order in which a CU received its workgroups.
usage: win_stamps.py [streams] [frames] [env KEY=VALUE ...]"""
import os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
os.environ['VND_TUNING'] = '1'
os.environ.setdefault('VND_WIN_STAMPS', '1024')
args = [a for a in sys.argv[1:] if '=' not in a]
for kv in sys.argv[1:]:
    if '=' in kv:
        k, v = kv.split('=', 1)
        os.environ[k] = v
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays

mine = int(args[0]) if args else 128
n = int(args[1]) if len(args) > 1 else 48000
CH = 2
mode = vnd.MODE_EXACT if os.environ.get('STAMPS_MODE') == 'exact' else vnd.MODE_FAST
ctx = _native.default_context()
which = os.environ.get('STAMPS_TABLE', 'cfg2')
if which == 'cfg5':
    CH = 8
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1)
elif which == 'cfg3':
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1)
else:
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
buffers = max(2, int(np.ceil(600e6 / (mine * n * CH * 4 * 2))))
xs = [torch.empty((mine, n, CH), dtype=torch.float32, device='cuda').uniform_(-1, 1) for _ in range(buffers)]
ys = [torch.empty_like(xs[0]) for _ in range(buffers)]
table.prepare(mine, n, CH, mode)
print(table.describe(mine, n, CH, mode))
side = torch.cuda.Stream()
st = side.cuda_stream
def step(i):
    table.convolve_device(xs[i % buffers].data_ptr(), ys[i % buffers].data_ptr(), mine, n, CH, mode, st)
# (a hipGraph replay: under VND_TUNING the host side of a launch re-reads its variables and may be slower than a small pass)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    for i in range(50):
        step(i)
    side.synchronize()
    with torch.cuda.graph(g, stream=side):
        for i in range(200):
            step(i)
    g.replay()
    side.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record()
torch.cuda.synchronize()
per_pass = e0.elapsed_time(e1) / 200 * 1e3
print(f'{per_pass:.2f} us per pass (stamped build)')
s = table.read_stamps(mine, n, CH, mode)
if s.size == 0:
    raise SystemExit('no stamps in this kernel')
live = s[:, 12] != 0
block = np.arange(len(s))[live]
s = s[live]
if not np.any(s[:, 0]):          # VND_WIN_STAMP_PHASES=0: entry and exit only (nothing held in registers: the build of the product)
    entry, exit_ = s[:, 12].astype(np.int64), s[:, 13].astype(np.int64)
    t0 = entry.min()
    print(f'{len(s)} workgroups; entries spread over {(entry.max() - t0) * 0.01:.2f} us; exits: first {(exit_.min() - t0) * 0.01:.2f}, mean {(exit_.mean() - t0) * 0.01:.2f}, '
          f'last {(exit_.max() - t0) * 0.01:.2f} us after the first entry  =>  in-kernel span {(exit_.max() - t0) * 0.01:.2f} us, boundary {per_pass - (exit_.max() - t0) * 0.01:.2f} us')
    second = (block >> 3) >= 32
    if second.any() and (~second).any():
        print(f'   first workgroup of a CU (blockIdx >> 3 < 32): exit mean {(exit_[~second].mean() - t0) * 0.01:.2f} max {(exit_[~second].max() - t0) * 0.01:.2f};  later ones: exit mean {(exit_[second].mean() - t0) * 0.01:.2f} max {(exit_[second].max() - t0) * 0.01:.2f}')
    raise SystemExit(0)
hw = s[:, 15]
t = s[:, :15].astype(np.int64)
t[:, 2:12][t[:, 2:12] == 0] = -1
entry = t[:, 12].copy()
t[:, 12:14] = -1
t0 = entry.min()
print(f'kernel entry of the first workgroup -> its span starts: {(t[:, 0] - entry).mean() * 0.01:.2f} us (mean over workgroups); last workgroup entered {(entry.max() - t0) * 0.01:.2f} us after the first')
rel = np.where(t > 0, (t - t0) * 0.01, np.nan)          # us
names = ['start', 'ring filled'] + [f'tile {k // 4}: {w}' for k in range(12) for w in [('taps done', 'window dead (quads, octets: stores issued)', 'outputs exchanged', 'refill published')[k % 4]]] + ['stores acknowledged']
print(f'{len(s)} workgroups stamped; last stamp of all {np.nanmax(rel):.2f} us after the first start')
print(f'{"phase":28s} {"mean":>8s} {"min":>8s} {"max":>8s}   (us after the first workgroup started)')
for k, name in enumerate(names):
    col = rel[:, k]
    if np.all(np.isnan(col)):
        continue
    print(f'{name:28s} {np.nanmean(col):8.2f} {np.nanmin(col):8.2f} {np.nanmax(col):8.2f}')
# phase lengths per workgroup
d = np.diff(rel[:, :14], axis=1)
print(f'last refill published -> stores acknowledged: mean {np.nanmean(rel[:, 14] - np.nanmax(rel[:, :14], axis=1)):.2f} us')
print('phase lengths (us, mean over workgroups):')
for k in range(d.shape[1]):
    if not np.all(np.isnan(d[:, k])):
        print(f'  {names[k]:24s} -> {names[k + 1]:28s} {np.nanmean(d[:, k]):7.2f}  (min {np.nanmin(d[:, k]):6.2f} max {np.nanmax(d[:, k]):6.2f})')
# by CU: HW_ID bits (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (se 2 bits on some), ...; XCC_ID low bits of the high word
cu = ((hw >> 8) & 0xf).astype(np.int64); sh = ((hw >> 12) & 1).astype(np.int64); se = ((hw >> 13) & 7).astype(np.int64); xcc = ((hw >> 32) & 0xf).astype(np.int64)
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
order = np.zeros(len(s), np.int64)
for kk in np.unique(key):
    idx = np.where(key == kk)[0]
    idx = idx[np.argsort(t[idx, 0])]
    order[idx] = np.arange(len(idx))
print(f'{len(np.unique(key))} distinct CUs seen; workgroups per CU: {np.bincount(np.bincount(np.unique(key, return_inverse=True)[1]))[1:]} (count of CUs with 1, 2, ... workgroups)')
for o in range(int(order.max()) + 1):
    sel = order == o
    line = '  '.join(f'{np.nanmean(rel[sel, k]):6.2f}' for k in range(15) if not np.all(np.isnan(rel[sel, k])))
    print(f'  workgroup #{o} of its CU (n={int(sel.sum())}): {line}')
second = (block >> 3) >= 32
print(f'blocks with (blockIdx >> 3) >= 32: {int(second.sum())}, of them not the first workgroup on their CU: {int((order[second] > 0).sum())}; '
      f'blocks below 32 that are not first: {int((order[~second] > 0).sum())}')
