#!/usr/bin/env python3
"""VND_MODE_EXACT through the WINDOW form of the per-table kernel against the pair-read per-table kernel (shifted plane
copies) and the generic ordered kernel: bit-equality on the whole pool, one stream against the C oracle, sustained rate.
usage: win_exact_try.py [seconds per variant] [cfg2|cfg3]"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
os.environ.setdefault('VND_TUNING', '1')      # geometry variables are read live
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
from oracle import c_oracle
from oracle import vnd_oracle as O

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
which = sys.argv[2] if len(sys.argv) > 2 else 'cfg2'
ctx = _native.default_context()
kw = dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
pool, n = 128, 480000
if which == 'cfg3':
    kw.update(num_impulses=128, log_distribution_strength=0.0)
    pool, n = 24, 2880000
fir = vnd.generate_velvet_noise(**kw)
arr = function_path_arrays(fir)
tables = {'function path': _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)}
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1, num_impulses=kw['num_impulses'],
                     log_distribution_strength=kw.get('log_distribution_strength', 1.0))
tables['class path'] = vn._device_table()
x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
GENERIC, WIN_OFF = 1 << 25, 1 << 5
WIN = {16: 2 << 5, 32: 3 << 5}


def run(table, variant):
    ctx.set_variant(variant)
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=0, stream=stream)
    torch.cuda.synchronize()
    return y.clone()


def rate(table, variant, label):
    ctx.set_variant(variant)
    desc = table.describe(pool, n, 2, 0)
    t0 = time.perf_counter(); best = []
    while time.perf_counter() - t0 < seconds:
        best.append(table.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=0, n_buffers=1, stride_elems=0,
                                      iters=100 if n * pool < 1e8 else 30, stream=stream))
    tail = best[len(best) // 2:]
    print(f'{label:40s} {np.mean(tail):.4f} ms/launch {8e-6 * pool * n * 2 / np.mean(tail):6.0f} GB/s  (min {min(best):.4f})  {desc[:150]}', flush=True)


for name, table in tables.items():
    y_gen = run(table, GENERIC)
    y_pair = run(table, WIN_OFF)
    print(f'== {name}: pair-read == generic ordered: {torch.equal(y_gen, y_pair)}', flush=True)
    if name == 'function path':
        want = c_oracle.convolve(x[pool - 1].cpu().numpy(), *O.fir_to_taps(fir))
        print('   generic == C oracle (last stream):', np.array_equal(y_gen[pool - 1].cpu().numpy(), want), flush=True)
    for M, env in ((32, {}), (32, dict(VND_SPEC_NT=128)), (16, {})):
        for k in ('VND_SPEC_NT',):
            os.environ.pop(k, None)
        os.environ.update({k: str(v) for k, v in env.items()})
        yw = run(table, WIN[M])
        desc = table.describe(pool, n, 2, 0)
        same = torch.equal(yw, y_gen)
        print(f'   window M={M} {env}: == generic: {same}   {desc[:150]}', flush=True)
        if not same:
            d = (yw != y_gen)
            print('      first differing (stream, frame, ch):', d.nonzero()[:8].tolist(), 'count', int(d.sum()), flush=True)
    for rep in range(2):
        for k in ('VND_SPEC_NT',):
            os.environ.pop(k, None)
        rate(table, GENERIC, f'{name}: generic ordered')
        rate(table, WIN_OFF, f'{name}: pair-read per-table')
        rate(table, WIN[32], f'{name}: window M=32')
        os.environ['VND_SPEC_NT'] = '128'
        rate(table, WIN[32], f'{name}: window M=32 nt=128')
        os.environ.pop('VND_SPEC_NT')
        rate(table, WIN[16], f'{name}: window M=16')
ctx.set_variant(-1)
