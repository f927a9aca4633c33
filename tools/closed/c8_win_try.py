#!/usr/bin/env python3
"""cfg5 (96 kHz, 8 channels, 64 taps; bench shape 16 x 10 s) through the WINDOW form on channel pairs (VW_C = 8)
against the pair-read per-table kernel: parity vs the exact kernel and sustained rate, fast and exact.
usage: c8_win_try.py [seconds per variant]"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
os.environ.setdefault('VND_TUNING', '1')
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
ctx = _native.default_context()
fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1)
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
pool, n, C = 16, 960000, 8
x = torch.empty((pool, n, C), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
WIN = {0: 1 << 5, 16: 2 << 5, 32: 3 << 5, 64: 4 << 5}


def run(variant, mode):
    ctx.set_variant(variant)
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, C, mode=mode, stream=st)
    torch.cuda.synchronize()
    return y.clone()


def rate(variant, mode, label):
    ctx.set_variant(variant)
    desc = table.describe(pool, n, C, mode)
    t0 = time.perf_counter(); best = []
    while time.perf_counter() - t0 < seconds:
        best.append(table.time_device(x.data_ptr(), y.data_ptr(), pool, n, C, mode=mode, n_buffers=1, stride_elems=0, iters=40, stream=st))
    tail = best[len(best) // 2:]
    print(f'{label:30s} {np.mean(tail):.4f} ms {8e-6 * pool * n * C / np.mean(tail):6.0f} GB/s (min {min(best):.4f})  {desc}', flush=True)


def env_set(**env):
    for k in ('VND_SPEC_NT', 'VND_SPEC_LA', 'VND_WIN_G'):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)


ye = run(1 << 25, 0)                       # the generic ordered kernel: oracle-identical (tests)
peak = float(ye.abs().max())
configs = [(0, {}), (32, {}), (32, dict(VND_SPEC_NT=128)), (32, dict(VND_SPEC_NT=256)), (16, {}), (16, dict(VND_SPEC_NT=192)), (16, dict(VND_SPEC_NT=128))]
ok = []
for M, env in configs:
    env_set(**env)
    try:
        yf = run(WIN[M], 2)
        e_fast = float((yf - ye).abs().max()) / peak
        yx = run(WIN[M], 0)
        same = bool(torch.equal(yx, ye))
        print(f'M={M} {env}: fast vs exact {e_fast:.2e} of peak; exact form bit-identical to the generic exact kernel: {same}   {table.describe(pool, n, C, 2)}', flush=True)
        ok.append((M, env))
    except Exception as e:
        print(f'M={M} {env}: {e!r}', flush=True)
for rep in range(2):
    for M, env in ok:
        env_set(**env)
        tag = ' '.join(f'{k[4:].lower()}={v}' for k, v in env.items())
        rate(WIN[M], 2, f'fast  M={M} {tag}')
        rate(WIN[M], 0, f'exact M={M} {tag}')
    env_set()
    rate(1 << 25, 2, 'fast  generic')
    rate(1 << 25, 0, 'exact generic')

# diagnosis (wrong results on purpose): where the time of the wide window form goes.  VND_WIN_DEBUG bit 0: no stores;
# bit 1: every load from the same 4 KB.  A fresh table per setting (a table keeps its built kernels).
if len(sys.argv) > 2 and sys.argv[2] == 'diagnose':
    for dbg in (0, 1, 2, 3):
        os.environ['VND_WIN_DEBUG'] = str(dbg)
        table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
        for M, env in ((32, dict(VND_SPEC_NT=128)), (16, {})):
            env_set(**env)
            rate(WIN[M], 2, f'debug={dbg} fast M={M}')
    os.environ.pop('VND_WIN_DEBUG')
