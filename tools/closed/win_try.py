#!/usr/bin/env python3
"""The WINDOW form of the per-table kernel (vnd_win.hpp) against the pair-read per-table kernel and the
generic one: parity (vs the exact kernel on a few streams, whole pool vs the pair-read kernel) and sustained rate.
usage: win_try.py [seconds per variant] [cfg2|cfg3|cfg4] [quick]"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
os.environ.setdefault('VND_TUNING', '1')      # geometry variables are read live
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
which = sys.argv[2] if len(sys.argv) > 2 else 'cfg2'
quick = len(sys.argv) > 3 and sys.argv[3] == 'quick'
ctx = _native.default_context()
if which == 'cfg3':
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000,
                                    log_distribution_strength=0.0, seed=1)
    pool, n = 24, 2880000
elif which == 'cfg4':
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
    pool, n = 1024, 48000
else:
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
    pool, n = 128, 480000
arr = function_path_arrays(fir)
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
GENERIC = 1 << 25
WIN = {0: 1 << 5, 16: 2 << 5, 32: 3 << 5, 64: 4 << 5}


def run(variant, mode=2):
    ctx.set_variant(variant)
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, stream=stream)
    torch.cuda.synchronize()
    return y.clone()


def rate(variant, label):
    ctx.set_variant(variant)
    desc = table.describe(pool, n, 2, 2)
    t0 = time.perf_counter(); best = []
    while time.perf_counter() - t0 < seconds:
        best.append(table.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=2, n_buffers=1, stride_elems=0,
                                      iters=100 if n * pool < 1e8 else 30, stream=stream))
    tail = best[len(best) // 2:]
    print(f'{label:34s} {np.mean(tail):.4f} ms/launch {8e-6 * pool * n * 2 / np.mean(tail):6.0f} GB/s  (min {min(best):.4f})  {desc}', flush=True)


ne = min(4, pool)
ye = torch.empty((ne, n, 2), dtype=torch.float32, device='cuda')
ctx.set_variant(-1)
table.convolve_device(x[:ne].contiguous().data_ptr(), ye.data_ptr(), ne, n, 2, mode=0, stream=stream)
torch.cuda.synchronize()
peak = float(ye.abs().max())
y_pair = run(WIN[0])
print('pair-read kernel:', table.describe(pool, n, 2, 2), f'err {float((y_pair[:ne] - ye).abs().max()) / peak:.2e} of peak', flush=True)


def env_set(**env):
    for k in ('VND_SPEC_NT', 'VND_SPEC_LA', 'VND_WIN_G'):
        os.environ.pop(k, None)
    for k, val in env.items():
        os.environ[k] = str(val)


def check(M, **env):
    env_set(**env)
    yw = run(WIN[M])
    desc = table.describe(pool, n, 2, 2)
    d = (yw[:ne] - ye).abs()
    err = float(d.max()) / peak
    dp = float((yw - y_pair).abs().max()) / peak
    worst = int(d.amax(dim=(0, 2)).argmax())
    tag = ' '.join(f'{k[4:].lower()}={v}' for k, v in env.items())
    print(f'window M={M} {tag}: vs exact {err:.2e} of peak (worst frame {worst}), whole pool vs pair-read {dp:.2e}   {desc}', flush=True)
    if not desc.startswith('conv_spec_window'):
        print('   !! the window kernel did not run:', desc, flush=True)
    if err > 2e-6 or dp > 4e-6:
        bad = (d.amax(dim=2) > 2e-6 * peak).nonzero()
        print('   !! frames off by more than 2e-6 of peak (stream, frame):', bad[:12].tolist(), 'count', len(bad), flush=True)
    return err


configs = [(32, dict(VND_SPEC_NT=256)), (32, dict(VND_SPEC_NT=128)), (32, dict(VND_SPEC_NT=192)), (16, dict(VND_SPEC_NT=256)),
           (64, dict(VND_SPEC_NT=128)), (64, dict(VND_SPEC_NT=64)), (32, dict(VND_SPEC_NT=64)), (32, dict(VND_SPEC_NT=256, VND_SPEC_LA=10)),
           (32, dict(VND_SPEC_NT=256, VND_SPEC_LA=3))]
if quick:
    configs = configs[:3]
ok = {}
for M, env in configs:
    try:
        ok[M, tuple(env.items())] = check(M, **env)
    except Exception as e:
        print(f'window M={M} {env}: {e!r}', flush=True)
for rep in range(2):
    env_set()
    rate(GENERIC, 'generic fast')
    rate(WIN[0], 'pair-read per-table')
    for M, env in configs:
        if (M, tuple(env.items())) not in ok:
            continue
        env_set(**env)
        tag = ' '.join(f'{k[4:].lower()}={v}' for k, v in env.items())
        rate(WIN[M], f'window M={M} {tag}')
