#!/usr/bin/env python3
"""The window form with its waves SPLIT over the two channels of a stereo signal (VW_S, vw_span_s: a wave computes one channel,
three waves per SIMD) against the plain window form (a lane computes both channels, two waves per SIMD):
(1) small ragged signals, span seams and stream tails against the NumPy oracle (fast: of peak; exact: bit for bit);
(2) cfg3 (24 x 60 s, 128 taps), cfg3 kappa 1, cfg2 (128 x 10 s, 30 taps), the class-path table: sustained rate, fast and exact.
usage: win_split_try.py [seconds per variant] [skip-small]"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
os.environ.setdefault('VND_TUNING', '1')
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
from oracle import vnd_oracle as O

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
ctx = _native.default_context()
FORCE = 1 << 23
WIN = {0: 1 << 5, 16: 2 << 5, 32: 3 << 5}
KW = {'cfg2': dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1),
      'cfg3': dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1),
      'cfg3k1': dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=1.0, seed=1)}


def make_table(fir):
    a = function_path_arrays(np.ascontiguousarray(fir))
    return _native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)


def env_set(**env):
    for k in ('VND_SPEC_NT', 'VND_SPEC_LA', 'VND_WIN_G', 'VND_WIN_SPLIT', 'VND_WIN_SPLIT_LATE', 'VND_WIN_ALLOW_SPILL'):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)


bad = 0
notsplit = set()
if 'skip-small' not in sys.argv:
    rng = np.random.default_rng(7)
    for name, M, nt in (('cfg3', 32, 384), ('cfg2', 32, 384), ('cfg2', 16, 256), ('cfg3k1', 32, 256), ('cfg2', 32, 128)):
        fir = vnd.generate_velvet_noise(**KW[name])
        table = make_table(fir)
        env_set(VND_SPEC_NT=nt, VND_WIN_SPLIT=2)
        T = (nt // 2) * M
        for n in sorted({1, 3, M + 1, T - 1, T, T + 1, 2 * T + 3, 5 * T + 17, 40003}):
            for batch in (1, 3):
                if batch > 1 and n % 2:
                    continue
                x = rng.uniform(-1, 1, (batch, n, 2)).astype(np.float32)
                want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(batch)])
                peak = float(np.abs(want).max()) or 1.0
                for min_span, rounds in ((1, 7), (2, 1)):
                    ctx.set_variant(FORCE | WIN[M] | (min_span << 20) | (rounds << 28))
                    for mode in (2, 0):
                        text = table.describe(batch, n, 2, mode)
                        if 'split-by-channel' not in text or f'tile={T} ' not in text:
                            notsplit.add((name, M, nt, mode))          # (a build that spills is rejected: the plain form runs)
                        got = table.convolve_host(x, mode)
                        if mode == 0:
                            ok = np.array_equal(got, want)
                            what = 'bit-identical' if ok else f'DIFFERS max {np.abs(got - want).max():.3e}'
                        else:
                            err = float(np.abs(got.astype(np.float64) - want).max()) / peak
                            ok = err <= (2e-6 if 'cfg3' in name else 1e-6)
                            what = f'{err:.2e} of peak'
                        if not ok:
                            bad += 1
                            w = np.argwhere(~np.isclose(got, want, rtol=0, atol=4e-6 * peak))
                            print(f'FAIL {name} M={M} nt={nt} n={n} batch={batch} spans=({min_span},{rounds}) mode={mode}: {what}; first bad {w[:4].tolist()} of {len(w)}', flush=True)
        print(f'{name} M={M} nt={nt}: small shapes done, failures so far {bad}', flush=True)
        ctx.set_variant(-1)
        table.close()
    env_set()
    print('not in the split form (build rejected):', sorted(notsplit))
    if bad:
        print('small-shape failures:', bad)
        sys.exit(1)

st = torch.cuda.current_stream().cuda_stream
POOLS = {'cfg3': (24, 2880000), 'cfg3k1': (24, 2880000), 'cfg2': (128, 480000), 'class': (128, 480000)}
if 'cfg3-only' in sys.argv:
    POOLS = {k: v for k, v in POOLS.items() if k.startswith('cfg3')}
if 'cfg2-only' in sys.argv:
    POOLS = {k: v for k, v in POOLS.items() if not k.startswith('cfg3')}
for name, (pool, n) in POOLS.items():
    if name == 'class':
        table = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)._device_table()
    else:
        table = make_table(vnd.generate_velvet_noise(**KW[name]))
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty_like(x)

    def run(mode):
        y.zero_()
        table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, stream=st)
        torch.cuda.synchronize()
        return y.clone()

    def rate(mode, label):
        desc = table.describe(pool, n, 2, mode)
        t0 = time.perf_counter(); best = []
        while time.perf_counter() - t0 < seconds:
            best.append(table.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, n_buffers=1, stride_elems=0, iters=30, stream=st))
        tail = best[len(best) // 2:]
        print(f'{name:7s} {label:28s} {np.mean(tail):.4f} ms {8e-6 * pool * n * 2 / np.mean(tail):6.0f} GB/s (min {min(best):.4f})  {desc[:175]}', flush=True)

    ctx.set_variant(-1)
    env_set(VND_WIN_SPLIT=0)
    ye, yf = run(0), run(2)
    peak = float(ye.abs().max())
    V64 = 4 << 5
    configs = [('plain', dict(VND_WIN_SPLIT=0)), ('default', {}),
               ('split 64x256 forced', dict(VND_WIN_SPLIT=2, VND_SPEC_NT=256, variant=V64))]
    if 'late' in sys.argv:           # the fast mode's 64-frame form with (almost) the whole refill loaded late
        configs += [('split 64 late=15 la=2', dict(VND_WIN_SPLIT=2, VND_SPEC_NT=256, VND_WIN_SPLIT_LATE=15, VND_SPEC_LA=2, variant=V64))]
    if 'all' in sys.argv:
        configs += [('split 32x384', dict(VND_WIN_SPLIT=2, VND_SPEC_NT=384)), ('split 32x256 la=3', dict(VND_WIN_SPLIT=2, VND_SPEC_LA=3)),
                    ('split 32x256 la=6', dict(VND_WIN_SPLIT=2, VND_SPEC_LA=6)), ('split 32x512', dict(VND_WIN_SPLIT=2, VND_SPEC_NT=512))]
    ok = []
    for label, env in configs:
        env = dict(env)
        ctx_variant = env.pop('variant', -1)
        ctx.set_variant(ctx_variant)
        env_set(**env)
        try:
            f, e = run(2), run(0)
            print(f'{name} {label}: fast vs exact {float((f - ye).abs().max()) / peak:.2e} of peak (fast vs plain fast {float((f - yf).abs().max()) / peak:.1e}); exact bit-identical to the plain form: {bool(torch.equal(e, ye))}', flush=True)
            ok.append((label, dict(env, variant=ctx_variant)))
        except Exception as exc:
            print(f'{name} {label}: {exc!r}', flush=True)
    for rep in range(2):
        for label, env in ok:
            env = dict(env)
            ctx.set_variant(env.pop('variant', -1))
            env_set(**env)
            rate(2, f'fast  {label}')
            rate(0, f'exact {label}')
    table.close()
    del x, y
    torch.cuda.empty_cache()
