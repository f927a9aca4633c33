#!/usr/bin/env python3
"""The window form on channel QUADS (VW_Q, vw_span_q: 16 bytes of every frame per workgroup) for signals of 4k channels:
(1) small ragged signals, span seams and stream tails against the NumPy oracle (fast: of peak; exact: bit for bit), C = 4, 8, 12;
(2) cfg5 (96 kHz, 8 channels, 64 taps; bench shape 16 x 10 s): parity vs the generic exact kernel and sustained rate against
    the pair-read per-table kernel and the window form on channel pairs, fast and exact.
usage: c8_quad_win_try.py [seconds per variant] [skip-small]"""
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
fir8 = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1)


def make_table(fir):
    a = function_path_arrays(np.ascontiguousarray(fir))
    return _native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)


def env_set(**env):
    for k in ('VND_SPEC_NT', 'VND_SPEC_LA', 'VND_WIN_G', 'VND_WIN_QUAD', 'VND_WIN_WIDE', 'VND_WIN_QUAD_M', 'VND_FORCE_NT', 'VND_WIN_QUAD_CU_PAIRS', 'VND_WIN_OCTET', 'VND_WIN_OCTET_SPLIT'):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)


bad = 0
if 'skip-small' not in sys.argv:
    rng = np.random.default_rng(5)
    SHAPES = ((8, 16, 512, 2), (16, 16, 512, 2), (8, 32, 256, 2), (8, 16, 256, 2), (8, 16, 256, 1), (4, 16, 256, 1), (12, 16, 128, 1), (8, 32, 128, 1), (4, 32, 256, 1))
    if 'csplit' in sys.argv:          # octets with the waves split over the channels (vw_span_qc): Q = 3 stands for it here
        SHAPES = ((8, 32, 512, 3), (16, 32, 512, 3), (8, 16, 512, 3), (4, 32, 256, 4), (12, 32, 256, 4), (4, 16, 256, 4))
    for C, M, nt, Q in SHAPES:
        if 'octets-only' in sys.argv and Q != 2:
            continue
        if C == 12 and Q == 4 and False:
            continue
        fir = np.concatenate([fir8, fir8[:, ::-1]], axis=1)[:, :C]
        table = make_table(fir)
        # (Q = 3: octets, Q = 4: quads - with the waves split over the channels)
        env_set(VND_SPEC_NT=nt, VND_WIN_OCTET=1 if Q in (2, 3) else 0, VND_WIN_OCTET_SPLIT=1 if Q >= 3 else 0)
        T = (nt // (2 * Q)) * M if Q < 3 else (nt // (8 if Q == 3 else 4)) * M
        for n in sorted({1, 3, M + 1, T - 1, T, T + 1, 2 * T + 3, 5 * T + 17, 40003}):
            for batch in (1, 3):
                x = rng.uniform(-1, 1, (batch, n, C)).astype(np.float32)
                want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(batch)])
                peak = float(np.abs(want).max()) or 1.0
                for min_span, rounds in ((1, 7), (2, 1)):
                    ctx.set_variant(FORCE | WIN[M] | (min_span << 20) | (rounds << 28))
                    for mode in (2, 0):
                        text = table.describe(batch, n, C, mode)
                        if ('channel-octets' if Q in (2, 3) else 'channel-quads') not in text or f'tile={T} ' not in text or (Q >= 3) != ('waves=split-by-channel' in text):
                            print('NOT QUAD:', C, M, nt, n, batch, text, flush=True); bad += 1
                            continue
                        got = table.convolve_host(x, mode)
                        if mode == 0:
                            ok = np.array_equal(got, want)
                            what = 'bit-identical' if ok else f'DIFFERS max {np.abs(got - want).max():.3e}'
                        else:
                            err = float(np.abs(got.astype(np.float64) - want).max()) / peak
                            ok = err <= 1e-6
                            what = f'{err:.2e} of peak'
                        if not ok:
                            bad += 1
                            w = np.argwhere(~np.isclose(got, want, rtol=0, atol=2e-6 * peak))
                            print(f'FAIL C={C} M={M} nt={nt} n={n} batch={batch} spans=({min_span},{rounds}) mode={mode}: {what}; first bad {w[:4].tolist()} of {len(w)}', flush=True)
        print(f'C={C} M={M} nt={nt} Q={Q}: small shapes done, failures so far {bad}', flush=True)
        ctx.set_variant(-1)
        table.close()
    env_set()
    if bad:
        print('small-shape failures:', bad)
        sys.exit(1)

table = make_table(fir8)
pool, n, C = 16, 960000, 8
x = torch.empty((pool, n, C), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream


def run(variant, mode):
    ctx.set_variant(variant)
    y.zero_()
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
    print(f'{label:34s} {np.mean(tail):.4f} ms {8e-6 * pool * n * C / np.mean(tail):6.0f} GB/s (min {min(best):.4f})  {desc}', flush=True)


env_set()
ye = run(1 << 25, 0)                       # the generic ordered kernel: oracle-identical (tests)
peak = float(ye.abs().max())
configs = [('octet csplit 32x512', -1, dict(VND_WIN_OCTET_SPLIT=1)), ('octet csplit 32x512 g=4', -1, dict(VND_WIN_OCTET_SPLIT=1, VND_WIN_G=4)),
           ('octet csplit 32x512 la=6', -1, dict(VND_WIN_OCTET_SPLIT=1, VND_SPEC_LA=6)), ('octet csplit 16x512', WIN[16], dict(VND_WIN_OCTET_SPLIT=1)),
           ('octet 16x512', -1, dict(VND_WIN_OCTET_SPLIT=0)), ('octet 16x512 nt-stores', -1, dict(VND_FORCE_NT=1)), ('octet 16x512 la=4', -1, dict(VND_SPEC_LA=4)),
           ('quad 16x256', -1, dict(VND_WIN_OCTET=0, VND_WIN_OCTET_SPLIT=0)), ('quad csplit 32x256', -1, dict(VND_WIN_OCTET=0)), ('quad 16x512', -1, dict(VND_WIN_OCTET=0, VND_SPEC_NT=512)),
           ('pair-read', -1, dict(VND_WIN_QUAD=0)), ('pair window 32x128', WIN[32], dict(VND_WIN_QUAD=0, VND_SPEC_NT=128))]
ok = []
for label, variant, env in configs:
    env_set(**env)
    try:
        yf = run(variant, 2)
        e_fast = float((yf - ye).abs().max()) / peak
        yx = run(variant, 0)
        same = bool(torch.equal(yx, ye))
        print(f'{label}: fast vs exact {e_fast:.2e} of peak; exact form bit-identical to the generic exact kernel: {same}   {table.describe(pool, n, C, 2)}', flush=True)
        ok.append((label, variant, env))
    except Exception as e:
        print(f'{label}: {e!r}', flush=True)
for rep in range(2):
    for label, variant, env in ok:
        env_set(**env)
        rate(variant, 2, f'fast  {label}')
        rate(variant, 0, f'exact {label}')

# the same pool bytes as FOUR-channel frames (the quad is the frame: whole lines per workgroup), first four channels' taps
table4 = make_table(fir8[:, :4])
x4 = x.view(2 * pool, n, 4); y4 = y.view(2 * pool, n, 4)
for split4 in (1, 0):
  env_set(VND_WIN_OCTET_SPLIT=split4)
  ctx.set_variant(-1)
  for mode in (2, 0):
    desc = table4.describe(2 * pool, n, 4, mode)
    best = [table4.time_device(x4.data_ptr(), y4.data_ptr(), 2 * pool, n, 4, mode=mode, n_buffers=1, stride_elems=0, iters=40, stream=st) for _ in range(12)]
    print(f'4-channel frames, channel-split {split4}, mode {mode}: {np.mean(best[6:]):.4f} ms (min {min(best):.4f})  {desc}', flush=True)
