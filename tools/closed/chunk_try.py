#!/usr/bin/env python3
"""Small one-round launches cut into CU chunks (VND_WIN_CHUNKS) against the uniform spans: every stream of the launch against the
exact kernel / the C oracle first, then the per-pass time of a hipGraph replay for each setting.
usage: chunk_try.py [streams frames]..."""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
os.environ['VND_TUNING'] = '1'
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
from oracle import c_oracle

ctx = _native.default_context()
arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
taps = (arr.tap_offsets, arr.tap_index, arr.tap_weight)
forced = [a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('len0=')]          # len0=4: also try a forced split of a CU's chunk (first workgroup's tiles)
sys.argv = [a for a in sys.argv if not a.startswith('len0=')]
shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(128, 48000), (64, 96000), (256, 30000), (128, 40000), (32, 200000)]
side = torch.cuda.Stream()


def per_pass(table, xs, ys, mine, n, mode, passes=400):
    buffers = len(xs)
    def step(i):
        table.convolve_device(xs[i % buffers].data_ptr(), ys[i % buffers].data_ptr(), mine, n, 2, mode, side.cuda_stream)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for i in range(20):
            step(i)
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            for i in range(passes):
                step(i)
        g.replay(); side.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record()
            side.synchronize()
            ts.append(e0.elapsed_time(e1) / passes * 1e3)
    return min(ts), sorted(ts)[2]


for mine, n in shapes:
    buffers = max(2, int(np.ceil(600e6 / (mine * n * 2 * 4 * 2))))
    xs = [torch.empty((mine, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1) for _ in range(buffers)]
    ys = [torch.empty_like(xs[0]) for _ in range(buffers)]
    for env in (dict(VND_WIN_CHUNKS='0'), dict(VND_WIN_CHUNKS='1'), dict(VND_WIN_CHUNKS='1', VND_WIN_STAGGER_TICKS='150'),
                dict(VND_WIN_CHUNKS='1', VND_WIN_STAGGER_TICKS='300'), dict(VND_WIN_CHUNKS='1', VND_WIN_STAGGER_TICKS='450')) + tuple(
                dict(VND_WIN_CHUNKS='1', VND_WIN_CHUNK_LEN0=f, VND_WIN_STAGGER_TICKS=t) for f in forced for t in ('300', '600')):
        for k in ('VND_WIN_CHUNKS', 'VND_WIN_STAGGER_TICKS', 'VND_WIN_CHUNK_LEN0'):
            os.environ.pop(k, None)
        os.environ.update(env)
        table = _native.TapTable.create(ctx, *taps)
        line = []
        for mode in (vnd.MODE_FAST, vnd.MODE_EXACT):
            table.prepare(mine, n, 2, mode)
            desc = table.describe(mine, n, 2, mode)
            # parity: the whole launch against the oracle on a few streams, and fast vs exact everywhere
            y = torch.zeros_like(xs[0])
            table.convolve_device(xs[0].data_ptr(), y.data_ptr(), mine, n, 2, mode, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = y.cpu().numpy()
            worst = 0.0
            for b in sorted({0, 1, mine // 2, mine - 1}):
                want = c_oracle.convolve(xs[0][b].cpu().numpy(), *taps, threads=8)
                if mode == vnd.MODE_EXACT:
                    assert np.array_equal(got[b], want), f'exact mode differs from the oracle: stream {b} {env}'
                else:
                    worst = max(worst, float(np.abs(got[b].astype(np.float64) - want).max() / np.abs(want).max()))
            if mode == vnd.MODE_EXACT:
                exact_all = got
            else:
                fast_all = got
            best, med = per_pass(table, xs, ys, mine, n, mode)
            line.append(f'{"fast" if mode == vnd.MODE_FAST else "exact"} {best:6.2f} / {med:6.2f} us' + (f' (<= {worst:.1e} of peak)' if mode == vnd.MODE_FAST else ' (bit-identical)'))
        err = float(np.abs(fast_all.astype(np.float64) - exact_all).max() / np.abs(exact_all).max())
        assert err <= 1e-6, f'fast vs exact over the whole launch: {err:.2e}'
        print(f'{mine:4d} x {n:6d}  {str(env):62s} {"   ".join(line)}   all streams fast vs exact {err:.1e}\n      {desc[desc.find("workgroups="):]}', flush=True)
        table.close()
    del xs, ys
    torch.cuda.empty_cache()
