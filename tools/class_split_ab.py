#!/usr/bin/env python3
"""Round 6: VelvetNoise.convolve's table (class path: per segment the negative taps, then the positive ones, one gain per segment -
decorrelation.py:402-414) in VND_MODE_EXACT on the headline's shape: the plain 32-frame window form (VND_WIN_SPLIT_CLASS=0) against the
split 64-frame form with the fast mode's late refill (=1), interleaved; stereo in, and a mono input fanned out; the last stream of
every leg bit for bit against the oracle's class path.  usage: class_split_ab.py [repeats=N] [pool=N]"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
os.environ['VND_TUNING'] = '1'
import numpy as np
import torch
import bench
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from oracle import vnd_oracle as O

repeats = next((int(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('repeats=')), 2)
pool = next((int(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('pool=')), 1024)
power = bench.PowerSampler(torch, 0)
n = 480000
st = torch.cuda.current_stream().cuda_stream
for name, kw, cx in (('class 30 taps, stereo', dict(), 2), ('class 30 taps, mono in', dict(), 1), ('class 128 taps, stereo', dict(num_impulses=128), 2)):
    vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1, **kw)
    table = vn._device_table()
    taps = O.generate_class_taps(sample_rate_hz=48000, seed=1, **kw)
    batch = pool if cx == 2 else pool // 4
    torch.manual_seed(21)
    x = torch.empty((batch, n, cx), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty((batch, n, 2), dtype=torch.float32, device='cuda')
    for r in range(repeats):
        for split in ('0', '1'):
            os.environ['VND_WIN_SPLIT_CLASS'] = split
            run = lambda k=0: table.convolve_device(x.data_ptr(), y.data_ptr(), batch, n, cx, vnd.MODE_EXACT, st)
            run(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.12:
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            iters = 40 if cx == 2 else 160
            e0.record()
            for _ in range(iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            board = bench.board_under(torch, power, run) or {}
            xs = x[batch - 1].cpu().numpy()
            if cx == 1:
                xs = np.ascontiguousarray(np.repeat(xs, 2, axis=1))
            want = O.class_convolve(xs, taps, (0.85, 0.55, 0.35, 0.2), 2)
            assert np.array_equal(y[batch - 1].cpu().numpy(), want), 'differs from the class-path oracle'
            d = table.describe(batch, n, cx, vnd.MODE_EXACT)
            per_frame = 16 if cx == 2 else 12
            print(f"{name:24s} [{r}] split_class={split}  {ms:.4f} ms  {per_frame * 1e-9 * batch * n / ms / 8.0:.4f} of 8 TB/s  {board.get('power_W')} W {board.get('sclk_MHz')} MHz  "
                  f"bit-identical  {d[:22]} {d[d.find('frames_per_lane'):][:48]} ... {d[d.find('threads='):]}", flush=True)
power.close()
