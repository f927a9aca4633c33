#!/usr/bin/env python3
"""Round 6: VelvetNoise.decorrelate of MONO signals (decorrelation.py:428-442: mono_to_stereo, the class path's convolution, side-channel
encode, RMS normaliser) over a resident pool, device to device - the stage a `VelvetNoise(...).decorrelate_batched(mono)` caller runs -
beside the same stage on stereo input.  Bytes the stage must move per FRAME: mono in 4 + out 8 (+ the exact stage's sums 12 and the
scale pass 16 = 40; fused fast 28); stereo 8 + 8 (+ 16 + 16 = 48; fused fast 32).  usage: f1_mono_rate.py [pool]"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
os.environ['VND_TUNING'] = '1'
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from oracle import vnd_oracle as O

pool = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lr = 'lr' in sys.argv[2:]            # LR mode: the normaliser alone, no side-channel encode
n = 480000
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1, **(dict(mode='LR') if 'lr' in sys.argv[2:] else {}))
table = vn._device_table()
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(3)
for cx in (1, 2):
    x = torch.empty((pool, n, cx), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
    ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    for label, mode, fan in (('exact', vnd.MODE_EXACT, '1'), ('exact', vnd.MODE_EXACT, '0'), ('exact', vnd.MODE_EXACT, '1'), ('fast', vnd.MODE_FAST, '1')):
        if cx == 2 and fan == '0' and not lr and pool < 256:
            continue
        os.environ['VND_EPI_SUMS_ONLY'] = fan
        os.environ['VND_WIN_FANOUT_EPI'] = fan          # (0: a mono input's exact stage through the pair-read form + a pass for the block sums, as until round 6)
        table.prepare(pool, n, cx, mode)
        run = lambda: table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, cx, mode=mode, ms_encode=not lr, width=None, normalize=1,
                                               workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        xs = x[pool - 1].cpu().numpy()
        want = O.decorrelate(xs[:, 0] if cx == 1 else xs, sample_rate_hz=48000, seed=1, **(dict(mode='LR') if lr else {}))
        got = y[pool - 1].cpu().numpy()
        ok = 'bit-identical' if np.array_equal(got, want) else f'{np.max(np.abs(got.astype(np.float64) - want)) / np.max(np.abs(want)):.1e} of peak'
        d = table.describe(pool, n, cx, mode)
        print(f"{'mono' if cx == 1 else 'stereo'} in, {label:5s} fanout_epi={fan}: {ms:.4f} ms per {pool} x 10 s  = {pool * n / ms / 1e6:.1f} Gframes/s  ({ok})   plain convolution would launch: {d[:60]} ... {d[d.find('threads='):][:60]}", flush=True)
