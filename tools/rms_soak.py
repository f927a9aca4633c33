#!/usr/bin/env python3
"""Soak of the exact stage's NumPy-order sums: the block-parallel kernels (tally: plain blocks whole, ulp(e + 1) tallies only near a
crossing) against the one-workgroup-per-stream kernel, bit for bit, over many long signals of awkward kinds - and four of each
kind against np.mean(np.square(.)) through the oracle's stage.  usage: rms_soak.py [rounds]"""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from oracle import vnd_oracle as O

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = _native.default_context()
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
table = vn._device_table()
st = torch.cuda.current_stream().cuda_stream
bad = 0
for r in range(rounds):
    rng = np.random.default_rng(100 + r)
    n = int(rng.integers(300000, 700000)) & ~1
    pool = 48
    x = np.empty((pool, n, 2), np.float32)
    t = np.arange(n)
    for b in range(pool):
        kind = b % 8
        if kind == 0: x[b] = rng.uniform(-1, 1, (n, 2))
        elif kind == 1: x[b] = np.round(rng.uniform(-1, 1, (n, 2)) * 32767) / 32768.0                      # 16-bit audio: ties
        elif kind == 2: x[b] = rng.uniform(-1, 1, (n, 2)) * np.linspace(1e-4, 1.0, n)[:, None]             # growing level: many binades
        elif kind == 3: x[b] = rng.uniform(-1, 1, (n, 2)) * (rng.random((n, 1)) < 0.01)                    # sparse
        elif kind == 4: x[b] = rng.uniform(-1, 1, (n, 2)) * 1e-18                                          # squares near the denormals
        elif kind == 5: x[b] = np.where(t[:, None] < n // 3, 0.0, rng.uniform(-1, 1, (n, 2)))              # a silent start
        elif kind == 6: x[b] = (rng.integers(-3, 4, (n, 2)) * 0.25)                                        # a few exact values: ties everywhere
        else: x[b] = rng.standard_normal((n, 2)) * 30.0                                                    # large
    xs = torch.from_numpy(x).cuda()
    ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    out = {}
    for name, variant in (('block_parallel', -1), ('per_stream', 1 << 19)):
        ctx.set_variant(variant)
        y = torch.empty_like(xs)
        table.decorrelate_device(xs.data_ptr(), y.data_ptr(), pool, n, 2, mode=vnd.MODE_EXACT, ms_encode=True, width=None, normalize=1,
                                 workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
        torch.cuda.synchronize()
        out[name] = y.cpu().numpy()
        ctx.set_variant(-1)
    same = [bool(np.array_equal(out['block_parallel'][b], out['per_stream'][b], equal_nan=True)) for b in range(pool)]
    ref = [bool(np.array_equal(out['block_parallel'][b], O.decorrelate(x[b], sample_rate_hz=48000, seed=1), equal_nan=True)) for b in range(8)]
    bad += same.count(False) + ref.count(False)
    print(f'round {r}: n = {n}, {pool} signals: block-parallel == per-stream for {same.count(True)} / {pool}; == NumPy stage for {ref.count(True)} / 8 kinds', flush=True)
print('FAILED' if bad else 'all identical')
sys.exit(1 if bad else 0)
