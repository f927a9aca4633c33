#!/usr/bin/env python3
"""Round 6: where the exact stage should switch from the block-parallel NumPy-order sums (tally + stitch, block sums from the
convolution's store phase) to the per-stream kernel: pools of 192 ... 1024 stereo signals x 10 s, default choice against the block-parallel
form forced (variant bit 17) and the per-stream kernel forced (bit 19); bit-identical to the oracle's stage on the last stream."""
import os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from oracle import vnd_oracle as O
n = 480000
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
table = vn._device_table()
ctx = _native.default_context()
st = torch.cuda.current_stream().cuda_stream
for pool in ([int(a) for a in sys.argv[1:]] or [192, 256, 320, 384, 512, 1024]):
    torch.manual_seed(pool)
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty_like(x)
    ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    want = O.decorrelate(x[pool - 1].cpu().numpy(), sample_rate_hz=48000, seed=1)
    for r in range(2):
        for label, variant in (('default', -1), ('block-parallel', 1 << 17), ('per-stream', 1 << 19)):
            ctx.set_variant(variant)
            run = lambda: table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=vnd.MODE_EXACT, ms_encode=True, width=None, normalize=1,
                                                   workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            ok = np.array_equal(y[pool - 1].cpu().numpy(), want)
            print(f'pool {pool:5d} [{r}] {label:15s} {ms:.4f} ms  {24e-9 * pool * n * 2 / ms / 8.0:.4f} of 8 TB/s  bit-identical: {ok}', flush=True)
    ctx.set_variant(-1)
    del x, y, ws
    torch.cuda.empty_cache()
