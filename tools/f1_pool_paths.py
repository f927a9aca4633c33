#!/usr/bin/env python3
"""Exact decorrelate stage over pools of 256 / 128 ten-second stereo signals by path: sequential sums (one workgroup per stream) against
block-parallel sums (variant bit 17), 16-byte against 8-byte staging loads of the sums kernels (VND_EPI_WIDE); interleaved repeats."""
import os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
os.environ['VND_TUNING'] = '1'
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
ctx = _native.default_context()
st = torch.cuda.current_stream().cuda_stream
n = 480000
for pool in (256, 128, 512):
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty_like(x)
    ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    for rep in range(2):
        for label, variant, wide in (('default', -1, '1'), ('default, 8-byte staging', -1, '0'), ('block-parallel sums', 1 << 17, '1'), ('block-parallel, 8-byte staging', 1 << 17, '0')):
            os.environ['VND_EPI_WIDE'] = wide
            ctx.set_variant(variant)
            table = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)._device_table()
            def run():
                table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=vnd.MODE_EXACT, ms_encode=True, width=None,
                                         normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
            for _ in range(5): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(30): run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 30
            print(f'pool {pool:4d} {label:32s}: {ms:.4f} ms per call = {24 * pool * n * 2 / ms / 1e9:.2f} TB/s = {24 * pool * n * 2 / ms / 8e9:.3f} of 8 TB/s', flush=True)
    ctx.set_variant(-1)
    del x, y, ws
    torch.cuda.empty_cache()
