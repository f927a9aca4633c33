#!/usr/bin/env python3
"""The exact RMS sums on batches: the one-workgroup-per-stream kernel against the block-parallel kernels forced
for any batch (variant bit 17), whole exact stage, 10 s stereo signals."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
ctx = _native.default_context()
table = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)._device_table()
st = torch.cuda.current_stream().cuda_stream
n = 480000
for pool in (16, 64, 128, 256, 512, 1024):
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty_like(x)
    ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    res = {}
    for label, variant in (('per-stream', 1 << 19), ('block-parallel', 1 << 17), ('per-stream', 1 << 19), ('block-parallel', 1 << 17)):
        ctx.set_variant(variant)
        def run():
            table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=0, ms_encode=True, width=None,
                                     normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(label, []).append(e0.elapsed_time(e1) / 10)
    print(f'pool {pool:5d}: per-stream {min(res["per-stream"]):.4f} ms   block-parallel {min(res["block-parallel"]):.4f} ms', flush=True)
    del x, y, ws
    torch.cuda.empty_cache()
ctx.set_variant(-1)
