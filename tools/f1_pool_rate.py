#!/usr/bin/env python3
"""The whole decorrelate stage (MS encode + RMS normaliser) over a resident pool of cfg2 signals, exact and fully fused fast mode:
the loop rocprofv3 wraps for profiles/r04_f1_pool_*.  usage: f1_pool_rate.py [pool ...]"""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
ctx = _native.default_context()
table = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)._device_table()
st = torch.cuda.current_stream().cuda_stream
n = 480000
for pool in [int(a) for a in sys.argv[1:]] or [256, 128]:
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty_like(x)
    ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    for label, mode, bps in (('exact', vnd.MODE_EXACT, 24), ('fast fused', vnd.MODE_FAST, 16)):
        table.prepare(pool, n, 2, mode)
        def run():
            table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, ms_encode=True, width=None,
                                     normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print(f'pool {pool:4d} {label:10s}: {ms:.4f} ms per call = {bps * pool * n * 2 / ms / 1e9:.2f} TB/s of its {bps} B/sample', flush=True)
    del x, y, ws
    torch.cuda.empty_cache()
