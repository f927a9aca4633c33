#!/usr/bin/env python3
"""Device-resident rate of the whole VelvetNoise.decorrelate stage (convolution +
epilogue, vnd_decorrelate_f32_dev) on the cfg2 pool, next to the convolution alone."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native

vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
table = vn._device_table()
pool, n = 128, 480000
x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def run(kind, mode):
    if kind == 'convolve':
        table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode, st)
    else:
        table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, ms_encode=True, width=None,
                                 normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
for kind in ('convolve', 'decorrelate'):
    for mode, name in ((2, 'fast'), (0, 'exact')):
        for _ in range(300): run(kind, mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): run(kind, mode)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
        print(f'{kind:12s} {name:5s} {ms:.4f} ms per pool  {x.numel()/ms/1e3:10.0f} Msamples/s  {8*x.numel()/ms/1e6:7.1f} GB/s algorithmic')
