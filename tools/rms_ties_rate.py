#!/usr/bin/env python3
"""The exact RMS sums on signals whose squares tie often (audio that came from 16-bit integers): block-parallel
kernels against the one-workgroup-per-stream kernel, one 10 s stereo signal."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native

ctx = _native.default_context()
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
table = vn._device_table()
st = torch.cuda.current_stream().cuda_stream
n = 480000
rng = np.random.default_rng(5)
signals = {
    'uniform float': rng.uniform(-1, 1, (1, n, 2)).astype(np.float32),
    'int16 / 32768': (rng.integers(-32768, 32767, (1, n, 2)) / 32768.0).astype(np.float32),
    'int16 music-like (sine mix) / 32768': (np.round(8000 * (np.sin(np.arange(n) * 0.01)[:, None] * [1.0, 0.7]
                                                             + 0.3 * rng.standard_normal((n, 2)))) / 32768.0).astype(np.float32)[None],
    'int16 values as floats': rng.integers(-32768, 32767, (1, n, 2)).astype(np.float32),
}
SEQ = 1 << 19
for name, sig in signals.items():
    x = torch.from_numpy(np.ascontiguousarray(sig)).cuda()
    y = torch.empty_like(x)
    ws_bytes = _native.decorrelate_workspace_bytes(1, n, 2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    outs = {}
    for label, variant in (('sequential', SEQ), ('parallel', -1)):
        ctx.set_variant(variant)
        def run():
            table.decorrelate_device(x.data_ptr(), y.data_ptr(), 1, n, 2, mode=0, ms_encode=True, width=None,
                                     normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
        for _ in range(10): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        outs[label] = y.clone()
        print(f'{name:38s} {label:10s} whole exact stage {e0.elapsed_time(e1) / 50:.4f} ms', flush=True)
    assert torch.equal(outs['sequential'], outs['parallel'])
ctx.set_variant(-1)
