#!/usr/bin/env python3
"""The exact (NumPy-order) RMS sums of VelvetNoise.decorrelate: block-parallel kernels (rms_par_*) against
the one-workgroup-per-stream kernel (variant bit 19), for one 10 s stereo signal and for the cfg2 pool.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel times committed in profiles/."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native

ctx = _native.default_context()
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
table = vn._device_table()
st = torch.cuda.current_stream().cuda_stream
SEQ = 1 << 19
for pool in (1, 128):
    n = 480000
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty_like(x)
    ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    outs = {}
    for name, variant in (('sequential', SEQ), ('parallel', -1), ('sequential', SEQ), ('parallel', -1)):
        ctx.set_variant(variant)
        def run():
            table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=0, ms_encode=True, width=None,
                                     normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
        for _ in range(20): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 100 if pool == 1 else 30
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        outs[name] = y.clone()
        print(f'pool {pool:4d}  {name:10s} whole exact stage {e0.elapsed_time(e1) / reps:.4f} ms', flush=True)
    assert torch.equal(outs['sequential'], outs['parallel']), 'parallel sums changed the result'
    print(f'pool {pool}: outputs bit-identical')
ctx.set_variant(-1)
