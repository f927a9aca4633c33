"""Round 6: the block-parallel NumPy-order sums with their staging loads on consecutive bytes per wave access (VND_EPI_PAR_COALESCED=1, the
default) against 32 consecutive bytes per lane as two accesses (=0), interleaved, the exact decorrelate stage over pools of 128 / 64 / 32
signals, stereo and mono; every output the same bits."""
import os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1])); os.environ['VND_TUNING'] = '1'
import numpy as np, torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
n = 480000
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
table = vn._device_table()
st = torch.cuda.current_stream().cuda_stream
for pool, cx in ((128, 2), (128, 1), (64, 2), (32, 2)):
    torch.manual_seed(3)
    x = torch.empty((pool, n, cx), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
    ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    table.prepare(pool, n, cx, vnd.MODE_EXACT)
    ref = None
    for r in range(2):
        for coal in ('0', '1'):
            os.environ['VND_EPI_PAR_COALESCED'] = coal
            run = lambda: table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, cx, mode=vnd.MODE_EXACT, ms_encode=True, width=None, normalize=1, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
            for _ in range(5): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(30): run()
            e1.record(); torch.cuda.synchronize()
            if ref is None: ref = y.clone()
            print(f'pool {pool} cx {cx} [{r}] coalesced={coal}: {e0.elapsed_time(e1)/30:.4f} ms  same bits: {bool(torch.equal(y, ref))}', flush=True)
