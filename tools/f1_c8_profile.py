import sys, pathlib
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
fs8, n8, pool8 = 96000, 960000, 16
vn8 = vnd.VelvetNoise(sample_rate_hz=fs8, num_outs=8, num_impulses=64, filtered_channels=tuple(range(8)), mode='LR', seed=1)
table = vn8._device_table()
st = torch.cuda.current_stream().cuda_stream
x = torch.empty((pool8, n8, 8), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
ws_bytes = _native.decorrelate_workspace_bytes(pool8, n8, 8)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
mode = vnd.MODE_EXACT if sys.argv[1] == 'exact' else vnd.MODE_FAST
table.prepare(pool8, n8, 8, mode)
for _ in range(30):
    table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool8, n8, 8, mode=mode, ms_encode=False, width=None, normalize=1, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
torch.cuda.synchronize()
