import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
table = vn._device_table()
for pool, n in ((128, 480000), (1, 480000), (1024, 48000)):
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty_like(x)
    ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    def run(norm):
        table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=0, ms_encode=True, width=None,
                                 normalize=norm, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
    res = []
    for norm in (False, True):
        for _ in range(20): run(norm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run(norm)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50)
    print(f'pool {pool:5d} n {n}: without normaliser {res[0]:.3f} ms, with {res[1]:.3f} ms -> normaliser {res[1]-res[0]:.3f} ms')
