import sys, time, pathlib
sys.path.insert(0, '.')
import numpy as np, torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
n = 300_000_000
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
table = vn._device_table()
x = torch.empty((1, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
y = torch.empty_like(x)
ws_bytes = _native.decorrelate_workspace_bytes(1, n, 2)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
ctx = _native.default_context()
for label, variant in (('parallel sums', -1), ('sequential sums', 1 << 19)):
    ctx.set_variant(variant)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        table.decorrelate_device(x.data_ptr(), y.data_ptr(), 1, n, 2, mode=0, ms_encode=True, width=None, normalize=True,
                                 workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
        torch.cuda.synchronize(); print(label, 'device stage', round((time.perf_counter() - t) * 1e3, 1), 'ms', flush=True)
ctx.set_variant(-1)
del x, y
xh = np.random.default_rng(0).uniform(-1, 1, (n // 4, 2)).astype(np.float32)
for rep in range(3):
    t = time.perf_counter(); out = vn.decorrelate(xh); print('host decorrelate n/4', round((time.perf_counter() - t) * 1e3, 1), 'ms', _native.pinned_pool.hits, _native.pinned_pool.misses, flush=True)
    del out
