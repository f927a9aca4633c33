import os, sys, time
sys.path.insert(0, '/root/repo')
os.environ['VND_TUNING'] = '1'; os.environ['VND_SPEC_VERBOSE'] = '1'
import numpy as np, torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
for which in ('cfg3', 'cfg2'):
    kw = dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1); pool, n = 128, 480000
    if which == 'cfg3':
        kw.update(num_impulses=128, log_distribution_strength=0.0); pool, n = 24, 2880000
    arr = function_path_arrays(vnd.generate_velvet_noise(**kw))
    table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1); y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    for nt, la in ((256, 4), (64, 2), (64, 1), (128, 2), (64, 3)):
        os.environ['VND_SPEC_NT'] = str(nt); os.environ['VND_SPEC_LA'] = str(la)
        ctx.set_variant(3 << 5)
        desc = table.describe(pool, n, 2, 2)
        best = []
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.6:
            best.append(table.time_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=2, n_buffers=1, stride_elems=0, iters=100 if which == 'cfg2' else 30, stream=st))
        tail = best[len(best) // 2:]
        print(f'{which} nt={nt} LA={la}  {np.mean(tail):.4f} ms  {8e-6 * pool * n * 2 / np.mean(tail):6.0f} GB/s  {desc[:120]}', flush=True)
    del x, y; table.close(); torch.cuda.empty_cache()
