import os, sys
sys.path.insert(0, '/root/repo')
os.environ['VND_TUNING'] = '1'
import numpy as np
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
WIN = {16: 2 << 5, 32: 3 << 5, 64: 4 << 5}
def table_of(**kw):
    a = function_path_arrays(vnd.generate_velvet_noise(**kw))
    return _native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)
t2 = table_of(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
t3 = table_of(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000, log_distribution_strength=0.0, seed=1)
vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
tc = vn._device_table()
for name, t, pool, n in (('cfg2', t2, 2048, 480000), ('cfg3', t3, 24, 2880000), ('cfg4', t2, 1024, 48000), ('class', tc, 2048, 480000)):
    for mode in (2, 0):
        ctx.set_variant(-1)
        print(name, mode, t.describe(pool, n, 2, mode)[:60], t.describe(pool, n, 2, mode).split('threads=')[-1])
for M, nt in ((32, 192), (32, 128), (16, 128), (16, 256), (64, 64), (64, 128)):
    os.environ['VND_SPEC_NT'] = str(nt)
    for name, t in (('cfg2', t2), ('cfg3', t3)):
        for mode in (2, 0):
            ctx.set_variant(WIN[M] | (1 << 23))
            d = t.describe(64, 480000, 2, mode)
            print(f'forced M={M} nt={nt} {name} mode={mode}:', d[:40], d.split('threads=')[-1])
