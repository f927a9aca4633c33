import os, sys
sys.path.insert(0, '/root/repo')
os.environ['VND_SPEC_VERBOSE'] = '1'
import numpy as np
os.environ.setdefault('VND_TUNING', '1')      # geometry variables are read live
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
a = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
t = _native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)
for nt in (64, 128, 192, 256):
    os.environ['VND_SPEC_NT'] = str(nt)
    for M, bits in ((32, 3 << 5), (16, 2 << 5), (64, 4 << 5)):
        ctx.set_variant((1 << 23) | bits)
        print(nt, M, t.describe(3, 40000, 2, 2)[:110], flush=True)
