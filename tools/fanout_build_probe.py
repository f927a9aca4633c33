import os, sys
sys.path.insert(0, '.')
os.environ['VND_TUNING']='1'; os.environ['VND_SPEC_VERBOSE']='1'
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
for env in ({}, {'VND_WIN_SPLIT_LATE':'6'}, {'VND_WIN_SPLIT_LATE':'4'}, {'VND_SPEC_LA':'1'}, {'VND_WIN_OPAQUE_IN':'0'}):
    os.environ.update(env)
    table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
    table.prepare(128, 480000, 1, 2)
    print(env, table.describe(128, 480000, 1, 2)[:120], flush=True)
    for k in env: os.environ.pop(k)
    table.close()
