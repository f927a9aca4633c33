#!/usr/bin/env python3
"""The N = 8 shard of cfg4 (128 streams x 1 s stereo, rotating buffers) and its larger siblings: per-pass time of the
automatic choice and of forced per-table variants, passes on ONE HIP stream and alternating over TWO (the tail of a pass
beside the head of the next: consecutive passes are independent batches).  usage: shard_try.py [streams ...]"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
os.environ.setdefault('VND_TUNING', '1')      # geometry variables are read live
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays
ctx = _native.default_context()
arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
n = 48000
FORCE, GENERIC = 1 << 23, 1 << 25
WIN = {0: 1 << 5, 16: 2 << 5, 32: 3 << 5}
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()


def span_bits(min_span, rounds):
    return (min_span << 20) | (rounds << 28)


def measure(mine, variant, env, two_streams):
    for k in ('VND_SPEC_NT',):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    ctx.set_variant(variant)
    buffers = max(2, int(np.ceil(600e6 / (mine * n * 2 * 4 * 2))))
    xs = [torch.empty((mine, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1) for _ in range(buffers)]
    ys = [torch.empty_like(xs[0]) for _ in range(buffers)]
    streams = [s1, s2] if two_streams else [s1]
    def step(i):
        st = streams[i % len(streams)]
        table.convolve_device(xs[i % buffers].data_ptr(), ys[i % buffers].data_ptr(), mine, n, 2, 2, st.cuda_stream)
    for i in range(200): step(i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(400): step(i)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 400)
    desc = table.describe(mine, n, 2, 2)
    del xs, ys
    torch.cuda.empty_cache()
    return best * 1e6, desc


for mine in [int(a) for a in sys.argv[1:]] or [128, 256]:
    variants = [('auto', -1, {}), ('generic', GENERIC, {}),
                ('pair-read forced span>=4', FORCE | WIN[0] | span_bits(4, 0), {}),
                ('window M=32 nt=128', FORCE | WIN[32], dict(VND_SPEC_NT=128)),
                ('window M=32 nt=64', FORCE | WIN[32], dict(VND_SPEC_NT=64)),
                ('window M=16 nt=128', FORCE | WIN[16], dict(VND_SPEC_NT=128)),
                ('window M=16 nt=64', FORCE | WIN[16], dict(VND_SPEC_NT=64)),
                ('window M=32 nt=256', FORCE | WIN[32], dict(VND_SPEC_NT=256))]
    for label, v, env in variants:
        for two in (False, True):
            us, desc = measure(mine, v, env, two)
            print(f'{mine:5d} streams  {label:28s} {"2 streams" if two else "1 stream "}  {us:7.1f} us/pass   {desc[:130]}', flush=True)
    # copy floor
    buffers = max(2, int(np.ceil(600e6 / (mine * n * 2 * 4 * 2))))
    xs = [torch.empty((mine, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1) for _ in range(buffers)]
    ys = [torch.empty_like(xs[0]) for _ in range(buffers)]
    for two in (False, True):
        sts = [s1, s2] if two else [s1]
        def cstep(i):
            with torch.cuda.stream(sts[i % len(sts)]):
                ys[i % buffers].copy_(xs[i % buffers])
        for i in range(100): cstep(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(400): cstep(i)
        torch.cuda.synchronize()
        print(f'{mine:5d} streams  device copy                  {"2 streams" if two else "1 stream "}  {(time.perf_counter() - t0) / 400 * 1e6:7.1f} us/pass', flush=True)
    del xs, ys
ctx.set_variant(-1)
