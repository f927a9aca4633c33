#!/usr/bin/env python3
"""Where a pass of cfg4's N = 8 shard (128 x 1 s stereo) spends its time: host issue cost per call, per-pass time of the
Python loop and of the C loop, the same as a hipGraph replay (torch.cuda.CUDAGraph around the C-ABI launches), and - under
`rocprofv3 --kernel-trace` (tools/shard_timeline.sh) - kernel durations against the gaps between them.
usage: shard_timeline.py [streams] [passes]"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd import _native
from vndecorrelate_amd.taps import function_path_arrays

mine = int(sys.argv[1]) if len(sys.argv) > 1 else 128
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 600
n, mode = 48000, vnd.MODE_FAST
ctx = _native.default_context()
arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
buffers = max(2, int(np.ceil(600e6 / (mine * n * 2 * 4 * 2))))
xs = [torch.empty((mine, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1) for _ in range(buffers)]
ys = [torch.empty_like(xs[0]) for _ in range(buffers)]
table.prepare(mine, n, 2, mode)
print(table.describe(mine, n, 2, mode), flush=True)
side = torch.cuda.Stream()
st = side.cuda_stream


def step(i):
    table.convolve_device(xs[i % buffers].data_ptr(), ys[i % buffers].data_ptr(), mine, n, 2, mode, st)


def loop(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        e0.record()
        t0 = time.perf_counter()
        for i in range(reps):
            step(i)
        t_issue = time.perf_counter() - t0
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, t_issue / reps * 1e6


for i in range(300):
    step(i)
torch.cuda.synchronize()
res = sorted(loop(passes) for _ in range(7))
print(f'python loop: us per pass min {res[0][0]:.2f} median {res[3][0]:.2f} max {res[-1][0]:.2f}; host issue {res[3][1]:.2f} us per call', flush=True)

# the same passes as ONE graph replay: no host in the loop, the runtime's own dependent-dispatch boundary
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    for i in range(8):
        step(i)
    side.synchronize()
    try:
        with torch.cuda.graph(g, stream=side):
            for i in range(passes):
                step(i)
        ok = True
    except Exception as exc:       # capture of a module launch may be refused
        print('graph capture failed:', repr(exc)[:200])
        ok = False
if ok:
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / passes * 1e3)
    ts.sort()
    print(f'hipGraph replay of {passes} passes: us per pass min {ts[0]:.2f} median {ts[3]:.2f} max {ts[-1]:.2f}', flush=True)

# copy floor, same loop shape
def cstep(i):
    ys[i % buffers].copy_(xs[i % buffers])
with torch.cuda.stream(side):
    for i in range(100): cstep(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    side.synchronize()
    e0.record()
    for i in range(passes): cstep(i)
    e1.record()
torch.cuda.synchronize()
print(f'torch copy of the shard: {e0.elapsed_time(e1) / passes * 1e3:.2f} us per pass', flush=True)
ms = ctx.time_copy(xs[0].data_ptr(), ys[0].data_ptr(), xs[0].numel(), passes, st)
print(f'library copy kernel (one buffer, cache-resident at this size): {ms * 1e3:.2f} us per pass', flush=True)
