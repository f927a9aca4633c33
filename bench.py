#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of velvet-noise decorrelation on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json configs[1], "cfg2"): synthetic 48 kHz stereo float32, 10 s per
signal, 30 taps over 30 ms (seed 1, the reference generator's defaults).  One cfg2 signal is only
3.84 MB, which lives in the 256 MiB Infinity Cache, so a *step* is one pass of the hot path over a
resident pool of ``--pool`` DISTINCT cfg2 signals in ONE batched launch (default 2048 signals:
7.9 GB read + 7.9 GB written per step - far past the cache, SURVEY.md 8d - and long enough that the
K = 20 steps the driver asks for are ~65 ms of timed kernel, not a 4 ms sample).
"1 sample" = one float32 output channel-sample.

One process per GPU (torch.distributed / RCCL when launched by torchrun).  The path shards by
independent streams: each rank owns its own pool, the only communication is one RCCL broadcast of
the serialized tap table before the timed region, and no collective sits on the data path
("scaling": "weak").

The JSON line also carries
  roofline      algorithmic HBM bytes (8 B per output sample) / kernel time measured with HIP
                events on the launch stream, vs 8 TB/s; `traffic` = HBM bytes per launch from the
                committed PMC passes of the same launch geometry (`traffic_source` says which file);
  cpu_baseline  the NumPy restatement of the reference's convolve_velvet_noise
                (oracle/vnd_oracle.py, single thread - NumPy slicing does not multithread) timed on
                this host on a bounded number of cfg2 signals; rank 0, N=1 only;
  secondary     the other BASELINE configs on one GPU (cfg3, cfg5, cfg4; kernel ms, GB/s, the
                binding limit named), each with one stream of the timed output checked against the
                C oracle (`parity_vs_oracle_of_peak`; the full-pool parity tests are tests/test_gpu_win.py);
  cfg4_strong   SURVEY 8(d) "Scaling (cfg4)": the 1024 x 1 s stereo batch split over the ranks with
                distributed.shard_range, timed from "table broadcast done, shards resident" to "all
                ranks done" (max over ranks); reported at every N, so N = 1/2/4/8 lines give the curve;
                at N = 1 `projection` times what ONE rank does at N = 1/2/4/8 (1024/512/256/128 streams
                per pass, rotating buffers) beside a plain device copy of the same shard;
  end_to_end    the synchronous host API (H2D + kernel + D2H) on pageable and pinned NumPy buffers,
                N=1 only - PCIe-inclusive, never part of `value`.

What is printed where.  stdout carries ONE short JSON line (a few KB): the contract's keys, `roofline`,
`cpu_baseline`, and in `config` one short numeric key per claim of README's table (`exact_frac`, `cfg3_frac`,
`cfg5_frac`, `proj_speedup`, ... - `COMPACT_KEYS` below says what each one is).  Everything else - launch
descriptions, notes, per-leg records - goes to `bench_detail.json` beside this file (and to `gpurun_out/` when
that directory exists) and, as one JSON document, to stderr BEFORE the line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks: the parent -
which never imports torch or touches a GPU - runs `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a CHILD process, relays rank 0's line and exits with the child's code.
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import sys
import time

import numpy as np

REPO = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

SAMPLE_RATE = 48000
SECONDS = 10
CHANNELS = 2
TAPS = 30
FIR_SECONDS = 0.03
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X FP32 vector peak (same guide): what bounds a table of more than ~80 taps per stereo frame
ALGO_BYTES_PER_SAMPLE = 8      # 4 B read + 4 B written per output channel-sample
MIN_TIMED_MS = 50.0            # the timed region of the headline must be at least this long


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--min-warmup-ms', type=float, default=150.0,
                    help='keep issuing untimed warm-up steps until this much wall time has passed: the GPU '
                         'ramps its clocks over the first ~40 ms of load (tools/sustain.py)')
    ap.add_argument('--pool', type=int, default=2048, help='distinct cfg2 signals per rank and step')
    ap.add_argument('--mode', choices=['exact', 'fma', 'fast'], default=os.environ.get('VND_BENCH_MODE', 'fast'))
    ap.add_argument('--cpu-seconds', type=float, default=10.0, help='budget of the CPU baseline leg')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-exact', action='store_true', help='skip the extra exact-mode timing')
    ap.add_argument('--no-secondary', action='store_true', help='skip cfg3/cfg5/cfg4, the audio leg, the next rows, end_to_end and the cfg4 strong-scaling leg')
    ap.add_argument('--no-strong', action='store_true', help='skip the cfg4 strong-scaling leg')
    ap.add_argument('--no-power', action='store_true', help='do not sample board power / shader clock')
    ap.add_argument('--detail', default=None, help='where the detail document goes (default: bench_detail.json beside bench.py)')
    ap.add_argument('--variant', type=int, default=-1, help='kernel variant override (tuning)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo for rehearsals)')
    return ap.parse_args()


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process tree and relay rank 0's
    line.  This process has not imported torch and makes no GPU call, before or after (the ranks are children, not
    an exec of this process): `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <the same arguments>`."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), str(pathlib.Path(__file__).resolve()), *argv]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL between processes needs it on this driver
    print('bench.py: starting', args.gpus, 'ranks:', ' '.join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for text in proc.stdout:                              # rank 0's JSON line goes to stdout as it is; anything else a rank printed, to stderr
        out = sys.stdout if text.lstrip().startswith('{"metric"') else sys.stderr
        out.write(text)
        out.flush()
    return proc.wait()


class PowerSampler:
    """Board power (W) and shader clock (MHz) of ONE card from its hwmon files, read by a thread every few
    milliseconds while the main thread launches kernels.  `window(t0, t1)` gives the median of the samples taken between
    two `time.perf_counter()` stamps.  The card is found by PCI address; a box that does not show the files reports null."""

    def __init__(self, torch, device_index, period_s=0.004):
        import glob
        import threading
        self.samples, self.stop_flag, self.thread, self.files = [], False, None, None
        self.cap_W = self.card = None
        self.period = period_s
        try:
            p = torch.cuda.get_device_properties(device_index)
            want = f'{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0'
            for card in glob.glob('/sys/class/drm/card*/device'):
                if os.path.realpath(card).endswith(want):
                    hw = glob.glob(card + '/hwmon/hwmon*')
                    if hw:
                        power = next((hw[0] + '/' + f for f in ('power1_input', 'power1_average') if os.path.exists(hw[0] + '/' + f)), None)
                        clock = hw[0] + '/freq1_input' if os.path.exists(hw[0] + '/freq1_input') else None
                        cap = hw[0] + '/power1_cap'
                        self.cap_W = int(open(cap).read()) / 1e6 if os.path.exists(cap) else None
                        if power:
                            self.files = (power, clock)
                            self.card = card.split('/')[-2]
        except Exception:
            self.files = None
        if self.files:
            self.thread = threading.Thread(target=self._run, daemon=True)
            self.thread.start()

    def _run(self):
        power, clock = self.files
        while not self.stop_flag:
            try:
                w = int(open(power).read()) / 1e6
                mhz = int(open(clock).read()) / 1e6 if clock else None
                self.samples.append((time.perf_counter(), w, mhz))
            except Exception:
                pass
            time.sleep(self.period)

    def window(self, t0, t1):
        rows = [(w, m) for t, w, m in self.samples if t0 <= t <= t1]
        if not rows:
            return None
        ws = sorted(w for w, _ in rows)
        ms = sorted(m for _, m in rows if m is not None)
        return {'power_W': round(ws[len(ws) // 2], 1), 'sclk_MHz': round(ms[len(ms) // 2]) if ms else None, 'samples': len(rows)}

    def close(self):
        self.stop_flag = True
        if self.thread:
            self.thread.join(timeout=1.0)


def cpu_baseline(budget_s: float) -> dict:
    """Reference CPU path (NumPy restatement, bit-identical to the reference) on
    whole cfg2 signals until ~budget_s of CPU time is spent."""
    from oracle import vnd_oracle as O
    fir = O.generate_velvet_noise(duration_seconds=FIR_SECONDS, num_impulses=TAPS, num_outs=CHANNELS,
                                  sample_rate_hz=SAMPLE_RATE, seed=1)
    x = np.random.default_rng(0).uniform(-1, 1, (SAMPLE_RATE * SECONDS, CHANNELS)).astype(np.float32)
    O.convolve_velvet_noise(x, fir)                      # warm-up
    reps, spent, best = 0, 0.0, float('inf')
    while spent < budget_s and reps < 1000:
        t = time.perf_counter()
        O.convolve_velvet_noise(x, fir)
        dt = time.perf_counter() - t
        spent += dt
        best = min(best, dt)
        reps += 1
    mean = spent / reps
    # the "fair CPU" line of SURVEY 8d: the C restatement (bit-identical too), one thread and many,
    # on a batch of 16 such signals (OpenMP over streams x tiles); ~2 s, reported beside the headline
    c_port = None
    try:
        from oracle import c_oracle
        offs, idx, w = O.fir_to_taps(fir)
        xb = np.ascontiguousarray(np.broadcast_to(x, (16,) + x.shape))
        threads = max(1, min(os.cpu_count() or 1, 64))
        c_port = {}
        for label, nthreads, batch in (('threads_1', 1, xb[:2]), (f'threads_{threads}', threads, xb)):
            c_oracle.convolve(batch, offs, idx, w, threads=nthreads)
            t = time.perf_counter()
            c_oracle.convolve(batch, offs, idx, w, threads=nthreads)
            c_port[label] = round(batch.size / (time.perf_counter() - t) / 1e6, 1)
    except Exception as exc:                            # the C oracle is optional test infrastructure
        c_port = {'unavailable': repr(exc)}
    return {'value': round(x.size / mean / 1e6, 3), 'unit': 'Msamples/s', 'cores': 1, 'kind': 'port',
            'c_restatement_Msamples_s': c_port,
            'sample': f'{reps} x one cfg2 signal ({x.shape[0]}x{x.shape[1]} f32, {TAPS} taps), '
                      f'NumPy restatement of convolve_velvet_noise, mean {mean * 1e3:.1f} ms, '
                      f'min {best * 1e3:.1f} ms, host cores available {os.cpu_count()}',
            'sample_short': f'{reps} x one cfg2 signal, NumPy restatement of convolve_velvet_noise, mean {mean * 1e3:.1f} ms, 1 thread',
            'best_value': round(x.size / best / 1e6, 3)}


def board_under(torch, power, fn, seconds=0.6):
    """The board's power (W) and shader clock (MHz) under `fn` launched back to back for `seconds` - NOT timed: the hwmon power
    figure is a slow average, so the reading is the median over the second half of the run.  None when the box shows no hwmon files."""
    if power is None or not power.files:
        return None
    s0, k = time.perf_counter(), 0
    while time.perf_counter() - s0 < seconds:
        fn(k); k += 1
        if k % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    s1 = time.perf_counter()
    return power.window(s0 + (s1 - s0) / 2, s1)


def device_rate(torch, table, shape, mode, *, min_ms=30.0, buffers=1, taps=None, exact_pool=False, power=None, xs_given=None, check_streams=None):
    """Kernel milliseconds per launch of `table` over a resident (batch, n, C) pool (HIP events on the
    launch stream, >= min_ms timed after a clock-settling warm-up).  `buffers` > 1 rotates distinct
    pools so that small shapes still stream from HBM, not from the 256 MiB Infinity Cache.
    `xs_given`: the pools to run on (default: uniform random in [-1, 1))."""
    batch, n, c = shape
    xs = xs_given if xs_given is not None else [torch.empty(shape, dtype=torch.float32, device='cuda').uniform_(-1.0, 1.0) for _ in range(buffers)]
    buffers = len(xs)
    ys = [torch.empty_like(xs[0]) for _ in range(buffers)]
    stream = torch.cuda.current_stream().cuda_stream

    def launch(i):
        table.convolve_device(xs[i % buffers].data_ptr(), ys[i % buffers].data_ptr(), batch, n, c, mode, stream)

    t0, i = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < 120.0:
        launch(i); i += 1
        if i % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    launch(0)
    torch.cuda.synchronize()
    iters = 4
    while True:
        e0.record()
        for k in range(iters):
            launch(k)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        if ms >= min_ms or iters >= 4096:
            break
        iters *= 2
    per = ms / iters
    bytes_per_launch = ALGO_BYTES_PER_SAMPLE * batch * n * c
    board = board_under(torch, power, launch)      # 0.6 s more of the same launches, not timed
    rec = {'kernel_ms': round(per, 4), 'board': board, 'achieved_GBs': round(bytes_per_launch / (per * 1e-3) / 1e9, 1),
           'frac_of_8TBs': round(bytes_per_launch / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           'Msamples_s': round(batch * n * c / (per * 1e-3) / 1e6, 1), 'launches_timed': iters,
           'launch': table.describe(batch, n, c, mode)}
    if taps is not None:
        # the arithmetic beside the bytes: one FMA per (frame, non-zero tap of its channel).  Above 157.3 / 8 = 19.7 flop per
        # byte (MI355X_MICROARCH.md: FP32 vector peak, v_pk_fma_f32 on every SIMD) the FP32 pipe bounds the launch, not HBM
        fmas = float(np.count_nonzero(np.asarray(taps[2]))) * batch * n
        rec['fp32_TFLOPs'] = round(2.0 * fmas / (per * 1e-3) / 1e12, 1)
        rec['frac_of_fp32_vector_peak'] = round(2.0 * fmas / (per * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4)
        rec['flop_per_byte'] = round(2.0 * fmas / bytes_per_launch, 1)
        rec['bound'] = 'fp32 vector' if 2.0 * fmas / bytes_per_launch > FP32_VECTOR_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS else 'hbm'
        # one stream of what the timed launches wrote, against the C oracle (the last stream: the highest addresses)
        last = (iters - 1) % buffers
        rec['parity_vs_oracle_of_peak'] = max(oracle_parity(xs[last][b], ys[last][b], taps, mode) for b in (check_streams or [batch - 1]))
        rec['parity_stream'] = check_streams or batch - 1
        if mode != 0 and exact_pool:
            # EVERY stream of the pool against the exact kernel (= the oracle, bit for bit: asserted on a stream of its own leg): the
            # worst stream's error as a fraction of the pool's output peak - the north star's 1e-6 is asserted on this number
            ye = torch.empty_like(ys[0])
            table.convolve_device(xs[0].data_ptr(), ye.data_ptr(), batch, n, c, 0, stream)
            table.convolve_device(xs[0].data_ptr(), ys[0].data_ptr(), batch, n, c, mode, stream)
            torch.cuda.synchronize()
            per_stream = (ys[0] - ye).abs().amax(dim=(1, 2)) / ye.abs().max()
            rec['parity_max_over_pool'] = float(per_stream.max())
            rec['parity_worst_stream'] = int(per_stream.argmax())
            assert rec['parity_max_over_pool'] <= 1e-6, f"fast mode off by {rec['parity_max_over_pool']:.2e} of peak on stream {rec['parity_worst_stream']}"
            del ye
    return rec


def oracle_parity(x_dev, y_dev, taps, mode, exact_mode=0):
    """max |y - oracle| / max |oracle| of one device stream (0.0 and an assert in the bit-exact mode)."""
    from oracle import c_oracle
    want = c_oracle.convolve(x_dev.cpu().numpy(), *taps, threads=8)
    got = y_dev.cpu().numpy()
    if mode == exact_mode:
        assert np.array_equal(got, want), 'exact mode differs from the oracle'
        return 0.0
    return float(np.max(np.abs(got.astype(np.float64) - want)) / np.max(np.abs(want)))


def secondary_configs(torch, vnd, _native, ctx, mode, power=None) -> dict:
    """The other BASELINE configs on one GPU (rates only; their parity is tests/test_gpu_*.py)."""
    from vndecorrelate_amd.taps import function_path_arrays
    out = {}

    def table_of(**kw):
        fir = vnd.generate_velvet_noise(**kw)
        a = function_path_arrays(fir)
        return _native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight), (a.tap_offsets, a.tap_index, a.tap_weight)

    specs = [
        ('cfg3', dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000,
                      log_distribution_strength=0.0, seed=1), (24, 2880000, 2), 1,
         '48 kHz stereo, 60 s, 128 taps (segmented decay, kappa 0); pool of 24',
         'FP32 vector pipe under the board power cap (32 flop per byte: past the 19.7 flop/B ridge); see `board` (W, MHz) of this leg and DESIGN.md 3.4'),
        ('cfg3_kappa1', dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000,
                             log_distribution_strength=1.0, seed=1), (24, 2880000, 2), 1,
         '48 kHz stereo, 60 s, 128 impulses log-distributed (kappa 1): the function path keeps 123 distinct taps per channel '
         '(last write wins on duplicate indices, SURVEY 8d); pool of 24',
         'as cfg3'),
        ('cfg5', dict(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1),
         (16, 960000, 8), 1, '96 kHz 8-channel, 10 s, 64 log-distributed taps; pool of 16',
         'LDS window reads (2.6 B per tap and output with 32-frame runs, bank conflicts 0.008 of the LDS cycles) and the FP32 pipe under the board '
         'power cap; one workgroup of 512 lanes per CU (a wave per channel; the 2720-frame halo of all 8 channels takes 158 KB of LDS), so a '
         'tile\'s exchange / store / refill phase is not covered by another workgroup (DESIGN.md 3.4, profiles/r05_cfg5_probe.txt)'),
        ('cfg4', dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1),
         (1024, 48000, 2), 2, '1024 independent 1 s stereo streams, 30 taps, one launch; 2 rotating pools',
         'board power cap, as cfg2'),
    ]
    for k, (name, kw, shape, buffers, what, limit) in enumerate(specs):
        try:
            torch.manual_seed(5000 + k)                 # the pools are seeded: a run's parity figures are a fact about the code, not about the draw
            t, taps = table_of(**kw)
            r = device_rate(torch, t, shape, mode, buffers=buffers, taps=taps, exact_pool=True, power=power)
            assert r['parity_vs_oracle_of_peak'] <= 1e-6, f"{name}: timed output off by {r['parity_vs_oracle_of_peak']:.2e} of peak"
            r.update({'workload': what, 'binding_limit': limit})
            if mode != vnd.MODE_EXACT and name != 'cfg3_kappa1':
                # the API's default mode on the same pool: bit-identical to the oracle (asserted inside), its own per-table kernel
                ex = device_rate(torch, t, shape, vnd.MODE_EXACT, buffers=buffers, taps=taps, power=power)
                r['exact_mode'] = {k: ex[k] for k in ('kernel_ms', 'achieved_GBs', 'frac_of_8TBs', 'board', 'launch')}
                r['exact_mode']['parity'] = 'bit-identical to the C oracle on the checked stream (asserted in this run)'
            out[name] = r
            t.close()
        except Exception as exc:                        # a secondary leg must never cost the headline
            out[name] = {'error': repr(exc)}
        torch.cuda.empty_cache()
    return out


def end_to_end(torch, vnd, mode) -> dict:
    """PCIe-inclusive rates of the synchronous host API: one cfg2 signal and the cfg4 batch."""
    out = {}
    fir = vnd.generate_velvet_noise(duration_seconds=FIR_SECONDS, num_impulses=TAPS, num_outs=CHANNELS,
                                    sample_rate_hz=SAMPLE_RATE, seed=1)
    rng = np.random.default_rng(3)
    for name, shape in (('cfg2_one_signal', (1, 480000, 2)), ('cfg4_batch', (1024, 48000, 2))):
        x = rng.uniform(-1, 1, shape).astype(np.float32)
        rec = {}
        for kind in ('pageable', 'pinned'):
            xin = x
            if kind == 'pinned':
                xin = torch.from_numpy(x).pin_memory().numpy()
            call = (lambda a: vnd.convolve_velvet_noise_batched(a, fir, mode=mode)) if shape[0] > 1 else \
                   (lambda a: vnd.convolve_velvet_noise(a[0], fir, mode=mode))
            call(xin)
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 0.5 and reps < 200:
                call(xin)
                reps += 1
            dt = (time.perf_counter() - t0) / reps
            rec[kind] = {'ms_per_call': round(dt * 1e3, 3), 'Msamples_s': round(x.size / dt / 1e6, 1),
                         'GBs_in_plus_out': round(2 * x.nbytes / dt / 1e9, 2)}
        out[name] = rec
    out['note'] = ('host pointers in, host pointers out; the result array is allocated by the call as the reference does (from a '
                   'page-locked pool); never part of `value`.  pageable: H2D + kernel + D2H inside the call, pipelined over groups of '
                   'streams (one group: the kernel writes the result in place).  pinned: no staging, the kernel reads and writes the '
                   'page-locked buffers in place across PCIe (profiles/r03_host_in_place.txt).  The pageable batch stays ahead: its '
                   'upload is staged by the CPU beside the SDMA download, page-locked copies share the SDMA path (57 GB/s in total)')
    return out


def audio_leg(torch, vnd, _native, ctx, pool, power=None) -> dict:
    """The path on AUDIO instead of uniform random floats (SURVEY 8d's synthetic input is the worst case for the board's power cap,
    DESIGN.md 3.4): BASELINE configs[0]'s material - the 1 s stereo excerpt of the reference's viola recording kept in
    tests/golden/golden.npz (`viola_excerpt_in`, 44.1 kHz; /root/reference/tests/test_example.py:19-49 runs the whole file) - tiled to
    the headline pool's shape (as many streams, as many frames), every stream starting at another frame of the excerpt; the table
    is that test's (20 ms, 30 impulses, seed 1).  Fast and exact kernels, three streams of each against the C oracle."""
    from vndecorrelate_amd.taps import function_path_arrays
    z = np.load(REPO / 'tests' / 'golden' / 'golden.npz')
    excerpt = torch.from_numpy(np.ascontiguousarray(z['viola_excerpt_in'])).cuda()          # (44100, 2) float32
    fs = int(excerpt.shape[0])
    # the headline pool's shape exactly - as many streams of as many frames (10.9 s at 44.1 kHz) - so that the data and the table are all that differ
    streams, n = max(3, pool), SAMPLE_RATE * SECONDS
    reps = -(-n // fs)
    base = excerpt.repeat(reps + 1, 1)
    x = torch.empty((streams, n, 2), dtype=torch.float32, device='cuda')
    for b in range(streams):
        off = (b * 1009) % fs
        x[b].copy_(base[off:off + n])
    arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.02, num_impulses=TAPS, num_outs=2, sample_rate_hz=fs, seed=1))
    taps = (arr.tap_offsets, arr.tap_index, arr.tap_weight)
    table = _native.TapTable.create(ctx, *taps)
    out = {'what': f'{streams} streams x {n} frames of the viola excerpt, tiled (44.1 kHz stereo, peak {float(excerpt.abs().max()):.3f}, rms {float(excerpt.square().mean().sqrt()):.4f}), '
                   f'one launch; table: 20 ms / 30 impulses / seed 1 at 44.1 kHz (tests/test_example.py:19-34 of the reference)',
           'streams': streams, 'frames': n}
    checked = sorted({0, streams // 2, streams - 1})
    for label, mode in (('fast', vnd.MODE_FAST), ('exact', vnd.MODE_EXACT)):
        r = device_rate(torch, table, (streams, n, 2), mode, taps=taps, power=power, xs_given=[x], check_streams=checked)
        assert r['parity_vs_oracle_of_peak'] <= 1e-6, f"audio {label}: timed output off by {r['parity_vs_oracle_of_peak']:.2e} of peak"
        out[label] = {k: r[k] for k in ('kernel_ms', 'achieved_GBs', 'frac_of_8TBs', 'board', 'parity_vs_oracle_of_peak', 'parity_stream', 'launch')}
    table.close()
    del x, base
    torch.cuda.empty_cache()
    return out


def next_rows(torch, vnd, _native, power=None) -> dict:
    """SURVEY 8(f) rows on one GPU, host to host or device resident as stated (rates only; parity is tests/)."""
    import contextlib
    import io
    out = {}
    rng = np.random.default_rng(5)
    torch.manual_seed(6000)
    n = SAMPLE_RATE * SECONDS

    def best_of(fn, reps):
        fn()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3

    # f1: VelvetNoise.decorrelate (side-channel encode + exact RMS normaliser), bit-identical mode
    try:
        vn = vnd.VelvetNoise(sample_rate_hz=SAMPLE_RATE, seed=1)
        table = vn._device_table()
        st = torch.cuda.current_stream().cuda_stream
        ws_bytes = _native.decorrelate_workspace_bytes(1, n, 2)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
        rec = {}
        music = np.round(8000 * (np.sin(np.arange(n) * 0.01)[:, None] * np.array([1.0, 0.7]) + 0.3 * rng.standard_normal((n, 2))))
        for name, sig in (('uniform_float', rng.uniform(-1, 1, (n, 2)).astype(np.float32)),
                          ('int16_music', (music / 32768.0).astype(np.float32))):
            x = torch.from_numpy(sig[None]).cuda()
            y = torch.empty_like(x)

            def run():
                table.decorrelate_device(x.data_ptr(), y.data_ptr(), 1, n, 2, mode=vnd.MODE_EXACT, ms_encode=True, width=None,
                                         normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
            for _ in range(10):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(100):
                run()
            e1.record()
            torch.cuda.synchronize()
            rec[name + '_device_resident_ms'] = round(e0.elapsed_time(e1) / 100, 4)
        host = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
        rec['uniform_float_host_to_host_ms'] = round(best_of(lambda: vn.decorrelate(host), 10), 3)
        rec['what'] = ('VelvetNoise.decorrelate of one 10 s 48 kHz stereo signal, VND_MODE_EXACT: bit-identical to the reference, '
                       'RMS normaliser in NumPy order (block-parallel sums, ties in integers)')
        out['f1_decorrelate_exact'] = rec
    except Exception as exc:
        out['f1_decorrelate_exact'] = {'error': repr(exc)}
    # f1 at pool scale: the whole stage (side-channel encode + RMS normaliser) over a resident pool of cfg2 signals in one call
    try:
        from oracle import vnd_oracle as O
        vn = vnd.VelvetNoise(sample_rate_hz=SAMPLE_RATE, seed=1)
        table = vn._device_table()
        st = torch.cuda.current_stream().cuda_stream
        rec = {}
        for pool in (256, 128):
            x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
            y = torch.empty_like(x)
            ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
            for label, mode, normalize, bytes_per_sample in (('exact', vnd.MODE_EXACT, 1, 24), ('fast_fused', vnd.MODE_FAST, 1, 16)):
                table.prepare(pool, n, 2, mode)

                def run():
                    table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, ms_encode=True, width=None,
                                             normalize=normalize, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
                for _ in range(5):
                    run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                reps = 20
                for _ in range(reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                board = board_under(torch, power, lambda k: run())
                moved = bytes_per_sample * pool * n * 2
                worst = 0.0
                for b in sorted({0, pool // 2, pool - 1}):
                    want = O.decorrelate(x[b].cpu().numpy(), sample_rate_hz=SAMPLE_RATE, seed=1)
                    got = y[b].cpu().numpy()
                    if mode == vnd.MODE_EXACT:
                        assert np.array_equal(got, want), f'f1_pool: exact stage differs from the oracle (stream {b})'
                    else:
                        worst = max(worst, float(np.max(np.abs(got.astype(np.float64) - want)) / np.max(np.abs(want))))
                assert worst <= 5e-4, f'f1_pool: fused fast stage off by {worst:.2e} of peak'
                rec[f'pool{pool}_{label}'] = {
                    'ms_per_call': round(ms, 4), 'bytes_per_sample_moved': bytes_per_sample, 'achieved_GBs': round(moved / (ms * 1e-3) / 1e9, 1),
                    'frac_of_8TBs': round(moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'Msamples_s': round(pool * n * 2 / (ms * 1e-3) / 1e6, 1), 'board': board,
                    'parity': ('bit-identical to the oracle\'s whole stage on streams 0 / middle / last (asserted in this run)' if mode == vnd.MODE_EXACT
                               else f'{worst:.1e} of peak from the oracle\'s stage (its RMS is the correctly rounded one: bar 5e-4)'),
                    'launch': table.describe(pool, n, 2, mode)[:200]}
            del x, y, ws
            torch.cuda.empty_cache()
        rec['what'] = ('vnd_decorrelate_f32_dev (MS encode + RMS normalise) over a resident pool of 10 s stereo signals, one call.  exact: convolution '
                       'with the pointwise steps in its store phase (8 B/sample) + NumPy-order sums (x and y read once: 8) + scale pass (8) = 24 B/sample; '
                       'pools below 256 streams take the block-parallel sums, whose per-block predictions the convolution leaves on its way; '
                       'fast_fused: the store phase leaves per-block sums of squares (a lane\'s 32 squares in float32, the wave and everything after in float64) + scale pass = 16 B/sample')
        out['f1_pool'] = rec
    except Exception as exc:
        out['f1_pool'] = {'error': repr(exc)}
    # f1 on MONO signals - the reference's canonical use (decorrelation.py:428-442: mono_to_stereo, convolve, side-channel encode, normalise):
    # the stage over a resident pool of 128 x 10 s mono signals, stereo out.  Bytes per FRAME the stage must move: convolution 4 in + 8 out,
    # NumPy-order sums 4 + 8 (exact only), scale pass 8 + 8: 40 exact, 28 fused fast
    try:
        from oracle import vnd_oracle as O
        vn = vnd.VelvetNoise(sample_rate_hz=SAMPLE_RATE, seed=1)
        table = vn._device_table()
        st = torch.cuda.current_stream().cuda_stream
        pool = 128
        x = torch.empty((pool, n, 1), dtype=torch.float32, device='cuda').uniform_(-1, 1)
        y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
        ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
        rec = {}
        for label, mode, bytes_per_frame in (('exact', vnd.MODE_EXACT, 40), ('fast_fused', vnd.MODE_FAST, 28)):
            table.prepare(pool, n, 1, mode)

            def run():
                table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, 1, mode=mode, ms_encode=True, width=None,
                                         normalize=1, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            board = board_under(torch, power, lambda k: run())
            worst = 0.0
            for b in (0, pool - 1):
                want = O.decorrelate(x[b, :, 0].cpu().numpy(), sample_rate_hz=SAMPLE_RATE, seed=1)
                got = y[b].cpu().numpy()
                if mode == vnd.MODE_EXACT:
                    assert np.array_equal(got, want), f'f1_mono: exact stage differs from the oracle (stream {b})'
                else:
                    worst = max(worst, float(np.max(np.abs(got.astype(np.float64) - want)) / np.max(np.abs(want))))
            assert worst <= 5e-4, f'f1_mono: fused fast stage off by {worst:.2e} of peak'
            rec[label] = {'ms_per_call': round(ms, 4), 'bytes_per_frame_moved': bytes_per_frame, 'achieved_GBs': round(bytes_per_frame * pool * n / (ms * 1e-3) / 1e9, 1),
                          'frac_of_8TBs': round(bytes_per_frame * pool * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'Gframes_s': round(pool * n / (ms * 1e-3) / 1e9, 1),
                          'parity_vs_oracle_stage_of_peak': worst, 'board': board}
        rec['what'] = ('vnd_decorrelate_fanout_f32_dev over 128 x 10 s MONO signals, stereo out, one call: exact - bit-identical to the reference\'s whole stage on the '
                       'checked streams (asserted) - and fused fast; the exact stage\'s convolution in the plain window form with the block sums in its store phase')
        out['f1_mono'] = rec
        del x, y, ws
        torch.cuda.empty_cache()
    except Exception as exc:
        out['f1_mono'] = {'error': repr(exc)}
    # f1 on cfg5's shape: VelvetNoise(num_outs=8, mode='LR', filtered_channels=0..7).decorrelate - how BASELINE's 8-channel config maps onto
    # the class API (decorrelation.py:417-442 with :433-440 reduced to the per-channel RMS normaliser) - over a resident pool
    try:
        from oracle import vnd_oracle as O
        fs8, n8, pool8 = 96000, 960000, 16
        kw8 = dict(sample_rate_hz=fs8, num_outs=8, num_impulses=64, filtered_channels=tuple(range(8)), mode='LR', seed=1)
        vn8 = vnd.VelvetNoise(**kw8)
        table = vn8._device_table()
        st = torch.cuda.current_stream().cuda_stream
        x = torch.empty((pool8, n8, 8), dtype=torch.float32, device='cuda').uniform_(-1, 1)
        y = torch.empty_like(x)
        ws_bytes = _native.decorrelate_workspace_bytes(pool8, n8, 8)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
        rec = {}
        for label, mode, bytes_per_sample in (('fast_fused', vnd.MODE_FAST, 16), ('exact', vnd.MODE_EXACT, 24)):
            table.prepare(pool8, n8, 8, mode)

            def run():
                table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool8, n8, 8, mode=mode, ms_encode=False, width=None,
                                         normalize=1, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            reps = 20
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            board = board_under(torch, power, lambda k: run())
            b = pool8 - 1
            want = O.decorrelate(x[b].cpu().numpy(), **kw8)
            got = y[b].cpu().numpy()
            if mode == vnd.MODE_EXACT:
                assert np.array_equal(got, want), 'f1_c8: exact stage differs from the oracle'
                worst = 0.0
            else:
                worst = float(np.max(np.abs(got.astype(np.float64) - want)) / np.max(np.abs(want)))
                assert worst <= 5e-4, f'f1_c8: fused fast stage off by {worst:.2e} of peak'
            moved = bytes_per_sample * pool8 * n8 * 8
            rec[label] = {'ms_per_call': round(ms, 4), 'bytes_per_sample_moved': bytes_per_sample, 'achieved_GBs': round(moved / (ms * 1e-3) / 1e9, 1),
                          'frac_of_8TBs': round(moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'Msamples_s': round(pool8 * n8 * 8 / (ms * 1e-3) / 1e6, 1),
                          'parity_vs_oracle_stage_of_peak': worst, 'board': board, 'launch': table.describe(pool8, n8, 8, mode)[:200]}
        rec['what'] = ('vnd_decorrelate_f32_dev of 16 x 10 s 96 kHz 8-channel signals (LR mode: the per-channel RMS normaliser alone), class-path table of 64 '
                       'taps per channel; fractions by the bytes the stage must move: 16 B per sample fused (convolution 8 + scale pass 8), 24 exact (+ 8: the '
                       'reference-order sums read x and y)')
        out['f1_c8'] = rec
        del x, y, ws
        torch.cuda.empty_cache()
    except Exception as exc:
        out['f1_c8'] = {'error': repr(exc)}
    # fan-out: mono in, stereo out (decorrelation.py:431-432) on the cfg2 shape, device resident, throughput mode
    try:
        from vndecorrelate_amd.taps import function_path_arrays
        ctx = _native.default_context()
        arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=FIR_SECONDS, num_impulses=TAPS, num_outs=2,
                                                             sample_rate_hz=SAMPLE_RATE, seed=1))
        table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
        pool = 128
        x = torch.empty((pool, n, 1), dtype=torch.float32, device='cuda').uniform_(-1, 1)
        y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
        st = torch.cuda.current_stream().cuda_stream

        def run(mode=vnd.MODE_FAST):
            table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 1, mode, st)
        rec = {}
        for label, mode in (('fast', vnd.MODE_FAST), ('exact', vnd.MODE_EXACT)):
            t0, i = time.perf_counter(), 0
            while (time.perf_counter() - t0) * 1e3 < 100.0:        # the clocks settle over the first ~40 ms of load
                run(mode); i += 1
                if i % 8 == 0:
                    torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(200):
                run(mode)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 200
            board = board_under(torch, power, lambda k: run(mode))
            # stream 0 and the last one of what the timed launches wrote, against the C oracle on the replicated input
            worst = 0.0
            for b in (0, pool - 1):
                xs2 = np.ascontiguousarray(np.repeat(x[b].cpu().numpy(), 2, axis=1))
                worst = max(worst, oracle_parity(torch.from_numpy(xs2), y[b], (arr.tap_offsets, arr.tap_index, arr.tap_weight), mode))
            assert worst <= 1e-6, f'mono_to_stereo {label}: off by {worst:.2e} of peak'
            rec[label] = {'kernel_ms': round(ms, 4), 'output_Msamples_s': round(pool * n * 2 / ms / 1e3, 1),
                          'achieved_GBs_12B_per_frame': round(12e-6 * pool * n / ms, 1), 'frac_of_8TBs': round(12e-9 * pool * n / ms / 8.0, 4),
                          'parity_vs_oracle_of_peak': worst, 'board': board, 'launch': table.describe(pool, n, 1, mode)}
        out['mono_to_stereo_fast'] = dict(rec['fast'], what='128 x 10 s mono signals in, stereo out, one launch (12 algorithmic bytes per frame: 4 read, 8 written)',
                                          exact_mode=rec['exact'])
        table.close()
        del x, y
    except Exception as exc:
        out['mono_to_stereo_fast'] = {'error': repr(exc)}
    # f3: the optimiser's candidate scan, host to host
    try:
        import vndecorrelate_amd.optimization as opt
        fs = 44100
        sig = rng.uniform(-1, 1, (int(fs * 5.7), 2)).astype(np.float32)
        cands = [vnd.VelvetNoise(sample_rate_hz=fs, duration_seconds=0.03, num_impulses=30, log_distribution_strength=k,
                                 normalizer=None, filtered_channels=(0,), mode='LR', seed=1) for k in np.linspace(0, 1, 400)]
        kw = dict(angle_limit=float(np.pi / 4), lambda_mean=5.0, lambda_skew=2.0, lambda_correlation=15.0, lambda_penalty=1e3)
        with contextlib.redirect_stdout(io.StringIO()):
            ms = best_of(lambda: opt.grid_scan(sig, cands, **kw), 5)
        out['f3_grid_scan'] = {'ms_host_to_host': round(ms, 3), 'what': 'grid_scan of 400 built candidates x 5.7 s of 44.1 kHz stereo'}
    except Exception as exc:
        out['f3_grid_scan'] = {'error': repr(exc)}
    # f4: velvet noise -> Haas chain, device resident, host to host
    try:
        host = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
        chain = (vnd.SignalChain(sample_rate_hz=SAMPLE_RATE, device_resident=True).velvet_noise(seed=1)
                 .haas_effect(delay_time_seconds=0.02, delayed_channel=1, mode='LR'))
        out['f4_resident_chain'] = {'ms_host_to_host': round(best_of(lambda: chain(host), 5), 3),
                                    'what': 'SignalChain(velvet noise -> Haas), 10 s stereo in, float64 (n + delay, 2) out, exact mode'}
    except Exception as exc:
        out['f4_resident_chain'] = {'error': repr(exc)}
    return out


def shard_projection(torch, table, n, mode, stream) -> dict:
    """What ONE rank does per pass of the cfg4 strong-scaling leg at N = 1, 2, 4, 8 (1024 / N streams, rotating
    buffers so that a 49 MB shard still streams from HBM), beside a plain device copy of the same shard: the
    one-GPU projection of the 8-GPU speed-up (no 8-GPU node is available to the builder).  Every size is timed WARM - the
    clocks settled on that size's own passes - as the median of 7 loops (min beside it); N = 1, the denominator of every
    speed-up, is measured last, behind 0.3 s of the other sizes' passes."""
    out = {}
    rows = {}
    for ranks in (8, 4, 2, 1):
        mine = 1024 // ranks
        buffers = max(1, int(np.ceil(600e6 / (mine * n * CHANNELS * 4 * 2))))
        xs = [torch.empty((mine, n, CHANNELS), dtype=torch.float32, device='cuda').uniform_(-1, 1) for _ in range(buffers)]
        ys = [torch.empty_like(xs[0]) for _ in range(buffers)]
        table.prepare(mine, n, CHANNELS, mode)             # a rank builds its shard's kernel once, before the passes

        def timed_loops(fn, reps, loops=7):
            t0, i = time.perf_counter(), 0
            while (time.perf_counter() - t0) * 1e3 < 60.0:      # this size's own warm-up: 60 ms of its passes
                fn(i); i += 1
                if i % 16 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            times = []
            for _ in range(loops):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    fn(i)
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) / reps)
            times.sort()
            return times[0], times[len(times) // 2]

        reps = 100 * ranks
        ms_min, ms_med = timed_loops(lambda i: table.convolve_device(xs[i % buffers].data_ptr(), ys[i % buffers].data_ptr(), mine, n, CHANNELS, mode, stream), reps)
        _, copy_ms = timed_loops(lambda i: ys[i % buffers].copy_(xs[i % buffers]), reps, loops=3)
        rows[ranks] = {'streams_per_rank': mine, 'us_per_pass': round(ms_med * 1e3, 2), 'us_per_pass_min': round(ms_min * 1e3, 2),
                       'device_copy_us': round(copy_ms * 1e3, 2), 'launch': table.describe(mine, n, CHANNELS, mode)[:160]}
        del xs, ys
        torch.cuda.empty_cache()
    base = rows[1]['us_per_pass']
    for ranks in (1, 2, 4, 8):
        rows[ranks]['speedup_vs_N1'] = round(base / rows[ranks]['us_per_pass'], 2)
        out[f'N={ranks}'] = rows[ranks]
    out['note'] = ('kernel time between HIP events on the launch stream, back-to-back passes over rotating buffers, median (and min) of 7 loops '
                   'per size, each size warmed on its own passes, N = 1 measured last; the 8-GPU job adds one barrier per timed region, not per pass')
    return out


def cfg4_strong(torch, dist, vnd, _native, ctx, table_image, mode, world, rank, device, backend, taps=None) -> dict:
    """SURVEY 8(d) scaling leg: 1024 x 1 s stereo streams, contiguous shards, no data-path collective."""
    from vndecorrelate_amd.distributed import shard_range
    table = _native.TapTable.from_bytes(ctx, table_image)
    streams, n = 1024, 48000
    _, mine = shard_range(streams, world, rank)            # (first stream, count) of this rank's contiguous block
    # rotate enough distinct shard buffers that a step streams from HBM (a 1/8 shard is 49 MB)
    buffers = max(1, int(np.ceil(600e6 / max(1, mine * n * CHANNELS * 4 * 2))))
    gen = torch.Generator(device=device)
    gen.manual_seed(99 + rank)
    xs = [torch.empty((mine, n, CHANNELS), dtype=torch.float32, device=device).uniform_(-1, 1, generator=gen)
          for _ in range(buffers)]
    ys = [torch.empty_like(xs[0]) for _ in range(buffers)]
    stream = torch.cuda.current_stream().cuda_stream
    table.prepare(mine, n, CHANNELS, mode)                 # small shards never stall for a hipRTC build inside a pass

    def step(i):
        table.convolve_device(xs[i % buffers].data_ptr(), ys[i % buffers].data_ptr(), mine, n, CHANNELS, mode, stream)

    t0, i = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < 100.0:
        step(i); i += 1
        if i % 8 == 0:
            torch.cuda.synchronize()
    steps = 200 if world == 1 else 200 * min(world, 4)     # (a shard's pass is short - 24 us at N = 8 - and the region ends in a barrier: 5-20 ms of passes per region either way)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per = elapsed / steps
    total = streams * n * CHANNELS
    parity = None
    if rank == 0 and taps is not None:
        last = (steps - 1) % buffers
        parity = oracle_parity(xs[last][mine - 1], ys[last][mine - 1], taps, mode)
        assert parity <= 1e-6, f'cfg4_strong: timed output off by {parity:.2e} of peak'
    projection = None
    if world == 1:
        del xs, ys
        torch.cuda.empty_cache()
        projection = shard_projection(torch, table, n, mode, stream)
    return {'streams': streams, 'frames_per_stream': n, 'ranks': world, 'streams_on_rank0': mine, 'steps': steps,
            'parity_vs_oracle_of_peak': parity, 'projection': projection,
            'ms_per_pass_max_over_ranks': round(per * 1e3, 4), 'Msamples_s': round(total / per / 1e6, 1),
            'aggregate_GBs': round(ALGO_BYTES_PER_SAMPLE * total / per / 1e9, 1),
            'timed_from': 'table broadcast done, shards resident; barrier + sync both sides, max over ranks',
            'launch_rank0': table.describe(mine, n, CHANNELS, mode)}


def dig(record, *path, default=None):
    """record[path[0]][path[1]]... or `default` when a leg is missing or failed."""
    for key in path:
        if not isinstance(record, dict) or key not in record:
            return default
        record = record[key]
    return record


# what the short keys of `config` are (the detail document carries the full records they come from)
COMPACT_KEYS = {
    'parity': 'headline: worst of the checked streams of the timed output vs the C oracle, fraction of the output peak',
    'cfg4_strong': 'this run\'s ranks on the 1024 x 1 s batch: max-over-ranks ms per pass, whole-job Msamples/s - the strong-scaling figure',
    'proj_N1_us / proj_N8_us / proj_copy_N8_us / proj_speedup': 'one-GPU projection of the cfg4 strong cut: a rank\'s pass at N = 1 and N = 8, a copy of the N = 8 shard, N1 / N8',
    'exact_ms / exact_frac': 'VND_MODE_EXACT (bit-identical, the API default) on the headline pool: kernel ms, fraction of 8 TB/s',
    'class_exact_frac': 'the class path\'s table (VelvetNoise.convolve) in exact mode on the same pool',
    'cfgK_ms / cfgK_frac / cfgK_exact_frac': 'BASELINE configs[K-1] on one GPU: kernel ms and fraction of 8 TB/s by 8 B per sample, fast and exact',
    'cfg3_fp32_frac': 'cfg3 against the FP32 vector peak (157.3 TFLOP/s): its binding limit',
    'cfg3k1_frac': 'cfg3 with kappa 1 (123 distinct taps per channel)',
    'cfgK_parity': 'fast mode: the worst stream of that timed pool against the exact kernel (= the reference), of the pool\'s peak',
    'worst_parity_over_pools': 'the largest of them',
    'audio_frac / audio_exact_frac': 'the headline pool\'s bytes as 10 s streams of the reference\'s viola recording (44.1 kHz stereo, its 20 ms table) instead of random floats',
    'f1_P_exact_frac / f1_P_fast_frac': 'the whole decorrelate stage over a pool of P 10 s stereo signals, by its own 24 / 16 B per sample',
    'f1_mono_exact_frac / f1_mono_fast_frac / f1_mono_exact_ms': 'VelvetNoise.decorrelate of 128 x 10 s MONO signals (stereo out), by 40 / 28 B per frame',
    'f1_c8_frac': 'VelvetNoise.decorrelate of 8-channel 96 kHz signals (LR mode, RMS normaliser), fused fast stage, by 16 B per sample',
    'm2s_frac / m2s_exact_frac': 'mono in, stereo out, 12 B per frame',
    'scan_ms / chain_ms': 'f3 grid scan of 400 candidates, f4 resident chain: host to host',
    'e2e_cfg2_ms / e2e_cfg4_ms': 'host API, pageable buffers, PCIe inclusive: one cfg2 signal, the cfg4 batch',
    'single_us': 'one cfg2 signal per launch, device resident: launch to synchronise, median (fast mode)',
    '<leg>_W / <leg>_MHz': 'board power and shader clock under that leg\'s kernel (hwmon, median of the second half of 0.6 s of launches)',
}

# the order of `config`: the driver's parsed record keeps the FIRST 24 keys - what the north star asks about comes first
COMPACT_ORDER = ('workload', 'pool', 'frames', 'channels', 'mode', 'parity', 'world_size', 'ranks_seen', 'backend', 'cfg4_strong', 'proj_speedup', 'proj_N1_us',
                 'proj_N8_us', 'proj_copy_N8_us', 'exact_frac', 'class_exact_frac', 'cfg3_frac', 'cfg3_fp32_frac', 'cfg5_frac', 'cfg4_frac', 'audio_frac',
                 'worst_parity_over_pools', 'cfg3_exact_frac', 'cfg5_exact_frac')


def compact(d: dict) -> dict:
    """The stdout line: the contract's keys plus one short numeric key per claim, all under `config`, `roofline` and
    `cpu_baseline` (which the driver's record keeps whole), no string above 120 characters."""
    cfg, roof, sec, nxt = d['config'], d['roofline'], d.get('secondary') or {}, d.get('next_rows') or {}
    c = {'workload': f"cfg2: 48 kHz stereo f32, 10 s, 30 taps / 30 ms, seed 1; pool of {cfg['pool_signals_per_gpu']} signals per GPU, one launch per step",
         'pool': cfg['pool_signals_per_gpu'], 'frames': cfg['frames'], 'channels': cfg['channels'], 'mode': cfg['arithmetic'],
         'parity': float(f"{cfg['parity_vs_oracle_of_peak']:.3g}"), 'timed_ms': d.get('timed_ms'), 'world_size': cfg['world_size'], 'backend': cfg['backend'],
         'ranks_seen': cfg['ranks_seen_by_all_reduce'],
         'exact_ms': dig(d, 'exact_mode', 'kernel_ms'), 'exact_frac': dig(d, 'exact_mode', 'frac_of_8TBs'),
         'class_exact_frac': dig(d, 'exact_mode', 'class_path_table', 'frac_of_8TBs')}

    def board(prefix, rec):
        if isinstance(rec, dict) and rec.get('power_W') is not None:
            c[prefix + '_W'], c[prefix + '_MHz'] = round(rec['power_W']), rec.get('sclk_MHz')
    board('exact', dig(d, 'exact_mode', 'board'))
    board('class_exact', dig(d, 'exact_mode', 'class_path_table', 'board'))
    worst = None
    for name, short in (('cfg3', 'cfg3'), ('cfg3_kappa1', 'cfg3k1'), ('cfg5', 'cfg5'), ('cfg4', 'cfg4')):
        rec = sec.get(name) or {}
        if 'error' in rec:
            c[short + '_error'] = rec['error'][:100]
            continue
        if not rec:
            continue
        if short != 'cfg3k1':
            c[short + '_ms'] = rec.get('kernel_ms')
        c[short + '_frac'] = rec.get('frac_of_8TBs')
        if short == 'cfg3':
            c['cfg3_fp32_frac'] = rec.get('frac_of_fp32_vector_peak')
        board(short, rec.get('board'))
        if 'exact_mode' in rec:
            c[short + '_exact_frac'] = dig(rec, 'exact_mode', 'frac_of_8TBs')
            board(short + '_exact', dig(rec, 'exact_mode', 'board'))
        if rec.get('parity_max_over_pool') is not None:
            worst = max(worst or 0.0, rec['parity_max_over_pool'])
            c[short + '_parity'] = float(f"{rec['parity_max_over_pool']:.3g}")
    if worst is not None:
        c['worst_parity_over_pools'] = float(f'{worst:.3g}')
    audio = d.get('audio') or {}
    if 'error' in audio:
        c['audio_error'] = audio['error'][:100]
    for label, short in (('fast', 'audio'), ('exact', 'audio_exact')):
        if dig(audio, label, 'frac_of_8TBs') is not None:
            c[short + '_frac'] = audio[label]['frac_of_8TBs']
            board(short, audio[label].get('board'))
    for pool in (256, 128):
        for label, short in (('exact', 'exact'), ('fast_fused', 'fast')):
            v = dig(nxt, 'f1_pool', f'pool{pool}_{label}', 'frac_of_8TBs')
            if v is not None:
                c[f'f1_{pool}_{short}_frac'] = v
                board(f'f1_{pool}_{short}', dig(nxt, 'f1_pool', f'pool{pool}_{label}', 'board'))
    for key, path in (('f1_mono_exact_frac', ('f1_mono', 'exact', 'frac_of_8TBs')), ('f1_mono_fast_frac', ('f1_mono', 'fast_fused', 'frac_of_8TBs')),
                      ('f1_mono_exact_ms', ('f1_mono', 'exact', 'ms_per_call')),
                      ('f1_c8_frac', ('f1_c8', 'fast_fused', 'frac_of_8TBs')), ('f1_c8_exact_frac', ('f1_c8', 'exact', 'frac_of_8TBs')),
                      ('m2s_frac', ('mono_to_stereo_fast', 'frac_of_8TBs')), ('m2s_exact_frac', ('mono_to_stereo_fast', 'exact_mode', 'frac_of_8TBs')),
                      ('scan_ms', ('f3_grid_scan', 'ms_host_to_host')), ('chain_ms', ('f4_resident_chain', 'ms_host_to_host'))):
        v = dig(nxt, *path)
        if v is not None:
            c[key] = v
    board('f1_mono_exact', dig(nxt, 'f1_mono', 'exact', 'board'))
    board('f1_c8', dig(nxt, 'f1_c8', 'fast_fused', 'board'))
    board('f1_c8_exact', dig(nxt, 'f1_c8', 'exact', 'board'))
    board('m2s', dig(nxt, 'mono_to_stereo_fast', 'board'))
    board('m2s_exact', dig(nxt, 'mono_to_stereo_fast', 'exact_mode', 'board'))
    for key, path in (('e2e_cfg2_ms', ('cfg2_one_signal', 'pageable', 'ms_per_call')), ('e2e_cfg4_ms', ('cfg4_batch', 'pageable', 'ms_per_call'))):
        v = dig(d, 'end_to_end', *path)
        if v is not None:
            c[key] = v
    proj = dig(d, 'cfg4_strong', 'projection')
    if proj:
        c.update({'proj_N1_us': dig(proj, 'N=1', 'us_per_pass'), 'proj_N8_us': dig(proj, 'N=8', 'us_per_pass'),
                  'proj_copy_N8_us': dig(proj, 'N=8', 'device_copy_us'), 'proj_speedup': dig(proj, 'N=8', 'speedup_vs_N1')})
    v = dig(cfg, 'single_launch_us', 'fast', 'launch_to_sync_us_median')
    if v is not None:
        c['single_us'] = v
    if d.get('cfg4_strong'):
        st = d['cfg4_strong']
        c['cfg4_strong'] = {k: st.get(k) for k in ('ranks', 'streams_on_rank0', 'ms_per_pass_max_over_ranks', 'Msamples_s')}
    c['detail'] = 'bench_detail.json; stderr'
    # what matters most first (COMPACT_ORDER), the rest as gathered, the *_W / *_MHz readings last
    rest = [k for k in c if k not in COMPACT_ORDER]
    ordered = [k for k in COMPACT_ORDER if k in c] + [k for k in rest if not k.endswith(('_W', '_MHz'))] + [k for k in rest if k.endswith(('_W', '_MHz'))]
    c = {k: c[k] for k in ordered}
    line = {k: d[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                              'dtype', 'data')}
    line['config'] = c
    pw = roof.get('power') or {}
    line['roofline'] = {'bound': roof['bound'], 'achieved': roof['achieved'], 'peak': roof['peak'], 'unit': roof['unit'], 'frac': roof['frac'],
                        'traffic': roof['traffic'], 'kernel_ms': roof['kernel_ms'], 'copy_GBs': roof['streaming_copy_GBs'],
                        'frac_of_copy': roof['frac_of_streaming_copy'],
                        'power_W': dig(pw, 'convolution', 'power_W'), 'sclk_MHz': dig(pw, 'convolution', 'sclk_MHz'),
                        'sclk_timed_MHz': dig(pw, 'convolution_timed_steps', 'sclk_MHz'),
                        'copy_power_W': dig(pw, 'copy_kernel', 'power_W'), 'copy_sclk_MHz': dig(pw, 'copy_kernel', 'sclk_MHz'), 'cap_W': pw.get('cap_W')}
    if d.get('cpu_baseline'):
        cb = d['cpu_baseline']
        cr = cb.get('c_restatement_Msamples_s') or {}
        line['cpu_baseline'] = {'value': cb['value'], 'unit': cb['unit'], 'cores': cb['cores'], 'kind': cb['kind'], 'sample': cb['sample_short'],
                                'c_port_1_thread': cr.get('threads_1'), 'c_port_all_threads': next((v for k, v in cr.items() if k != 'threads_1'), None),
                                'host_cores': os.cpu_count()}
    return line


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))          # before torch is imported: this process never touches a GPU
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: start the ranks with plain `python bench.py --gpus N` '
                         '(it launches them itself) or with torchrun --nproc-per-node N')
    if 'VND_BENCH_FORCE_DEVICE' in os.environ:                 # rehearsing N ranks on a 1-GPU box
        local_rank = int(os.environ['VND_BENCH_FORCE_DEVICE'])
    os.environ['VND_DEVICE'] = str(local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
    # what the process group itself says: a sum of ones over the ranks (RCCL when the backend is nccl)
    ranks_seen = 1
    if world > 1:
        ones = torch.ones(1, dtype=torch.int64, device=device if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        assert ranks_seen == dist.get_world_size() == world, (ranks_seen, dist.get_world_size(), world)
    power = PowerSampler(torch, local_rank) if (rank == 0 and not args.no_power) else None

    import vndecorrelate_amd.decorrelation as vnd
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays

    ctx = _native.default_context()
    ctx.set_variant(args.variant)
    mode = {'exact': vnd.MODE_EXACT, 'fma': vnd.MODE_FMA, 'fast': vnd.MODE_FAST}[args.mode]

    # ---- shared impulse table: built on rank 0, broadcast over RCCL/xGMI ------
    from vndecorrelate_amd.distributed import broadcast_bytes
    image = None
    if rank == 0:
        fir = vnd.generate_velvet_noise(duration_seconds=FIR_SECONDS, num_impulses=TAPS,
                                        num_outs=CHANNELS, sample_rate_hz=SAMPLE_RATE, seed=1)
        arrays = function_path_arrays(fir)
        image = arrays.to_bytes()
    image = broadcast_bytes(image, src=0, device=device if args.backend == 'nccl' else None)
    table = _native.TapTable.from_bytes(ctx, image)

    # ---- resident synthetic pool: (pool, N, C) float32 in HBM --------------------
    n = SAMPLE_RATE * SECONDS
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    x = torch.empty((args.pool, n, CHANNELS), dtype=torch.float32, device=device)
    x.uniform_(-1.0, 1.0, generator=gen)
    y = torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    samples_per_step = args.pool * n * CHANNELS

    def run(m):
        table.convolve_device(x.data_ptr(), y.data_ptr(), args.pool, n, CHANNELS, m, stream)

    warmups = []

    def timed(m, steps, warmup):
        """warmup, barrier+sync, `steps` back-to-back steps, sync+barrier; wall seconds and the
        mean kernel time between two events on the launch stream."""
        run(m)                           # the first launch of a table may compile its kernel (hipRTC): not warm-up time
        torch.cuda.synchronize()
        done, t_w = 0, time.perf_counter()
        while done < warmup or (time.perf_counter() - t_w) * 1e3 < args.min_warmup_ms:
            run(m)
            done += 1
            if done % 4 == 0:
                torch.cuda.synchronize()
        warmups.append(done)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()                     # same stream the kernels are launched on
        for _ in range(steps):
            run(m)
        ev1.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        stamps.append((t_w, t0, t1))
        return t1 - t0, ev0.elapsed_time(ev1) / steps

    stamps = []
    elapsed, kernel_ms = timed(mode, args.steps, args.warmup)
    timed_ms = elapsed * 1e3
    checked = sorted({0, args.pool // 2, args.pool - 1})      # first, middle, last stream (the last: past 4 GiB of offsets)
    y_timed = {b: y[b].cpu().numpy() for b in checked} if rank == 0 else None

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # spot-check the timed output against the oracle on three streams (rank 0)
    if rank == 0:
        from oracle import c_oracle
        parity = 0.0
        for b in checked:
            xs = x[b].cpu().numpy()
            want = c_oracle.convolve(xs, arrays.tap_offsets, arrays.tap_index, arrays.tap_weight, threads=8)
            if mode == vnd.MODE_EXACT:
                assert np.array_equal(y_timed[b], want), f'bench output differs from the oracle (stream {b})'
            else:
                err = float(np.max(np.abs(y_timed[b].astype(np.float64) - want)) / np.max(np.abs(want)))
                assert err <= 1e-6, f'bench output off by {err:.2e} of peak (stream {b})'
                parity = max(parity, err)

    # the bit-exact mode on the same pool, reported beside the headline (not part of `value`)
    exact_info = None
    if mode != vnd.MODE_EXACT and not args.no_exact:
        e_elapsed, e_kernel_ms = timed(vnd.MODE_EXACT, max(args.steps // 2, 1), 2)
        if rank == 0:
            assert np.array_equal(y[args.pool - 1].cpu().numpy(), want), 'exact mode differs from the oracle'
            exact_board = board_under(torch, power, lambda k: run(vnd.MODE_EXACT))
            # the class path's table (VelvetNoise.convolve: every weight +-1, segment gains; decorrelation.py:393-415) in the
            # same bit-exact mode on the same pool: one packed add per tap, per-table kernel built by default
            cls_info = None
            try:
                cls_table = vnd.VelvetNoise(sample_rate_hz=SAMPLE_RATE, seed=1)._device_table()
                def run_cls():
                    cls_table.convolve_device(x.data_ptr(), y.data_ptr(), args.pool, n, CHANNELS, vnd.MODE_EXACT, stream)
                run_cls(); torch.cuda.synchronize()
                for _ in range(3):
                    run_cls()
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                c0.record()
                reps = max(args.steps // 2, 1)
                for _ in range(reps):
                    run_cls()
                c1.record()
                torch.cuda.synchronize()
                c_ms = c0.elapsed_time(c1) / reps
                cls_board = board_under(torch, power, lambda k: run_cls())
                from oracle import vnd_oracle as O
                taps = O.generate_class_taps(sample_rate_hz=SAMPLE_RATE, seed=1)
                want_cls = O.class_convolve(xs, taps, (0.85, 0.55, 0.35, 0.2), 2)
                assert np.array_equal(y[args.pool - 1].cpu().numpy(), want_cls), 'class-path exact mode differs from the oracle'
                cls_info = {'kernel_ms': round(c_ms, 4),
                            'achieved_GBs': round(ALGO_BYTES_PER_SAMPLE * samples_per_step / (c_ms * 1e-3) / 1e9, 1),
                            'frac_of_8TBs': round(ALGO_BYTES_PER_SAMPLE * samples_per_step / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            'board': cls_board,
                            'launch': cls_table.describe(args.pool, n, CHANNELS, vnd.MODE_EXACT)}
            except Exception as exc:
                cls_info = {'error': repr(exc)}
            exact_info = {'kernel_ms': round(e_kernel_ms, 4),
                          'achieved_GBs': round(ALGO_BYTES_PER_SAMPLE * samples_per_step / (e_kernel_ms * 1e-3) / 1e9, 1),
                          'frac_of_8TBs': round(ALGO_BYTES_PER_SAMPLE * samples_per_step / (e_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          'warmup_actual': warmups[-1], 'board': exact_board,
                          'parity': 'bit-identical to the oracle (sha-checked in tests)',
                          'launch': table.describe(args.pool, n, CHANNELS, vnd.MODE_EXACT),
                          'class_path_table': cls_info}

    # the device's own streaming ceiling on the same pool: a plain device copy (4 B read + 4 B
    # written per sample, the kernel's algorithmic traffic), the honest companion of the 8 TB/s figure
    copy_gbs = stream_copy_gbs = None
    power_info = None
    if rank == 0:
        for _ in range(3):
            y.copy_(x)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        c0.record()
        for _ in range(10):
            y.copy_(x)
        c1.record()
        torch.cuda.synchronize()
        copy_gbs = ALGO_BYTES_PER_SAMPLE * samples_per_step / (c0.elapsed_time(c1) / 10 * 1e-3) / 1e9
        # ... and the best plain copy this library knows (16-byte non-temporal accesses, tools/micro/copy_ceiling.hip)
        stream_ptr = torch.cuda.current_stream().cuda_stream
        ctx.time_copy(x.data_ptr(), y.data_ptr(), x.numel(), 3, stream_ptr)
        stream_copy_ms = ctx.time_copy(x.data_ptr(), y.data_ptr(), x.numel(), 20, stream_ptr)
        stream_copy_gbs = ALGO_BYTES_PER_SAMPLE * samples_per_step / (stream_copy_ms * 1e-3) / 1e9
        # board power and shader clock (the convolution sits on the board's power cap, the copy does not).  The hwmon power
        # figure is a running average that takes a few hundred ms to settle, so beside the readings taken DURING the timed steps each
        # kernel gets 1 s of back-to-back launches after them and reports the median of that run's second half
        if power is not None and power.files:
            t_w, t0, t1 = stamps[0]

            def sustained(fn, seconds=1.0):
                s0, k = time.perf_counter(), 0
                while time.perf_counter() - s0 < seconds:
                    fn(); k += 1
                    if k % 8 == 0:
                        torch.cuda.synchronize()
                torch.cuda.synchronize()
                s1 = time.perf_counter()
                return power.window(s0 + (s1 - s0) / 2, s1)
            power_info = {'convolution_timed_steps': power.window(t0, t1),
                          'convolution': sustained(lambda: run(mode)),
                          'copy_kernel': sustained(lambda: ctx.time_copy(x.data_ptr(), y.data_ptr(), x.numel(), 4, stream_ptr)),
                          'cap_W': power.cap_W, 'card': power.card,
                          'how': f'hwmon power1_input / freq1_input of the card, a reading every {power.period * 1e3:.0f} ms from a thread; medians over the second '
                                 'half of 1 s of back-to-back launches of each kernel after the timed region (the power figure is a slow average), and over '
                                 'the timed steps themselves'}

    launch_text = table.describe(args.pool, n, CHANNELS, mode)
    # what a one-file caller sees (the reference's own use, tests/test_example.py:19-49): ONE device-resident cfg2 signal per launch -
    # never builds a per-table kernel by itself (small launches do not stall for hipRTC), so this is a generic kernel's latency
    single_launch = None
    if rank == 0:
        single_launch = {}
        for label, m in (('fast', vnd.MODE_FAST), ('exact', vnd.MODE_EXACT)):
            lat = []
            for i in range(60):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                table.convolve_device(x[i % min(16, args.pool)].data_ptr(), y[i % min(16, args.pool)].data_ptr(), 1, n, CHANNELS, m, stream)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t0) * 1e6)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(200):
                table.convolve_device(x[i % min(64, args.pool)].data_ptr(), y[i % min(64, args.pool)].data_ptr(), 1, n, CHANNELS, m, stream)
            e1.record()
            torch.cuda.synchronize()
            single_launch[label] = {'launch_to_sync_us_median': round(sorted(lat[10:])[25], 1), 'back_to_back_us': round(e0.elapsed_time(e1) / 200 * 1e3, 2),
                                    'launch': table.describe(1, n, CHANNELS, m)[:90]}
        single_launch['what'] = 'one 10 s stereo signal per launch, device resident: host launch + kernel + synchronise (median of 50), and kernel time back to back over 64 signals'
    del x, y
    torch.cuda.empty_cache()

    strong = None
    if not (args.no_strong or args.no_secondary):           # (--no-secondary: the headline alone)
        strong = cfg4_strong(torch, dist, vnd, _native, ctx, image, mode, world, rank, device, args.backend,
                             taps=(arrays.tap_offsets, arrays.tap_index, arrays.tap_weight) if rank == 0 else None)

    if rank == 0:
        value = world * samples_per_step * args.steps / elapsed / 1e6
        achieved = ALGO_BYTES_PER_SAMPLE * samples_per_step / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes (separate --pmc runs, FETCH_SIZE x 2 +
        # WRITE_SIZE as MI355X_MICROARCH.md prescribes), scaled to this pool when they were taken on
        # the same kernel geometry per stream
        traffic, traffic_source = None, None
        prof = REPO / 'profiles' / 'hbm_traffic.json'
        if prof.exists():
            rec = json.loads(prof.read_text())
            # the counters are a property of ONE kernel and launch geometry: the committed passes count only for a launch that describes
            # itself exactly as the profiled one did (reads_ahead aside: hipRTC's register allocation under rocprofv3 differs by one read,
            # the bytes do not); anything else reports no traffic rather than another kernel's
            strip = lambda t: ' '.join(w for w in t.split() if not w.startswith('reads_ahead='))
            if rec.get('launch') and strip(rec['launch']) == strip(launch_text) and rec.get('bytes_per_stream'):
                traffic = int(rec['bytes_per_stream'] * args.pool)
                traffic_source = (f"profiles/hbm_traffic.json: rocprofv3 --pmc passes of this kernel (source tree {rec.get('commit', 'unrecorded')}) on a pool of "
                                  f"{rec.get('pool')} streams ({rec.get('source')}), scaled per stream - not re-measured in this run")
            else:
                traffic_source = ('profiles/hbm_traffic.json was taken on another launch (' + str(rec.get('launch'))[:120] + ' ...): no traffic figure '
                                  'for this one - re-run tools/profile.sh')
                print('bench.py: profiles/hbm_traffic.json does not describe this launch; roofline.traffic is null', file=sys.stderr)
        detail = {
            'metric': 'Msamples/sec decorrelated (stereo, 30 taps) + achieved HBM GB/s vs roofline',
            'value': round(value, 1), 'unit': 'Msamples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'warmup_actual': warmups[0], 'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'timed_ms': round(timed_ms, 2), 'min_timed_ms_met': timed_ms >= MIN_TIMED_MS,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'cfg2: 48 kHz stereo float32, 10 s, 30 taps / 30 ms velvet FIR (seed 1); '
                                   f'{args.pool} distinct signals resident per GPU, one batched launch per step',
                       'pool_signals_per_gpu': args.pool, 'frames': n, 'channels': CHANNELS,
                       'Mframes_per_s': round(value / CHANNELS, 1),
                       'arithmetic': args.mode, 'parity_vs_oracle_of_peak': (0.0 if mode == vnd.MODE_EXACT else parity),
                       'parity_streams_checked': checked,
                       'launch': launch_text, 'exact_mode': 'top-level key `exact_mode`', 'single_launch_us': single_launch,
                       'world_size': world, 'backend': (args.backend if world > 1 else None), 'ranks_seen_by_all_reduce': ranks_seen,
                       'sharding': 'independent streams per rank; RCCL broadcast of the tap table only'},
            # VND_MODE_EXACT - the drop-in API's DEFAULT arithmetic, bit-identical to the reference - on the same pool, function path
            # (convolve_velvet_noise) and class path (VelvetNoise.convolve): not part of `value`
            'exact_mode': exact_info,
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                         'traffic_source': traffic_source,
                         'read_only_frac': round(achieved / 2 / HBM_PEAK_GBS, 4),
                         'kernel_ms': round(kernel_ms, 4), 'device_copy_GBs': round(copy_gbs, 1),
                         'streaming_copy_GBs': round(stream_copy_gbs, 1),
                         'frac_of_streaming_copy': round(achieved / stream_copy_gbs, 4),
                         'streaming_copy_what': 'a plain copy of the same pool in this run (vnd_time_copy_f32_dev: 16-byte '
                                                'non-temporal loads and stores; device_copy_GBs is torch\'s Tensor.copy_): what a '
                                                'kernel that reads 4 B and writes 4 B per sample can reach on this box',

                         'limit': 'board power cap (`power`: watts and shader clock under this kernel and under the plain copy, this run) on boxes whose copy '
                                  'runs 6.4-6.6 TB/s, HBM itself on boxes whose copy runs 5.6 TB/s (`frac_of_streaming_copy`): DESIGN.md 3.4',
                         'power': power_info,
                         'algorithmic_bytes_per_launch': ALGO_BYTES_PER_SAMPLE * samples_per_step},
        }
        if strong is not None:
            detail['cfg4_strong'] = strong
        if world == 1 and not args.no_secondary:
            detail['secondary'] = secondary_configs(torch, vnd, _native, ctx, mode, power)
            try:
                detail['end_to_end'] = end_to_end(torch, vnd, mode)
            except Exception as exc:                    # as above: never at the headline's expense
                detail['end_to_end'] = {'error': repr(exc)}
            try:
                detail['audio'] = audio_leg(torch, vnd, _native, ctx, args.pool, power)
            except Exception as exc:
                detail['audio'] = {'error': repr(exc)}
            detail['next_rows'] = next_rows(torch, vnd, _native, power)
        if power is not None:
            power.close()
        if world == 1 and not args.no_cpu:
            detail['cpu_baseline'] = cpu_baseline(args.cpu_seconds)
        # the long record: a side file (and gpurun_out/, which travels back from a GPU box) and stderr; the short line: stdout, last
        text = json.dumps(detail, indent=1)
        targets = [pathlib.Path(args.detail)] if args.detail else [REPO / 'bench_detail.json']
        if not args.detail and (REPO / 'gpurun_out').is_dir():
            targets.append(REPO / 'gpurun_out' / 'bench_detail.json')
        for target in targets:
            try:
                target.write_text(text)
            except OSError as exc:
                print(f'bench.py: could not write {target}: {exc}', file=sys.stderr)
        print(json.dumps(detail), file=sys.stderr, flush=True)
        line = compact(detail)
        text = json.dumps(line, separators=(',', ':'))
        if len(text) >= 4096:                            # the driver keeps a tail: shed the least important keys rather than lose the run
            print(f'bench.py: the line grew to {len(text)} bytes; dropping trailing config keys', file=sys.stderr)
            keys = list(line['config'])
            while len(text) >= 4096 and len(keys) > 12:
                line['config'].pop(keys.pop())
                text = json.dumps(line, separators=(',', ':'))
        print(text, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
