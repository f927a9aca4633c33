#!/usr/bin/env python3
"""Tuning aid: time kernel variants of the velvet-noise tap sum in ONE process,
interleaved rounds (cdna guide rule 24).  Prints GB/s of algorithmic traffic.

    python tools/sweep.py [--config cfg2|cfg3|cfg4|cfg5] [--rounds 5]
"""
import argparse
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))

CONFIGS = {   # fs, fir seconds, taps, kappa, frames, channels, pool
    'cfg2': (48000, 0.03, 30, 1.0, 480000, 2, 128),
    'cfg3': (48000, 0.03, 128, 0.0, 2880000, 2, 24),
    'cfg4': (48000, 0.03, 30, 1.0, 48000, 2, 1024),
    'cfg5': (96000, 0.03, 64, 1.0, 960000, 8, 16),
    'mono': (48000, 0.03, 30, 1.0, 480000, 1, 256),
}


def variant(r_log2=None, dual=None, cg=0, direct=False, nt=0, nostream=0, percu=0):
    v = 0
    if r_log2 is not None:          # frame pairs per lane (the name is historical)
        v |= r_log2
    return v | (cg << 8) | (int(direct) << 12) | ({0: 0, 256: 0, 128: 1, 512: 2, 1024: 3}[nt] << 16) | (nostream << 18) | (percu << 20)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='cfg2')
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--cgs', default='')
    ap.add_argument('--rs', default='2,4,8', help='frame pairs per lane to try')
    ap.add_argument('--modes', default='0,1,2')
    ap.add_argument('--direct', action='store_true')
    ap.add_argument('--nts', default='256')
    ap.add_argument('--stream', default='0', help=argparse.SUPPRESS)
    args = ap.parse_args()
    import torch
    import vndecorrelate_amd.decorrelation as vnd
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays

    fs, dur, taps, kappa, n, ch, pool = CONFIGS[args.config]
    ctx = _native.default_context()
    fir = vnd.generate_velvet_noise(duration_seconds=dur, num_impulses=taps, num_outs=ch,
                                    sample_rate_hz=fs, log_distribution_strength=kappa, seed=1)
    arr = function_path_arrays(fir)
    table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
    x = torch.empty((pool, n, ch), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y = torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    nbytes = 8 * x.numel()
    cgs = [int(c) for c in args.cgs.split(',')] if args.cgs else ([2] if ch % 2 == 0 else [1])
    cases = []
    for mode in [int(m) for m in args.modes.split(',')]:
        for cg in cgs:
            for r in [int(v) for v in args.rs.split(',')]:
                for dual in (0,):
                    for nt in ([int(v) for v in args.nts.split(',')] if mode == 2 else [0]):
                        for sv in ([int(v) for v in args.stream.split(',')] if mode == 2 else [0]):
                            cases.append((mode, cg, r, dual, False, nt, sv))
        if args.direct:
            cases.append((mode, 0, None, None, True, 0, 0))
    results = {c: [] for c in cases}
    for rnd in range(args.rounds + 1):
        for c in cases:
            mode, cg, r, dual, direct, nt, sv = c
            ctx.set_variant(variant(r, dual, cg, direct, nt, int(sv == 0), max(sv, 0)))
            iters = 2 if direct else args.iters
            ms = table.time_device(x.data_ptr(), y.data_ptr(), pool, n, ch, mode=mode, n_buffers=1,
                                   stride_elems=0, iters=iters, stream=stream)
            if rnd:
                results[c].append(ms)
    print(f'# {args.config}: pool={pool} n={n} C={ch} taps={taps} algorithmic bytes/launch={nbytes/1e6:.1f} MB')
    print('mode cg pairs dual direct   med_ms   min_ms   GB/s(med)  frac_of_8TB/s  launch')
    for c in cases:
        mode, cg, r, dual, direct, nt, sv = c
        ctx.set_variant(variant(r, dual, cg, direct, nt, int(sv == 0), max(sv, 0)))
        desc = table.describe(pool, n, ch, mode)
        med, mn = float(np.median(results[c])), float(np.min(results[c]))
        gbs = nbytes / med / 1e6
        print(f'{("exact", "fma  ", "fast ")[mode]} {cg:2d} {r if r is not None else 0:5d} '
              f'{dual if dual is not None else "-":>4} {int(direct):6d} {med:8.4f} {mn:8.4f} '
              f'{gbs:10.1f} {gbs / 8000:10.4f}   {desc}')
    ctx.set_variant(-1)


if __name__ == '__main__':
    main()
