#!/usr/bin/env python3
"""One row per bench run (= per box: every gpurun call lands on a fresh MI355X) from the short stdout lines kept under gpurun_out/:
the headline beside the box's own copy kernel, with the board's power and shader clock under each - and the other claims of the line.
usage: bench_by_box.py gpurun_out/r6_*_bench.out > profiles/r06_bench_by_box.txt"""
import json, sys

rows = []
for path in sys.argv[1:]:
    for text in open(path):
        if text.startswith('{"metric"'):
            rows.append((path, json.loads(text)))
print('# bench.py (--steps 20 --warmup 5) on the boxes this round\'s runs landed on, oldest first; the round\'s final kernels (fast mode: adds per segment).')
print('# Under the convolution every box sits on its 1400 W cap with the clock pulled down; under the plain copy kernel of the same run none does.')
print('#')
print('# run                      headline      copy kernel     conv board      copy board      exact (class)  cfg3 (W, MHz)        cfg5 (W, MHz)        cfg4    audio f/e      f1 256 e/f     f1 128 e   f1 c8 f/e       m2s f/e        N=8 proj       cfg3 parity')
for path, d in rows:
    c, r = d['config'], d['roofline']
    g = lambda k, fmt='{:.3f}': ('-' if c.get(k) is None else fmt.format(c[k]))
    board = lambda w, m: f'{w:.0f} W {m:.0f} MHz' if w is not None and m is not None else '-'
    print(f"{path.split('/')[-1]:24s}  {r['frac']:.4f}        {r['copy_GBs']:.0f} GB/s {r['frac_of_copy']:.3f}  {board(r.get('power_W'), r.get('sclk_MHz')):15s} "
          f"{board(r.get('copy_power_W'), r.get('copy_sclk_MHz')):15s} {g('exact_frac')} ({g('class_exact_frac')})  {g('cfg3_frac')} ({g('cfg3_W', '{:.0f}')}, {g('cfg3_MHz', '{:.0f}')})   "
          f"{g('cfg5_frac')} ({g('cfg5_W', '{:.0f}')}, {g('cfg5_MHz', '{:.0f}')})   {g('cfg4_frac')}   {g('audio_frac')}/{g('audio_exact_frac')}    {g('f1_256_exact_frac')}/{g('f1_256_fast_frac')}    {g('f1_128_exact_frac')}      "
          f"{g('f1_c8_frac')}/{g('f1_c8_exact_frac')}     {g('m2s_frac')}/{g('m2s_exact_frac')}    {g('proj_N8_us', '{:.1f}')} us {g('proj_speedup', '{:.2f}')} x   {g('cfg3_parity', '{:.2e}')}")
