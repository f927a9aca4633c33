#!/usr/bin/env python3
"""Condense tools/power_probe.py logs: per section the launch time and, for the card whose
power moves, the board power and shader clock while the kernel runs."""
import ast, re, sys
sec, runs, rates = None, {}, {}
for line in open(sys.argv[1]):
    if line.startswith('=='):
        sec = line.strip('= \n'); runs[sec] = []; rates[sec] = []; continue
    m = re.match(r'(RUN |idle)\s+(-?[\d.]+) (\{.*\})', line)
    if m and m.group(1) == 'RUN ':
        runs[sec].append(ast.literal_eval(m.group(3)))
    m = re.match(r't=\s*([\d.]+)s\s+([\d.]+) ms', line)
    if m and float(m.group(1)) > 0.5:
        rates[sec].append(float(m.group(2)))
for sec, samples in runs.items():
    if not samples:
        print(sec, 'no samples'); continue
    cards = sorted({k.split(':')[0] for k in samples[0]})
    hot = max(cards, key=lambda c: max(s.get(c + ':power1_input', 0) for s in samples))
    pw = [s[hot + ':power1_input'] / 1e6 for s in samples[1:]] or [samples[0][hot + ':power1_input'] / 1e6]
    fq = [s[hot + ':freq1_input'] / 1e6 for s in samples[1:]] or [samples[0][hot + ':freq1_input'] / 1e6]
    ms = sum(rates[sec]) / max(1, len(rates[sec]))
    print(f'{sec:10s} {ms:.4f} ms/launch  {983.04 / ms:5.0f} GB/s(cfg2 bytes)   power {sum(pw) / len(pw):6.0f} W   sclk {sum(fq) / len(fq):5.0f} MHz  ({hot})')
