#!/bin/bash
# round 3: board power and shader clock under the WINDOW form of the per-table kernel against the pair-read form
# (tools/power_probe.py, 3 s back to back per kernel; cfg2 pool of 128 and cfg3 pool of 24)
out=gpurun_out/r3_power_a.log
: > $out
for cfg in cfg2 cfg3; do
  for k in pair win32; do
    echo "== $cfg-$k" >> $out
    if [ $k = win32 ]; then export VND_WIN_M=32 VND_SPEC_NT=256; else unset VND_WIN_M VND_SPEC_NT; fi
    VND_PROBE_CFG=$cfg timeout -k 10 120 python tools/power_probe.py 2 3 >> $out 2>&1 || exit 1
  done
done
unset VND_WIN_M VND_SPEC_NT
echo "== copy" >> $out
timeout -k 10 120 python tools/power_probe.py copy 3 >> $out 2>&1
python tools/summarize_power.py $out
