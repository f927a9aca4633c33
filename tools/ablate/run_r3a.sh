#!/bin/bash
# round 3: board power and shader clock under the WINDOW form of the per-table kernel (the default for stereo tables in
# fast mode) against the pair-read form (VND_WIN_M=0) and a plain device copy
# (tools/power_probe.py, 3 s back to back per kernel; cfg2 pool of 128 and cfg3 pool of 24)
out=gpurun_out/r3_power.log
: > $out
for cfg in cfg2 cfg3; do
  for k in pair window; do
    echo "== $cfg-$k" >> $out
    if [ $k = pair ]; then export VND_WIN_M=0; else unset VND_WIN_M; fi
    VND_PROBE_CFG=$cfg timeout -k 10 120 python tools/power_probe.py 2 3 >> $out 2>&1 || exit 1
  done
done
unset VND_WIN_M
echo "== copy" >> $out
timeout -k 10 120 python tools/power_probe.py copy 3 >> $out 2>&1
python tools/summarize_power.py $out
