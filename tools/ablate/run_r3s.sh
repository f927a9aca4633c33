#!/bin/bash
# round 3 (last hours): board power and shader clock under the window form with 64-frame runs and the waves split over the two
# channels (the default now) against the plain 32-frame form (VND_WIN_SPLIT=0), the pair-read form (VND_WIN_M=0) and a plain copy
# (tools/power_probe.py, 3 s back to back per kernel; cfg2 pool of 128 and cfg3 pool of 24; fast mode, then exact for cfg2)
out=gpurun_out/r3s_power.log
: > $out
for cfg in cfg2 cfg3; do
  for k in pair plain32 split64; do
    echo "== $cfg-$k" >> $out
    unset VND_WIN_M VND_WIN_SPLIT
    [ $k = pair ] && export VND_WIN_M=0
    [ $k = plain32 ] && export VND_WIN_SPLIT=0
    VND_PROBE_CFG=$cfg timeout -k 10 120 python tools/power_probe.py 2 3 >> $out 2>&1 || exit 1
  done
done
for k in plain32 split64; do
  echo "== cfg2exact-$k" >> $out
  unset VND_WIN_M VND_WIN_SPLIT
  [ $k = plain32 ] && export VND_WIN_SPLIT=0
  VND_PROBE_CFG=cfg2 timeout -k 10 120 python tools/power_probe.py 0 3 >> $out 2>&1 || exit 1
done
unset VND_WIN_M VND_WIN_SPLIT
echo "== copy" >> $out
timeout -k 10 120 python tools/power_probe.py copy 3 >> $out 2>&1
python tools/summarize_power.py $out
