#!/bin/bash
# board power and shader clock under back-to-back launches of cfg2 (pool of 128), cfg3 and cfg5, fast and exact: which configs sit on the 1400 W cap
for cfg in cfg2 cfg3 cfg5; do for mode in 2 0; do
  echo "== $cfg mode $mode"; VND_PROBE_CFG=$cfg timeout -k 10 120 python tools/power_probe.py $mode 4 2>&1 | grep -v amdgpu.ids | grep '^conv\|^t=\|^RUN\|sclk clock' | cut -c1-2000
done; done
